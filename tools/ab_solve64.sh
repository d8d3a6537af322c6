#!/bin/bash
# On the GPU box: forward / backward solves of a workload with NRHS right-hand sides, product against tools/libparsy_prev.bin
# (PARSY_LIB), alternating.  tools/ab_solve64.sh OUT WORKLOAD NRHS
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$1; W=${2:-flan}; Q=${3:-64}
cd $R
for rep in 1 2; do
  for lib in "" "$R/tools/libparsy_prev.bin"; do
    echo "== lib=${lib:-product} $W nrhs=$Q" >> $out
    ( [ -n "$lib" ] && export PARSY_LIB=$lib; timeout -k 10 280 python3 tools/one_factor.py $W 1 4 $Q 2>&1 | grep -v amdgpu.ids >> $out ) || echo failed >> $out
  done
done
cat $out
