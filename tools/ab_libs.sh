#!/bin/bash
# On the GPU box: tools/one_factor.py WORKLOAD 1 3 NRHS under each of the given libraries ('-' = the product).
# tools/ab_libs.sh OUT WORKLOAD NRHS LIB...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$1; W=$2; Q=$3; shift 3
cd $R
for lib in "$@"; do
  echo "== lib=$lib $W nrhs=$Q" >> $out
  ( [ "$lib" != "-" ] && export PARSY_LIB=$R/$lib; timeout -k 10 280 python3 tools/one_factor.py $W 1 3 $Q 2>&1 | grep -v amdgpu.ids >> $out ) || echo failed >> $out
done
cat $out
