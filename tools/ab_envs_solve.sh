#!/bin/bash
# On the GPU box: tools/one_factor.py WORKLOAD 1 3 NRHS under each of the given environments ('-' = nothing set).
# tools/ab_envs_solve.sh OUT WORKLOAD NRHS "VAR=V ..." ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$1; W=$2; Q=$3; shift 3
cd $R
for envs in "$@"; do
  echo "== $envs $W nrhs=$Q" >> $out
  if [ "$envs" = "-" ]; then timeout -k 10 280 python3 tools/one_factor.py $W 1 3 $Q 2>&1 | grep -v amdgpu.ids >> $out
  else ( export $envs; timeout -k 10 280 python3 tools/one_factor.py $W 1 3 $Q 2>&1 | grep -v amdgpu.ids >> $out ); fi
done
