#!/bin/bash
# On the GPU box: serialised per-kind times of a Flan-class factorization under different environments.
# tools/ab_env.sh OUT "VAR=V [VAR=V ...]" ...      ('-' = nothing set)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$1; shift
cd $R
for envs in "$@"; do
  echo "== $envs" >> $out
  if [ "$envs" = "-" ]; then
    timeout -k 10 280 python3 tools/kinds.py flan >> $out 2>&1 || echo "failed" >> $out
  else
    ( export $envs; timeout -k 10 280 python3 tools/kinds.py flan >> $out 2>&1 ) || echo "failed" >> $out
  fi
done
cat $out
