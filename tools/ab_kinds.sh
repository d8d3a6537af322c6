#!/bin/bash
# On the GPU box: serialised per-kind times of a Flan-class factorization with diagnostic builds of the library
# (results may be wrong: ablations).  tools/ab_kinds.sh OUT NAME [NAME ...]   ('-' = the product)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$1; shift
cd $R
for name in "$@"; do
  if [ "$name" = "-" ]; then unset PARSY_LIB; else export PARSY_LIB=$R/tools/libparsy_$name.bin; fi
  timeout -k 10 280 python3 tools/kinds.py flan >> $out 2>&1 || echo "$name failed" >> $out
done
cat $out
