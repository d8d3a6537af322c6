#!/bin/bash
# On the GPU box: bench.py (Flan-class, short) with the product library and with diagnostic builds of it, one after
# the other: tools/ab_lib.sh OUT NAME [NAME ...]  (tools/libparsy_NAME.bin; '-' = the product)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$1; shift
cd $R
for name in "$@"; do
  if [ "$name" = "-" ]; then unset PARSY_LIB; else export PARSY_LIB=$R/tools/libparsy_$name.bin; fi
  for rep in 1 2; do
    timeout -k 10 280 python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), {k: round(v,1) for k,v in d['roofline']['kind_ms_per_factorization_serialized'].items()})" >> $out || exit 1
  done
done
cat $out
