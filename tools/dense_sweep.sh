#!/bin/bash
# On the GPU box: the thresholds of the dense / ragged split (schedule.cpp) swept on the Flan-class input:
# overlapped ms of a factorization + serialised time per kind.   tools/dense_sweep.sh OUT
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$1
cd $R
run() { echo "$*" >> $out; env "$@" timeout -k 10 200 python3 tools/kinds.py flan 2>&1 | grep -v amdgpu.ids >> $out; }
run PARSY_BIG_DENSE=1
run PARSY_DENSE_FILL=85
run PARSY_DENSE_FILL=70
run PARSY_DENSE_FILL=55
run PARSY_DENSE_MIN_SHARE=50
run PARSY_DENSE_MIN_SHARE=70
run PARSY_DENSE_ALL_SHARE=15
run PARSY_DENSE_ALL_SHARE=30
run PARSY_DENSE_FILL=70 PARSY_DENSE_ALL_SHARE=15
cat $out
