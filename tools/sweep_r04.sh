#!/bin/bash
# On the GPU box: schedule knobs re-swept with the round-4 kernels (Flan-class: overlapped ms + serialised ms per kind)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/r04/sweep_super.txt
rm -f $out
cd $R
run() { echo "$*" >> $out; env "$@" timeout -k 10 200 python3 tools/kinds.py flan 2>&1 | grep -v amdgpu.ids >> $out; }
run PARSY_BIG_SUPER_MIN=12288
run PARSY_BIG_SUPER_MIN=8192
run PARSY_BIG_SUPER_MIN=6144
run PARSY_BIG_SUPER_MIN=4096
run PARSY_BIG_SUPER_MIN=2048
run PARSY_BIG_SUPER_MIN=6144 PARSY_BIG_SUPER_FILL=95
run PARSY_BIG_SUPER_MIN=4096 PARSY_DENSE_FILL=60
cat $out
