#!/bin/bash
# On the GPU box: schedule knobs on the nd24k-class input with the round-4 kernels (overlapped ms + serialised ms per kind)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/r04/sweep_nd24k.txt
rm -f $out
cd $R
run() { echo "$*" >> $out; env "$@" timeout -k 10 200 python3 tools/kinds.py nd24k 2>&1 | grep -v amdgpu.ids >> $out; }
run PARSY_X=0
run PARSY_BIG_MINK=128
run PARSY_BIG_MINK=128 PARSY_PIECE_WIDTH=512
run PARSY_BIG_MINK=128 PARSY_PIECE_WIDTH=256
run PARSY_BIG_MINK=64 PARSY_PIECE_WIDTH=256
run PARSY_BIG_MINK=128 PARSY_PIECE_WIDTH=384 PARSY_CHAIN_SPLIT=0
run PARSY_BIG_MINK=128 PARSY_PIECE_WIDTH=512 PARSY_CHAIN_SPLIT=2
cat $out
