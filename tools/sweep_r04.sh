#!/bin/bash
# On the GPU box: schedule knobs re-swept with the round-4 kernels (Flan-class: overlapped ms + serialised ms per kind)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/r04/sweep_piece.txt
rm -f $out
cd $R
run() { echo "$*" >> $out; env "$@" timeout -k 10 200 python3 tools/kinds.py flan 2>&1 | grep -v amdgpu.ids >> $out; }
run PARSY_PIECE_WIDTH=512
run PARSY_PIECE_WIDTH=768
run PARSY_PIECE_WIDTH=896
run PARSY_PIECE_WIDTH=1024
run PARSY_PIECE_WIDTH=768 PARSY_DENSE_ALL_SHARE=12
run PARSY_PIECE_WIDTH=768 PARSY_BIG_MINK=64
run PARSY_PIECE_WIDTH=512
run PARSY_PIECE_WIDTH=768
cat $out
