#!/bin/bash
# On the GPU box: kernel timeline of the last of 3 factorizations (rocprofv3 --kernel-trace) -> gpurun_out/timeline*.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-nd24k}
TAG=${2:-}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl$TAG -o tl -- python3 $R/tools/one_factor.py $W 3 > $R/gpurun_out/tl$TAG.log 2>&1
cd $R && python3 tools/timeline.py gpurun_out/tl$TAG/tl_kernel_trace.csv > gpurun_out/timeline$TAG.txt
