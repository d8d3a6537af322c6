#!/bin/bash
# On the GPU box: shader clock during each kernel of one Flan-class factorization -> gpurun_out/r04/clock_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-x}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_clk
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_clk -o p -- python3 $R/tools/one_factor.py flan 2 > $R/gpurun_out/r04/clock_$T.log 2>&1 || exit 1
python3 $R/tools/clock_pmc.py /tmp/pmc_clk/p_counter_collection.csv /tmp/pmc_clk/p_kernel_trace.csv > $R/gpurun_out/r04/clock_$T.txt
cat $R/gpurun_out/r04/clock_$T.txt
