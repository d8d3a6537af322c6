#!/bin/bash
# On the GPU box: rocprofv3 --pmc passes that say what the waves of k_chol_big wait for (one group per run, 2
# factorizations each) -> gpurun_out/stall_<tag>_<group>.json.  Usage: collect_stall.sh TAG [WORKLOAD]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-x}
W=${2:-flan}
cd /tmp && export TMPDIR=/tmp
run() {  # group name, counters...
    g=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/stall_$g -o p -- python3 $R/tools/one_factor.py $W 2 > $R/gpurun_out/stall_${T}_$g.log 2>&1 || return 1
    python3 $R/tools/pmc_summary.py /tmp/stall_$g/p_counter_collection.csv > $R/gpurun_out/stall_${T}_$g.json
}
run g1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE &&
run g2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_IFETCH &&
run g3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_MFMA &&
run g4 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL
