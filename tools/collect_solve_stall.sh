#!/bin/bash
# On the GPU box: rocprofv3 --pmc passes over the solves of a workload (one group per run) ->
# gpurun_out/r05/sstall_<tag>_<group>.json.  Usage: collect_solve_stall.sh TAG WORKLOAD NRHS
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-x}
W=${2:-parabolic_fem}
NR=${3:-64}
mkdir -p $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
run() {  # group name, counters...
    g=$1; shift
    rm -rf /tmp/sstall_$g
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/sstall_$g -o p -- python3 $R/tools/one_factor.py $W 1 2 $NR > $R/gpurun_out/r05/sstall_${T}_$g.log 2>&1 || return 1
    python3 $R/tools/pmc_summary.py /tmp/sstall_$g/p_counter_collection.csv > $R/gpurun_out/r05/sstall_${T}_$g.json
}
run g1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE &&
run g2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_IFETCH &&
run g3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU &&
run g4 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
if [ -n "$4" ]; then
run m1 TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run m2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum
run m3 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum TCC_READ_sum TCC_WRITE_sum
fi
