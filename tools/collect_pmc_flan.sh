#!/bin/bash
# On the GPU box: rocprofv3 --pmc passes (one counter group per run) over 2 factorizations of a workload
# (tools/one_factor.py), summed per kernel -> gpurun_out/pmc_<workload>_<group>.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-flan}
cd /tmp && export TMPDIR=/tmp
run() {  # group name, counters...
    g=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$g -o p -- python3 $R/tools/one_factor.py $W 2 > $R/gpurun_out/pmc_${W}_$g.log 2>&1 || return 1
    python3 $R/tools/pmc_summary.py /tmp/pmc_$g/p_counter_collection.csv > $R/gpurun_out/pmc_${W}_$g.json
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run l2 TCC_HIT_sum TCC_MISS_sum
