#!/bin/bash
# On the GPU box: L2 hit / miss and FETCH_SIZE of 2 factorizations under the environment of the caller
# -> gpurun_out/pmc_<workload>_<tag>_{l2,fetch}.json      usage: collect_l2.sh WORKLOAD TAG
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-flan}
T=${2:-x}
cd /tmp && export TMPDIR=/tmp
run() {  # group name, counters...
    g=$1; shift
    rm -rf /tmp/pmc_$g
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$g -o p -- python3 $R/tools/one_factor.py $W 2 > $R/gpurun_out/pmc_${W}_${T}_$g.log 2>&1 || return 1
    python3 $R/tools/pmc_summary.py /tmp/pmc_$g/p_counter_collection.csv > $R/gpurun_out/pmc_${W}_${T}_$g.json
}
run l2 TCC_HIT_sum TCC_MISS_sum &&
run fetch FETCH_SIZE
