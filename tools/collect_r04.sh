#!/bin/bash
# On the GPU box: the evidence files of round 4 for one workload (default flan) -> gpurun_out/r04/ :
#   kernel_stats.csv     rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1` (no CPU legs)
#   pmc_mfma.json        SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_BUSY_CYCLES, ... per kernel
#   pmc_traffic.json     FETCH_SIZE / WRITE_SIZE per kernel, calibrated (tools/pmc_traffic2.py)
#   pmc_l2.json          TCC_HIT_sum / TCC_MISS_sum per kernel
# One counter group per run (FETCH_SIZE and WRITE_SIZE do not fit one pass); counters never together with the
# hip/hsa trace domains.  The profiled program is tools/one_factor.py <workload> 2 2: 2 factorizations, 2 forward
# and 2 backward solves.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-flan}
O=$R/gpurun_out/${OUTDIR:-r04}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
H=$(python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_hash())")
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_stats -o st -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/stats_bench.json 2> $O/stats_bench.err
cp /tmp/r04_stats/st_kernel_stats.csv $O/${W}_kernel_stats.csv
# the same command with every launch on one stream (as in bench.py's profiled steps: roofline.avg_launch_ms)
PARSY_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_stats_s -o st -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/stats_bench_serialized.json 2> $O/stats_bench_serialized.err
cp /tmp/r04_stats_s/st_kernel_stats.csv $O/${W}_kernel_stats_serialized.csv
pmc() {  # group name, program..., -- counters
    g=$1; shift
    rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/r04_$g -o p -- "$@" > $O/pmc_$g.log 2>&1
    python3 $R/tools/pmc_summary.py /tmp/r04_$g/p_counter_collection.csv > $O/pmc_$g.json
}
PMC="FETCH_SIZE" pmc calib_fetch $R/tools/pmc_calib.bin
tail -1 $O/pmc_calib_fetch.log > /dev/null
$R/tools/pmc_calib.bin > $O/calib_stdout.json
PMC="WRITE_SIZE" pmc calib_write $R/tools/pmc_calib.bin
PMC="FETCH_SIZE" pmc fetch python3 $R/tools/one_factor.py $W 2 2
PMC="WRITE_SIZE" pmc write python3 $R/tools/one_factor.py $W 2 2
PMC="TCC_HIT_sum TCC_MISS_sum" pmc l2 python3 $R/tools/one_factor.py $W 2 2
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" pmc mfma python3 $R/tools/one_factor.py $W 2 2
python3 $R/tools/pmc_traffic2.py $O/calib_stdout.json $O/pmc_calib_fetch.json $O/pmc_calib_write.json $O/pmc_fetch.json $O/pmc_write.json $W $H $O/${W}_pmc_traffic.json > /dev/null
cp $O/pmc_mfma.json $O/${W}_pmc_mfma.json
cp $O/pmc_l2.json $O/${W}_pmc_l2.json
ls -la $O
