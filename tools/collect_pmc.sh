#!/bin/bash
# On the GPU box: the four rocprofv3 --pmc passes (calibration + bench.py, FETCH_SIZE and WRITE_SIZE in
# separate runs: they do not fit one pass) and the per-launch HBM traffic of the tile kernel.
# Usage: bash tools/collect_pmc.sh OUT.json   (writes the raw csv under gpurun_out/pmc_*)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/pmc_traffic.json}
NF=4  # factorizations in the profiled bench run: 1 warm-up + 3 steps
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cf -o cf -- $R/tools/pmc_calib.bin > $R/gpurun_out/pmc_calib.json 2> $R/gpurun_out/pmc_cf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cw -o cw -- $R/tools/pmc_calib.bin > /dev/null 2> $R/gpurun_out/pmc_cw.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bf -o bf -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $R/gpurun_out/pmc_bf.json 2> $R/gpurun_out/pmc_bf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bw -o bw -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > $R/gpurun_out/pmc_bw.json 2> $R/gpurun_out/pmc_bw.err
cd $R
python3 tools/pmc_traffic.py gpurun_out/pmc_calib.json gpurun_out/pmc_cf/cf_counter_collection.csv gpurun_out/pmc_cw/cw_counter_collection.csv gpurun_out/pmc_bf/bf_counter_collection.csv gpurun_out/pmc_bw/bw_counter_collection.csv $NF $OUT
