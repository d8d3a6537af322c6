# kernel timeline of the solves of a workload: solve_tl.sh WORKLOAD NRHS OUTNAME [env assignments...]
R=${GRAFT_REPO_ROOT}
W=$1; NR=$2; OUT=$3; shift 3
for kv in "$@"; do export "$kv"; done
mkdir -p $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl_$OUT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$OUT -o tl -- python3 $R/tools/one_factor.py $W 1 2 $NR > $R/gpurun_out/r05/tl_$OUT.log 2>&1
cd $R && python3 tools/timeline.py gpurun_out/tl_$OUT/tl_kernel_trace.csv solve > gpurun_out/r05/${OUT}_timeline.txt
rm -rf $R/gpurun_out/tl_$OUT
