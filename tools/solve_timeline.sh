R=${GRAFT_REPO_ROOT}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl_solve
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_solve -o tl -- python3 $R/tools/one_factor.py flan 1 2 1 > $R/gpurun_out/r04/tl_solve.log 2>&1
cd $R && python3 tools/timeline.py gpurun_out/tl_solve/tl_kernel_trace.csv solve > gpurun_out/r04/flan_solve_timeline.txt
cat gpurun_out/r04/flan_solve_timeline.txt
