R=${GRAFT_REPO_ROOT}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_l2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc_l2 -o p -- python3 $R/tools/one_factor.py flan 1 > $R/gpurun_out/r04/l2disp.log 2>&1 || exit 1
python3 $R/tools/pmc_per_dispatch.py /tmp/pmc_l2/p_counter_collection.csv k_chol_dense > $R/gpurun_out/r04/l2_dense_per_dispatch.txt
python3 $R/tools/pmc_per_dispatch.py /tmp/pmc_l2/p_counter_collection.csv k_chol_big > $R/gpurun_out/r04/l2_big_per_dispatch.txt
