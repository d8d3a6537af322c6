# On the GPU box: HBM fetch and L2 hit counters of one Flan-class factorization, per kernel (two passes).
R=${GRAFT_REPO_ROOT}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  n=$(echo $g | tr ' ' '_')
  rm -rf /tmp/pmc_$n
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmc_$n -o p -- python3 $R/tools/one_factor.py flan 1 > $O/pmc_bigk_$n.log 2>&1 || { echo "pass $n failed"; tail -5 $O/pmc_bigk_$n.log; continue; }
  python3 $R/tools/pmc_summary.py /tmp/pmc_$n/p_counter_collection.csv > $O/pmc_bigk_$n.json
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/pmc_bigk_*.json")):
    d=json.load(open(f))
    print(f.split("/")[-1])
    for k,v in d.items():
        if any(s in k for s in ("k_chol_big","k_chol_dense")): print("  ",k,v)
PY
