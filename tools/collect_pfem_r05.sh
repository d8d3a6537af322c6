#!/bin/bash
# On the GPU box: the many-right-hand-side solves of the parabolic_fem-class input (configs[3]) before / after round 5 ->
# gpurun_out/r05/: kernel timelines (64 and 8 right-hand sides; PARSY_SUB_MRHS_MIN=0 = the level kernels' subtree form, no
# bands: round 4) and WRITE_SIZE per kernel (rocprofv3 --pmc, its own pass: the bytes of the atomics + stores).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
bash $R/tools/solve_tl.sh parabolic_fem 64 pfem64_new
bash $R/tools/solve_tl.sh parabolic_fem 64 pfem64_old PARSY_SUB_MRHS_MIN=0
bash $R/tools/solve_tl.sh parabolic_fem 8 pfem8_new
bash $R/tools/solve_tl.sh parabolic_fem 8 pfem8_old PARSY_SUB_MRHS_MIN=0
cd /tmp && export TMPDIR=/tmp
for tag in new old; do
  if [ $tag = old ]; then export PARSY_SUB_MRHS_MIN=0; else unset PARSY_SUB_MRHS_MIN; fi
  rm -rf /tmp/pfw_$tag
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pfw_$tag -o p -- python3 $R/tools/one_factor.py parabolic_fem 1 2 64 > $O/pfem_write_$tag.log 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pfw_$tag/p_counter_collection.csv > $O/pfem_write_$tag.json
done
