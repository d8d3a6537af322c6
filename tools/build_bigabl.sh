#!/bin/bash
# Diagnostic builds of the library with one phase of k_chol_big removed (results are wrong by
# construction; only kernel times are read).  Never the product library.
set -e
cd "$(dirname "$0")/.."
for v in "$@"; do
python3 - <<PY
from pathlib import Path
from parsy_bench_amd.build import build_native
print(build_native(force=True, extra_flags=["-DPARSY_BIGABL_$v"], out=Path("tools/libparsy_abl_$v.bin").resolve(),
                   objdir=Path("tools/build_abl_$v").resolve()))
PY
done
