#!/bin/bash
# Diagnostic build with in-kernel phase stamps (never the product library).
set -e
cd "$(dirname "$0")/.."
python3 - <<'PY'
from pathlib import Path
from parsy_bench_amd.build import build_native
print(build_native(force=True, extra_flags=["-DPARSY_STAMPS"], out=Path("tools/libparsy_stamps.bin").resolve(),
                   objdir=Path("tools/build_stamps").resolve()))
PY
