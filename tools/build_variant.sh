#!/bin/bash
# Diagnostic build of the library with extra compiler flags: tools/build_variant.sh NAME -DFLAG[=V] ...
# -> tools/libparsy_NAME.bin (load it with PARSY_LIB=...).  Never the product library.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python3 - "$name" "$@" <<'PY'
import sys
from pathlib import Path
from parsy_bench_amd.build import build_native
name, flags = sys.argv[1], sys.argv[2:]
print(build_native(force=True, extra_flags=flags, out=Path(f"tools/libparsy_{name}.bin").resolve(),
                   objdir=Path(f"tools/build_var_{name}").resolve()))
PY
