"""Diagnostic: run a few factorizations with the stamped build (tools/build_stamps.sh) and print, for
the last supernode, the per-block-column timeline of the chain: diagonal tile (J,J) and tile (J+1,J)."""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import _native as N
N._LIB_PATH = ROOT / "tools" / "libparsy_stamps.bin"
from parsy_bench_amd import api, inspector as I, matrices as M
A, perm = M.workload(sys.argv[1] if len(sys.argv) > 1 else "nd24k")
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
for _ in range(3):
    lv, sec = plan.factor(sym.A2x)
print("factor ms", sec * 1e3, "status", plan.status())
tr = np.zeros(16 * 512, dtype=np.uint64)
N.lib().parsy_debug_trace(tr.ctypes.data_as(C.c_void_p))
t = tr.reshape(512, 16).astype(np.float64) / 100.0  # us
w = int(np.diff(sym.super)[-1]); nbc = (w + 63) // 64
t0 = t[0, 0]
print("root: %d block columns; times in us relative to the start of diag tile 0" % nbc)
print("  J | diag: start  stream_end  potrf_end  published | below: start  saw_flag  trsm_end  published | step")
prev = None
for J in range(nbc):
    d = t[J]
    step = (d[3] - prev) if prev is not None else 0.0
    print("%3d | %8.1f %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f %8.1f | %6.1f" % (
        J, d[0] - t0, d[1] - t0, d[2] - t0, d[3] - t0, d[7] - t0, d[4] - t0, d[5] - t0, d[6] - t0, step))
    prev = d[3]
    if J > 0:
        pb = t[J - 1][6]
        print("      diag wave0 last entry: below published %.1f | first poll miss %.1f  flag seen %.1f  acquired %.1f  round(more) %.1f  last round %.1f  | stream_end %.1f" % (
            pb - t0, d[8] - t0, d[9] - t0, d[10] - t0, d[11] - t0, d[12] - t0, d[1] - t0))
