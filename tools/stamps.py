"""Diagnostic: run a few factorizations with the stamped build (tools/build_stamps.sh) and print, for
the last supernode, the per-block-column timeline of the chain: diagonal tile (J,J) and tile (J+1,J)."""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import _native as N
import os
N._LIB_PATH = Path(os.environ.get("PARSY_STAMPS_LIB", str(ROOT / "tools" / "libparsy_stamps.bin")))
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
print("root: %d block columns; walker times in us relative to its first POTRF start" % nbc)
print("  J |   start  potrf_end  prep_seen  diag_pub  trsm_end  syrk_end | step  prefetched")
prev = None
for J in range(nbc):
    d = t[J]
    step = (d[0] - prev) if prev is not None else 0.0
    print("%3d | %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f | %6.1f  %d" % (
        J, d[0] - t0, d[2] - t0, d[4] - t0, d[3] - t0, d[5] - t0, d[6] - t0, step, int(tr.reshape(512, 16)[J, 7])))
    prev = d[0]
    if J in (3, 4, 9):
        print("      after potrf: +%.1f prep seen, +%.1f diag published, +%.1f inverted, +%.1f solved, +%.1f X stored, +%.1f half SYRK + X flag, +%.1f SYRK products, +%.1f subtracted + barrier" % (
            d[4] - d[2], d[3] - d[4], d[10] - d[3], d[5] - d[10], d[11] - d[5], d[12] - d[11], d[13] - d[12], d[6] - d[13]))
    if J == 5:
        raw = tr.reshape(512, 16)[J]
        cyc = float(raw[9]) - float(raw[8])
        print("      POTRF: %.0f shader clocks in %.2f us -> %.2f GHz" % (cyc, d[2] - d[0], cyc / (d[2] - d[0]) / 1e3))
pb = np.zeros(8, dtype=np.uint64)
N.lib().parsy_debug_probe(pb.ctypes.data_as(C.c_void_p))
q = pb.astype(np.float64)
print("POTRF step 4, thread (6,4) [block column 4 = the factoring column], shader clocks: write+barrier1 %.0f  factor+solve %.0f  barrier2 %.0f  update %.0f" % (
    q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3]))
print("SMALL kernel, workgroup 0 of a launch, by number of updates: w r | load  updates  potrf  trsm  store (us)")
T = tr.reshape(512, 16)
for nu in range(0, 64):
    row = T[256 + nu]
    if row[5] == 0:
        continue
    x = row[:6].astype(np.float64) / 100.0
    print("  nupd %2d: w %2d r %3d | %5.1f %6.1f %6.1f %6.1f %5.1f" % (nu, int(row[8]), int(row[9]), x[1]-x[0], x[2]-x[1], x[3]-x[2], x[4]-x[3], x[5]-x[4]))
