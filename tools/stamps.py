"""Diagnostic: run one factorization with the stamped build and print phase times (us)
of the last PANEL workgroup 0 and of the last diagonal-tile workgroup."""
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
st = np.zeros(32, dtype=np.uint64)
N.lib().parsy_debug_stamps(st.ctypes.data_as(C.c_void_p))
t = st.astype(np.float64) / 100.0  # 100 MHz -> us
print("factor ms", sec * 1e3)
print("PANEL wg0: load %.2f  trsm %.2f  store %.2f us" % (t[1] - t[0], t[2] - t[1], t[3] - t[2]))
print("diag tile: update %.2f  writeback+sync %.2f  gather %.2f  potrf %.2f  park %.2f us" % (
    t[9] - t[8], t[10] - t[9], t[11] - t[10], t[12] - t[11], t[13] - t[12]))
print("root tile(0,0) thread0: entries %d chunks %d | store %.1f issue %.1f consume %.1f barrier %.1f | loop %.1f us" % (
    st[22], st[21], st[16] / 100.0, st[17] / 100.0, st[18] / 100.0, st[19] / 100.0, st[20] / 100.0))
print("tile below diag: wait %.2f  load Dg %.2f  trsm %.2f  store %.2f us;  diag publish->(wait end) %.2f" % (
    t[25] - t[24], t[26] - t[25], t[27] - t[26], t[28] - t[27], t[25] - t[13]))
