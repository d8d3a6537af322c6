"""Host-side (no GPU): the ragged entries of the BIG launches (what k_chol_big multiplies) by what they are next to --
a row remainder BEHIND a dense block of the same source and column window in the same task (same src, ja, nj, K;
ia = the dense block's ia + 128), a column remainder BESIDE one (same ia, ja = dense ja + 128), or neither -- with
their products (16 x 16 x 4) and the operand bytes k_chol_big stages for them.  Usage: pair_stats.py [WORKLOAD]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import _native as N, inspector as I, matrices as M  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "flan"
A, perm = M.workload(name)
sym = I.analyze(A, perm)
lib = N.lib()
lib.parsy_plan_from_symbolic.restype = C.c_void_p
h = lib.parsy_plan_from_symbolic(sym._handle, -1)
assert h, N.last_error()
lib.parsy_debug_big_windows.restype = C.c_int64
lib.parsy_debug_big_windows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
n = lib.parsy_debug_big_windows(h, None, 0)
out = np.zeros((n, 9), dtype=np.int64)
lib.parsy_debug_big_windows(h, out.ctypes.data, n)
task, launch, K, mi, nj, ia, ja, src, dense = out.T
dense = dense.astype(bool)
flops = 2.0 * K * mi * nj
bytes_ = 8.0 * K * (mi + nj)
print(f"{name}: {n} entries; dense {dense.sum()} ({flops[dense].sum():.3e} flops, {bytes_[dense].sum() / 1e9:.1f} GB staged); "
      f"ragged {(~dense).sum()} ({flops[~dense].sum():.3e} flops, {bytes_[~dense].sum() / 1e9:.1f} GB staged)")
# index the dense blocks by (task, src, ia, ja)
key = {}
for i in np.where(dense)[0]:
    key[(task[i], src[i], ia[i], ja[i])] = i
cls = np.zeros(n, dtype=np.int8)     # 1 = row remainder behind a dense block, 2 = column remainder beside one, 3 = corner of both
for i in np.where(~dense)[0]:
    below = (task[i], src[i], ia[i] - 128, ja[i]) in key
    beside = (task[i], src[i], ia[i], ja[i] - 128) in key
    corner = (task[i], src[i], ia[i] - 128, ja[i] - 128) in key
    cls[i] = 1 if below else 2 if beside else 3 if corner else 0
for c, what in ((1, "row remainder behind a dense block (C window staged already)"),
                (2, "column remainder beside a dense block (R window staged already)"),
                (3, "corner of a dense block's remainders"), (0, "next to no dense block of its task")):
    m = (~dense) & (cls == c)
    if not m.any():
        print(f"  {what}: none")
        continue
    saved = 8.0 * K[m] * (nj[m] if c == 1 else mi[m] if c == 2 else 0)
    print(f"  {what}: {m.sum()} entries, {flops[m].sum():.3e} flops ({flops[m].sum() / flops[~dense].sum():.2f} of the ragged), "
          f"staged {bytes_[m].sum() / 1e9:.1f} GB of which shared with the dense block {saved.sum() / 1e9:.1f} GB; "
          f"rows mean {mi[m].mean():.0f} cols mean {nj[m].mean():.0f} K mean {K[m].mean():.0f}")
    for lo, hi in ((1, 16), (17, 32), (33, 64), (65, 127)):
        ext = mi if c == 1 else nj
        q = m & (ext >= lo) & (ext <= hi)
        if q.any():
            print(f"      {'rows' if c == 1 else 'columns'} {lo:3d}..{hi:3d}: {q.sum():8d} entries {flops[q].sum():.3e} flops {bytes_[q].sum() / 1e9:7.1f} GB")
