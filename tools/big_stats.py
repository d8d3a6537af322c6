"""Host-side statistics of the BIG launches (no GPU): how full the 128 x 128 windows of the (task, source) entries
are, how the k extent is distributed, and what share of the issued MFMA time the busiest wave of a workgroup
determines (the other waves wait for it at the chunk barrier).  Usage: big_stats.py [WORKLOAD]"""
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
lib.parsy_debug_big_entries.restype = C.c_int64
lib.parsy_debug_big_entries.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
n = lib.parsy_debug_big_entries(h, None, 0)
out = np.zeros((n, 7), dtype=np.int32)
lib.parsy_debug_big_entries(h, out.ctypes.data, n)
task, launch, K, mi, nj, ident, diag = out.T.astype(np.int64)
chunks = (K + 15) // 16
print(f"{name}: {task.max() + 1} tasks, {n} entries, {chunks.sum():.3e} chunks; entries per task "
      f"{n / (task.max() + 1):.2f}")
flops = 2.0 * K * mi * nj
print(f"flops (full windows, incl. upper parts of diagonal tiles) {flops.sum():.4e}")
# waves: 2 x 4, 64 x 32 outputs; fragments of wave (wr, wc)
tot_issue = np.zeros(n)
max_issue = np.zeros(n)
for wr in range(2):
    for wc in range(4):
        nfr = np.clip((mi - 64 * wr + 15) >> 4, 0, 4)
        nfc = np.clip((nj - 32 * wc + 15) >> 4, 0, 2)
        skip = (diag == 1) & (64 * wr + 63 < 32 * wc)
        f = np.where(skip, 0, nfr * nfc)
        tot_issue += f
        max_issue = np.maximum(max_issue, f)
# per SIMD two waves of the workgroup: waves w and w + 4 share SIMD (w & 3)?  Use the max over waves as the bound.
w = chunks
print(f"issued fragments / (8 waves x 8): {(tot_issue * w).sum() / (64 * w.sum()):.3f}; "
      f"busiest wave / 8: {(max_issue * w).sum() / (8 * w.sum()):.3f}")
print("chunk-weighted histogram of mi (rows in the row window):")
for lo, hi in ((1, 32), (33, 64), (65, 96), (97, 127), (128, 128)):
    m = (mi >= lo) & (mi <= hi)
    print(f"  mi {lo:3d}-{hi:3d}: {w[m].sum() / w.sum():.3f}")
print("chunk-weighted histogram of nj:")
for lo, hi in ((1, 32), (33, 64), (65, 96), (97, 127), (128, 128)):
    m = (nj >= lo) & (nj <= hi)
    print(f"  nj {lo:3d}-{hi:3d}: {w[m].sum() / w.sum():.3f}")
print("full windows (mi = nj = 128):", f"{w[(mi == 128) & (nj == 128)].sum() / w.sum():.3f}")
print("chunk-weighted histogram of K:")
for lo, hi in ((1, 127), (128, 255), (256, 511), (512, 512), (513, 1 << 30)):
    m = (K >= lo) & (K <= hi)
    print(f"  K {lo}-{hi}: chunks {w[m].sum() / w.sum():.3f} entries {m.sum() / n:.3f}")
print("identity-map entries:", f"chunks {w[ident == 1].sum() / w.sum():.3f} entries {(ident == 1).sum() / n:.3f}")
print("diagonal-tile entries:", f"chunks {w[diag == 1].sum() / w.sum():.3f}")
# consecutive entries of one task with the identity map (accumulators could be carried)
same = (task[1:] == task[:-1]) & (ident[1:] == 1) & (ident[:-1] == 1)
print("identity entries that follow an identity entry of the same task:", same.sum(), "of", n)
ept = np.bincount(task)
print("entries per task percentiles 50/90/99/max:", np.percentile(ept, [50, 90, 99]), ept.max())
cpt = np.bincount(task, weights=w)
print("chunks per task percentiles 10/50/90/max:", np.percentile(cpt, [10, 50, 90]), cpt.max())

# ---- what a per-entry choice of the wave grid would give: busiest wave's fragments per chunk, chunk-weighted mean
NR, NC = (mi + 15) >> 4, (nj + 15) >> 4
tot = NR * NC   # (upper parts of diagonal tiles are skipped by whole waves only; ignored here)
def busiest(WR, WC, even=True):
    if even:
        return -(-NR // WR) * -(-NC // WC)
    fr = np.minimum(NR, 8 // WR)
    fc = np.minimum(NC, 8 // WC)
    return fr * fc
print(f"mean fragments per chunk {np.average(tot, weights=w):.2f} (/8 = {np.average(tot, weights=w) / 8:.2f} per wave)")
print("current (2x4, fixed 64x32 blocks):", f"{np.average(busiest(2, 4, False), weights=w):.3f}")
grids = [(2, 4), (1, 8), (4, 2), (8, 1)]
for k in range(1, 5):
    import itertools
    for sel in itertools.combinations(grids, k):
        if (2, 4) not in sel:
            continue
        best = np.minimum.reduce([busiest(a, b) for a, b in sel])
        print("grids", sel, f"busiest wave {np.average(best, weights=w):.3f}")
# two waves of a workgroup share a SIMD (w and w + 4): SIMD load = sum of the two
