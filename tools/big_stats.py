"""Host-side statistics of the BIG launches (no GPU): how full the 128 x 128 blocks of the (task, source) entries
are, how the k extent is distributed, and how many 16 x 16 fragments the busiest wave of a workgroup multiplies per k
step (the other waves wait for it at the chunk barrier).  Usage: [PARSY_BIG_SUPER=RxC] big_stats.py [WORKLOAD]"""
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
task, launch, K, mi, nj, ident, dij = out.T.astype(np.int64)
w = (K + 15) // 16   # chunks
ntask = task.max() + 1
print(f"{name}: {ntask} tasks, {n} entries, {w.sum():.3e} chunks; entries per task {n / ntask:.2f}")
NR, NC = (mi + 15) >> 4, (nj + 15) >> 4


def dealt(even):
    """fragments per wave (2 x 4 waves) as k_chol_big deals them; returns (sum over waves, busiest wave)"""
    tot, mx = np.zeros(n), np.zeros(n)
    frb = (NR + 1) >> 1 if even else np.full(n, 4)
    fcb = (NC + 3) >> 2 if even else np.full(n, 2)
    for wr in range(2):
        for wc in range(4):
            nfr = np.clip(NR - frb * wr, 0, frb)
            nfc = np.clip(NC - fcb * wc, 0, fcb)
            above = dij + 16 * frb * wr + 16 * nfr - 1 < 16 * fcb * wc
            f = np.where(above, 0, nfr * nfc)
            tot += f
            mx = np.maximum(mx, f)
    return tot, mx


for even in (False, True):
    tot, mx = dealt(even)
    print(f"{'even deal' if even else 'fixed 64 x 32 blocks'}: fragments per chunk {np.average(tot, weights=w):.2f} "
          f"(per wave {np.average(tot, weights=w) / 8:.2f}), busiest wave {np.average(mx, weights=w):.3f}; "
          f"MFMA k steps issued by the busiest waves {(mx * w).sum() * 4:.4e}")
print("full blocks (mi = nj = 128):", f"{w[(mi == 128) & (nj == 128)].sum() / w.sum():.3f} of the chunks")
for nm, v in (("mi", mi), ("nj", nj)):
    print(f"chunk-weighted histogram of {nm}:", " ".join(
        f"{lo}-{hi}: {w[(v >= lo) & (v <= hi)].sum() / w.sum():.3f}" for lo, hi in ((1, 32), (33, 64), (65, 96), (97, 127), (128, 128))))
print("chunk-weighted histogram of K:", " ".join(
    f"{lo}-{hi}: {w[(K >= lo) & (K <= hi)].sum() / w.sum():.3f}/{((K >= lo) & (K <= hi)).sum() / n:.3f}"
    for lo, hi in ((1, 127), (128, 255), (256, 511), (512, 512), (513, 1 << 30))), "(chunks/entries)")
print("identity-map entries:", f"chunks {w[ident == 1].sum() / w.sum():.3f} entries {(ident == 1).sum() / n:.3f}")
ept = np.bincount(task)
print("entries per task percentiles 50/90/99/max:", np.percentile(ept, [50, 90, 99]), ept.max())
cpt = np.bincount(task, weights=w)
print("chunks per task percentiles 10/50/90/max:", np.percentile(cpt, [10, 50, 90]), cpt.max())
# tasks per launch
tl = np.zeros(ntask, dtype=np.int64)
tl[task] = launch
tpl_all = np.bincount(tl, minlength=int(launch.max()) + 1)
tpl = tpl_all[tpl_all > 0]
print(f"{len(tpl)} launches; tasks per launch percentiles 10/50/90: {np.percentile(tpl, [10, 50, 90])}; "
      f"launches with fewer than 512 / 1024 tasks: {(tpl < 512).sum()} / {(tpl < 1024).sum()}")

# ---- per launch: tasks, chunks, issued 16x16x4 products (even deal) -> with the measured times of tools/big_launches.py
# (argv[2], optional) the share of the MFMA pipes' time each launch keeps busy
if len(sys.argv) > 2:
    tot, mx = dealt(True)
    rows = [l.split() for l in open(sys.argv[2]) if l.startswith("level")]
    ms = {(int(r[1]), int(r[3])): float(r[7]) for r in rows}
    prod = np.bincount(launch, weights=tot * w * 4)
    crit = np.bincount(launch, weights=mx * w * 4)
    nch = np.bincount(launch, weights=w)
    print("level side | tasks chunks | ms | MFMA busy | busiest-wave bound ms | chunks/slot us/chunk")
    for lid in np.nonzero(prod)[0]:
        lev, side = lid >> 1, lid & 1
        key = (lev + 2, 1) if side else (lev + 1, 0)   # the level whose launches wait for a side launch: s + 2
        if key not in ms:
            continue
        t = ms[key] * 1e-3
        busy = prod[lid] * 64 / (1024 * 2.37e9 * t)
        bound = crit[lid] * 64 * 2 / (512 * 2.37e9) * 1e3 / 2   # 8 waves on 4 SIMDs: two waves per SIMD and workgroup
        print(f"{lev:3d} {side} | {tpl_all[lid]:6d} {int(nch[lid]):9d} | {ms[key]:8.3f} | {busy:.3f} | {bound:8.3f} | "
              f"{nch[lid] / 512:8.0f} {t * 1e6 / max(nch[lid] / 512, 1):6.2f}")
