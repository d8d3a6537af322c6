"""Host-side (no GPU): which share of the BIG launches' products sits in DENSE entries -- full 128 x 128 blocks of a
source's rows that lie entirely on or below the target's diagonal (what k_chol_dense takes) -- per launch and in all,
for the super-tile choice in the environment.  Usage: [PARSY_BIG_SUPER=RxC] dense_stats.py [WORKLOAD]"""
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
NR, NC = (mi + 15) >> 4, (nj + 15) >> 4
# issued 16x16x4 products of an entry (fragments on or below the diagonal x k steps of four)
kst = (K + 3) // 4
fr = np.arange(8)[:, None, None]
fc = np.arange(8)[None, :, None]
have = (fr < NR[None, None, :]) & (fc < NC[None, None, :]) & (dij[None, None, :] + 16 * fr + 15 >= 16 * fc)
frags = have.sum(axis=(0, 1))
prod = frags * kst
dense = (mi == 128) & (nj == 128)
print(f"{name}: {task.max() + 1} tasks, {n} entries; products in dense entries: {prod[dense].sum() / prod.sum():.3f}"
      f" (entries {dense.mean():.3f})")
full = (mi == 128) & (nj == 128)
print(f"  full blocks incl. diagonal ones: {prod[full].sum() / prod.sum():.3f}")
tl = np.unique(launch)
print("launch(src level, push) tasks  products(1e9)  dense share | dense tasks, ragged tasks")
for lid in tl:
    m = launch == lid
    p = prod[m].sum()
    if p < 0.005 * prod.sum():
        continue
    td = len(np.unique(task[m & dense]))
    tr = len(np.unique(task[m & ~dense]))
    print(f"  {lid >> 1:3d} {'push' if lid & 1 else 'next'} {len(np.unique(task[m])):7d} {p / 1e9:10.3f} {prod[m & dense].sum() / p:8.3f} | {td:7d} {tr:7d}")

# ---- load balance of the dense launches: chunks (8 k) per task in launch order, list-scheduled onto 512 workgroup
# slots (2 per compute unit): makespan against the perfectly balanced total / 512
import heapq
lib.parsy_debug_dense_tasks.restype = C.c_int64
lib.parsy_debug_dense_tasks.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
nt = lib.parsy_debug_dense_tasks(h, None, 0)
dt = np.zeros((nt, 2), dtype=np.int32)
lib.parsy_debug_dense_tasks(h, dt.ctypes.data, nt)
tot_bal = tot_mk = 0.0
rows = []
for lid in np.unique(dt[:, 0]):
    ch = dt[dt[:, 0] == lid, 1].astype(np.int64)
    slots = [0] * 512
    heapq.heapify(slots)
    for c in ch:
        heapq.heappush(slots, heapq.heappop(slots) + int(c) + 6)   # (+ ~6 chunk times of task start / end)
    mk = max(slots)
    bal = ch.sum() / 512.0
    tot_bal += bal
    tot_mk += mk
    rows.append((lid, len(ch), ch.sum(), ch.max(), mk / max(bal, 1)))
print(f"dense launches: {len(rows)}; balanced chunk-times {tot_bal:.3e}, list-scheduled makespan {tot_mk:.3e}: x{tot_mk / tot_bal:.3f}")
for r in rows:
    if r[2] > 0.01 * sum(x[2] for x in rows):
        print(f"  launch {r[0]:4d}: tasks {r[1]:6d} chunks {r[2]:9d} longest {r[3]:6d} makespan / balanced {r[4]:.3f}")
