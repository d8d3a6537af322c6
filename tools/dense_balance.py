"""What the DENSE launches lose to their ends: per launch of one profiled Flan-class factorization (every launch alone on the
device) the tasks' chunk counts (8-wide k chunks; a task is one workgroup, 512 run side by side: 2 per compute unit), the
makespan of a greedy deal of those tasks over 512 slots against total / 512, and the measured time.
Usage: dense_balance.py [WORKLOAD]"""
import ctypes as C
import heapq
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import _native as N, api, inspector as I, matrices as M
A, perm = M.workload(sys.argv[1] if len(sys.argv) > 1 else "flan")
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
for _ in range(2):
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
plan.profile(2)
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
plan.profile_collect()
lib = N.lib()
lib.parsy_debug_launch_times.restype = C.c_int64
lib.parsy_debug_launch_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
n = lib.parsy_debug_launch_times(plan._h, None, 0)
lt = np.zeros((n, 4))
lib.parsy_debug_launch_times(plan._h, lt.ctypes.data, n)
dense_ms = lt[lt[:, 0] == 9][:, 3]
lib.parsy_debug_dense_tasks.restype = C.c_int64
lib.parsy_debug_dense_tasks.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
m = lib.parsy_debug_dense_tasks(plan._h, None, 0)
dt = np.zeros((m, 2), dtype=np.int32)
lib.parsy_debug_dense_tasks(plan._h, dt.ctypes.data, m)
dt[:, 1] &= (1 << 30) - 1
SLOTS = 512
tot_ms = tot_bal = 0.0
rows = []
for li in range(int(dt[:, 0].max()) + 1):
    ch = np.sort(dt[dt[:, 0] == li][:, 1])[::-1]
    ch = ch[ch > 0]
    if len(ch) == 0 or li >= len(dense_ms):
        continue
    heap = [0] * SLOTS
    for c in ch:
        heapq.heapreplace(heap, heap[0] + int(c))
    mk, ideal = max(heap), ch.sum() / SLOTS
    ms = float(dense_ms[li])
    rows.append((li, len(ch), int(ch.sum()), ideal, mk, ms))
    tot_ms += ms
    tot_bal += ms * max(ideal, ch.max()) / mk
print(f"{len(rows)} DENSE launches, {tot_ms:.1f} ms; with every launch's chunks dealt evenly over {SLOTS} slots (never below its longest task): {tot_bal:.1f} ms")
print("launch tasks chunks ideal/slot greedy-makespan longest ms  us/chunk(makespan)")
for li, nt, tc, ideal, mk, ms in sorted(rows, key=lambda r: -r[5])[:40]:
    ch = np.sort(dt[dt[:, 0] == li][:, 1])[::-1]
    print(f"{li:4d} {nt:6d} {tc:9d} {ideal:9.1f} {mk:7d} {int(ch.max()):6d} {ms:7.3f}  {ms * 1e3 / mk:6.3f}")
