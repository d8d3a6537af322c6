"""On the GPU box: every THIN launch of a profiled factorization (each launch alone on the device): tasks, total and
longest task weight (k steps of four + 8 per piece), measured ms, and what the longest task alone would take at the
launch's mean rate.  Usage: thin_launches.py [WORKLOAD]"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, _native as N, inspector as I, matrices as M
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
lib.parsy_debug_thin_tasks.restype = C.c_int64
lib.parsy_debug_thin_tasks.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
nt = lib.parsy_debug_thin_tasks(plan._h, None, 0)
tt = np.zeros((nt, 2), dtype=np.int32)
lib.parsy_debug_thin_tasks(plan._h, tt.ctypes.data, nt)
KIND_THIN = api.KIND_NAMES.index("THIN")
# (the profile lists launches in enqueue order; the schedule's chol list is main-stream order with the early launches
# spliced in: the THIN launches keep their relative order in both)
thin_rows = lt[lt[:, 0] == KIND_THIN]
print(f"THIN launches {len(thin_rows)}, tasks {nt}, ms {thin_rows[:, 3].sum():.2f}")
tot_tail = 0.0
for li, row in enumerate(thin_rows):
    w = tt[tt[:, 0] == li, 1].astype(np.int64)
    w = w[w > 0]
    if len(w) == 0:
        continue
    rate = w.sum() / max(row[3], 1e-9)          # weight per ms with the whole device
    slots = 256 * 4 * 2
    print(f"  launch {li:3d} level/side {int(row[1]):4d}: tasks {len(w):7d} weight {w.sum():10d} longest {w.max():7d}"
          f" mean {w.mean():7.1f}  ms {row[3]:7.3f}  weight/us/slot {w.sum() / (row[3] * 1e3) / slots:6.3f}"
          f"  balanced bound {max(w.max(), w.sum() / slots):9.0f}")
