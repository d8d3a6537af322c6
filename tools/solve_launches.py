"""On the GPU box: every launch of a profiled forward / backward solve (each launch alone on the device):
kind, level, work items, ms; and of the factorization when it has at most 64 launches.  Usage: solve_launches.py WORKLOAD NRHS [NRHS ...]"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, _native as N, inspector as I, matrices as M
name = sys.argv[1]
A, perm = M.workload(name)
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
lib0 = N.lib()
lib0.parsy_debug_launch_times.restype = C.c_int64
lib0.parsy_debug_launch_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
plan.profile(2)
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize(); plan.profile_collect()
n = lib0.parsy_debug_launch_times(plan._h, None, 0)
lt = np.zeros((n, 4))
lib0.parsy_debug_launch_times(plan._h, lt.ctypes.data, n)
plan.profile(0)
print(f"{name} factorization: {lt[:, 3].sum():.3f} ms serialised in {n} launches (overlapped: {plan.last_factor_ms():.3f} ms)")
if n <= 64:
    for row in lt:
        k = int(row[0])
        print(f"   {str(api.KIND_NAMES[k] if 0 <= k < len(api.KIND_NAMES) else k):12s} level {int(row[1]) >> 1 if row[1] >= 0 else -1:3d} items {int(row[2]):7d}  {row[3]:7.3f} ms")
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
b = torch.empty(sym.n, dtype=torch.float64, device=dev)
plan.rhs_ones_device(L.data_ptr(), b.data_ptr(), 0)
lib = N.lib()
lib.parsy_debug_launch_times.restype = C.c_int64
lib.parsy_debug_launch_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
for nrhs in [int(v) for v in sys.argv[2:]]:
    B = b.repeat(nrhs).contiguous()
    X = torch.empty_like(B)
    for back in (0, 1):
        fn = plan.backsolve_device if back else plan.solve_device
        for _ in range(2):
            X.copy_(B); fn(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
        torch.cuda.synchronize()
        plan.profile(2)
        X.copy_(B); fn(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
        torch.cuda.synchronize(); plan.profile_collect()
        n = lib.parsy_debug_launch_times(plan._h, None, 0)
        lt = np.zeros((n, 4))
        lib.parsy_debug_launch_times(plan._h, lt.ctypes.data, n)
        plan.profile(0)
        print(f"{name} nrhs {nrhs} {'backward' if back else 'forward'}: {lt[:, 3].sum():.3f} ms serialised")
        for row in lt:
            k = int(row[0])
            print(f"   {str(api.KIND_NAMES[k] if 0 <= k < len(api.KIND_NAMES) else k):12s} level {int(row[1]) >> 1 if row[1] >= 0 else -1:3d} items {int(row[2]):7d}  {row[3]:7.3f} ms")
