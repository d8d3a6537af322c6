"""Per-launch times of the BIG (or, with KIND = 0 SMALL / 1 TILES / 2 CHAIN, another kind's) launches of one profiled
factorization (every launch alone on the device).
Usage: [PARSY_BIG_SUPER=..] big_launches.py [WORKLOAD [KIND]] -> rows (level, side, tasks, ms)."""
import ctypes as C
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
out = np.zeros((n, 4))
lib.parsy_debug_launch_times(plan._h, out.ctypes.data, n)
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 3
big = out[out[:, 0] == kind]
print("launches of kind", kind, ":", len(big), "total ms", big[:, 3].sum())
for r in big:
    print(f"level {int(r[1]) >> 1:3d} side {int(r[1]) & 1} tasks {int(r[2]):7d} ms {r[3]:8.3f}")
