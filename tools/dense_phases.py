"""Diagnostic: shader clocks per phase of k_chol_dense's chunk loop, summed over every wave of every dense launch of
one factorization (tools/build_variant.sh densestamps -DPARSY_DENSESTAMPS).  Usage: dense_phases.py [WORKLOAD]"""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import _native as N
N._LIB_PATH = ROOT / "tools" / "libparsy_densestamps.bin"
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
A, perm = M.workload(sys.argv[1] if len(sys.argv) > 1 else "flan")
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
out = np.zeros(16, dtype=np.uint64)
for rep in range(2):
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
    torch.cuda.synchronize()
    N.lib().parsy_debug_densephase(out.ctypes.data_as(C.c_void_p), 1)
print("status", plan.status(), "ms", plan.last_factor_ms())
names = ["loop top", "products A", "wait own DMA", "barrier", "fetch (DMA issue)", "read A' + wait B", "mask + products B",
         "epilogue (entry end)", "read B' + wait A' + mask"]
chunks = float(out[9])
tot = float(out[:9].sum())
print(f"chunks (wave x chunk): {chunks:.3e}; clocks per chunk and wave: {tot / chunks:.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:28s} {float(out[i]) / chunks:8.1f}  {float(out[i]) / tot:6.3f}")
wt = float(out[11])
if wt > 0:
    print(f"tasks (wave x task): {wt:.3e}; chunks per task {chunks / wt:.1f}; clocks from kernel entry to the first products, per task: "
          f"{float(out[10]) / wt:.0f}; in-loop clocks per task {tot / wt:.0f}")
