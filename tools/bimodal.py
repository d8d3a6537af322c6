"""Diagnostic (review item 7): the forward ONE-launch solve of a large plan, forced (PARSY_SOLVE_ONE=2), call by call --
its device time per call, the addresses of L and x, for several processes side by side (run it N times).
Usage: bimodal.py [WORKLOAD [CALLS]]"""
import os
import sys
from pathlib import Path
import numpy as np
os.environ.setdefault("PARSY_SOLVE_ONE", "2")
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
name = sys.argv[1] if len(sys.argv) > 1 else "flan"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
A, perm = M.workload(name)
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
b = torch.ones(sym.n, dtype=torch.float64, device=dev)
x = torch.empty_like(b)
tf, tb = [], []
for i in range(calls):
    x.copy_(b)
    plan.solve_device(L.data_ptr(), x.data_ptr(), 1, sym.n, 0)
    torch.cuda.synchronize()
    tf.append(plan.last_solve_ms())
for i in range(calls):
    x.copy_(b)
    plan.backsolve_device(L.data_ptr(), x.data_ptr(), 1, sym.n, 0)
    torch.cuda.synchronize()
    tb.append(plan.last_solve_ms())
print(f"{name} solve_one={plan.info['solve_one']} L@{L.data_ptr():#x} x@{x.data_ptr():#x} status {plan.solve_status()}")
print("  forward  ms:", " ".join(f"{t:.3f}" for t in tf))
print("  backward ms:", " ".join(f"{t:.3f}" for t in tb))
