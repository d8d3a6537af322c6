"""Minimal driver for profiling: N factorizations (+ optional solves) of a workload, device-resident.
Usage: one_factor.py WORKLOAD [FACTORIZATIONS [SOLVES [NRHS]]]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
name = sys.argv[1] if len(sys.argv) > 1 else "nd24k"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
A, perm = M.workload(name)
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
for _ in range(reps):
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
print("status", plan.status(), "ms", plan.last_factor_ms())
nsolve = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nrhs = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if nsolve:
    x = torch.ones(sym.n * nrhs, dtype=torch.float64, device=dev)
    for _ in range(nsolve):
        plan.solve_device(L.data_ptr(), x.data_ptr(), nrhs, sym.n, 0)
        torch.cuda.synchronize()
    print("forward solve ms", plan.last_solve_ms())
    for _ in range(nsolve):
        plan.backsolve_device(L.data_ptr(), x.data_ptr(), nrhs, sym.n, 0)
        torch.cuda.synchronize()
    print("backward solve ms", plan.last_solve_ms())
