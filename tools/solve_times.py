"""Diagnostic: forward / backward solve times of a workload for several numbers of right-hand sides.
Usage: python tools/solve_times.py WORKLOAD NRHS[,NRHS...]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
name = sys.argv[1]
A, perm = M.workload(name)
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
assert plan.status() == 0
b = torch.empty(sym.n, dtype=torch.float64, device=dev)
plan.rhs_ones_device(L.data_ptr(), b.data_ptr(), 0)
for nrhs in [int(v) for v in sys.argv[2].split(",")]:
    B = b.repeat(nrhs).contiguous()
    X = torch.empty_like(B)
    f, bk = [], []
    for _ in range(4):
        X.copy_(B)
        plan.solve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
        torch.cuda.synchronize()
        f.append(plan.last_solve_ms())
    err = float((X - 1).abs().max())
    for _ in range(4):
        X.copy_(B)
        plan.backsolve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
        torch.cuda.synchronize()
        bk.append(plan.last_solve_ms())
    print(f"{name} nrhs={nrhs}: forward {min(f[1:]):.3f} ms (err {err:.1e}) backward {min(bk[1:]):.3f} ms status {plan.solve_status()}", flush=True)
