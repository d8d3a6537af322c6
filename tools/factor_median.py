"""Diagnostic: device time of the numeric factorization of a workload, median / min of N synchronised runs after a warm-up
(PARSY_LIB selects a diagnostic build).  Usage: factor_median.py WORKLOAD [RUNS]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
name = sys.argv[1]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 40
A, perm = M.workload(name)
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
t = []
for i in range(runs + 5):
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
    torch.cuda.synchronize()
    if i >= 5:
        t.append(plan.last_factor_ms())
assert plan.status() == 0
print(f"{name}: median {np.median(t):.4f} ms, min {min(t):.4f}, p90 {np.percentile(t, 90):.4f} ({runs} runs)")
