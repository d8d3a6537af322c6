"""Serialised time per launch kind of one profiled factorization (every launch alone on the device); no checks, so
that ablation builds with wrong results can be timed.  Usage: [PARSY_LIB=...] kinds.py [WORKLOAD]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
A, perm = M.workload(sys.argv[1] if len(sys.argv) > 1 else "flan")
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
for _ in range(2):
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
ms = plan.last_factor_ms()
plan.profile(2)
for _ in range(2):
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
    torch.cuda.synchronize()
    plan.profile_collect()
pr = plan.profile_get()
print(os.environ.get("PARSY_LIB", "product").split("/")[-1], "overlapped ms %.1f" % ms,
      {k: round(v / pr["runs"], 1) for k, v in pr["ms"].items() if v > 0})
