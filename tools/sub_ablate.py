"""Diagnostic: time of the subtree launches of the many-right-hand-side solves per ablation mask (PARSY_SUB_ABL, needs the
build of tools/build_variant.sh subabl -DPARSY_SUBABL loaded through PARSY_LIB).  Usage: sub_ablate.py WORKLOAD NRHS MASK[,MASK...]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
name, nrhs = sys.argv[1], int(sys.argv[2])
A, perm = M.workload(name)
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
X = torch.ones(sym.n * nrhs, dtype=torch.float64, device=dev)
for mask in sys.argv[3].split(","):
    os.environ["PARSY_SUB_ABL"] = mask
    res = []
    for fn in (plan.solve_device, plan.backsolve_device):
        t = []
        for _ in range(4):
            X.fill_(1.0)
            plan.profile(1)
            fn(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
            torch.cuda.synchronize()
            plan.profile_collect()
            t.append(plan.last_solve_ms())
        res.append(min(t[1:]))
    print(f"mask {int(mask):3d}: forward {res[0]:.3f} ms backward {res[1]:.3f} ms", flush=True)
