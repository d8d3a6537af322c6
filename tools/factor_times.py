"""On the GPU box: ms per factorization (50 back to back) of the named workloads with the product's defaults."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
for name in sys.argv[1:]:
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    dev = torch.device("cuda", 0)
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
    plan = api.Plan(sym, 0)
    reps = 50 if sym.n < 200000 else 8
    for _ in range(3):
        plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / reps * 1e3:.4f} ms per factorization, status {plan.status()}, launches {plan.info['chol_launches']}")
