"""On the GPU box: the small configurations as bench.py times them (N calls back to back, one synchronisation at the
end): factorization, forward and backward solve, with the ONE-launch solves and with the level launches.
Usage: small_jobs.py [WORKLOAD ...]   (default: ex15)"""
import os
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M


def timed(fn, warm, reps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name in sys.argv[1:] or ["ex15"]:
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    dev = torch.device("cuda", 0)
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
    for mode in os.environ.get("SMALL_JOBS_MODES", "0,1").split(","):
        os.environ["PARSY_SOLVE_ONE"] = mode
        plan = api.Plan(sym, 0)
        f = timed(lambda: plan.factor_device(values.data_ptr(), L.data_ptr(), 0), 5, 50)
        b = torch.empty(sym.n, dtype=torch.float64, device=dev)
        plan.rhs_ones_device(L.data_ptr(), b.data_ptr(), 0)
        for nrhs in [int(v) for v in os.environ.get("SMALL_JOBS_NRHS", "1,4").split(",")]:
            B = b.repeat(nrhs).contiguous()
            X = B.clone()
            fs = timed(lambda: plan.solve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0), 5, 50)
            X.copy_(B)
            plan.solve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
            torch.cuda.synchronize()
            err = float((X - 1.0).abs().max())
            gpu_f = plan.last_solve_ms()
            bs = timed(lambda: plan.backsolve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0), 5, 50)
            gpu_b = plan.last_solve_ms()
            print(f"{name} PARSY_SOLVE_ONE={mode} solve_one={plan.info['solve_one']} nrhs {nrhs}: factorization {f:.4f} ms, forward {fs:.4f} ms"
                  f" (events around one: {gpu_f:.4f}), backward {bs:.4f} ms ({gpu_b:.4f}); forward error vs ones {err:.2e}, status {plan.solve_status()}")
        del plan
