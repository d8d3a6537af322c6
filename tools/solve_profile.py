"""Diagnostic: per-kind launch times of the forward / backward solve of a workload for several nrhs."""
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
b = torch.empty(sym.n, dtype=torch.float64, device=dev)
plan.rhs_ones_device(L.data_ptr(), b.data_ptr(), 0)
info = plan.info
print(name, "n", sym.n, "nsuper", sym.nsuper, "levels", sym.nlevels, "solve launches", info["solve_launches"], "maxw", sym.maxSupWid)
w = np.diff(sym.super); r = np.diff(sym.i_ptr[sym.super].astype(np.int64))
print("xsize", sym.xsize, "in supernodes wider than 64:", int((w * r)[w > 64].sum()), "count", int((w > 64).sum()))
for nrhs in [int(v) for v in sys.argv[2:]]:
    B = b.repeat(nrhs).contiguous()
    X = torch.empty_like(B)
    for back in (0, 1):
        fn = plan.backsolve_device if back else plan.solve_device
        for _ in range(2):
            X.copy_(B); fn(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(5):
            fn(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5 * 1e3
        plan.profile(2)
        for _ in range(3):
            X.copy_(B); fn(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
            torch.cuda.synchronize(); plan.profile_collect()
        p = plan.profile_get(); plan.profile(0)
        print(f"nrhs {nrhs} {'backward' if back else 'forward'}: {dt:.3f} ms; serialized kinds:",
              {k: round(v / p['runs'], 3) for k, v in p['ms'].items() if v > 0},
              {k: v // p['runs'] for k, v in p['launches'].items() if v > 0}, "status", plan.solve_status())
