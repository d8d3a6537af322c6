"""Soak test of the in-launch hand-offs: many factorizations of several inputs, one plan alone and two
plans concurrently (uneven load on the chip), every factor compared bitwise with the first one and the
status word checked.  Bounded: a few thousand launches, one process."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
dev = torch.device("cuda", 0)
bad = 0
for name, reps in (("mid3d", 1500), ("lap30", 600), ("nd24k", 600)):
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    plans = [api.Plan(sym, 0), api.Plan(sym, 0)]
    Ls = [torch.empty(int(sym.xsize), dtype=torch.float64, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    plans[0].factor_device(values.data_ptr(), Ls[0].data_ptr(), streams[0].cuda_stream)
    torch.cuda.synchronize()
    ref = Ls[0].clone()
    torch.cuda.synchronize()  # (the clone runs on torch's stream: it must be done before Ls[0] is rewritten)
    t0 = time.time()
    for mode in ("alone", "two in flight"):
        mism = stat = 0
        for i in range(reps):
            if mode == "alone":
                plans[0].factor_device(values.data_ptr(), Ls[0].data_ptr(), streams[0].cuda_stream)
                torch.cuda.synchronize()
                ok = bool(torch.equal(Ls[0], ref))
                st = plans[0].status()
            else:
                for j in range(2):
                    plans[j].factor_device(values.data_ptr(), Ls[j].data_ptr(), streams[j].cuda_stream)
                torch.cuda.synchronize()
                ok = bool(torch.equal(Ls[0], ref)) and bool(torch.equal(Ls[1], ref))
                st = plans[0].status() or plans[1].status()
            mism += not ok
            stat += st != 0
        print(f"{name:6s} {mode:14s}: {reps} rounds, {mism} factor mismatches, {stat} bad status, {time.time() - t0:.1f} s", flush=True)
        bad += mism + stat
# the solve chains (forward: flags + published x blocks, atomics on shared rows -> equal to rounding;
# backward: fixed order -> bitwise)
for name, reps in (("lap30", 1000), ("nd24k", 1000)):
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    plan = api.Plan(sym, 0)
    L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
    torch.cuda.synchronize()
    rng = np.random.default_rng(3)
    b = torch.from_numpy(rng.standard_normal(sym.n)).to(dev)
    x = b.clone(); plan.solve_device(L.data_ptr(), x.data_ptr(), 1, sym.n, 0); torch.cuda.synchronize()
    xf = x.clone()
    y = b.clone(); plan.backsolve_device(L.data_ptr(), y.data_ptr(), 1, sym.n, 0); torch.cuda.synchronize()
    yb = y.clone()
    torch.cuda.synchronize()
    worst = 0.0; bmis = 0; stat = 0
    for i in range(reps):
        x.copy_(b); plan.solve_device(L.data_ptr(), x.data_ptr(), 1, sym.n, 0)
        y.copy_(b); plan.backsolve_device(L.data_ptr(), y.data_ptr(), 1, sym.n, 0)
        torch.cuda.synchronize()
        worst = max(worst, float((x - xf).abs().max() / xf.abs().max()))
        bmis += not bool(torch.equal(y, yb))
        stat += plan.status() != 0
    print(f"{name:6s} solves: {reps} rounds, forward max rel deviation {worst:.2e}, backward mismatches {bmis}, bad status {stat}", flush=True)
    bad += bmis + stat + (worst > 1e-12)
print("SOAK", "OK" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
