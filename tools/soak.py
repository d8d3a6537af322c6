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
FACTOR_REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
SOLVE_REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for name, reps in (("mid3d", 2 * FACTOR_REPS), ("lap30", FACTOR_REPS), ("nd24k", FACTOR_REPS)):
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
# the solves (forward: atomics on shared rows -> equal to rounding; backward: fixed order -> bitwise), with one
# right-hand side (wave kernels, armed hand-off buffer) and with five (workgroup kernels, flags); lap30 a second
# time with subtree launches forced
import os
# (ex15, small3d: small plans -- blocks of <= 8 right-hand sides go through the ONE-launch kernels, hand-off buffers used in turn)
for name, reps, subtrees in (("ex15", SOLVE_REPS, None), ("small3d", SOLVE_REPS, None),
                             ("lap30", SOLVE_REPS, None), ("lap30", SOLVE_REPS // 2, "2"), ("nd24k", SOLVE_REPS, None),
                             ("parabolic_fem", SOLVE_REPS // 2, None)):
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    if subtrees is not None:
        os.environ["PARSY_SUBTREES"] = subtrees
    plan = api.Plan(sym, 0)
    os.environ.pop("PARSY_SUBTREES", None)
    L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
    torch.cuda.synchronize()
    rng = np.random.default_rng(3)
    for nrhs in (1, 5, 64):
        b = torch.from_numpy(rng.standard_normal(sym.n * nrhs)).to(dev)
        x = b.clone(); plan.solve_device(L.data_ptr(), x.data_ptr(), nrhs, sym.n, 0); torch.cuda.synchronize()
        xf = x.clone()
        y = b.clone(); plan.backsolve_device(L.data_ptr(), y.data_ptr(), nrhs, sym.n, 0); torch.cuda.synchronize()
        yb = y.clone()
        torch.cuda.synchronize()
        worst = 0.0; bmis = 0; stat = 0
        n_rounds = reps if nrhs == 1 else (reps // 5 if nrhs == 5 else reps // 20)
        for i in range(n_rounds):
            x.copy_(b); plan.solve_device(L.data_ptr(), x.data_ptr(), nrhs, sym.n, 0)
            torch.cuda.synchronize()
            stat += plan.solve_status() != 0
            y.copy_(b); plan.backsolve_device(L.data_ptr(), y.data_ptr(), nrhs, sym.n, 0)
            torch.cuda.synchronize()
            stat += plan.solve_status() != 0
            worst = max(worst, float((x - xf).abs().max() / xf.abs().max()))
            bmis += not bool(torch.equal(y, yb))
        print(f"{name:13s} subtrees {subtrees or 'auto':4s} nrhs {nrhs}: {n_rounds} rounds, forward max rel deviation {worst:.2e}, "
              f"backward mismatches {bmis}, bad status {stat}", flush=True)
        bad += bmis + stat + (worst > 1e-12)
# two plans of a small input, their ONE-launch solves in flight together on two streams (persistent workgroups of two
# launches share the device: each waits only for earlier tickets of its own launch)
for name in ("ex15", "small3d"):
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    plans = [api.Plan(sym, 0), api.Plan(sym, 0)]
    assert plans[0].info["solve_one"] == 3
    Ls = [torch.empty(int(sym.xsize), dtype=torch.float64, device=dev) for _ in plans]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for p, Lq in zip(plans, Ls):
        p.factor_device(values.data_ptr(), Lq.data_ptr(), 0)
    torch.cuda.synchronize()
    rng = np.random.default_rng(4)
    b = torch.from_numpy(rng.standard_normal(sym.n)).to(dev)
    ref_f = b.clone(); plans[0].solve_device(Ls[0].data_ptr(), ref_f.data_ptr(), 1, sym.n, 0)
    ref_b = b.clone(); plans[0].backsolve_device(Ls[0].data_ptr(), ref_b.data_ptr(), 1, sym.n, 0)
    torch.cuda.synchronize()
    xs = [b.clone(), b.clone()]
    worst = 0.0; bmis = 0; stat = 0
    for i in range(SOLVE_REPS):
        back = i & 1
        for p, Lq, x, st_ in zip(plans, Ls, xs, streams):
            x.copy_(b)
        torch.cuda.synchronize()
        for p, Lq, x, st_ in zip(plans, Ls, xs, streams):
            (p.backsolve_device if back else p.solve_device)(Lq.data_ptr(), x.data_ptr(), 1, sym.n, st_.cuda_stream)
        torch.cuda.synchronize()
        for p, x in zip(plans, xs):
            stat += p.solve_status() != 0
            if back:
                bmis += not bool(torch.equal(x, ref_b))
            else:
                worst = max(worst, float((x - ref_f).abs().max() / ref_f.abs().max()))
    print(f"{name:13s} two ONE-launch solves in flight: {SOLVE_REPS} rounds, forward max rel deviation {worst:.2e}, "
          f"backward mismatches {bmis}, bad status {stat}", flush=True)
    bad += bmis + stat + (worst > 1e-12)
print("SOAK", "OK" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
