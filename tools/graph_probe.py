"""Probe: one factorization (and one forward solve) of a small workload replayed as a captured HIP graph (through
torch.cuda.CUDAGraph on the stream the library launches on) against the plain launch sequence.
Usage: graph_probe.py [WORKLOAD]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
name = sys.argv[1] if len(sys.argv) > 1 else "ex15"
A, perm = M.workload(name)
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
x = torch.ones(sym.n, dtype=torch.float64, device=dev)
s = torch.cuda.Stream(device=dev)
def timed(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
with torch.cuda.stream(s):
    h = s.cuda_stream
    plain_f = timed(lambda: plan.factor_device(values.data_ptr(), L.data_ptr(), h))
    Lref = L.clone()
    plain_s = timed(lambda: plan.solve_device(L.data_ptr(), x.data_ptr(), 1, sym.n, h))
    print(f"{name}: plain factorization {plain_f:.4f} ms, forward solve {plain_s:.4f} ms (wall per call, back to back)")
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            plan.factor_device(values.data_ptr(), L.data_ptr(), s.cuda_stream)
        gf = timed(g.replay)
        torch.cuda.synchronize()
        print(f"graph factorization {gf:.4f} ms; factor equal: {bool(torch.equal(L, Lref))}")
    except Exception as e:
        print("graph capture of the factorization failed:", repr(e)[:300])
    try:
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            plan.solve_device(L.data_ptr(), x.data_ptr(), 1, sym.n, s.cuda_stream)
        gs = timed(g2.replay)
        print(f"graph forward solve {gs:.4f} ms")
    except Exception as e:
        print("graph capture of the solve failed:", repr(e)[:300])
