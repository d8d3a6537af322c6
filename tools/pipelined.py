"""Experiment: throughput of independent factorizations issued on two plans / two streams alternately
(each plan has its own flags, tickets, scratch and side stream), against one after the other."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
name = sys.argv[1] if len(sys.argv) > 1 else "nd24k"
nplans = int(sys.argv[2]) if len(sys.argv) > 2 else 2
A, perm = M.workload(name)
sym = I.analyze(A, perm)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
plans = [api.Plan(sym, 0) for _ in range(nplans)]
Ls = [torch.empty(int(sym.xsize), dtype=torch.float64, device=dev) for _ in range(nplans)]
streams = [torch.cuda.Stream(device=dev) for _ in range(nplans)]
def run(k, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        j = i % k
        plans[j].factor_device(values.data_ptr(), Ls[j].data_ptr(), streams[j].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(nplans, 4)
print("one plan, sequential: %.3f ms per factorization" % run(1, 20))
print("%d plans alternating:   %.3f ms per factorization" % (nplans, run(nplans, 20)))
print("status", [p.status() for p in plans])
ref = Ls[0].clone()
print("max diff between the plans' factors", float((Ls[-1] - ref).abs().max()))
