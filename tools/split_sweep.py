"""On the GPU box: factorization time of small / mid inputs against the split of the early wave streams
(PARSY_SPLIT_CHUNKS / _TARGET / _MAX).  Usage: split_sweep.py WORKLOAD [WORKLOAD ...]"""
import os, sys, time
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
    ref = None
    for ch, tg, mx in ((192, 128, 4), (96, 64, 4), (64, 32, 8), (32, 24, 8), (32, 16, 16), (16, 12, 16)):
        os.environ["PARSY_SPLIT_CHUNKS"], os.environ["PARSY_SPLIT_TARGET"], os.environ["PARSY_SPLIT_MAX"] = str(ch), str(tg), str(mx)
        plan = api.Plan(sym, 0)
        for _ in range(5):
            plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 50 * 1e3
        lv = L.cpu().numpy().copy()
        if ref is None:
            ref = lv
        print(f"{name} split > {ch} chunks into parts of {tg} (max {mx}): {ms:.4f} ms per factorization, status {plan.status()}, launches {plan.info['chol_launches']},"
              f" max rel diff vs first {np.abs(lv - ref).max() / np.abs(ref).max():.2e}")
        del plan
