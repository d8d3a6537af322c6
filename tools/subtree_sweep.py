"""Diagnostic: factorization / forward / backward solve times of a workload for several values of PARSY_SUBTREES
(subtrees aimed at per compute unit; 0 = level launches only).  The factor and both solves of every setting are
compared with those of the first one.
Usage: python tools/subtree_sweep.py WORKLOAD[,WORKLOAD...] 0 2 4 8 ..."""
import os
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M

dev = torch.device("cuda", 0)
for name in sys.argv[1].split(","):
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
    ref = None
    for cfg in sys.argv[2:]:
        os.environ["PARSY_SUBTREES"] = cfg
        plan = api.Plan(sym, 0)
        info = plan.info
        fms = []
        for _ in range(6):
            plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
            torch.cuda.synchronize()
            fms.append(plan.last_factor_ms())
        out = {"L": L.clone() if sym.xsize < 5e8 else L[-2000000:].clone()}
        for nrhs in (1, 64):
            b = torch.zeros(nrhs * sym.n, dtype=torch.float64, device=dev)
            plan.rhs_ones_device(L.data_ptr(), b.data_ptr(), 0)
            if nrhs > 1:
                b.view(nrhs, sym.n)[1:] = b[:sym.n] * torch.arange(2, nrhs + 1, device=dev, dtype=torch.float64)[:, None]
            sms, bms = [], []
            for _ in range(4):
                x = b.clone()
                plan.solve_device(L.data_ptr(), x.data_ptr(), nrhs, sym.n, 0)
                torch.cuda.synchronize()
                sms.append(plan.last_solve_ms())
                fw = x.clone()
                plan.backsolve_device(L.data_ptr(), x.data_ptr(), nrhs, sym.n, 0)
                torch.cuda.synchronize()
                bms.append(plan.last_solve_ms())
            out[f"fw{nrhs}"], out[f"bw{nrhs}"] = fw, x
            out[f"t{nrhs}"] = (min(sms[1:]), min(bms[1:]))
            # forward solve of b = L 1 (scaled per column) gives x = 1, 2, ...
            want = torch.arange(1, nrhs + 1, device=dev, dtype=torch.float64)[:, None].expand(nrhs, sym.n)
            out[f"e{nrhs}"] = float((fw.view(nrhs, sym.n) - want).abs().max())
        if ref is None:
            ref = out
        diff = {k: float((out[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in ("L", "fw1", "bw1", "fw64", "bw64")}
        print(f"{name} subtrees/CU={cfg}: factor {min(fms[1:]):.3f} ms | solve fw/bw nrhs=1 {out['t1'][0]:.3f}/{out['t1'][1]:.3f} "
              f"nrhs=64 {out['t64'][0]:.3f}/{out['t64'][1]:.3f} ms | status {plan.status()} {plan.solve_status()} | "
              f"launches chol {info['chol_launches']} fw {info['solve_launches']} bw {info['backsolve_launches']} | "
              f"subtrees chol {info['chol_subtrees']} ({info['chol_subtree_supernodes']} sn) solve {info['solve_subtrees']} "
              f"({info['solve_subtree_supernodes']} sn) | fw err {out['e1']:.1e} {out['e64']:.1e} | "
              f"diff vs first " + " ".join(f"{k} {v:.1e}" for k, v in diff.items()), flush=True)
        del plan
