"""Diagnostic: time the factorization of a workload under several schedule knobs (read when a plan is built).
Usage: python tools/sweep.py WORKLOAD "PW,G,MINK" ["PW,G,MINK" ...]   -> one line per setting"""
import os
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
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
ref = None
for cfg in sys.argv[2:]:
    pw, g, mink = cfg.split(",")
    os.environ["PARSY_PIECE_WIDTH"] = pw
    os.environ["PARSY_PUSH_GROUP"] = g
    os.environ["PARSY_BIG_MINK"] = mink
    plan = api.Plan(sym, 0)
    info = plan.info
    ms = []
    for _ in range(4):
        plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
        torch.cuda.synchronize()
        ms.append(plan.last_factor_ms())
    st = plan.status()
    chk = L[-200000:].clone()
    err = 0.0 if ref is None else float((chk - ref).abs().max() / ref.abs().max())
    if ref is None:
        ref = chk
    print(f"{name} pw={pw} g={g} mink={mink}: best {min(ms[1:]):.3f} ms  (runs {['%.2f' % m for m in ms]}) status {st} "
          f"levels {info['chol_levels']} launches {info['chol_launches']} big_tasks {info['big_tasks']} "
          f"big {info['big_flops']:.3e} wave {info['tile_update_flops']:.3e} inner {info['inner_flops']:.3e} "
          f"tail_diff {err:.1e}", flush=True)
    del plan
