"""Diagnostic: time the factorization of a workload under several environment settings that are read when a
plan is built (schedule knobs) -- all plans in ONE process, timed in interleaved rounds, plus the per-kind
launch times of a profiled (serialised) run and a bitwise comparison of the factors.
Usage: python tools/envsweep.py WORKLOAD "K=V,K=V" ["K=V" ...]   ('-' = no setting)"""
import os
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
name = sys.argv[1]
rounds = int(os.environ.get("SWEEP_ROUNDS", "3"))
reps = int(os.environ.get("SWEEP_REPS", "3"))
A, perm = M.workload(name)
sym = I.analyze(A, perm)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
plans = []
for cfg in sys.argv[2:]:
    kv = [x.split("=", 1) for x in cfg.split(",") if "=" in x]
    for k, v in kv:
        os.environ[k] = v
    plans.append((cfg, api.Plan(sym, 0)))
    for k, _ in kv:
        del os.environ[k]
times = {cfg: [] for cfg, _ in plans}
ref = None
same = {}
for rnd in range(rounds):
    for cfg, plan in plans:
        for _ in range(reps):
            plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
            torch.cuda.synchronize()
            times[cfg].append(plan.last_factor_ms())
        if rnd == 0:
            chk = L[-400000:].clone()
            if ref is None:
                ref = chk
            same[cfg] = bool(torch.equal(chk, ref))
            assert plan.status() == 0, (cfg, plan.status())
for cfg, plan in plans:
    plan.profile(2)
    for _ in range(2):
        plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
        torch.cuda.synchronize()
        plan.profile_collect()
    pf = plan.profile_get()
    plan.profile(0)
    kinds = {k: round(v / pf["runs"], 2) for k, v in pf["ms"].items() if v > 0}
    info = plan.info
    t = np.array(times[cfg][1:])
    print(f"{name} [{cfg}]: median {np.median(t):.2f} min {t.min():.2f} ms | serialized kinds {kinds} | "
          f"launches {info['chol_launches']} big_tasks {info['big_tasks']} | tail bitwise = first: {same[cfg]}", flush=True)
