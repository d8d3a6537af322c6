"""Diagnostic: the distributed factorization with N ranks sharing device 0 (parsy_mg), compared bitwise with the
single-device factor; per-rank device time of a run where the ranks do not wait for each other's kernels
(they share the device, so the times are not a scaling measurement -- tools/mg_rank_times.py is).
Usage: python tools/mg_emulate.py WORKLOAD N [block] ["K=V,K=V"]"""
import os
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import api, inspector as I, matrices as M
name, nr = sys.argv[1], int(sys.argv[2])
block = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for kv in (sys.argv[4].split(",") if len(sys.argv) > 4 else []):
    k, v = kv.split("=")
    os.environ[k] = v
A, perm = M.workload(name)
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
ref, sec = plan.factor(sym.A2x)
assert plan.status() == 0
print(f"{name}: single device {sec * 1e3:.2f} ms, pieces {plan.info['n_pieces']} levels {plan.info['chol_levels']}", flush=True)
del plan
t0 = time.time()
mg = api.MultiDevice(sym, [0] * nr, block)
i = mg.dist.info
print(f"N={nr} block={block}: create {time.time() - t0:.1f}s subtrees {i['n_subtrees']} root pieces {i['n_root_pieces']} "
      f"max rank share {i['max_rank_cost'] / i['total_cost']:.3f} messages {i['n_messages']} "
      f"exchange {i['exchange_elements'] * 8 / 1e9:.2f} GB", flush=True)
mg.set_values(sym.A2x)
for rep in range(3):
    st, s = mg.factor()
    print(f"  run {rep}: status {st} wall {s * 1e3:.2f} ms rank ms {np.round(mg.rank_ms(), 2).tolist()}", flush=True)
got = mg.gather()
print("bitwise equal to the single-device factor:", bool(np.array_equal(got, ref)),
      "max abs diff", float(np.abs(got - ref).max()))
