"""What the sharded solves (multigpu.LeveledShardedSolve / ShardedSolve) would take with every rank on a device of its
own, measured on ONE device: the ranks' plans take turns, so each rank's launches run alone and are timed (the plan's
own hipEvents, median of RUNS) --
  sub[r]       = rank r's subtrees (parsy_solve_device / parsy_backsolve_device with its mask)
  top[r][l]    = rank r's supernodes of level l above the cut (parsy_solve_levels_device, one level per call)
  leveled      = max_r sub[r] + sum_l (max_r top[r][l] + COLLECTIVE_US) + COLLECTIVE_US      (the final reduce of x)
  root form    = max_r sub[r] + COLLECTIVE_US + (all supernodes above the cut on rank 0) + COLLECTIVE_US
against the single-device solve.  COLLECTIVE_US is an ASSUMED latency of one small all-reduce / reduce over xGMI (the
messages are kilobytes); nothing here is a multi-GPU measurement.  The factor is the single-device one (the distributed
factor is bitwise the same).  Usage: python tools/mg_solve_model.py WORKLOAD N[,N...] [NRHS]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
COLLECTIVE_US = 30.0
RUNS = 5
name = sys.argv[1]
ranks = [int(v) for v in sys.argv[2].split(",")]
nrhs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
A, perm = M.workload(name)
sym = I.analyze(A, perm)
dev = torch.device("cuda", 0)
plan = api.Plan(sym, 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
assert plan.status() == 0
del values
B = torch.ones(sym.n * nrhs, dtype=torch.float64, device=dev)
X = torch.empty_like(B)


def med(fn):
    t = []
    for i in range(RUNS + 1):
        X.copy_(B)
        fn()
        torch.cuda.synchronize()
        if i:
            t.append(plan.last_solve_ms())
    return float(np.median(t))


t1f = med(lambda: plan.solve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0))
t1b = med(lambda: plan.backsolve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0))
print(f"{name}, {nrhs} right-hand side(s): single device forward {t1f:.3f} ms, backward {t1b:.3f} ms "
      f"(assumed per collective: {COLLECTIVE_US:.0f} us)", flush=True)
pieces = plan.pieces()
level_of = plan.solve_levels()
first = np.concatenate([[True], np.diff(pieces["supernode"]) != 0])
w = np.diff(sym.super).astype(np.int64)
r = np.diff(sym.i_ptr[sym.super].astype(np.int64))
for nr in ranks:
    D = api.Dist(plan, nr)
    sn_owner, below = D.owner[first], D.in_subtree[first]
    above = below == 0
    top_levels = sorted(set(level_of[above].tolist()))
    sub_f, sub_b = [], []
    top_f = np.zeros((nr, len(top_levels)))
    top_b = np.zeros((nr, len(top_levels)))
    held = []
    for rk in range(nr):
        mask = ((below == 1) & (sn_owner == rk)).astype(np.uint8)
        plan.set_active(mask)
        sub_f.append(med(lambda: plan.solve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)))
        sub_b.append(med(lambda: plan.backsolve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)))
        tmask = (above & (sn_owner == rk)).astype(np.uint8)
        held.append(float(((mask | tmask).astype(bool) * w * r).sum() * 8 / 1e9))
        plan.set_active(tmask)
        for backward, out in ((False, top_f), (True, top_b)):
            for i, lev in enumerate(top_levels):
                if not (tmask.astype(bool) & (level_of == lev)).any():
                    continue
                # (each level as a solve of its own: FIRST | LAST -- the arming of the hand-off buffer is in every step)
                out[rk, i] = med(lambda: plan.solve_levels_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0, lev, lev + 1,
                                                                   True, True, backward))
    plan.set_active(above.astype(np.uint8))
    root_f = med(lambda: plan.solve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0))
    root_b = med(lambda: plan.backsolve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0))
    plan.set_active(None)
    c = COLLECTIVE_US / 1e3
    lev_f = max(sub_f) + (top_f.max(axis=0) + c).sum() + c
    lev_b = max(sub_b) + (top_b.max(axis=0) + c).sum() + c
    rf = max(sub_f) + c + root_f + c
    rb = max(sub_b) + c + root_b + c
    print(f"N={nr}: supernodes above the cut {int(above.sum())} on {len(top_levels)} levels, solved by "
          f"{len(set(sn_owner[above].tolist()))} ranks; panels held per rank (GB) {np.round(held, 2).tolist()} "
          f"(root-rank form: rank 0 holds {float((above * w * r).sum() * 8 / 1e9 + held[0]) - float(((above & (sn_owner == 0)) * w * r).sum() * 8 / 1e9):.2f})\n"
          f"   forward : subtrees per rank {np.round(sub_f, 3).tolist()} ms; levels above the cut (max over ranks) "
          f"{np.round(top_f.max(axis=0), 3).tolist()} ms -> leveled {lev_f:.3f} ms = {t1f / lev_f:.2f}x of one device; "
          f"root-rank form {rf:.3f} ms = {t1f / rf:.2f}x (above the cut on rank 0: {root_f:.3f})\n"
          f"   backward: subtrees per rank {np.round(sub_b, 3).tolist()} ms; levels {np.round(top_b.max(axis=0), 3).tolist()} ms "
          f"-> leveled {lev_b:.3f} ms = {t1b / lev_b:.2f}x; root-rank form {rb:.3f} ms = {t1b / rb:.2f}x "
          f"(above the cut on rank 0: {root_b:.3f})", flush=True)
