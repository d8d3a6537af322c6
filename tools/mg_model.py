"""What each rank of the distributed factorization would spend on a device of its own, measured on ONE device:
parsy_mg_profile lets the ranks take turns, so that every launch of every rank runs alone and is timed (hipEvents),
and the factor is compared bitwise with the single-device one.  From those times:
  busy[r]     = all launches of rank r + the copies it receives (on one device: device-to-device)
  bound_busy  = T(1 rank, same method) / max_r busy[r]              -- compute-only speed-up bound
  path        = the longest chain through the ranks' MAIN-stream launches: a rank starts level l + 1 when it has
                finished level l and every message of level l addressed to it has arrived (sender's end of level l +
                bytes / LINK_GBS: each pair of GPUs has its own link, a sender's messages of one level travel side
                by side)                                                -- critical-path estimate
  est         = max(max_r busy[r], path)
Usage: python tools/mg_model.py WORKLOAD N[,N...] [block] ["K=V,K=V"]"""
import os
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import api, inspector as I, matrices as M
LINK_GBS = 50.0   # one xGMI link, one direction, payload rate assumed for the model (7 links per GPU)
name = sys.argv[1]
ranks = [int(v) for v in sys.argv[2].split(",")]
block = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for kv in (sys.argv[4].split(",") if len(sys.argv) > 4 else []):
    k, v = kv.split("=")
    os.environ[k] = v
A, perm = M.workload(name)
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
ref, sec = plan.factor(sym.A2x)
assert plan.status() == 0
print(f"{name}: single device, overlapped streams: {sec * 1e3:.2f} ms; pieces {plan.info['n_pieces']} "
      f"levels {plan.info['chol_levels']}", flush=True)
del plan
t1 = None
for nr in [1] + [n for n in ranks if n > 1]:
    mg = api.MultiDevice(sym, [0] * nr, block)
    mg.set_values(sym.A2x)
    mg.profile()                                  # warm-up
    st, main, side, copy = mg.profile()
    same = bool(np.array_equal(mg.gather(), ref))
    busy = (main + side + copy).sum(axis=1)
    if nr == 1:
        t1 = float(busy[0])
    nl = main.shape[1]
    start = np.zeros(nr)          # when each rank can start the level at hand
    link_sum = 0.0
    for lev in range(nl):
        end = start + main[:, lev]
        nxt = end.copy()
        for (src, dst, off, ln, pk, total) in mg.dist.messages(lev):
            t = total * 8 / (LINK_GBS * 1e6)
            if end[src] + t > nxt[dst]:
                link_sum += end[src] + t - nxt[dst]
                nxt[dst] = end[src] + t
        start = nxt
    path = float(start.max())
    i = mg.dist.info
    est = max(float(busy.max()), path)
    print(f"N={nr} block={block}: status {st} bitwise = single device: {same} | busy ms {np.round(busy, 1).tolist()} "
          f"(main {np.round(main.sum(axis=1), 1).tolist()} side {np.round(side.sum(axis=1), 1).tolist()} "
          f"copies {np.round(copy.sum(axis=1), 1).tolist()}) | T1 {t1:.1f} ms bound_busy {t1 / busy.max():.2f}x | "
          f"path {path:.1f} ms (lockstep main chain {main.max(axis=0).sum():.1f}; waits for messages at {LINK_GBS:.0f} GB/s: {link_sum:.1f}) | "
          f"est {est:.1f} ms = {t1 / est:.2f}x | exchange {i['exchange_elements'] * 8 / 1e9:.2f} GB in "
          f"{i['n_messages']} messages, subtrees {i['n_subtrees']}, root pieces {i['n_root_pieces']}", flush=True)
    mg.close()
