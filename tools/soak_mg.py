"""Soak of the distributed factorization (parsy_mg, N ranks sharing device 0): many runs, the factor collected from
the ranks compared bitwise with the single-plan factor every time (the ranks' streams run concurrently on the one
device: uneven progress of the ranks is the rule).  Usage: python tools/soak_mg.py [REPS]"""
import os
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import api, inspector as I, matrices as M
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
for name, nr, env in (("lap30", 4, {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "32"}), ("mid3d", 6, {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "16", "PARSY_DIST_MIN_SUBTREES": "16"}),
                      ("nd24k", 4, {}), ("nd24k", 3, {"PARSY_PIECE_WIDTH": "256", "PARSY_BIG_MINK": "64"})):
    os.environ.update(env)
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    plan = api.Plan(sym, 0)
    ref, _ = plan.factor(sym.A2x)
    del plan
    mg = api.MultiDevice(sym, [0] * nr)
    for k in env:
        del os.environ[k]
    mg.set_values(sym.A2x)
    t0 = time.time()
    mism = stat = 0
    for _ in range(REPS):
        st, _ = mg.factor()
        stat += st != 0
        mism += not np.array_equal(mg.gather(), ref)
    bad += mism + stat
    i = mg.dist.info
    print(f"{name} {nr} ranks {env}: {REPS} distributed factorizations, {i['n_messages']} messages each, "
          f"{i['n_root_pieces']} pieces above the cut: mismatches {mism}, bad status {stat}, {time.time() - t0:.1f} s", flush=True)
    mg.close()
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
