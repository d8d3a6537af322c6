"""Soak of the BIG launches (tile updates as memory-side atomic adds): inputs on which every update of the split
supernodes goes through k_chol_big with SMALL K and several sources per tile (forced with the schedule knobs: the
case in which the order of two sources' adds into one entry of L must not depend on timing), a mid-size input where the
schedule picks the BIG launches by itself, and the Flan-class input; one plan alone and two plans in flight on two
streams, every factor compared bitwise with the first.  Usage: python tools/soak_big.py [REPS]"""
import os
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
dev = torch.device("cuda", 0)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
CASES = [("lap30", {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "16"}, REPS),
         ("nd24k", {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "16"}, REPS),
         ("nd24k", {"PARSY_PIECE_WIDTH": "256", "PARSY_BIG_MINK": "64", "PARSY_PUSH_GROUP": "2"}, REPS),
         ("nd24k", {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "16", "PARSY_BIG_SUPER": "2"}, REPS),
         ("lap30", {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "16", "PARSY_BIG_SUPER": "4x2"}, REPS),
         # round 4: the dense / ragged split -- every entry through k_chol_dense (ragged windows included), the split forced
         # wherever a full block exists, and the split as the schedule takes it
         ("nd24k", {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "16", "PARSY_BIG_DENSE": "4"}, REPS),
         ("lap30", {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "32", "PARSY_BIG_DENSE": "2", "PARSY_BIG_SUPER": "2"}, REPS),
         ("nd24k", {"PARSY_PIECE_WIDTH": "256", "PARSY_BIG_MINK": "64", "PARSY_BIG_DENSE": "2"}, REPS),
         # round 5: strips riding with their dense blocks (2 x 2 super-tiles, the dense kernel wherever a full block exists)
         ("nd24k", {"PARSY_PIECE_WIDTH": "256", "PARSY_BIG_MINK": "64", "PARSY_BIG_DENSE": "2", "PARSY_BIG_SUPER": "2"}, REPS),
         ("64x64x64", {}, max(REPS // 3, 10)),
         ("flan", {}, max(REPS // 15, 5))]
for name, env, reps in CASES:
    os.environ.update(env)
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    plans = [api.Plan(sym, 0), api.Plan(sym, 0)]
    for k in env:
        del os.environ[k]
    info = plans[0].info
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    two = int(sym.xsize) * 8 * 3 < 200e9
    Ls = [torch.empty(int(sym.xsize), dtype=torch.float64, device=dev) for _ in range(2 if two else 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    plans[0].factor_device(values.data_ptr(), Ls[0].data_ptr(), streams[0].cuda_stream)
    torch.cuda.synchronize()
    chk = int(Ls[0].view(torch.int64).sum().item())
    t0 = time.time()
    mism = stat = 0
    for i in range(reps):
        nplans = 2 if (two and i % 2) else 1
        for j in range(nplans):
            plans[j].factor_device(values.data_ptr(), Ls[j].data_ptr(), streams[j].cuda_stream)
        torch.cuda.synchronize()
        for j in range(nplans):
            mism += int(Ls[j].view(torch.int64).sum().item()) != chk
            stat += plans[j].status() != 0
    bad += mism + stat
    print(f"{name} {env}: {reps} rounds (alone / two in flight alternating), big_tasks {info['big_tasks']} "
          f"big_entries {info['big_entries']} dense_entries {info['dense_entries']} strips {info['dense_strip_entries']} pieces {info['n_pieces']}: checksum mismatches {mism}, bad status {stat}, "
          f"{time.time() - t0:.1f} s", flush=True)
    del plans, Ls
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
