"""Per BIG launch of a profiled factorization (every launch alone on the device): time, flops, rate, and what bounds it
on paper -- the matrix-pipe time of all its tasks spread over the workgroup slots, and its longest single task.
Usage: big_rates.py [WORKLOAD]"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import _native as N, api, inspector as I, matrices as M
A, perm = M.workload(sys.argv[1] if len(sys.argv) > 1 else "flan")
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
for _ in range(2):
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
plan.profile(2)
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
plan.profile_collect()
lib = N.lib()
lib.parsy_debug_launch_times.restype = C.c_int64
lib.parsy_debug_launch_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
n = lib.parsy_debug_launch_times(plan._h, None, 0)
lt = np.zeros((n, 4))
lib.parsy_debug_launch_times(plan._h, lt.ctypes.data, n)
big = lt[lt[:, 0] == 3]
lib.parsy_debug_big_entries.restype = C.c_int64
lib.parsy_debug_big_entries.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
ne = lib.parsy_debug_big_entries(plan._h, None, 0)
ent = np.zeros((ne, 7), dtype=np.int32)
lib.parsy_debug_big_entries(plan._h, ent.ctypes.data, ne)
task, launch, K, mi, nj, ident, dij = ent.T.astype(np.int64)
ch = (K + 15) // 16
NR, NC = (mi + 15) >> 4, (nj + 15) >> 4
frb, fcb = (NR + 1) >> 1, (NC + 3) >> 2
mx = np.zeros(ne)
for wr in range(2):
    for wc in range(4):
        nfr = np.clip(NR - frb * wr, 0, frb)
        nfc = np.clip(NC - fcb * wc, 0, fcb)
        above = dij + 16 * frb * wr + 16 * nfr - 1 < 16 * fcb * wc
        mx = np.maximum(mx, np.where(above, 0, nfr * nfc))
# matrix-pipe clocks of an entry for the busiest wave: chunks x 4 k steps x fragments x 64 clocks (+ ~2300 per chunk
# of staging / barrier / bookkeeping measured in profiles/r03_big_chunk_timeline.txt is NOT included)
clk = ch * 4 * mx * 64
flops = 2.0 * K * mi * nj
ids = np.unique(launch)
rows = []
for lid in ids:
    m = launch == lid
    tk = task[m]
    ntask = len(np.unique(tk))
    tclk = np.bincount(tk - tk.min(), weights=clk[m])
    tclk = tclk[tclk > 0]
    rows.append((lid, ntask, flops[m].sum(), tclk.sum(), tclk.max()))
# a measured launch row is (kind, level << 1 | side, work items, ms): a NEXT launch (main stream) carries its targets'
# level = source level + 1, a PUSH launch (side stream) the level that waits for it = source level + 2
meas = {}
for k_, ls, cnt, ms in big:
    lev, side = int(ls) >> 1, int(ls) & 1
    meas[((lev - 2) * 2 + 1) if side else ((lev - 1) * 2)] = (int(cnt), ms)
print("launch (source level)   tasks      ms   TFLOP/s | pipe-bound ms   frac | longest task ms")
ghz = 2.4e9
tot_ms = tot_pipe = 0.0
for lid, ntask, fl, csum, cmax in rows:
    if lid not in meas:
        print("  unmatched", lid, ntask)
        continue
    cnt, ms = meas[lid]
    # a task keeps 2 waves on each of the 4 matrix pipes of its compute unit busy (and shares them with the one other
    # task resident there): pipe time of the launch = sum over tasks of (busiest wave's clocks x 2) / 256 compute units
    pipe_ms = csum * 2 / 256 / ghz * 1e3
    long_ms = cmax * 2 / ghz * 1e3
    tot_ms += ms
    tot_pipe += pipe_ms
    print(f"  {lid >> 1:3d} {'push' if lid & 1 else 'next'}  {ntask:7d} {ms:8.3f} {fl / ms / 1e9:8.1f} | {pipe_ms:8.3f}  {pipe_ms / ms:5.2f} | {long_ms:8.3f}")
print("sum ms", tot_ms, "pipe-bound sum", tot_pipe, "frac", tot_pipe / tot_ms)
