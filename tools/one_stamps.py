"""On the GPU box, diagnostic build -DPARSY_ONESTAMPS (tools/build_variant.sh onestamps -DPARSY_ONESTAMPS;
PARSY_LIB=tools/libparsy_onestamps.bin): where the ONE-launch forward solve's time goes -- per supernode the wall clock
at start / ready to gather / gathered / solved / published, and the chain of last-arriving descendants from the root
down.  Usage: one_stamps.py [WORKLOAD]"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, _native as N, inspector as I, matrices as M
name = sys.argv[1] if len(sys.argv) > 1 else "ex15"
A, perm = M.workload(name)
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
assert plan.info["solve_one"] & 1
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
b = torch.empty(sym.n, dtype=torch.float64, device=dev)
plan.rhs_ones_device(L.data_ptr(), b.data_ptr(), 0)
x = b.clone()
for _ in range(5):
    x.copy_(b)
    plan.solve_device(L.data_ptr(), x.data_ptr(), 1, sym.n, 0)
torch.cuda.synchronize()
st = np.zeros((4096, 8), dtype=np.uint64)
N.lib().parsy_debug_onestamps(C.c_void_p(st.ctypes.data))
sup = np.asarray(sym.super)
iptr = np.asarray(sym.i_ptr).astype(np.int64)
rows = np.asarray(sym.s)
col2sup = np.asarray(sym.col2Sup)
# the blocks in ticket order (schedule.cpp: build_solve_one): level by level, block columns left to right
blocks = []      # (supernode, cb, wbk)
blk_of_col = np.full(sym.n, -1, dtype=np.int64)
for tsn in np.asarray(sym.levelSet)[: sym.nsuper]:
    c0, c1 = int(sup[tsn]), int(sup[tsn + 1])
    for cb in range(0, c1 - c0, 64):
        wbk = min(64, c1 - c0 - cb)
        blk_of_col[c0 + cb: c0 + cb + wbk] = len(blocks)
        blocks.append((int(tsn), cb, wbk))
nb = len(blocks)
st = st[:nb, :5].astype(np.int64)
t0 = st[:, 0].min()
us = (st - t0) / 100.0
print(f"{name}: {sym.nsuper} supernodes, {nb} blocks; first start 0, last published {us[:, 4].max():.2f} us; starts spread over {us[:, 0].max():.2f} us")
feeds = [set() for _ in range(nb)]
for p, (tsn, cb, wbk) in enumerate(blocks):
    c0, c1 = int(sup[tsn]), int(sup[tsn + 1])
    r = int(iptr[c1] - iptr[c0]) if c1 < sym.n else int(sym.ssize - iptr[c0])
    rr = rows[int(iptr[c0]): int(iptr[c0]) + r]
    for col in rr[cb + wbk:]:
        feeds[int(blk_of_col[col])].add(p)
t = int(np.argmax(us[:, 4]))
print("   blk  sn  width  start  ready  gathered  solved  published | ready->gathered (wait), gathered->solved, solved->published")
while True:
    tsn, cb, wbk = blocks[t]
    print(f"  {t:4d} {tsn:4d} {wbk:5d} {us[t, 0]:7.2f} {us[t, 1]:7.2f} {us[t, 2]:8.2f} {us[t, 3]:7.2f} {us[t, 4]:9.2f} |"
          f" {us[t, 2] - us[t, 1]:6.2f} {us[t, 3] - us[t, 2]:6.2f} {us[t, 4] - us[t, 3]:6.2f}")
    if not feeds[t]:
        break
    d = max(feeds[t], key=lambda q: us[q, 4])
    print(f"        last feeding block {d} published at {us[d, 4]:.2f}: gathered {us[t, 2] - us[d, 4]:.2f} us later")
    t = d
