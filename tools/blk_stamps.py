"""On the GPU box, diagnostic build -DPARSY_BLKSTAMPS (tools/build_variant.sh blkstamps -DPARSY_BLKSTAMPS; PARSY_LIB=...):
the chain of block-column tasks of k_solve_blocks_mrhs in the last launch of a many-right-hand-side forward solve (the top
supernode) -- per hop: publication of X_(jb-1) -> seen by the task of jb, its products, the product with the inverse block,
its stores.  Usage: blk_stamps.py WORKLOAD NRHS"""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M, _native as N
name, nrhs = sys.argv[1], int(sys.argv[2])
A, perm = M.workload(name)
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
X = torch.ones(sym.n * nrhs, dtype=torch.float64, device=dev)
for _ in range(3):
    X.fill_(1.0)
    plan.solve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
    torch.cuda.synchronize()
print("forward solve ms", plan.last_solve_ms())
st = np.zeros(512 * 8, dtype=np.uint64)
N.lib().parsy_debug_blkstamps(C.c_void_p(st.ctypes.data))
st = st.reshape(512, 8).astype(np.int64)
nb = int((st[:, 3] > 0).sum())
us = lambda v: v / 100.0          # 100-MHz clock
hop = us(st[2:nb, 3] - st[1:nb - 1, 3])
print(f"{name} nrhs {nrhs}: {nb} block columns in the last launch's chain; hop (stores to stores) median {np.median(hop):.2f} us, mean {hop.mean():.2f}")
seg = [("published -> seen", st[2:nb, 0] - st[1:nb - 1, 3]), ("products with X_(jb-1)", st[2:nb, 1] - st[2:nb, 0]),
       ("inverse block", st[2:nb, 2] - st[2:nb, 1]), ("stores", st[2:nb, 3] - st[2:nb, 2])]
for n_, v in seg:
    print(f"   {n_:26s} median {np.median(us(v)):6.2f} us  p10 {np.percentile(us(v), 10):6.2f}  p90 {np.percentile(us(v), 90):6.2f}")
