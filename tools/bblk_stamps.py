"""On the GPU box, diagnostic build -DPARSY_BLKSTAMPS (tools/build_variant.sh blkstamps -DPARSY_BLKSTAMPS; PARSY_LIB=...):
the chain of block-column tasks of k_bsolve_chain_mrhs in the last chain launch of a many-right-hand-side BACKWARD solve
that takes that kernel -- per hop (block column jb + 1 -> jb): the task's own work before its last wait (the later blocks'
products, the reduction of the waves' parts, P and M's operands), how long it then waits for X_(jb+1), the products behind
the wait, the stores.  Usage: bblk_stamps.py WORKLOAD NRHS"""
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
    plan.backsolve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
    torch.cuda.synchronize()
print("backward solve ms", plan.last_solve_ms())
st = np.zeros(512 * 8, dtype=np.uint64)
N.lib().parsy_debug_blkstamps(C.c_void_p(st.ctypes.data))
st = st.reshape(512, 8).astype(np.int64)
ok = np.where((st[:, 3] > 0) & (st[:, 0] > 0))[0]
us = lambda v: v / 100.0          # 100-MHz clock
# hop jb+1 -> jb for consecutive stamped block columns
jbs = [j for j in ok if j + 1 in set(ok.tolist())]
jb = np.array(jbs)
print(f"{name} nrhs {nrhs}: {len(ok)} block columns stamped (the LAST launch of the kernel); hops {len(jb)}")
hop = us(st[jb, 3] - st[jb + 1, 3])
print(f"  hop (armed-buffer stores of jb + 1 -> of jb): median {np.median(hop):.2f} us, mean {hop.mean():.2f}")
seg = [("reduction of the waves' parts (4 -> 5)", st[jb, 5] - st[jb, 4]),
       ("T from LDS, P, M's operands (5 -> 6)", st[jb, 6] - st[jb, 5]),
       ("ready -> X_(jb+1) seen (6 -> 0)", st[jb, 0] - st[jb, 6]),
       ("   of which after X_(jb+1) was stored", st[jb, 0] - np.maximum(st[jb, 6], st[jb + 1, 3])),
       ("products behind the wait (0 -> 1)", st[jb, 1] - st[jb, 0]),
       ("armed-buffer stores (1 -> 3)", st[jb, 3] - st[jb, 1]),
       ("X_(jb+2) stored -> this task's later blocks done (-> 4)", st[jb[jb + 2 < 512], 4] - st[jb[jb + 2 < 512] + 2, 3])]
for n_, v in seg:
    v = us(v)
    print(f"   {n_:52s} median {np.median(v):7.2f} us  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}")
