"""On the GPU box, diagnostic build -DPARSY_SUBSTAMPS (tools/build_variant.sh substamps -DPARSY_SUBSTAMPS; PARSY_LIB=...):
where a wave of k_solve_sub_mrhs spends its time.  Usage: sub_stamps.py WORKLOAD NRHS"""
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
for _ in range(2):
    X.fill_(1.0)
    plan.solve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
    torch.cuda.synchronize()
st = np.zeros(2048 * 64, dtype=np.uint64)
N.lib().parsy_debug_substamps(C.c_void_p(st.ctypes.data))
st = st.reshape(2048, 64).astype(np.int64)
ok = st[:, 3] > 0
st = st[ok]
print("waves", len(st), " (the LAST launch of the kernel leaves its stamps: the band above the subtrees where there is one)")
tot = st[:, 3] - st[:, 0]
print(f"wave lifetime: median {np.median(tot):.0f} cycles, p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}")
print(f"start -> first loads issued + LDS zeroed: {np.median(st[:, 1] - st[:, 0]):.0f}")
print(f"members done -> end (flush): {np.median(st[:, 3] - st[:, 2]):.0f}")
names = ["top wait (loads of this member)", "issue next member's loads", "diag block -> 4x4 inverses", "x, solve products", "stores of x", "rows below (products, LDS adds)"]
for m in range(7):
    b = 8 + 8 * m
    have = st[:, b + 5] > 0
    if have.sum() < 8:
        break
    prev = st[have, 1] if m == 0 else st[have, b - 8 + 5]
    seg = [st[have, b] - prev] + [st[have, b + i] - st[have, b + i - 1] for i in range(1, 6)]
    print(f"member {m} ({have.sum()} waves): total {np.median(st[have, b + 5] - prev):.0f}: " +
          ", ".join(f"{n} {np.median(v):.0f}" for n, v in zip(names, seg)))
