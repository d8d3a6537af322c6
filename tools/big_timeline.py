"""Diagnostic: the phases of k_chol_big's chunk loop for one workgroup of one launch, stamped in-kernel (shader
clocks; tools/build_variant.sh bigstamps -DPARSY_BIGSTAMPS).  Usage: big_timeline.py WORKLOAD GRID BLOCK"""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import _native as N
N._LIB_PATH = ROOT / "tools" / "libparsy_bigstamps.bin"
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
A, perm = M.workload(sys.argv[1] if len(sys.argv) > 1 else "flan")
grid, block = int(sys.argv[2]), int(sys.argv[3])
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
N.lib().parsy_debug_bigstamp_cfg(grid, block)
for _ in range(2):
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
print("status", plan.status(), "ms", plan.last_factor_ms())
tr = np.zeros(2 * 8 * 1024, dtype=np.uint64)
N.lib().parsy_debug_bigtrace(tr.ctypes.data_as(C.c_void_p))
T = tr.reshape(2, 1024, 8).astype(np.int64)
for slot, wv in ((0, 0), (1, 5)):
    t = T[slot]
    nz = np.nonzero(t[:, 0])[0]
    if len(nz) == 0:
        print("wave", wv, "no stamps"); continue
    last = nz.max()
    t0 = t[0, 0]
    print(f"wave {wv}: {last + 1} chunks; columns: chunk | top-t0 | fetch mult update barrier | period | frags k/K")
    d = []
    for n in range(last + 1):
        r = t[n]
        per = (t[n + 1, 0] - r[0]) if n < last else 0
        d.append([r[2] - r[0], r[3] - r[2], r[4] - r[3], r[5] - r[4], per, r[6]])
        if n < 40 or n % 50 == 0:
            print(f"{n:4d} | {r[0] - t0:9d} | {r[2] - r[0]:6d} {r[3] - r[2]:6d} {r[4] - r[3]:6d} {r[5] - r[4]:6d} | {per:6d} | {r[6]} {r[7] & 0xffffffff}/{r[7] >> 32}")
    d = np.array(d[:-1], dtype=np.float64)
    if len(d):
        print("mean: fetch %.0f mult %.0f update %.0f barrier %.0f | period %.0f | frags %.2f" % tuple(d.mean(axis=0)))
        full = d[d[:, 5] == 8]
        if len(full):
            print("full chunks (8 fragments, %d): fetch %.0f mult %.0f update %.0f barrier %.0f | period %.0f" % ((len(full),) + tuple(full.mean(axis=0)[:5])))
