"""Diagnostic: phase sums of the ordinary tiles of the chain launches (stamped build: tools/build_stamps.sh).
Usage: tile_phases.py [WORKLOAD]"""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import _native as N
N._LIB_PATH = ROOT / "tools" / "libparsy_stamps.bin"
import torch
from parsy_bench_amd import api, inspector as I, matrices as M
A, perm = M.workload(sys.argv[1] if len(sys.argv) > 1 else "flan")
sym = I.analyze(A, perm)
plan = api.Plan(sym, 0)
dev = torch.device("cuda", 0)
values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
for _ in range(2):
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
ph = np.zeros(16, dtype=np.uint64)
N.lib().parsy_debug_tilephase(ph.ctypes.data_as(C.c_void_p), 1)
plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
torch.cuda.synchronize()
print("factor ms", plan.last_factor_ms(), "status", plan.status())
N.lib().parsy_debug_tilephase(ph.ctypes.data_as(C.c_void_p), 0)
n, nb = float(ph[8]), float(ph[9])
names = ["load + descendants' stream", "own block columns", "wait for the diagonal tile", "load of the diagonal tile",
         "TRSM", "write + publish"]
print("ordinary chain tiles: %d, block columns per tile %.2f" % (n, nb / max(n, 1)))
tot = 0.0
for i, nm in enumerate(names):
    us = float(ph[i]) / 100.0 / max(n, 1)
    tot += us
    print("  %-28s %7.2f us per tile" % (nm, us))
print("  %-28s %7.2f us per tile (%.2f us per block column in phase 1)" % ("sum", tot, float(ph[1]) / 100.0 / max(nb, 1)))
