"""Diagnostic: the drop-in operator (host buffers in and out) on a workload: wall time of the call against the device
time of the numeric kernels -- the PCIe-inclusive rate quoted in DESIGN.md (never bench.py's `value`)."""
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import api, inspector as I, matrices as M
name = sys.argv[1] if len(sys.argv) > 1 else "flan"
A, perm = M.workload(name)
sym = I.analyze(A, perm)
nl, levelPtr, parPtr, partition = I.trivial_hlevel(sym)
lv = np.zeros(int(sym.xsize))
timing = np.zeros(8)
for it in range(3):
    lv[:] = 0.0
    t0 = time.perf_counter()
    ok = api.cholesky_left_par_05(sym.n, sym.A2p, sym.A2i, sym.A2x, sym.p, sym.s, sym.i_ptr, lv, sym.super, sym.nsuper,
                                  timing, sym.sParent, sym.A1p, sym.A1i, sym.col2Sup, nl, levelPtr, None, 0, parPtr,
                                  partition, 1, 1, sym.maxSupWid + 1, sym.maxCol + 1)
    dt = time.perf_counter() - t0
    print(f"{name} call {it}: ok={ok} wall {dt:.3f} s  timing[0] {timing[0]:.3f} s (operator wall)  timing[2] {timing[2]:.3f} s "
          f"(device kernels)  lValues {sym.xsize * 8 / 1e9:.2f} GB", flush=True)
api.dropin_reset()
