"""Host-side (no GPU): the BIG entries that are THIN (windows hold at most a share of 128 x 128) -- how many, what they
cost k_chol_big (16-k chunks) and what pieces of PR x PC of the source's rows would cost one wave each (k steps of
four: operand loads and matrix instructions).  Usage: thin_stats.py [WORKLOAD] [FILL_PERCENT]"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

os.environ["PARSY_THIN_FILL"] = "0"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from parsy_bench_amd import _native as N, inspector as I, matrices as M  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "flan"
fill = int(sys.argv[2]) if len(sys.argv) > 2 else 25
A, perm = M.workload(name)
sym = I.analyze(A, perm)
lib = N.lib()
lib.parsy_plan_from_symbolic.restype = C.c_void_p
h = lib.parsy_plan_from_symbolic(sym._handle, -1)
assert h, N.last_error()
lib.parsy_debug_big_entries.restype = C.c_int64
lib.parsy_debug_big_entries.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
n = lib.parsy_debug_big_entries(h, None, 0)
out = np.zeros((n, 7), dtype=np.int32)
lib.parsy_debug_big_entries(h, out.ctypes.data, n)
task, launch, K, mi, nj, ident, dij = out.T.astype(np.int64)
chunks = (K + 15) // 16
full = (mi * nj >= 0.70 * 16384) & (K >= 8)
rag = ~full
thin = rag & (mi * nj <= fill * 16384 // 100)
print(f"{name}: entries {n}, chunks {chunks.sum():.3e}; ragged entries {rag.sum()} chunks {chunks[rag].sum():.3e};"
      f" thin(<= {fill} %) entries {thin.sum()} chunks {chunks[thin].sum():.3e}")
print(f"  thin: K mean {K[thin].mean():.1f} median {np.median(K[thin]):.0f}; mi mean {mi[thin].mean():.1f} nj mean {nj[thin].mean():.1f}")
flops = 2.0 * K * mi * nj
print(f"  flops (rectangles): ragged {flops[rag].sum():.3e}, thin {flops[thin].sum():.3e}, ragged-not-thin {flops[rag & ~thin].sum():.3e}")
tasks_with = len(np.unique(task[thin]))
print(f"  tasks with thin entries: {tasks_with}; thin entries per such task: mean {thin.sum() / tasks_with:.1f}")
kst = (K + 3) // 4
for PR, PC in ((32, 64), (32, 32), (16, 64), (64, 32), (16, 128), (32, 128)):
    m, nn, ks = mi[thin], nj[thin], kst[thin]
    npr, npc = -(-m // PR), -(-nn // PC)
    # pieces: full ones and the remainders
    pieces = npr * npc
    # fragments per piece row/col: count over the grid
    def frag_sum(ext, P):
        nfull, rem = ext // P, ext % P
        return nfull * (P // 16) + (-(-rem // 16)), nfull + (rem > 0)
    fr_tot, _ = frag_sum(m, PR)
    fc_tot, _ = frag_sum(nn, PC)
    mf = fr_tot * fc_tot * ks              # matrix instructions
    loads = (fr_tot * npc + fc_tot * npr) * ks   # operand fragment loads
    # per-task serial time model: per piece 8 steps of latency overhead
    print(f"  pieces {PR:3d} x {PC:3d}: {pieces.sum():9d} pieces, mfma {mf.sum():.3e} ({mf.sum() * 64 / 2.4e9 / 1024 * 1e3:.1f} ms of 1024 SIMDs),"
          f" loads {loads.sum():.3e} ({loads.sum() * 512 / 1e9:.1f} GB), loads / mfma {loads.sum() / mf.sum():.2f}")
# shape histogram of 32 x 64 pieces
m, nn, ks = mi[thin], nj[thin], kst[thin]
hist = {}
for r0 in range(0, 128, 32):
    for c0 in range(0, 128, 64):
        pm = np.clip(m - r0, 0, 32)
        pn = np.clip(nn - c0, 0, 64)
        ok = (pm > 0) & (pn > 0)
        nf = ((pm + 15) // 16)[ok] * 10 + ((pn + 15) // 16)[ok]
        for key in np.unique(nf):
            sel = nf == key
            a = hist.setdefault(int(key), [0, 0])
            a[0] += int(sel.sum())
            a[1] += int(ks[ok][sel].sum())
print("  32 x 64 pieces by (row fragments, column fragments): count, k steps")
for key in sorted(hist):
    print(f"    {key // 10} x {key % 10}: {hist[key][0]:9d} {hist[key][1]:11d}")
