"""Host-side statistics of the tile kernel's update stream (no GPU): per tiled supernode, how many
(tile, descendant) pairs and 16-wide k chunks the schedule produces, how many useful flops a
chunk carries, and the longest tile stream per level."""
import sys

import numpy as np

sys.path.insert(0, ".")
from parsy_bench_amd import inspector, matrices  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "nd24k"
A, perm = matrices.workload(name)
sym = inspector.analyze(A, perm)
ns = sym.nsuper
sup = sym.super.astype(np.int64)
w = np.diff(sup)
iptr = sym.i_ptr.astype(np.int64)
r = iptr[sup[1:]] - iptr[sup[:-1]] if False else np.diff(iptr[sup])
rows = sym.s
lev = np.zeros(ns, dtype=np.int32)
for l in range(sym.nlevels):
    lev[sym.levelSet[sym.levelPtr[l]:sym.levelPtr[l + 1]]] = l

T = 64
tot_pairs = tot_chunks = 0
tot_flops = 0.0
hist_flops = []
per_level = {}
for s in range(ns):
    small = w[s] <= 64 and r[s] * w[s] <= 6144
    if small:
        continue
    c0 = sup[s]
    trow = rows[iptr[c0]:iptr[c0] + r[s]]  # row pattern of target
    pos = {}
    u0, u1 = sym.updPtr[s], sym.updPtr[s + 1]
    ntr, ntc = (r[s] + T - 1) // T, (w[s] + T - 1) // T
    chunks_tile = np.zeros((ntr, ntc), dtype=np.int64)
    flops_tile = np.zeros((ntr, ntc))
    for u in range(u0, u1):
        d = sym.updSn[u]
        lb, ub = sym.updLb[u], sym.updUb[u]
        K = w[d]
        drow = rows[iptr[sup[d]] + lb: iptr[sup[d]] + r[d]]
        rel = np.searchsorted(trow, drow)
        ncol = ub - lb + 1
        ti = rel // T
        cnt_i = np.bincount(ti, minlength=ntr)
        cnt_j = np.bincount(ti[:ncol], minlength=ntr)[:ntc]
        nch = (K + 15) // 16
        ii, jj = np.nonzero(np.outer(cnt_i, cnt_j))
        keep = ii >= jj
        ii, jj = ii[keep], jj[keep]
        chunks_tile[ii, jj] += nch
        f = 2.0 * K * cnt_i[ii] * cnt_j[jj]
        flops_tile[ii, jj] += f
        hist_flops.append(f / nch)
        tot_pairs += len(ii)
        tot_chunks += nch * len(ii)
        tot_flops += f.sum()
    L = int(lev[s])
    pl = per_level.setdefault(L, dict(sn=0, tiles=0, chunks=0, maxchunks=0, flops=0.0))
    pl["sn"] += 1
    pl["tiles"] += int((np.arange(ntr)[:, None] >= np.arange(ntc)[None, :]).sum())
    pl["chunks"] += int(chunks_tile.sum())
    pl["maxchunks"] = max(pl["maxchunks"], int(chunks_tile.max()))
    pl["flops"] += flops_tile.sum()

hf = np.concatenate(hist_flops)
print(f"{name}: tiled pairs={tot_pairs} chunks={tot_chunks} flops={tot_flops:.3e} "
      f"flops/chunk mean={tot_flops / tot_chunks:.0f} (full 64x64x16 chunk = 131072)")
print("flops/chunk percentiles 10/50/90:", np.percentile(hf, [10, 50, 90]))
for L in sorted(per_level):
    p = per_level[L]
    print(f"level {L}: sn={p['sn']} tiles={p['tiles']} chunks={p['chunks']} maxchunks/tile={p['maxchunks']} "
          f"flops={p['flops']:.3e} flops/chunk={p['flops'] / max(p['chunks'], 1):.0f}")
