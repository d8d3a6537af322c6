"""From a timeline (tools/timeline.py output): when is NO update kernel (k_chol_big / k_chol_dense) running, and what
runs then -- the exposed part of the chain / tile / small launches.  Usage: timeline_idle.py gpurun_out/timeline*.txt"""
import sys, re
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+) us\s+dur\s+([\d.]+)\s+gap\s+(-?[\d.]+)\s+wgs\s+(\d+)\s+q=(\S+)\s+(\S+)", l)
    if m:
        rows.append((float(m.group(1)), float(m.group(2)), m.group(6), int(m.group(4))))
end = max(s + d for s, d, n, g in rows)
upd = sorted((s, s + d) for s, d, n, g in rows if n in ("k_chol_big", "k_chol_dense"))
# union of update-kernel intervals
u = []
for a, b in upd:
    if u and a <= u[-1][1]:
        u[-1][1] = max(u[-1][1], b)
    else:
        u.append([a, b])
busy = sum(b - a for a, b in u)
print(f"span {end / 1e3:.1f} ms; an update kernel runs {busy / 1e3:.1f} ms; none runs {(end - busy) / 1e3:.1f} ms")
# the gaps, largest first, with what runs inside
gaps = []
prev = 0.0
for a, b in u + [[end, end]]:
    if a - prev > 20:
        inside = {}
        for s, d, n, g in rows:
            o = min(s + d, a) - max(s, prev)
            if o > 0 and n not in ("k_chol_big", "k_chol_dense"):
                inside[n] = inside.get(n, 0) + o
        gaps.append((a - prev, prev, inside))
    prev = max(prev, b)
gaps.sort(reverse=True)
print("gaps > 20 us:", len(gaps), "sum", round(sum(g[0] for g in gaps) / 1e3, 2), "ms")
by_third = [0, 0, 0, 0]
for g in gaps:
    by_third[min(3, int(4 * g[1] / end))] += g[0]
print("gap time by quarter of the factorization (ms):", [round(x / 1e3, 2) for x in by_third])
for g in gaps[:25]:
    print(f"  at {g[1] / 1e3:8.2f} ms: {g[0]:7.1f} us  " + ", ".join(f"{k} {v:.0f}" for k, v in sorted(g[2].items(), key=lambda x: -x[1])))
