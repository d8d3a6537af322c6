"""Print the kernel timeline of the LAST factorization found in a rocprofv3 --kernel-trace csv
(start offset, duration, gap to the previous kernel end on any stream, grid, name)."""
import csv
import sys

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
ks = []
for r in rows:
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
               int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", 256) or 256),
               r.get("Stream_Id", r.get("Queue_Id", "?"))))
ks.sort()
# a factorization starts with k_scatter_a
starts = [i for i, k in enumerate(ks) if "k_scatter_a" in k[2]]
if not starts:
    sys.exit("no k_scatter_a in trace")
SOLVE = len(sys.argv) > 2 and sys.argv[2] == "solve"
b = starts[-1]
e = len(ks)
if SOLVE:  # the last forward + backward solve instead: from the last k_diag_inverse on
    # (both solves start with k_diag_inverse: take the last one that a forward-solve kernel follows)
    dinv = [i for i, k in enumerate(ks) if "k_diag_inverse" in k[2]]
    fwd = [i for i in dinv if i + 1 < len(ks) and ("k_solve" in ks[i + 1][2] or "k_transpose_x" in ks[i + 1][2])]
    b = (fwd or dinv)[-1]
t0 = ks[b][0]
prev_end = t0
tot = {}
for s, en, name, grid, wg, q in ks[b:e]:
    short = name.split("(")[0].replace("parsy::", "").replace("void ", "")
    if "solve" in short and not SOLVE:
        break
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(en - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  wgs {grid // max(wg, 1):6d}  q={q}  {short}")
    prev_end = max(prev_end, en)
    tot[short] = tot.get(short, 0) + (en - s) / 1e3
print("total span us", (prev_end - t0) / 1e3)
for k, v in sorted(tot.items(), key=lambda x: -x[1]):
    print(f"  {k}: {v:.1f} us")
