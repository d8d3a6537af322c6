"""Per dispatch of one kernel: the counters of a rocprofv3 --pmc run (CSV) side by side, in dispatch order.
Usage: pmc_per_dispatch.py COUNTER_COLLECTION.csv KERNEL_SUBSTRING"""
import csv, sys
from collections import defaultdict, OrderedDict
rows = OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]:
        continue
    d = rows.setdefault(int(r["Dispatch_Id"]), {"grid": r.get("Grid_Size", "?"), "wg": r.get("Workgroup_Size", "?")})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({k for d in rows.values() for k in d if k not in ("grid", "wg")})
print("dispatch grid", *names, "hit_rate")
for k, d in rows.items():
    h, m = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
    print(k, int(d["grid"]) // max(int(d["wg"]), 1) if d["grid"] != "?" else "?", *[f"{d.get(n, 0):.4g}" for n in names], f"{h / max(h + m, 1):.3f}")
