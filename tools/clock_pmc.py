"""Shader clock and MFMA-pipe occupancy per kernel from a rocprofv3 run with --pmc GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace: clock = GRBM_GUI_ACTIVE / kernel duration.
Usage: clock_pmc.py DIR/p_counter_collection.csv DIR/p_kernel_trace.csv"""
import csv, sys
from collections import defaultdict
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = defaultdict(lambda: defaultdict(float))
seen = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("parsy::", "")
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen[name]:
        seen[name].add(r["Dispatch_Id"])
        acc[name]["ns"] += dur.get(r["Dispatch_Id"], (0, ""))[0]
print("kernel launches ms clock_GHz mfma_busy_of_simd_cycles")
for name, c in sorted(acc.items(), key=lambda kv: -kv[1]["ns"]):
    if c["ns"] <= 0:
        continue
    ghz = c.get("GRBM_GUI_ACTIVE", 0) / c["ns"]
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("GRBM_GUI_ACTIVE", 1) * 128 * 8, 1)   # per XCD counters summed?
    print(f"{name:28s} {len(seen[name]):5d} {c['ns'] / 1e6:9.3f} {ghz:6.3f} raw_mfma_busy/gui {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(c.get('GRBM_GUI_ACTIVE', 1), 1):8.2f}")
