"""Sum the counters of a rocprofv3 --pmc run per kernel (one JSON object per kernel: launches, counter sums).
Usage: python tools/pmc_summary.py COUNTER_COLLECTION.csv [more.csv ...] > summary.json"""
import csv
import json
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("parsy::", "")
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(name, r["Counter_Name"])].add((path, r["Dispatch_Id"]))
out = {}
for name, c in tot.items():
    out[name] = {k: v for k, v in c.items()}
    out[name]["launches"] = max(len(disp[(name, k)]) for k in c)
json.dump(out, sys.stdout, indent=1)
