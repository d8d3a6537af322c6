"""On the GPU box: rocprofv3 --kernel-trace --stats of a command, summed per kernel (total / calls / average).
Usage: python tools/kstats.py OUT.csv -- python3 tools/one_factor.py ..."""
import csv
import glob
import os
import shutil
import subprocess
import sys
out = sys.argv[1]
cmd = sys.argv[3:]
d = "/tmp/kstats_run"
shutil.rmtree(d, ignore_errors=True)
env = dict(os.environ, TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd,
               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
f = glob.glob(d + "/**/p_kernel_stats.csv", recursive=True)[0]
shutil.copy(f, out)
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "").replace("parsy::", "")
    if float(r["TotalDurationNs"]) > 2e4:
        print(f"{n[:64]:64s} calls {r['Calls']:>6s} total_us {float(r['TotalDurationNs']) / 1e3:10.1f} "
              f"avg_us {float(r['AverageNs']) / 1e3:9.1f}")
