"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (calibration binary + bench.py) into the HBM
traffic of the tile kernel per launch.  Usage:
  python tools/pmc_traffic.py CALIB_STDOUT.json CALIB_FETCH.csv CALIB_WRITE.csv BENCH_FETCH.csv BENCH_WRITE.csv N_FACT OUT.json
The counters' unit and their bias for the tile kernel's access shapes (8-byte lanes, 128-byte segments;
8-byte write-through stores) are taken from the calibration kernels, which move a known byte count
(MI355X_MICROARCH.md, HBM section: calibrate on a known byte count in your own access pattern)."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    return tot, cnt


def pick(d, key):
    keys = key if isinstance(key, tuple) else (key,)
    return sum(v for k, v in d.items() if any(q in k for q in keys))


calib = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
cf, cfn = per_kernel(sys.argv[2], "FETCH_SIZE")
cw, cwn = per_kernel(sys.argv[3], "WRITE_SIZE")
bf, bfn = per_kernel(sys.argv[4], "FETCH_SIZE")
bw, bwn = per_kernel(sys.argv[5], "WRITE_SIZE")
nfact = int(pick(bfn, "k_scatter_a")) or int(sys.argv[6])  # one A scatter per factorization

bytes_per_fetch_8 = calib["calib_read8_mfma_bytes"] * pick(cfn, "calib_read8_mfma") / pick(cf, "calib_read8_mfma")
bytes_per_fetch_16 = calib["calib_read16_bytes"] * pick(cfn, "calib_read16") / pick(cf, "calib_read16")
bytes_per_write_8 = calib["calib_write8_sc1_bytes"] * pick(cwn, "calib_write8_sc1") / pick(cw, "calib_write8_sc1")
out = {
    "workload": sys.argv[8] if len(sys.argv) > 8 else "nd24k",
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 (one pass per counter; tools/collect_pmc.sh)",
    "calibration": {
        "bytes_per_FETCH_SIZE_count_8B_lanes_128B_segments": bytes_per_fetch_8,
        "bytes_per_FETCH_SIZE_count_16B_lanes_contiguous": bytes_per_fetch_16,
        "bytes_per_WRITE_SIZE_count_8B_sc1_stores": bytes_per_write_8,
    },
    "factorizations_in_the_profiled_run": nfact,
    "kernels": {},
}
# k_chol_tiles and k_chol_chain are the two entry points of the same tile kernel (tile_task<>)
for name, key in (("k_chol_tiles", ("k_chol_tiles", "k_chol_chain")), ("k_chol_small", "k_chol_small"),
                  ("k_scatter_a", "k_scatter_a")):
    launches = pick(bfn, key)
    if not launches:
        continue
    rd = pick(bf, key) * bytes_per_fetch_8
    wr = pick(bw, key) * bytes_per_write_8
    out["kernels"][name] = {
        "launches_per_factorization": launches / nfact,
        "read_bytes_per_factorization": rd / nfact,
        "write_bytes_per_factorization": wr / nfact,
        "hbm_bytes_per_launch": (rd + wr) / launches,
    }
json.dump(out, open(sys.argv[7], "w"), indent=1)
print(json.dumps(out, indent=1))
