"""Round-2 form of the HBM-traffic summary: per kernel, bytes per launch from the rocprofv3 --pmc FETCH_SIZE and
WRITE_SIZE passes (separate runs; tools/collect_r02.sh), with the counter unit calibrated on kernels that move a
known byte count in the same access shapes (tools/pmc_calib.hip; MI355X_MICROARCH.md, HBM: FETCH_SIZE reads half
the bytes of wide coalesced reads -- the calibration measures exactly that factor per shape).
Usage: pmc_traffic2.py CALIB_STDOUT CALIB_FETCH.json CALIB_WRITE.json RUN_FETCH.json RUN_WRITE.json WORKLOAD HASH OUT
(the .json inputs are tools/pmc_summary.py outputs)."""
import json
import sys

calib = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
cf, cw, rf, rw = (json.load(open(p)) for p in sys.argv[2:6])
workload, khash, out_path = sys.argv[6], sys.argv[7], sys.argv[8]


def unit(summary, kernel, counter, known):
    k = summary[kernel]
    return known * k["launches"] / k[counter]


u_read8 = unit(cf, "calib_read8", "FETCH_SIZE", calib["calib_read8_bytes"])
u_read8m = unit(cf, "calib_read8_mfma", "FETCH_SIZE", calib["calib_read8_mfma_bytes"])
u_read16 = unit(cf, "calib_read16", "FETCH_SIZE", calib["calib_read16_bytes"])
u_dma16 = unit(cf, "calib_dma16", "FETCH_SIZE", calib["calib_dma16_bytes"]) if "calib_dma16" in cf else u_read16
u_write8 = unit(cw, "calib_write8", "WRITE_SIZE", calib["calib_write8_bytes"])
u_write8s = unit(cw, "calib_write8_sc1", "WRITE_SIZE", calib["calib_write8_sc1_bytes"])
nfact = rf.get("k_scatter_a", {}).get("launches", 1)
out = {
    "workload": workload, "kernel_source_hash": khash,
    "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 tools/one_factor.py <workload> 2 2 "
               "(one pass per counter; tools/collect_r04.sh)",
    "calibration_bytes_per_count": {"FETCH_SIZE, 8-B lanes contiguous (k_chol_big staging, solve rows)": u_read8,
                                    "FETCH_SIZE, 8-B lanes in 128-B segments (wave streams)": u_read8m,
                                    "FETCH_SIZE, 16-B lanes contiguous": u_read16,
                                    "FETCH_SIZE, LDS-DMA 16-B lanes contiguous, 8-byte aligned (k_chol_big staging, round 3)": u_dma16,
                                    "WRITE_SIZE, plain 8-B stores": u_write8,
                                    "WRITE_SIZE, 8-B sc1 stores": u_write8s},
    "factorizations_in_the_profiled_run": nfact, "kernels": {},
}
shape = {"k_chol_big": (u_dma16, u_write8), "k_chol_dense": (u_dma16, u_write8), "k_bsolve_below": (u_read8, u_write8), "k_chol_tiles": (u_read8m, u_write8), "k_chol_chain": (u_read8m, u_write8s),
         "k_chol_small": (u_read8, u_write8), "k_scatter_a": (u_read8, u_write8)}
for name in sorted(set(rf) | set(rw)):
    if not name.startswith("k_"):
        continue
    base = name.split("<")[0]
    ur, uw = shape.get(base, (u_read8, u_write8))
    launches = max(rf.get(name, {}).get("launches", 0), rw.get(name, {}).get("launches", 0))
    rd = rf.get(name, {}).get("FETCH_SIZE", 0.0) * ur
    wr = rw.get(name, {}).get("WRITE_SIZE", 0.0) * uw
    out["kernels"][name] = {"launches_in_run": launches, "read_bytes_in_run": rd, "write_bytes_in_run": wr,
                            "hbm_bytes_per_launch": (rd + wr) / max(launches, 1)}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
