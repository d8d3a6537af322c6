#!/usr/bin/env python3
"""Benchmark of the hot path: supernodal Cholesky factorizations/s (+ BCSC forward
solves/s) on a SuiteSparse-class SPD matrix, numeric phase only, inputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one numeric factorization of the workload matrix (pattern analysed and
uploaded once, outside the timed region -- the reference times only the executor call
too: examples/choleskyTest01.cpp:209-229).  After the K factorization steps, K forward
solves are timed the same way and reported as `solves_per_sec`.

N = 1 workload: the Flan_1565-class stand-in (BASELINE.json configs[2], the largest
single-GPU configuration; the SuiteSparse file itself cannot be fetched offline): 3-D
27-point stencil 116^3, geometric nested dissection, n = 1 560 896, 19.4 GB of lValues.
The nd24k-class (configs[1]) and parabolic_fem-class (configs[3]: solves with 1 / 8 / 64
right-hand sides) inputs are measured after it and reported as extra objects, never as
`value`.  N > 1: the same matrix, ONE factorization over the ranks (strong scaling, as north_star
describes it): etree subtrees below a cut on one rank each, the pieces of the separators above it
dealt over all ranks, finished pieces sent point to point (RCCL over xGMI) after every level to
the ranks that read them (csrc/dist.cpp, multigpu.py).  The N > 1 path is correct by construction (gloo tests
with 2 and 4 ranks, N ranks sharing the one device: bitwise the single-device factor) but its RCCL messages between
DISTINCT devices have not run on hardware yet -- the builder's box has one GPU; the line it prints carries
`ranks_seen` and per-rank busy times so that a reader can tell.

PyTorch is plumbing here: device buffers, the HIP stream, torch.distributed.  All
numerics run in libparsy_amd.so through its C ABI.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix, datasheet (the microarch guide lists no f64 row)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
SOLVE_TOL = 1e-9               # max|x - 1| of the solve of L x = L 1 (reference testTriangular: 1e-3)
BACK_TOL = 1e-9                # max|x - 1| of forward + backward solve of L L' x = (P A P') 1
PMC_ROUND = "r05"              # profiles/<round>_<workload>_pmc_traffic.json: the committed counter summary that is quoted


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_hash() -> str:
    """Identifies the kernels a committed PMC summary was measured on (profiles/*pmc*.json carry it)."""
    h = hashlib.sha256()
    for p in sorted((ROOT / "parsy_bench_amd" / "csrc").glob("*")):
        if p.suffix in (".hip", ".cpp", ".hpp"):
            h.update(p.read_bytes())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------
# CPU baseline: the oracle (oracle/, test infrastructure: a plain-C restatement of the reference's
# executors, -O3 -fopenmp, dense kernels bound to the image's MKL when present -- what the reference
# links) timed on this host.  Only this leg of bench.py touches oracle/.
# ---------------------------------------------------------------------------------------
def _subtree_ranges(sym):
    """first[s] = first supernode of the subtree rooted at s (supernodes are postordered: the subtree is
    the contiguous range first[s]..s); cost[s] = flops on the stored structure of that subtree."""
    ns = sym.nsuper
    first = np.arange(ns)
    par = np.asarray(sym.sParent)
    for s in range(ns):
        p = par[s]
        if p >= 0 and first[s] < first[p]:
            first[p] = first[s]
    w = np.diff(sym.super).astype(np.float64)
    r = np.diff(sym.i_ptr[sym.super].astype(np.int64)).astype(np.float64)
    own = w * r * r - w * (w - 1) * r + w * (w - 1) * (2 * w - 1) / 6.0   # sum_t (r - t)^2
    cum = np.concatenate([[0.0], np.cumsum(own)])
    return first, cum[np.arange(ns) + 1] - cum[first], float(own.sum())


def cpu_baseline(sym, threads: int, budget_s: float = 8.0):
    """Time the CPU port on a BOUNDED sample of the same matrix: one etree subtree (a complete
    factorization of its own columns, same mix of narrow and wide supernodes) sized from a calibration
    run to about `budget_s` seconds, median of 3 runs after a warm-up; the whole matrix when it fits the
    budget.  value = (executed flops of the sample / seconds) / executed flops of the whole job."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as O
    O.lib()
    blas = O.bind_system_blas()
    first, sub, total = _subtree_ranges(sym)
    ns = sym.nsuper

    def run(root, reps):
        ts = []
        for _ in range(reps):
            ok, dt = O.cholesky_wavefront_subtree(sym, sym.A2x, int(root), int(first[root]), threads=threads)
            if not ok:
                raise RuntimeError("CPU port reported a non-positive pivot")
            ts.append(dt)
        return float(np.median(ts))

    def pick(limit):
        ok = np.where(sub <= limit)[0]
        return int(ok[np.argmax(sub[ok])])

    # calibration (doubles as the warm-up of the BLAS threads and of the page cache)
    cal = pick(max(total * 1e-3, min(total, 3e10)))
    run(cal, 1)
    rate = sub[cal] / run(cal, 1)
    # The sample is the SAME subtree in every run (a sample sized from the measured rate moved from run
    # to run): the largest subtree of at most 1.2e12 executed flops (about 8 s at the 150 GFLOP/s this port reaches on
    # 16 cores), halved only while the calibration says it would take more than 4 x the budget on this host.
    limit = 1.2e12
    while limit > sub[cal] and limit / rate > 4.0 * budget_s:
        limit *= 0.5
    root = pick(max(sub[cal], limit))
    whole = sub[root] >= 0.999 * total
    secs = run(root, 3)
    sample = (f"{'the whole matrix' if whole else 'one etree subtree'}: supernodes {int(first[root])}..{root} of "
              f"{ns} = {100.0 * sub[root] / total:.2f} % of the executed flops ({sub[root]:.3e} of {total:.3e}), "
              f"median of 3 runs after a warm-up; oracle/parsy_oracle.c wavefront executor, -O3 -fopenmp, "
              f"{threads} OpenMP threads over the supernodes of a level, dense kernels: "
              f"{blas + ' (MKL_THREADING_LAYER=' + os.environ.get('MKL_THREADING_LAYER', '?') + ')' if blas else 'built-in loops'}"
              f"{'' if whole else '; value = sample flop rate / flops of the whole job (the top separators, absent from the sample, run at a higher BLAS-3 rate on a CPU too: read it as a lower bound)'}")
    out = {"value": (sub[root] / secs) / total, "unit": "factorizations/s", "cores": threads, "kind": "port",
           "sample": sample, "sample_seconds": secs, "sample_gflops": sub[root] / secs / 1e9,
           "host_cpus_visible": len(os.sched_getaffinity(0))}
    O.unbind_blas()
    return out


def cpu_baseline_top(sym, lv, threads: int, rest: dict):
    """Second sample: the TOP SEPARATOR's supernode alone -- the oracle's cholesky_left_par_05 root
    phase (one supernode, dense kernels on `threads` BLAS threads: what the reference does with its last l-level,
    parallel_PB_Cholesky_05.h:269-411) on the factor the GPU has just computed for everything below it -- and the
    flop-weighted combination with the subtree sample: seconds = flops(rest) / rate(subtree sample) + flops(top) /
    rate(top sample).  lv: the GPU's factor on the host (the top separator's panel is overwritten)."""
    import time
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as O
    blas = O.bind_system_blas()
    try:
        ns = sym.nsuper
        root = ns - 1
        c0, c1 = int(sym.super[root]), int(sym.super[root + 1])
        w = c1 - c0
        p0, p1 = int(sym.p[c0]), int(sym.xsize)
        # left-looking work of this target: DSYRK on the rows of every descendant inside its columns + its own POTRF
        u0, u1 = int(sym.updPtr[root]), int(sym.updPtr[root + 1])
        d = np.asarray(sym.updSn[u0:u1], dtype=np.int64)
        K = np.diff(sym.super)[d].astype(np.float64)
        n1 = (np.asarray(sym.updUb[u0:u1], dtype=np.float64) - np.asarray(sym.updLb[u0:u1], dtype=np.float64) + 1.0)
        f_top = float((K * n1 * (n1 + 1.0)).sum() + w ** 3 / 3.0)
        ok_t = []
        for _ in range(2):   # (a warm-up of the BLAS threads, then the measured run)
            lv[p0:p1] = 0.0
            timing = np.zeros(8 + max(threads, 1))
            arrs = [np.ascontiguousarray(a, dtype=t) for a, t in (
                (sym.A2p, np.int32), (sym.A2i, np.int32), (sym.A2x, np.float64), (sym.p, np.uint64), (sym.s, np.int32),
                (sym.i_ptr, np.uint64), (sym.super, np.int32), (sym.sParent, np.int32), (sym.A1p, np.int32),
                (sym.A1i, np.int32), (sym.col2Sup, np.int32))]
            lp, pp, pt = np.array([0, 1], np.int32), np.array([0, 1], np.int32), np.array([root], np.int32)
            O.lib().oracle_set_threads(threads)
            t0 = time.perf_counter()
            ok = O.lib().oracle_cholesky_left_par_05(
                sym.n, O.P(arrs[0]), O.P(arrs[1]), O.P(arrs[2]), O.P(arrs[3]), O.P(arrs[4]), O.P(arrs[5]), O.P(lv),
                O.P(arrs[6]), ns, O.P(timing), O.P(arrs[7]), O.P(arrs[8]), O.P(arrs[9]), O.P(arrs[10]), 1, O.P(lp), None,
                0, O.P(pp), O.P(pt), 1, threads, sym.maxSupWid + 1, sym.maxCol + 1, None)
            ok_t.append((bool(ok), time.perf_counter() - t0))
        ok, secs = ok_t[-1]
        if not ok:
            raise RuntimeError("CPU port reported a non-positive pivot in the top separator")
        total = rest["sample_gflops"] * 1e9 / rest["value"] if rest.get("value") else None   # executed flops of the whole job
        out = {"supernode": root, "width": w, "descendants": int(u1 - u0), "flops": f_top, "seconds": secs,
               "gflops": f_top / secs / 1e9,
               "sample": f"the top separator's supernode alone ({w} columns, {u1 - u0} descendants, {f_top:.3e} flops: DSYRK per "
                         f"descendant + POTRF) on the GPU's factor of everything below it; oracle cholesky_left_par_05 root phase, "
                         f"{threads} BLAS threads ({blas or 'built-in loops'}), second of two runs"}
        if total:
            est = max(total - f_top, 0.0) / (rest["sample_gflops"] * 1e9) + secs
            out["combined_value"] = 1.0 / est
            out["combined"] = ("factorizations/s = 1 / (flops outside the top separator / subtree-sample rate + measured seconds of "
                               "the top separator)")
        return out
    finally:
        O.unbind_blas()


def granted_cpus():
    """CPUs this process may really use: the cgroup CPU quota where one is set (v2 cpu.max, v1 cfs quota),
    else the affinity mask -- capped at the GPU box's documented per-GPU share of 16 CPUs when neither says
    less (the pool's process guard enforces that share; os.cpu_count() reports the whole host).  Returns
    (count, how it was found)."""
    aff = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, min(aff, int(np.ceil(int(q) / int(per))))), f"cgroup v2 cpu.max = {q} {per}"
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, min(aff, int(np.ceil(q / per)))), f"cgroup v1 cfs quota {q}/{per}"
    except (OSError, ValueError):
        pass
    if aff <= 16:
        return aff, f"affinity mask ({aff} CPUs), no cgroup quota"
    return 16, f"no cgroup quota, affinity mask has {aff} CPUs: the pool's documented share of 16 CPUs per GPU"


def cpu_baseline_ex15(threads: int = 1):
    """configs[0]: the reference's own CPU-runnable case (ex15-class), as scripts/eval.sh:5-21 times it:
    chunk = 1, costParam = threads, levelParam in {2,1,0,-1,-2}, finalSeqNode in {2,4}; per setting 5
    factorizations, the median; the best setting is reported.  The H-level partitions are the reference's own
    LBC partitioner's output, read from a committed fixture; without one for this thread count every supernode is
    its own w-partition (the factor is schedule independent)."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as O
    from parsy_bench_amd import inspector as I, matrices as M
    O.lib()
    blas = O.bind_system_blas()
    A, perm = M.workload("ex15")
    sym = I.analyze(A, perm)
    # The partitions are a committed fixture (tests/golden/lbc_ex15.npz, written by tests/golden/make_golden.py from
    # the reference's getCoarseLevelSet_6 in the build container): nothing compiled from the reference is loaded here.
    settings = []
    fixture = ROOT / "tests" / "golden" / "lbc_ex15.npz"
    if fixture.exists():
        g = np.load(fixture)
        for lev in (2, 1, 0, -1, -2):
            for fin in (2, 4):
                key = f"c{threads}_l{lev}_f{fin}"
                if key + "_levelPtr" in g:
                    settings.append(((lev, fin), (int(g[key + "_nLevels"][0]), g[key + "_levelPtr"], g[key + "_parPtr"],
                                                  g[key + "_partition"])))
    lbc = bool(settings)
    if not settings:
        settings = [((None, None), I.trivial_hlevel(sym))]
    O.cholesky_05(sym, sym.A2x, settings[0][1], threads=threads)
    sweep, best, lo = [], None, None
    for (lev, fin), hl in settings:
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            ok, lo, _ = O.cholesky_05(sym, sym.A2x, hl, threads=threads)
            ts.append(time.perf_counter() - t0)
            if not ok:
                raise RuntimeError("CPU port reported a non-positive pivot")
        med = float(np.median(ts))
        sweep.append({"levelParam": lev, "finalSeqNode": fin, "l_levels": int(hl[0]), "w_partitions": len(hl[2]) - 1,
                      "median_s": med})
        if best is None or med < best["median_s"]:
            best = sweep[-1]
    b = O.rhs_init_blocked(sym, lo)
    t0 = time.perf_counter()
    x = O.blocked_lsolve(sym, lo, b, "serial")
    ts_solve = time.perf_counter() - t0
    O.unbind_blas()
    return {"workload": "ex15-class stand-in (83x83 5-point grid, n = 6 889)", "threads": threads,
            "factorizations_per_sec": 1.0 / best["median_s"], "median_of": 5, "dense_kernels": blas or "built-in loops",
            "schedule": ("reference LBC partitions (getCoarseLevelSet_6; committed fixture tests/golden/lbc_ex15.npz), sweep of scripts/eval.sh, best of"
                         if lbc else "one w-partition per supernode (no fixture for this thread count)"),
            "best": best, "sweep": sweep,
            "solves_per_sec": 1.0 / ts_solve, "solve_max_abs_err": float(np.abs(x - 1.0).max())}


def cpu_baseline_whole(name: str, threads: int):
    """A whole matrix (no sampling) on all granted cores: the oracle's wavefront executor, median of 3 after
    a warm-up."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle as O
    from parsy_bench_amd import inspector as I, matrices as M
    O.lib()
    blas = O.bind_system_blas()
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    O.cholesky_wavefront(sym, sym.A2x, threads=threads)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ok, _, _ = O.cholesky_wavefront(sym, sym.A2x, threads=threads)
        ts.append(time.perf_counter() - t0)
        if not ok:
            raise RuntimeError("CPU port reported a non-positive pivot")
    O.unbind_blas()
    med = float(np.median(ts))
    return {"workload": f"{name}-class stand-in, the whole matrix", "threads": threads, "n": sym.n,
            "flops_F": sym.flops_colcount, "factorizations_per_sec": 1.0 / med, "median_of": 3,
            "gflops_F": sym.flops_colcount / med / 1e9, "dense_kernels": blas or "built-in loops"}


# ---------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="flan")
    ap.add_argument("--mtx", default=None, help="a MatrixMarket file (e.g. the real Flan_1565.mtx) instead of the "
                    "synthetic stand-in: lower triangle taken, ordered by --order FILE or by parsy_order_nd")
    ap.add_argument("--order", default=None, help="ordering file for --mtx (dimension, then n entries)")
    ap.add_argument("--nrhs", type=int, default=1)
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the nd24k / parabolic_fem / ex15 extra objects")
    ap.add_argument("--no-solve", action="store_true", help="factorizations only (profiling runs)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="also time this many independent factorizations in flight (extra field; 1 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the executor has no CPU fallback)")
    # one rank per GPU; a rehearsal with more ranks than GPUs (PARSY_DIST_BACKEND=gloo) shares devices
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("PARSY_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # every rank takes part in one collective before the first point-to-point exchange (only some ranks take
        # part in those: with RCCL the first call on a group must be made by all of its ranks)
        t_init = torch.zeros(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t_init)
    local_rank = dev_index

    from parsy_bench_amd import api, inspector as I, matrices as M, multigpu as MG

    stream = torch.cuda.current_stream().cuda_stream

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, warmup, steps, collective=True):
        for _ in range(warmup):
            fn()
        fence() if collective else torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        fence() if collective else torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1 and collective:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def measure_solves(plan, sym, L, nrhs, warmup, steps):
        """b = L 1 (the reference's rhsInitBlocked) repeated nrhs times; forward solves timed like the
        factorizations; every column must come back as ones."""
        b = torch.empty(sym.n, dtype=torch.float64, device=dev)
        plan.rhs_ones_device(L.data_ptr(), b.data_ptr(), stream)
        B = b.repeat(nrhs).contiguous()
        X = torch.empty_like(B)

        def solve_step():
            X.copy_(B)
            plan.solve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, stream)

        dt = timed(solve_step, warmup, steps, collective=False)
        if plan.solve_status() != 0:
            raise SystemExit(f"forward solve: a hand-off wait timed out (solve status {plan.solve_status()}, nrhs = {nrhs})")
        err = float((X - 1.0).abs().max().item())
        if not (err <= SOLVE_TOL):
            raise SystemExit(f"forward solve of L x = L 1 is off: max|x - 1| = {err:.3e} (nrhs = {nrhs})")

        def bsolve_step():
            X.copy_(B)
            plan.backsolve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, stream)

        dt_b = timed(bsolve_step, warmup, steps, collective=False)
        if plan.solve_status() != 0:
            raise SystemExit(f"backward solve: a hand-off wait timed out (solve status {plan.solve_status()}, nrhs = {nrhs})")
        # (the timed steps above include the copy that restores the right-hand side -- the solve works in place --: n x nrhs
        # doubles read and written, 0.1 ms of a 64-right-hand-side step on the parabolic_fem-class input.  Beside them: the
        # device time of the solve's own launches, hipEvents around one solve, best of three)
        dev_ms = []
        for fn in (plan.solve_device, plan.backsolve_device):
            t = []
            for _ in range(3):
                X.copy_(B)
                fn(L.data_ptr(), X.data_ptr(), nrhs, sym.n, stream)
                torch.cuda.synchronize()
                t.append(plan.last_solve_ms())
            dev_ms.append(min(t))
        measure_solves.device_ms = tuple(dev_ms)
        # the backward solve is validated once, untimed: c = (P A P') 1 from A itself (host, scipy), then
        # forward + backward solve of L L' x = c must return ones in every column
        import scipy.sparse as sp
        A2 = sp.csc_matrix((sym.A2x, sym.A2i, sym.A2p), shape=(sym.n, sym.n))
        ones = np.ones(sym.n)
        c = A2 @ ones + A2.T @ ones - A2.diagonal()
        C = torch.from_numpy(c).to(dev).repeat(nrhs).contiguous()
        plan.solve_device(L.data_ptr(), C.data_ptr(), nrhs, sym.n, stream)
        plan.backsolve_device(L.data_ptr(), C.data_ptr(), nrhs, sym.n, stream)
        torch.cuda.synchronize()
        if plan.solve_status() != 0:
            raise SystemExit(f"forward + backward solve: solve status {plan.solve_status()} (nrhs = {nrhs})")
        err_b = float((C - 1.0).abs().max().item())
        if not (err_b <= BACK_TOL):
            raise SystemExit(f"backward solve is off: max|x - 1| = {err_b:.3e} for L L' x = (P A P') 1 (nrhs = {nrhs})")
        return dt, dt_b, err, (B, X), err_b

    # ---- the headline workload -----------------------------------------------------------
    t0 = time.perf_counter()
    if args.mtx:
        # a real SuiteSparse file when it is present on the box: own reader, own ordering (the
        # reference orders with METIS, absent here) -- no extras and no committed PMC summary apply to it
        A = M.read_mtx(args.mtx)
        perm = M.read_ordering(args.order, A.n) if args.order else I.order_nd(A)
        args.workload = Path(args.mtx).stem
        args.no_extras = True
    else:
        A, perm = M.workload(args.workload)
    sym = I.analyze(A, perm)
    t_inspect = time.perf_counter() - t0
    t0 = time.perf_counter()
    plan = api.Plan(sym, local_rank)
    t_plan = time.perf_counter() - t0
    info = plan.info
    if rank == 0:
        log(f"[bench] workload={args.workload} n={sym.n} nnz(A)={sym.nnzA} nsuper={sym.nsuper} "
            f"nnz(L)={sym.nnzL} xsize={sym.xsize} F={sym.flops_colcount:.4e} "
            f"executed={sym.flops_stored:.4e} levels={sym.nlevels} (Cholesky view: {info['n_pieces']} pieces, "
            f"{info['chol_levels']} levels) maxw={sym.maxSupWid} launches/factor={info['chol_launches']} "
            f"inspect={t_inspect:.2f}s plan={t_plan:.2f}s")

    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)

    DF = engine = pieces = D = None
    if world > 1:
        # the library's distribution (parsy_dist): subtrees below a cut on one rank each, the pieces above it dealt
        # over the ranks; after every level the finished pieces travel to the ranks that read them (RCCL point to point)
        D = api.Dist(plan, world, int(os.environ.get("PARSY_DIST_BLOCK", "0")))   # 0: the library's default (2)
        if D.check(plan) != 0:
            raise SystemExit("the distribution is inconsistent: " + str(D.check(plan)))
        pieces = plan.pieces()
        plan.set_active_pieces(D.mask(rank))
        DF = MG.DistributedFactorization(D, rank, dist, dev, stage_on_host=(backend != "nccl"))
        engine = MG.PlanEngine(plan, values.data_ptr())
        if rank == 0:
            i = D.info
            log(f"[bench] distribution over {world} ranks: {i['n_subtrees']} subtrees below the cut, {i['n_root_pieces']} "
                f"pieces above it ({i['root_cost'] / i['total_cost']:.3f} of the flops), rank shares "
                f"{np.round(D.rank_cost / D.rank_cost.sum(), 3).tolist()}, {i['n_messages']} messages, "
                f"{i['exchange_elements'] * 8 / 1e9:.2f} GB travel per factorization")

    def factor_step():
        if world == 1:
            plan.factor_device(values.data_ptr(), L.data_ptr(), stream)
        else:
            DF.factor(engine, L, stream)

    # ---- factorizations ------------------------------------------------------------
    dt_f = timed(factor_step, args.warmup, args.steps)
    status = plan.status()
    if world > 1:   # every rank factors its own pieces: the first failing pivot / error of any rank
        t = torch.tensor([float(status if status > 0 else (1 << 40) if status == 0 else -1)], dtype=torch.float64,
                         device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        status = 0 if t.item() >= float(1 << 40) else int(t.item())
    if status != 0:
        raise SystemExit(f"factorization failed: status {status} (> 0: non-positive pivot at that column; "
                         f"< 0: a hand-off wait inside a launch timed out)")

    # ---- N > 1: who took part (evidence for the reader of the line: ranks seen by a collective, their devices,
    # what each rank's device spent on the last factorization, what each rank sent) -----
    multi = None
    if world > 1:
        import socket
        props = torch.cuda.get_device_properties(dev)
        ident = f"{socket.gethostname()}/cuda:{dev_index}/{getattr(props, 'uuid', '')}/{props.name}"
        mine = torch.tensor([float(rank), float(dev_index), float(int(hashlib.sha256(ident.encode()).hexdigest()[:12], 16)),
                             float(plan.last_factor_ms()), float(DF.sent_elements * 8)], dtype=torch.float64,
                            device=dev if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = [t.cpu().tolist() for t in allr]
        multi = {"ranks_seen": len({int(r[0]) for r in rows}),
                 "distinct_devices": len({(int(r[1]), int(r[2])) for r in rows}),
                 "rank_device_index": [int(r[1]) for r in rows],
                 "rank_device_ms_last_factorization": [round(r[3], 3) for r in rows],
                 "rank_bytes_sent_per_factorization": [int(r[4]) for r in rows],
                 "backend": backend + (" (= RCCL)" if backend == "nccl" else " (rehearsal: host-staged messages)"),
                 "note": "device ms = hipEvents around the rank's own launches and exchanges of the last timed factorization; "
                         "distinct_devices < ranks_seen means ranks shared a GPU (a rehearsal, not a scaling measurement)"}

    # ---- two independent factorizations in flight (reported beside `value`, never as `value`) -----
    pipelined = None
    if world == 1 and args.in_flight > 1 and int(sym.xsize) * 8 * args.in_flight < 64e9:
        plans2 = [plan] + [api.Plan(sym, local_rank) for _ in range(args.in_flight - 1)]
        Ls2 = [L] + [torch.empty_like(L) for _ in range(args.in_flight - 1)]
        streams2 = [torch.cuda.Stream(device=dev) for _ in range(args.in_flight)]
        counter = [0]

        def pipelined_step():
            j = counter[0] % args.in_flight
            counter[0] += 1
            plans2[j].factor_device(values.data_ptr(), Ls2[j].data_ptr(), streams2[j].cuda_stream)

        dt_p = timed(pipelined_step, max(args.warmup, args.in_flight), args.steps)
        same = all(bool(torch.equal(Ls2[0], x)) for x in Ls2[1:])
        ok = all(p.status() == 0 for p in plans2)
        pipelined = {"in_flight": args.in_flight, "value": args.steps / dt_p, "unit": "factorizations/s",
                     "ms_per_step": dt_p / args.steps * 1e3, "factors_bitwise_equal": same and ok,
                     "note": "independent factorizations alternating over separate plans, lValues buffers and "
                             "streams; not the headline value (that is one factorization after the other)"}
        del plans2[1:], Ls2[1:]

    # ---- forward / backward solves -------------------------------------------------------------
    dt_s = dt_b = solve_err = back_err = None
    nrhs = args.nrhs
    solve_plan = None
    BX = None
    sharded = None
    if world == 1:
        if not args.no_solve:
            solve_plan = plan
            dt_s, dt_b, solve_err, BX, back_err = measure_solves(solve_plan, sym, L, nrhs, args.warmup, args.steps)
    elif not args.no_solve:
        # the factor is distributed (every piece final on its owner): the solves run sharded -- every rank its own
        # subtrees; the supernodes above the cut are solved WHERE THEY WERE FACTORED, the ranks going through the etree
        # levels above the cut together with one all-reduce of the level's x rows per level (LeveledShardedSolve; only
        # the pieces of a SPLIT supernode move, once per factorization, untimed).  Beside it, for comparison, the form of
        # rounds 3-4: the panels above the cut collected on rank 0, which solves them alone (ShardedSolve).
        import scipy.sparse as sp
        A2 = sp.csc_matrix((sym.A2x, sym.A2i, sym.A2p), shape=(sym.n, sym.n))
        ones = np.ones(sym.n)
        c = A2 @ ones + A2.T @ ones - A2.diagonal()
        C = torch.from_numpy(c).to(dev).repeat(nrhs).contiguous()
        holder = {}
        top_plan = api.Plan(sym, local_rank)
        LS = MG.LeveledShardedSolve(sym, pieces, D, rank, dist, MG.PlanSolver(plan), MG.PlanSolver(top_plan),
                                    top_plan.solve_levels(), stage_on_host=(backend != "nccl"))
        moved_lev = LS.gather_solve_parts(L)
        held = torch.tensor([float(sum(int(pieces["value_end"][p] - pieces["value_begin"][p]) for p in range(len(D.owner))
                                       if D.owner[p] == rank or (D.in_subtree[p] == 0 and LS.sn_owner[pieces["supernode"][p]] == rank)))],
                            dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        held_all = [torch.zeros_like(held) for _ in range(world)]
        dist.all_gather(held_all, held)
        SS = LS

        def fwd():
            holder["y"] = SS.forward(L, C, nrhs, stream)

        def bwd():
            holder["x"] = SS.backward(L, holder["y"], nrhs, stream)

        def check(what):
            if plan.solve_status() != 0 or top_plan.solve_status() != 0:
                raise SystemExit(f"sharded solve ({what}): a hand-off wait timed out")
            if rank == 0:
                err = float((holder["x"] - 1.0).abs().max().item())
                if not (err <= BACK_TOL):
                    raise SystemExit(f"sharded solves ({what}) are off: max|x - 1| = {err:.3e} for L L' x = (P A P') 1")
                return err
            return None

        dt_s = timed(fwd, args.warmup, args.steps)
        dt_b = timed(bwd, args.warmup, args.steps)
        back_err = check("leveled")
        exchanged = LS.exchanged
        # the comparison form (its plan for the part above the cut: rank 0's top_plan with another mask)
        SS = MG.ShardedSolve(sym, pieces, D, rank, dist, MG.PlanSolver(plan), MG.PlanSolver(top_plan) if rank == 0 else None,
                             stage_on_host=(backend != "nccl"))
        moved = SS.gather_root_part(L)
        dt_s_root = timed(fwd, args.warmup, args.steps)
        dt_b_root = timed(bwd, args.warmup, args.steps)
        check("root rank")
        if rank == 0:
            sharded = {"form": "leveled: supernodes above the cut solved by the ranks that factored them",
                       "supernodes_above_the_cut": int(LS.root_mask.sum()), "levels_above_the_cut": len(LS.top_levels),
                       "x_entries_all_reduced_per_solve": int(exchanged),
                       "elements_moved_once_per_factorization": int(moved_lev),
                       "factor_entries_held_per_rank": [int(t.item()) for t in held_all],
                       "root_rank_form": {"forward_ms": dt_s_root / args.steps * 1e3, "backward_ms": dt_b_root / args.steps * 1e3,
                                          "elements_moved_once_per_factorization": int(moved),
                                          "note": "rounds 3-4: panels above the cut collected on rank 0, one reduce / "
                                                  "broadcast of x at the cut"},
                       "note": "forward: subtree solves on their owners, then per etree level above the cut one all-reduce of "
                               "the level's x rows and the owners' level launches (parsy_solve_levels_device); backward: the "
                               "same from the top level down, then the subtree solves and one reduce that collects x"}
        dist.barrier()

    # ---- per-kernel timing (hipEvents on the launch stream, launches serialised) for the roofline ----
    prof = None
    if rank == 0 and world == 1 and args.profile_steps > 0:
        plan.profile(2)
        for _ in range(args.profile_steps):
            plan.factor_device(values.data_ptr(), L.data_ptr(), stream)
            torch.cuda.synchronize()
            plan.profile_collect()
        prof_f = plan.profile_get()
        prof_s = None
        if BX is not None:
            B, X = BX
            plan.profile(2)
            for _ in range(args.profile_steps):
                X.copy_(B)
                plan.solve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, stream)
                torch.cuda.synchronize()
                plan.profile_collect()
            prof_s = plan.profile_get()
            plan.profile(2)
            for _ in range(args.profile_steps):
                X.copy_(B)
                plan.backsolve_device(L.data_ptr(), X.data_ptr(), nrhs, sym.n, stream)
                torch.cuda.synchronize()
                plan.profile_collect()
            prof_b = plan.profile_get()
        plan.profile(0)
        prof = (prof_f, prof_s, prof_b if BX is not None else None)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = dt_f / args.steps * 1e3
    fact_per_s = args.steps / dt_f
    grid = M.WORKLOADS.get(args.workload)
    out = {
        "metric": "Cholesky factorizations/sec (+ SpTRSV solves/sec), SuiteSparse SPD set",
        "value": fact_per_s,
        "unit": "factorizations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",   # N > 1 shards ONE factorization (total work fixed); N = 1 is its base point
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic" if not args.mtx else "file",
        "config": {
            "workload": (f"{args.workload}-class stand-in (BASELINE.json configs[2] when flan: Flan_1565): grid "
                         f"{grid[:3]} {grid[3]}-point stencil, geometric nested dissection" if grid else
                         (f"{args.mtx} (MatrixMarket file), ordering: " + (args.order or "parsy_order_nd (graph nested "
                          "dissection)") if args.mtx else args.workload)),
            "n": sym.n, "nnz_A_lower": int(sym.nnzA), "nsuper": sym.nsuper, "nnz_L": int(sym.nnzL),
            "xsize": int(sym.xsize), "flops_F": sym.flops_colcount, "flops_executed": sym.flops_stored,
            "etree_levels": sym.nlevels, "cholesky_view": {k: info[k] for k in (
                "n_pieces", "chol_levels", "piece_width", "big_min_k", "big_tasks", "big_entries", "dense_entries", "dense_tasks")},
            "launches_per_factorization": info["chol_launches"],
            "parallelism": "1 GPU" if world == 1 else (
                f"{world} GPUs: etree subtrees below a cut on one rank each, the pieces of the separators above it dealt "
                f"over all ranks; after every level of the Cholesky view the finished pieces travel point to point "
                f"(RCCL) to the ranks that read them"),
        },
        "gflops_F": sym.flops_colcount / (dt_f / args.steps) / 1e9,
        "solves_per_sec": (args.steps * nrhs / dt_s) if dt_s else None,
        "solve_nrhs": nrhs,
        "solve_ms": (dt_s / args.steps * 1e3) if dt_s else None,
        "solve_max_abs_err_vs_ones": solve_err,
        "backward_solve_ms": (dt_b / args.steps * 1e3) if dt_b else None,
        "forward_backward_max_abs_err_vs_ones": back_err,
        "throughput_in_flight": pipelined,
        "multi_gpu": multi,
        "sharded_solves": sharded,
        "inspect_seconds": t_inspect, "plan_seconds": t_plan,
    }

    if prof is not None:
        pf, ps, pb = prof
        runs = pf["runs"]
        kinds = {k: v / runs for k, v in pf["ms"].items() if v > 0}
        launches = {k: v // runs for k, v in pf["launches"].items() if v > 0}
        tile_ms = kinds.get("TILES", 0.0) + kinds.get("CHAIN", 0.0)
        tile_flops = info["tile_update_flops"] + info["inner_flops"]
        tiles = {"kernel": "tile_task<> (k_chol_tiles + k_chol_chain: per-wave update streams, POTRF/TRSM of the tiles)",
                 "achieved": tile_flops / (tile_ms * 1e-3) / 1e12 if tile_ms > 0 else 0.0, "unit": "TFLOP/s",
                 "algorithmic_flops_per_factorization": tile_flops, "kernel_ms_per_factorization_serialized": tile_ms}
        tiles["frac"] = tiles["achieved"] / FP64_MFMA_PEAK_TFLOPS
        big_ms, big_n = kinds.get("BIG", 0.0), launches.get("BIG", 0)
        dense_ms, dense_n = kinds.get("DENSE", 0.0), launches.get("DENSE", 0)
        ragged_flops = info["big_flops"] - info["dense_flops"]
        # the two kernels of the BIG updates side by side (algorithmic flops = the reference's DSYRK + DGEMM counts of
        # the updates they apply: K n1 (n1 + 1) + 2 K (m - n1) n1 per (target, descendant) pair, exact from the schedule,
        # split by entry: full 128 x 128 blocks -> k_chol_dense, the ragged rest -> k_chol_big)
        big_kernels = {
            "k_chol_dense": {"ms_per_factorization_serialized": dense_ms, "launches": int(dense_n), "algorithmic_flops": info["dense_flops"],
                             "achieved": info["dense_flops"] / (dense_ms * 1e-3) / 1e12 if dense_ms > 0 else 0.0},
            "k_chol_big": {"ms_per_factorization_serialized": big_ms, "launches": int(big_n), "algorithmic_flops": ragged_flops,
                           "achieved": ragged_flops / (big_ms * 1e-3) / 1e12 if big_ms > 0 else 0.0},
            "both": {"ms_per_factorization_serialized": big_ms + dense_ms, "algorithmic_flops": info["big_flops"],
                     "achieved": info["big_flops"] / ((big_ms + dense_ms) * 1e-3) / 1e12 if big_ms + dense_ms > 0 else 0.0},
        }
        for v in big_kernels.values():
            v["frac"] = v["achieved"] / FP64_MFMA_PEAK_TFLOPS
        if dense_ms > big_ms and dense_ms > tile_ms:
            # dominant kernel: k_chol_dense
            name, dom_ms, dom_n, dom_flops = ("k_chol_dense (the full 128x128 blocks of the reference's DSYRK/DGEMM of wide "
                                              "descendants and between the pieces of split supernodes: 4-slot LDS-DMA ring, "
                                              "FP64 MFMA, software-pipelined operand reads)"), dense_ms, dense_n, info["dense_flops"]
            key = "k_chol_dense"
        elif big_ms > 0 and info["big_flops"] >= tile_flops:
            # dominant kernel: k_chol_big
            name, dom_ms, dom_n, dom_flops = ("k_chol_big (LDS-staged 128x128 FP64-MFMA update tiles: the reference's "
                                              "DSYRK/DGEMM of wide descendants and between the pieces of split "
                                              "supernodes)"), big_ms, big_n, ragged_flops
            key = "k_chol_big"
        else:
            name, dom_ms, dom_n, dom_flops, key = tiles["kernel"], tile_ms, launches.get("TILES", 0) + launches.get(
                "CHAIN", 0), tile_flops, "k_chol_tiles"
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        # HBM-side bytes per launch from the PMC passes of the same kernels (rocprofv3 --pmc FETCH_SIZE and
        # WRITE_SIZE in separate runs, tools/collect_pmc.sh): counters cannot be collected inside this run, so the
        # committed summary is quoted -- only while the kernel sources are the ones it was measured on
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(ROOT / "profiles" / f"{PMC_ROUND}_{args.workload}_pmc_traffic.json"))
            if pmc.get("kernel_source_hash") == kernel_source_hash() and key in pmc.get("kernels", {}):
                traffic = pmc["kernels"][key]["hbm_bytes_per_launch"]
                traffic_src = (f"profiles/{PMC_ROUND}_{args.workload}_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                               f"separate passes, calibrated; same kernel sources)")
            elif key in pmc.get("kernels", {}):
                traffic_src = "profiles summary is from other kernel sources: not quoted"
        except (OSError, ValueError):
            pass
        out["roofline"] = {
            "kernel": name, "bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "bytes per launch",
            "traffic_source": traffic_src,
            "launches_per_factorization": int(dom_n),
            "algorithmic_flops_per_launch": dom_flops / max(dom_n, 1),
            "algorithmic_flops_per_factorization": dom_flops,
            "avg_launch_ms": dom_ms / max(dom_n, 1),
            "kernel_ms_per_factorization_serialized": dom_ms,
            "timing": "hipEvents around every launch on the launch stream, launches serialised on one stream "
                      "(in the timed steps the side stream overlaps them: the sum over kinds exceeds ms_per_step)",
            "whole_job": {"flops_F": sym.flops_colcount, "achieved": sym.flops_colcount / (ms_per_step * 1e-3) / 1e12,
                          "frac": sym.flops_colcount / (ms_per_step * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS},
            "kind_ms_per_factorization_serialized": kinds,
            "big_update_kernels": big_kernels,
            "tile_kernel": tiles,
        }
        if ps is not None:
            sruns = ps["runs"]
            solve_traffic = None
            try:
                pmc = json.load(open(ROOT / "profiles" / f"{PMC_ROUND}_{args.workload}_pmc_traffic.json"))
                if pmc.get("kernel_source_hash") == kernel_source_hash() and nrhs == 1:
                    ks = pmc["kernels"]
                    # the profiled program runs as many forward (and backward) solves as factorizations; the inverse
                    # diagonal blocks are formed once per solve of either kind
                    fwd = [v for k, v in ks.items() if k.startswith(("k_solve_small", "k_solve_tiny", "k_solve_chain", "k_solve_one"))]
                    nsolves = pmc["factorizations_in_the_profiled_run"]
                    solve_traffic = (sum(v["read_bytes_in_run"] + v["write_bytes_in_run"] for v in fwd) / max(nsolves, 1)
                                     + ks.get("k_diag_inverse", {}).get("hbm_bytes_per_launch", 0.0))
            except (OSError, ValueError, KeyError):
                pass
            solve_bytes = 8.0 * sym.xsize + 4.0 * sym.ssize + 16.0 * sym.n * nrhs
            s_ms = sum(ps["ms"].values()) / sruns
            out["roofline_solve"] = {
                "bound": "hbm", "achieved": solve_bytes / (s_ms * 1e-3) / 1e9 if s_ms > 0 else 0.0,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (solve_bytes / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if s_ms > 0 else 0.0,
                "algorithmic_bytes_per_solve": solve_bytes, "traffic": solve_traffic,
                "traffic_unit": "HBM-side bytes per forward solve (k_diag_inverse + k_solve_*, same PMC summary)",
                "kind_ms_per_solve": {k: v / sruns for k, v in ps["ms"].items() if v > 0},
            }

        if pb is not None:
            bruns = pb["runs"]
            back_traffic = None
            try:
                pmc = json.load(open(ROOT / "profiles" / f"{PMC_ROUND}_{args.workload}_pmc_traffic.json"))
                if pmc.get("kernel_source_hash") == kernel_source_hash() and nrhs == 1:
                    ks = pmc["kernels"]
                    bwd = [v for k, v in ks.items() if k.startswith("k_bsolve")]
                    nsolves = pmc["factorizations_in_the_profiled_run"]
                    back_traffic = (sum(v["read_bytes_in_run"] + v["write_bytes_in_run"] for v in bwd) / max(nsolves, 1)
                                    + ks.get("k_diag_inverse", {}).get("hbm_bytes_per_launch", 0.0))
            except (OSError, ValueError, KeyError):
                pass
            # the backward solve L' x = y reads the same bytes as the forward one: every stored value of L once, the
            # row ids once, x read and written once per right-hand side
            back_bytes = 8.0 * sym.xsize + 4.0 * sym.ssize + 16.0 * sym.n * nrhs
            b_ms = sum(pb["ms"].values()) / bruns
            out["roofline_backsolve"] = {
                "bound": "hbm", "achieved": back_bytes / (b_ms * 1e-3) / 1e9 if b_ms > 0 else 0.0,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (back_bytes / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if b_ms > 0 else 0.0,
                "algorithmic_bytes_per_solve": back_bytes, "traffic": back_traffic,
                "traffic_unit": "HBM-side bytes per backward solve (k_diag_inverse + k_bsolve_*, same PMC summary)",
                "ms_per_solve_serialized": b_ms,
                "kind_ms_per_solve": {k: v / bruns for k, v in pb["ms"].items() if v > 0},
            }

    # (the CPU baseline's second sample works on this factor: the extras below reuse the buffer)
    L_host = None
    if world == 1 and not args.no_cpu_baseline and sym.nsuper > 1:
        try:
            L_host = L.cpu().numpy()
        except Exception as e:
            log(f"no host copy of the factor for the top-separator sample: {e!r}")
    # ---- extra objects: the other single-GPU configurations of BASELINE.json (never `value`) ----
    del plan, solve_plan, BX
    if world == 1 and not args.no_extras and args.workload == "flan":
        L_big = L
        try:
            extras = {}
            # configs[1] nd24k-class: factorizations/s + solves/s
            A1, p1 = M.workload("nd24k")
            s1 = I.analyze(A1, p1)
            pl1 = api.Plan(s1, local_rank)
            v1 = torch.from_numpy(np.ascontiguousarray(s1.A2x)).to(dev)
            L1 = L_big[: int(s1.xsize)]
            d1 = timed(lambda: pl1.factor_device(v1.data_ptr(), L1.data_ptr(), stream), 3, 20, collective=False)
            if pl1.status() != 0:
                raise RuntimeError(f"nd24k factorization status {pl1.status()}")
            ds1, db1, e1, _, eb1 = measure_solves(pl1, s1, L1, 1, 3, 20)
            extras["nd24k"] = {"workload": "nd24k-class stand-in (configs[1]): grid (42, 42, 42) 27-point stencil",
                               "n": s1.n, "flops_F": s1.flops_colcount, "factorizations_per_sec": 20 / d1,
                               "ms_per_factorization": d1 / 20 * 1e3, "gflops_F": s1.flops_colcount / (d1 / 20) / 1e9,
                               "solves_per_sec": 20 / ds1, "solve_ms": ds1 / 20 * 1e3,
                               "backward_solve_ms": db1 / 20 * 1e3, "solve_max_abs_err_vs_ones": e1,
                               "forward_backward_max_abs_err_vs_ones": eb1}
            del pl1
            # configs[0] ex15-class (the reference's own CPU-runnable case; its 1-thread CPU figures: cpu_baseline.ex15_1_thread)
            A0, p0 = M.workload("ex15")
            s0 = I.analyze(A0, p0)
            pl0 = api.Plan(s0, local_rank)
            v0 = torch.from_numpy(np.ascontiguousarray(s0.A2x)).to(dev)
            L0 = L_big[: int(s0.xsize)]
            d0 = timed(lambda: pl0.factor_device(v0.data_ptr(), L0.data_ptr(), stream), 5, 50, collective=False)
            if pl0.status() != 0:
                raise RuntimeError(f"ex15 factorization status {pl0.status()}")
            ds0, db0, e0, _, eb0 = measure_solves(pl0, s0, L0, 1, 5, 50)
            extras["ex15"] = {"workload": "ex15-class stand-in (configs[0]): 83 x 83 5-point grid", "n": s0.n,
                              "flops_F": s0.flops_colcount, "factorizations_per_sec": 50 / d0,
                              "ms_per_factorization": d0 / 50 * 1e3, "solves_per_sec": 50 / ds0,
                              "solve_ms": ds0 / 50 * 1e3, "backward_solve_ms": db0 / 50 * 1e3,
                              "solve_max_abs_err_vs_ones": e0, "forward_backward_max_abs_err_vs_ones": eb0,
                              "solve_launches_per_direction": 1 if pl0.info["solve_one"] == 3 else pl0.info["solve_launches"],
                              "note": "a job of a few launch latencies: 20 launches per factorization; the solves are ONE launch "
                                      "each (k_solve_one / k_bsolve_one: values handed over as the data itself)"}
            del pl0
            # configs[3] parabolic_fem-class: BCSC lower-triangular solve only, many right-hand sides
            A3, p3 = M.workload("parabolic_fem")
            s3 = I.analyze(A3, p3)
            pl3 = api.Plan(s3, local_rank)
            v3 = torch.from_numpy(np.ascontiguousarray(s3.A2x)).to(dev)
            L3 = L_big[: int(s3.xsize)]
            pl3.factor_device(v3.data_ptr(), L3.data_ptr(), stream)
            torch.cuda.synchronize()
            if pl3.status() != 0:
                raise RuntimeError(f"parabolic_fem factorization status {pl3.status()}")
            pf3 = {"workload": "parabolic_fem-class stand-in (configs[3]): grid (725, 725) 5-point stencil, solve only",
                   "n": s3.n, "xsize": int(s3.xsize), "nrhs": {}}
            for q in (1, 8, 64):
                dsq, dbq, eq, _, ebq = measure_solves(pl3, s3, L3, q, 2, 10)
                bytes_q = 8.0 * s3.xsize + 4.0 * s3.ssize + 16.0 * s3.n * q
                pf3["nrhs"][str(q)] = {"solves_per_sec": 10 * q / dsq, "ms_per_block_solve": dsq / 10 * 1e3,
                                       "algorithmic_GBps": bytes_q / (dsq / 10) / 1e9,
                                       "frac_of_hbm_peak": bytes_q / (dsq / 10) / 1e9 / HBM_PEAK_GBS,
                                       "backward_ms_per_block_solve": dbq / 10 * 1e3, "max_abs_err_vs_ones": eq,
                                       "forward_backward_max_abs_err_vs_ones": ebq,
                                       "device_ms_forward": measure_solves.device_ms[0],
                                       "device_ms_backward": measure_solves.device_ms[1],
                                       "timing": "ms_per_block_solve: 10 steps back to back, each step = a copy that restores the "
                                                 "right-hand side (the solve is in place) + the solve; device_ms_*: hipEvents around "
                                                 "one solve's launches, best of three"}
            extras["parabolic_fem"] = pf3
            del pl3
            out["other_configs"] = extras
        except Exception as e:  # extras never cost the headline
            out["other_configs"] = {"failed": repr(e)}

    if not args.no_cpu_baseline and world == 1:
        try:
            share, share_src = granted_cpus()
            out["cpu_baseline"] = cpu_baseline(sym, share)
            out["cpu_baseline"]["cores_source"] = share_src
            if L_host is not None:
                try:
                    out["cpu_baseline"]["top_separator"] = cpu_baseline_top(sym, L_host, share, out["cpu_baseline"])
                except Exception as e:
                    out["cpu_baseline"]["top_separator"] = {"failed": repr(e)}
                L_host = None
            out["cpu_baseline"]["ex15_1_thread"] = cpu_baseline_ex15(1)
            out["cpu_baseline"]["nd24k_whole_matrix"] = cpu_baseline_whole("nd24k", share)
        except Exception as e:  # the baseline is reporting only; never lose the GPU numbers to it
            out["cpu_baseline"] = {"value": None, "unit": "factorizations/s", "cores": 0, "kind": "port",
                                   "sample": f"failed: {e!r}"}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
