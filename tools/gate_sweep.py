"""Gate sweep: every size / shape gate of the schedule against its alternatives on a set of
generated inputs that is wider than the five stand-ins the gates were set on -- 2-D and 3-D grids, 5 / 7 / 9 / 27-point
stencils, two sizes each, one with the graph nested dissection (parsy_order_nd) instead of the geometric one, one banded.
Per input and setting: factorization, forward / backward solve with 1, 8 and 64 right-hand sides (best of 3, device
time).  Prints one table and, per gate, the inputs where the default loses more than 10 % to an alternative.
Usage: python tools/gate_sweep.py [quick]"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from parsy_bench_amd import api, inspector as I, matrices as M

QUICK = len(sys.argv) > 1 and sys.argv[1] == "quick"
INPUTS = [
    # name, (nx, ny, nz, stencil), ordering
    ("2d5 200x200", (200, 200, 1, 5), "geo"), ("2d5 500x500", (500, 500, 1, 5), "geo"),
    ("2d9 200x200", (200, 200, 1, 9), "geo"), ("2d9 400x400", (400, 400, 1, 9), "geo"),
    ("3d7 30^3", (30, 30, 30, 7), "geo"), ("3d7 56^3", (56, 56, 56, 7), "geo"),
    ("3d27 24^3", (24, 24, 24, 27), "geo"), ("3d27 48^3", (48, 48, 48, 27), "geo"),
    ("3d27 64^3", (64, 64, 64, 27), "geo"), ("3d27 96x96x12", (96, 96, 12, 27), "geo"),
    ("2d5 300x300 graph-ND", (300, 300, 1, 5), "nd"), ("3d7 32^3 graph-ND", (32, 32, 32, 7), "nd"),
    ("banded 20000x6", (20000, 6, 1, 5), "natural"),
]
if QUICK:
    INPUTS = INPUTS[:2] + INPUTS[4:5] + INPUTS[-1:]
# gate -> alternatives to the default (environment of the plan build and of the solves)
GATES = {
    "ONE-launch solves (kOne*)": [{"PARSY_SOLVE_ONE": "0"}, {"PARSY_SOLVE_ONE": "2"}],
    "subtree launches (kSubtree*)": [{"PARSY_SUBTREES": "0"}, {"PARSY_SUBTREES": "16"}],
    "many-RHS subtree / band kernels (kSubTier*)": [{"PARSY_SUB_MRHS_MIN": "0"}, {"PARSY_SUB_TIER_MIN_TREES": "0"},
                                                    {"PARSY_SUB_TIER_MIN_TREES": "512"}, {"PARSY_SUB_TIER_MIN_TREES": "128"}],
    "row-major X from 16 right-hand sides (200 entries per row)": [{"PARSY_XT_MIN": "0"}, {"PARSY_XT_MIN": "16"}],
    "many-RHS kernels from 6 / 2 (PARSY_MRHS_MIN)": [{"PARSY_MRHS_MIN": "16"}],
    "backward rows below (kBelow*)": [{"PARSY_BSOLVE_BELOW": "0"}, {"PARSY_BSOLVE_BELOW": "2"}],
    "dense update kernel (kDense*)": [{"PARSY_BIG_DENSE": "0"}, {"PARSY_BIG_DENSE": "2"}],
    "two chain launches per level (kChainSplitAutoFlops)": [{"PARSY_CHAIN_SPLIT": "0"}, {"PARSY_CHAIN_SPLIT": "2"}],
}
NRHS = (1, 4, 8, 64)
dev = torch.device("cuda", 0)


def measure(sym, values, Lbuf, env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        plan = api.Plan(sym, 0)
        L = Lbuf[: int(sym.xsize)]
        t = []
        for _ in range(4):
            plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
            torch.cuda.synchronize()
            t.append(plan.last_factor_ms())
        if plan.status() != 0:
            raise RuntimeError("factorization status %d" % plan.status())
        res = {"factor": min(t[1:])}
        b = torch.empty(sym.n, dtype=torch.float64, device=dev)
        plan.rhs_ones_device(L.data_ptr(), b.data_ptr(), 0)
        for nrhs in NRHS:
            B = b.repeat(nrhs).contiguous()
            X = torch.empty_like(B)
            for tag, fn in (("f", plan.solve_device), ("b", plan.backsolve_device)):
                tt = []
                for _ in range(4):
                    X.copy_(B)
                    fn(L.data_ptr(), X.data_ptr(), nrhs, sym.n, 0)
                    torch.cuda.synchronize()
                    tt.append(plan.last_solve_ms())
                if tag == "f" and float((X - 1).abs().max()) > 1e-8:
                    raise RuntimeError(f"forward solve off with {env}")
                if plan.solve_status() != 0:
                    raise RuntimeError(f"solve status with {env}")
                res[f"{tag}{nrhs}"] = min(tt[1:])
        plan.close()
        return res
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    keys = ["factor"] + [f"{t}{n}" for n in NRHS for t in "fb"]
    losses = {g: [] for g in GATES}
    print("times in ms: " + " ".join(keys))
    for name, (nx, ny, nz, st), order in INPUTS:
        A = M.grid_spd(nx, ny, nz, st, 0.1)
        perm = M.grid_nd(nx, ny, nz) if order == "geo" else I.order_nd(A) if order == "nd" else np.arange(A.n, dtype=np.int32)
        t0 = time.time()
        sym = I.analyze(A, perm)
        values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
        Lbuf = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
        base = measure(sym, values, Lbuf, {})
        print(f"\n== {name}: n {sym.n}, supernodes {sym.nsuper}, entries of L {sym.xsize} ({sym.xsize / sym.n:.0f} per row), "
              f"levels {sym.nlevels} (analysis {time.time() - t0:.1f} s)")
        print(f"   {'default':58s} " + " ".join(f"{base[k]:8.3f}" for k in keys), flush=True)
        for gate, alts in GATES.items():
            for env in alts:
                try:
                    r = measure(sym, values, Lbuf, env)
                except Exception as e:
                    print(f"   {str(env):58s} failed: {e!r}", flush=True)
                    continue
                marks = []
                for k in keys:
                    if base[k] > 1.10 * r[k] and base[k] - r[k] > 0.005:
                        marks.append(f"{k}: {base[k]:.3f} vs {r[k]:.3f}")
                print(f"   {str(env):58s} " + " ".join(f"{r[k]:8.3f}" for k in keys) + ("   <-- default loses: " + ", ".join(marks) if marks else ""),
                      flush=True)
                if marks:
                    losses[gate].append(f"{name} {env}: " + ", ".join(marks))
        del values, Lbuf
        torch.cuda.empty_cache()
    print("\n== defaults that lose more than 10 % to an alternative")
    for gate, ls in losses.items():
        print(f"{gate}: {len(ls)}")
        for l in ls:
            print("   " + l)


main()
