"""Generates tests/golden/*.npz from the reference itself (run in the build container only).

Every vector here is an OUTPUT of reference code compiled from /root/reference by
oracle/Makefile into oracle/_ref/libparsy_ref.so (see oracle/ref_harness.cpp for exactly
which reference functions that is), on inputs produced by this repo's deterministic
generators.  The fixtures are data (inputs + expected outputs); no reference source text.

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import oracle as O  # noqa: E402
from parsy_bench_amd import matrices as M  # noqa: E402

OUT = Path(__file__).resolve().parent
R = O.ref()
P = O.P


def ref_analyze(A, perm, nrelax=(4, 16, 48), zrelax=(0.8, 0.1, 0.05), cost=4, level=0, final=2):
    Ap = np.ascontiguousarray(A.Ap, np.int32)
    Ai = np.ascontiguousarray(A.Ai, np.int32)
    Ax = np.ascontiguousarray(A.Ax)
    pm = np.ascontiguousarray(perm, np.int32)
    nr = np.array(nrelax, np.int32)
    zr = np.array(zrelax, np.float64)
    h = R.ref_analyze(A.n, P(Ap), P(Ai), P(Ax), P(pm), P(nr), P(zr), cost, level, final)
    assert h
    out = {}
    for name in ("Perm", "Parent", "ColCount", "super", "col2Sup", "sParent", "s", "A1p", "A1i", "A2p",
                 "A2i", "levelPtr", "parPtr", "partition", "prunePtr", "pruneSet", "wfLevelPtr",
                 "wfLevelSet"):
        n = R.ref_get_int(h, name.encode(), None)
        a = np.zeros(n, np.int32)
        R.ref_get_int(h, name.encode(), P(a))
        out[name] = a
    for name in ("p", "i_ptr"):
        n = R.ref_get_size(h, name.encode(), None)
        a = np.zeros(n, np.uint64)
        R.ref_get_size(h, name.encode(), P(a))
        out[name] = a
    n = R.ref_get_double(h, b"A2x", None)
    a = np.zeros(n)
    R.ref_get_double(h, b"A2x", P(a))
    out["A2x"] = a
    sc = np.zeros(9, np.int64)
    R.ref_get_scalars(h, P(sc))
    out["scalars"] = sc  # n nsuper ssize xsize maxSupWid maxCol nLevels nPar wfLevels
    out["s"] = out["s"][: int(sc[2])]
    return out


def inspector_cases():
    cases = {
        "tiny2d": M.workload("tiny2d"),
        "small3d": M.workload("small3d"),
        "grid2d_9pt": (M.grid_spd(17, 13, 1, 9, 0.5), M.grid_nd(17, 13, 1)),
        "lap7_natural": (M.grid_spd(6, 5, 4, 7, 0.01), np.arange(120, dtype=np.int32)),
        "random60": (M.random_spd(60, 0.08, seed=3), np.random.default_rng(5).permutation(60).astype(np.int32)),
    }
    for name, (A, perm) in cases.items():
        g = ref_analyze(A, perm)
        np.savez_compressed(OUT / f"inspector_{name}.npz", Ap=A.Ap, Ai=A.Ai, Ax=A.Ax, perm=perm, **g)
        print("inspector", name, g["scalars"])
    # a second supernode-relaxation setting (examples/choleskyTest03.cpp:107)
    A, perm = M.workload("small3d")
    g = ref_analyze(A, perm, nrelax=(4, 16, 0))
    np.savez_compressed(OUT / "inspector_small3d_relax0.npz", Ap=A.Ap, Ai=A.Ai, Ax=A.Ax, perm=perm, **g)


def dense_kernel_cases():
    rng = np.random.default_rng(11)
    out = {}
    # triangularSolve/BLAS.h: dlsolve_blas_nonUnit / dmatvec_blas on every unroll tail
    for ncol in (1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 23):
        ldm = ncol + 5
        Mx = np.asfortranarray(rng.uniform(-1, 1, (ldm, ncol)))
        Mx[np.arange(ncol), np.arange(ncol)] += 4.0
        rhs = rng.uniform(-1, 1, ncol)
        flat = np.ascontiguousarray(Mx.T.reshape(-1))  # column-major storage
        x = rhs.copy()
        R.ref_dlsolve_blas_nonUnit(ldm, ncol, P(flat), P(x))
        out[f"dlsolve_{ncol}_M"] = flat
        out[f"dlsolve_{ncol}_rhs"] = rhs
        out[f"dlsolve_{ncol}_x"] = x
        nrow = 5
        vec = rng.uniform(-1, 1, ncol)
        y = rng.uniform(-1, 1, nrow)
        y0 = y.copy()
        Mv = np.ascontiguousarray(flat[ncol:])  # rows below the triangle, same ldm
        R.ref_dmatvec_blas(ldm, nrow, ncol, P(Mv), P(vec), P(y))
        out[f"dmatvec_{ncol}_vec"] = vec
        out[f"dmatvec_{ncol}_y0"] = y0
        out[f"dmatvec_{ncol}_y"] = y
    # cholesky/MyBLAS.h: Cholesky_col + lSolve_dense_col (the reference's readable POTRF/TRSM)
    for dim, n in ((1, 1), (3, 5), (8, 8), (13, 20), (32, 40)):
        B = rng.uniform(-1, 1, (n, n))
        Sx = B @ B.T + n * np.eye(n)
        a = np.ascontiguousarray(Sx.T.reshape(-1))
        a_in = a.copy()
        R.ref_Cholesky_col(n, dim, P(a))
        out[f"cholcol_{dim}_{n}_in"] = a_in
        out[f"cholcol_{dim}_{n}_out"] = a
    np.savez_compressed(OUT / "dense_kernels.npz", **out)
    print("dense kernels:", len(out), "arrays")


def lbc_ex15_cases():
    """The reference's own LBC partitions (getCoarseLevelSet_6, cholesky/InspectionLevel_06.h:18) of the ex15-class
    stand-in for the sweep of scripts/eval.sh:5-21 (levelParam in {2,1,0,-1,-2}, finalSeqNode in {2,4}) at
    costParam = 1, 4, 8, 16 threads: what bench.py's cpu_baseline_ex15 schedules the CPU port by."""
    A, perm = M.workload("ex15")
    out = {}
    for cost in (1, 4, 8, 16):
        for lev in (2, 1, 0, -1, -2):
            for fin in (2, 4):
                nl, levelPtr, parPtr, partition = O.ref_hlevel(A, perm, cost, lev, fin)
                key = f"c{cost}_l{lev}_f{fin}"
                out[key + "_nLevels"] = np.array([nl], np.int32)
                out[key + "_levelPtr"] = levelPtr
                out[key + "_parPtr"] = parPtr
                out[key + "_partition"] = partition
    np.savez_compressed(OUT / "lbc_ex15.npz", **out)
    print("lbc_ex15:", len(out) // 4, "settings")


if __name__ == "__main__":
    inspector_cases()
    dense_kernel_cases()
    lbc_ex15_cases()
