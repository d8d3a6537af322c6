"""ctypes face of the CPU oracle (oracle/parsy_oracle.c) and, where it was built,
of the reference-backed checker oracle/_ref/libparsy_ref.so.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under parsy_bench_amd/ imports this.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "libparsy_oracle.so"
REF_SO = HERE / "_ref" / "libparsy_ref.so"
REFERENCE_ROOT = Path("/root/reference")

_oracle = None
_ref = None
vp = C.c_void_p


def build(ref: bool = True) -> None:
    """Compile the oracle (and oracle/_ref when the reference tree is present)."""
    targets = [str(ORACLE_SO)]
    subprocess.run(["make", "-s", "-C", str(HERE), *targets], check=True)
    if ref and REFERENCE_ROOT.is_dir():
        subprocess.run(["make", "-s", "-C", str(HERE), "ref"], check=True)


def P(a):
    return None if a is None else a.ctypes.data_as(vp)


def lib():
    global _oracle
    if _oracle is None:
        if not ORACLE_SO.exists():
            build(ref=False)
        L = C.CDLL(str(ORACLE_SO))
        for name in ("oracle_cholesky_left_par_05", "oracle_cholesky_left_par_waveFront",
                     "oracle_blockedLsolve", "oracle_leveledBlockedLsolve",
                     "oracle_H2LeveledBlockedLsolve", "oracle_H2LeveledBlockedLsolve_Peeled",
                     "oracle_ereach_sn", "oracle_getLevelSet", "oracle_testTriangular",
                     "oracle_dpotrf_l", "oracle_max_threads", "oracle_bcsc2csc"):
            getattr(L, name).restype = C.c_int
        L.oracle_cholesky_left_par_05.argtypes = (
            [C.c_int] + [vp] * 8 + [C.c_int] + [vp] * 5 + [C.c_int, vp, vp, C.c_int, vp, vp]
            + [C.c_int] * 4 + [vp])
        L.oracle_cholesky_left_par_waveFront.argtypes = (
            [C.c_int] + [vp] * 8 + [C.c_int] + [vp] * 5 + [C.c_int, vp, vp] + [C.c_int] * 4)
        base = [C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp]
        L.oracle_blockedLsolve.argtypes = base
        L.oracle_leveledBlockedLsolve.argtypes = base + [C.c_int, vp, vp, C.c_int]
        L.oracle_H2LeveledBlockedLsolve.argtypes = base + [C.c_int, vp, vp, C.c_int, vp, vp, C.c_int]
        L.oracle_H2LeveledBlockedLsolve_Peeled.argtypes = base + [C.c_int, vp, vp, C.c_int, vp, vp,
                                                                 C.c_int, C.c_int]
        L.oracle_blockedLTsolve.restype = C.c_int
        L.oracle_blockedLTsolve.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_int, vp]
        L.oracle_ereach_sn.argtypes = [C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
        L.oracle_getLevelSet.argtypes = [C.c_size_t, vp, vp, vp]
        L.oracle_dlsolve_blas_nonUnit.argtypes = [C.c_int, C.c_int, vp, vp]
        L.oracle_dmatvec_blas.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp]
        L.oracle_rhsInitBlocked.argtypes = [C.c_size_t, C.c_size_t, vp, vp, vp, vp, vp]
        L.oracle_testTriangular.argtypes = [C.c_size_t, vp]
        L.oracle_bcsc2csc.argtypes = [C.c_size_t, C.c_size_t, vp, vp, vp, vp, vp, vp, vp, vp]
        L.oracle_dsyrk_ln.argtypes = [C.c_int, C.c_int, vp, C.c_int, vp, C.c_int]
        L.oracle_dgemm_nt.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int]
        L.oracle_dpotrf_l.argtypes = [C.c_int, vp, C.c_int]
        L.oracle_dtrsm_rltn.argtypes = [C.c_int, C.c_int, vp, C.c_int, vp, C.c_int]
        L.oracle_bind_blas.argtypes = [vp, vp, vp, vp]
        L.oracle_set_threads.argtypes = [C.c_int]
        _oracle = L
    return _oracle


def have_ref() -> bool:
    """True where the reference-backed checker exists -- or can be built now: it is made on demand (first use by a
    CPU test or by tests/golden/make_golden.py) where /root/reference is present, never by __graft_entry__.build()."""
    if not REF_SO.exists() and REFERENCE_ROOT.is_dir():
        subprocess.run(["make", "-s", "-C", str(HERE), "ref"], check=False)
    return REF_SO.exists()


def ref():
    """The reference-backed checker (only where /root/reference was available to build it)."""
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref is not built and /root/reference is not here to build it from")
        R = C.CDLL(str(REF_SO))
        R.ref_ereach_sn.restype = C.c_int
        R.ref_ereach_sn.argtypes = [C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
        R.ref_getLevelSet.restype = C.c_int
        R.ref_getLevelSet.argtypes = [C.c_size_t, vp, vp, vp]
        R.ref_dlsolve_blas_nonUnit.argtypes = [C.c_int, C.c_int, vp, vp]
        R.ref_dmatvec_blas.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp]
        R.ref_Cholesky_col.argtypes = [C.c_int, C.c_int, vp]
        R.ref_lSolve_dense_col.argtypes = [C.c_int, C.c_int, vp, vp]
        R.ref_analyze.restype = vp
        R.ref_analyze.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]
        for nm in ("ref_get_int", "ref_get_size", "ref_get_double"):
            getattr(R, nm).restype = C.c_longlong
            getattr(R, nm).argtypes = [vp, C.c_char_p, vp]
        R.ref_get_scalars.argtypes = [vp, vp]
        _ref = R
    return _ref


# ---------------------------------------------------------------------------
# convenience wrappers over reference-shaped arrays (a parsy_bench_amd Symbolic or
# anything with the same attribute names)
# ---------------------------------------------------------------------------
def _sz(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def cholesky_05(sym, values, hlevel, threads: int = 1):
    """Run the oracle's cholesky_left_par_05; returns (ok, lValues, timing)."""
    nl, levelPtr, parPtr, partition = hlevel
    lValues = np.zeros(int(sym.xsize), dtype=np.float64)
    timing = np.zeros(8 + max(threads, 1), dtype=np.float64)
    vals = np.ascontiguousarray(values, dtype=np.float64)
    arrs = dict(c=_i32(sym.A2p), r=_i32(sym.A2i), lC=_sz(sym.p), lR=_i32(sym.s), Li=_sz(sym.i_ptr),
                bs=_i32(sym.super), aT=_i32(sym.sParent), cT=_i32(sym.A1p), rT=_i32(sym.A1i),
                c2s=_i32(sym.col2Sup), lp=_i32(levelPtr), pp=_i32(parPtr), pt=_i32(partition))
    lib().oracle_set_threads(threads)
    ok = lib().oracle_cholesky_left_par_05(
        sym.n, P(arrs["c"]), P(arrs["r"]), P(vals), P(arrs["lC"]), P(arrs["lR"]), P(arrs["Li"]),
        P(lValues), P(arrs["bs"]), sym.nsuper, P(timing), P(arrs["aT"]), P(arrs["cT"]),
        P(arrs["rT"]), P(arrs["c2s"]), nl, P(arrs["lp"]), None, 0, P(arrs["pp"]), P(arrs["pt"]),
        1, threads, sym.maxSupWid + 1, sym.maxCol + 1, None)
    return bool(ok), lValues, timing


def cholesky_wavefront(sym, values, threads: int = 1):
    lValues = np.zeros(int(sym.xsize), dtype=np.float64)
    timing = np.zeros(8 + max(threads, 1), dtype=np.float64)
    vals = np.ascontiguousarray(values, dtype=np.float64)
    a = [_i32(sym.A2p), _i32(sym.A2i), _sz(sym.p), _i32(sym.s), _sz(sym.i_ptr), _i32(sym.super),
         _i32(sym.sParent), _i32(sym.A1p), _i32(sym.A1i), _i32(sym.col2Sup), _i32(sym.levelPtr),
         _i32(sym.levelSet)]
    lib().oracle_set_threads(threads)
    ok = lib().oracle_cholesky_left_par_waveFront(
        sym.n, P(a[0]), P(a[1]), P(vals), P(a[2]), P(a[3]), P(a[4]), P(lValues), P(a[5]),
        sym.nsuper, P(timing), P(a[6]), P(a[7]), P(a[8]), P(a[9]), sym.nlevels, P(a[10]), P(a[11]),
        1, threads, sym.maxSupWid + 1, sym.maxCol + 1)
    return bool(ok), lValues, timing


def cholesky_wavefront_subtree(sym, values, root: int, first: int, threads: int = 1):
    """The oracle's wavefront executor on ONE etree subtree: supernodes first..root (contiguous, the
    supernodes are postordered), with level sets restricted to them.  A subtree is a complete
    factorization of its own columns (a target only reads its descendants, common/Reach.h:122-135), so
    this is a bounded sample of the whole job with the same mix of small and wide supernodes (bench.py's
    CPU baseline on inputs whose full CPU factorization takes minutes).  Returns (ok, seconds)."""
    import time
    sub = np.arange(first, root + 1)
    lev = np.zeros(sym.nsuper, dtype=np.int64)
    for l in range(sym.nlevels):
        lev[sym.levelSet[sym.levelPtr[l]:sym.levelPtr[l + 1]]] = l
    order = sub[np.argsort(lev[sub], kind="stable")]
    nl = int(lev[sub].max()) + 1
    levelPtr = np.zeros(nl + 1, dtype=np.int32)
    np.cumsum(np.bincount(lev[sub], minlength=nl), out=levelPtr[1:])
    w = np.diff(sym.super)[sub]
    r = np.diff(sym.i_ptr[sym.super].astype(np.int64))[sub]
    lValues = np.zeros(int(sym.xsize), dtype=np.float64)  # untouched pages are never committed
    timing = np.zeros(8 + max(threads, 1), dtype=np.float64)
    vals = np.ascontiguousarray(values, dtype=np.float64)
    a = [_i32(sym.A2p), _i32(sym.A2i), _sz(sym.p), _i32(sym.s), _sz(sym.i_ptr), _i32(sym.super),
         _i32(sym.sParent), _i32(sym.A1p), _i32(sym.A1i), _i32(sym.col2Sup), levelPtr, _i32(order)]
    lib().oracle_set_threads(threads)
    t0 = time.perf_counter()
    ok = lib().oracle_cholesky_left_par_waveFront(
        sym.n, P(a[0]), P(a[1]), P(vals), P(a[2]), P(a[3]), P(a[4]), P(lValues), P(a[5]),
        sym.nsuper, P(timing), P(a[6]), P(a[7]), P(a[8]), P(a[9]), nl, P(a[10]), P(a[11]),
        1, threads, int(w.max()) + 1, int(r.max()) + 1)
    return bool(ok), time.perf_counter() - t0


def rhs_init_blocked(sym, lValues):
    b = np.zeros(sym.n, dtype=np.float64)
    a = [_sz(sym.p), _i32(sym.s), _sz(sym.i_ptr), np.ascontiguousarray(lValues)]
    lib().oracle_rhsInitBlocked(sym.n, sym.nsuper, P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(b))
    return b


def blocked_lsolve(sym, lValues, x, variant: str = "serial", hlevel=None, threads: int = 1):
    x = np.ascontiguousarray(x, dtype=np.float64).copy()
    a = [_sz(sym.p), _i32(sym.s), np.ascontiguousarray(lValues), _sz(sym.i_ptr), _i32(sym.col2Sup),
         _i32(sym.super)]
    base = (sym.n, P(a[0]), P(a[1]), P(a[2]), int(sym.xsize), P(a[3]), P(a[4]), P(a[5]), sym.nsuper, P(x))
    L = lib()
    L.oracle_set_threads(threads)
    if variant == "serial":
        rc = L.oracle_blockedLsolve(*base)
    elif variant == "H1":
        lp, ls = _i32(sym.levelPtr), _i32(sym.levelSet)
        rc = L.oracle_leveledBlockedLsolve(*base, sym.nlevels, P(lp), P(ls), 1)
    else:
        nl, levelPtr, parPtr, partition = hlevel
        lp, pp, pt = _i32(levelPtr), _i32(parPtr), _i32(partition)
        if variant == "H2":
            rc = L.oracle_H2LeveledBlockedLsolve(*base, nl, P(lp), None, 0, P(pp), P(pt), 1)
        elif variant == "H2peeled":
            rc = L.oracle_H2LeveledBlockedLsolve_Peeled(*base, nl, P(lp), None, 0, P(pp), P(pt), 1, threads)
        else:
            raise ValueError(variant)
    assert rc == 1
    return x


def blocked_ltsolve(sym, lValues, y):
    """Backward solve L' x = y (checker of the product's backward solve; not a reference function)."""
    x = np.ascontiguousarray(y, dtype=np.float64).copy()
    a = [_sz(sym.p), _i32(sym.s), np.ascontiguousarray(lValues), _sz(sym.i_ptr), _i32(sym.super)]
    rc = lib().oracle_blockedLTsolve(sym.n, P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(a[4]), sym.nsuper, P(x))
    assert rc == 1
    return x


def bind_system_blas() -> str:
    """Point the oracle's four dense operations at a Fortran-convention BLAS/LAPACK
    present on this machine (MKL first -- what the reference links -- then scipy's
    OpenBLAS).  Returns a label of what was bound ('' = built-in loops kept)."""
    import os
    # MKL's default (Intel OpenMP) threading layer next to libgomp computes garbage
    # (SURVEY.md 8c trap 2): make MKL use the GNU runtime this oracle is built with.
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
    cands = [("/opt/conda/lib/libmkl_rt.so", ("dsyrk", "dgemm", "dpotrf", "dtrsm"), "mkl_rt")]
    try:
        import scipy
        for pth in (Path(scipy.__file__).parent.parent / "scipy.libs").glob("libscipy_openblas*.so"):
            cands.append((str(pth), ("scipy_dsyrk_", "scipy_dgemm_", "scipy_dpotrf_", "scipy_dtrsm_"),
                          "scipy_openblas"))
    except Exception:
        pass
    for path, names, label in cands:
        if not Path(path).exists():
            continue
        try:
            B = C.CDLL(path, mode=C.RTLD_GLOBAL)
            addrs = [C.cast(getattr(B, nm), vp) for nm in names]
        except (OSError, AttributeError):
            continue
        lib().oracle_bind_blas(*addrs)
        bind_system_blas._keep = B
        return label
    return ""


def unbind_blas() -> None:
    lib().oracle_bind_blas(None, None, None, None)


def ref_hlevel(A, perm, cost: int = 4, level: int = 0, final: int = 2, nrelax=(4, 16, 48),
               zrelax=(0.8, 0.1, 0.05)):
    """H-level schedule (nLevels, levelPtr, parPtr, partition) of the REFERENCE's own LBC partitioner
    (getCoarseLevelSet_6, cholesky/InspectionLevel_06.h, through oracle/_ref) for the parameters of
    scripts/eval.sh: costParam, levelParam, finalSeqNode.  Only where oracle/_ref was built."""
    R = ref()
    Ap = np.ascontiguousarray(A.Ap, np.int32)
    Ai = np.ascontiguousarray(A.Ai, np.int32)
    Ax = np.ascontiguousarray(A.Ax)
    pm = np.ascontiguousarray(perm, np.int32)
    nr = np.array(nrelax, np.int32)
    zr = np.array(zrelax, np.float64)
    h = R.ref_analyze(A.n, P(Ap), P(Ai), P(Ax), P(pm), P(nr), P(zr), cost, level, final)
    if not h:
        raise RuntimeError("ref_analyze failed")
    out = {}
    for name in ("levelPtr", "parPtr", "partition"):
        n = R.ref_get_int(h, name.encode(), None)
        a = np.zeros(max(int(n), 1), np.int32)
        R.ref_get_int(h, name.encode(), P(a))
        out[name] = a[: int(n)]
    sc = np.zeros(9, np.int64)
    R.ref_get_scalars(h, P(sc))
    return int(sc[6]), out["levelPtr"], out["parPtr"], out["partition"]
