"""Python face of the host inspector (csrc/inspector.cpp): produces the symbolic
objects the reference's `analyze_p2` (cholesky/LSparsity.h:256) hands to its
executors, plus the hoisted update lists."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _native as N


@dataclass
class Symbolic:
    """Arrays named as in the reference's BCSC (common/def.h:117-204)."""
    n: int
    nsuper: int
    ssize: int
    xsize: int
    nnzL: int
    nnzA: int
    maxSupWid: int
    maxCol: int
    flops_colcount: float
    flops_stored: float
    Perm: np.ndarray
    Parent: np.ndarray
    ColCount: np.ndarray
    super: np.ndarray
    col2Sup: np.ndarray
    sParent: np.ndarray
    p: np.ndarray        # uint64 (size_t), n+1
    i_ptr: np.ndarray    # uint64 (size_t), n+1
    s: np.ndarray
    A1p: np.ndarray
    A1i: np.ndarray
    A2p: np.ndarray
    A2i: np.ndarray
    A2x: np.ndarray
    A2src: np.ndarray
    levelPtr: np.ndarray
    levelSet: np.ndarray
    updPtr: np.ndarray
    updSn: np.ndarray
    updLb: np.ndarray
    updUb: np.ndarray
    _handle: object = field(default=None, repr=False)

    @property
    def nlevels(self) -> int:
        return len(self.levelPtr) - 1

    def permute_values(self, Ax: np.ndarray) -> np.ndarray:
        """Values of the lower triangle of P A P' (A2 order) for new values on the same pattern."""
        return np.ascontiguousarray(Ax[self.A2src])

    def __del__(self):
        h, self._handle = self._handle, None
        if h:
            try:
                N.lib().parsy_symbolic_free(h)
            except Exception:
                pass


def order_nd(A, leaf: int = 0) -> np.ndarray:
    """Fill-reducing ordering of an arbitrary pattern (graph nested dissection, csrc/ordering.cpp);
    perm[new] = old.  For matrices that come without an ordering (the reference calls METIS)."""
    Ap = np.ascontiguousarray(A.Ap, dtype=np.int32)
    Ai = np.ascontiguousarray(A.Ai, dtype=np.int32)
    perm = np.empty(A.n, dtype=np.int32)
    if N.lib().parsy_order_nd(A.n, N.ptr(Ap), N.ptr(Ai), leaf, N.ptr(perm)) != 0:
        raise RuntimeError(N.last_error())
    return perm


def analyze(A, perm=None, nrelax=(4, 16, 48), zrelax=(0.8, 0.1, 0.05)) -> Symbolic:
    lib = N.lib()
    Ap = np.ascontiguousarray(A.Ap, dtype=np.int32)
    Ai = np.ascontiguousarray(A.Ai, dtype=np.int32)
    Ax = np.ascontiguousarray(A.Ax, dtype=np.float64)
    pm = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
    nr = np.asarray(nrelax, dtype=np.int32)
    zr = np.asarray(zrelax, dtype=np.float64)
    h = lib.parsy_analyze(A.n, N.ptr(Ap), N.ptr(Ai), N.ptr(Ax), N.ptr(pm), N.ptr(nr), N.ptr(zr))
    if not h:
        raise RuntimeError(N.last_error())
    v = N.SymbolicView()
    lib.parsy_symbolic_get(h, C.byref(v))
    n, ns = v.n, v.nsuper
    g = N.view_array
    return Symbolic(
        n=n, nsuper=ns, ssize=v.ssize, xsize=v.xsize, nnzL=v.nnzL, nnzA=v.nnzA,
        maxSupWid=v.maxSupWid, maxCol=v.maxCol,
        flops_colcount=v.flops_colcount, flops_stored=v.flops_stored,
        Perm=g(v.Perm, n, np.int32), Parent=g(v.Parent, n, np.int32),
        ColCount=g(v.ColCount, n, np.int32), super=g(v.super, ns + 1, np.int32),
        col2Sup=g(v.col2Sup, n, np.int32), sParent=g(v.sParent, ns, np.int32),
        p=g(v.p, n + 1, np.uint64), i_ptr=g(v.i_ptr, n + 1, np.uint64),
        s=g(v.s, v.ssize, np.int32),
        A1p=g(v.A1p, n + 1, np.int32), A1i=g(v.A1i, v.nnzA, np.int32),
        A2p=g(v.A2p, n + 1, np.int32), A2i=g(v.A2i, v.nnzA, np.int32),
        A2x=g(v.A2x, v.nnzA, np.float64), A2src=g(v.A2src, v.nnzA, np.int32),
        levelPtr=g(v.levelPtr, v.nlevels + 1, np.int32), levelSet=g(v.levelSet, ns, np.int32),
        updPtr=g(v.updPtr, ns + 1, np.int64), updSn=g(v.updSn, v.n_updates, np.int32),
        updLb=g(v.updLb, v.n_updates, np.int32), updUb=g(v.updUb, v.n_updates, np.int32),
        _handle=h,
    )


def trivial_hlevel(sym: Symbolic):
    """A valid H-level schedule for the `_05` / H2 entry points built from the etree
    level sets: every supernode its own w-partition (nLevels, levelPtr, parPtr,
    partition as `getCoarseLevelSet_6` lays them out, cholesky/InspectionLevel_06.h:18)."""
    nl = sym.nlevels
    levelPtr = sym.levelPtr.astype(np.int32).copy()
    parPtr = np.arange(sym.nsuper + 1, dtype=np.int32)
    partition = sym.levelSet.astype(np.int32).copy()
    return nl, levelPtr, parPtr, partition


def bcsc_to_dense(sym: Symbolic, lValues: np.ndarray) -> np.ndarray:
    """Expand BCSC values to a dense lower-triangular matrix (tests only)."""
    L = np.zeros((sym.n, sym.n))
    for sn in range(sym.nsuper):
        c0, c1 = int(sym.super[sn]), int(sym.super[sn + 1])
        b, e = int(sym.i_ptr[c0]), int(sym.i_ptr[c1])
        rows = sym.s[b:e]
        r = e - b
        for c in range(c0, c1):
            base = int(sym.p[c])
            L[rows, c] = lValues[base:base + r]
    return L
