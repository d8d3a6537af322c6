"""Deterministic synthetic SPD matrices standing in for the SuiteSparse classes of
BASELINE.json (the files themselves -- reference scripts/dlMat.sh:5-21 -- cannot be
fetched offline) and geometric nested-dissection orderings (METIS, reference
cholesky/LSparsity.h:597, is absent).  See SURVEY.md 8(d).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _native as N


@dataclass
class LowerCSC:
    """Lower triangle of a symmetric matrix, CSC, rows sorted (what the
    reference's reader produces: common/Util.h:77-179)."""
    n: int
    Ap: np.ndarray  # int32 n+1
    Ai: np.ndarray  # int32 nnz
    Ax: np.ndarray  # float64 nnz

    @property
    def nnz(self) -> int:
        return int(self.Ap[-1])

    def to_dense(self) -> np.ndarray:
        A = np.zeros((self.n, self.n))
        for j in range(self.n):
            sl = slice(self.Ap[j], self.Ap[j + 1])
            A[self.Ai[sl], j] = self.Ax[sl]
        return A + np.tril(A, -1).T

    def to_scipy(self):
        import scipy.sparse as sp
        Lo = sp.csc_matrix((self.Ax, self.Ai, self.Ap), shape=(self.n, self.n))
        return (Lo + sp.tril(Lo, -1).T).tocsc()


def read_mtx(path: str) -> LowerCSC:
    """A MatrixMarket file as the reference reads it (common/Util.h:77-179: coordinate real, lower triangle,
    column-sorted; README.md:31).  Accepts general / symmetric files with either triangle or both: the lower triangle
    is taken and sorted (what examples/MakingLowerHalf.cpp prepares for the reference)."""
    import scipy.io
    import scipy.sparse as sp
    M_ = scipy.io.mmread(path)
    if M_.shape[0] != M_.shape[1]:
        raise ValueError(f"{path}: not a square matrix")
    M_ = sp.coo_matrix(M_)
    lo = M_.row >= M_.col
    up = M_.row < M_.col
    if up.any() and not lo.sum() > M_.shape[0]:   # an upper-triangular file: mirror it
        M_ = sp.coo_matrix((M_.data, (M_.col, M_.row)), shape=M_.shape)
        lo = M_.row >= M_.col
    L_ = sp.csc_matrix((M_.data[lo], (M_.row[lo], M_.col[lo])), shape=M_.shape)
    L_.sum_duplicates()
    L_.sort_indices()
    return LowerCSC(int(L_.shape[0]), L_.indptr.astype(np.int32), L_.indices.astype(np.int32),
                    L_.data.astype(np.float64))


def read_ordering(path: str, n: int) -> np.ndarray:
    """An ordering file as the reference's readOrdering takes it (common/Util.h:199-221): the dimension, then n
    permutation entries (new -> old, 0-based)."""
    vals = np.loadtxt(path, dtype=np.int64, comments="%").ravel()
    if len(vals) != n + 1 or vals[0] != n:
        raise ValueError(f"{path}: expected the dimension {n} followed by {n} entries")
    perm = vals[1:].astype(np.int32)
    if not np.array_equal(np.sort(perm), np.arange(n)):
        raise ValueError(f"{path}: not a permutation of 0..{n - 1}")
    return perm


def grid_spd(nx: int, ny: int, nz: int = 1, stencil: int = 5, shift: float = 0.1) -> LowerCSC:
    lib = N.lib()
    n = nx * ny * nz
    Ap = np.zeros(n + 1, dtype=np.int32)
    nnz = lib.parsy_grid_spd_lower(nx, ny, nz, stencil, shift, N.ptr(Ap), None, None)
    if nnz < 0:
        raise ValueError(N.last_error())
    Ai = np.zeros(nnz, dtype=np.int32)
    Ax = np.zeros(nnz, dtype=np.float64)
    lib.parsy_grid_spd_lower(nx, ny, nz, stencil, shift, N.ptr(Ap), N.ptr(Ai), N.ptr(Ax))
    return LowerCSC(n, Ap, Ai, Ax)


def grid_nd(nx: int, ny: int, nz: int = 1, leaf: int | None = None) -> np.ndarray:
    if leaf is None:
        leaf = 8 if nz == 1 else 27
    perm = np.zeros(nx * ny * nz, dtype=np.int32)
    if N.lib().parsy_grid_nested_dissection(nx, ny, nz, leaf, N.ptr(perm)) != 0:
        raise ValueError(N.last_error())
    return perm


def random_spd(n: int, density: float = 0.05, seed: int = 0) -> LowerCSC:
    """Small random sparse SPD matrix (diagonally dominant), for ragged-pattern tests."""
    rng = np.random.default_rng(seed)
    M = np.tril((rng.random((n, n)) < density) * rng.uniform(-1.0, 1.0, (n, n)), -1)
    A = M + M.T
    A[np.arange(n), np.arange(n)] = np.abs(A).sum(axis=1) + rng.uniform(0.5, 1.5, n)
    Ap = [0]
    Ai = []
    Ax = []
    for j in range(n):
        rows = np.nonzero(A[j:, j])[0] + j
        Ai.extend(rows.tolist())
        Ax.extend(A[rows, j].tolist())
        Ap.append(len(Ai))
    return LowerCSC(n, np.array(Ap, np.int32), np.array(Ai, np.int32), np.array(Ax, np.float64))


# name -> (nx, ny, nz, stencil, shift): the stand-ins of BASELINE.json's configs
WORKLOADS = {
    "ex15": (83, 83, 1, 5, 0.1),             # configs[0]: n = 6 889   (ex15: 6 867)
    "nd24k": (42, 42, 42, 27, 0.1),          # configs[1]: n = 74 088  (nd24k: 72 000)
    "flan": (116, 116, 116, 27, 0.1),        # configs[2]: n = 1 560 896 (Flan_1565: 1 564 794)
    "parabolic_fem": (725, 725, 1, 5, 0.1),  # configs[3]: n = 525 625 (parabolic_fem: 525 825)
    # small ones for tests
    "tiny2d": (12, 12, 1, 5, 0.1),
    "small3d": (10, 10, 10, 27, 0.1),
    "mid3d": (20, 20, 20, 27, 0.1),
    "lap30": (30, 30, 30, 7, 0.01),          # the survey's probe problem
}


def workload(name: str):
    """A named workload, or "NXxNYxNZ[:stencil]" for an ad-hoc grid (diagnostics)."""
    if name not in WORKLOADS and "x" in name:
        dims, _, st = name.partition(":")
        nx, ny, nz = (int(v) for v in dims.split("x"))
        return grid_spd(nx, ny, nz, int(st or 27), 0.1), grid_nd(nx, ny, nz)
    nx, ny, nz, st, sh = WORKLOADS[name]
    return grid_spd(nx, ny, nz, st, sh), grid_nd(nx, ny, nz)
