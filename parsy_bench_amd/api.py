"""Host-side mirror of the reference's operator interface for the hot path.

Two layers, both thin ctypes shims over libparsy_amd.so (the C ABI in
include/parsy_amd.h) -- no numerics happen in Python:

* the drop-in operators, with the reference's names, argument order and return
  values (cholesky/parallel_PB_Cholesky_05.h:27, Parallel_PB_Cholesky_wavefront.h:10,
  triangularSolve/Triangular_BCSC.h:14/115/171/238), taking numpy arrays where the
  reference takes raw pointers;
* `Plan`, the pattern-resident handle API (device pointers in/out) used by
  bench.py and by multi-GPU sharding.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N

KIND_NAMES = ["SMALL", "TILES", "CHAIN", "BIG", "BACK_BELOW", "SOLVE_SMALL", "SOLVE_PANEL", "SOLVE_FIXUP", "BACK_BLOCK", "DENSE"]


def device_count() -> int:
    return int(N.lib().parsy_device_count())


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _sz(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------------------------------------------------------------------
# drop-in operators (host arrays in, host arrays out)
# ---------------------------------------------------------------------------
def cholesky_left_par_05(n, c, r, values, lC, lR, Li_ptr, lValues, blockSet, supNo, timing, aTree,
                         cT, rT, col2Sup, nLevels, levelPtr, levelSet, nPar, parPtr, partition,
                         chunk, threads, super_max, col_max, nodCost=None) -> bool:
    """lValues (zeroed by the caller) receives the factor; returns False on a
    non-positive pivot or if the HIP path could not run."""
    assert lValues.dtype == np.float64 and lValues.flags["C_CONTIGUOUS"]
    a = [_i32(c), _i32(r), _f64(values), _sz(lC), _i32(lR), _sz(Li_ptr), _i32(blockSet), _i32(aTree),
         _i32(cT), _i32(rT), _i32(col2Sup), _i32(levelPtr), _i32(parPtr), _i32(partition)]
    ls = None if levelSet is None else _i32(levelSet)
    return bool(N.lib().cholesky_left_par_05(
        n, N.ptr(a[0]), N.ptr(a[1]), N.ptr(a[2]), N.ptr(a[3]), N.ptr(a[4]), N.ptr(a[5]), N.ptr(lValues),
        N.ptr(a[6]), supNo, N.ptr(timing), N.ptr(a[7]), N.ptr(a[8]), N.ptr(a[9]), N.ptr(a[10]), nLevels,
        N.ptr(a[11]), N.ptr(ls), nPar, N.ptr(a[12]), N.ptr(a[13]), chunk, threads, super_max, col_max,
        None))


def cholesky_left_par_05_prune(n, c, r, values, lC, lR, Li_ptr, lValues, blockSet, supNo, timing, prunePtr,
                               pruneSet, nLevels, levelPtr, levelSet, nPar, parPtr, partition, chunk, threads,
                               super_max, col_max, nodCost=None) -> bool:
    """The reference's PRUNE build of cholesky_left_par_05: update lists instead of etree + upper pattern."""
    assert lValues.dtype == np.float64 and lValues.flags["C_CONTIGUOUS"]
    a = [_i32(c), _i32(r), _f64(values), _sz(lC), _i32(lR), _sz(Li_ptr), _i32(blockSet), _i32(prunePtr),
         _i32(pruneSet), _i32(levelPtr), _i32(parPtr), _i32(partition)]
    ls = None if levelSet is None else _i32(levelSet)
    return bool(N.lib().cholesky_left_par_05_prune(
        n, N.ptr(a[0]), N.ptr(a[1]), N.ptr(a[2]), N.ptr(a[3]), N.ptr(a[4]), N.ptr(a[5]), N.ptr(lValues),
        N.ptr(a[6]), supNo, N.ptr(timing), N.ptr(a[7]), N.ptr(a[8]), nLevels, N.ptr(a[9]), N.ptr(ls), nPar,
        N.ptr(a[10]), N.ptr(a[11]), chunk, threads, super_max, col_max, None))


def cholesky_left_par_waveFront(n, c, r, values, lC, lR, Li_ptr, lValues, blockSet, supNo, timing,
                                aTree, cT, rT, col2Sup, nLevels, levelPtr, levelSet, chunk, threads,
                                super_max, col_max) -> bool:
    assert lValues.dtype == np.float64 and lValues.flags["C_CONTIGUOUS"]
    a = [_i32(c), _i32(r), _f64(values), _sz(lC), _i32(lR), _sz(Li_ptr), _i32(blockSet), _i32(aTree),
         _i32(cT), _i32(rT), _i32(col2Sup), _i32(levelPtr), _i32(levelSet)]
    return bool(N.lib().cholesky_left_par_waveFront(
        n, N.ptr(a[0]), N.ptr(a[1]), N.ptr(a[2]), N.ptr(a[3]), N.ptr(a[4]), N.ptr(a[5]), N.ptr(lValues),
        N.ptr(a[6]), supNo, N.ptr(timing), N.ptr(a[7]), N.ptr(a[8]), N.ptr(a[9]), N.ptr(a[10]), nLevels,
        N.ptr(a[11]), N.ptr(a[12]), chunk, threads, super_max, col_max))


def _solve_base(n, Lp, Li, Lx, NNZ, Li_ptr, col2sup, sup2col, supNo, x):
    assert x is None or (x.dtype == np.float64 and x.flags["C_CONTIGUOUS"])
    keep = [None if Lp is None else _sz(Lp), None if Li is None else _i32(Li), _f64(Lx), _sz(Li_ptr),
            _i32(col2sup), _i32(sup2col)]
    return keep, (n, N.ptr(keep[0]), N.ptr(keep[1]), N.ptr(keep[2]), int(NNZ), N.ptr(keep[3]),
                  N.ptr(keep[4]), N.ptr(keep[5]), supNo, N.ptr(x))


def blockedLsolve(n, Lp, Li, Lx, NNZ, Li_ptr, col2sup, sup2col, supNo, x) -> int:
    keep, base = _solve_base(n, Lp, Li, Lx, NNZ, Li_ptr, col2sup, sup2col, supNo, x)
    return int(N.lib().blockedLsolve(*base))


def leveledBlockedLsolve(n, Lp, Li, Lx, NNZ, Li_ptr, col2sup, sup2col, supNo, x, levels, levelPtr,
                         levelSet, chunk) -> int:
    keep, base = _solve_base(n, Lp, Li, Lx, NNZ, Li_ptr, col2sup, sup2col, supNo, x)
    lp, ls = _i32(levelPtr), _i32(levelSet)
    return int(N.lib().leveledBlockedLsolve(*base, levels, N.ptr(lp), N.ptr(ls), chunk))


def H2LeveledBlockedLsolve(n, Lp, Li, Lx, NNZ, Li_ptr, col2sup, sup2col, supNo, x, levels, levelPtr,
                           levelSet, parts, parPtr, partition, chunk) -> int:
    keep, base = _solve_base(n, Lp, Li, Lx, NNZ, Li_ptr, col2sup, sup2col, supNo, x)
    lp, pp, pt = _i32(levelPtr), _i32(parPtr), _i32(partition)
    ls = None if levelSet is None else _i32(levelSet)
    return int(N.lib().H2LeveledBlockedLsolve(*base, levels, N.ptr(lp), N.ptr(ls), parts, N.ptr(pp),
                                              N.ptr(pt), chunk))


def H2LeveledBlockedLsolve_Peeled(n, Lp, Li, Lx, NNZ, Li_ptr, col2sup, sup2col, supNo, x, levels,
                                  levelPtr, levelSet, parts, parPtr, partition, chunk, threads) -> int:
    keep, base = _solve_base(n, Lp, Li, Lx, NNZ, Li_ptr, col2sup, sup2col, supNo, x)
    lp, pp, pt = _i32(levelPtr), _i32(parPtr), _i32(partition)
    ls = None if levelSet is None else _i32(levelSet)
    return int(N.lib().H2LeveledBlockedLsolve_Peeled(*base, levels, N.ptr(lp), N.ptr(ls), parts,
                                                     N.ptr(pp), N.ptr(pt), chunk, threads))


def dropin_reset() -> None:
    N.lib().parsy_dropin_reset()


# ---------------------------------------------------------------------------
# plan API
# ---------------------------------------------------------------------------
class Plan:
    """Pattern + launch schedule resident on one device (parsy_plan)."""

    def __init__(self, sym, device: int = 0):
        lib = N.lib()
        self.sym = sym
        self.device = device
        if getattr(sym, "_handle", None):
            h = lib.parsy_plan_from_symbolic(sym._handle, device)
        else:
            a = [_i32(sym.super), _sz(sym.p), _sz(sym.i_ptr), _i32(sym.s), _i32(sym.sParent),
                 _i32(sym.col2Sup), _i32(sym.A1p), _i32(sym.A1i), _i32(sym.A2p), _i32(sym.A2i)]
            h = lib.parsy_plan_create(sym.n, sym.nsuper, *[N.ptr(v) for v in a], device)
        if not h:
            raise RuntimeError("parsy_plan_create failed: " + N.last_error())
        self._h = h

    def close(self):
        h, self._h = self._h, None
        if h:
            N.lib().parsy_plan_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def info(self) -> dict:
        pi = N.PlanInfo()
        N.lib().parsy_plan_get_info(self._h, C.byref(pi))
        return pi.as_dict()

    def set_active(self, mask) -> None:
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        if N.lib().parsy_plan_set_active(self._h, N.ptr(m)) != 0:
            raise RuntimeError(N.last_error())

    # host-buffer conveniences -------------------------------------------------
    def factor(self, values, out=None):
        """Returns (lValues, device_seconds). Raises on a HIP failure; check `status()`.  out: a float64 array of
        xsize entries to receive lValues (every entry is written)."""
        vals = _f64(values)
        lValues = np.zeros(int(self.sym.xsize), dtype=np.float64) if out is None else out
        assert lValues.dtype == np.float64 and lValues.size == int(self.sym.xsize) and lValues.flags.c_contiguous
        sec = C.c_double(0)
        if N.lib().parsy_factor_host(self._h, N.ptr(vals), N.ptr(lValues), C.byref(sec)) != 0:
            raise RuntimeError("parsy_factor_host failed: " + N.last_error())
        return lValues, sec.value

    def status(self) -> int:
        return int(N.lib().parsy_factor_status(self._h))

    def solve_status(self) -> int:
        """0 = the last solve on this plan completed; -1 = a hand-off wait inside it timed out."""
        return int(N.lib().parsy_solve_status(self._h))

    def solve(self, lValues, b):
        """Forward solve; b is (n,) or (n, nrhs) (any layout); returns x of the same shape."""
        b = np.asarray(b, dtype=np.float64)
        one = b.ndim == 1
        X = np.asfortranarray(b.reshape(self.sym.n, -1)).copy(order="F")
        nrhs = X.shape[1]
        lv = _f64(lValues)
        sec = C.c_double(0)
        rc = N.lib().parsy_solve_host(self._h, N.ptr(lv), X.ctypes.data_as(C.c_void_p), nrhs,
                                      self.sym.n, C.byref(sec))
        if rc != 0:
            raise RuntimeError("parsy_solve_host failed: " + N.last_error())
        return (X[:, 0].copy() if one else np.ascontiguousarray(X)), sec.value

    def solve2(self, lValues, b, forward: bool = True):
        """Backward solve L' x = y (forward=False) or the full L L' x = b (forward=True) of the
        PERMUTED system; returns (x, device_seconds)."""
        b = np.asarray(b, dtype=np.float64)
        one = b.ndim == 1
        X = np.asfortranarray(b.reshape(self.sym.n, -1)).copy(order="F")
        lv = _f64(lValues)
        sec = C.c_double(0)
        rc = N.lib().parsy_solve2_host(self._h, N.ptr(lv), X.ctypes.data_as(C.c_void_p), X.shape[1],
                                       self.sym.n, 1 if forward else 0, C.byref(sec))
        if rc != 0:
            raise RuntimeError("parsy_solve2_host failed: " + N.last_error())
        return (X[:, 0].copy() if one else np.ascontiguousarray(X)), sec.value

    def solve_spd(self, lValues, b):
        """x with A x = b for the ORIGINAL matrix: x = P' L'^-1 L^-1 P b (Perm from the inspector)."""
        b = np.asarray(b, dtype=np.float64)
        pb = b[self.sym.Perm] if b.ndim == 1 else b[self.sym.Perm, :]
        z, sec = self.solve2(lValues, pb, forward=True)
        x = np.empty_like(z)
        x[self.sym.Perm] = z
        return x, sec

    def backsolve_device(self, d_lValues: int, d_x: int, nrhs: int, ldx: int, stream: int = 0) -> None:
        if N.lib().parsy_backsolve_device(self._h, d_lValues, d_x, nrhs, ldx, stream) != 0:
            raise RuntimeError("parsy_backsolve_device failed: " + N.last_error())

    def rhs_ones_device(self, d_lValues: int, d_b: int, stream: int = 0) -> None:
        """d_b = L * 1 on the stored structure (the reference's rhsInitBlocked, common/Util.h:277)."""
        if N.lib().parsy_rhs_ones_device(self._h, d_lValues, d_b, stream) != 0:
            raise RuntimeError("parsy_rhs_ones_device failed: " + N.last_error())

    # device-pointer API ---------------------------------------------------------
    def factor_device(self, d_values: int, d_lValues: int, stream: int = 0, init: bool = True) -> None:
        if N.lib().parsy_factor_device_ex(self._h, d_values, d_lValues, stream, 0 if init else 1) != 0:
            raise RuntimeError("parsy_factor_device failed: " + N.last_error())

    def solve_device(self, d_lValues: int, d_x: int, nrhs: int, ldx: int, stream: int = 0) -> None:
        if N.lib().parsy_solve_device(self._h, d_lValues, d_x, nrhs, ldx, stream) != 0:
            raise RuntimeError("parsy_solve_device failed: " + N.last_error())

    def solve_levels(self) -> np.ndarray:
        """The etree level of every supernode, as the launches of the solves go by it (parsy_plan_solve_levels)."""
        out = np.zeros(self.sym.nsuper, dtype=np.int32)
        N.lib().parsy_plan_solve_levels(self._h, N.ptr(out))
        return out

    def solve_levels_device(self, d_lValues: int, d_x: int, nrhs: int, ldx: int, stream: int, level_begin: int, level_end: int,
                            first: bool, last: bool, backward: bool = False) -> None:
        """One step of a solve that goes level by level (parsy_solve_levels_device): the active supernodes of the etree
        levels [level_begin, level_end)."""
        flags = (1 if first else 0) | (2 if last else 0) | (4 if backward else 0)
        if N.lib().parsy_solve_levels_device(self._h, d_lValues, d_x, nrhs, ldx, stream, level_begin, level_end, flags) != 0:
            raise RuntimeError("parsy_solve_levels_device failed: " + N.last_error())

    # level by level (the steps of a multi-device run; parsy_factor_device is exactly this sequence) ----
    def factor_begin(self, d_values: int, d_lValues: int, stream: int = 0, init: bool = True) -> None:
        if N.lib().parsy_factor_begin(self._h, d_values, d_lValues, stream, 0 if init else 1) != 0:
            raise RuntimeError("parsy_factor_begin failed: " + N.last_error())

    def factor_level(self, level: int, d_lValues: int, stream: int = 0) -> None:
        if N.lib().parsy_factor_level(self._h, level, d_lValues, stream) != 0:
            raise RuntimeError("parsy_factor_level failed: " + N.last_error())

    def factor_end(self, stream: int = 0) -> None:
        if N.lib().parsy_factor_end(self._h, stream) != 0:
            raise RuntimeError("parsy_factor_end failed: " + N.last_error())

    def pieces(self) -> dict:
        """The pieces of the Cholesky view: supernode, level, col0, width, rows, value_begin, value_end."""
        n = int(N.lib().parsy_plan_pieces(self._h, *([None] * 7)))
        out = {k: np.zeros(n, dtype=np.int32) for k in ("supernode", "level", "col0", "width", "rows")}
        out.update({k: np.zeros(n, dtype=np.int64) for k in ("value_begin", "value_end")})
        N.lib().parsy_plan_pieces(self._h, *[N.ptr(out[k]) for k in (
            "supernode", "level", "col0", "width", "rows", "value_begin", "value_end")])
        return out

    def set_active_pieces(self, mask) -> None:
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        if N.lib().parsy_plan_set_active_pieces(self._h, N.ptr(m)) != 0:
            raise RuntimeError(N.last_error())

    def check(self) -> int:
        return int(N.lib().parsy_plan_check(self._h))

    def last_factor_ms(self) -> float:
        return float(N.lib().parsy_last_factor_ms(self._h))

    def last_solve_ms(self) -> float:
        return float(N.lib().parsy_last_solve_ms(self._h))

    def profile(self, mode: int) -> None:
        N.lib().parsy_plan_profile(self._h, mode)

    def profile_collect(self) -> None:
        if N.lib().parsy_plan_profile_collect(self._h) != 0:
            raise RuntimeError("profile_collect: no profiled run to collect")

    def profile_get(self) -> dict:
        ms = np.zeros(10, dtype=np.float64)
        cnt = np.zeros(10, dtype=np.int32)
        runs = C.c_int(0)
        N.lib().parsy_plan_profile_get(self._h, N.ptr(ms), N.ptr(cnt), C.byref(runs))
        return {"runs": runs.value, "ms": dict(zip(KIND_NAMES, ms.tolist())),
                "launches": dict(zip(KIND_NAMES, cnt.tolist()))}


# ---------------------------------------------------------------------------
# distribution of one factorization over the devices of a node
# ---------------------------------------------------------------------------
class Dist:
    """Ownership of the pieces and the messages that follow every level (parsy_dist; host logic)."""

    def __init__(self, plan: "Plan", nranks: int, block: int = 0, _borrowed=None):
        self._own = _borrowed is None
        self._h = _borrowed or N.lib().parsy_dist_create(plan._h, nranks, block)
        if not self._h:
            raise RuntimeError("parsy_dist_create failed: " + N.last_error())
        di = N.DistInfo()
        N.lib().parsy_dist_get_info(self._h, C.byref(di))
        self.info = di.as_dict()
        self.nranks, self.nlevels = di.nranks, di.nlevels
        self.owner = np.zeros(di.n_pieces, dtype=np.int32)
        self.in_subtree = np.zeros(di.n_pieces, dtype=np.uint8)
        self.rank_cost = np.zeros(di.nranks)
        self.level_cost = np.zeros((di.nlevels, di.nranks))
        N.lib().parsy_dist_get(self._h, N.ptr(self.owner), N.ptr(self.in_subtree), N.ptr(self.rank_cost),
                               N.ptr(self.level_cost))

    def close(self):
        h, self._h = self._h, None
        if h and self._own:
            N.lib().parsy_dist_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, plan: "Plan") -> int:
        return int(N.lib().parsy_dist_check(plan._h, self._h))

    def mask(self, rank: int) -> np.ndarray:
        return (self.owner == rank).astype(np.uint8)

    def messages(self, level: int, rank: int | None = None):
        """The messages that follow `level` (those `rank` sends or receives when given):
        (src, dst, off int64[], len int32[], packed int64[], total)."""
        lib = N.lib()
        out = []
        for k in range(max(int(lib.parsy_dist_level_messages(self._h, level)), 0)):
            src, dst = C.c_int32(0), C.c_int32(0)
            nseg, total = C.c_int64(0), C.c_int64(0)
            # (who talks to whom first: the arrays of a message are copied only for the ranks that take part in it)
            if lib.parsy_dist_message(self._h, level, k, C.byref(src), C.byref(dst), C.byref(nseg), C.byref(total),
                                      None, None, None) != 0:
                raise RuntimeError(N.last_error())
            if rank is not None and rank not in (src.value, dst.value):
                continue
            off, ln, pk = N.c_i64_p(), N.c_int_p(), N.c_i64_p()
            lib.parsy_dist_message(self._h, level, k, None, None, None, None, C.byref(off), C.byref(ln), C.byref(pk))
            out.append((src.value, dst.value, N.view_array(off, nseg.value, np.int64),
                        N.view_array(ln, nseg.value, np.int32), N.view_array(pk, nseg.value, np.int64), total.value))
        return out


class MultiDevice:
    """One process driving several devices (parsy_mg): `devices` lists one HIP device per rank and may
    repeat a device (several ranks share it)."""

    def __init__(self, sym, devices, block: int = 0):
        if not getattr(sym, "_handle", None):
            raise RuntimeError("MultiDevice needs an inspector result (parsy_symbolic)")
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        self.sym = sym
        self.nranks = len(dv)
        self._h = N.lib().parsy_mg_create(sym._handle, len(dv), N.ptr(dv), block)
        if not self._h:
            raise RuntimeError("parsy_mg_create failed: " + N.last_error())
        self.dist = Dist(None, len(dv), _borrowed=N.lib().parsy_mg_dist(self._h))

    def close(self):
        h, self._h = self._h, None
        if h:
            self.dist._h = None
            N.lib().parsy_mg_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_values(self, values) -> None:
        v = _f64(values)
        if N.lib().parsy_mg_set_values(self._h, N.ptr(v)) != 0:
            raise RuntimeError("parsy_mg_set_values failed: " + N.last_error())

    def factor(self):
        """One distributed factorization; returns (status, wall seconds)."""
        sec = C.c_double(0)
        st = int(N.lib().parsy_mg_factor(self._h, C.byref(sec)))
        if st < 0:
            raise RuntimeError("parsy_mg_factor failed: " + N.last_error())
        return st, sec.value

    def profile(self):
        """One factorization with the ranks taking turns (every step alone on its device) and every launch timed:
        (status, main_ms, side_ms, copy_ms), each nranks x levels."""
        shape = (self.nranks, self.dist.nlevels)
        main, side, copy = np.zeros(shape), np.zeros(shape), np.zeros(shape)
        st = int(N.lib().parsy_mg_profile(self._h, N.ptr(main), N.ptr(side), N.ptr(copy)))
        if st < 0:
            raise RuntimeError("parsy_mg_profile failed: " + N.last_error())
        return st, main, side, copy

    def rank_ms(self) -> np.ndarray:
        out = np.zeros(self.nranks)
        N.lib().parsy_mg_rank_ms(self._h, N.ptr(out))
        return out

    def gather(self) -> np.ndarray:
        lv = np.zeros(int(self.sym.xsize), dtype=np.float64)
        if N.lib().parsy_mg_gather_host(self._h, N.ptr(lv)) != 0:
            raise RuntimeError("parsy_mg_gather_host failed: " + N.last_error())
        return lv
