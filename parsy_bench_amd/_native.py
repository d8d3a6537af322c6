"""ctypes binding of libparsy_amd.so (declared in include/parsy_amd.h).

The library is the product; there is no Python or CPU fallback behind it.  If it
is missing it is built on first use (hipcc cross-compiles without a GPU); if that
fails, importing callers get a loud RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
# PARSY_LIB: a diagnostic build of the same library (tools/: ablations, tuning variants) instead of the product
_LIB_PATH = Path(os.environ["PARSY_LIB"]) if os.environ.get("PARSY_LIB") else _PKG / "libparsy_amd.so"
_lib = None

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)
c_size_p = C.POINTER(C.c_size_t)
c_i64_p = C.POINTER(C.c_int64)


class SymbolicView(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("nsuper", C.c_int32), ("nlevels", C.c_int32),
        ("maxSupWid", C.c_int32), ("maxCol", C.c_int32),
        ("ssize", C.c_int64), ("xsize", C.c_int64), ("nnzL", C.c_int64),
        ("nnzA", C.c_int64), ("n_updates", C.c_int64),
        ("flops_colcount", C.c_double), ("flops_stored", C.c_double),
        ("Perm", c_int_p), ("Parent", c_int_p), ("ColCount", c_int_p),
        ("super", c_int_p), ("col2Sup", c_int_p), ("sParent", c_int_p),
        ("p", c_size_p), ("i_ptr", c_size_p), ("s", c_int_p),
        ("A1p", c_int_p), ("A1i", c_int_p),
        ("A2p", c_int_p), ("A2i", c_int_p), ("A2x", c_dbl_p), ("A2src", c_int_p),
        ("levelPtr", c_int_p), ("levelSet", c_int_p),
        ("updPtr", c_i64_p), ("updSn", c_int_p), ("updLb", c_int_p), ("updUb", c_int_p),
    ]


class PlanInfo(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("nsuper", C.c_int32), ("nlevels", C.c_int32),
        ("max_width", C.c_int32), ("max_rows", C.c_int32),
        ("n_small", C.c_int32), ("n_big", C.c_int32),
        ("chol_launches", C.c_int32), ("solve_launches", C.c_int32),
        ("nnzA", C.c_int64), ("ssize", C.c_int64), ("xsize", C.c_int64), ("nnzL", C.c_int64),
        ("n_updates", C.c_int64), ("relpos_len", C.c_int64), ("device_bytes", C.c_int64),
        ("flops_stored", C.c_double), ("update_flops", C.c_double), ("reread_bytes", C.c_double),
        ("inner_flops", C.c_double), ("tile_update_flops", C.c_double),
        ("big_flops", C.c_double), ("big_entries", C.c_int64), ("big_tasks", C.c_int32),
        ("n_pieces", C.c_int32), ("chol_levels", C.c_int32), ("piece_width", C.c_int32),
        ("big_min_k", C.c_int32),
        ("chol_subtrees", C.c_int32), ("chol_subtree_supernodes", C.c_int32),
        ("solve_subtrees", C.c_int32), ("solve_subtree_supernodes", C.c_int32),
        ("backsolve_launches", C.c_int32), ("dense_tasks", C.c_int32),
        ("dense_flops", C.c_double), ("dense_entries", C.c_int64),
        ("solve_one", C.c_int32), ("solve_one_blocks", C.c_int32),
        ("sub_mrhs_trees", C.c_int32), ("sub_mrhs_slots", C.c_int32),
        ("sub_mrhs_tiers", C.c_int32), ("sub_mrhs_cover_level", C.c_int32),
        ("dense_strip_entries", C.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/parsy_amd.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = [
    "cholesky_left_par_05", "cholesky_left_par_05_prune", "cholesky_left_par_waveFront", "blockedLsolve",
    "leveledBlockedLsolve", "H2LeveledBlockedLsolve", "H2LeveledBlockedLsolve_Peeled",
    "parsy_dropin_reset", "parsy_plan_create", "parsy_plan_destroy", "parsy_plan_get_info",
    "parsy_plan_set_active", "parsy_plan_chain_check", "parsy_factor_device", "parsy_factor_status", "parsy_solve_device",
    "parsy_factor_host", "parsy_solve_host", "parsy_last_factor_ms", "parsy_last_solve_ms",
    "parsy_last_error", "parsy_device_count", "parsy_analyze", "parsy_symbolic_free",
    "parsy_symbolic_get", "parsy_plan_from_symbolic", "parsy_grid_spd_lower",
    "parsy_grid_nested_dissection", "parsy_order_nd", "parsy_plan_profile", "parsy_plan_profile_collect",
    "parsy_plan_profile_get", "parsy_factor_device_ex", "parsy_backsolve_device", "parsy_solve_levels_device", "parsy_plan_solve_levels", "parsy_solve2_host",
    "parsy_rhs_ones_device", "parsy_solve_status", "parsy_copy_segments_device", "parsy_plan_check",
    "parsy_factor_begin", "parsy_factor_level", "parsy_factor_end", "parsy_plan_pieces",
    "parsy_plan_set_active_pieces", "parsy_dist_create", "parsy_dist_destroy", "parsy_dist_get_info",
    "parsy_dist_get", "parsy_dist_level_messages", "parsy_dist_message", "parsy_dist_check",
    "parsy_mg_create", "parsy_mg_destroy", "parsy_mg_set_values", "parsy_mg_factor", "parsy_mg_rank_ms",
    "parsy_mg_gather_host", "parsy_mg_dist", "parsy_mg_plan", "parsy_mg_profile", "parsy_plan_profile_levels",
]


class DistInfo(C.Structure):
    _fields_ = [
        ("nranks", C.c_int32), ("nlevels", C.c_int32), ("n_pieces", C.c_int32), ("n_subtrees", C.c_int32),
        ("n_root_pieces", C.c_int32), ("n_messages", C.c_int32), ("exchange_elements", C.c_int64),
        ("total_cost", C.c_double), ("root_cost", C.c_double), ("max_rank_cost", C.c_double),
        ("lockstep_cost", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def _declare(lib):
    vp = C.c_void_p
    lib.parsy_last_error.restype = C.c_char_p
    lib.parsy_device_count.restype = C.c_int
    lib.parsy_analyze.restype = vp
    lib.parsy_analyze.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp]
    lib.parsy_symbolic_free.argtypes = [vp]
    lib.parsy_symbolic_get.argtypes = [vp, C.POINTER(SymbolicView)]
    lib.parsy_grid_spd_lower.restype = C.c_int64
    lib.parsy_grid_spd_lower.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp, vp, vp]
    lib.parsy_grid_nested_dissection.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.parsy_order_nd.argtypes = [C.c_int, vp, vp, C.c_int, vp]
    lib.parsy_plan_create.restype = vp
    lib.parsy_plan_create.argtypes = [C.c_int, C.c_int] + [vp] * 10 + [C.c_int]
    lib.parsy_plan_from_symbolic.restype = vp
    lib.parsy_plan_from_symbolic.argtypes = [vp, C.c_int]
    lib.parsy_plan_destroy.argtypes = [vp]
    lib.parsy_plan_get_info.argtypes = [vp, C.POINTER(PlanInfo)]
    lib.parsy_plan_set_active.argtypes = [vp, vp]
    lib.parsy_plan_chain_check.argtypes = [vp, C.c_int]
    lib.parsy_plan_chain_check.restype = C.c_longlong
    lib.parsy_plan_check.argtypes = [vp]
    lib.parsy_plan_check.restype = C.c_longlong
    lib.parsy_factor_device.argtypes = [vp, vp, vp, vp]
    lib.parsy_factor_device_ex.argtypes = [vp, vp, vp, vp, C.c_int]
    lib.parsy_factor_status.argtypes = [vp]
    lib.parsy_solve_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
    lib.parsy_plan_solve_levels.argtypes = [vp, vp]
    lib.parsy_solve_levels_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    lib.parsy_backsolve_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
    lib.parsy_solve2_host.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.parsy_rhs_ones_device.argtypes = [vp, vp, vp, vp]
    lib.parsy_solve_status.argtypes = [vp]
    lib.parsy_copy_segments_device.argtypes = [vp, vp, vp, vp, vp, C.c_int64, vp]
    lib.parsy_factor_begin.argtypes = [vp, vp, vp, vp, C.c_int]
    lib.parsy_factor_level.argtypes = [vp, C.c_int, vp, vp]
    lib.parsy_factor_end.argtypes = [vp, vp]
    lib.parsy_plan_pieces.argtypes = [vp] + [vp] * 7
    lib.parsy_plan_set_active_pieces.argtypes = [vp, vp]
    lib.parsy_dist_create.restype = vp
    lib.parsy_dist_create.argtypes = [vp, C.c_int, C.c_int]
    lib.parsy_dist_destroy.argtypes = [vp]
    lib.parsy_dist_get_info.argtypes = [vp, C.POINTER(DistInfo)]
    lib.parsy_dist_get.argtypes = [vp, vp, vp, vp, vp]
    lib.parsy_dist_level_messages.argtypes = [vp, C.c_int]
    lib.parsy_dist_message.argtypes = [vp, C.c_int, C.c_int, c_int_p, c_int_p, c_i64_p, c_i64_p,
                                       C.POINTER(c_i64_p), C.POINTER(c_int_p), C.POINTER(c_i64_p)]
    lib.parsy_dist_check.argtypes = [vp, vp]
    lib.parsy_dist_check.restype = C.c_longlong
    lib.parsy_mg_create.restype = vp
    lib.parsy_mg_create.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.parsy_mg_destroy.argtypes = [vp]
    lib.parsy_mg_set_values.argtypes = [vp, vp]
    lib.parsy_mg_factor.argtypes = [vp, vp]
    lib.parsy_mg_rank_ms.argtypes = [vp, vp]
    lib.parsy_mg_profile.argtypes = [vp, vp, vp, vp]
    lib.parsy_plan_profile_levels.argtypes = [vp, vp, vp]
    lib.parsy_mg_gather_host.argtypes = [vp, vp]
    lib.parsy_mg_dist.restype = vp
    lib.parsy_mg_dist.argtypes = [vp]
    lib.parsy_mg_plan.restype = vp
    lib.parsy_mg_plan.argtypes = [vp, C.c_int]
    lib.parsy_factor_host.argtypes = [vp, vp, vp, vp]
    lib.parsy_solve_host.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
    lib.parsy_last_factor_ms.restype = C.c_double
    lib.parsy_last_factor_ms.argtypes = [vp]
    lib.parsy_last_solve_ms.restype = C.c_double
    lib.parsy_last_solve_ms.argtypes = [vp]
    lib.parsy_plan_profile.argtypes = [vp, C.c_int]
    lib.parsy_plan_profile_collect.argtypes = [vp]
    lib.parsy_plan_profile_get.argtypes = [vp, vp, vp, vp]
    lib.cholesky_left_par_05.restype = C.c_bool
    lib.cholesky_left_par_05.argtypes = (
        [C.c_int] + [vp] * 8 + [C.c_int] + [vp] * 5 + [C.c_int, vp, vp, C.c_int, vp, vp]
        + [C.c_int] * 4 + [vp])
    lib.cholesky_left_par_05_prune.restype = C.c_bool
    lib.cholesky_left_par_05_prune.argtypes = (
        [C.c_int] + [vp] * 8 + [C.c_int] + [vp] * 3 + [C.c_int, vp, vp, C.c_int, vp, vp]
        + [C.c_int] * 4 + [vp])
    lib.cholesky_left_par_waveFront.restype = C.c_bool
    lib.cholesky_left_par_waveFront.argtypes = (
        [C.c_int] + [vp] * 8 + [C.c_int] + [vp] * 5 + [C.c_int, vp, vp] + [C.c_int] * 4)
    base = [C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp]
    lib.blockedLsolve.argtypes = base
    lib.leveledBlockedLsolve.argtypes = base + [C.c_int, vp, vp, C.c_int]
    lib.H2LeveledBlockedLsolve.argtypes = base + [C.c_int, vp, vp, C.c_int, vp, vp, C.c_int]
    lib.H2LeveledBlockedLsolve_Peeled.argtypes = base + [C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int]
    for name in ("blockedLsolve", "leveledBlockedLsolve", "H2LeveledBlockedLsolve",
                 "H2LeveledBlockedLsolve_Peeled"):
        getattr(lib, name).restype = C.c_int


def lib():
    """Load (building if needed) libparsy_amd.so. Raises if it cannot be had."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists() or os.environ.get("PARSY_REBUILD"):
        from .build import build_native
        build_native()
    try:
        loaded = C.CDLL(str(_LIB_PATH), mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"libparsy_amd.so could not be loaded ({e}); the HIP executor is the "
                           "only implementation of this package -- there is no fallback") from e
    _declare(loaded)
    _lib = loaded
    return _lib


def last_error() -> str:
    msg = lib().parsy_last_error()
    return msg.decode() if msg else ""


def ptr(a):
    """Raw pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def view_array(p, n, dtype):
    """Copy n elements from a ctypes pointer into a fresh numpy array."""
    if n == 0 or not p:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(p, shape=(int(n),)).astype(dtype, copy=True)
