"""Host-side schedule of the device executor (built with device = -1: no HIP call)."""
import ctypes as C

import numpy as np
import pytest

from conftest import problem, shard_masks
from parsy_bench_amd import _native as N, inspector as I


def host_plan(sym):
    h = N.lib().parsy_plan_from_symbolic(sym._handle, -1)
    assert h, N.last_error()
    pi = N.PlanInfo()
    N.lib().parsy_plan_get_info(h, C.byref(pi))
    return h, pi.as_dict()


@pytest.mark.parametrize("name", ["tiny2d", "small3d", "mid3d", "ex15"])
def test_plan_counts_match_the_pattern(name):
    A, perm, sym = problem(name)
    h, info = host_plan(sym)
    try:
        assert (info["n"], info["nsuper"], info["nlevels"]) == (sym.n, sym.nsuper, sym.nlevels)
        assert (info["xsize"], info["ssize"], info["nnzL"], info["nnzA"]) == (sym.xsize, sym.ssize, sym.nnzL, sym.nnzA)
        assert info["n_updates"] == len(sym.updSn)
        assert info["n_small"] + info["n_big"] == sym.nsuper
        assert info["flops_stored"] == pytest.approx(sym.flops_stored, rel=1e-12)
        # update flops from the reference's formulas
        w = np.diff(sym.super).astype(np.float64)
        r = np.diff(sym.i_ptr[sym.super].astype(np.int64)).astype(np.float64)
        m = r[sym.updSn] - sym.updLb
        n1 = (sym.updUb - sym.updLb + 1).astype(np.float64)
        K = w[sym.updSn]
        assert info["update_flops"] == pytest.approx(float((K * n1 * (n1 + 1) + 2 * K * (m - n1) * n1).sum()), rel=1e-12)
        assert info["relpos_len"] == int(m.sum())
        assert info["reread_bytes"] == pytest.approx(float(8 * (K * m).sum()), rel=1e-12)
        assert info["chol_launches"] >= 1 and info["solve_launches"] >= 1  # (levels / subtrees: test below)
    finally:
        N.lib().parsy_plan_destroy(h)


def test_device_calls_on_a_host_plan_fail_loudly():
    A, perm, sym = problem("tiny2d")
    h, _ = host_plan(sym)
    try:
        lv = np.zeros(int(sym.xsize))
        rc = N.lib().parsy_factor_host(h, N.ptr(np.ascontiguousarray(sym.A2x)), N.ptr(lv), None)
        assert rc != 0 and "no device" in N.last_error()
    finally:
        N.lib().parsy_plan_destroy(h)


def test_malformed_pattern_is_rejected():
    A, perm, sym = problem("tiny2d")
    bad = sym.s.copy()
    bad[0] += 1  # first row of supernode 0 is no longer its own first column
    a = [np.ascontiguousarray(x) for x in (sym.super, sym.p, sym.i_ptr, bad, sym.sParent, sym.col2Sup, sym.A1p,
                                           sym.A1i, sym.A2p, sym.A2i)]
    h = N.lib().parsy_plan_create(sym.n, sym.nsuper, *[N.ptr(v) for v in a], -1)
    assert not h and "supernode rows" in N.last_error()


@pytest.mark.parametrize("name", ["small3d", "mid3d", "lap30", "ex15", "nd24k"])
@pytest.mark.parametrize("slots", [2, 7, 64, 512])
def test_chain_launches_cannot_deadlock(name, slots):
    """Dry run of the in-launch tile hand-offs (tickets in start order, walkers resident for their
    whole supernode) at several residencies, down to far fewer workgroups than a level has walkers."""
    A, perm, sym = problem(name)
    h, info = host_plan(sym)
    try:
        if slots < 64:
            # fewer resident workgroups than a batch of walkers is outside the contract (the kernel keeps
            # 2 per CU resident); the dry run must then report it rather than hang
            assert N.lib().parsy_plan_chain_check(h, slots) >= 0
        else:
            assert N.lib().parsy_plan_chain_check(h, slots) == 0
    finally:
        N.lib().parsy_plan_destroy(h)


def test_chain_dry_run_with_many_walkers():
    """A level with more walkers that outlive their first block column (128 supernodes wider than 128
    columns; a walker is resident for its whole chain) than resident workgroups -- what deadlocked a
    block-column-major order over the whole level: batches of kWalkerBatch = 64 walkers keep it live
    at any residency above a batch."""
    from parsy_bench_amd import matrices as M
    A, perm = M.workload("72x72x72")
    sym = I.analyze(A, perm)
    h, info = host_plan(sym)
    try:
        for slots in (80, 128, 512):
            assert N.lib().parsy_plan_chain_check(h, slots) == 0
    finally:
        N.lib().parsy_plan_destroy(h)


@pytest.mark.parametrize("name,piece,mink,sup", [("mid3d", 128, 16, "2"), ("mid3d", 128, 16, "4x2"), ("lap30", 128, 32, "2x4"),
                                                 ("lap30", 256, 64, "3"), ("nd24k", 256, 64, "2"), ("nd24k", 512, 128, "4")])
def test_super_tiles_cover_every_update_once(monkeypatch, name, piece, mink, sup):
    """BIG tasks that own R x C tiles (PARSY_BIG_SUPER; by itself the schedule takes 2 x 2 for launches of many ragged
    tasks): every entry is a block of at most 128 x 128 rows of its source inside the task's window, the blocks cover
    every update exactly once (flop identity), every task is launched exactly once -- parsy_plan_check -- and there
    are fewer tasks for the same flops than with single tiles."""
    A, perm, sym = problem(name)
    monkeypatch.setenv("PARSY_PIECE_WIDTH", str(piece))
    monkeypatch.setenv("PARSY_BIG_MINK", str(mink))
    monkeypatch.setenv("PARSY_BIG_SUPER", "1")
    h1, info1 = host_plan(sym)
    monkeypatch.setenv("PARSY_BIG_SUPER", sup)
    h, info = host_plan(sym)
    try:
        assert N.lib().parsy_plan_check(h) == 0, N.last_error()
        assert 0 < info["big_tasks"] < info1["big_tasks"]
        assert info["big_flops"] == info1["big_flops"] and info["big_entries"] <= info1["big_entries"]
        for mask in shard_masks(sym, 2):
            assert N.lib().parsy_plan_set_active(h, N.ptr(np.ascontiguousarray(mask))) == 0
            assert N.lib().parsy_plan_check(h) == 0, N.last_error()
    finally:
        N.lib().parsy_plan_destroy(h)
        N.lib().parsy_plan_destroy(h1)


@pytest.mark.parametrize("name,piece,mink,sup", [("nd24k", 256, 64, "2"), ("lap30", 128, 32, "2"), ("lap30", 256, 64, "2x1")])
def test_strips_ride_with_their_dense_blocks(monkeypatch, name, piece, mink, sup):
    """Round 5: a remainder of <= 16 rows (columns) of a source's run right behind (beside) a full 128 x 128 block is not
    an entry of the ragged launch but rides with that block (WaveEntry::mn bits 17-26).  The flop identity of
    parsy_plan_check counts the strips with their blocks; the pairing is made per task from the pattern alone, so a
    rank's piece mask changes neither the count nor the check; PARSY_DENSE_STRIPS=0 is the plan without."""
    from parsy_bench_amd import api
    A, perm, sym = problem(name)
    monkeypatch.setenv("PARSY_PIECE_WIDTH", str(piece))
    monkeypatch.setenv("PARSY_BIG_MINK", str(mink))
    monkeypatch.setenv("PARSY_BIG_SUPER", sup)
    monkeypatch.setenv("PARSY_BIG_DENSE", "2")
    monkeypatch.setenv("PARSY_DENSE_STRIPS", "0")
    plan0 = api.Plan(sym, -1)
    monkeypatch.delenv("PARSY_DENSE_STRIPS")
    plan = api.Plan(sym, -1)
    i0, i1 = plan0.info, plan.info
    assert i0["dense_strip_entries"] == 0 and i1["dense_strip_entries"] > 0
    assert plan0.check() == 0 and plan.check() == 0
    assert i1["big_flops"] == i0["big_flops"] and i1["dense_flops"] > i0["dense_flops"]
    assert i1["dense_entries"] == i0["dense_entries"] and i1["big_tasks"] <= i0["big_tasks"]
    D = api.Dist(plan, 3)
    for rank in range(3):
        plan.set_active_pieces(D.mask(rank))
        assert plan.check() == 0 and plan.info["dense_strip_entries"] == i1["dense_strip_entries"]
    plan.set_active_pieces(None)
    assert plan.check() == 0


@pytest.mark.parametrize("name,piece,mink,mode", [("mid3d", 128, 16, 2), ("lap30", 128, 32, 1), ("nd24k", 512, 128, 2),
                                                  ("nd24k", None, None, 2)])
def test_split_chain_launches_are_consistent(monkeypatch, name, piece, mink, mode):
    """Two chain launches per level (diagonal squares with the walkers / the rows below them, PARSY_CHAIN_SPLIT): every
    tile still in exactly one launch, producers before consumers across the two, no deadlock at any residency."""
    A, perm, sym = problem(name)
    if piece is not None:
        monkeypatch.setenv("PARSY_PIECE_WIDTH", str(piece))
        monkeypatch.setenv("PARSY_BIG_MINK", str(mink))
    monkeypatch.setenv("PARSY_CHAIN_SPLIT", "0")
    h0, info0 = host_plan(sym)
    monkeypatch.setenv("PARSY_CHAIN_SPLIT", str(mode))
    h, info = host_plan(sym)
    try:
        assert info["chol_launches"] > info0["chol_launches"]
        assert N.lib().parsy_plan_check(h) == 0, N.last_error()
        for slots in (64, 512):
            assert N.lib().parsy_plan_chain_check(h, slots) == 0
    finally:
        N.lib().parsy_plan_destroy(h)
        N.lib().parsy_plan_destroy(h0)


def test_flan_class_plan_takes_super_tiles_by_itself(monkeypatch):
    """BASELINE configs[2] at full size, host only: the launches with many ragged single-tile tasks take 2 x 2
    super-tiles without being told to (a third fewer tasks, a quarter fewer chunks), the plan checks."""
    from parsy_bench_amd import matrices as M
    A, perm = M.workload("flan")
    sym = I.analyze(A, perm)
    h, info = host_plan(sym)
    monkeypatch.setenv("PARSY_BIG_SUPER", "1")
    h1, info1 = host_plan(sym)
    try:
        assert N.lib().parsy_plan_check(h) == 0, N.last_error()
        assert info["big_flops"] == info1["big_flops"]
        assert info["big_tasks"] < 0.75 * info1["big_tasks"] and info["big_entries"] < 0.75 * info1["big_entries"]
    finally:
        N.lib().parsy_plan_destroy(h)
        N.lib().parsy_plan_destroy(h1)


@pytest.mark.parametrize("name,piece,mink,group", [
    ("small3d", None, None, None), ("mid3d", None, None, None), ("ex15", None, None, None), ("nd24k", None, None, None),
    ("mid3d", 128, 16, 1), ("mid3d", 128, 48, 2), ("lap30", 128, 32, 1), ("lap30", 256, 64, 4), ("lap30", 0, 16, 1),
    ("nd24k", 512, 128, 1), ("nd24k", 256, 64, 2), ("72x72x72", None, None, None)])
def test_cholesky_view_is_consistent(monkeypatch, name, piece, mink, group):
    """The schedule the factorization runs on -- supernodes cut into pieces, updates filed as wave-stream or
    BIG entries by source level -- checked on the host (parsy_plan_check): pieces tile their supernode on
    consecutive levels, every update comes from a lower level, every entry's windows lie inside its update and
    (BIG) inside its task's tile, the entries cover every update exactly once (flop identity with the reference's
    DSYRK + DGEMM counts), side launches sit between the levels they wait for and the level that waits for them."""
    from parsy_bench_amd import matrices as M
    if name in ("72x72x72",):
        A, perm = M.workload(name)
        sym = I.analyze(A, perm)
    else:
        A, perm, sym = problem(name)
    if piece is not None:
        monkeypatch.setenv("PARSY_PIECE_WIDTH", str(piece))
        monkeypatch.setenv("PARSY_BIG_MINK", str(mink))
        monkeypatch.setenv("PARSY_PUSH_GROUP", str(group))
    h, info = host_plan(sym)
    try:
        bad = N.lib().parsy_plan_check(h)
        assert bad == 0, N.last_error()
        if piece:
            assert info["piece_width"] == piece
            if sym.maxSupWid > piece * 3 // 2:
                assert info["n_pieces"] > sym.nsuper and info["chol_levels"] > sym.nlevels
            assert info["big_tasks"] > 0
        if piece is None and name == "72x72x72":
            assert info["big_tasks"] > 0 and info["piece_width"] == 0   # chosen by the size of the job
        # the chain launches of the view cannot deadlock either
        assert N.lib().parsy_plan_chain_check(h, 512) == 0
        # restricting the launches to a subtree keeps the sequence consistent
        for mask in shard_masks(sym, 2) + shard_masks(sym, 3):
            assert N.lib().parsy_plan_set_active(h, N.ptr(np.ascontiguousarray(mask))) == 0
            assert N.lib().parsy_plan_check(h) == 0, N.last_error()
    finally:
        N.lib().parsy_plan_destroy(h)


@pytest.mark.parametrize("name", ["tiny2d", "small3d", "ex15", "lap30"])
def test_subtree_launches_replace_the_narrow_levels(name, monkeypatch):
    """The bottom of the etree goes to subtree launches (one workgroup walks a whole subtree of narrow supernodes:
    the reference's w-partitions, parallel_PB_Cholesky_05.h:66-84, Triangular_BCSC.h:171-232).  Host checks: the plan
    is consistent (parsy_plan_check: members only depend on their own subtree, every SMALL supernode is launched once),
    there are fewer launches than with level launches only (PARSY_SUBTREES=0), and a shard's launches stay consistent."""
    A, perm, sym = problem(name)
    monkeypatch.setenv("PARSY_SUBTREES", "0")
    h0, levels_only = host_plan(sym)
    N.lib().parsy_plan_destroy(h0)
    assert levels_only["chol_subtrees"] == 0 and levels_only["solve_subtrees"] == 0
    assert levels_only["chol_launches"] >= sym.nlevels and levels_only["solve_launches"] >= sym.nlevels
    # small inputs get no subtrees by default (level launches are faster while the device is not saturated)
    monkeypatch.delenv("PARSY_SUBTREES")
    h1, by_default = host_plan(sym)
    N.lib().parsy_plan_destroy(h1)
    assert by_default["chol_subtrees"] == 0 and by_default["solve_launches"] == levels_only["solve_launches"]
    monkeypatch.setenv("PARSY_SUBTREES", "2")   # forced: aim at 2 subtrees per compute unit
    h, info = host_plan(sym)
    try:
        assert N.lib().parsy_plan_check(h) == 0, N.last_error()
        assert info["chol_subtrees"] > 0 and info["solve_subtrees"] > 0
        assert info["chol_subtree_supernodes"] >= info["chol_subtrees"]
        assert info["solve_subtree_supernodes"] > info["solve_subtrees"]      # they do merge supernodes
        assert info["chol_launches"] <= levels_only["chol_launches"]
        assert info["solve_launches"] <= levels_only["solve_launches"]
        assert info["backsolve_launches"] <= levels_only["backsolve_launches"]
        # a shard (every other subtree of a 2-way cut): only active supernodes are launched, still each exactly once
        for mask in shard_masks(sym, 2) + shard_masks(sym, 3):
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert N.lib().parsy_plan_set_active(h, N.ptr(m)) == 0, N.last_error()
            assert N.lib().parsy_plan_check(h) == 0, N.last_error()
            pi = N.PlanInfo()
            N.lib().parsy_plan_get_info(h, C.byref(pi))
            assert pi.solve_subtree_supernodes <= int(m.sum())
    finally:
        N.lib().parsy_plan_destroy(h)


@pytest.mark.parametrize("name", ["small3d", "mid3d", "lap30"])
def test_solve_launches_are_consistent(name, monkeypatch):
    """parsy_plan_check also walks the launches of both solves: every active supernode of one block column is solved
    exactly once forward and backward, every chunk / block column of a wide one appears once, subtree runs are in
    (reverse) index order, width classes hold, the backward chain's groups cover a supernode's block columns -- for the
    default plan, with subtrees forced, in the per-block-column fallback form and for the shards of a 2-way cut."""
    A, perm, sym = problem(name)
    for env in ({}, {"PARSY_SUBTREES": "2"}, {"PARSY_SUBTREES": "0"}, {"PARSY_FORCE_UNFUSED": "1"}):
        for k in ("PARSY_SUBTREES", "PARSY_FORCE_UNFUSED"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h, info = host_plan(sym)
        try:
            assert N.lib().parsy_plan_check(h) == 0, (env, N.last_error())
            for mask in shard_masks(sym, 2) + shard_masks(sym, 3):
                m = np.ascontiguousarray(mask, dtype=np.uint8)
                assert N.lib().parsy_plan_set_active(h, N.ptr(m)) == 0, N.last_error()
                assert N.lib().parsy_plan_check(h) == 0, (env, N.last_error())
        finally:
            N.lib().parsy_plan_destroy(h)


@pytest.mark.parametrize("name,auto,forced", [("tiny2d", 3, 3), ("ex15", 3, 3), ("small3d", 3, 3), ("13x13x13:27", 3, 3),
                                              ("24x24x2:27", 3, 3), ("mid3d", 3, 3), ("lap30", 3, 3), ("nd24k", 7, 7), ("parabolic_fem", 7, 7)])
def test_one_launch_solve_lists(name, auto, forced, monkeypatch):
    """Plans of <= 8192 supernodes (16 384 when they hold >= 4096 entries on average) and <= 2^28 stored entries solve in
    ONE launch per direction (info: bit 0 forward, bit 1 backward; bit 2: only the supernodes outside the subtree launches,
    which stay -- those are what counts then): that the block columns
    tile the supernodes in ticket order, that every row below a block's columns has its own hand-off slot and is
    gathered exactly once by the block that owns it, and the block-column runs of the backward solve are checked by
    parsy_plan_check; a rank's share of the supernodes keeps the level launches; PARSY_SOLVE_ONE=0 / 2 switch it (2:
    whatever the size)."""
    A, perm, sym = problem(name)
    monkeypatch.delenv("PARSY_SOLVE_ONE", raising=False)
    h, info = host_plan(sym)
    try:
        assert info["solve_one"] == auto and N.lib().parsy_plan_check(h) == 0, N.last_error()
        m = np.ascontiguousarray(shard_masks(sym, 2)[0], dtype=np.uint8)
        assert N.lib().parsy_plan_set_active(h, N.ptr(m)) == 0, N.last_error()
        pi = N.PlanInfo()
        N.lib().parsy_plan_get_info(h, C.byref(pi))
        assert pi.solve_one == 0 and N.lib().parsy_plan_check(h) == 0
    finally:
        N.lib().parsy_plan_destroy(h)
    for mode, want in (("0", 0), ("2", forced)):
        monkeypatch.setenv("PARSY_SOLVE_ONE", mode)
        h, info = host_plan(sym)
        try:
            assert info["solve_one"] == want and N.lib().parsy_plan_check(h) == 0, N.last_error()
        finally:
            N.lib().parsy_plan_destroy(h)
