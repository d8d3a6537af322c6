"""Parity of the HIP path with the CPU oracle (the -m gpu tier).

Everything goes through the C ABI of libparsy_amd.so.  Tolerances: the factor is
compared entry-wise with the oracle's lValues, relative to the largest entry of L
(bound 1e-11; FP64 results differ only by summation order), the residual
max|L L' - P A P'| / max|A| <= 1e-10 as north_star states, and solves by
max|x - x_oracle| <= 1e-10 * max|x|.
"""
import numpy as np
import pytest

from conftest import problem

pytestmark = pytest.mark.gpu

FACTOR_TOL = 1e-11
RESID_TOL = 1e-10
SOLVE_TOL = 1e-10


def _factor_both(api, oracle, name):
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem(name)
    plan = api.Plan(sym, 0)
    lv, sec = plan.factor(sym.A2x)
    assert plan.status() == 0
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert ok
    return A, sym, plan, lv, lo


@pytest.mark.parametrize("name", ["tiny2d", "small3d", "ex15", "mid3d", "lap30"])
def test_factor_matches_oracle(api, oracle, name):
    A, sym, plan, lv, lo = _factor_both(api, oracle, name)
    scale = np.abs(lo).max()
    err = np.abs(lv - lo).max() / scale
    assert err <= FACTOR_TOL, f"{name}: max|L_gpu - L_oracle|/max|L| = {err:.3e}"
    # padding above the diagonal of every diagonal block stays exactly zero
    for sn in range(sym.nsuper):
        c0, c1 = int(sym.super[sn]), int(sym.super[sn + 1])
        r = int(sym.i_ptr[c1] - sym.i_ptr[c0]) if c1 < sym.n else int(sym.ssize - sym.i_ptr[c0])
        base = int(sym.p[c0])
        for c in range(1, min(c1 - c0, 8)):
            assert not lv[base + c * r: base + c * r + c].any()


@pytest.mark.parametrize("name", ["tiny2d", "small3d"])
def test_factor_residual_dense(api, oracle, name):
    from parsy_bench_amd import inspector as I
    A, sym, plan, lv, lo = _factor_both(api, oracle, name)
    Ld = I.bcsc_to_dense(sym, lv)
    Ad = A.to_dense()[np.ix_(sym.Perm, sym.Perm)]
    assert np.abs(Ld @ Ld.T - Ad).max() / np.abs(Ad).max() <= RESID_TOL
    assert np.abs(Ld - np.linalg.cholesky(Ad)).max() <= 1e-12 * np.abs(Ld).max() * sym.n


@pytest.mark.parametrize("name", ["tiny2d", "small3d", "ex15", "mid3d", "lap30"])
@pytest.mark.parametrize("nrhs,one", [(1, "0"), (1, "1"), (2, "0"), (3, "1"), (3, "0"), (5, "1"), (5, "0"), (6, "0"), (8, "1"), (8, "0"),
                                      (16, "1"), (19, "1"), (64, "1"), (70, "1")])
def test_solve_matches_oracle(api, oracle, monkeypatch, name, nrhs, one):
    """From 2 right-hand sides on the wide supernodes' chain, from 6 on the narrow supernodes too take the
    many-right-hand-side kernels (64 per pass over a panel, MFMA products): every column is checked against the
    oracle's one-vector solve.  one = "1": the product (blocks of <= 8 right-hand sides of these plans
    go through the ONE-launch kernels), "0": the level launches for every block."""
    monkeypatch.setenv("PARSY_SOLVE_ONE", one)
    A, sym, plan, lv, lo = _factor_both(api, oracle, name)
    rng = np.random.default_rng(1)
    b1 = oracle.rhs_init_blocked(sym, lo)  # b = L * 1  (common/Util.h:277)
    B = np.stack([b1] + [rng.standard_normal(sym.n) for _ in range(nrhs - 1)], axis=1)
    X, sec = plan.solve(lo, B)
    for q in range(nrhs):
        xo = oracle.blocked_lsolve(sym, lo, B[:, q], "serial")
        assert np.abs(X[:, q] - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())
    assert np.abs(X[:, 0] - 1.0).max() <= 1e-9
    assert oracle.lib().oracle_testTriangular(sym.n, oracle.P(np.ascontiguousarray(X[:, 0]))) == 1


@pytest.mark.parametrize("name", ["ex15", "mid3d", "lap30"])
@pytest.mark.parametrize("nrhs", [6, 16, 19, 64, 70])
def test_solve_on_row_major_x_matches_oracle(api, oracle, monkeypatch, name, nrhs):
    """Forward solves with many right-hand sides work on X with the right-hand sides of a row contiguous (transposed in
    and out; by itself only for factors with >= 200 entries per row and >= 16 right-hand sides): forced here
    (PARSY_XT_MIN), every column against the oracle, and the result must equal the right-hand-side-major path's to
    rounding (the scatter into x is made of atomics either way)."""
    monkeypatch.setenv("PARSY_SOLVE_ONE", "0")   # (the level launches: the ONE-launch kernels keep the caller's layout)
    A, sym, plan, lv, lo = _factor_both(api, oracle, name)
    rng = np.random.default_rng(3)
    B = rng.standard_normal((sym.n, nrhs))
    monkeypatch.setenv("PARSY_XT_MIN", "0")
    X0, _ = plan.solve(lo, B)
    monkeypatch.setenv("PARSY_XT_MIN", "6")
    X, _ = plan.solve(lo, B)
    assert plan.solve_status() == 0
    for q in range(nrhs):
        xo = oracle.blocked_lsolve(sym, lo, B[:, q], "serial")
        assert np.abs(X[:, q] - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())
    assert np.abs(X - X0).max() <= 1e-12 * max(1.0, np.abs(X0).max())


@pytest.mark.parametrize("name,env", [("ex15", {}), ("lap30", {}), ("nd24k", {}), ("mid3d", {"PARSY_FORCE_UNFUSED": "1"}),
                                      ("parabolic_fem", {})])
@pytest.mark.parametrize("nrhs", [1, 3, 8, 19])
def test_solve_in_steps_of_levels_matches_oracle(api, oracle, monkeypatch, name, env, nrhs):
    """parsy_solve_levels_device (round 5: the steps of a solve that is distributed above the cut, Triangular_BCSC.h:115-164's
    level sets one call at a time): the whole solve as one call per etree level, and as three calls of level ranges, forward
    and backward, every column against the oracle.  PARSY_FORCE_UNFUSED: the per-block-column form of the wide supernodes,
    whose x goes from scratch into x in the LAST step."""
    import torch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    A, perm, sym = problem(name)
    plan = api.Plan(sym, 0)
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    nl = int(plan.solve_levels().max()) + 1
    assert nl == sym.nlevels
    dev = torch.device("cuda", 0)
    Ld = torch.from_numpy(lv).to(dev)
    rng = np.random.default_rng(7)
    B = rng.standard_normal((nrhs, sym.n))
    cuts = sorted({0, nl // 3, (2 * nl) // 3, nl})
    for backward in (False, True):
        want = [oracle.blocked_ltsolve(sym, lv, B[q]) if backward else oracle.blocked_lsolve(sym, lv, B[q], "serial") for q in range(nrhs)]
        for ranges in ([(l, l + 1) for l in range(nl)], list(zip(cuts[:-1], cuts[1:]))):
            X = torch.from_numpy(B.reshape(-1).copy()).to(dev)
            order = ranges[::-1] if backward else ranges
            for i, (a, b) in enumerate(order):
                plan.solve_levels_device(Ld.data_ptr(), X.data_ptr(), nrhs, sym.n, 0, a, b, i == 0, i == len(order) - 1, backward)
            torch.cuda.synchronize()
            assert plan.solve_status() == 0
            Xh = X.cpu().numpy().reshape(nrhs, sym.n)
            for q in range(nrhs):
                assert np.abs(Xh[q] - want[q]).max() <= SOLVE_TOL * max(1.0, np.abs(want[q]).max()), \
                    f"{name}: {'backward' if backward else 'forward'} solve in {len(ranges)} steps, column {q}"
    # a refused call says why
    with pytest.raises(RuntimeError, match="level_begin <= level_end"):
        plan.solve_levels_device(Ld.data_ptr(), Ld.data_ptr(), 1, sym.n, 0, 3, 2, True, True)
    with pytest.raises(RuntimeError, match="needs an open solve"):
        plan.solve_levels_device(Ld.data_ptr(), X.data_ptr(), nrhs, sym.n, 0, 0, 1, False, False)


@pytest.mark.parametrize("name", ["ex15", "lap30", "nd24k", "parabolic_fem"])
@pytest.mark.parametrize("nrhs", [6, 7, 8, 16])
def test_product_gate_with_row_major_x_forced(api, oracle, monkeypatch, name, nrhs):
    """The product's own choice between the ONE-launch kernels and the level launches (no PARSY_SOLVE_ONE) together
    with the row-major layout of X forced from 6 right-hand sides on.  (Round 4: an experiment that ran the level
    launches of a partly ONE-launch solve on the caller's buffer while the plan still said "row-major, stride ldq"
    wrote past the end of that buffer -- n x nrhs doubles addressed as n x 16 -- and the process aborted in this very
    call; DESIGN section 7.  The ONE-launch path keeps the caller's layout and says so to every launch it makes.)"""
    from parsy_bench_amd import inspector as I
    monkeypatch.delenv("PARSY_SOLVE_ONE", raising=False)
    monkeypatch.setenv("PARSY_XT_MIN", "6")
    A, perm, sym = problem(name)
    plan = api.Plan(sym, 0)
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    rng = np.random.default_rng(5)
    B = rng.standard_normal((sym.n, nrhs))
    for forward in (True, False):
        X, _ = plan.solve(lv, B) if forward else plan.solve2(lv, B, forward=False)
        assert plan.solve_status() == 0
        for q in range(nrhs):
            xo = oracle.blocked_lsolve(sym, lv, B[:, q], "serial") if forward else oracle.blocked_ltsolve(sym, lv, B[:, q])
            assert np.abs(X[:, q] - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max()), (forward, q)


@pytest.mark.parametrize("nrhs", [3, 19])
def test_solve_flag_protocol_chain_matches_oracle(api, oracle, monkeypatch, nrhs):
    """PARSY_OLD_MRHS_CHAIN=1: the chain launches of rounds 1-2 (flags + staged copies per block column; 8 right-hand
    sides per pass below 16) stay available as a fallback and stay correct."""
    monkeypatch.setenv("PARSY_OLD_MRHS_CHAIN", "1")
    monkeypatch.setenv("PARSY_SOLVE_ONE", "0")
    monkeypatch.setenv("PARSY_MRHS_MIN", "16")
    A, sym, plan, lv, lo = _factor_both(api, oracle, "lap30")
    rng = np.random.default_rng(2)
    B = rng.standard_normal((sym.n, nrhs))
    X, _ = plan.solve(lo, B)
    assert plan.solve_status() == 0
    for q in range(nrhs):
        xo = oracle.blocked_lsolve(sym, lo, B[:, q], "serial")
        assert np.abs(X[:, q] - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())


# ---------------------------------------------------------------------------
# drop-in operators: the reference's own argument lists (host pointers)
# ---------------------------------------------------------------------------
def _dropin_args(sym):
    from parsy_bench_amd import inspector as I
    nl, levelPtr, parPtr, partition = I.trivial_hlevel(sym)
    return nl, levelPtr, parPtr, partition


@pytest.mark.parametrize("name", ["small3d", "mid3d"])
def test_dropin_cholesky_operators(api, oracle, name):
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem(name)
    nl, levelPtr, parPtr, partition = _dropin_args(sym)
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, (nl, levelPtr, parPtr, partition))
    timing = np.zeros(8)
    lv = np.zeros(int(sym.xsize))
    assert api.cholesky_left_par_05(sym.n, sym.A2p, sym.A2i, sym.A2x, sym.p, sym.s, sym.i_ptr, lv, sym.super,
                                    sym.nsuper, timing, sym.sParent, sym.A1p, sym.A1i, sym.col2Sup, nl, levelPtr,
                                    None, 0, parPtr, partition, 1, 4, sym.maxSupWid + 1, sym.maxCol + 1) is True
    assert np.abs(lv - lo).max() <= FACTOR_TOL * np.abs(lo).max()
    assert timing[0] > 0 and timing[2] > 0 and timing[1] == 0
    lw = np.zeros(int(sym.xsize))
    assert api.cholesky_left_par_waveFront(sym.n, sym.A2p, sym.A2i, sym.A2x, sym.p, sym.s, sym.i_ptr, lw,
                                           sym.super, sym.nsuper, timing, sym.sParent, sym.A1p, sym.A1i,
                                           sym.col2Sup, sym.nlevels, sym.levelPtr, sym.levelSet, 1, 4,
                                           sym.maxSupWid + 1, sym.maxCol + 1) is True
    assert np.array_equal(lw, lv)  # same plan, same schedule: bitwise
    # the PRUNE build's signature: update lists (getBlockedPruneSet) instead of etree + upper pattern
    lp = np.zeros(int(sym.xsize))
    assert api.cholesky_left_par_05_prune(sym.n, sym.A2p, sym.A2i, sym.A2x, sym.p, sym.s, sym.i_ptr, lp,
                                          sym.super, sym.nsuper, timing, sym.updPtr, sym.updSn, nl, levelPtr,
                                          None, 0, parPtr, partition, 1, 4, sym.maxSupWid + 1,
                                          sym.maxCol + 1) is True
    assert np.array_equal(lp, lv)  # same update lists, same schedule: bitwise
    # a partition that is not a permutation of the supernodes is refused
    bad = partition.copy()
    bad[0] = bad[1]
    lz = np.zeros(int(sym.xsize))
    assert api.cholesky_left_par_05(sym.n, sym.A2p, sym.A2i, sym.A2x, sym.p, sym.s, sym.i_ptr, lz, sym.super,
                                    sym.nsuper, timing, sym.sParent, sym.A1p, sym.A1i, sym.col2Sup, nl, levelPtr,
                                    None, 0, parPtr, bad, 1, 4, sym.maxSupWid + 1, sym.maxCol + 1) is False
    api.dropin_reset()


@pytest.mark.parametrize("name", ["small3d", "lap30"])
def test_dropin_solve_operators(api, oracle, name):
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem(name)
    nl, levelPtr, parPtr, partition = _dropin_args(sym)
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, (nl, levelPtr, parPtr, partition))
    b = oracle.rhs_init_blocked(sym, lo)
    xo = oracle.blocked_lsolve(sym, lo, b, "serial")
    base = (sym.n, sym.p, sym.s, lo, int(sym.xsize), sym.i_ptr, sym.col2Sup, sym.super, sym.nsuper)
    x1 = b.copy()
    assert api.blockedLsolve(*base, x1) == 1
    x2 = b.copy()
    assert api.leveledBlockedLsolve(*base, x2, sym.nlevels, sym.levelPtr, sym.levelSet, 1) == 1
    x3 = b.copy()
    assert api.H2LeveledBlockedLsolve(*base, x3, nl, levelPtr, None, 0, parPtr, partition, 1) == 1
    x4 = b.copy()
    assert api.H2LeveledBlockedLsolve_Peeled(*base, x4, nl, levelPtr, None, 0, parPtr, partition, 1, 4) == 1
    for x in (x1, x2, x3, x4):
        assert np.abs(x - xo).max() <= SOLVE_TOL
        assert oracle.lib().oracle_testTriangular(sym.n, oracle.P(x)) == 1
    # NULL inputs: 0, as the reference (Triangular_BCSC.h:24)
    assert api.blockedLsolve(sym.n, None, sym.s, lo, 0, sym.i_ptr, sym.col2Sup, sym.super, sym.nsuper, x1) == 0
    api.dropin_reset()


# ---------------------------------------------------------------------------
# failure reporting and edge cases
# ---------------------------------------------------------------------------
def test_not_positive_definite_reports_the_column(api, oracle):
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem("small3d")
    vals = sym.A2x.copy()
    # make one diagonal entry of P A P' negative: find the diagonal of column 700
    col = 700
    d = int(sym.A2p[col])
    assert sym.A2i[d] == col
    vals[d] = -5.0
    plan = api.Plan(sym, 0)
    lv, _ = plan.factor(vals)
    st = plan.status()
    ok, lo, _ = oracle.cholesky_05(sym, vals, I.trivial_hlevel(sym))
    assert not ok
    assert st == col + 1, f"status {st}, expected first failing column {col + 1}"
    # and a clean re-factorization afterwards resets it
    lv2, _ = plan.factor(sym.A2x)
    assert plan.status() == 0


@pytest.mark.parametrize("where", ["first_block", "second_block", "small_supernode"])
def test_not_positive_definite_in_every_kind_of_supernode(api, oracle, where):
    """A failing pivot inside a wide supernode (the walker's POTRF, first and later block columns) and
    inside an LDS-resident one: the first failing column is reported, nothing hangs, and the plan is
    usable afterwards."""
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem("mid3d")
    w = np.diff(sym.super)
    wide = int(np.argmax(w))  # the root separator: several block columns
    assert w[wide] > 128
    if where == "first_block":
        col = int(sym.super[wide]) + 5
    elif where == "second_block":
        col = int(sym.super[wide]) + 64 + 17
    else:
        col = int(sym.super[0]) + (int(w[0]) - 1)  # last column of the first (leaf) supernode
    vals = sym.A2x.copy()
    d = int(sym.A2p[col])
    assert sym.A2i[d] == col
    vals[d] = -1e3
    plan = api.Plan(sym, 0)
    plan.factor(vals)
    ok, lo, _ = oracle.cholesky_05(sym, vals, I.trivial_hlevel(sym))
    assert not ok
    assert plan.status() == col + 1
    lv2, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert ok and np.abs(lv2 - lo).max() <= FACTOR_TOL * np.abs(lo).max()


def _check_small(api, oracle, A, perm, nrhs=2):
    from parsy_bench_amd import inspector as I
    sym = I.analyze(A, perm)
    plan = api.Plan(sym, 0)
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert ok and np.abs(lv - lo).max() <= FACTOR_TOL * max(1.0, np.abs(lo).max())
    Ld = I.bcsc_to_dense(sym, lv)
    Ad = A.to_dense()[np.ix_(sym.Perm, sym.Perm)]
    assert np.abs(Ld @ Ld.T - Ad).max() <= RESID_TOL * np.abs(Ad).max()
    rng = np.random.default_rng(7)
    # forward and backward solves with one right-hand side (the wave kernels), with `nrhs` and with a block of 17
    # (the many-right-hand-side kernels), against the dense factor
    for q in sorted({1, nrhs, 17}):
        B = rng.standard_normal((sym.n, q))
        X, _ = plan.solve(lv, B if q > 1 else B[:, 0])
        X = X.reshape(sym.n, -1)
        assert plan.solve_status() == 0
        assert np.abs(Ld @ X - B).max() <= 1e-10 * max(1.0, np.abs(X).max()), f"forward, nrhs={q}"
        Xb, _ = plan.solve2(lv, B, forward=False)
        assert plan.solve_status() == 0
        assert np.abs(Ld.T @ Xb - B).max() <= 1e-10 * max(1.0, np.abs(Xb).max()), f"backward, nrhs={q}"
    return sym


def test_edge_one_by_one(api, oracle):
    from parsy_bench_amd.matrices import LowerCSC
    A = LowerCSC(1, np.array([0, 1], np.int32), np.array([0], np.int32), np.array([9.0]))
    sym = _check_small(api, oracle, A, None)
    assert sym.nsuper == 1


def test_edge_diagonal_matrix(api, oracle):
    from parsy_bench_amd.matrices import LowerCSC
    n = 37
    A = LowerCSC(n, np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.linspace(1.0, 5.0, n))
    _check_small(api, oracle, A, None)


def test_edge_dense_matrix_one_wide_supernode(api, oracle):
    """A dense SPD matrix is a single supernode wider than a tile: TILES is empty, the
    PANEL / INNER chain does all the work (incl. a last block column narrower than 64)."""
    from parsy_bench_amd.matrices import random_spd
    A = random_spd(150, density=1.0, seed=4)
    sym = _check_small(api, oracle, A, None, nrhs=5)
    assert sym.nsuper == 1 and sym.maxSupWid == 150


def test_edge_banded_chain_etree(api, oracle):
    """Tridiagonal matrix in natural order: the etree is a chain (no level parallelism)."""
    from parsy_bench_amd.matrices import LowerCSC
    n = 300
    Ap = np.zeros(n + 1, np.int32)
    Ai, Ax = [], []
    for j in range(n):
        Ai.append(j); Ax.append(4.0)
        if j + 1 < n:
            Ai.append(j + 1); Ax.append(-1.0)
        Ap[j + 1] = len(Ai)
    A = LowerCSC(n, Ap, np.array(Ai, np.int32), np.array(Ax))
    _check_small(api, oracle, A, None)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ragged_random_patterns(api, oracle, seed):
    from parsy_bench_amd.matrices import random_spd
    A = random_spd(220, density=0.03, seed=seed)
    perm = np.random.default_rng(seed).permutation(220).astype(np.int32)
    _check_small(api, oracle, A, perm, nrhs=9)


def test_new_values_same_pattern(api, oracle):
    """Refactorisation: same plan, new numeric values (the executor holds no numeric state)."""
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem("mid3d")
    plan = api.Plan(sym, 0)
    rng = np.random.default_rng(3)
    for _ in range(2):
        Ax = A.Ax * rng.uniform(0.9, 1.1)
        Ax[A.Ap[:-1]] += rng.uniform(0.0, 1.0, A.n)  # keep it diagonally dominant
        vals = sym.permute_values(Ax)
        lv, _ = plan.factor(vals)
        ok, lo, _ = oracle.cholesky_05(sym, vals, I.trivial_hlevel(sym))
        assert ok and plan.status() == 0
        assert np.abs(lv - lo).max() <= FACTOR_TOL * np.abs(lo).max()


# ---------------------------------------------------------------------------
# the distributed factorization, N ranks sharing this one device (parsy_mg: what a node does, with
# device-to-device copies in place of xGMI peer copies)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,nranks,block,env", [
    ("mid3d", 2, 1, {}), ("lap30", 4, 1, {}),
    ("lap30", 4, 1, {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "32"}),
    ("lap30", 3, 2, {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "32", "PARSY_SUBTREES": "2"}),
    ("mid3d", 6, 1, {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "16", "PARSY_DIST_MIN_SUBTREES": "16"}),
    ("nd24k", 4, 1, {}),
])
def test_distributed_factorization_on_one_device(api, oracle, monkeypatch, name, nranks, block, env):
    """Subtrees below the cut on one rank each, the pieces above it dealt over the ranks (split supernodes: their
    pieces on different ranks), finished pieces copied to the ranks that read them after every level: the factor
    collected from the owners equals the single-plan factor BIT FOR BIT, and the oracle to rounding."""
    from parsy_bench_amd import inspector as I
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    A, perm, sym = problem(name)
    plan = api.Plan(sym, 0)
    ref, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    mg = api.MultiDevice(sym, [0] * nranks, block)
    try:
        info = mg.dist.info
        assert info["n_subtrees"] >= nranks and info["n_messages"] > 0
        if "PARSY_PIECE_WIDTH" in env:
            assert plan.info["n_pieces"] > sym.nsuper
        mg.set_values(sym.A2x)
        for _ in range(2):          # (twice: epochs, flags and tickets of the shards are reusable)
            st, _ = mg.factor()
            assert st == 0
            got = mg.gather()
            assert np.array_equal(got, ref)
        above = np.where(mg.dist.in_subtree == 0)[0]
        if len(above) >= 2:
            assert len(set(mg.dist.owner[above].tolist())) >= 2
        # the profiled form (ranks take turns, every launch timed) computes the same factor
        st, main, side, copy = mg.profile()
        assert st == 0 and np.array_equal(mg.gather(), ref)
        assert main.shape == (nranks, plan.info["chol_levels"]) and (main.sum(axis=1) > 0).all()
    finally:
        mg.close()
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert ok and np.abs(ref - lo).max() <= FACTOR_TOL * np.abs(lo).max()


@pytest.mark.parametrize("where", ["subtree", "above the cut"])
def test_distributed_factorization_reports_a_bad_pivot(api, oracle, where):
    """A non-positive pivot on any rank ends the distributed factorization with the failing column (LAPACK's info, as the
    reference returns false on it: parallel_PB_Cholesky_05.h:204-209) -- in a subtree owned by one rank or in a piece
    above the cut."""
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem("lap30")
    mg = api.MultiDevice(sym, [0] * 3)
    try:
        pieces = api.Plan(sym, -1).pieces()
        sel = np.where(mg.dist.in_subtree == (1 if where == "subtree" else 0))[0]
        p = int(sel[len(sel) // 2])
        col = int(pieces["col0"][p]) + 1
        vals = sym.A2x.copy()
        assert sym.A2i[sym.A2p[col]] == col          # the diagonal entry comes first in its column
        vals[sym.A2p[col]] = -5.0
        mg.set_values(vals)
        st, _ = mg.factor()
        ok, lo, bad = oracle.cholesky_05(sym, vals, I.trivial_hlevel(sym))
        assert not ok and st == col + 1
        mg.set_values(sym.A2x)                       # the handle is reusable afterwards
        st, _ = mg.factor()
        assert st == 0
    finally:
        mg.close()


def test_distributed_flan_class_factorization_on_one_device(api):
    """BASELINE configs[4] at full size, 8 ranks sharing this device (8 x 19.4 GB of lValues): BIG launches and
    pieces active, the top separator's 40 pieces dealt over all ranks, 45 GB of pieces copied between the ranks'
    buffers.  Checked bit for bit against the single-plan factor on the device (the input is too large for the CPU
    checker; the single-plan factor is what the size-independent tests below check)."""
    import torch
    from parsy_bench_amd import inspector as I, matrices as M
    A, perm = M.workload("flan")
    sym = I.analyze(A, perm)
    dev = torch.device("cuda", 0)
    plan = api.Plan(sym, 0)
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    Lref = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
    plan.factor_device(values.data_ptr(), Lref.data_ptr(), 0)
    torch.cuda.synchronize()
    assert plan.status() == 0
    del plan
    mg = api.MultiDevice(sym, [0] * 8)
    try:
        D = mg.dist
        assert D.rank_cost.max() / D.rank_cost.sum() <= 1.08 / 8
        mg.set_values(sym.A2x)
        st, _ = mg.factor()
        assert st == 0
        # collected from the owners (19.4 GB through the host), compared on the device in slices
        got = mg.gather()
        for a in range(0, len(got), 1 << 27):
            b = min(len(got), a + (1 << 27))
            assert bool(torch.equal(torch.from_numpy(got[a:b]).to(dev), Lref[a:b]))
    finally:
        mg.close()


# ---------------------------------------------------------------------------
# BASELINE.json's full sizes: size-independent properties
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["nd24k", "parabolic_fem"])
def test_full_size_round_trip(api, oracle, name):
    """b := L 1 on the stored structure, then L x = b must give x = 1 (the reference's own
    check, common/Util.h:277-306), for 1 and for many right-hand sides; and the factor agrees
    with the CPU port.  nd24k = configs[1], parabolic_fem = configs[3] stand-ins."""
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem(name)
    plan = api.Plan(sym, 0)
    lv, sec = plan.factor(sym.A2x)
    assert plan.status() == 0
    b = oracle.rhs_init_blocked(sym, lv)
    for nrhs in (1, 16):
        B = np.repeat(b[:, None], nrhs, axis=1)
        X, _ = plan.solve(lv, B)
        assert np.abs(X - 1.0).max() <= 1e-9
    assert oracle.lib().oracle_testTriangular(sym.n, oracle.P(np.ascontiguousarray(X[:, 0]))) == 1
    # linearity of the solve: L (2x + y) = 2 L x + L y
    rng = np.random.default_rng(0)
    y = rng.standard_normal(sym.n)
    Xy, _ = plan.solve(lv, np.stack([y, 2 * b + y], axis=1))
    assert np.abs(Xy[:, 1] - (2.0 + Xy[:, 0])).max() <= 1e-9 * max(1.0, np.abs(Xy).max())
    blas = oracle.bind_system_blas()
    try:
        ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym), threads=8)
    finally:
        oracle.unbind_blas()
    assert ok and np.abs(lv - lo).max() <= 1e-10 * np.abs(lo).max()


def test_first_factorization_on_a_callers_non_blocking_stream(api, oracle):
    """A fresh plan used at once on a non-blocking stream (what torch.cuda.Stream() creates): the
    plan's uploads (pageable hipMemcpy, hipMemset: ordered against the NULL stream only) must have
    landed before parsy_plan_create returns.  The first factor must equal the default-stream one bitwise."""
    torch = pytest.importorskip("torch")
    A, perm, sym = problem("nd24k")
    dev = torch.device("cuda", 0)
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    L0 = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
    L1 = torch.full((int(sym.xsize),), float("nan"), dtype=torch.float64, device=dev)
    plan0 = api.Plan(sym, 0)
    plan0.factor_device(values.data_ptr(), L0.data_ptr(), 0)
    torch.cuda.synchronize()
    assert plan0.status() == 0
    st = torch.cuda.Stream(device=dev)
    for _ in range(3):  # fresh plan each time: the race is between plan creation and its first use
        plan = api.Plan(sym, 0)
        plan.factor_device(values.data_ptr(), L1.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        assert plan.status() == 0
        assert bool(torch.equal(L0, L1))
        L1.fill_(float("nan"))
        plan.close()


def test_flan_class_size_on_device(api):
    """BASELINE configs[2] stand-in (n = 1 560 896, 2.0e9 factor entries: too large for the CPU
    checker in a test): size-independent properties on the device only -- b := L 1 on the stored
    structure, forward solve gives 1; backward after forward solves L L' x = b; status 0."""
    torch = pytest.importorskip("torch")
    from parsy_bench_amd import inspector as I, matrices as M
    A, perm = M.workload("flan")
    sym = I.analyze(A, perm)
    assert sym.nnzL > 2_000_000_000
    dev = torch.device("cuda", 0)
    plan = api.Plan(sym, 0)
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
    plan.factor_device(values.data_ptr(), L.data_ptr(), 0)
    torch.cuda.synchronize()
    assert plan.status() == 0
    rows = torch.from_numpy(sym.s.astype(np.int64)).to(dev)
    w = np.diff(sym.super)
    r = np.diff(sym.i_ptr[sym.super].astype(np.int64))
    b = torch.zeros(sym.n, dtype=torch.float64, device=dev)
    for sn in range(sym.nsuper):  # b[rows of the panel] += row sums of the panel (common/Util.h:277)
        c0 = int(sym.super[sn])
        rs = slice(int(sym.i_ptr[c0]), int(sym.i_ptr[c0]) + int(r[sn]))
        panel = L[int(sym.p[c0]): int(sym.p[c0]) + int(w[sn] * r[sn])].view(int(w[sn]), int(r[sn]))
        b.index_add_(0, rows[rs], panel.sum(dim=0))
    x = b.clone()
    plan.solve_device(L.data_ptr(), x.data_ptr(), 1, sym.n, 0)
    torch.cuda.synchronize()
    assert float((x - 1.0).abs().max()) <= 1e-9
    # L L' z = b  =>  L' z = 1: apply L' to z on the stored structure and compare with 1
    plan.backsolve_device(L.data_ptr(), x.data_ptr(), 1, sym.n, 0)
    torch.cuda.synchronize()
    t = torch.zeros(sym.n, dtype=torch.float64, device=dev)
    for sn in range(sym.nsuper):  # t[cols of the panel] = panel' z[rows]
        c0, c1 = int(sym.super[sn]), int(sym.super[sn + 1])
        rs = slice(int(sym.i_ptr[c0]), int(sym.i_ptr[c0]) + int(r[sn]))
        panel = L[int(sym.p[c0]): int(sym.p[c0]) + int(w[sn] * r[sn])].view(int(w[sn]), int(r[sn]))
        t[c0:c1] = panel @ x[rows[rs]]
    assert float((t - 1.0).abs().max()) <= 1e-8
    assert plan.status() == 0 and plan.solve_status() == 0
    # and the factor itself against the ORIGINAL matrix: x = P' L'^-1 L^-1 P b, residual of A x = b on the host
    # (north_star: 1e-10 relative) -- an oracle-independent check of factorization + both solves at full size
    rng = np.random.default_rng(5)
    bh = rng.standard_normal(sym.n)
    perm_t = torch.from_numpy(sym.Perm.astype(np.int64)).to(dev)
    z = torch.from_numpy(bh).to(dev)[perm_t].contiguous()
    plan.solve_device(L.data_ptr(), z.data_ptr(), 1, sym.n, 0)
    plan.backsolve_device(L.data_ptr(), z.data_ptr(), 1, sym.n, 0)
    torch.cuda.synchronize()
    assert plan.solve_status() == 0
    xh = np.empty(sym.n)
    xh[sym.Perm] = z.cpu().numpy()
    As = A.to_scipy()
    res = As @ xh - bh
    assert np.abs(res).max() <= RESID_TOL * (abs(As).max() * np.abs(xh).max() + np.abs(bh).max())


def test_levels_with_hundreds_of_walkers(api, oracle):
    """A larger 3-D grid: levels with hundreds of wide supernodes, each with a workgroup that stays
    resident for the whole supernode (the walker).  The launches must neither deadlock nor time out
    (status < 0).  (The host dry run in test_schedule_host.py covers residencies below a level's walkers.)"""
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem("56x56x56")
    plan = api.Plan(sym, 0)
    assert plan.info["n_big"] > 512
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    blas = oracle.bind_system_blas()
    try:
        ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym), threads=8)
    finally:
        oracle.unbind_blas()
    assert ok and np.abs(lv - lo).max() <= 1e-10 * np.abs(lo).max()


def test_unfused_fallback_paths(api, oracle, monkeypatch):
    """The solve schedule used when a chain launch would not be resident (per-block-column
    launches + fix-up) gives the same answers."""
    from parsy_bench_amd import inspector as I
    monkeypatch.setenv("PARSY_FORCE_UNFUSED", "1")
    A, perm, sym = problem("lap30")
    plan = api.Plan(sym, 0)
    info = plan.info
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert np.abs(lv - lo).max() <= FACTOR_TOL * np.abs(lo).max()
    b = oracle.rhs_init_blocked(sym, lo)
    X, _ = plan.solve(lo, np.stack([b, 2 * b, b + 1.0], axis=1))
    xo = oracle.blocked_lsolve(sym, lo, b + 1.0, "serial")
    assert np.abs(X[:, 0] - 1.0).max() <= 1e-9 and np.abs(X[:, 2] - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())
    # the backward solve's fallback (one launch per block-column index) against the checker
    y = np.linspace(-1.0, 1.0, sym.n)
    Xb, _ = plan.solve2(lo, y[:, None], forward=False)
    xb = oracle.blocked_ltsolve(sym, lo, y)
    assert np.abs(Xb[:, 0] - xb).max() <= SOLVE_TOL * max(1.0, np.abs(xb).max())
    monkeypatch.delenv("PARSY_FORCE_UNFUSED")
    plan2 = api.Plan(sym, 0)
    assert plan2.info["solve_launches"] < info["solve_launches"]  # the chain schedule has fewer launches
    lv2, _ = plan2.factor(sym.A2x)
    assert np.abs(lv2 - lv).max() <= FACTOR_TOL * np.abs(lo).max()


# ---------------------------------------------------------------------------
# backward solve and the end-to-end solve A x = b
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny2d", "small3d", "ex15", "mid3d", "lap30"])
@pytest.mark.parametrize("nrhs,one", [(1, "0"), (1, "1"), (5, "0"), (5, "1"), (16, "1"), (19, "1"), (64, "1"), (70, "1")])
def test_backward_solve_matches_checker(api, oracle, monkeypatch, name, nrhs, one):
    """One, a few (4 per pass) and many right-hand sides (from 16 on: 64 per pass over L, products on the matrix
    cores -- k_bsolve_block_mrhs): every column against the checker's transposed solve.  one: as in
    test_solve_matches_oracle."""
    monkeypatch.setenv("PARSY_SOLVE_ONE", one)
    A, sym, plan, lv, lo = _factor_both(api, oracle, name)
    rng = np.random.default_rng(9)
    Y = rng.standard_normal((sym.n, nrhs))
    X, _ = plan.solve2(lo, Y, forward=False)
    for q in range(nrhs):
        xo = oracle.blocked_ltsolve(sym, lo, Y[:, q])
        assert np.abs(X[:, q] - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())


@pytest.mark.parametrize("name", ["ex15", "mid3d", "lap30", "nd24k"])
def test_backward_solve_with_the_rows_below_summed_first(api, oracle, monkeypatch, name):
    """One right-hand side: the sums of a wide supernode's block columns over the rows below its own columns go to
    k_bsolve_below (512-row chunks over the whole device, partial sums added up by the chain in a fixed order) where a
    level's chain launch has few workgroups; PARSY_BSOLVE_BELOW=2 takes that path for every wide supernode, 0 never.
    Each against the checker, bitwise reproducible, and the plan check covers the slots."""
    monkeypatch.setenv("PARSY_SOLVE_ONE", "0")   # (the level launches: k_bsolve_below belongs to them)
    A, perm, sym = problem(name)
    rng = np.random.default_rng(12)
    y = rng.standard_normal(sym.n)
    got = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("PARSY_BSOLVE_BELOW", mode)
        plan = api.Plan(sym, 0)
        assert plan.check() == 0
        lo, _ = plan.factor(sym.A2x)
        assert plan.status() == 0
        x1, _ = plan.solve2(lo, y, forward=False)
        x2, _ = plan.solve2(lo, y, forward=False)
        assert plan.solve_status() == 0 and np.array_equal(x1, x2)
        xo = oracle.blocked_ltsolve(sym, lo, y)
        assert np.abs(x1.ravel() - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())
        got[mode] = (x1, plan.info["backsolve_launches"])
    if sym.maxSupWid > 64:
        assert got["2"][1] >= got["0"][1]         # (the extra launches, where a wide supernode has rows below)
    assert np.abs(got["2"][0] - got["0"][0]).max() <= 1e-12 * max(1.0, np.abs(got["0"][0]).max())


@pytest.mark.parametrize("name,nrhs", [("lap30", 64), ("mid3d", 19), ("nd24k", 70)])
def test_backward_solve_many_rhs_without_the_chain(api, oracle, monkeypatch, name, nrhs):
    """The per-block-column form (PARSY_FORCE_UNFUSED: what runs where a chain launch is not wanted) of the many-
    right-hand-side backward kernel: later blocks are read from x, the diagonal solve is the blocked substitution."""
    monkeypatch.setenv("PARSY_FORCE_UNFUSED", "1")
    A, perm, sym = problem(name)
    plan = api.Plan(sym, 0)
    lo, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    rng = np.random.default_rng(10)
    Y = rng.standard_normal((sym.n, nrhs))
    X, _ = plan.solve2(lo, Y, forward=False)
    for q in (0, 1, nrhs // 2, nrhs - 1):
        xo = oracle.blocked_ltsolve(sym, lo, Y[:, q])
        assert np.abs(X[:, q] - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())


@pytest.mark.parametrize("name", ["small3d", "ex15", "lap30", "nd24k"])
def test_end_to_end_solve_residual(api, oracle, name):
    """Factor on the GPU, then x = P' L'^-1 L^-1 P b: the residual of the ORIGINAL system is an
    oracle-independent check of the whole path (north_star: 1e-10 relative residual)."""
    A, perm, sym = problem(name)
    plan = api.Plan(sym, 0)
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    As = A.to_scipy()
    rng = np.random.default_rng(12)
    B = rng.standard_normal((sym.n, 3))
    X, _ = plan.solve_spd(lv, B)
    R = As @ X - B
    assert np.abs(R).max() <= RESID_TOL * (np.abs(As).max() * np.abs(X).max() + np.abs(B).max())
    xs, _ = plan.solve_spd(lv, As @ np.ones(sym.n))
    assert np.abs(xs - 1.0).max() <= 1e-9


# ---------------------------------------------------------------------------
# BIG launches (LDS-staged GEMM updates from wide descendants) and the pieces of split
# supernodes: forced onto small inputs with the diagnostic knobs of the schedule
# (PARSY_PIECE_WIDTH / PARSY_BIG_MINK are read when a plan is built)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,piece,mink", [("small3d", 128, 16), ("mid3d", 128, 16), ("mid3d", 128, 48),
                                             ("lap30", 128, 32), ("lap30", 256, 64), ("ex15", 128, 16),
                                             ("mid3d", 0, 32), ("lap30", 512, 128)])
def test_factor_with_pieces_and_big_launches(api, oracle, monkeypatch, name, piece, mink):
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem(name)
    monkeypatch.setenv("PARSY_PIECE_WIDTH", str(piece))
    monkeypatch.setenv("PARSY_BIG_MINK", str(mink))
    plan = api.Plan(sym, 0)
    info = plan.info
    assert info["piece_width"] == piece and info["big_min_k"] == mink
    if piece and sym.maxSupWid > piece * 3 // 2:
        assert info["n_pieces"] > sym.nsuper and info["chol_levels"] > sym.nlevels
    assert info["big_tasks"] > 0 and info["big_flops"] > 0
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert ok
    err = np.abs(lv - lo).max() / np.abs(lo).max()
    assert err <= FACTOR_TOL, f"{name} piece={piece} mink={mink}: {err:.3e}"
    lv2, _ = plan.factor(sym.A2x)
    assert np.array_equal(lv, lv2)  # fixed summation order: bitwise reproducible
    # the schedule is a rearrangement of the same work: flop counts add up
    monkeypatch.delenv("PARSY_PIECE_WIDTH")
    monkeypatch.delenv("PARSY_BIG_MINK")
    ref = api.Plan(sym, 0).info
    tot = info["big_flops"] + info["tile_update_flops"] + info["inner_flops"]
    tot_ref = ref["big_flops"] + ref["tile_update_flops"] + ref["inner_flops"]
    assert tot <= tot_ref * (1 + 1e-12) + 1 and info["update_flops"] == ref["update_flops"]


# ---------------------------------------------------------------------------
# super-tiles of the BIG launches (a task owns R x C tiles; a source's rows inside the larger window are cut into
# 128 x 128 blocks in the source's own row order): forced onto small inputs (PARSY_BIG_SUPER; by itself the schedule
# takes 2 x 2 only for launches of >= 12 288 tasks).  Which task forms a product changes nothing in its value and the
# sources keep their order: the factor must be BITWISE the single-tile one, and agree with the oracle.
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,piece,mink,sup", [("mid3d", 128, 16, "2"), ("mid3d", 128, 16, "4x2"), ("lap30", 128, 32, "2x4"),
                                                 ("lap30", 256, 64, "3"), ("ex15", 128, 16, "2"), ("mid3d", 0, 32, "8")])
def test_factor_with_super_tiles(api, oracle, monkeypatch, name, piece, mink, sup):
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem(name)
    monkeypatch.setenv("PARSY_PIECE_WIDTH", str(piece))
    monkeypatch.setenv("PARSY_BIG_MINK", str(mink))
    monkeypatch.setenv("PARSY_BIG_SUPER", "1")
    # (which entries are dense blocks depends on the windows: the dense / ragged order of sums is tested below)
    monkeypatch.setenv("PARSY_BIG_DENSE", "0")
    plan1 = api.Plan(sym, 0)
    lv1, _ = plan1.factor(sym.A2x)
    assert plan1.status() == 0
    monkeypatch.setenv("PARSY_BIG_SUPER", sup)
    plan = api.Plan(sym, 0)
    assert plan.info["big_tasks"] > 0 and plan.info["big_flops"] == plan1.info["big_flops"]
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    assert np.array_equal(lv, lv1), f"{name} super {sup}: the factor differs from the single-tile one"
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert ok and np.abs(lv - lo).max() <= FACTOR_TOL * np.abs(lo).max()


# ---------------------------------------------------------------------------
# host buffers in and out (parsy_factor_host, what the drop-in operators call): large factors are downloaded band of
# levels by band of levels BEHIND the kernels of the levels above (a worker thread; PARSY_HOST_PIPELINE=0: kernels, then
# one download; 2: pipelined whatever the size).  Same kernels, same order: the factors must be bitwise equal, and
# every entry of lValues must have been downloaded (the buffer is poisoned first).
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,env", [("lap30", {}), ("mid3d", {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "16"}),
                                      ("nd24k", {}), ("tiny2d", {})])
def test_host_factorization_with_pipelined_download(api, monkeypatch, name, env):
    A, perm, sym = problem(name)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("PARSY_HOST_PIPELINE", "0")
    plan = api.Plan(sym, 0)
    ref, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    monkeypatch.setenv("PARSY_HOST_PIPELINE", "2")
    for _ in range(2):
        out = np.full(int(sym.xsize), np.nan)
        got, _ = plan.factor(sym.A2x, out=out)
        assert plan.status() == 0
        assert not np.isnan(got).any(), "a part of lValues was never downloaded"
        assert np.array_equal(got, ref)


# ---------------------------------------------------------------------------
# DENSE launches (k_chol_dense): the full 128 x 128 blocks among a BIG task's entries go through a
# kernel of their own, launched right before the task's ragged rest.  PARSY_BIG_DENSE=2 takes it wherever a dense
# entry exists, 4 hands EVERY entry to it (it multiplies a ragged window as a whole block and drops what the window
# does not have: what launches with a tiny ragged rest do by themselves), 0 never; the order of sums per entry of L differs between the two (dense entries first), so the
# factors agree to rounding, each is bitwise reproducible, and both agree with the oracle.
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,piece,mink,sup,mode", [("mid3d", 128, 16, "0", 2), ("lap30", 128, 32, "0", 2), ("lap30", 128, 32, "2", 2),
                                                      ("lap30", 256, 64, "2x1", 2), ("nd24k", 0, 64, "0", 2), ("nd24k", 256, 64, "2", 2),
                                                      ("nd24k", 200, 24, "0", 2),
                                                      # mode 4: EVERY entry through k_chol_dense -- ragged windows, K tails
                                                      ("small3d", 128, 16, "0", 4), ("mid3d", 128, 16, "0", 4), ("mid3d", 0, 32, "2", 4),
                                                      ("lap30", 128, 32, "0", 4), ("ex15", 128, 16, "0", 4), ("nd24k", 200, 24, "2", 4)])
def test_factor_with_dense_launches(api, oracle, monkeypatch, name, piece, mink, sup, mode):
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem(name)
    monkeypatch.setenv("PARSY_PIECE_WIDTH", str(piece))
    monkeypatch.setenv("PARSY_BIG_MINK", str(mink))
    if sup != "0":
        monkeypatch.setenv("PARSY_BIG_SUPER", sup)
    monkeypatch.setenv("PARSY_BIG_DENSE", "0")
    plan0 = api.Plan(sym, 0)
    assert plan0.info["dense_entries"] == 0 and plan0.info["dense_tasks"] == 0
    lv0, _ = plan0.factor(sym.A2x)
    assert plan0.status() == 0
    monkeypatch.setenv("PARSY_BIG_DENSE", str(mode))
    plan = api.Plan(sym, 0)
    info = plan.info
    assert info["dense_entries"] > 0 and info["dense_tasks"] > 0 and 0 < info["dense_flops"] <= info["big_flops"]
    if mode == 4:
        # (all but the updates narrower than one 8-wide k chunk: narrow descendants of a split supernode's pieces)
        assert info["dense_entries"] >= 0.9 * info["big_entries"] and info["dense_flops"] >= 0.9 * info["big_flops"]
    assert info["big_flops"] == plan0.info["big_flops"] and plan.check() == 0
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert ok
    scale = np.abs(lo).max()
    assert np.abs(lv - lo).max() <= FACTOR_TOL * scale, f"{name}: dense launches vs oracle {np.abs(lv - lo).max() / scale:.3e}"
    assert np.abs(lv - lv0).max() <= FACTOR_TOL * scale
    for _ in range(2):
        lv2, _ = plan.factor(sym.A2x)
        assert np.array_equal(lv, lv2)   # fixed summation order: bitwise reproducible
    # padding above the diagonal of every diagonal block stays exactly zero
    for sn in range(sym.nsuper):
        c0, c1 = int(sym.super[sn]), int(sym.super[sn + 1])
        r = int(sym.i_ptr[c1] - sym.i_ptr[c0]) if c1 < sym.n else int(sym.ssize - sym.i_ptr[c0])
        base = int(sym.p[c0])
        for c in range(1, min(c1 - c0, 8)):
            assert not lv[base + c * r: base + c * r + c].any()


# ---------------------------------------------------------------------------
# Strips (round 5): the remainder of a source's row run right behind a full 128 x 128 block (<= 16 rows), or of its
# column run beside it, rides with the block through k_chol_dense instead of being an entry of the ragged launch.
# PARSY_DENSE_STRIPS=0 is the form without: the same products in another order of sums -- rounding apart, each bitwise
# reproducible, both against the oracle.
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,piece,mink,sup", [("lap30", 128, 32, "2"), ("lap30", 256, 64, "2x1"), ("nd24k", 256, 64, "2"),
                                                 ("nd24k", 128, 24, "2"), ("mid3d", 128, 16, "2")])
def test_factor_with_dense_strips(api, oracle, monkeypatch, name, piece, mink, sup):
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem(name)
    monkeypatch.setenv("PARSY_PIECE_WIDTH", str(piece))
    monkeypatch.setenv("PARSY_BIG_MINK", str(mink))
    monkeypatch.setenv("PARSY_BIG_SUPER", sup)
    monkeypatch.setenv("PARSY_BIG_DENSE", "2")
    monkeypatch.setenv("PARSY_DENSE_STRIPS", "0")
    plan0 = api.Plan(sym, 0)
    assert plan0.info["dense_strip_entries"] == 0 and plan0.check() == 0
    lv0, _ = plan0.factor(sym.A2x)
    assert plan0.status() == 0
    monkeypatch.delenv("PARSY_DENSE_STRIPS")
    plan = api.Plan(sym, 0)
    info = plan.info
    assert plan.check() == 0 and info["big_flops"] == plan0.info["big_flops"]
    if name != "mid3d":
        assert info["dense_strip_entries"] > 0, "this case is meant to have strips"
        assert info["dense_flops"] > plan0.info["dense_flops"] and info["big_entries"] == plan0.info["big_entries"]
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert ok
    scale = np.abs(lo).max()
    assert np.abs(lv - lo).max() <= FACTOR_TOL * scale, f"{name}: strips vs oracle {np.abs(lv - lo).max() / scale:.3e}"
    assert np.abs(lv - lv0).max() <= FACTOR_TOL * scale
    for _ in range(2):
        lv2, _ = plan.factor(sym.A2x)
        assert np.array_equal(lv, lv2)   # fixed summation order: bitwise reproducible


# ---------------------------------------------------------------------------
# two chain launches per level (the tiles of the diagonal squares with the walkers, then the tiles of the rows below
# them): what the largest jobs take by themselves, forced here (PARSY_CHAIN_SPLIT).  The same tiles, updates and sums:
# the factor must be bitwise the one-launch factor.
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,piece,mink,mode", [("mid3d", 128, 16, 2), ("lap30", 128, 32, 1), ("lap30", 0, 0, 2),
                                                  ("nd24k", 0, 0, 2), ("ex15", 128, 16, 2)])
def test_factor_with_split_chain_launches(api, oracle, monkeypatch, name, piece, mink, mode):
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem(name)
    if piece:
        monkeypatch.setenv("PARSY_PIECE_WIDTH", str(piece))
        monkeypatch.setenv("PARSY_BIG_MINK", str(mink))
    monkeypatch.setenv("PARSY_CHAIN_SPLIT", "0")
    plan0 = api.Plan(sym, 0)
    lv0, _ = plan0.factor(sym.A2x)
    assert plan0.status() == 0
    monkeypatch.setenv("PARSY_CHAIN_SPLIT", str(mode))
    plan = api.Plan(sym, 0)
    assert plan.info["chol_launches"] > plan0.info["chol_launches"]
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    assert np.array_equal(lv, lv0)
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert ok and np.abs(lv - lo).max() <= FACTOR_TOL * np.abs(lo).max()


# ---------------------------------------------------------------------------
# a hand-off wait of the solve's chain launches that times out is REPORTED: own status word, host
# conveniences and drop-in solves fail, the factorization's status is untouched
# ---------------------------------------------------------------------------
def test_solve_timeout_is_reported_not_returned_as_success(api, oracle, monkeypatch):
    from parsy_bench_amd import inspector as I
    monkeypatch.setenv("PARSY_SOLVE_ONE", "0")   # (the level launches; the ONE-launch kernels' timeout: test further down)
    A, perm, sym = problem("lap30")   # has supernodes wider than a tile: chain launches in both solves
    plan = api.Plan(sym, 0)
    assert plan.info["max_width"] > 64
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    b = oracle.rhs_init_blocked(sym, lv)
    x, _ = plan.solve(lv, b)
    assert plan.solve_status() == 0 and np.abs(x - 1.0).max() < 1e-9
    monkeypatch.setenv("PARSY_DEBUG_SOLVE_STALL", "1")   # waiters wait for an epoch nobody publishes
    with pytest.raises(RuntimeError, match="timed out"):
        plan.solve(lv, b)
    assert plan.solve_status() == -1
    assert plan.status() == 0                               # the factorization's status word is its own
    with pytest.raises(RuntimeError, match="timed out"):
        plan.solve2(lv, b, forward=False)
    # the drop-in operator returns 0 (the reference's failure value) instead of a wrong x
    xs = b.copy()
    rc = api.leveledBlockedLsolve(sym.n, sym.p, sym.s, lv, int(sym.xsize), sym.i_ptr, sym.col2Sup, sym.super,
                                  sym.nsuper, xs, sym.nlevels, sym.levelPtr, sym.levelSet, 1)
    assert rc == 0 and np.array_equal(xs, b)
    monkeypatch.delenv("PARSY_DEBUG_SOLVE_STALL")
    x2, _ = plan.solve(lv, b)                               # the next solve starts clean
    assert plan.solve_status() == 0 and np.abs(x2 - 1.0).max() < 1e-9
    rc = api.leveledBlockedLsolve(sym.n, sym.p, sym.s, lv, int(sym.xsize), sym.i_ptr, sym.col2Sup, sym.super,
                                  sym.nsuper, xs, sym.nlevels, sym.levelPtr, sym.levelSet, 1)
    assert rc == 1 and np.abs(xs - 1.0).max() < 1e-9
    api.dropin_reset()


@pytest.mark.parametrize("nrhs", [8, 40])
def test_many_rhs_solve_timeout_is_reported(api, oracle, monkeypatch, nrhs):
    """The same with many right-hand sides: the block-column tasks of k_solve_blocks_mrhs (8: the waves share one group's
    rows and leave together behind a barrier; 40: a wave per group) and the backward chain kernels run into their bounded
    waits, the solve reports -1 instead of hanging or returning a wrong x, and the next solve is clean."""
    monkeypatch.setenv("PARSY_SOLVE_ONE", "0")
    A, perm, sym = problem("lap30")
    plan = api.Plan(sym, 0)
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    rng = np.random.default_rng(11)
    B = rng.standard_normal((sym.n, nrhs))
    X, _ = plan.solve(lv, B)
    assert plan.solve_status() == 0
    monkeypatch.setenv("PARSY_DEBUG_SOLVE_STALL", "1")
    with pytest.raises(RuntimeError, match="timed out"):
        plan.solve(lv, B)
    assert plan.solve_status() == -1
    with pytest.raises(RuntimeError, match="timed out"):
        plan.solve2(lv, B, forward=False)
    monkeypatch.delenv("PARSY_DEBUG_SOLVE_STALL")
    X2, _ = plan.solve(lv, B)
    assert plan.solve_status() == 0 and np.array_equal(X2, X) or np.abs(X2 - X).max() <= 1e-12 * max(1.0, np.abs(X).max())
    Xb, _ = plan.solve2(lv, B, forward=False)
    assert plan.solve_status() == 0
    for q in (0, nrhs - 1):
        xb = oracle.blocked_ltsolve(sym, lv, B[:, q])
        assert np.abs(Xb[:, q] - xb).max() <= SOLVE_TOL * max(1.0, np.abs(xb).max())


# ---------------------------------------------------------------------------
# ONE-launch solves of small plans (k_solve_one, k_bsolve_one): one workgroup per block column taken by ticket
# in level order, every value handed over as the data itself (an armed buffer) instead of level launches.  By itself
# for plans of <= 8192 supernodes (16 384 large ones) and blocks of <= 8 right-hand sides: every input of this file but the
# Flan- and parabolic_fem-class ones.  Here: inputs whose top separators are up to 41 block columns wide; repeated solves
# (two hand-off buffers used in turn, no memset; the buffers grow with the widest block seen), both directions alternating,
# against the oracle / the checker and against the level launches (PARSY_SOLVE_ONE=0).
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny2d", "small3d", "ex15", "13x13x13:27", "24x24x2:27", "mid3d", "lap30", "nd24k"])
@pytest.mark.parametrize("nrhs", [1, 3, 8])
def test_one_launch_solves(api, oracle, monkeypatch, name, nrhs):
    A, perm, sym = problem(name)
    monkeypatch.setenv("PARSY_SOLVE_ONE", "0")
    plan0 = api.Plan(sym, 0)
    assert plan0.info["solve_one"] == 0
    lv, _ = plan0.factor(sym.A2x)
    assert plan0.status() == 0
    monkeypatch.setenv("PARSY_SOLVE_ONE", "2")
    plan = api.Plan(sym, 0)
    # (nd24k-class: since round 5 the solves cut subtrees from 2 048 narrow supernodes on -- bit 2: that launch stays beside)
    assert plan.info["solve_one"] == (7 if name == "nd24k" else 3) and plan.check() == 0
    rng = np.random.default_rng(5)
    b1 = oracle.rhs_init_blocked(sym, lv)
    for rep in range(3):
        B = np.stack([b1] + [rng.standard_normal(sym.n) for _ in range(nrhs - 1)], axis=1)
        X, _ = plan.solve(lv, B)
        assert plan.solve_status() == 0
        X0, _ = plan0.solve(lv, B)
        for q in range(nrhs):
            xo = oracle.blocked_lsolve(sym, lv, B[:, q], "serial")
            assert np.abs(X[:, q] - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())
            assert np.abs(X[:, q] - X0[:, q]).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())
        assert np.abs(X[:, 0] - 1.0).max() <= 1e-9
        Y = rng.standard_normal((sym.n, nrhs))
        for _ in range(1 + rep):   # (an odd and an even number of solves between two forward ones)
            Z, _ = plan.solve2(lv, Y, forward=False)
            assert plan.solve_status() == 0
        for q in range(nrhs):
            zo = oracle.blocked_ltsolve(sym, lv, Y[:, q])
            assert np.abs(Z[:, q] - zo).max() <= SOLVE_TOL * max(1.0, np.abs(zo).max())
    # more right-hand sides than one pass holds: the level launches, then ONE launch again
    B9 = np.stack([b1] * 9, axis=1)
    X9, _ = plan.solve(lv, B9)
    assert np.abs(X9 - 1.0).max() <= 1e-9
    x1, _ = plan.solve(lv, b1)
    assert plan.solve_status() == 0 and np.abs(x1 - 1.0).max() <= 1e-9


# Plans with subtree launches (thousands of tiny supernodes, one wave per subtree): those launches stay -- first in the
# forward solve, last in the backward one -- and the ONE launch holds everything above them.  The parabolic_fem-class input
# takes that form by itself (forward: blocks of <= 4 right-hand sides, backward: one; PARSY_SOLVE_ONE=2: every block of <= 8);
# forced onto others with PARSY_SUBTREES=2.
@pytest.mark.parametrize("name,env", [("parabolic_fem", {}), ("parabolic_fem", {"PARSY_SOLVE_ONE": "2"}),
                                      ("lap30", {"PARSY_SUBTREES": "2", "PARSY_SOLVE_ONE": "2"}),
                                      ("nd24k", {"PARSY_SUBTREES": "2", "PARSY_SOLVE_ONE": "2"}),
                                      ("mid3d", {"PARSY_SUBTREES": "2"}), ("ex15", {"PARSY_SUBTREES": "2"})])
@pytest.mark.parametrize("nrhs", [1, 4, 8])
def test_one_launch_solves_beside_the_subtree_launches(api, oracle, monkeypatch, name, env, nrhs):
    A, perm, sym = problem(name)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    plan = api.Plan(sym, 0)
    info = plan.info
    assert info["solve_one"] == 7 and info["solve_subtrees"] > 0 and plan.check() == 0
    assert 0 < info["solve_one_blocks"]
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    monkeypatch.setenv("PARSY_SOLVE_ONE", "0")
    plan0 = api.Plan(sym, 0)
    rng = np.random.default_rng(11)
    b1 = oracle.rhs_init_blocked(sym, lv)
    for rep in range(2):
        B = np.stack([b1] + [rng.standard_normal(sym.n) for _ in range(nrhs - 1)], axis=1)
        X, _ = plan.solve(lv, B)
        assert plan.solve_status() == 0
        X0, _ = plan0.solve(lv, B)
        for q in range(nrhs):
            xo = oracle.blocked_lsolve(sym, lv, B[:, q], "serial")
            assert np.abs(X[:, q] - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())
            assert np.abs(X[:, q] - X0[:, q]).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())
        assert np.abs(X[:, 0] - 1.0).max() <= 1e-9
        Y = rng.standard_normal((sym.n, nrhs))
        Z, _ = plan.solve2(lv, Y, forward=False)
        assert plan.solve_status() == 0
        for q in range(nrhs):
            zo = oracle.blocked_ltsolve(sym, lv, Y[:, q])
            assert np.abs(Z[:, q] - zo).max() <= SOLVE_TOL * max(1.0, np.abs(zo).max())


def test_one_launch_solve_timeout_is_reported(api, oracle, monkeypatch):
    A, perm, sym = problem("ex15")
    plan = api.Plan(sym, 0)
    assert plan.info["solve_one"] == 3    # (small enough: taken by itself, both directions)
    lv, _ = plan.factor(sym.A2x)
    b = oracle.rhs_init_blocked(sym, lv)
    x, _ = plan.solve(lv, b)
    assert plan.solve_status() == 0 and np.abs(x - 1.0).max() < 1e-9
    monkeypatch.setenv("PARSY_DEBUG_SOLVE_STALL", "1")   # every dependency wait runs into its bound
    with pytest.raises(RuntimeError, match="timed out"):
        plan.solve(lv, b)
    assert plan.solve_status() == -1 and plan.status() == 0
    with pytest.raises(RuntimeError, match="timed out"):
        plan.solve2(lv, b, forward=False)
    monkeypatch.delenv("PARSY_DEBUG_SOLVE_STALL")
    x2, _ = plan.solve(lv, b)                             # the next solve starts clean
    assert plan.solve_status() == 0 and np.abs(x2 - 1.0).max() < 1e-9
    z, _ = plan.solve2(lv, b, forward=False)
    assert plan.solve_status() == 0


@pytest.mark.parametrize("nrhs,one", [(1, "0"), (1, "1"), (5, "0"), (5, "1"), (64, "1")])
def test_right_hand_side_with_the_armed_nan_pattern_does_not_stall(api, oracle, monkeypatch, nrhs, one):
    """The chain launches of the solves hand x over as the data itself (a hand-off buffer armed with a signalling-NaN
    pattern: a value is valid once it differs).  A right-hand side that happens to carry that very pattern must not
    turn into a 2 s timeout: what is published is the result of arithmetic (a quiet NaN), and a value equal to the
    pattern would be published as a quiet NaN -- the solve completes, status 0, the NaN propagates to the rows that
    depend on it and nowhere else."""
    import time
    monkeypatch.setenv("PARSY_SOLVE_ONE", one)             # ("1": blocks of <= 8 go through the ONE-launch kernels' buffers)
    A, perm, sym = problem("lap30")
    plan = api.Plan(sym, 0)
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0 and sym.maxSupWid > 64     # (wide supernodes: the chain launches run)
    armed = np.frombuffer(np.array([0xFFF7A5A5FFF7A5A5], dtype=np.uint64).tobytes(), dtype=np.float64)[0]
    rng = np.random.default_rng(5)
    B = rng.standard_normal((sym.n, nrhs))
    poisoned = int(sym.super[sym.nsuper - 1]) + 3          # a column of the root supernode
    B[poisoned, 0] = armed
    t0 = time.perf_counter()
    X, _ = plan.solve(lv, B if nrhs > 1 else B[:, 0])
    Xb, _ = plan.solve2(lv, B, forward=False)
    assert time.perf_counter() - t0 < 1.5 and plan.solve_status() == 0
    X = X.reshape(sym.n, -1)
    assert np.isnan(X[poisoned, 0]) and np.isnan(Xb[poisoned, 0])
    # forward: nothing before the poisoned entry's 64-column block can see it (inside the block the product with the
    # inverse diagonal block spreads a NaN: 0 x NaN)
    assert not np.isnan(X[:int(sym.super[sym.nsuper - 1]), :]).any()
    if nrhs > 1:
        assert not np.isnan(X[:, 1:]).any() and not np.isnan(Xb[:, 1:]).any()
        xo = oracle.blocked_lsolve(sym, lv, B[:, 1], "serial")
        assert np.abs(X[:, 1] - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())


# ---------------------------------------------------------------------------
# device helpers of the C ABI
# ---------------------------------------------------------------------------
def test_rhs_ones_device_is_the_references_rhsInitBlocked(api, oracle):
    """parsy_rhs_ones_device: b = L 1 on the stored structure (common/Util.h:277-288), against the oracle's
    restatement, on a factor with supernodes of every kind."""
    import torch
    A, perm, sym = problem("lap30")
    plan = api.Plan(sym, 0)
    lv, _ = plan.factor(sym.A2x)
    dev = torch.device("cuda", 0)
    L = torch.from_numpy(lv).to(dev)
    b = torch.full((sym.n,), float("nan"), dtype=torch.float64, device=dev)
    plan.rhs_ones_device(L.data_ptr(), b.data_ptr(), 0)
    torch.cuda.synchronize()
    want = oracle.rhs_init_blocked(sym, lv)
    assert np.abs(b.cpu().numpy() - want).max() <= 1e-12 * np.abs(want).max()


def test_copy_segments_device_packs_and_unpacks(api):
    import torch
    from parsy_bench_amd import _native as N
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    src = torch.from_numpy(rng.standard_normal(100_000)).to(dev)
    ln = rng.integers(0, 70, size=3000).astype(np.int32)          # some empty runs too
    src_off = np.sort(rng.choice(100_000 - 70, size=3000, replace=False)).astype(np.int64)
    dst_off = np.concatenate([[0], np.cumsum(ln[:-1])]).astype(np.int64)
    total = int(ln.sum())
    t = [torch.from_numpy(a).to(dev) for a in (dst_off, src_off, ln)]
    packed = torch.zeros(total, dtype=torch.float64, device=dev)
    assert N.lib().parsy_copy_segments_device(packed.data_ptr(), src.data_ptr(), t[0].data_ptr(), t[1].data_ptr(),
                                              t[2].data_ptr(), len(ln), 0) == 0
    torch.cuda.synchronize()
    h = src.cpu().numpy()
    want = np.concatenate([h[o:o + l] for o, l in zip(src_off, ln)])
    assert np.array_equal(packed.cpu().numpy(), want)
    back = torch.zeros_like(src)
    assert N.lib().parsy_copy_segments_device(back.data_ptr(), packed.data_ptr(), t[1].data_ptr(), t[0].data_ptr(),
                                              t[2].data_ptr(), len(ln), 0) == 0
    torch.cuda.synchronize()
    mask = np.zeros(100_000, bool)
    for o, l in zip(src_off, ln):
        mask[o:o + l] = True
    got = back.cpu().numpy()
    assert np.array_equal(got[mask], h[mask]) and not got[~mask].any()
    assert N.lib().parsy_copy_segments_device(None, None, None, None, None, 0, 0) == 0   # nothing to do
    assert N.lib().parsy_copy_segments_device(None, src.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                              5, 0) != 0                                  # null destination


# ---------------------------------------------------------------------------
# subtree launches (the reference's w-partitions: one workgroup walks a subtree of narrow supernodes in order,
# parallel_PB_Cholesky_05.h:66-84, Triangular_BCSC.h:171-232), forced onto small inputs with PARSY_SUBTREES
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny2d", "small3d", "ex15", "mid3d", "lap30"])
@pytest.mark.parametrize("per_cu", [1, 16])
def test_subtree_launches_match_level_launches(api, oracle, monkeypatch, name, per_cu):
    A, sym, plan0, lv0, lo = _factor_both(api, oracle, name)   # level launches, compared with the oracle
    assert plan0.info["chol_subtrees"] == 0
    monkeypatch.setenv("PARSY_SUBTREES", str(per_cu))
    plan = api.Plan(sym, 0)
    info = plan.info
    assert info["chol_subtrees"] > 0 and info["solve_subtrees"] > 0
    if per_cu == 1:
        assert info["solve_subtree_supernodes"] > info["solve_subtrees"]   # walks of more than one supernode
        assert info["solve_launches"] <= plan0.info["solve_launches"]
        assert info["backsolve_launches"] <= plan0.info["backsolve_launches"]
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    assert np.array_equal(lv, lv0)   # same kernels, same order of sums per entry: bitwise
    rng = np.random.default_rng(21)
    for nrhs in (1, 5, 64, 70):
        B = rng.standard_normal((sym.n, nrhs))
        X, _ = plan.solve(lo, B if nrhs > 1 else B[:, 0])
        Xb, _ = plan.solve2(lo, B, forward=False)
        assert plan.solve_status() == 0
        X = X.reshape(sym.n, -1)
        for q in range(0, nrhs, 7):
            xo = oracle.blocked_lsolve(sym, lo, B[:, q], "serial")
            assert np.abs(X[:, q] - xo).max() <= SOLVE_TOL * max(1.0, np.abs(xo).max())
            xb = oracle.blocked_ltsolve(sym, lo, B[:, q])
            assert np.abs(Xb[:, q] - xb).max() <= SOLVE_TOL * max(1.0, np.abs(xb).max())


# The subtree launch of a solve with many right-hand sides: one wave per (subtree, 16 right-hand sides), what the members
# hand to each other in LDS, the outside rows as one atomic per subtree (k_solve_sub_mrhs / k_bsolve_sub_mrhs; taken from 6
# right-hand sides on, PARSY_SUB_MRHS_MIN=0: the level kernels' subtree form).  Every column against the oracle, both forms,
# both layouts of X.
# Bands of levels above the subtrees take the same form (tiers: one launch per band instead of one or two per level; wider
# supernodes as chains of 16-column blocks), by themselves where a band leaves >= 256 trees: forced here with
# PARSY_SUB_TIER_MIN_TREES (0: the subtree launch only).
@pytest.mark.parametrize("name,per_cu,min_trees", [("tiny2d", 1, 0), ("ex15", 1, 4), ("ex15", 16, 0), ("ex15", 16, 16), ("mid3d", 1, 2),
                                                   ("lap30", 2, 8), ("lap30", 2, 0), ("24x24x2:27", 1, 2), ("nd24k", 16, 16)])
@pytest.mark.parametrize("nrhs", [6, 8, 16, 19, 64, 70])
def test_subtree_launches_with_many_right_hand_sides(api, oracle, monkeypatch, name, per_cu, min_trees, nrhs):
    if name == "nd24k" and nrhs not in (8, 70):
        pytest.skip("the larger input: two block sizes")
    monkeypatch.setenv("PARSY_SOLVE_ONE", "0")
    monkeypatch.setenv("PARSY_SUBTREES", str(per_cu))
    monkeypatch.setenv("PARSY_SUB_TIER_MIN_TREES", str(min_trees))
    A, sym, plan, lv, lo = _factor_both(api, oracle, name)
    assert plan.info["solve_subtrees"] > 0 and plan.info["sub_mrhs_tiers"] >= (2 if min_trees else 1)
    assert (plan.info["sub_mrhs_cover_level"] >= 0) == (min_trees > 0)
    rng = np.random.default_rng(77)
    B = rng.standard_normal((sym.n, nrhs))
    ref_f = np.stack([oracle.blocked_lsolve(sym, lo, B[:, q], "serial") for q in range(nrhs)], axis=1)
    ref_b = np.stack([oracle.blocked_ltsolve(sym, lo, B[:, q]) for q in range(nrhs)], axis=1)
    for sub_min, xt in (("0", "0"), ("6", "0"), ("6", "6")):
        monkeypatch.setenv("PARSY_SUB_MRHS_MIN", sub_min)
        monkeypatch.setenv("PARSY_XT_MIN", xt)
        X, _ = plan.solve(lo, B)
        assert plan.solve_status() == 0
        assert np.abs(X - ref_f).max() <= SOLVE_TOL * max(1.0, np.abs(ref_f).max()), (sub_min, xt)
        Xb, _ = plan.solve2(lo, B, forward=False)
        assert plan.solve_status() == 0
        assert np.abs(Xb - ref_b).max() <= SOLVE_TOL * max(1.0, np.abs(ref_b).max()), (sub_min, xt)


def test_subtree_launches_report_a_bad_pivot(api, oracle, monkeypatch):
    """A non-positive pivot inside a subtree: same failing column as the reference's (first one in column order)."""
    from parsy_bench_amd import inspector as I
    A, perm, sym = problem("ex15")
    monkeypatch.setenv("PARSY_SUBTREES", "1")
    plan = api.Plan(sym, 0)
    assert plan.info["chol_subtree_supernodes"] > 0
    vals = sym.A2x.copy()
    col = int(sym.super[0])                      # first column of a leaf: surely inside a subtree
    assert sym.A2i[sym.A2p[col]] == col          # the diagonal entry comes first in its column
    vals[sym.A2p[col]] = -1.0
    lv, _ = plan.factor(vals)
    ok, lo, bad = oracle.cholesky_05(sym, vals, I.trivial_hlevel(sym))
    assert not ok and plan.status() == col + 1
