"""Fill-reducing ordering for matrices that come without one (csrc/ordering.cpp; the reference calls
METIS, cholesky/LSparsity.h, which this build does not have)."""
import numpy as np
import pytest

from conftest import problem
from parsy_bench_amd import inspector as I, matrices as M


def _is_permutation(p, n):
    return p.shape == (n,) and np.array_equal(np.sort(p), np.arange(n))


@pytest.mark.parametrize("name", ["tiny2d", "ex15", "small3d", "mid3d", "lap30"])
def test_graph_nested_dissection_on_grids(name):
    A, geo, _ = problem(name)
    p = I.order_nd(A)
    assert _is_permutation(p, A.n)
    s_nd, s_nat, s_geo = I.analyze(A, p), I.analyze(A, None), I.analyze(A, geo)
    # it must beat the natural ordering clearly and stay within 2x of the geometric dissection that
    # knows the grid (27-point stencils give it L-infinity shells instead of planes)
    assert s_nd.nnzL < 0.9 * s_nat.nnzL
    assert s_nd.flops_colcount < 2.0 * s_geo.flops_colcount


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_ordering_of_ragged_and_disconnected_patterns(seed):
    rng = np.random.default_rng(seed)
    n = 700
    A = M.random_spd(n, density=0.004 if seed % 2 else 0.02, seed=seed)  # sparse ones fall apart into components
    p = I.order_nd(A, leaf=int(rng.integers(1, 40)))
    assert _is_permutation(p, n)
    sym = I.analyze(A, p)  # the inspector accepts it
    assert sym.n == n and sym.nnzL >= A.n


def test_ordering_edge_cases():
    one = M.random_spd(1, density=1.0, seed=0)
    assert I.order_nd(one).tolist() == [0]
    diag = M.LowerCSC(5, np.arange(6, dtype=np.int32), np.arange(5, dtype=np.int32), np.ones(5))
    assert _is_permutation(I.order_nd(diag), 5)
    # a path graph: separators are single vertices
    n = 300
    Ap = np.zeros(n + 1, dtype=np.int32)
    Ai, Ax = [], []
    for j in range(n):
        Ai.append(j); Ax.append(2.5)
        if j + 1 < n:
            Ai.append(j + 1); Ax.append(-1.0)
        Ap[j + 1] = len(Ai)
    path = M.LowerCSC(n, Ap, np.array(Ai, dtype=np.int32), np.array(Ax))
    p = I.order_nd(path, leaf=8)
    assert _is_permutation(p, n)
    assert I.analyze(path, p).nnzL <= 12 * n  # a chain barely fills in (the count includes the relaxed supernodes' zeros)


@pytest.mark.gpu
def test_factor_and_solve_with_the_graph_ordering(api, oracle):
    """End to end on a matrix the generator's geometric ordering does not know: order_nd -> inspector
    -> GPU factor + solves, residual of the ORIGINAL system."""
    A, _, _ = problem("lap30")
    p = I.order_nd(A)
    sym = I.analyze(A, p)
    plan = api.Plan(sym, 0)
    lv, _ = plan.factor(sym.A2x)
    assert plan.status() == 0
    ok, lo, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert ok and np.abs(lv - lo).max() <= 1e-11 * np.abs(lo).max()
    rng = np.random.default_rng(5)
    b = rng.standard_normal(A.n)
    x, _ = plan.solve_spd(lv, b)
    r = A.to_scipy() @ x - b
    assert np.abs(r).max() <= 1e-10 * max(1.0, np.abs(b).max())
