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


@pytest.fixture(scope="module")
def api():
    from parsy_bench_amd import api as A
    if A.device_count() < 1:
        pytest.fail("no HIP device visible: the -m gpu tier must run on the GPU box")
    return A


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
@pytest.mark.parametrize("nrhs", [1, 3, 8, 19])
def test_solve_matches_oracle(api, oracle, name, nrhs):
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
