import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O  # oracle/oracle.py -- test infrastructure, never imported by the package
    O.lib()
    return O


@pytest.fixture(scope="session")
def have_gpu():
    from parsy_bench_amd import api
    return api.device_count() > 0


def problem(name):
    """(A, perm, sym) for a named synthetic workload, cached per session."""
    from parsy_bench_amd import inspector as I, matrices as M
    if name not in _CACHE:
        A, perm = M.workload(name)
        _CACHE[name] = (A, perm, I.analyze(A, perm))
    return _CACHE[name]


_CACHE = {}


def shard_masks(sym, nranks):
    """Supernode masks of an nranks-way distribution (parsy_dist on a host-only plan): one mask per rank -- the
    supernodes whose first piece the rank owns -- for the tests that restrict a plan's launches to a shard."""
    import numpy as np
    from parsy_bench_amd import api
    plan = api.Plan(sym, -1)
    D = api.Dist(plan, nranks)
    pieces = plan.pieces()
    first = np.concatenate([[True], np.diff(pieces["supernode"]) != 0])
    own = D.owner[first]
    return [(own == r).astype(np.uint8) for r in range(nranks)]


@pytest.fixture(scope="module")
def api():
    """The product's Python binding; the -m gpu tier fails loudly without a device."""
    from parsy_bench_amd import api as A
    if A.device_count() < 1:
        pytest.fail("no HIP device visible: the -m gpu tier must run on the GPU box")
    return A
