"""The host inspector against vectors produced by the reference's own inspector functions
(tests/golden/inspector_*.npz, see make_golden.py) -- bit-exact integer / index work --
and, where oracle/_ref was built (this container), against the reference live."""
from pathlib import Path

import numpy as np
import pytest

from parsy_bench_amd import inspector as I
from parsy_bench_amd.matrices import LowerCSC

GOLD = Path(__file__).resolve().parent / "golden"
CASES = sorted(p.stem.replace("inspector_", "") for p in GOLD.glob("inspector_*.npz"))

# name in the fixture -> attribute of parsy_bench_amd.inspector.Symbolic
FIELDS = {"Perm": "Perm", "Parent": "Parent", "ColCount": "ColCount", "super": "super",
          "col2Sup": "col2Sup", "sParent": "sParent", "s": "s", "A1p": "A1p", "A1i": "A1i",
          "A2p": "A2p", "A2i": "A2i", "p": "p", "i_ptr": "i_ptr", "A2x": "A2x",
          "wfLevelPtr": "levelPtr", "wfLevelSet": "levelSet", "prunePtr": "updPtr", "pruneSet": "updSn"}


@pytest.mark.parametrize("case", CASES)
def test_matches_reference_inspector(case):
    g = np.load(GOLD / f"inspector_{case}.npz")
    A = LowerCSC(len(g["Ap"]) - 1, g["Ap"], g["Ai"], g["Ax"])
    nrelax = (4, 16, 0) if case.endswith("relax0") else (4, 16, 48)
    sym = I.analyze(A, g["perm"], nrelax=nrelax)
    n, nsuper, ssize, xsize, maxw, maxc = (int(v) for v in g["scalars"][:6])
    assert (sym.n, sym.nsuper, sym.ssize, sym.xsize, sym.maxSupWid, sym.maxCol) == (n, nsuper, ssize, xsize, maxw, maxc)
    for gname, attr in FIELDS.items():
        ours = np.asarray(getattr(sym, attr))
        ref = g[gname]
        assert ours.shape == ref.shape, gname
        assert np.array_equal(ours.astype(ref.dtype), ref), f"{case}: {gname} differs from the reference"


@pytest.mark.parametrize("case", CASES)
def test_symbolic_invariants(case):
    g = np.load(GOLD / f"inspector_{case}.npz")
    A = LowerCSC(len(g["Ap"]) - 1, g["Ap"], g["Ai"], g["Ax"])
    sym = I.analyze(A, g["perm"])
    assert sorted(sym.Perm.tolist()) == list(range(sym.n))
    w = np.diff(sym.super)
    r = np.diff(sym.i_ptr[sym.super].astype(np.int64))
    assert (w > 0).all() and (r >= w).all()
    assert int((w * r).sum()) == sym.xsize and int(r.sum()) == sym.ssize
    assert sym.nnzL == int((w * r - w * (w - 1) // 2).sum())
    assert sym.flops_colcount == float((sym.ColCount.astype(np.float64) ** 2).sum())
    for sn in range(sym.nsuper):
        rows = sym.s[int(sym.i_ptr[sym.super[sn]]): int(sym.i_ptr[sym.super[sn]]) + int(r[sn])]
        assert np.array_equal(rows[: w[sn]], np.arange(sym.super[sn], sym.super[sn + 1]))
        assert (np.diff(rows) > 0).all()
        if sym.sParent[sn] >= 0:
            assert sym.sParent[sn] > sn                      # postordered
            assert sym.col2Sup[rows[w[sn]]] == sym.sParent[sn]  # parent = supernode of first off-diagonal row
    # every level-set member appears once, children strictly below parents
    lev = np.empty(sym.nsuper, int)
    for l in range(sym.nlevels):
        lev[sym.levelSet[sym.levelPtr[l]: sym.levelPtr[l + 1]]] = l
    assert sorted(sym.levelSet.tolist()) == list(range(sym.nsuper))
    for sn in range(sym.nsuper):
        if sym.sParent[sn] >= 0:
            assert lev[sn] < lev[sym.sParent[sn]]


def test_live_reference_on_a_larger_problem(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built here (no /root/reference): the golden vectors stand in")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", GOLD / "make_golden.py")
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from parsy_bench_amd import matrices as M
    A, perm = M.workload("mid3d")
    g = mg.ref_analyze(A, perm)
    sym = I.analyze(A, perm)
    for gname, attr in FIELDS.items():
        assert np.array_equal(np.asarray(getattr(sym, attr)).astype(g[gname].dtype), g[gname]), gname
