"""Host side of the multi-device factorization (no GPU): the distribution of the pieces over ranks (parsy_dist), its
fan-out messages, and the level-by-level launch lists of the shards (parsy_plan_set_active_pieces).  The reference
has no counterpart; what it fixes is the independence argument (a target only reads its descendants,
common/Reach.h:122-135) that makes whole subtrees go to one rank."""
import numpy as np
import pytest

from conftest import problem
from parsy_bench_amd import _native as N, api


CASES = [("small3d", 2, {}), ("mid3d", 4, {}), ("ex15", 8, {}), ("lap30", 3, {}),
         ("lap30", 4, {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "32"}),
         ("mid3d", 2, {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "16", "PARSY_SUBTREES": "2"}),
         ("nd24k", 8, {})]


@pytest.mark.parametrize("name,nranks,env", CASES)
def test_distribution_is_consistent(monkeypatch, name, nranks, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    A, perm, sym = problem(name)
    plan = api.Plan(sym, -1)
    pieces = plan.pieces()
    npieces = len(pieces["supernode"])
    if "PARSY_PIECE_WIDTH" in env:
        assert npieces > sym.nsuper, "this case is meant to have split supernodes"
    for block in (1, 2):
        D = api.Dist(plan, nranks, block)
        # the library's own check: every update across ranks finds its source rows in a message that follows the
        # source's level; subtrees walked by one workgroup are not split
        assert D.check(plan) == 0, N.last_error()
        assert len(D.owner) == npieces and D.owner.min() >= 0 and D.owner.max() < nranks
        assert D.info["n_root_pieces"] == int((D.in_subtree == 0).sum())
        # below the cut: every piece has its whole etree subtree on its own rank (parents inside a subtree keep the
        # owner; the subtree roots' parents are above the cut)
        sn, lev = pieces["supernode"], pieces["level"]
        par = np.asarray(sym.sParent)
        first_piece = np.searchsorted(sn, np.arange(sym.nsuper))
        for p in np.where(D.in_subtree == 1)[0]:
            s = sn[p]
            if par[s] >= 0:
                q = first_piece[par[s]]
                assert D.in_subtree[q] == 0 or D.owner[q] == D.owner[p]
        # above the cut is upward closed
        for p in np.where(D.in_subtree == 0)[0]:
            s = sn[p]
            if par[s] >= 0:
                assert D.in_subtree[first_piece[par[s]]] == 0
        # the ranks' shares add up, and no rank carries more than the whole
        assert D.rank_cost.sum() == pytest.approx(D.info["total_cost"], rel=1e-12)
        assert D.level_cost.sum() == pytest.approx(D.info["total_cost"], rel=1e-12)
        # messages: senders own what they send, it travels right after its level, packed offsets are prefix sums
        total = 0
        for level in range(D.nlevels):
            for (src, dst, off, ln, pk, tot) in D.messages(level):
                assert src != dst and len(off) == len(ln) == len(pk) and tot == int(ln.sum())
                assert np.array_equal(pk, np.concatenate([[0], np.cumsum(ln[:-1])]))
                p = np.searchsorted(pieces["value_begin"], off, side="right") - 1
                assert (D.owner[p] == src).all() and (lev[p] == level).all()
                assert ((off >= pieces["value_begin"][p]) & (off + ln <= pieces["value_end"][p])).all()
                total += tot
        assert total == D.info["exchange_elements"]
        # every rank's shard is a consistent launch schedule; together the shards launch every BIG task and every
        # piece exactly once (parsy_plan_check counts them per active set)
        launched = 0
        for r in range(nranks):
            plan.set_active_pieces(D.mask(r))
            assert plan.check() == 0, N.last_error()
            launched += plan.info["chol_launches"]
        plan.set_active_pieces(None)
        assert plan.check() == 0
        assert launched >= plan.info["chol_launches"]


def test_one_rank_owns_everything_and_sends_nothing():
    A, perm, sym = problem("mid3d")
    plan = api.Plan(sym, -1)
    D = api.Dist(plan, 1)
    assert D.check(plan) == 0 and (D.owner == 0).all() and D.info["n_messages"] == 0


def test_flan_class_distribution_balances_the_ranks():
    """BASELINE configs[4]: the Flan-class pattern over 2 / 4 / 8 ranks.  Host only (the schedule of the 1.56 M-column
    pattern takes a second): no rank carries more than 1.08 x its share, and the top separators -- more than half of
    the flops -- are spread over all ranks."""
    from parsy_bench_amd import inspector as I, matrices as M
    A, perm = M.workload("flan")
    sym = I.analyze(A, perm)
    plan = api.Plan(sym, -1)
    assert plan.info["n_pieces"] > sym.nsuper
    for nranks in (2, 4, 8):
        D = api.Dist(plan, nranks)
        assert D.check(plan) == 0, N.last_error()
        share = D.rank_cost / D.rank_cost.sum()
        assert share.max() <= 1.08 / nranks
        above = D.in_subtree == 0
        assert D.info["root_cost"] / D.info["total_cost"] > 0.3
        assert len(set(D.owner[above].tolist())) == nranks
