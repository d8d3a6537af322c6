"""The N > 1 path on CPU: gloo ranks run the library's distribution (parsy_dist: subtree cut + the pieces above it
dealt over the ranks) level by level, with the fan-out messages of multigpu.DistributedFactorization over
torch.distributed in between.  The numeric work of a level is done by the CPU oracle here (no GPU in this tier);
ownership, messages, packing, the point-to-point exchange and the final gather are the product's code and are
what is under test.  Result: bitwise the single-process factor, with the part above the cut factored by more than
one rank.
"""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _rank_main(rank, world, port, name, out_dir, min_subtrees):
    if min_subtrees:
        os.environ["PARSY_DIST_MIN_SUBTREES"] = str(min_subtrees)
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import oracle as O
    from parsy_bench_amd import api, inspector as I, matrices as M, multigpu as MG

    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    plan = api.Plan(sym, -1)              # host schedule only: pieces, levels, update lists
    pieces = plan.pieces()
    assert len(pieces["supernode"]) == sym.nsuper, "the oracle factors whole supernodes: no split ones in this test"
    D = api.Dist(plan, world)
    assert D.check(plan) == 0
    a = dict(c=sym.A2p, r=sym.A2i, v=np.ascontiguousarray(sym.A2x), lC=sym.p, lR=sym.s, Li=sym.i_ptr,
             bs=sym.super, aT=sym.sParent, cT=sym.A1p, rT=sym.A1i, c2s=sym.col2Sup)
    a = {k: np.ascontiguousarray(v) for k, v in a.items()}

    class OracleEngine:
        """One level of the Cholesky view = the rank's supernodes of that level, each a w-partition of its own."""

        def begin(self, L, stream):
            L.zero_()

        def level(self, lev, L, stream):
            mine = np.where((pieces["level"] == lev) & (D.owner == rank))[0]
            if len(mine) == 0:
                return
            partition = pieces["supernode"][mine].astype(np.int32)
            parPtr = np.arange(len(partition) + 1, dtype=np.int32)
            levelPtr = np.array([0, len(partition), len(partition)], dtype=np.int32)   # + an empty root phase
            timing = np.zeros(16)
            ok = O.lib().oracle_cholesky_left_par_05(
                sym.n, O.P(a["c"]), O.P(a["r"]), O.P(a["v"]), O.P(a["lC"]), O.P(a["lR"]), O.P(a["Li"]),
                O.P(L.numpy()), O.P(a["bs"]), sym.nsuper, O.P(timing), O.P(a["aT"]), O.P(a["cT"]), O.P(a["rT"]),
                O.P(a["c2s"]), 2, O.P(levelPtr), None, 0, O.P(parPtr), O.P(partition), 1, 1,
                sym.maxSupWid + 1, sym.maxCol + 1, None)
            assert ok

        def end(self, L, stream):
            pass

    L = torch.zeros(int(sym.xsize), dtype=torch.float64)
    DF = MG.DistributedFactorization(D, rank, dist, None)
    DF.factor(OracleEngine(), L)
    # every piece is final on its owner: before the gather, compare the rank's own pieces with the reference
    np.save(Path(out_dir) / f"own_{rank}.npy", L.numpy().copy())
    moved = MG.gather_factor(L, pieces, D.owner, rank, dist)
    if rank == 0:
        np.save(Path(out_dir) / "gathered.npy", L.numpy())
        np.save(Path(out_dir) / "meta.npy", np.array([moved, D.info["n_subtrees"], D.info["n_root_pieces"],
                                                      D.info["exchange_elements"], D.info["n_messages"]]))
    sent = torch.tensor([float(DF.sent_elements)], dtype=torch.float64)
    dist.all_reduce(sent)
    if rank == 0:
        np.save(Path(out_dir) / "sent.npy", sent.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world,min_subtrees", [("small3d", 2, 0), ("mid3d", 2, 4), ("lap30", 4, 0), ("mid3d", 4, 0),
                                                     ("lap30", 2, 8)])
def test_distributed_factorization_matches_single_process(tmp_path, oracle, monkeypatch, name, world, min_subtrees):
    import torch.multiprocessing as mp
    from conftest import problem
    from parsy_bench_amd import api, inspector as I
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(world, port, name, str(tmp_path), min_subtrees), nprocs=world, join=True)
    if min_subtrees:
        monkeypatch.setenv("PARSY_DIST_MIN_SUBTREES", str(min_subtrees))
    A, perm, sym = problem(name)
    ok, ref, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    got = np.load(tmp_path / "gathered.npy")
    moved, nsub, nroot, exch, nmsg = np.load(tmp_path / "meta.npy")
    assert ok and np.array_equal(got, ref)
    assert nsub >= world and nroot >= 1 and moved > 0 and nmsg > 0
    # what the ranks sent is what the distribution announced
    assert float(np.load(tmp_path / "sent.npy")[0]) == float(exch)
    # every rank's own pieces were final (bitwise) before the gather, and the part above the cut was factored by
    # more than one rank
    plan = api.Plan(sym, -1)
    pieces = plan.pieces()
    D = api.Dist(plan, world)
    own = [np.load(tmp_path / f"own_{r}.npy") for r in range(world)]
    for p in range(len(D.owner)):
        a, b = int(pieces["value_begin"][p]), int(pieces["value_end"][p])
        assert np.array_equal(own[D.owner[p]][a:b], ref[a:b])
    cost_share = D.rank_cost / D.rank_cost.sum()
    assert cost_share.max() < (0.75 if world == 2 else 0.55)
    above = np.where(D.in_subtree == 0)[0]
    assert len(above) == int(nroot)
    if nroot >= 2:
        assert len(set(D.owner[above].tolist())) >= 2, "the pieces above the cut all went to one rank"
    if min_subtrees or world >= 4:
        assert nroot >= 2, "this case is meant to have the part above the cut factored by more than one rank"


# ---------------------------------------------------------------------------
# sharded solves on the distributed factor (multigpu.ShardedSolve): subtree solves on their owners, one reduce /
# broadcast of x at the cut, the supernodes above the cut on the root rank
# ---------------------------------------------------------------------------
def _solve_rank_main(rank, world, port, name, out_dir, leveled=False):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import oracle as O
    from parsy_bench_amd import api, inspector as I, matrices as M, multigpu as MG

    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    plan = api.Plan(sym, -1)
    pieces = plan.pieces()
    D = api.Dist(plan, world)
    ok, lv, _ = O.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    assert ok
    # every rank keeps only what the distributed factorization leaves it: its own pieces (the rest is poisoned)
    L = torch.full((int(sym.xsize),), float("nan"), dtype=torch.float64)
    for p in np.where(D.owner == rank)[0]:
        a, b = int(pieces["value_begin"][p]), int(pieces["value_end"][p])
        L[a:b] = torch.from_numpy(lv[a:b])
    w = np.diff(sym.super)
    r = np.diff(sym.i_ptr[sym.super].astype(np.int64))

    class OracleSolver:
        """The checker's solves restricted to a set of supernodes (the reference's leveled forward solve takes any
        level set, Triangular_BCSC.h:115-164; the backward one is written out here for the small test inputs)."""

        def set_mask(self, mask):
            self.mask = np.asarray(mask, dtype=bool)

        def forward(self, Lt, X, nrhs, n, stream):
            lev = np.zeros(sym.nsuper, dtype=np.int64)
            for l in range(sym.nlevels):
                lev[sym.levelSet[sym.levelPtr[l]: sym.levelPtr[l + 1]]] = l
            sel = np.where(self.mask)[0]
            order = sel[np.argsort(lev[sel], kind="stable")].astype(np.int32)
            levelPtr = np.zeros(sym.nlevels + 1, dtype=np.int32)
            np.cumsum(np.bincount(lev[sel], minlength=sym.nlevels), out=levelPtr[1:])
            a = [np.ascontiguousarray(sym.p, dtype=np.uint64), np.ascontiguousarray(sym.s, dtype=np.int32),
                 Lt.numpy(), np.ascontiguousarray(sym.i_ptr, dtype=np.uint64),
                 np.ascontiguousarray(sym.col2Sup, dtype=np.int32), np.ascontiguousarray(sym.super, dtype=np.int32)]
            Xn = X.numpy()
            for q in range(nrhs):
                xq = Xn[q * n: (q + 1) * n]
                O.lib().oracle_set_threads(1)
                rc = O.lib().oracle_leveledBlockedLsolve(n, O.P(a[0]), O.P(a[1]), O.P(a[2]), int(sym.xsize), O.P(a[3]),
                                                         O.P(a[4]), O.P(a[5]), sym.nsuper, O.P(xq), sym.nlevels,
                                                         O.P(levelPtr), O.P(order), 1)
                assert rc == 1

        def backward(self, Lt, X, nrhs, n, stream):
            Ln, Xn = Lt.numpy(), X.numpy()
            for s in np.where(self.mask)[0][::-1]:
                c0, ws, rs = int(sym.super[s]), int(w[s]), int(r[s])
                panel = Ln[int(sym.p[c0]): int(sym.p[c0]) + ws * rs].reshape(ws, rs).T      # rows x columns
                rows = sym.s[int(sym.i_ptr[c0]): int(sym.i_ptr[c0]) + rs]
                for q in range(nrhs):
                    xq = Xn[q * n: (q + 1) * n]
                    t = xq[c0: c0 + ws] - panel[ws:, :].T @ xq[rows[ws:]]
                    xq[c0: c0 + ws] = np.linalg.solve(np.tril(panel[:ws, :]).T, t)

        def levels(self, Lt, X, nrhs, n, stream, level_begin, level_end, first, last, back):
            """One step of the leveled form: the masked supernodes of the etree levels [level_begin, level_end)."""
            keep = self.mask.copy()
            self.mask = keep & (level_of >= level_begin) & (level_of < level_end)
            try:
                (self.backward if back else self.forward)(Lt, X, nrhs, n, stream)
            finally:
                self.mask = keep

    if leveled:
        # above the cut every supernode is solved where it was factored: rank 0 never sees the other ranks' panels
        level_of = plan.solve_levels()
        SS = MG.LeveledShardedSolve(sym, pieces, D, rank, dist, OracleSolver(), OracleSolver(), level_of)
        moved = SS.gather_solve_parts(L)
        held = ~torch.isnan(L)
        first = np.concatenate([[True], np.diff(pieces["supernode"]) != 0])
        for p in np.where(D.in_subtree == 0)[0]:       # a panel above the cut is on exactly one rank: its solver
            a, b = int(pieces["value_begin"][p]), int(pieces["value_end"][p])
            want = D.owner[first][pieces["supernode"][p]] == rank or D.owner[p] == rank
            assert bool(held[a:b].all()) == bool(want) or b == a
    else:
        SS = MG.ShardedSolve(sym, pieces, D, rank, dist, OracleSolver(), OracleSolver() if rank == 0 else None)
        moved = SS.gather_root_part(L)
    rng = np.random.default_rng(17)            # (same seed on every rank: the right-hand side is replicated)
    nrhs = 3
    B = torch.from_numpy(rng.standard_normal(sym.n * nrhs))
    Xf = SS.forward(L, B, nrhs)
    Xb = SS.backward(L, B, nrhs)
    if rank == 0:
        np.save(Path(out_dir) / "B.npy", B.numpy())
        np.save(Path(out_dir) / "Xf.npy", Xf.numpy())
        np.save(Path(out_dir) / "Xb.npy", Xb.numpy())
        np.save(Path(out_dir) / "moved.npy", np.array([moved, int(SS.root_mask.sum()), int(SS.sub_mask.sum())]))
    if leveled:
        np.save(Path(out_dir) / f"top_{rank}.npy", np.array([int(SS.top_mask.sum()), len(SS.top_levels), SS.exchanged]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world,leveled", [("mid3d", 2, False), ("lap30", 4, False), ("mid3d", 2, True), ("lap30", 4, True)])
def test_sharded_solves_match_the_serial_solves(tmp_path, oracle, name, world, leveled):
    import torch.multiprocessing as mp
    from conftest import problem
    from parsy_bench_amd import inspector as I
    port = 31500 + (os.getpid() % 2000) + (2000 if leveled else 0)
    mp.spawn(_solve_rank_main, args=(world, port, name, str(tmp_path), leveled), nprocs=world, join=True)
    A, perm, sym = problem(name)
    ok, lv, _ = oracle.cholesky_05(sym, sym.A2x, I.trivial_hlevel(sym))
    B = np.load(tmp_path / "B.npy").reshape(3, sym.n)
    Xf = np.load(tmp_path / "Xf.npy").reshape(3, sym.n)
    Xb = np.load(tmp_path / "Xb.npy").reshape(3, sym.n)
    moved, nroot, nsub0 = np.load(tmp_path / "moved.npy")
    assert nroot >= 1 and nsub0 >= 1
    if leveled:
        tops = np.array([np.load(tmp_path / f"top_{r}.npy") for r in range(world)])
        assert tops[:, 0].sum() == nroot, "every supernode above the cut is solved by exactly one rank"
        assert tops[0, 1] >= 1 and tops[0, 2] > 0
        if world >= 4:
            assert (tops[:, 0] > 0).sum() >= 2, "the supernodes above the cut were all solved by one rank"
    for q in range(3):
        xo = oracle.blocked_lsolve(sym, lv, B[q], "serial")
        assert np.abs(Xf[q] - xo).max() <= 1e-11 * max(1.0, np.abs(xo).max())
        xb = oracle.blocked_ltsolve(sym, lv, B[q])
        assert np.abs(Xb[q] - xb).max() <= 1e-11 * max(1.0, np.abs(xb).max())
