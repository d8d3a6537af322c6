"""The one-process-per-GPU path on the GPU box: two (three) ranks share the one device, torch.distributed with
gloo and host-staged messages in place of RCCL (one device cannot host several RCCL ranks) -- everything else is what
`bench.py --gpus N` runs: the library's distribution, plans restricted to the ranks' pieces, level-by-level
factorization with the fan-out messages in between (multigpu.DistributedFactorization), then the sharded solves
(multigpu.ShardedSolve, and multigpu.LeveledShardedSolve: the supernodes above the cut solved where they were
factored).  Checked on rank 0 against a single plan: factor bitwise, solves to rounding."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _rank_main(rank, world, port, name, env, out_dir, distinct=False):
    """distinct = False: the ranks share device 0, gloo + host-staged messages.  distinct = True (needs >= `world`
    devices): rank r on device r, backend nccl (= RCCL), device-to-device messages -- the path bench.py --gpus N runs."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(env)
    import torch
    import torch.distributed as dist
    from parsy_bench_amd import api, inspector as I, matrices as M, multigpu as MG

    di = rank if distinct else 0
    torch.cuda.set_device(di)
    dev = torch.device("cuda", di)
    if distinct:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    stream = torch.cuda.current_stream().cuda_stream
    A, perm = M.workload(name)
    sym = I.analyze(A, perm)
    plan = api.Plan(sym, di)
    pieces = plan.pieces()
    D = api.Dist(plan, world)
    assert D.check(plan) == 0
    plan.set_active_pieces(D.mask(rank))
    values = torch.from_numpy(np.ascontiguousarray(sym.A2x)).to(dev)
    L = torch.empty(int(sym.xsize), dtype=torch.float64, device=dev)
    DF = MG.DistributedFactorization(D, rank, dist, dev, stage_on_host=not distinct)
    engine = MG.PlanEngine(plan, values.data_ptr())
    for _ in range(2):
        DF.factor(engine, L, stream)
    torch.cuda.synchronize()
    assert plan.status() == 0
    own = L.cpu().numpy().copy()
    # sharded solves on the distributed factor
    sub_plan = api.Plan(sym, di)
    root_plan = api.Plan(sym, di) if rank == 0 else None
    SS = MG.ShardedSolve(sym, pieces, D, rank, dist, MG.PlanSolver(sub_plan),
                         MG.PlanSolver(root_plan) if rank == 0 else None, stage_on_host=not distinct)
    SS.gather_root_part(L)
    rng = np.random.default_rng(23)
    nrhs = 3
    Bh = rng.standard_normal(sym.n * nrhs)
    B = torch.from_numpy(Bh).to(dev)
    Xf = SS.forward(L, B, nrhs, stream)
    Xb = SS.backward(L, B, nrhs, stream)
    torch.cuda.synchronize()
    assert sub_plan.solve_status() == 0
    # ... and with the supernodes above the cut solved where they were factored, level by level with one all-reduce of
    # the level's rows in between (parsy_solve_levels_device): no rank holds the other ranks' panels (NaN there)
    L2 = torch.full_like(L, float("nan"))
    for p in np.where(D.owner == rank)[0]:
        a, b = int(pieces["value_begin"][p]), int(pieces["value_end"][p])
        L2[a:b] = torch.from_numpy(own[a:b]).to(dev)
    top_plan = api.Plan(sym, di)
    LS = MG.LeveledShardedSolve(sym, pieces, D, rank, dist, MG.PlanSolver(sub_plan), MG.PlanSolver(top_plan),
                                top_plan.solve_levels(), stage_on_host=not distinct)
    LS.gather_solve_parts(L2)
    lev_out = {}
    for nr in (1, 3, 8):
        Bq = B if nr == 3 else torch.from_numpy(rng.standard_normal(sym.n * nr)).to(dev)
        lev_out[nr] = (Bq.cpu().numpy(), LS.forward(L2, Bq, nr, stream), LS.backward(L2, Bq, nr, stream))
        torch.cuda.synchronize()
        assert sub_plan.solve_status() == 0 and top_plan.solve_status() == 0
    # rank 0 collects the factor and checks everything against a single plan
    np.save(Path(out_dir) / f"own_{rank}.npy", own)
    dist.barrier()
    if rank == 0:
        full = api.Plan(sym, di)
        ref, _ = full.factor(sym.A2x)
        for p in range(len(D.owner)):
            a, b = int(pieces["value_begin"][p]), int(pieces["value_end"][p])
            o = np.load(Path(out_dir) / f"own_{D.owner[p]}.npy", mmap_mode="r")
            assert np.array_equal(o[a:b], ref[a:b]), f"piece {p} differs from the single-plan factor"
        xf, _ = full.solve(ref, Bh.reshape(nrhs, sym.n).T)
        xb, _ = full.solve2(ref, Bh.reshape(nrhs, sym.n).T, forward=False)
        gf = Xf.cpu().numpy().reshape(nrhs, sym.n).T
        gb = Xb.cpu().numpy().reshape(nrhs, sym.n).T
        assert np.abs(gf - xf).max() <= 1e-10 * max(1.0, np.abs(xf).max())
        assert np.abs(gb - xb).max() <= 1e-10 * max(1.0, np.abs(xb).max())
        for nr, (bq, lf, lb) in lev_out.items():
            bq = bq.reshape(nr, sym.n).T
            xf, _ = full.solve(ref, bq)
            xb, _ = full.solve2(ref, bq, forward=False)
            gf, gb = lf.cpu().numpy().reshape(nr, sym.n).T, lb.cpu().numpy().reshape(nr, sym.n).T
            assert np.abs(gf - xf.reshape(sym.n, nr)).max() <= 1e-10 * max(1.0, np.abs(xf).max()), f"leveled forward solve, {nr} right-hand sides"
            assert np.abs(gb - xb.reshape(sym.n, nr)).max() <= 1e-10 * max(1.0, np.abs(xb).max()), f"leveled backward solve, {nr} right-hand sides"
        np.save(Path(out_dir) / "ok.npy", np.array([D.info["n_messages"], D.info["n_root_pieces"]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,env", [
    ("mid3d", 2, {}),
    ("lap30", 3, {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "32"}),
    # 2 x 2 super-tiles and the dense kernel wherever a dense entry exists: strips ride with their blocks (round 5) -- the
    # distributed factor must still be bitwise the single-plan one
    ("nd24k", 3, {"PARSY_PIECE_WIDTH": "256", "PARSY_BIG_MINK": "64", "PARSY_BIG_SUPER": "2", "PARSY_BIG_DENSE": "2"}),
])
def test_ranks_sharing_the_device_factor_and_solve(tmp_path, name, world, env):
    import torch.multiprocessing as mp
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(world, port, name, env, str(tmp_path)), nprocs=world, join=True)
    nmsg, nroot = np.load(tmp_path / "ok.npy")
    assert nmsg > 0 and nroot >= 1


def _device_count():
    import torch
    return torch.cuda.device_count()      # (does not initialise the GPU on this image)


@pytest.mark.gpu
def test_two_distinct_devices_one_process_bitwise():
    """the distinct-device path of parsy_mg (hipDeviceEnablePeerAccess, cross-device
    hipStreamWaitEvent, the peer-reading copy kernel) -- runs only where two devices are visible; on the one-GPU
    box it is SKIPPED, and until it has run once that path stays unverified on hardware (include/parsy_amd.h
    section 5 says so)."""
    if _device_count() < 2:
        pytest.skip("needs two HIP devices: the distinct-device path of parsy_mg is unverified on this box")
    from parsy_bench_amd import api, inspector as I, matrices as M
    for name, env in (("mid3d", {}), ("lap30", {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "32"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            A, perm = M.workload(name)
            sym = I.analyze(A, perm)
            plan = api.Plan(sym, 0)
            ref, _ = plan.factor(sym.A2x)
            assert plan.status() == 0
            mg = api.MultiDevice(sym, [0, 1])
            try:
                assert mg.dist.info["n_messages"] > 0
                mg.set_values(sym.A2x)
                for _ in range(2):
                    st, _ = mg.factor()
                    assert st == 0
                    assert np.array_equal(mg.gather(), ref), "two-device factor differs from the single-plan factor"
            finally:
                mg.close()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


@pytest.mark.gpu
def test_two_distinct_devices_two_processes_rccl(tmp_path):
    """The one-process-per-GPU path with backend nccl (= RCCL) and device-to-device point-to-point messages: rank r
    on device r.  Skipped where fewer than two devices are visible (see the test above)."""
    if _device_count() < 2:
        pytest.skip("needs two HIP devices: RCCL cannot host two ranks on one GPU")
    import torch.multiprocessing as mp
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(2, port, "lap30", {"PARSY_PIECE_WIDTH": "128", "PARSY_BIG_MINK": "32"}, str(tmp_path), True),
             nprocs=2, join=True)
    nmsg, nroot = np.load(tmp_path / "ok.npy")
    assert nmsg > 0 and nroot >= 1
