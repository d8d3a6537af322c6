"""One factorization over the GPUs of a node, one process per GPU (torch.distributed: backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

The distribution itself is the library's (include/parsy_amd.h, parsy_dist_*; csrc/dist.cpp): below a cut whole
etree subtrees go to one rank each -- disjoint subtrees are independent in left-looking Cholesky, a target only
reads its descendants (reference common/Reach.h:122-135; the independence the reference's w-partitions rest on,
cholesky/InspectionLevel_06.h:196-217) --, above it the pieces of the Cholesky view (supernodes, the very wide
separators cut into column ranges) are dealt over the ranks level by level.  A rank applies every update into
the pieces it owns; after a level of the Cholesky view is complete, each of its pieces travels to the ranks
that own a target it updates (fan-out; only the rows those targets read).  This module is the plumbing of that
exchange: the segments of a message are packed into one buffer (the library's copy kernel on device tensors),
sent point to point -- with RCCL every pair of GPUs has its own xGMI link --, and unpacked into the same
positions of the receiver's lValues.  The same distribution is run by ONE process over several devices in
csrc/mg.hip (parsy_mg_*, peer-to-peer copies instead of messages).

Numerics stay in the plans: every target receives the same updates in the same order from the same kernels
whatever the number of ranks, so the distributed factor is bitwise the single-device one.
"""
from __future__ import annotations

import numpy as np


def _copy_segments(dst, src, dst_off, src_off, ln, dev_arrays, stream):
    """dst[dst_off[q] + i] = src[src_off[q] + i], i < ln[q]: the library's copy kernel on device tensors (segment
    arrays uploaded once: dev_arrays), numpy on host tensors."""
    if dst.is_cuda:
        from . import _native as N
        d_dst_off, d_src_off, d_len = dev_arrays
        if N.lib().parsy_copy_segments_device(dst.data_ptr(), src.data_ptr(), d_dst_off.data_ptr(), d_src_off.data_ptr(),
                                              d_len.data_ptr(), len(ln), stream) != 0:
            raise RuntimeError("parsy_copy_segments_device failed: " + N.last_error())
        return
    d, s = dst.numpy(), src.numpy()
    for q in range(len(ln)):
        d[dst_off[q]: dst_off[q] + ln[q]] = s[src_off[q]: src_off[q] + ln[q]]


class PlanEngine:
    """The numeric steps of one rank on its device: a plan restricted to the pieces the rank owns."""

    def __init__(self, plan, values_ptr: int):
        self.plan, self.values_ptr = plan, values_ptr

    def begin(self, L, stream):
        self.plan.factor_begin(self.values_ptr, L.data_ptr(), stream)

    def level(self, lev, L, stream):
        self.plan.factor_level(lev, L.data_ptr(), stream)

    def end(self, L, stream):
        self.plan.factor_end(stream)


class DistributedFactorization:
    """The per-rank driver: level steps of `engine` interleaved with the messages of parsy_dist.

    engine: begin(L, stream) / level(lev, L, stream) / end(L, stream) -- PlanEngine on a GPU; the CPU tests plug
    in the oracle.  dist_mod: torch.distributed (initialised).  stage_on_host: bounce device buffers through
    host copies (rehearsals with gloo on device tensors)."""

    def __init__(self, dist_info, rank: int, dist_mod, device, stage_on_host: bool = False):
        import torch
        self.D, self.rank, self.dist, self.device = dist_info, rank, dist_mod, device
        self.stage_on_host = stage_on_host
        self.levels = []          # per level: list of (src, dst, off, len, packed, total, dev_arrays, pool offset)
        self.send_pool_len = self.recv_pool_len = 0
        on_dev = device is not None and str(device) != "cpu"
        for lev in range(dist_info.nlevels):
            mine, s_used, r_used = [], 0, 0
            for (src, dst, off, ln, pk, total) in dist_info.messages(lev, rank):
                dev_arrays = None
                if on_dev:
                    dev_arrays = tuple(torch.from_numpy(a).to(device) for a in (off, pk, ln))
                if src == rank:
                    mine.append((src, dst, off, ln, pk, total, dev_arrays, s_used))
                    s_used += total
                else:
                    mine.append((src, dst, off, ln, pk, total, dev_arrays, r_used))
                    r_used += total
            self.levels.append(mine)
            self.send_pool_len = max(self.send_pool_len, s_used)
            self.recv_pool_len = max(self.recv_pool_len, r_used)
        self._pools = None
        self.sent_elements = sum(m[5] for lv in self.levels for m in lv if m[0] == rank)

    def _ensure_pools(self, like):
        import torch
        if self._pools is None or self._pools[0].device != like.device:
            self._pools = (torch.empty(max(self.send_pool_len, 1), dtype=torch.float64, device=like.device),
                           torch.empty(max(self.recv_pool_len, 1), dtype=torch.float64, device=like.device))
        return self._pools

    def exchange(self, lev: int, L, stream: int = 0):
        """The messages that follow level `lev`: pack -> point to point -> unpack.  `stream` must be the current
        torch stream (the sends / receives are ordered on it)."""
        import torch
        mine = self.levels[lev]
        if not mine:
            return
        if L.is_cuda:
            assert stream == torch.cuda.current_stream(L.device).cuda_stream, \
                "DistributedFactorization.exchange: pass the CURRENT torch stream (RCCL orders its sends / receives on it)"
        send_pool, recv_pool = self._ensure_pools(L)
        ops, landing = [], []
        for (src, dst, off, ln, pk, total, dev, base) in mine:
            if src == self.rank:
                buf = send_pool[base: base + total]
                # pack: segment q of L -> packed[q]   (dst offsets = packed, src offsets = off)
                _copy_segments(buf, L, pk, off, ln, None if dev is None else (dev[1], dev[0], dev[2]), stream)
                ops.append(self.dist.P2POp(self.dist.isend, buf.cpu() if self.stage_on_host else buf, dst))
            else:
                buf = recv_pool[base: base + total]
                host = torch.empty(total, dtype=torch.float64) if self.stage_on_host else None
                ops.append(self.dist.P2POp(self.dist.irecv, host if self.stage_on_host else buf, src))
                landing.append((buf, host, off, ln, pk, dev))
        for req in self.dist.batch_isend_irecv(ops):
            req.wait()
        for (buf, host, off, ln, pk, dev) in landing:
            if host is not None:
                buf.copy_(host)
            # unpack: packed[q] -> segment q of L
            _copy_segments(L, buf, off, pk, ln, None if dev is None else (dev[0], dev[1], dev[2]), stream)

    def factor(self, engine, L, stream: int = 0):
        engine.begin(L, stream)
        for lev in range(self.D.nlevels):
            engine.level(lev, L, stream)
            self.exchange(lev, L, stream)
        engine.end(L, stream)


def gather_factor(L, pieces, owner, rank: int, dist_mod, root: int = 0, stage_on_host: bool = False):
    """After a distributed factorization every piece is final on its owner: collect the whole factor on `root`
    (for a single-GPU solve, or to compare).  Consecutive pieces of one owner are one contiguous range of lValues
    (pieces are in column order).  Returns the number of elements moved."""
    vb, ve = pieces["value_begin"], pieces["value_end"]
    runs, p, n = [], 0, len(owner)
    while p < n:
        q = p
        while q + 1 < n and owner[q + 1] == owner[p]:
            q += 1
        runs.append((int(owner[p]), int(vb[p]), int(ve[q])))
        p = q + 1
    ops, landing, moved = [], [], 0
    for own, a, b in runs:
        if own == root or b <= a:
            continue
        moved += b - a
        view = L[a:b]
        if rank == root:
            buf = view.cpu() if stage_on_host else view
            if stage_on_host:
                landing.append((view, buf))
            ops.append(dist_mod.P2POp(dist_mod.irecv, buf, own))
        elif rank == own:
            ops.append(dist_mod.P2POp(dist_mod.isend, view.cpu() if stage_on_host else view, root))
    if ops:
        for req in dist_mod.batch_isend_irecv(ops):
            req.wait()
    for view, buf in landing:
        view.copy_(buf)
    return moved


class PlanSolver:
    """The solves of one rank on its device: a plan restricted (supernode mask) to the rank's subtrees, and -- on the
    root rank -- one restricted to the supernodes above the cut."""

    def __init__(self, plan):
        self.plan = plan

    def set_mask(self, mask):
        self.plan.set_active(mask)

    def forward(self, L, X, nrhs, n, stream):
        self.plan.solve_device(L.data_ptr(), X.data_ptr(), nrhs, n, stream)

    def backward(self, L, X, nrhs, n, stream):
        self.plan.backsolve_device(L.data_ptr(), X.data_ptr(), nrhs, n, stream)

    def levels(self, L, X, nrhs, n, stream, level_begin, level_end, first, last, backward):
        """One step of a solve that goes level by level (LeveledShardedSolve): parsy_solve_levels_device."""
        self.plan.solve_levels_device(L.data_ptr(), X.data_ptr(), nrhs, n, stream, level_begin, level_end, first, last, backward)


class ShardedSolve:
    """Forward / backward solves on the distributed factor: every rank solves the subtrees it owns --
    a supernode of a subtree only writes rows of its ancestors (forward) and only reads them (backward) --, the
    supernodes above the cut are solved by the root rank, and what crosses the cut is one vector: the forward solve
    REDUCES the partial x (every rank's updates of the rows above the cut; its own subtree columns solved) onto the
    root rank, the backward solve BROADCASTS the root rank's x of those rows.  The panels above the cut are collected
    on the root rank once per factorization (`gather_root_part`); the subtree panels stay where they were factored.

    sub_solver / root_solver: objects with set_mask(mask) / forward(L, X, nrhs, n, stream) / backward(...) --
    PlanSolver on a GPU; the CPU tests plug in the oracle.  X is n x nrhs column-major (1-D tensor), replicated on
    entry (every rank passes the same right-hand side); the result is complete on the root rank."""

    def __init__(self, sym, pieces, dist_info, rank: int, dist_mod, sub_solver, root_solver=None, root: int = 0,
                 stage_on_host: bool = False, _root_mask: bool = True):
        import torch
        self.n, self.rank, self.root, self.dist = sym.n, rank, root, dist_mod
        self.stage_on_host = stage_on_host
        self.pieces, self.D = pieces, dist_info
        first = np.concatenate([[True], np.diff(pieces["supernode"]) != 0])
        sn_owner, sn_below = dist_info.owner[first], dist_info.in_subtree[first]
        self.sub_mask = ((sn_below == 1) & (sn_owner == rank)).astype(np.uint8)
        self.root_mask = (sn_below == 0).astype(np.uint8)
        w = np.diff(sym.super)
        mine = np.repeat(self.sub_mask.astype(bool), w)
        above = np.repeat(self.root_mask.astype(bool), w)
        self.keep = torch.from_numpy(mine | (above if rank == root else np.zeros_like(above)))   # columns whose b / x this rank carries
        self.sub, self.rootsolver = sub_solver, root_solver
        self.sub.set_mask(self.sub_mask)
        if rank == root and _root_mask:
            if root_solver is None:
                raise ValueError("the root rank needs a solver for the supernodes above the cut")
            root_solver.set_mask(self.root_mask)

    def gather_root_part(self, L):
        """The pieces above the cut that other ranks factored -> the root rank (runs of consecutive pieces)."""
        vb, ve = self.pieces["value_begin"], self.pieces["value_end"]
        owner, below = self.D.owner, self.D.in_subtree
        ops, landing, moved, p, n = [], [], 0, 0, len(owner)
        while p < n:
            q = p
            while q + 1 < n and owner[q + 1] == owner[p] and below[q + 1] == below[p]:
                q += 1
            a, b, own = int(vb[p]), int(ve[q]), int(owner[p])
            if below[p] == 0 and own != self.root and b > a:
                moved += b - a
                view = L[a:b]
                if self.rank == self.root:
                    buf = view.cpu() if self.stage_on_host else view
                    if self.stage_on_host:
                        landing.append((view, buf))
                    ops.append(self.dist.P2POp(self.dist.irecv, buf, own))
                elif self.rank == own:
                    ops.append(self.dist.P2POp(self.dist.isend, view.cpu() if self.stage_on_host else view, self.root))
            p = q + 1
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        for view, buf in landing:
            view.copy_(buf)
        return moved

    def _masked(self, X, nrhs):
        if self.keep.device != X.device:
            self.keep = self.keep.to(X.device)
        return (X.view(nrhs, self.n) * self.keep).view(-1)

    def _collect(self, X, op_reduce=True):
        if self.stage_on_host and X.is_cuda:
            h = X.cpu()
            self.dist.reduce(h, self.root) if op_reduce else self.dist.broadcast(h, self.root)
            X.copy_(h)
        elif op_reduce:
            self.dist.reduce(X, self.root)
        else:
            self.dist.broadcast(X, self.root)

    def forward(self, L, B, nrhs: int = 1, stream: int = 0):
        """L x = b.  Returns X: complete on the root rank."""
        X = self._masked(B, nrhs).contiguous()
        self.sub.forward(L, X, nrhs, self.n, stream)      # own subtrees: solved columns + updates of ancestor rows
        self._collect(X, op_reduce=True)                   # sum over ranks = x after every subtree, on the root
        if self.rank == self.root:
            self.rootsolver.forward(L, X, nrhs, self.n, stream)
        return X

    def backward(self, L, Y, nrhs: int = 1, stream: int = 0):
        """L' x = y.  Returns X: complete on the root rank."""
        X = Y.clone()
        if self.rank == self.root:
            self.rootsolver.backward(L, X, nrhs, self.n, stream)   # above the cut first: it reads nothing below
        self._collect(X, op_reduce=False)                           # everybody gets the rows above the cut
        self.sub.backward(L, X, nrhs, self.n, stream)
        X = self._masked(X, nrhs).contiguous()                      # own subtree columns (+ above the cut on the root)
        self._collect(X, op_reduce=True)
        return X


class LeveledShardedSolve(ShardedSolve):
    """ShardedSolve with the supernodes above the cut solved WHERE THEY WERE FACTORED instead of on the root rank: no
    rank ever holds the whole top of the factor.  A supernode above the cut is solved by the owner of its first piece
    (`gather_solve_parts` brings the other pieces of a split supernode there, once per factorization -- the panels of
    whole supernodes do not move at all), and the ranks go through the etree levels above the cut together (the level
    sets of the reference's leveled solve, Triangular_BCSC.h:115-164), one exchange per level:

    forward   x rows of the level's supernodes: every rank holds the updates ITS supernodes made (subtrees and lower
              levels) -> all-reduce(SUM) of those rows, then the owners solve (parsy_solve_levels_device) and the other
              ranks drop the rows;
    backward  the owners solve the level (top level first), then the level's x rows -- zero on the other ranks -- are
              all-reduced: every rank has the x its own supernodes below read.

    top_solver: set_mask(mask) / levels(L, X, nrhs, n, stream, level_begin, level_end, first, last, backward) --
    PlanSolver on a GPU (a second plan of the rank, restricted to the supernodes it solves above the cut).
    level_of: the etree level of every supernode as the plan's launches go by it (Plan.solve_levels())."""

    def __init__(self, sym, pieces, dist_info, rank: int, dist_mod, sub_solver, top_solver, level_of, root: int = 0,
                 stage_on_host: bool = False):
        import torch
        super().__init__(sym, pieces, dist_info, rank, dist_mod, sub_solver, top_solver, root, stage_on_host, _root_mask=False)
        first = np.concatenate([[True], np.diff(pieces["supernode"]) != 0])
        self.sn_owner = dist_info.owner[first].astype(np.int64)          # above the cut: who solves the supernode
        above = self.root_mask.astype(bool)
        self.top_mask = (above & (self.sn_owner == rank)).astype(np.uint8)
        self.top = top_solver
        self.top.set_mask(self.top_mask)
        w = np.diff(sym.super)
        mine = np.repeat(self.sub_mask.astype(bool), w)
        top_cols = np.repeat(self.top_mask.astype(bool), w)
        self.keep_in = torch.from_numpy(mine | top_cols)     # forward: the right-hand side of a row enters on ONE rank
        level_of = np.asarray(level_of)
        self.top_levels = sorted(set(level_of[above].tolist()))
        self.level_cols, self.level_own = {}, {}
        for lev in self.top_levels:
            sns = np.where(above & (level_of == lev))[0]
            cols = np.concatenate([np.arange(sym.super[t], sym.super[t + 1]) for t in sns])
            own = np.concatenate([np.full(int(w[t]), self.sn_owner[t] == rank) for t in sns])
            self.level_cols[lev] = torch.from_numpy(cols.astype(np.int64))
            self.level_own[lev] = torch.from_numpy(own.astype(np.float64))
        self.exchanged = 0        # entries of x all-reduced by the last solve

    def gather_solve_parts(self, L):
        """The pieces of a split supernode above the cut -> the rank that solves the supernode (the owner of its first
        piece); runs of consecutive pieces with one sender and one receiver are one message.  Returns the entries moved."""
        vb, ve = self.pieces["value_begin"], self.pieces["value_end"]
        owner, below, sn = self.D.owner, self.D.in_subtree, self.pieces["supernode"]
        dest = self.sn_owner[sn]
        ops, landing, moved, p, n = [], [], 0, 0, len(owner)
        while p < n:
            q = p
            while q + 1 < n and owner[q + 1] == owner[p] and below[q + 1] == below[p] and dest[q + 1] == dest[p]:
                q += 1
            a, b, own, to = int(vb[p]), int(ve[q]), int(owner[p]), int(dest[p])
            if below[p] == 0 and own != to and b > a:
                moved += b - a
                view = L[a:b]
                if self.rank == to:
                    buf = view.cpu() if self.stage_on_host else view
                    if self.stage_on_host:
                        landing.append((view, buf))
                    ops.append(self.dist.P2POp(self.dist.irecv, buf, own))
                elif self.rank == own:
                    ops.append(self.dist.P2POp(self.dist.isend, view.cpu() if self.stage_on_host else view, to))
            p = q + 1
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        for view, buf in landing:
            view.copy_(buf)
        return moved

    def gather_root_part(self, L):
        raise RuntimeError("LeveledShardedSolve keeps the panels above the cut on the ranks that solve them: gather_solve_parts")

    def _level_sum(self, Xv, lev, own_only: bool):
        """All-reduce(SUM) of the x rows of the supernodes of level `lev` above the cut (own_only: a rank contributes the
        rows it owns, zeros elsewhere)."""
        if self.level_cols[lev].device != Xv.device:       # (once: the index lists live where X lives)
            self.level_cols[lev] = self.level_cols[lev].to(Xv.device)
            self.level_own[lev] = self.level_own[lev].to(Xv.device)
        idx, own = self.level_cols[lev], self.level_own[lev]
        buf = Xv[:, idx]
        if own_only:
            buf = buf * own
        buf = buf.contiguous()
        if self.stage_on_host and buf.is_cuda:
            h = buf.cpu()
            self.dist.all_reduce(h)
            buf.copy_(h)
        else:
            self.dist.all_reduce(buf)
        self.exchanged += buf.numel()
        return idx, own, buf

    def forward(self, L, B, nrhs: int = 1, stream: int = 0):
        """L x = b.  Returns X: complete on the root rank."""
        if self.keep_in.device != B.device:
            self.keep_in = self.keep_in.to(B.device)
        X = (B.view(nrhs, self.n) * self.keep_in).view(-1).contiguous()
        Xv = X.view(nrhs, self.n)
        self.exchanged = 0
        self.sub.forward(L, X, nrhs, self.n, stream)
        nl = len(self.top_levels)
        for i, lev in enumerate(self.top_levels):
            idx, own, buf = self._level_sum(Xv, lev, own_only=False)
            Xv[:, idx] = buf * own                         # the sum where the supernode is solved, nothing elsewhere
            self.top.levels(L, X, nrhs, self.n, stream, lev, lev + 1, i == 0, i == nl - 1, False)
        self._collect(X, op_reduce=True)
        return X

    def backward(self, L, Y, nrhs: int = 1, stream: int = 0):
        """L' x = y.  Returns X: complete on the root rank."""
        X = Y.clone()
        Xv = X.view(nrhs, self.n)
        self.exchanged = 0
        nl = len(self.top_levels)
        for i, lev in enumerate(reversed(self.top_levels)):
            self.top.levels(L, X, nrhs, self.n, stream, lev, lev + 1, i == 0, i == nl - 1, True)
            idx, own, buf = self._level_sum(Xv, lev, own_only=True)
            Xv[:, idx] = buf
        self.sub.backward(L, X, nrhs, self.n, stream)
        X = self._masked(X, nrhs).contiguous()             # own subtree columns (+ above the cut on the root)
        self._collect(X, op_reduce=True)
        return X
