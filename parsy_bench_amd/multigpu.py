"""Etree-subtree sharding of one factorization over the GPUs of a node.

Disjoint subtrees of the supernodal etree are independent in left-looking
Cholesky (a target only reads descendants: reference common/Reach.h:122-135) --
the same independence the reference exploits for its w-partitions
(cholesky/InspectionLevel_06.h:196-217).  Each rank factors the subtrees it owns;
the supernodes above the cut ("root part") are factored by rank 0 after ONE
exchange step: the owners' panels (contiguous slices of lValues, because
supernodes are numbered in postorder) are gathered onto rank 0.

This module is host logic + torch.distributed plumbing only (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests); numerics stay in the plans.
"""
from __future__ import annotations

import heapq
from dataclasses import dataclass

import numpy as np


@dataclass
class SubtreeCut:
    owner: np.ndarray          # per supernode: owning rank, -1 = root part (rank 0, after the gather)
    subtrees: list             # (first_sn, last_sn, rank, cost) -- supernodes first..last inclusive
    root_nodes: np.ndarray     # supernodes above the cut
    cost: np.ndarray           # per-supernode flop estimate
    rank_cost: np.ndarray      # summed subtree cost per rank

    def mask(self, rank: int) -> np.ndarray:
        return (self.owner == rank).astype(np.uint8)

    def root_mask(self) -> np.ndarray:
        return (self.owner < 0).astype(np.uint8)

    def slices(self, sym):
        """(rank, start, stop) value ranges of lValues for every subtree."""
        out = []
        for first, last, rank, _ in self.subtrees:
            start = int(sym.p[sym.super[first]])
            stop = int(sym.p[sym.super[last + 1]])
            out.append((rank, start, stop))
        return out


def supernode_costs(sym) -> np.ndarray:
    """Executed flops per target supernode: its updates + POTRF/TRSM on the stored panel."""
    w = np.diff(sym.super).astype(np.float64)
    r = np.diff(sym.i_ptr[sym.super].astype(np.int64)).astype(np.float64)
    cost = w * r * r  # ~ sum_t (r-t)^2 for the panel itself (upper bound, fine for balancing)
    rd = r[sym.updSn]
    m = rd - sym.updLb
    n1 = (sym.updUb - sym.updLb + 1).astype(np.float64)
    K = w[sym.updSn]
    per_upd = K * n1 * (n1 + 1) + 2.0 * K * (m - n1) * n1
    tgt = np.repeat(np.arange(sym.nsuper), np.diff(sym.updPtr))
    np.add.at(cost, tgt, per_upd)
    return cost


def cut_subtrees(sym, nranks: int, oversub: int = 4) -> SubtreeCut:
    """Walk down from the etree roots, always opening the most expensive subtree, until
    there are >= oversub*nranks subtrees; then longest-processing-time bin packing."""
    ns = sym.nsuper
    par = sym.sParent
    cost = supernode_costs(sym)
    sub = cost.copy()
    size = np.ones(ns, dtype=np.int64)
    children = [[] for _ in range(ns)]
    for s in range(ns):  # postorder: children before parents
        p = int(par[s])
        if p >= 0:
            sub[p] += sub[s]
            size[p] += size[s]
            children[p].append(s)
    heap = [(-sub[s], s) for s in range(ns) if par[s] < 0]
    heapq.heapify(heap)
    root_nodes = []
    target = max(1, oversub * nranks) if nranks > 1 else 1
    while heap and len(heap) < target:
        negc, s = heap[0]
        if not children[s]:
            break
        heapq.heappop(heap)
        root_nodes.append(s)
        for c in children[s]:
            heapq.heappush(heap, (-sub[c], c))
    owner = np.full(ns, -1, dtype=np.int32)
    rank_cost = np.zeros(max(nranks, 1))
    subtrees = []
    for negc, s in sorted(heap):  # most expensive first
        rk = int(np.argmin(rank_cost))
        rank_cost[rk] += -negc
        first = s - int(size[s]) + 1
        owner[first:s + 1] = rk
        subtrees.append((first, s, rk, -negc))
    return SubtreeCut(owner=owner, subtrees=subtrees, root_nodes=np.array(sorted(root_nodes), dtype=np.int32),
                      cost=cost, rank_cost=rank_cost)


def gather_to_root(lvalues, cut: SubtreeCut, sym, rank: int, dist, root: int = 0, stage_on_host: bool = False):
    """The one exchange step: every subtree slice of lValues travels from its owner to `root`
    (point-to-point, all owners concurrently -- with RCCL each pair uses its own xGMI link).
    `lvalues` is a 1-D torch tensor of xsize doubles (device memory with nccl; with
    `stage_on_host` -- gloo rehearsals -- device slices are bounced through host copies)."""
    ops, landing = [], []
    for owner, start, stop in cut.slices(sym):
        if owner == root or stop <= start:
            continue
        view = lvalues[start:stop]
        if rank == root:
            buf = view.cpu() if stage_on_host else view
            if stage_on_host:
                landing.append((view, buf))
            ops.append(dist.P2POp(dist.irecv, buf, owner))
        elif rank == owner:
            ops.append(dist.P2POp(dist.isend, view.cpu() if stage_on_host else view, root))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for view, buf in landing:
        view.copy_(buf)
    return sum(stop - start for owner, start, stop in cut.slices(sym) if owner != root)


class PackedExchange:
    """The exchange step with only what the root part reads.

    A root-part target reads, from a supernode d of a subtree, the rows of d that lie in the target's
    columns and below (parallel_PB_Cholesky_05.h:137-149: rows lb..end of the descendant's panel).
    Every root-part column comes after the subtree's last column (postorder), and a panel's rows are
    sorted, so what can ever be read of d is the TAIL of each of its panel columns: rows with index >=
    first column after the subtree.  Those tails -- one contiguous run per panel column -- are packed
    into one send buffer per subtree, travel point to point, and are unpacked into the same positions of
    the root rank's lValues.  The panels' upper parts stay with their owners (the factor is then
    distributed: subtree panels on the owners, root part on the root rank; `gather_to_root` collects
    it where one rank needs all of it, e.g. for a single-GPU solve).
    """

    def __init__(self, sym, cut: SubtreeCut, root: int = 0):
        self.root = root
        self.items = []  # per subtree: (owner, src_off int64[], len int32[], packed_off int64[], total)
        w = np.diff(sym.super).astype(np.int64)
        r = np.diff(sym.i_ptr[sym.super].astype(np.int64))
        pi = sym.i_ptr[sym.super[:-1]].astype(np.int64)
        p = sym.p.astype(np.int64)
        self.full_elements = 0
        self.packed_elements = 0
        for first, last, owner, _ in cut.subtrees:
            limit = int(sym.super[last + 1])
            src, ln = [], []
            for d in range(first, last + 1):
                rows = sym.s[pi[d]: pi[d] + r[d]]
                lb = int(np.searchsorted(rows, limit))
                if lb >= r[d]:
                    continue
                c0 = int(sym.super[d])
                src.append(p[c0: c0 + w[d]] + lb)
                ln.append(np.full(int(w[d]), r[d] - lb, dtype=np.int32))
            src = np.concatenate(src) if src else np.zeros(0, np.int64)
            ln = np.concatenate(ln) if ln else np.zeros(0, np.int32)
            off = np.zeros(len(ln), dtype=np.int64)
            if len(ln):
                np.cumsum(ln[:-1], out=off[1:])
            total = int(ln.sum())
            self.items.append((int(owner), src, ln, off, total))
            if owner != root:
                self.full_elements += int(p[sym.super[last + 1]] - p[sym.super[first]])
                self.packed_elements += total
        self._dev = {}   # device copies of the segment arrays, per item
        self._buf = {}

    def _segments(self, k, like):
        """Segment arrays of item k on the device of tensor `like` (uploaded once)."""
        import torch
        if k not in self._dev:
            _, src, ln, off, _ = self.items[k]
            self._dev[k] = tuple(torch.from_numpy(a).to(like.device) for a in (src, ln, off))
        return self._dev[k]

    def _copy(self, dst, src, dst_off, src_off, ln, k, stream):
        """dst[dst_off[q] + i] = src[src_off[q] + i], i < ln[q] (the host arrays of item k, in either role):
        the library's copy kernel on device tensors, numpy on host tensors."""
        if dst.is_cuda:
            from . import _native as N
            s_src, s_len, s_off = self._segments(k, dst)
            packing = dst_off is self.items[k][3]   # packed offsets on the destination side: pack, else unpack
            a_dst, a_src = (s_off, s_src) if packing else (s_src, s_off)
            if N.lib().parsy_copy_segments_device(dst.data_ptr(), src.data_ptr(), a_dst.data_ptr(), a_src.data_ptr(),
                                                  s_len.data_ptr(), len(ln), stream) != 0:
                raise RuntimeError("parsy_copy_segments_device failed: " + N.last_error())
            return
        d, s = dst.numpy(), src.numpy()
        for q in range(len(ln)):
            d[dst_off[q]: dst_off[q] + ln[q]] = s[src_off[q]: src_off[q] + ln[q]]

    def run(self, lvalues, rank: int, dist, stream: int = 0, stage_on_host: bool = False):
        """pack (owners) -> point-to-point -> unpack (root).  `lvalues`: 1-D torch tensor of xsize doubles.
        The caller's stream must be the current torch stream (the send / receive are ordered on it)."""
        import torch
        ops, landing = [], []
        for k, (owner, src, ln, off, total) in enumerate(self.items):
            if owner == self.root or total == 0 or rank not in (owner, self.root):
                continue
            if k not in self._buf or self._buf[k].device != lvalues.device:
                self._buf[k] = torch.empty(total, dtype=torch.float64, device=lvalues.device)
            buf = self._buf[k]
            if rank == owner:
                self._copy(buf, lvalues, off, src, ln, k, stream)
                ops.append(dist.P2POp(dist.isend, buf.cpu() if stage_on_host else buf, self.root))
            else:
                host = torch.empty(total, dtype=torch.float64) if stage_on_host else None
                ops.append(dist.P2POp(dist.irecv, host if stage_on_host else buf, owner))
                landing.append((k, host))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for k, host in landing:
            _, src, ln, off, _ = self.items[k]
            if host is not None:
                self._buf[k].copy_(host)
            self._copy(lvalues, self._buf[k], src, off, ln, k, stream)
        return self.packed_elements
