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
