// Distribution of one factorization over the devices of a node (see dist.hpp).
#include "dist.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <queue>
#include <stdexcept>

namespace parsy {

namespace {

// The source pieces of update u of target piece t and the first row (counted in the source SUPERNODE's panel)
// the update reads: an external update reads rows lb.. of every column of the descendant -- all its pieces --,
// an identity update (piece -> piece of one split supernode) reads the target's rows of the source pieces'
// columns.
struct Source { int32_t first_piece, last_piece, row0; };

Source update_source(const Schedule& S, int t, int64_t u) {
    const UpdDesc& U = S.upd[(size_t)u];
    const int last = S.upd_src[(size_t)u];
    if (U.rel >= 0) {
        const int d = S.csn_real[(size_t)last];
        return Source{S.piece0[(size_t)d], last, (int32_t)(U.src - S.sn[(size_t)d].px)};
    }
    int first = last, K = U.K - S.csn[(size_t)last].w;
    while (K > 0 && first > S.piece0[(size_t)S.csn_real[(size_t)t]]) K -= S.csn[(size_t)--first].w;
    if (K != 0) throw std::runtime_error("dist: an identity update does not match the pieces to its left");
    return Source{first, last, S.csn[(size_t)t].rbias};
}

double update_flops(const UpdDesc& U) {
    return (double)U.K * U.n1 * (U.n1 + 1.0) + 2.0 * U.K * (double)(U.m - U.n1) * U.n1;
}

}  // namespace

void build_dist(const Schedule& S, int nranks, int block, Dist& D) {
    D = Dist();
    if (nranks < 1) throw std::runtime_error("dist: need at least one rank");
    if (S.solve_only) throw std::runtime_error("dist: the plan has no factorization schedule (solve-only)");
    block = std::max(1, block);
    const int ns = S.nsuper, nc = (int)S.csn.size();
    D.nranks = nranks;
    D.nlevels = S.cnlevels;
    D.npieces = nc;
    // ---- cost of every piece: the updates into it + its own POTRF / TRSM
    D.cost.assign((size_t)nc, 0.0);
    for (int t = 0; t < nc; ++t) {
        const SnDesc& C = S.csn[(size_t)t];
        double c = (double)C.w * C.r * C.r / 3.0 + (double)C.w * C.w * (C.r - C.w);
        for (int64_t u = C.upd0; u < C.upd0 + C.nupd; ++u) c += update_flops(S.upd[(size_t)u]);
        D.cost[(size_t)t] = c;
        D.total_cost += c;
    }
    // ---- the cut: open the most expensive subtrees from the roots down until the subtrees, packed heaviest
    // first, give no rank more than its share of the whole job (the pieces above the cut then even the ranks
    // out).  A subtree that one workgroup walks (Schedule::chol_subtree) is never opened.
    std::vector<double> sub((size_t)ns, 0.0);
    std::vector<int32_t> size((size_t)ns, 1), lowest((size_t)ns);
    for (int s = 0; s < ns; ++s) lowest[(size_t)s] = s;
    std::vector<std::vector<int32_t>> children((size_t)ns);
    for (int s = 0; s < ns; ++s)
        for (int p = S.piece0[(size_t)s]; p < S.piece0[(size_t)s + 1]; ++p) sub[(size_t)s] += D.cost[(size_t)p];
    for (int s = 0; s < ns; ++s) {
        const int p = S.sparent[(size_t)s];
        if (p < 0) continue;
        if (p <= s) throw std::runtime_error("dist: the supernodal etree is not postordered");
        // (children come before their parent, so lowest[s] / size[s] are final here) a subtree is handed out below as
        // the index range [s - size + 1, s]: that needs a true postorder -- every subtree contiguous -- not merely
        // parent > child
        if (lowest[(size_t)s] != s - size[(size_t)s] + 1)
            throw std::runtime_error("dist: the supernodal etree is topologically ordered but not postordered "
                                     "(the subtree of supernode " + std::to_string(s) + " is not a contiguous index range)");
        sub[(size_t)p] += sub[(size_t)s];
        size[(size_t)p] += size[(size_t)s];
        lowest[(size_t)p] = std::min(lowest[(size_t)p], lowest[(size_t)s]);
        children[(size_t)p].push_back(s);
    }
    for (int s = 0; s < ns; ++s)
        if (S.sparent[(size_t)s] < 0 && lowest[(size_t)s] != s - size[(size_t)s] + 1)
            throw std::runtime_error("dist: the supernodal etree is not postordered (root " + std::to_string(s) + ")");
    using Item = std::pair<double, int32_t>;
    std::priority_queue<Item> heap;
    for (int s = 0; s < ns; ++s)
        if (S.sparent[(size_t)s] < 0) heap.push({sub[(size_t)s], s});
    std::vector<uint8_t> opened((size_t)ns, 0);
    auto packed_max = [&](std::priority_queue<Item> h) {   // (copy: heaviest first onto the least loaded rank)
        std::vector<double> load((size_t)nranks, 0.0);
        while (!h.empty()) {
            *std::min_element(load.begin(), load.end()) += h.top().first;
            h.pop();
        }
        return *std::max_element(load.begin(), load.end());
    };
    const double share = D.total_cost / nranks;
    // PARSY_DIST_MIN_SUBTREES=k: keep opening until there are at least k subtrees (diagnostics, tests: a deeper cut)
    int min_subtrees = nranks;
    if (const char* e = std::getenv("PARSY_DIST_MIN_SUBTREES")) min_subtrees = std::max(nranks, std::atoi(e));
    while (nranks > 1 && !heap.empty()) {
        if ((int)heap.size() >= min_subtrees && packed_max(heap) <= share) break;
        const int s = heap.top().second;
        if (children[(size_t)s].empty() || S.chol_subtree[(size_t)s] >= 0) break;
        if ((int)heap.size() >= 64 * nranks) break;
        heap.pop();
        opened[(size_t)s] = 1;
        for (int c : children[(size_t)s]) heap.push({sub[(size_t)c], c});
    }
    D.owner.assign((size_t)nc, -1);
    D.in_subtree.assign((size_t)nc, 0);
    D.rank_cost.assign((size_t)nranks, 0.0);
    D.n_subtrees = (int)heap.size();
    while (!heap.empty()) {
        const auto [c, s] = heap.top();
        heap.pop();
        const int rk = (int)(std::min_element(D.rank_cost.begin(), D.rank_cost.end()) - D.rank_cost.begin());
        D.rank_cost[(size_t)rk] += c;
        for (int q = s - size[(size_t)s] + 1; q <= s; ++q)
            for (int p = S.piece0[(size_t)q]; p < S.piece0[(size_t)q + 1]; ++p) {
                D.owner[(size_t)p] = rk;
                D.in_subtree[(size_t)p] = 1;
            }
    }
    // ---- above the cut: the pieces level by level, heaviest first, each to the least loaded rank; `block`
    // consecutive pieces of one split supernode stay together (fewer hand-offs on its chain)
    for (int lev = 0; lev < S.cnlevels; ++lev) {
        std::vector<int32_t> here;
        for (int q = S.clevelPtr[(size_t)lev]; q < S.clevelPtr[(size_t)lev + 1]; ++q)
            if (D.owner[(size_t)S.clevelSet[(size_t)q]] < 0) here.push_back(S.clevelSet[(size_t)q]);
        std::stable_sort(here.begin(), here.end(), [&](int a, int b) { return D.cost[(size_t)a] > D.cost[(size_t)b]; });
        for (int p : here) {
            const int j = p - S.piece0[(size_t)S.csn_real[(size_t)p]];
            int rk;
            if (j % block != 0) rk = D.owner[(size_t)p - 1];
            else rk = (int)(std::min_element(D.rank_cost.begin(), D.rank_cost.end()) - D.rank_cost.begin());
            D.owner[(size_t)p] = rk;
            D.rank_cost[(size_t)rk] += D.cost[(size_t)p];
            D.root_cost += D.cost[(size_t)p];
            ++D.n_root_pieces;
        }
    }
    D.level_cost.assign((size_t)S.cnlevels * nranks, 0.0);
    for (int p = 0; p < nc; ++p) D.level_cost[(size_t)S.level_of[(size_t)p] * nranks + D.owner[(size_t)p]] += D.cost[(size_t)p];

    // ---- what travels: per (source piece, consuming rank) the first row any of that rank's targets reads
    std::vector<int32_t> need((size_t)nc * nranks, INT_MAX);
    for (int t = 0; t < nc && nranks > 1; ++t) {
        const SnDesc& C = S.csn[(size_t)t];
        const int rk = D.owner[(size_t)t];
        for (int64_t u = C.upd0; u < C.upd0 + C.nupd; ++u) {
            const Source src = update_source(S, t, u);
            for (int q = src.first_piece; q <= src.last_piece; ++q) {
                if (D.owner[(size_t)q] == rk) continue;
                int32_t& nd = need[(size_t)q * nranks + rk];
                nd = std::min(nd, src.row0);
            }
        }
    }
    D.level_msg0.assign((size_t)S.cnlevels + 1, 0);
    for (int lev = 0; lev < S.cnlevels && nranks > 1; ++lev) {
        D.level_msg0[(size_t)lev] = (int64_t)D.msgs.size();
        std::vector<int32_t> slot((size_t)nranks * nranks, -1);
        for (int qi = S.clevelPtr[(size_t)lev]; qi < S.clevelPtr[(size_t)lev + 1]; ++qi) {
            const int q = S.clevelSet[(size_t)qi];
            const int a = D.owner[(size_t)q];
            const SnDesc& R = S.sn[(size_t)S.csn_real[(size_t)q]];
            const SnDesc& Q = S.csn[(size_t)q];
            for (int k = 0; k < nranks; ++k) {
                const int32_t row0 = need[(size_t)q * nranks + k];
                if (row0 == INT_MAX) continue;
                if (row0 < 0 || row0 >= R.r) throw std::runtime_error("dist: a needed row lies outside its panel");
                int32_t& sl = slot[(size_t)a * nranks + k];
                if (sl < 0) {
                    sl = (int32_t)D.msgs.size();
                    D.msgs.push_back(DistMessage());
                    D.msgs.back().level = lev;
                    D.msgs.back().src = a;
                    D.msgs.back().dst = k;
                }
                DistMessage& M = D.msgs[(size_t)sl];
                for (int c = Q.rbias; c < Q.rbias + Q.w; ++c) {
                    M.off.push_back(R.px + (int64_t)c * R.r + row0);
                    M.len.push_back(R.r - row0);
                    M.packed.push_back(M.total);
                    M.total += R.r - row0;
                }
            }
        }
        // (src, dst) order inside a level
        std::stable_sort(D.msgs.begin() + D.level_msg0[(size_t)lev], D.msgs.end(), [](const DistMessage& x, const DistMessage& y) {
            return x.src != y.src ? x.src < y.src : x.dst < y.dst;
        });
    }
    for (int lev = 0; lev < S.cnlevels && nranks == 1; ++lev) D.level_msg0[(size_t)lev] = 0;
    D.level_msg0[(size_t)S.cnlevels] = (int64_t)D.msgs.size();
    for (const DistMessage& M : D.msgs) D.exchange_elements += M.total;
}

int64_t check_dist(const Schedule& S, const Dist& D, std::string& what) {
    int64_t bad = 0;
    auto fail = [&](const std::string& msg) {
        if (bad++ == 0) what = msg;
    };
    const int nc = (int)S.csn.size(), nr = D.nranks;
    if ((int)D.owner.size() != nc) {
        fail("owner array does not cover the pieces");
        return bad;
    }
    for (int p = 0; p < nc; ++p)
        if (D.owner[(size_t)p] < 0 || D.owner[(size_t)p] >= nr) fail("piece " + std::to_string(p) + " has no owner");
    if (bad) return bad;
    // a subtree that one workgroup walks belongs to one rank
    {
        std::vector<int32_t> st_owner((size_t)std::max(S.n_chol_subtrees, 1), -1);
        for (int t = 0; t < S.nsuper; ++t) {
            const int st = S.chol_subtree[(size_t)t];
            if (st < 0) continue;
            const int o = D.owner[(size_t)S.piece0[(size_t)t]];
            if (st_owner[(size_t)st] < 0) st_owner[(size_t)st] = o;
            else if (st_owner[(size_t)st] != o) fail("subtree " + std::to_string(st) + " of narrow supernodes is split over ranks");
        }
    }
    // the part below the cut is descendant-closed per rank: a supernode marked in_subtree has every child marked too
    // and owned by the same rank (what the sharded solves rely on when they take in_subtree as "my subtrees")
    if ((int)D.in_subtree.size() == nc)
        for (int t = 0; t < S.nsuper; ++t) {
            const int par = S.sparent[(size_t)t];
            if (par < 0 || !D.in_subtree[(size_t)S.piece0[(size_t)par]]) continue;
            if (!D.in_subtree[(size_t)S.piece0[(size_t)t]])
                fail("supernode " + std::to_string(par) + " lies below the cut but its child " + std::to_string(t) + " does not");
            else if (D.owner[(size_t)S.piece0[(size_t)t]] != D.owner[(size_t)S.piece0[(size_t)par]])
                fail("supernode " + std::to_string(par) + " and its child " + std::to_string(t) + " lie below the cut on different ranks");
        }
    // delivered[(piece, rank)] = first row delivered to that rank, from the messages of the piece's level
    std::vector<int32_t> delivered((size_t)nc * nr, INT_MAX);
    std::vector<int32_t> col2piece;  // per real supernode: piece of every column (built on demand below)
    for (const DistMessage& M : D.msgs) {
        if (M.src == M.dst) fail("a message goes from a rank to itself");
        if (M.off.size() != M.len.size() || M.off.size() != M.packed.size()) {
            fail("message arrays differ in length");
            continue;
        }
        int64_t run = 0;
        for (size_t k = 0; k < M.off.size(); ++k) {
            if (M.packed[k] != run) fail("packed offsets are not the prefix sums of the lengths");
            run += M.len[k];
            // locate the supernode / column / row of the segment
            const int64_t off = M.off[k];
            int lo = 0, hi = S.nsuper - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) / 2;
                if (S.sn[(size_t)mid].px <= off) lo = mid;
                else hi = mid - 1;
            }
            const SnDesc& R = S.sn[(size_t)lo];
            const int64_t rel = off - R.px;
            const int col = (int)(rel / R.r), row = (int)(rel % R.r);
            if (col >= R.w || row + M.len[k] != R.r) {
                fail("a segment is not the tail of a panel column");
                continue;
            }
            int p = S.piece0[(size_t)lo];
            while (p + 1 < S.piece0[(size_t)lo + 1] && S.csn[(size_t)p + 1].rbias <= col) ++p;
            if (D.owner[(size_t)p] != M.src) fail("a message carries a piece its sender does not own");
            if (S.level_of[(size_t)p] != M.level) fail("a piece travels after another level than its own");
            if (col == S.csn[(size_t)p].rbias) delivered[(size_t)p * nr + M.dst] = std::min(delivered[(size_t)p * nr + M.dst], row);
            else if (delivered[(size_t)p * nr + M.dst] != row) fail("the columns of a piece travel from different rows");
        }
        if (run != M.total) fail("message total differs from the sum of its segments");
    }
    for (int t = 0; t < nc; ++t) {
        const SnDesc& C = S.csn[(size_t)t];
        const int rk = D.owner[(size_t)t];
        for (int64_t u = C.upd0; u < C.upd0 + C.nupd; ++u) {
            const Source src = update_source(S, t, u);
            for (int q = src.first_piece; q <= src.last_piece; ++q) {
                if (D.owner[(size_t)q] == rk) continue;
                if (delivered[(size_t)q * nr + rk] > src.row0)
                    fail("update " + std::to_string(u) + " of piece " + std::to_string(t) + " reads rows of piece " +
                         std::to_string(q) + " that never reach rank " + std::to_string(rk));
                if (S.level_of[(size_t)q] >= S.level_of[(size_t)t]) fail("a source is not below its target");
            }
        }
    }
    return bad;
}

}  // namespace parsy
