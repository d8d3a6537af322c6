// Host-side construction of the device schedule (see schedule.hpp).
#include "schedule.hpp"

#include <cstdio>
#include <map>
#include <algorithm>
#include <queue>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>

namespace parsy {

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

static int env_int(const char* name, int dflt) {
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    return std::atoi(v);
}

// Subtrees of the etree that consist of eligible supernodes only and cost at most `cap`, as large as
// possible: subtree[s] = its subtree (numbered from the last root down) or -1.  `parent` must be a
// postordered forest (parent[s] > s); otherwise no subtree is formed.
// slots (optional): per supernode (columns, rows below); a subtree's columns plus the rows below its root must not
// exceed slot_cap (the LDS slots of the solves' subtree kernels for many right-hand sides).
static int find_subtrees(const std::vector<int>& parent, const std::vector<uint8_t>& eligible,
                         const std::vector<double>& cost, double cap, std::vector<int32_t>& subtree,
                         const std::vector<std::pair<int32_t, int32_t>>* slots = nullptr, int64_t slot_cap = 0) {
    const int ns = (int)parent.size();
    subtree.assign(ns, -1);
    std::vector<uint8_t> whole(eligible);  // the supernode and everything below it is eligible
    std::vector<double> sub(cost);
    std::vector<int64_t> cols(slots ? ns : 0);
    for (int s = 0; s < ns && slots; ++s) cols[s] = (*slots)[s].first;
    for (int s = 0; s < ns; ++s) {
        const int p = parent[s];
        if (p < 0) continue;
        if (p <= s) return 0;
        if (!whole[s]) whole[p] = 0;
        sub[p] += sub[s];
        if (slots) cols[p] += cols[s];
    }
    int count = 0;
    for (int s = ns - 1; s >= 0; --s) {
        const int p = parent[s];
        if (p >= 0 && subtree[p] >= 0) subtree[s] = subtree[p];
        else if (whole[s] && sub[s] <= cap && (!slots || cols[s] + (*slots)[s].second <= slot_cap)) subtree[s] = count++;
    }
    return count;
}

void build_schedule(const PatternRef& P, const size_t* lC, const int* A2p, const int* A2i,
                    const uint8_t* active, Schedule& S, int compute_units) {
    S = Schedule();
    // resident workgroups of the chain kernel = 2 per CU; walkers of one batch take at most a quarter
    if (compute_units > 0) S.walker_batch = std::max(4, std::min(kWalkerBatch, compute_units / 2));
    const int n = P.n, ns = P.nsuper;
    S.n = n;
    S.nsuper = ns;
    S.solve_only = (A2p == nullptr);
    S.nnzA = A2p ? A2p[n] : 0;
    S.rows.assign(P.s, P.s + P.i_ptr[n]);
    S.ssize = (int64_t)P.i_ptr[n];
    S.xsize = (int64_t)lC[n];
    // --- supernode descriptors ------------------------------------------------
    S.sn.resize(ns);
    for (int t = 0; t < ns; ++t) {
        SnDesc& d = S.sn[t];
        d.c0 = P.super[t];
        d.w = P.super[t + 1] - P.super[t];
        d.r = (int)(P.i_ptr[P.super[t + 1]] - P.i_ptr[P.super[t]]);
        d.ld = d.r;
        d.rbias = 0;
        d.px = (int64_t)lC[d.c0];
        d.pi = (int64_t)P.i_ptr[d.c0];
        d.a0 = A2p ? A2p[d.c0] : 0;
        d.a1 = A2p ? A2p[P.super[t + 1]] : 0;
        d.upd0 = 0;
        d.nupd = 0;
        d.dslot = -1;
        d.tflag0 = -1;
        if (d.w <= 0 || d.r < d.w) throw std::runtime_error("schedule: malformed supernode");
        if ((int64_t)d.w * d.r > 0x7fffffffLL)
            throw std::runtime_error("schedule: a single panel exceeds 2^31 entries");
        for (int col = d.c0; col < d.c0 + d.w; ++col)
            if (lC[col + 1] - lC[col] != (size_t)d.r || P.i_ptr[col] != P.i_ptr[d.c0])
                throw std::runtime_error("schedule: lC / Li_ptr are not a column-major panel layout");
        S.max_width = std::max(S.max_width, d.w);
        S.max_rows = std::max(S.max_rows, d.r);
        S.nnzL += (int64_t)d.w * d.r - (int64_t)d.w * (d.w - 1) / 2;
        for (int k = 0; k < d.w; ++k) S.flops_stored += (double)(d.r - k) * (double)(d.r - k);
        for (int k = 0; k < d.w; ++k)
            if (S.rows[d.pi + k] != d.c0 + k)
                throw std::runtime_error("schedule: supernode rows do not start with its own columns");
        // solve: one scratch slot (inverse diagonal block) per block column of the wide supernodes
        if (d.w > kTile) {
            d.dslot = (int32_t)S.n_dslots;
            S.n_dslots += ceil_div(d.w, kTile);
        }
    }

    // --- update lists of the supernodes, relative indices, A scatter map -----------
    std::vector<int64_t> uptr(ns + 1, 0);
    std::vector<int> usn, ulb, uub;
    if (!S.solve_only) build_update_lists(P, uptr, usn, ulb, uub);
    std::vector<int64_t> urel(usn.size(), 0);  // per (target, descendant): offset into relpos of row lb
    S.a_dst.assign((size_t)S.nnzA, 0);
    std::vector<int> map(n, -1), stamp(n, -1);
    for (int t = 0; t < ns; ++t) {
        const SnDesc& T = S.sn[t];
        for (int k = 0; k < T.r; ++k) {
            const int row = S.rows[T.pi + k];
            map[row] = k;
            stamp[row] = t;
        }
        for (int col = T.c0; !S.solve_only && col < T.c0 + T.w; ++col)
            for (int q = A2p[col]; q < A2p[col + 1]; ++q) {
                const int row = A2i[q];
                if (stamp[row] != t) throw std::runtime_error("schedule: A entry outside the pattern of L");
                S.a_dst[q] = (int64_t)lC[col] + map[row];
            }
        for (int64_t u = uptr[t]; u < uptr[t + 1]; ++u) {
            const SnDesc& D = S.sn[usn[u]];
            urel[u] = (int64_t)S.relpos.size();
            for (int k = ulb[u]; k < D.r; ++k) {
                const int row = S.rows[D.pi + k];
                if (stamp[row] != t)
                    throw std::runtime_error("schedule: descendant row missing from the target's pattern");
                S.relpos.push_back(map[row]);
            }
            const double K = D.w, m = D.r - ulb[u], n1 = uub[u] - ulb[u] + 1;
            S.update_flops += K * n1 * (n1 + 1) + 2.0 * K * (m - n1) * n1;
            S.reread_bytes += 8.0 * K * m;
        }
    }
    if (S.relpos.size() > 0x7fffffffULL) throw std::runtime_error("schedule: relative index array exceeds int32");

    // Which form the updates take follows the size of the job (measured on MI355X, 27-point grids with
    // nested dissection): below about 1e11 update flops a factorization is a chain of latencies and the
    // wave streams inside the chain launches are the shortest path; above it the LDS-staged BIG launches
    // win (56^3 grid: 11.8 vs 13.1 ms); cutting the very wide supernodes into pieces pays from about 2e12
    // (96^3 grid: 163 vs 174 ms).  PARSY_BIG_MINK / PARSY_PIECE_WIDTH override (diagnostics, tests).
    S.big_min_k = S.update_flops >= kBigAutoFlops ? kBigMinK : INT_MAX;
    S.piece_width = S.update_flops >= kPieceAutoFlops ? kPieceWidth : 0;
    if (std::getenv("PARSY_BIG_MINK")) S.big_min_k = std::max(1, env_int("PARSY_BIG_MINK", kBigMinK));
    if (std::getenv("PARSY_PIECE_WIDTH")) S.piece_width = env_int("PARSY_PIECE_WIDTH", kPieceWidth);
    if (S.piece_width > 0) S.piece_width = ceil_div(std::max(S.piece_width, kBigTile), kBigTile) * kBigTile;
    if (S.piece_width < S.big_min_k) S.piece_width = 0;  // the pieces update each other through the BIG launches
    S.push_group = std::max(1, env_int("PARSY_PUSH_GROUP", kPushGroup));
    // (early wave streams longer than split_chunks 16-wide chunks are cut into parts of about split_target, at most split_max)
    const int split_chunks = env_int("PARSY_SPLIT_CHUNKS", kSplitChunks), split_target = std::max(1, env_int("PARSY_SPLIT_TARGET", kSplitTarget)),
              split_max = std::max(1, env_int("PARSY_SPLIT_MAX", kSplitMaxParts));
    const bool split_env = std::getenv("PARSY_SPLIT_CHUNKS") != nullptr;   // (set: one rule for every level -- diagnostics)

    std::vector<int> tree(P.sparent, P.sparent + ns);
    S.sparent = tree;
    level_sets(tree, S.levelPtr, S.levelSet);
    S.nlevels = (int)S.levelPtr.size() - 1;

    // --- Cholesky view: pieces of the very wide supernodes -------------------------------
    S.piece0.assign(ns + 1, 0);
    for (int t = 0; t < ns; ++t) {
        const SnDesc& T = S.sn[t];
        S.piece0[t] = (int32_t)S.csn.size();
        int np = 1;
        const int pw = S.piece_width;
        if (!S.solve_only && pw > 0 && T.w > pw + pw / 2) {
            np = ceil_div(T.w, pw);
            if (T.w - (np - 1) * pw < pw / 2) --np;  // a short remainder stays with the last piece
        }
        for (int j = 0; j < np; ++j) {
            const int off = j * pw;
            SnDesc c = T;
            c.c0 = T.c0 + off;
            c.w = (j == np - 1) ? T.w - off : pw;
            c.r = T.r - off;
            c.ld = T.r;
            c.rbias = off;
            c.px = T.px + (int64_t)off * T.r + off;
            c.pi = T.pi + off;
            c.dslot = -1;
            S.csn.push_back(c);
            S.csn_real.push_back(t);
        }
    }
    const int nc = (int)S.csn.size();
    S.piece0[ns] = nc;
    {
        std::vector<int> ctree(nc, -1);
        for (int t = 0; t < ns; ++t)
            for (int p = S.piece0[t]; p < S.piece0[t + 1]; ++p)
                ctree[p] = (p + 1 < S.piece0[t + 1]) ? p + 1 : (tree[t] >= 0 ? S.piece0[tree[t]] : -1);
        level_sets(ctree, S.clevelPtr, S.clevelSet);
    }
    S.cnlevels = (int)S.clevelPtr.size() - 1;
    S.level_of.assign(nc, 0);
    for (int l = 0; l < S.cnlevels; ++l)
        for (int q = S.clevelPtr[l]; q < S.clevelPtr[l + 1]; ++q) S.level_of[S.clevelSet[q]] = l;

    // --- update descriptors per piece: the supernode's descendants restricted to the piece's columns
    // (reference order), then the pieces to its left (identity row map) ------------------------
    for (int t = 0; t < ns && !S.solve_only; ++t) {
        const SnDesc& T = S.sn[t];
        for (int p = S.piece0[t]; p < S.piece0[t + 1]; ++p) {
            SnDesc& C = S.csn[p];
            C.upd0 = (int64_t)S.upd.size();
            const int col_lo = C.c0, col_hi = C.c0 + C.w;
            for (int64_t u = uptr[t]; u < uptr[t + 1]; ++u) {
                const SnDesc& D = S.sn[usn[u]];
                int lb = ulb[u], ub = uub[u];
                if (S.piece0[t + 1] - S.piece0[t] > 1) {
                    while (lb <= ub && S.rows[D.pi + lb] < col_lo) ++lb;
                    while (ub >= lb && S.rows[D.pi + ub] >= col_hi) --ub;
                    if (lb > ub) continue;
                }
                UpdDesc U;
                U.src = D.px + lb;
                U.ld = D.r;
                U.K = D.w;
                U.m = D.r - lb;
                U.n1 = ub - lb + 1;
                U.rel = urel[u] + (lb - ulb[u]);
                S.upd.push_back(U);
                S.upd_src.push_back(S.piece0[usn[u] + 1] - 1);
            }
            // the pieces to the left: the one right before this piece by itself (it is applied by the NEXT
            // launch of its level), the earlier ones in aligned groups of push_group pieces -- adjacent
            // columns of the same panel, one update with the summed width, due when the group's last
            // piece is complete (a group is cut short where this piece's turn comes first)
            const int j = p - S.piece0[t];
            for (int g0 = 0; g0 < j; ) {
                int g1 = (g0 == j - 1) ? j : std::min(j - 1, (g0 / S.push_group + 1) * S.push_group);
                const SnDesc& Q0 = S.csn[S.piece0[t] + g0];
                int K = 0;
                for (int q = g0; q < g1; ++q) K += S.csn[S.piece0[t] + q].w;
                UpdDesc U;
                U.src = T.px + (int64_t)Q0.rbias * T.r + C.rbias;
                U.ld = T.r;
                U.K = K;
                U.m = C.r;
                U.n1 = C.w;
                U.rel = -1;
                S.upd.push_back(U);
                S.upd_src.push_back(S.piece0[t] + g1 - 1);
                g0 = g1;
            }
            C.nupd = (int)((int64_t)S.upd.size() - C.upd0);
        }
    }
    // --- subtrees walked by one workgroup each (see schedule.hpp).  Cost in flop equivalents: what the
    // SMALL kernel executes plus a fixed amount per supernode and update (the small ones are latency);
    // solves: panel entries.  The cap aims at kSubtreesPerCu subtrees per compute unit.
    // A workgroup walks the supernodes of its subtree one after the other, siblings included, which level
    // launches would run side by side: that only pays when the bottom of the etree saturates the device anyway
    // (measured on MI355X: 45 000 narrow supernodes: factor -3 %, backward solve -7 %, forward solve -3 %;
    // 1 300 .. 3 300 of them: +1 .. +9 % with subtrees), so by default subtrees are cut only where the
    // eligible supernodes number at least kSubtreeMinPerSlot per subtree aimed at.
    // PARSY_SUBTREES=k: always cut, aiming at k subtrees per compute unit (0: never; diagnostics, tests).
    const bool forced = std::getenv("PARSY_SUBTREES") != nullptr;
    const int per_cu = env_int("PARSY_SUBTREES", kSubtreesPerCu);
    if (per_cu > 0) {
        const int cus = compute_units > 0 ? compute_units : 256;
        std::vector<uint8_t> elig(ns, 0);
        auto cut = [&](std::vector<double>& cost, double min_cost, std::vector<int32_t>& subtree,
                       const std::vector<std::pair<int32_t, int32_t>>* slots = nullptr, int64_t min_members = -1, int aim_per_cu = 0) {
            double total = 0;
            int64_t members = 0;
            {   // everything that could be in a subtree at all
                std::vector<int32_t> all;
                find_subtrees(tree, elig, cost, 1e300, all, slots, kSubMaxSlots);
                for (int t = 0; t < ns; ++t)
                    if (all[t] >= 0) {
                        total += cost[t];
                        ++members;
                    }
            }
            if (!forced && members < (min_members >= 0 ? min_members : (int64_t)kSubtreeMinPerSlot * per_cu * cus)) return 0;
            return find_subtrees(tree, elig, cost, std::max(min_cost, total / ((double)(aim_per_cu > 0 ? aim_per_cu : per_cu) * cus)), subtree, slots,
                                 kSubMaxSlots);
        };
        if (!S.solve_only) {
            S.chol_cost.assign(ns, 0.0);
            for (int t = 0; t < ns; ++t) {
                const SnDesc& C = S.csn[S.piece0[t]];
                elig[t] = S.piece0[t + 1] - S.piece0[t] == 1 && is_small(C);
                double c = 5e4 + (double)C.w * C.r * C.r;
                for (int64_t u = C.upd0; u < C.upd0 + C.nupd && elig[t]; ++u)
                    c += 1e4 + 2.0 * S.upd[u].K * (double)S.upd[u].m * S.upd[u].n1;
                S.chol_cost[t] = c;
            }
            // (the factorization's subtrees aim at more, smaller ones than the threshold above counts with: a workgroup
            // keeps a whole panel in LDS, one or two per compute unit, and the launch runs alone on the device -- Flan-class
            // 3.8 -> 2.9 ms with 64 per unit, 3.2 with 128: profiles/r05_dense_thresholds.txt.  PARSY_CHOL_SUBTREES)
            S.n_chol_subtrees = cut(S.chol_cost, kSubtreeMinCost, S.chol_subtree, nullptr, -1,
                                    forced ? 0 : std::max(1, env_int("PARSY_CHOL_SUBTREES", kCholSubtreesPerCu)));
        }
        S.solve_cost.assign(ns, 0.0);
        // solves: the subtrees of TINY supernodes (one wave walks one: the wave-per-supernode solve kernel)
        std::vector<std::pair<int32_t, int32_t>> slots((size_t)ns);
        for (int t = 0; t < ns; ++t) {
            elig[t] = S.sn[t].w <= kTinyWidth;
            S.solve_cost[t] = 2e3 + (double)S.sn[t].w * S.sn[t].r;
            slots[(size_t)t] = {S.sn[t].w, S.sn[t].r - S.sn[t].w};
        }
        // (the solves cut subtrees from far fewer supernodes on than the factorization: tools/gate_sweep.py, round 5 -- grids of
        // 3 400 .. 9 000 supernodes: forward solve -12 .. -23 %, 8 right-hand sides up to -44 %; 750 .. 1 450: +-8 %)
        S.n_solve_subtrees = cut(S.solve_cost, kSubtreeMinCost / 16, S.solve_subtree, &slots, kSolveSubtreeMinMembers);
        // (the backward solve walks the same subtrees from their roots down)
        S.bsolve_subtree = S.solve_subtree;
        S.n_bsolve_subtrees = S.n_solve_subtrees;
    }
    if (S.chol_subtree.empty()) S.chol_subtree.assign(ns, -1);
    if (S.solve_subtree.empty()) S.solve_subtree.assign(ns, -1);
    if (S.bsolve_subtree.empty()) S.bsolve_subtree.assign(ns, -1);

    // Updates that go through the BIG launches: wide descendants, and EVERY update of a split supernode's
    // pieces (what little reaches those from narrow descendants would otherwise be a TILES launch per
    // piece, in between the PUSH launches of the side stream).
    auto is_big = [&](const UpdDesc& U, int target_piece) {
        const int real = S.csn_real[target_piece];
        return U.rel < 0 || U.K >= S.big_min_k || S.piece0[real + 1] - S.piece0[real] > 1;
    };

    // --- tiled supernodes: scratch slots and the per-wave update streams ---------------
    // Every (target, descendant) update is cut along the 32-row windows of the target's tiles:
    // the descendant rows that fall into row window I32 times those (among its first n1 rows)
    // that fall into column window J32 are one WaveEntry of sub-tile (I32, J32), I32 >= J32.
    // Lists keep the reference's update order, so the sum order per entry of L is fixed.
    // Updates from wide descendants (is_big) do not enter these lists: BIG launches below.
    S.sn_wp0.assign(nc, -1);
    S.sn_tw0.assign(nc, -1);
    struct Group { int32_t win, first, len; };
    std::vector<Group> groups, cgroups;
    std::vector<int64_t> cursor;
    struct BigKeyed { int64_t launch_tile; WaveEntry e; int16_t sr, sc; };  // launch id << 40 | global tile index
    std::vector<BigKeyed> bigk;
    std::vector<int64_t> big_tile0(nc + 1, 0);  // first 128x128 tile index of every tiled piece
    // Edge of a BIG task's super-tile in 128 x 128 tiles, rows x columns, per launch (source level, kind).  A launch
    // with many tasks whose 128 x 128 windows are mostly ragged (a 128-row window of the target holds about 64 rows of
    // a typical source) takes 2 x 2 super-tiles -- a source's rows inside the larger windows are cut into FULL
    // 128 x 128 blocks in its own row order: a quarter fewer chunks for the same products -- the others (pushes
    // between the pieces of a separator: full windows anyway; launches of few tasks: they end with their longest
    // task) stay with single tiles.  Measured per launch on the Flan-class input (tools/big_launches.py): early levels
    // 10.7 -> 7.8 ms, the big pushes 37 -> 34 ms, launches below ~12 000 tasks 3-10 % slower with super-tiles.
    // PARSY_BIG_SUPER=RxC (or R) forces one size everywhere (diagnostics, tests).
    int env_sr = 0, env_sc = 0;
    if (const char* e = std::getenv("PARSY_BIG_SUPER")) {
        int a = 0, b = 0;
        const int got = std::sscanf(e, "%dx%d", &a, &b);
        if (got >= 1 && a >= 1 && a <= 64) {
            env_sr = a;
            env_sc = (got == 2 && b >= 1 && b <= 64) ? b : a;
        }
    }
    std::vector<int8_t> launch_super((size_t)2 * std::max(S.cnlevels, 1), 1);
    if (env_sr == 0 && !S.solve_only) {
        // (PARSY_BIG_SUPER_MIN / PARSY_BIG_SUPER_FILL: the two thresholds, diagnostics; fill in percent)
        const int64_t super_min_tasks = env_int("PARSY_BIG_SUPER_MIN", kBigSuperMinTasks);
        const double super_max_fill = env_int("PARSY_BIG_SUPER_FILL", (int)(kBigSuperMaxFill * 100 + 0.5)) / 100.0;
        const int super_hi = std::max(1, std::min(8, env_int("PARSY_BIG_SUPER_HI", 2)));   // (diagnostics: the edge such launches take)
        std::vector<int64_t> ltasks(launch_super.size(), 0);
        std::vector<double> lfrag(launch_super.size(), 0.0), lchunks(launch_super.size(), 0.0);
        std::vector<int64_t> keys;
        for (int t = 0; t < nc; ++t) {
            const SnDesc& T = S.csn[t];
            if (is_small(T)) continue;
            const int lev_t = S.level_of[t], nbr128 = ceil_div(T.r, kBigTile);
            keys.clear();
            for (int64_t u = T.upd0; u < T.upd0 + T.nupd; ++u) {
                const UpdDesc& U = S.upd[u];
                if (!is_big(U, t)) continue;
                const int lev_s = S.level_of[S.upd_src[u]];
                const int64_t launch = (int64_t)lev_s * 2 + (lev_s == lev_t - 1 ? 0 : 1);
                groups.clear();
                for (int k = 0; k < U.m;) {
                    const int win = (U.rel < 0 ? k : S.relpos[(size_t)U.rel + k] - T.rbias) / kBigTile;
                    int k1 = k + 1;
                    if (U.rel < 0) k1 = std::min(U.m, (win + 1) * kBigTile);
                    else while (k1 < U.m && (S.relpos[(size_t)U.rel + k1] - T.rbias) / kBigTile == win) ++k1;
                    groups.push_back(Group{win, k, k1 - k});
                    k = k1;
                }
                const double chunks = ceil_div(U.K, 16);
                for (const Group& gc : groups) {
                    if (gc.first >= U.n1) break;
                    const int nj = std::min(gc.len, U.n1 - gc.first);
                    for (const Group& gr : groups) {
                        if (gr.win < gc.win) continue;
                        keys.push_back((launch << 40) | ((int64_t)gc.win * nbr128 + gr.win));
                        lfrag[(size_t)launch] += chunks * ceil_div(gr.len, 16) * ceil_div(nj, 16);
                        lchunks[(size_t)launch] += chunks;
                    }
                }
            }
            std::sort(keys.begin(), keys.end());
            keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
            for (int64_t k : keys) ltasks[(size_t)(k >> 40)]++;
        }
        for (size_t l = 0; l < launch_super.size(); ++l)
            if (ltasks[l] >= super_min_tasks && lfrag[l] < super_max_fill * 64.0 * lchunks[l]) launch_super[l] = (int8_t)super_hi;
    }
    S.big_super_r = env_sr;
    S.big_super_c = env_sc;
    // tiles per level: a level with few of them leaves the device empty while its longest early stream runs (ex15-class:
    // 2 - 12 workgroups for 47 - 76 us per level, the critical path of the upper half of the tree) -- there the streams
    // are cut finer (kSplitFew*)
    std::vector<int64_t> level_tiles((size_t)std::max(S.cnlevels, 1), 0);
    if (!S.solve_only)
        for (int t = 0; t < nc; ++t)
            if (!is_small(S.csn[t])) {
                const int nbc = ceil_div(S.csn[t].w, kTile), nbr = ceil_div(S.csn[t].r, kTile);
                level_tiles[(size_t)S.level_of[t]] += (int64_t)nbc * nbr - (int64_t)nbc * (nbc - 1) / 2;
            }
    for (int t = 0; t < nc; ++t) {
        SnDesc& T = S.csn[t];
        big_tile0[t + 1] = big_tile0[t];
        if (S.solve_only) continue;
        if (is_small(T)) {
            S.n_small++;
            continue;
        }
        S.n_big++;
        const int nbc = ceil_div(T.w, kTile), nbr = ceil_div(T.r, kTile);
        if (S.n_tflags + (int64_t)nbc * nbr > 0x7fffffffLL) throw std::runtime_error("schedule: too many tiles");
        T.tflag0 = (int32_t)S.n_tflags;
        S.n_tflags += (int64_t)nbc * nbr;
        for (int jb = 1; jb < nbc; ++jb) {
            const double K = (double)jb * kTile, wb = std::min(kTile, T.w - jb * kTile);
            S.inner_flops += K * wb * (wb + 1) + 2.0 * K * (double)(T.r - jb * kTile - wb) * wb;
        }
        const int lev_t = S.level_of[t];
        auto rel_at = [&](const UpdDesc& U, int k) { return U.rel < 0 ? k : S.relpos[(size_t)U.rel + k] - T.rbias; };
        // key of a list: ((J * nbr + I) * 2 + phase) * 4 + wave
        const size_t nkeys = (size_t)nbc * nbr * 8;
        auto for_each_entry = [&](auto&& fn) {
            for (int64_t u = T.upd0; u < T.upd0 + T.nupd; ++u) {
                const UpdDesc& U = S.upd[u];
                if (is_big(U, t)) continue;
                // early: the descendant is complete before the level below the target even starts
                const int phase = (S.level_of[S.upd_src[u]] <= lev_t - 2) ? 0 : 1;
                groups.clear();
                for (int k = 0; k < U.m;) {
                    const int win = rel_at(U, k) / kSub;
                    int k1 = k + 1;
                    while (k1 < U.m && rel_at(U, k1) / kSub == win) ++k1;
                    groups.push_back(Group{win, k, k1 - k});
                    k = k1;
                }
                for (const Group& gc : groups) {
                    if (gc.first >= U.n1) break;
                    const int nj = std::min(gc.len, U.n1 - gc.first);
                    for (const Group& gr : groups) {
                        if (gr.win < gc.win) continue;
                        const int I = gr.win / 2, J = gc.win / 2;
                        const size_t key = (((size_t)J * nbr + I) * 2 + phase) * 4 + (gr.win & 1) * 2 + (gc.win & 1);
                        fn(key, U, gr.first, gr.len, gc.first, nj);
                    }
                }
            }
        };
        cursor.assign(nkeys + 1, 0);
        for_each_entry([&](size_t key, const UpdDesc&, int, int, int, int) { cursor[key + 1]++; });
        const int64_t base = (int64_t)S.wave_entries.size();
        for (size_t k = 0; k < nkeys; ++k) cursor[k + 1] += cursor[k];
        S.sn_wp0[t] = (int64_t)S.wave_ptr.size();
        for (size_t k = 0; k <= nkeys; ++k) S.wave_ptr.push_back(base + cursor[k]);
        S.wave_entries.resize((size_t)(base + cursor[nkeys]));
        for_each_entry([&](size_t key, const UpdDesc& U, int ia, int mi, int ja, int nj) {
            S.wave_entries[(size_t)(base + cursor[key]++)] =
                WaveEntry{U.src, (int32_t)U.rel, U.ld, U.K, ia, ja, mi | (nj << 8)};
        });
        // ---- BIG: the wide descendants, cut along the 128-row windows of the target's panel
        const int nbr128 = ceil_div(T.r, kBigTile), nbc128 = ceil_div(T.w, kBigTile);
        big_tile0[t + 1] = big_tile0[t] + (int64_t)nbr128 * nbc128;
        for (int64_t u = T.upd0; u < T.upd0 + T.nupd; ++u) {
            const UpdDesc& U = S.upd[u];
            const double f = (double)U.K * U.n1 * (U.n1 + 1) + 2.0 * U.K * (double)(U.m - U.n1) * U.n1;
            if (!is_big(U, t)) {
                S.tile_update_flops += f;
                continue;
            }
            S.big_flops += f;
            const int lev_s = S.level_of[S.upd_src[u]];
            if (lev_s >= lev_t) throw std::runtime_error("schedule: an update source is not below its target");
            const int64_t launch = (int64_t)lev_s * 2 + (lev_s == lev_t - 1 ? 0 : 1);
            const int sr = env_sr ? env_sr : launch_super[(size_t)launch], sc = env_sr ? env_sc : launch_super[(size_t)launch];
            // A task owns a SUPER-TILE of sr x sc 128 x 128 tiles of the target.  The source's rows inside the
            // super-tile's row window (a run of its panel rows) times its rows inside the column window are cut into
            // 128 x 128 blocks IN THE SOURCE'S ROW ORDER -- full blocks but for the last of a run -- one entry each:
            // a 128-row window of the target holds only about 64 rows of a typical source (a face of a sub-box on a
            // separator plane), so per-tile windows were two-thirds ragged (tools/big_stats.py).  The blocks of one
            // source touch different entries of L; the sources stay in update order.
            auto make_groups = [&](int win_rows, std::vector<Group>& out) {
                out.clear();
                for (int k = 0; k < U.m;) {
                    const int win = rel_at(U, k) / win_rows;
                    int k1 = k + 1;
                    if (U.rel < 0) k1 = std::min(U.m, (win + 1) * win_rows);
                    else while (k1 < U.m && rel_at(U, k1) / win_rows == win) ++k1;
                    out.push_back(Group{win, k, k1 - k});
                    k = k1;
                }
            };
            make_groups(kBigTile * sr, groups);
            if (sc == sr) cgroups = groups;
            else make_groups(kBigTile * sc, cgroups);
            for (const Group& gc : cgroups) {
                if (gc.first >= U.n1) break;
                const int njt = std::min(gc.len, U.n1 - gc.first);
                for (const Group& gr : groups) {
                    // (the target's rows ascend with the source's: a window strictly above the diagonal holds nothing)
                    if ((int64_t)(gr.win + 1) * sr * kBigTile - 1 < (int64_t)gc.win * sc * kBigTile) continue;
                    const int64_t tile = big_tile0[t] + (int64_t)gc.win * sc * nbr128 + (int64_t)gr.win * sr;
                    for (int cb = 0; cb < njt; cb += kBigTile)
                        for (int rb = 0; rb < gr.len; rb += kBigTile) {
                            const int mi = std::min(kBigTile, gr.len - rb), nj = std::min(kBigTile, njt - cb);
                            const int ia = gr.first + rb, ja = gc.first + cb;
                            if (ia + mi - 1 < ja) continue;   // block strictly above the diagonal
                            bigk.push_back(BigKeyed{(launch << 40) | tile,
                                                    WaveEntry{U.src, (int32_t)std::max<int64_t>(U.rel, 0), U.ld, U.K, ia, ja,
                                                              mi | (nj << 8) | ((U.rel < 0) << 16)},
                                                    (int16_t)sr, (int16_t)sc});
                        }
                }
            }
        }
        // weight of every tile = 16-wide k chunks of its longest wave stream
        S.sn_tw0[t] = (int64_t)S.tile_w.size();
        S.tile_w.resize(S.tile_w.size() + (size_t)nbc * nbr * 2, 0);
        S.tile_split.resize(S.tile_w.size() / 2, -1);
        int32_t* tw = &S.tile_w[S.sn_tw0[t]];
        int64_t* tsplit = &S.tile_split[S.sn_tw0[t] / 2];
        const int64_t* wp = &S.wave_ptr[S.sn_wp0[t]];
        for (size_t tp = 0; tp < (size_t)nbc * nbr * 2; ++tp) {
            int64_t wchunks[4];
            for (int q = 0; q < 4; ++q) {
                int64_t chunks = 0;
                for (int64_t e = wp[tp * 4 + q]; e < wp[tp * 4 + q + 1]; ++e)
                    chunks += ceil_div(S.wave_entries[(size_t)e].K, 16);
                wchunks[q] = chunks;
                tw[tp] = std::max<int32_t>(tw[tp], (int32_t)std::min<int64_t>(chunks, INT32_MAX));
            }
            // A launch ends with its longest wave stream: long EARLY streams are cut into parts that
            // separate workgroups apply to partial tiles (summed, in a fixed order, when the chain
            // launch loads the tile).  Every wave's list is cut where its own chunk count reaches p/nparts.
            const bool few = level_tiles[(size_t)lev_t] <= kSplitFewTiles && !split_env;
            if ((tp & 1) == 0 && tw[tp] > (few ? kSplitFewChunks : split_chunks)) {
                const int nparts = few ? std::min<int>(kSplitFewMaxParts, ceil_div(tw[tp], kSplitFewTarget))
                                       : std::min<int>(split_max, ceil_div(tw[tp], split_target));
                Schedule::SplitDesc sd{(int64_t)S.split_ranges.size(), S.n_split_doubles, nparts, 0};
                S.n_split_doubles += (int64_t)(nparts - 1) * kTile * kTile;
                S.split_ranges.resize(S.split_ranges.size() + (size_t)nparts * 8, 0);
                int64_t* rg = &S.split_ranges[(size_t)sd.ranges];
                for (int q = 0; q < 4; ++q) {
                    int64_t e = wp[tp * 4 + q], done = 0;
                    for (int part = 0; part < nparts; ++part) {
                        rg[part * 8 + 2 * q] = e;
                        const int64_t goal = wchunks[q] * (part + 1) / nparts;
                        while (e < wp[tp * 4 + q + 1] && (part == nparts - 1 || done < goal))
                            done += ceil_div(S.wave_entries[(size_t)e++].K, 16);
                        rg[part * 8 + 2 * q + 1] = e;
                    }
                }
                tsplit[tp / 2] = (int64_t)S.split_desc.size();
                S.split_desc.push_back(sd);
            }
        }
    }
    // ---- BIG tasks: one per (launch, tile), entries in update order (stable sort)
    if (!bigk.empty()) {
        std::stable_sort(bigk.begin(), bigk.end(),
                         [](const BigKeyed& a, const BigKeyed& b) { return a.launch_tile < b.launch_tile; });
        // Dense entries: full 128 x 128 blocks of the source's rows (on or below the target's diagonal; a block that
        // straddles it -- the diagonal blocks of the pushes between a separator's pieces -- is multiplied whole and its
        // upper part dropped when the tile is updated: 1.4 % more products there, and no ragged launch of a few hundred
        // diagonal blocks behind every such push).  A launch hands them to k_chol_dense when
        // they carry at least kDenseMinShare of its products; decided per launch over ALL targets, so that the order of
        // sums does not depend on which pieces a rank owns.
        auto is_full = [](const WaveEntry& E) {
            const int mi = E.mn & 255, nj = (E.mn >> 8) & 255;
            return mi == kBigTile && nj == kBigTile && E.K >= kDenseChunk;
        };
        const int dense_mode = env_int("PARSY_BIG_DENSE", 1);
        std::vector<double> lprod(launch_super.size(), 0.0), ldense(launch_super.size(), 0.0);
        auto products_of = [](const WaveEntry& E) {
            const int mi = E.mn & 255, nj = (E.mn >> 8) & 255;
            return (double)ceil_div(E.K, 4) * ceil_div(mi, 16) * ceil_div(nj, 16);
        };
        for (const BigKeyed& b : bigk) {
            const size_t l = (size_t)(b.launch_tile >> 40);
            lprod[l] += products_of(b.e);
            if (is_full(b.e)) ldense[l] += products_of(b.e);
        }
        // per launch: 0 = k_chol_big only; 1 = full blocks to k_chol_dense, the rest to k_chol_big; 2 = everything to
        // k_chol_dense (it multiplies a ragged window as a whole block and drops what the window does not have) --
        // where the ragged rest is so small (the last, partial row block of the pushes between a separator's pieces: a
        // handful of tasks) that a launch of its own would cost more than the wasted products: measured 0.07 - 0.1 ms
        // per such launch, 110 of them per Flan-class factorization
        std::vector<uint8_t> launch_dense(launch_super.size(), 0);
        for (size_t l = 0; l < launch_dense.size(); ++l) {
            if (dense_mode == 4) {   // (tests: every entry of every launch through k_chol_dense, ragged windows and all)
                launch_dense[l] = lprod[l] > 0 ? 2 : 0;
                continue;
            }
            if (dense_mode == 0 || ldense[l] <= 0) continue;
            if (dense_mode >= 2 || ldense[l] >= kDenseMinShare * lprod[l]) launch_dense[l] = 1;
            if (launch_dense[l] && lprod[l] - ldense[l] <= kDenseAllShare * lprod[l] && dense_mode != 3) launch_dense[l] = 2;
        }
        // (PARSY_DENSE_MIN_SHARE / PARSY_DENSE_ALL_SHARE / PARSY_DENSE_FILL, percent: the thresholds, diagnostics)
        const double min_share = env_int("PARSY_DENSE_MIN_SHARE", (int)(kDenseMinShare * 100 + 0.5)) / 100.0;
        const double all_share = env_int("PARSY_DENSE_ALL_SHARE", (int)(kDenseAllShare * 100 + 0.5)) / 100.0;
        const int fill_min = env_int("PARSY_DENSE_FILL", (int)(kDenseMinFill * 100 + 0.5)) * kBigTile * kBigTile / 100;
        for (size_t l = 0; l < launch_dense.size(); ++l) {   // (re-decide with the thresholds of the environment)
            if (dense_mode == 4 || dense_mode == 0 || ldense[l] <= 0) continue;
            launch_dense[l] = (dense_mode >= 2 || ldense[l] >= min_share * lprod[l]) ? 1 : 0;
            if (launch_dense[l] && lprod[l] - ldense[l] <= all_share * lprod[l] && dense_mode != 3) launch_dense[l] = 2;
        }
        // mode 1: full blocks, and windows filled well enough that a whole-block product beats the ragged kernel's rate
        auto is_dense = [&](const WaveEntry& E, int mode) {
            if (mode == 2) return E.K >= kDenseChunk;
            const int mi = E.mn & 255, nj = (E.mn >> 8) & 255;
            return mode == 1 && E.K >= kDenseChunk && mi * nj >= fill_min;
        };
        S.big_entries.resize(bigk.size());
        const bool dense_strips = env_int("PARSY_DENSE_STRIPS", 1) != 0;
        std::map<std::pair<int64_t, int64_t>, size_t> strip_of;   // (source, ia << 32 | ja) of a task's full dense blocks
        std::vector<uint8_t> absorbed;
        int t = 0;
        for (size_t i = 0; i < bigk.size();) {
            size_t j = i;
            while (j < bigk.size() && bigk[j].launch_tile == bigk[i].launch_tile) ++j;
            const int64_t launch = bigk[i].launch_tile >> 40, tile = bigk[i].launch_tile & ((1LL << 40) - 1);
            // dense entries first, each part in update order
            size_t at = i;
            int64_t weight = 0, dweight = 0, dchunks = 0;
            const int lmode = launch_dense[(size_t)launch];
            if (lmode)
                for (size_t q = i; q < j; ++q)
                    if (is_dense(bigk[q].e, lmode)) {
                        S.big_entries[at++] = bigk[q].e;
                        dchunks += ceil_div(bigk[q].e.K, kDenseChunk);
                        dweight += ceil_div(bigk[q].e.K, 16) + 2;
                        {   // (the pairs (i, j), i >= j, of the block: a block that straddles the diagonal has fewer)
                            const WaveEntry& E = bigk[q].e;
                            const int mi = E.mn & 255, nj = (E.mn >> 8) & 255;
                            double pairs = 0;
                            for (int jj = E.ja; jj < E.ja + nj; ++jj) pairs += std::max(0, E.ia + mi - std::max(E.ia, jj));
                            S.dense_flops += 2.0 * E.K * pairs;
                        }
                    }
            const size_t mid = at;
            // Strips: the remainder of a source's row run right behind a full block (<= kStripMax rows x the block's 128
            // columns) or of its column run beside it (the block's 128 rows x <= kStripMax columns) rides with the block
            // -- k_chol_dense has that block's operands staged and fetches 16 rows more per chunk, where k_chol_big
            // staged the 128-wide window again for a sliver of products (Flan-class: 62 000 such entries, a third of the
            // bytes the ragged launches staged; tools/pair_stats.py).  PARSY_DENSE_STRIPS=0: never.
            int32_t nstrips = 0;
            absorbed.assign(j - i, 0);
            if (lmode && dense_strips && mid > i) {
                strip_of.clear();
                for (size_t d = i; d < mid; ++d) {
                    const WaveEntry& E = S.big_entries[d];
                    if ((E.mn & 0xffff) == (kBigTile | (kBigTile << 8))) strip_of[{E.src, ((int64_t)E.ia << 32) | (uint32_t)E.ja}] = d;
                }
                for (size_t q = i; q < j && !strip_of.empty(); ++q) {
                    const WaveEntry& R = bigk[q].e;
                    if (is_dense(R, lmode)) continue;
                    const int mi = R.mn & 255, nj = (R.mn >> 8) & 255;
                    const bool row_strip = mi <= kStripMax && nj == kBigTile, col_strip = nj <= kStripMax && mi == kBigTile;
                    if (!row_strip && !col_strip) continue;
                    const auto it = strip_of.find({R.src, ((int64_t)(R.ia - (row_strip ? kBigTile : 0)) << 32) |
                                                          (uint32_t)(R.ja - (col_strip ? kBigTile : 0))});
                    if (it == strip_of.end()) continue;
                    WaveEntry& E = S.big_entries[it->second];
                    if (E.K != R.K || ((E.mn >> 16) & 1) != ((R.mn >> 16) & 1) || E.rel != R.rel || E.ld != R.ld) continue;
                    if ((row_strip ? big_strip_rows(E) : big_strip_cols(E)) != 0) continue;
                    if (big_strip_rows(E) == 0 && big_strip_cols(E) == 0) ++nstrips;
                    E.mn |= row_strip ? mi << 17 : nj << 22;
                    absorbed[q - i] = 1;
                    {
                        double pairs = 0;
                        for (int jj = R.ja; jj < R.ja + nj; ++jj) pairs += std::max(0, R.ia + mi - std::max(R.ia, jj));
                        S.dense_flops += 2.0 * R.K * pairs;
                    }
                    dweight += ceil_div(R.K, 64);
                }
            }
            for (size_t q = i; q < j; ++q)
                if (!is_dense(bigk[q].e, lmode) && !absorbed[q - i]) {
                    S.big_entries[at++] = bigk[q].e;
                    weight += ceil_div(bigk[q].e.K, 16) + 4;
                }
            const size_t jend = at;     // (entries that ride with a dense block leave unused slots up to j)
            S.n_dense_entries += (int64_t)(mid - i);
            S.n_strip_entries += nstrips;
            while (t + 1 < nc && big_tile0[t + 1] <= tile) ++t;   // tiles ascend within a launch ...
            if (tile < big_tile0[t]) {                             // ... and start over with the next one
                t = 0;
                while (big_tile0[t + 1] <= tile) ++t;
            }
            const int nbr128 = ceil_div(S.csn[t].r, kBigTile);
            const int64_t local = tile - big_tile0[t];
            S.big_all.push_back(Schedule::BigTask{t, (int32_t)(local % nbr128) * kBigTile,
                                                  (int32_t)(local / nbr128) * kBigTile,
                                                  (int32_t)std::min<int64_t>(weight, INT32_MAX), (int64_t)i, (int64_t)jend,
                                                  (int32_t)(launch >> 1), (int32_t)((launch & 1) == 0), bigk[i].sr, bigk[i].sc,
                                                  (int64_t)mid, (int32_t)std::min<int64_t>(dweight, INT32_MAX),
                                                  (int32_t)std::min<int64_t>(dchunks, INT32_MAX), nstrips});
            i = j;
        }
    }

    build_launches(S, active);
}

static void build_solve_one(Schedule& S, bool sharded);

// The lists of the solves' subtree launches for many right-hand sides (schedule.hpp: SubMember, SubTree, SubTier).
// append_sub_tier lays out one tier from its trees (supernodes in index order: descendants first).  A row below a member
// that is a column of a member of the same tree maps to that column's slot, any other row to the tree's list of outside
// rows (for a whole subtree: the rows below its root).
// Column slots are a STACK: a member's columns start where its parent's end (the root's at 0), so the slots in use at
// any time are those of a path to the root -- the members are walked in postorder, and everything a walk has pending
// (forward: sums for members not yet solved; backward: x of the members already solved that a later one reads) belongs
// to the ancestors of the member in hand.  Siblings share slot numbers; the forward kernel hands a slot back as zero
// when it reads it.  Both walks are replayed here and the tier is not made if a slot would be shared at the same time.
static bool append_sub_tier(Schedule& S, const std::vector<std::vector<int32_t>>& trees, int top_level, int max_slots) {
    const size_t members0 = S.sub_members.size(), trees0 = S.sub_trees.size(), slots0 = S.sub_slots.size(),
                 outs0 = S.sub_out_rows.size();
    auto drop = [&] {
        S.sub_members.resize(members0);
        S.sub_trees.resize(trees0);
        S.sub_slots.resize(slots0);
        S.sub_out_rows.resize(outs0);
        return false;
    };
    if (trees.empty()) return false;
    std::vector<int32_t> col_slot((size_t)S.n, -1);   // per column: its slot in the tree being laid out
    std::vector<int32_t> out_slot((size_t)S.n, -1);   // per row: its outside slot, likewise
    std::vector<int32_t> root_slot((size_t)S.nsuper, -1);  // per supernode of the tree: slot of its first column
    std::vector<int32_t> owner;                       // replay: the column a slot belongs to (-1: free)
    SubTier tier{(int32_t)trees0, 0, 0, top_level};
    for (const std::vector<int32_t>& tree : trees) {
        SubTree T{};
        T.m0 = (int32_t)S.sub_members.size();
        T.out0 = (int32_t)S.sub_out_rows.size();
        // slots: parents before children
        for (size_t q = tree.size(); q-- > 0;) {
            const int32_t t = tree[q];
            const SnDesc& D = S.sn[t];
            const int p = S.sparent[(size_t)t];
            const int32_t off = (p >= 0 && root_slot[(size_t)p] >= 0) ? root_slot[(size_t)p] + S.sn[p].w : 0;
            root_slot[(size_t)t] = off;
            for (int c = 0; c < D.w; ++c) col_slot[(size_t)D.c0 + c] = off + c;
            T.ncols = std::max(T.ncols, off + D.w);
        }
        // outside rows in ascending order
        std::vector<int32_t> outs;
        for (int32_t t : tree) {
            const SnDesc& D = S.sn[t];
            for (int k = D.w; k < D.r; ++k) {
                const int32_t row = S.rows[(size_t)D.pi + k];
                if (col_slot[(size_t)row] < 0 && out_slot[(size_t)row] < 0) {
                    out_slot[(size_t)row] = 0;
                    outs.push_back(row);
                }
            }
        }
        std::sort(outs.begin(), outs.end());
        T.nout = (int32_t)outs.size();
        for (int32_t j = 0; j < T.nout; ++j) out_slot[(size_t)outs[(size_t)j]] = T.ncols + j;
        S.sub_out_rows.insert(S.sub_out_rows.end(), outs.begin(), outs.end());
        const int32_t trash = T.ncols + T.nout;
        bool ok = trash + 1 <= max_slots && trash + 1 <= 65535;
        // members: a supernode's 16-column blocks from left to right (row k of block j = row 16 j + k of the panel)
        auto row_id = [&](const SnDesc& D, int k) { return k < D.w ? D.c0 + k : S.rows[(size_t)D.pi + k]; };
        auto slot_of = [&](int32_t row) { return col_slot[(size_t)row] >= 0 ? col_slot[(size_t)row] : out_slot[(size_t)row]; };
        for (size_t q = 0; q < tree.size() && ok; ++q) {
            const SnDesc& D = S.sn[tree[q]];
            for (int j0 = 0; j0 < D.w; j0 += kTinyWidth) {
                const int wb = std::min(kTinyWidth, D.w - j0);
                SubMember M{D.px + (int64_t)j0 * D.r + j0, D.c0 + j0, wb, D.r - j0, root_slot[(size_t)tree[q]] + j0,
                            (int32_t)(S.sub_slots.size() / 16), D.r};
                for (int k0 = wb; k0 < M.r; k0 += 16)
                    for (int kq = 0; kq < 4; ++kq)
                        for (int v = 0; v < 4; ++v) {
                            const int k = k0 + 4 * v + kq;
                            S.sub_slots.push_back((uint16_t)(k < M.r ? slot_of(row_id(D, j0 + k)) : trash));
                        }
                S.sub_members.push_back(M);
            }
        }
        T.m1 = (int32_t)S.sub_members.size();
        // replay of the forward walk: a column slot is taken by the first sum that goes to it and freed when its member
        // is solved; of the backward walk: written when its member is solved, read by the members after it
        owner.assign((size_t)T.ncols, -1);
        for (size_t q = 0; q < tree.size() && ok; ++q) {
            const SnDesc& D = S.sn[tree[q]];
            for (int c = 0; c < D.w && ok; ++c) {
                int32_t& o = owner[(size_t)col_slot[(size_t)D.c0 + c]];
                ok = o < 0 || o == D.c0 + c;
                o = -1;
            }
            for (int k = D.w; k < D.r && ok; ++k) {
                const int32_t row = S.rows[(size_t)D.pi + k];
                if (col_slot[(size_t)row] < 0) continue;
                int32_t& o = owner[(size_t)col_slot[(size_t)row]];
                ok = o < 0 || o == row;
                o = row;
            }
        }
        for (int32_t o : owner) ok = ok && o < 0;   // (every sum was consumed: its column's member came later)
        owner.assign((size_t)T.ncols, -1);
        for (size_t q = tree.size(); q-- > 0 && ok;) {
            const SnDesc& D = S.sn[tree[q]];
            for (int k = D.w; k < D.r && ok; ++k) {
                const int32_t row = S.rows[(size_t)D.pi + k];
                if (col_slot[(size_t)row] >= 0) ok = owner[(size_t)col_slot[(size_t)row]] == row;
            }
            for (int c = 0; c < D.w; ++c) owner[(size_t)col_slot[(size_t)D.c0 + c]] = D.c0 + c;
        }
        for (int32_t t : tree) {
            const SnDesc& D = S.sn[t];
            for (int c = 0; c < D.w; ++c) col_slot[(size_t)D.c0 + c] = -1;
            root_slot[(size_t)t] = -1;
        }
        for (int32_t row : outs) out_slot[(size_t)row] = -1;
        if (!ok) return drop();
        tier.max_slots = std::max(tier.max_slots, trash + 1);
        S.sub_trees.push_back(T);
    }
    tier.ntrees = (int32_t)(S.sub_trees.size() - trees0);
    S.sub_tiers.push_back(tier);
    S.sub_max_slots = std::max(S.sub_max_slots, tier.max_slots);
    return true;
}

// Tier 0: the subtrees of the forward solve's subtree launch as solve_small_list / solve_small_ranges lay them out.
// Tiers above: bands of levels whose active supernodes are all at most kSubTierMaxWidth wide, every supernode not yet in a
// tier in the tree of its highest ancestor inside the band; a band grows while that leaves at least kSubTierMinTrees
// trees (PARSY_SUB_TIER_MIN_TREES; 0: no bands) whose slots fit.
static void build_sub_tiers(Schedule& S) {
    S.sub_members.clear();
    S.sub_trees.clear();
    S.sub_slots.clear();
    S.sub_out_rows.clear();
    S.sub_tiers.clear();
    S.sub_max_slots = 0;
    S.sub_cover_level = -1;
    const int ns = S.nsuper;
    std::vector<uint8_t> covered((size_t)ns, 0);
    {
        std::vector<std::vector<int32_t>> trees(S.solve_small_ranges.size() / 2);
        for (size_t b = 0; b < trees.size(); ++b)
            trees[b].assign(S.solve_small_list.begin() + S.solve_small_ranges[2 * b],
                            S.solve_small_list.begin() + S.solve_small_ranges[2 * b + 1]);
        if (trees.empty() || !append_sub_tier(S, trees, -1, kSubMaxSlots + 1)) return;
        for (const auto& tr : trees)
            for (int32_t t : tr) covered[(size_t)t] = 1;
    }
    const int min_trees = env_int("PARSY_SUB_TIER_MIN_TREES", kSubTierMinTrees);
    if (min_trees <= 0) return;
    // Bands pay where the level launches they replace are chains of latencies -- factors of few entries per row (2-D
    // problems: parabolic_fem-class, 67 entries per row: levels 3-6 of 64 right-hand sides 307 -> 133 us forward).  The
    // trees of a 3-D factor carry hundreds of outside rows each (Flan-class: 442 slots = 63 KB of LDS per wave, two waves
    // per compute unit: 4.4 ms for what the level kernels do in 2.1), so there only the subtrees keep this form (3.7 ->
    // 2.1 ms).  PARSY_SUB_TIER_MIN_TREES set: whatever the factor (tests).
    if (std::getenv("PARSY_SUB_TIER_MIN_TREES") == nullptr && S.xsize >= (int64_t)kSubTierMaxDensity * S.n) return;
    std::vector<int> level((size_t)ns, 0);
    for (int l = 0; l < S.nlevels; ++l)
        for (int q = S.levelPtr[l]; q < S.levelPtr[l + 1]; ++q) level[(size_t)S.levelSet[q]] = l;
    // bands may reach up to the level below the first one with a wide (or, shards: with an inactive) supernode
    int lmax = S.nlevels - 1;
    for (int t = 0; t < ns; ++t)
        if (!covered[(size_t)t] && (S.sn[t].w > kSubTierMaxWidth || !S.active[t])) lmax = std::min(lmax, level[(size_t)t] - 1);
    // (a band must also cover every level below it completely: nothing uncovered may lie under a tier-0 subtree's level)
    auto trees_of_band = [&](int top, std::vector<std::vector<int32_t>>& trees) {
        std::vector<int32_t> root_of((size_t)ns, -1), tree_of((size_t)ns, -1);
        trees.clear();
        for (int t = ns - 1; t >= 0; --t) {   // parents first
            if (covered[(size_t)t] || level[(size_t)t] > top) continue;
            const int p = S.sparent[(size_t)t];
            if (p >= 0 && tree_of[(size_t)p] >= 0) tree_of[(size_t)t] = tree_of[(size_t)p];
            else {
                tree_of[(size_t)t] = (int32_t)trees.size();
                trees.emplace_back();
            }
        }
        for (int t = 0; t < ns; ++t)
            if (tree_of[(size_t)t] >= 0) trees[(size_t)tree_of[(size_t)t]].push_back(t);
    };
    int next = 0;   // lowest level with an uncovered supernode
    while (true) {
        next = S.nlevels;
        for (int t = 0; t < ns; ++t)
            if (!covered[(size_t)t]) next = std::min(next, level[(size_t)t]);
        if (next > lmax) break;
        // the highest top with enough trees
        // (where not even one level leaves min_trees trees -- the narrow levels right below the wide supernodes -- a last
        // band of at least a quarter as many: parabolic_fem-class, levels 7-8 as 128 trees)
        std::vector<std::vector<int32_t>> trees, best;
        int best_top = -1;
        // (the HIGHEST top that leaves enough trees: a handful of stragglers on the lowest uncovered level -- wider supernodes
        // among the leaves -- are few trees by themselves and many once the level above joins them)
        for (int need : {min_trees, std::max(min_trees / 4, 1)}) {
            for (int top = next; top <= lmax; ++top) {
                trees_of_band(top, trees);
                if ((int)trees.size() < need) continue;
                best.swap(trees);
                best_top = top;
            }
            if (best_top >= 0) break;
        }
        if (best_top < 0) break;
        // heaviest trees first
        std::vector<double> cost(best.size(), 0.0);
        for (size_t b = 0; b < best.size(); ++b)
            for (int32_t t : best[b]) cost[b] += 2e3 + (double)S.sn[t].w * S.sn[t].r;
        std::vector<size_t> order(best.size());
        for (size_t b = 0; b < order.size(); ++b) order[b] = b;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
        std::vector<std::vector<int32_t>> sorted(best.size());
        for (size_t b = 0; b < order.size(); ++b) sorted[b].swap(best[order[b]]);
        bool ok = append_sub_tier(S, sorted, best_top, kSubTierMaxSlots + 1);
        while (!ok && best_top > next) {   // (slots: a lower band)
            --best_top;
            trees_of_band(best_top, sorted);
            ok = append_sub_tier(S, sorted, best_top, kSubTierMaxSlots + 1);
        }
        if (!ok) break;
        for (const auto& tr : sorted)
            for (int32_t t : tr) covered[(size_t)t] = 1;
        S.sub_cover_level = best_top;
    }
}

void build_launches(Schedule& S, const uint8_t* active, const uint8_t* active_pieces) {
    const int ns = S.nsuper;
    // PARSY_FORCE_UNFUSED=1 schedules the solve's fallback form everywhere (per-block-column
    // launches): the path taken when a chain launch would not be resident.  Used by the tests.
    const char* fu = std::getenv("PARSY_FORCE_UNFUSED");
    const bool force_unfused = fu && fu[0] == '1';
    // chain launches take their work by ticket (producers first): any number of workgroups is fine
    const int max_chain = force_unfused ? -1 : INT_MAX;
    S.n_solve_chain_launches = 0;
    S.active.assign(ns, 1);
    if (active) S.active.assign(active, active + ns);
    // the factorization goes by PIECES of the Cholesky view (a multi-device run deals the pieces of a split
    // supernode to different devices); without a piece mask a piece is active with its supernode
    const int npieces = (int)S.csn.size();
    S.active_piece.assign((size_t)npieces, 1);
    for (int t = 0; t < npieces; ++t)
        S.active_piece[t] = active_pieces ? (active_pieces[t] != 0) : S.active[S.csn_real[t]];
    // A subtree launch of the solves walks its subtree without waiting for anything outside it.  Under a supernode mask
    // (a rank's share) that holds only for the subtrees the mask contains whole: the active members of the others go to
    // the level launches, which a solve in steps of levels (plan_solve_levels) separates by its exchange steps.
    if (S.solve_subtree_all.empty()) {
        S.solve_subtree_all = S.solve_subtree;
        S.bsolve_subtree_all = S.bsolve_subtree;
    }
    S.solve_subtree = S.solve_subtree_all;
    S.bsolve_subtree = S.bsolve_subtree_all;
    if (active)
        for (auto* sub : {&S.solve_subtree, &S.bsolve_subtree}) {
            const int count = sub == &S.solve_subtree ? S.n_solve_subtrees : S.n_bsolve_subtrees;
            std::vector<uint8_t> whole((size_t)std::max(count, 0), 1);
            for (int t = 0; t < ns; ++t)
                if ((*sub)[t] >= 0 && !S.active[t]) whole[(size_t)(*sub)[t]] = 0;
            for (int t = 0; t < ns; ++t)
                if ((*sub)[t] >= 0 && !whole[(size_t)(*sub)[t]]) (*sub)[t] = -1;
        }
    S.small_list.clear();
    S.tiles.clear();
    S.n_chain_launches = 0;
    S.chol.clear();
    S.solve_small_list.clear();
    S.solve_panels.clear();
    S.solve_mtasks.clear();
    S.solve_fix_list.clear();
    S.solve_wide_list.clear();
    S.solve.clear();
    S.n_solve_wide = 0;

    // Side-stream launches -- TILES(lev): the early wave streams of level lev's tiles, PUSH(s): the BIG
    // updates from level s's supernodes into targets at levels >= s + 2 -- are enqueued right before the
    // main-stream launches of the level in between (lev - 1 = s + 1), so that they run while the main
    // stream works through that level's chain; Launch::level = lev = s + 2 is the level whose main-stream
    // launches wait for them.
    S.big_tasks.clear();
    std::vector<Launch> early_launches;
    std::vector<size_t> level_begin;  // index in S.chol where each level's launches start
    std::vector<int> bigs, sbigs;
    // BIG tasks of every (source level, kind), active targets only, heaviest first
    std::vector<std::vector<const Schedule::BigTask*>> big_next(S.cnlevels), big_push(S.cnlevels);
    for (const Schedule::BigTask& b : S.big_all)
        if (S.active_piece[b.sn]) (b.next ? big_next : big_push)[b.src_level].push_back(&b);
    // Order of a BIG launch's tasks.  Workgroups are dealt round-robin over the 8 XCDs (block b runs on the
    // XCD of every block b + 8k; observed placement, used for speed only) and each XCD has its own L2, so
    // tiles that read the same rows of a source should share an XCD at the same time: the tasks are grouped
    // into g x g super-tiles of one target's tile grid (g row windows + g column windows of the source feed
    // g*g tasks), the groups are dealt to 8 sequences -- heaviest group first, each to the least loaded
    // sequence -- and sequence x fills the positions 8s + x (shorter sequences are padded with empty tasks,
    // which return at once).  Small launches keep the plain heaviest-first order.  PARSY_BIG_GROUP=g
    // (0: never group).
    const int big_group = env_int("PARSY_BIG_GROUP", kBigGroup);
    // (dense: the tasks' dense parts for k_chol_dense -- TileDesc::part = its 8-wide k chunks --, else their ragged rests)
    auto emit_big = [&](const std::vector<const Schedule::BigTask*>& all, Launch L, bool dense) {
        std::vector<const Schedule::BigTask*> v;
        for (const Schedule::BigTask* b : all)
            if (dense ? b->em > b->e0 : b->e1 > b->em) v.push_back(b);
        if (v.empty()) return L;
        L.first = (int32_t)S.big_tasks.size();
        auto wt = [dense](const Schedule::BigTask* b) { return dense ? b->dweight : b->weight; };
        auto desc = [dense](const Schedule::BigTask* b) {
            return dense ? TileDesc{b->sn, b->row0, b->col0, b->dchunks | (b->strips > 0 ? kDenseStripTask : 0), b->e0, b->em}
                         : TileDesc{b->sn, b->row0, b->col0, 0, b->em, b->e1};
        };
        auto by_weight = [&](const Schedule::BigTask* a, const Schedule::BigTask* b) { return wt(a) > wt(b); };
        // group edge: as large as leaves every XCD at least kBigGroupsPerXcd groups to balance with
        int g = big_group;
        // (a group of g x g tiles holds g * g / (sr * sc) tasks)
        const int per_task = v[0]->sr * v[0]->sc;
        while (g > 1 && (int64_t)v.size() * per_task < (int64_t)8 * kBigGroupsPerXcd * g * g) g /= 2;
        if (g < std::max(v[0]->sr, v[0]->sc)) g = 1;
        if (g <= 1) {
            std::stable_sort(v.begin(), v.end(), by_weight);
            for (const Schedule::BigTask* b : v) S.big_tasks.push_back(desc(b));
            L.count = (int32_t)v.size();
            return L;
        }
        struct Group { int64_t key, weight; int32_t maxw; std::vector<const Schedule::BigTask*> tasks; };
        std::vector<const Schedule::BigTask*> seq[8];
        int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto least = [&] {
            int x = 0;
            for (int q = 1; q < 8; ++q)
                if (load[q] < load[x]) x = q;
            return x;
        };
        // the bulk goes in g x g groups; the lightest kBigTailGroups groups are dealt again at half the edge
        // (and so on down to single tasks), so that the sequences end together
        std::vector<const Schedule::BigTask*> rest(v);
        for (; !rest.empty(); g /= 2) {
            if (g <= 1) {
                std::stable_sort(rest.begin(), rest.end(), by_weight);
                for (const Schedule::BigTask* b : rest) {
                    const int x = least();
                    seq[x].push_back(b);
                    load[x] += wt(b);
                }
                break;
            }
            std::vector<std::pair<int64_t, const Schedule::BigTask*>> keyed;
            keyed.reserve(rest.size());
            for (const Schedule::BigTask* b : rest) {
                const int64_t gi = b->row0 / (kBigTile * g), gj = b->col0 / (kBigTile * g);
                keyed.push_back({((int64_t)b->sn << 32) | (gj << 16) | gi, b});
            }
            std::stable_sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            std::vector<Group> groups;
            for (const auto& kb : keyed) {
                if (groups.empty() || groups.back().key != kb.first) groups.push_back(Group{kb.first, 0, 0, {}});
                Group& G = groups.back();
                G.tasks.push_back(kb.second);
                G.weight += wt(kb.second);
                G.maxw = std::max(G.maxw, wt(kb.second));
            }
            std::stable_sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) {
                return a.maxw != b.maxw ? a.maxw > b.maxw : a.weight > b.weight;
            });
            const size_t bulk = groups.size() > (size_t)kBigTailGroups ? groups.size() - kBigTailGroups : 0;
            rest.clear();
            for (size_t q = 0; q < groups.size(); ++q) {
                Group& G = groups[q];
                if (q >= bulk) {
                    rest.insert(rest.end(), G.tasks.begin(), G.tasks.end());
                    continue;
                }
                const int x = least();
                std::stable_sort(G.tasks.begin(), G.tasks.end(), by_weight);
                seq[x].insert(seq[x].end(), G.tasks.begin(), G.tasks.end());
                load[x] += G.weight;
            }
        }
        size_t len = 0;
        for (int x = 0; x < 8; ++x) len = std::max(len, seq[x].size());
        for (size_t s = 0; s < len; ++s)
            for (int x = 0; x < 8; ++x) {
                if (s < seq[x].size()) {
                    S.big_tasks.push_back(desc(seq[x][s]));
                } else {
                    S.big_tasks.push_back(TileDesc{0, 0, 0, 0, 0, 0});  // padding: no entries
                }
            }
        L.count = (int32_t)(8 * len);
        return L;
    };
    // The active supernodes of every subtree in index order (descendants first), subtrees by falling cost:
    // fn(supernode) appends to the kind's list; returns the (begin, end) pairs of the list positions.
    auto subtree_ranges = [&](const std::vector<int32_t>& subtree, int count, const std::vector<double>& cost,
                              auto&& position, auto&& append, std::vector<int32_t>& ranges, bool chol = false) {
        std::vector<std::vector<int32_t>> members((size_t)count);
        std::vector<double> total((size_t)count, 0.0);
        for (int t = 0; t < ns; ++t)
            if (subtree[t] >= 0 && (chol ? S.active_piece[S.piece0[t]] : S.active[t])) {
                members[(size_t)subtree[t]].push_back(t);
                total[(size_t)subtree[t]] += cost[t];
            }
        std::vector<int32_t> order;
        for (int k = 0; k < count; ++k)
            if (!members[(size_t)k].empty()) order.push_back(k);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return total[(size_t)a] > total[(size_t)b]; });
        for (int k : order) {
            ranges.push_back(position());
            for (int32_t t : members[(size_t)k]) append(t);
            ranges.push_back(position());
        }
        return (int32_t)order.size();
    };
    S.small_ranges.clear();
    S.solve_small_ranges.clear();
    S.bsolve_ranges.clear();
    S.sub_members.clear();
    S.sub_trees.clear();
    S.sub_slots.clear();
    S.sub_out_rows.clear();
    S.sub_max_slots = 0;
    S.bsolve_blocks.clear();
    S.bsolve_pairs.clear();
    for (int lev = 0; lev < S.cnlevels && !S.solve_only; ++lev) {
        level_begin.push_back(S.chol.size());
        bigs.clear();
        // ---- Cholesky -------------------------------------------------------------
        if (lev == 0 && S.n_chol_subtrees > 0) {
            // the subtrees of SMALL supernodes, one workgroup each: LDS and stage sized for the largest member
            Launch L{kLaunchSmall, 0, 0, 0, 0, 0, 2, 0, -1, 0};
            L.count = subtree_ranges(
                S.chol_subtree, S.n_chol_subtrees, S.chol_cost, [&] { return (int32_t)S.small_list.size(); },
                [&](int32_t t) {
                    const SnDesc& T = S.csn[S.piece0[t]];
                    S.small_list.push_back(S.piece0[t]);
                    L.lds_bytes = std::max<int32_t>(L.lds_bytes, T.w * T.r * (int)sizeof(double));
                    for (int64_t u = T.upd0; u < T.upd0 + T.nupd; ++u)
                        L.jb = std::max<int32_t>(L.jb, std::min<int64_t>((int64_t)S.upd[u].m * S.upd[u].K, 2048));
                },
                S.small_ranges, true);
            if (L.count > 0) S.chol.push_back(L);
        }
        {
            Launch L{kLaunchSmall, (int32_t)S.small_list.size(), 0, lev, 0, 0, 0, 0, -1, 0};
            for (int q = S.clevelPtr[lev]; q < S.clevelPtr[lev + 1]; ++q) {
                const int t = S.clevelSet[q];
                if (!S.active_piece[t] || S.chol_subtree[S.csn_real[t]] >= 0) continue;
                const SnDesc& T = S.csn[t];
                if (is_small(T)) {
                    S.small_list.push_back(t);
                    L.lds_bytes = std::max<int32_t>(L.lds_bytes, T.w * T.r * (int)sizeof(double));
                    // jb carries the stage size of the SMALL launch: the largest descendant block,
                    // capped (kernel: kSmallStage = 2048 doubles)
                    for (int64_t u = T.upd0; u < T.upd0 + T.nupd; ++u)
                        L.jb = std::max<int32_t>(L.jb, std::min<int64_t>((int64_t)S.upd[u].m * S.upd[u].K, 2048));
                } else {
                    bigs.push_back(t);
                }
            }
            L.count = (int32_t)S.small_list.size() - L.first;
            if (L.count > 0) S.chol.push_back(L);
        }
        // ---- PUSH(lev): this level's wide supernodes update everything at least two levels up (side
        // stream, once the level is complete); NEXT(lev - 1): the level below updates this level's tiles
        if (lev + 2 < S.cnlevels) {
            Launch Ld = emit_big(big_push[lev], Launch{kLaunchDense, 0, 0, lev + 2, 0, 0, 0, 1, lev, 1}, true);
            if (Ld.count > 0) early_launches.push_back(Ld);
            Launch Lp = emit_big(big_push[lev], Launch{kLaunchBig, 0, 0, lev + 2, 0, 0, 0, 1, lev, 1}, false);
            if (Lp.count > 0) early_launches.push_back(Lp);
        }
        if (lev > 0) {
            Launch Ld = emit_big(big_next[lev - 1], Launch{kLaunchDense, 0, 0, lev, 0, 0, 0, 0, -1, 0}, true);
            if (Ld.count > 0) S.chol.push_back(Ld);
            Launch Ln = emit_big(big_next[lev - 1], Launch{kLaunchBig, 0, 0, lev, 0, 0, 0, 0, -1, 0}, false);
            if (Ln.count > 0) S.chol.push_back(Ln);
        }
        if (!bigs.empty()) {
            // ---- TILES: the early part of the external updates, longest streams first ------
            {
                Launch Lt{kLaunchTiles, (int32_t)S.tiles.size(), 0, lev, 0, 0, 0, 1, lev - 2, 1};
                std::vector<std::pair<int32_t, TileDesc>> wt;  // (weight, tile)
                for (int t : bigs) {
                    const SnDesc& T = S.csn[t];
                    const int nbc = ceil_div(T.w, kTile), nbr = ceil_div(T.r, kTile);
                    const int32_t* tw = &S.tile_w[S.sn_tw0[t]];
                    for (int J = 0; J < nbc; ++J)
                        for (int I = J; I < nbr; ++I) {
                            const int32_t wgt = tw[((size_t)J * nbr + I) * 2];
                            if (wgt <= 0) continue;
                            const int64_t si = S.tile_split[S.sn_tw0[t] / 2 + (size_t)J * nbr + I];
                            if (si < 0) {
                                wt.push_back({wgt, TileDesc{t, I * kTile, J * kTile, 0,
                                                            S.sn_wp0[t] + (((int64_t)J * nbr + I) * 2) * 4, -1}});
                            } else {
                                const Schedule::SplitDesc& sd = S.split_desc[(size_t)si];
                                for (int part = 0; part < sd.nparts; ++part)
                                    wt.push_back({ceil_div(wgt, sd.nparts),
                                                  TileDesc{t, I * kTile, J * kTile, part | (sd.nparts << 8),
                                                           sd.ranges + part * 8, sd.sp}});
                            }
                        }
                }
                std::stable_sort(wt.begin(), wt.end(),
                                 [](const auto& a, const auto& b) { return a.first > b.first; });
                for (auto& x : wt) S.tiles.push_back(x.second);
                Lt.count = (int32_t)S.tiles.size() - Lt.first;
                if (Lt.count > 0) early_launches.push_back(Lt);
            }
            // ---- CHAIN: every tile of the level's tiled supernodes, producers before consumers.
            // Group J of a supernode: the walker (J = 0 only: it owns every diagonal tile), the two tiles
            // the walker needs prepared for its step J -- (J+1,J) and (J+1,J+1) -- and then the other
            // tiles of block column J, which wait for diagonal tile J.
            // On large jobs the chain takes two launches per level: the tiles of the diagonal squares with the walkers
            // first, the tiles of the rows below the squares afterwards.  In one launch those tiles -- hundreds to
            // thousands per piece of a top separator -- were dispatched long before the walker reached their block
            // column and waited for it holding a workgroup slot (72 KB of LDS: one of the two k_chol_big workgroups of
            // the PUSH launch that runs beside the chain cannot be resident on that compute unit meanwhile); in a launch
            // of their own every diagonal tile is there when they start, no flag is waited for, and the kernel of the
            // second launch (k_chol_chain_rows: no walker, no prepared tile) gets by with 52 KB of LDS: three workgroups
            // per compute unit.  Flan-class input: 371.2 -> 365.8 ms with the same kernel for both launches, 347.6 ms
            // with k_chol_chain_rows; 27-point grids 56^3 ... 80^3: -3 ... -6 %; 48^3: +-0, 40^3: +9 %, nd24k-class
            // 3.92 -> 4.21 ms (small jobs without BIG launches: the chain is the critical path), so the split is taken
            // from kChainSplitAutoFlops update flops on.  Splitting only the pieces of split supernodes loses everywhere
            // (mode 1).  PARSY_CHAIN_SPLIT=0: never; 2: always (diagnostics).
            const int chain_split_mode = env_int("PARSY_CHAIN_SPLIT", S.update_flops >= kChainSplitAutoFlops ? 2 : 0);
            auto splits = [&](int t) {
                const int real = S.csn_real[t];
                return chain_split_mode == 2 || (chain_split_mode == 1 && S.piece0[real + 1] - S.piece0[real] > 1);
            };
            bool chain_split = false;
            for (int t : bigs) chain_split = chain_split || splits(t);
            for (int part = 0; part < (chain_split ? 2 : 1); ++part) {
            Launch Lc{kLaunchChain, (int32_t)S.tiles.size(), 0, lev, S.n_chain_launches++, 0, 0, 0, -1, 0};
            Lc.fused = part;   // 1: only tiles below the squares (k_chol_chain_rows: three workgroups per compute unit)
            // Walkers stay resident for their whole chain, so the block-column-major interleaving is
            // done per batch of at most walker_batch supernodes: what a walker waits for then lies at most
            // one batch of tiles ahead of it in ticket order, and the walkers of the batches in flight
            // never take more than a fraction of the resident workgroups.
            for (size_t b0 = 0; b0 < bigs.size(); b0 += (size_t)S.walker_batch) {
              const size_t b1 = std::min(bigs.size(), b0 + (size_t)S.walker_batch);
              int maxnb = 0;
              for (size_t q = b0; q < b1; ++q) maxnb = std::max(maxnb, ceil_div(S.csn[bigs[q]].w, kTile));
              for (int J = 0; J < maxnb; ++J)
                for (size_t q = b0; q < b1; ++q) {
                    const int t = bigs[q];
                    const SnDesc& T = S.csn[t];
                    const int nbc = ceil_div(T.w, kTile), nbr = ceil_div(T.r, kTile);
                    if (J >= nbc) continue;
                    auto push = [&](int I, int Jc) {
                        if (chain_split && ((I < nbc || !splits(t)) != (part == 0))) return;   // square / rows below it
                        const int64_t si = S.tile_split[S.sn_tw0[t] / 2 + (size_t)Jc * nbr + I];
                        S.tiles.push_back(TileDesc{t, I * kTile, Jc * kTile,
                                                   si < 0 ? 0 : S.split_desc[(size_t)si].nparts,
                                                   S.sn_wp0[t] + (((int64_t)Jc * nbr + I) * 2 + 1) * 4,
                                                   si < 0 ? -1 : S.split_desc[(size_t)si].sp});
                    };
                    if (J == 0) push(0, 0);
                    const bool next_diag = J + 1 < nbc;
                    if (next_diag) {
                        push(J + 1, J);
                        push(J + 1, J + 1);
                    }
                    for (int I = next_diag ? J + 2 : J + 1; I < nbr; ++I) push(I, J);
                }
            }
            Lc.count = (int32_t)S.tiles.size() - Lc.first;
            if (Lc.count > 0) S.chol.push_back(Lc);
            else --S.n_chain_launches;
            }
        }
    }
    // splice the side launches in front of the level before the one that waits for them (stable: PUSH
    // before TILES of the same level)
    // chol_level_begin[lev]: where level lev's launches start in S.chol (its side launches first) -- the steps
    // of a factorization that is run level by level (parsy_factor_level)
    S.chol_level_begin.assign(level_begin.begin(), level_begin.end());
    if (!early_launches.empty()) {
        std::vector<Launch> merged;
        size_t e = 0;
        std::stable_sort(early_launches.begin(), early_launches.end(),
                         [](const Launch& a, const Launch& b) { return a.level < b.level; });
        for (int lev = 0; lev < S.cnlevels; ++lev) {
            S.chol_level_begin[(size_t)lev] = merged.size();
            while (e < early_launches.size() && early_launches[e].level - 1 <= lev) merged.push_back(early_launches[e++]);
            const size_t b0 = level_begin[lev], b1 = lev + 1 < S.cnlevels ? level_begin[lev + 1] : S.chol.size();
            merged.insert(merged.end(), S.chol.begin() + b0, S.chol.begin() + b1);
        }
        S.chol.swap(merged);
    }
    S.chol_level_begin.push_back(S.chol.size());
    for (int lev = 0; lev < S.nlevels; ++lev) {
        sbigs.clear();
        // ---- forward solve ----------------------------------------------------------
        if (lev == 0 && S.n_solve_subtrees > 0) {
            Launch L{kLaunchSolveSmall, 0, 0, 0, 0, 0, 2, 0, -1, 0};
            L.count = subtree_ranges(
                S.solve_subtree, S.n_solve_subtrees, S.solve_cost, [&] { return (int32_t)S.solve_small_list.size(); },
                [&](int32_t t) {
                    S.solve_small_list.push_back(t);
                    L.jb = std::max<int32_t>(L.jb, S.sn[t].w);
                },
                S.solve_small_ranges);
            if (L.count > 0) S.solve.push_back(L);
        }
        {
            // supernodes of one block column: the tiny ones (width <= kTinyWidth: one wave each) in a launch of
            // their own when there are enough of them, the others one workgroup each
            // (forward: the second width class of the one-wave kernel measured no faster than the workgroup kernel)
            std::vector<int32_t> tiny, narrow;
            for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
                const int t = S.levelSet[q];
                if (!S.active[t] || S.solve_subtree[t] >= 0) continue;
                if (S.sn[t].w <= kTinyWidth) tiny.push_back(t);
                else if (S.sn[t].w <= kTile) narrow.push_back(t);
                else sbigs.push_back(t);
            }
            if (!narrow.empty() && tiny.size() < 64) {
                narrow.insert(narrow.end(), tiny.begin(), tiny.end());
                tiny.clear();
            }
            for (const std::vector<int32_t>* group : {&tiny, &narrow}) {
                if (group->empty()) continue;
                Launch L{kLaunchSolveSmall, (int32_t)S.solve_small_list.size(), (int32_t)group->size(), lev, 0, 0, 0, 0, -1, 0};
                for (int32_t t : *group) {
                    S.solve_small_list.push_back(t);
                    L.jb = std::max<int32_t>(L.jb, S.sn[t].w);  // widest supernode of the launch
                }
                S.solve.push_back(L);
            }
        }
        if (!sbigs.empty()) {
            S.n_solve_wide += (int)sbigs.size();
            int chain_wgs = 0;
            for (int t : sbigs) chain_wgs += ceil_div(S.sn[t].r, kSolveRows);
            if (chain_wgs <= max_chain) {
                // one launch: every 256-row chunk of every wide supernode of the level
                Launch Lc{kLaunchSolvePanel, (int32_t)S.solve_panels.size(), 0, lev, S.n_solve_chain_launches++, 0, 1, 0, -1, 0};
                for (int t : sbigs)
                    for (int c = 0; c * kSolveRows < S.sn[t].r; ++c)
                        S.solve_panels.push_back(PanelDesc{t, c, c * kSolveRows, 0});
                Lc.count = (int32_t)S.solve_panels.size() - Lc.first;
                // the same launch for many right-hand sides (k_solve_blocks_mrhs): the block columns of the triangles
                // first (producers, ascending), then the row chunks below the supernodes' own columns
                Lc.lds_bytes = (int32_t)S.solve_mtasks.size();
                for (int t : sbigs)
                    for (int jb = 0; jb * kTile < S.sn[t].w; ++jb) S.solve_mtasks.push_back(PanelDesc{t, jb, -1, 0});
                for (int t : sbigs)
                    for (int c = 0, row0 = S.sn[t].w; row0 < S.sn[t].r; ++c, row0 += kSolveRowsMrhs)
                        S.solve_mtasks.push_back(PanelDesc{t, c, row0, 0});
                Lc.wait_level = (int32_t)S.solve_mtasks.size() - Lc.lds_bytes;
                S.solve.push_back(Lc);
                for (int t : sbigs) {
                    for (int jb = 0; jb * kTile < S.sn[t].w; ++jb) {
                        S.solve_wide_list.push_back(t);
                        S.solve_wide_list.push_back(jb);
                    }
                }
            } else {
                int maxnb = 0;
                for (int t : sbigs) maxnb = std::max(maxnb, ceil_div(S.sn[t].w, kTile));
                for (int jb = 0; jb < maxnb; ++jb) {
                    Launch Lp{kLaunchSolvePanel, (int32_t)S.solve_panels.size(), 0, lev, jb, 0, 0, 0, -1, 0};
                    for (int t : sbigs) {
                        const SnDesc& T = S.sn[t];
                        if (ceil_div(T.w, kTile) <= jb) continue;
                        const int wb = std::min(kTile, T.w - jb * kTile);
                        S.solve_panels.push_back(PanelDesc{t, jb, -1, 0});
                        for (int row0 = jb * kTile + wb; row0 < T.r; row0 += kSolveRows)
                            S.solve_panels.push_back(PanelDesc{t, jb, row0, 0});
                    }
                    Lp.count = (int32_t)S.solve_panels.size() - Lp.first;
                    S.solve.push_back(Lp);
                }
                S.solve_fix_list.insert(S.solve_fix_list.end(), sbigs.begin(), sbigs.end());
            }
        }
    }
    build_sub_tiers(S);
    // ---- backward solve: root level first.  Per level one chain launch for the block columns of
    // the wide supernodes (last block column first: block jb waits for the published x of blocks
    // jb+1.. of its supernode; every workgroup of the launch must be resident) and one launch for the
    // supernodes of a single block; when the chain would not be resident, one launch per block-column
    // index from the last one down.
    S.bsolve.clear();
    S.bsolve_below.clear();
    S.n_bpart_slots = 0;
    // PARSY_BSOLVE_BELOW=0: never split the rows below off; 2: for every wide supernode with rows below (tests)
    const int below_mode = env_int("PARSY_BSOLVE_BELOW", 1);
    auto in_level_launches = [&](int t) { return S.active[t] && S.bsolve_subtree[t] < 0; };
    // the supernodes of one block column of a level: the tiny ones (one wave each: Launch::early = 1) in a launch
    // of their own when there are enough of them, the others one workgroup each
    auto narrow_launches = [&](int lev) {
        std::vector<int32_t> tiny, tiny2, narrow;
        for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
            const int t = S.levelSet[q];
            if (!in_level_launches(t) || S.sn[t].w > kTile) continue;
            (S.sn[t].w <= kTinyWidth ? tiny : S.sn[t].w <= kTinyWidth2 ? tiny2 : narrow).push_back(t);
        }
        // (a group too small for a launch of its own joins the next wider one; the second width class of the
        // one-wave kernel pays from a few hundred supernodes on: Flan-class backward solve 7.60 -> 7.32 ms, but
        // +6 % on the nd24k-class input with an extra launch per level)
        if (tiny2.size() < 512 && !narrow.empty()) {
            narrow.insert(narrow.end(), tiny2.begin(), tiny2.end());
            tiny2.clear();
        }
        if (tiny.size() < 64 && !(tiny2.empty() && narrow.empty())) {
            (tiny2.empty() ? narrow : tiny2).insert((tiny2.empty() ? narrow : tiny2).end(), tiny.begin(), tiny.end());
            tiny.clear();
        }
        for (const std::vector<int32_t>* group : {&tiny, &tiny2, &narrow}) {
            if (group->empty()) continue;
            Launch Ln{kLaunchBackBlock, (int32_t)S.bsolve_blocks.size(), (int32_t)group->size(), lev, 0, 0, 0, 0, -1,
                      group == &tiny ? 1 : group == &tiny2 ? 2 : 0};
            for (int32_t t : *group) S.bsolve_blocks.push_back(PanelDesc{t, 0, 0, 0});
            S.bsolve.push_back(Ln);
        }
    };
    for (int lev = S.nlevels - 1; lev >= 0; --lev) {
        int maxnb = 0, wide_blocks = 0;
        for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
            const int t = S.levelSet[q];
            if (!in_level_launches(t)) continue;
            const int nbc = ceil_div(S.sn[t].w, kTile);
            maxnb = std::max(maxnb, nbc);
            if (nbc > 1) wide_blocks += nbc;
        }
        if (wide_blocks > 0 && wide_blocks <= max_chain) {
            Launch Lc{kLaunchBackBlock, (int32_t)S.bsolve_blocks.size(), 0, lev, S.n_solve_chain_launches++,
                      (int32_t)S.bsolve_pairs.size(), 1, 0, -1, 0};
            // One right-hand side: a block column's sums over the rows BELOW the supernode's own columns (x final there)
            // need no hand-off, but in the chain launch only the block column's own four waves stream them -- a level of
            // few, tall supernodes (the separators below the root: 53 workgroups for 0.9 GB of panel) ran at 1.2 TB/s.
            // There the rows below go to a launch of their own, 512-row chunks over the whole device, partial sums into
            // slots that the chain adds up in a fixed order (bitwise reproducible).
            int64_t groups_here = 0;
            for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
                const int t = S.levelSet[q];
                if (S.active[t] && S.sn[t].w > kTile) groups_here += ceil_div(ceil_div(S.sn[t].w, kTile), kBackGroup);
            }
            const bool below_level = below_mode >= 2 || (below_mode == 1 && groups_here <= kBelowMaxGroups);
            Launch Lw{kLaunchBackBelow, (int32_t)S.bsolve_below.size(), 0, lev, 0, 0, 0, 0, -1, 0};
            for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
                const int t = S.levelSet[q];
                const int nbc = ceil_div(S.sn[t].w, kTile);
                if (!S.active[t] || nbc < 2) continue;
                const int below = S.sn[t].r - S.sn[t].w;
                int32_t slot1 = 0;
                if (below_level && below >= (below_mode >= 2 ? 1 : kBelowRows) &&
                    S.n_bpart_slots + (int64_t)nbc * ceil_div(below, kBelowRows) < 0x7fffffffLL) {
                    const int nch = ceil_div(below, kBelowRows);
                    slot1 = (int32_t)S.n_bpart_slots + 1;
                    for (int jb = 0; jb < nbc; ++jb)
                        for (int c = 0; c < nch; ++c)
                            S.bsolve_below.push_back(PanelDesc{t, jb, S.sn[t].w + c * kBelowRows, (int32_t)(S.n_bpart_slots + (int64_t)jb * nch + c)});
                    S.n_bpart_slots += (int64_t)nbc * nch;
                }
                for (int jb = nbc - 1; jb >= 0; --jb) S.bsolve_blocks.push_back(PanelDesc{t, jb, 0, 0});
                for (int jb = nbc - 1; jb >= 0; jb -= kBackGroup)
                    S.bsolve_pairs.push_back(PanelDesc{t, jb, std::min(kBackGroup, jb + 1), slot1});
            }
            Lw.count = (int32_t)S.bsolve_below.size() - Lw.first;
            if (Lw.count > 0) S.bsolve.push_back(Lw);
            Lc.count = (int32_t)S.bsolve_blocks.size() - Lc.first;
            Lc.wait_level = (int32_t)S.bsolve_pairs.size() - Lc.lds_bytes;
            S.bsolve.push_back(Lc);
            narrow_launches(lev);
            continue;
        }
        for (int jb = maxnb - 1; jb >= 1; --jb) {
            Launch Lb{kLaunchBackBlock, (int32_t)S.bsolve_blocks.size(), 0, lev, jb, 0, 0, 0, -1, 0};
            for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
                const int t = S.levelSet[q];
                if (in_level_launches(t) && ceil_div(S.sn[t].w, kTile) > jb) S.bsolve_blocks.push_back(PanelDesc{t, jb, 0, 0});
            }
            Lb.count = (int32_t)S.bsolve_blocks.size() - Lb.first;
            if (Lb.count > 0) S.bsolve.push_back(Lb);
        }
        if (maxnb > 1) {   // block column 0 of the wide supernodes (their other block columns: the launches above)
            Launch Lb{kLaunchBackBlock, (int32_t)S.bsolve_blocks.size(), 0, lev, 0, 0, 0, 0, -1, 0};
            for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
                const int t = S.levelSet[q];
                if (in_level_launches(t) && ceil_div(S.sn[t].w, kTile) > 1) S.bsolve_blocks.push_back(PanelDesc{t, 0, 0, 0});
            }
            Lb.count = (int32_t)S.bsolve_blocks.size() - Lb.first;
            if (Lb.count > 0) S.bsolve.push_back(Lb);
        }
        narrow_launches(lev);
    }
    if (S.n_bsolve_subtrees > 0) {
        // the subtrees last: every ancestor outside them is final; inside, a workgroup walks from the subtree's
        // root down (the forward order reversed)
        Launch L{kLaunchBackBlock, 0, 0, 0, 0, 0, 2, 0, -1, 1};   // (tiny supernodes: one wave per subtree)
        std::vector<int32_t> members;
        L.count = subtree_ranges(
            S.bsolve_subtree, S.n_bsolve_subtrees, S.solve_cost,
            [&] {
                for (size_t q = members.size(); q-- > 0;) S.bsolve_blocks.push_back(PanelDesc{members[q], 0, 0, 0});
                members.clear();
                return (int32_t)S.bsolve_blocks.size();
            },
            [&](int32_t t) { members.push_back(t); }, S.bsolve_ranges);
        if (L.count > 0) S.bsolve.push_back(L);
    }
    if (!S.solve_fix_list.empty())
        S.solve.push_back(Launch{kLaunchSolveFixup, 0, (int32_t)S.solve_fix_list.size(), S.nlevels, 0, 0, 0, 0, -1, 0});
    build_solve_one(S, active != nullptr || active_pieces != nullptr);
}

// ONE-launch solves (schedule.hpp: Schedule::OneLists).  The lists of the launch that holds the supernodes of `member`:
// the supernodes are taken block column by block column (<= 64 columns, a window of the supernode's panel like a piece of
// the Cholesky view); what block p subtracts from the x of a row below its columns -- a later column of its supernode or
// a row of an ancestor -- goes into slot slot0[p] + (row - w) of a hand-off buffer (written once, so the data can be its
// own flag); the block that owns that row gathers its slots -- listed here per owner: (slot, column of the owner).
static void build_one_lists(const Schedule& S, const std::vector<uint8_t>& member, Schedule::OneLists& O) {
    const int ns = S.nsuper;
    O.clear();
    O.member = member;
    // the blocks in ticket order: level by level, a supernode's block columns from left to right
    std::vector<int32_t> blk_of_col((size_t)S.n, -1), blk_sn;
    for (int q = 0; q < ns; ++q) {
        const int t = S.levelSet[(size_t)q];
        if (!member[(size_t)t]) continue;
        const SnDesc& T = S.sn[t];
        for (int cb = 0; cb < T.w; cb += kTile) {
            SnDesc B = T;
            B.c0 = T.c0 + cb;
            B.w = std::min(kTile, T.w - cb);
            B.r = T.r - cb;
            B.px = T.px + (int64_t)cb * T.r + cb;
            B.pi = T.pi + cb;
            B.ld = T.r;
            for (int c = B.c0; c < B.c0 + B.w; ++c) blk_of_col[(size_t)c] = (int32_t)O.sn.size();
            O.slot0.push_back(O.nslots);
            O.wleft.push_back(T.w - cb);
            O.nslots += B.r - B.w;
            O.sn.push_back(B);
            blk_sn.push_back(t);
        }
    }
    const int nb = (int)O.sn.size();
    if (O.nslots > INT32_MAX / 2) throw std::runtime_error("schedule: too many hand-off slots for a ONE-launch solve");
    // column of the row k (counted inside block p's window of the panel): a later column of the supernode, or lR
    auto col_of = [&](int p, int k) {
        const SnDesc& T = S.sn[blk_sn[(size_t)p]];
        const int kk = (O.sn[(size_t)p].c0 - T.c0) + k;   // row of the supernode's panel
        return kk < T.w ? T.c0 + kk : S.rows[(size_t)T.pi + kk];
    };
    O.pull_ptr.assign((size_t)nb + 1, 0);
    for (int p = 0; p < nb; ++p)
        for (int k = O.sn[(size_t)p].w; k < O.sn[(size_t)p].r; ++k) {
            const int owner = blk_of_col[(size_t)col_of(p, k)];
            if (owner <= p) throw std::runtime_error("schedule: a row below a block's columns is not owned by a later block of the launch");
            O.pull_ptr[(size_t)owner + 1]++;
        }
    for (int p = 0; p < nb; ++p) O.pull_ptr[(size_t)p + 1] += O.pull_ptr[(size_t)p];
    O.pull_slot.resize((size_t)O.pull_ptr[(size_t)nb]);
    O.pull_pos.resize(O.pull_slot.size());
    std::vector<int32_t> at(O.pull_ptr.begin(), O.pull_ptr.end() - 1);
    for (int p = 0; p < nb; ++p)
        for (int k = O.sn[(size_t)p].w; k < O.sn[(size_t)p].r; ++k) {
            const int col = col_of(p, k), owner = blk_of_col[(size_t)col];
            const int32_t e = at[(size_t)owner]++;
            O.pull_slot[(size_t)e] = (int32_t)(O.slot0[(size_t)p] + (k - O.sn[(size_t)p].w));
            O.pull_pos[(size_t)e] = col - O.sn[(size_t)owner].c0;
        }
}

// Which solves go into ONE launch (schedule.hpp).  PARSY_SOLVE_ONE=0: never, 1: by size, 2: whatever the size (tests).
static void build_solve_one(Schedule& S, bool sharded) {
    const int ns = S.nsuper;
    S.solve_one = S.solve_one_back = S.one_subtrees = S.one_forced = S.one_big = false;
    S.one_f.clear();
    S.one_b.clear();
    const int mode = env_int("PARSY_SOLVE_ONE", 1);
    if (mode == 0 || sharded || ns == 0 || (int)S.levelSet.size() != ns) return;   // (a rank's share of the supernodes: level launches)
    // with subtree launches: the supernodes outside them (the launch must be the first of `solve` / the last of `bsolve`)
    const bool sub = S.n_solve_subtrees > 0 || S.n_bsolve_subtrees > 0;
    std::vector<uint8_t> mf((size_t)ns, 1), mb((size_t)ns, 1);
    int64_t nf = ns, nbk = ns;
    if (sub) {
        nf = nbk = 0;
        for (int t = 0; t < ns; ++t) {
            mf[(size_t)t] = S.solve_subtree[t] < 0;
            mb[(size_t)t] = S.bsolve_subtree[t] < 0;
            nf += mf[(size_t)t];
            nbk += mb[(size_t)t];
        }
        int nfused_f = 0, nfused_b = 0;
        for (const Launch& l : S.solve) nfused_f += l.fused == 2 && l.kind == kLaunchSolveSmall;
        for (const Launch& l : S.bsolve) nfused_b += l.fused == 2 && l.kind == kLaunchBackBlock;
        const bool first_ok = nfused_f == (S.n_solve_subtrees > 0) && (nfused_f == 0 || S.solve.front().fused == 2);
        const bool last_ok = nfused_b == (S.n_bsolve_subtrees > 0) && (nfused_b == 0 || S.bsolve.back().fused == 2);
        if (!first_ok || !last_ok || nf == 0 || nbk == 0) return;
    }
    auto fits = [&](int64_t members) {
        return S.xsize <= kOneMaxEntries && members <= 2 * kOneMaxSupernodes &&
               (members <= kOneMaxSupernodes || S.xsize >= kOneLargeEntries * (int64_t)ns);
    };
    // (the backward solve also above the subtree launch of a much larger plan, one right-hand side at a time: Flan-class, 15 000
    // supernodes outside it, 5.76 -> 4.55 ms.  The forward solve there: 4.27 or 4.65 ms from one process to the next against
    // 4.60 - 4.64 of the level launches, and 560 MB of hand-off slots: it stays with the level launches)
    const bool ok_f = mode == 2 || fits(nf);
    const bool ok_b = mode == 2 || fits(nbk) || (sub && nbk <= kOneMaxSupernodesBig);
    S.one_big = mode != 2 && !fits(nbk);
    if (!ok_f && !ok_b) return;
    // (a pattern whose rows are not owned by later members -- not an etree's -- or too many slots: the plan keeps its
    // level launches instead of failing as a whole)
    try {
        if (ok_f) build_one_lists(S, mf, S.one_f);
        if (ok_b && (!ok_f || mb != mf)) build_one_lists(S, mb, S.one_b);
    } catch (const std::runtime_error&) {
        S.one_f.clear();
        S.one_b.clear();
        S.one_big = false;
        return;
    }
    S.one_subtrees = sub;
    S.one_forced = mode == 2;
    S.solve_one = ok_f;
    S.solve_one_back = ok_b;
}

int64_t simulate_chain(const Schedule& S, int slots) {
    if (slots < 1) slots = 1;
    std::vector<uint8_t> fin((size_t)std::max<int64_t>(S.n_tflags, 1), 0), prep(fin.size(), 0);
    struct Task {
        int32_t sn, I, J, role;  // role 0 walker, 1 prepared tile left of a diagonal tile, 2 prepared diagonal tile, 3 other
        int32_t k;               // walker: current step; others: block columns already seen published
    };
    int64_t stuck = 0;
    for (const Launch& L : S.chol) {
        if (L.kind != kLaunchChain) continue;
        std::vector<Task> resident;
        int64_t next = 0;
        const int64_t count = L.count;
        bool progress = true;
        while (progress && (next < count || !resident.empty())) {
            progress = false;
            while ((int)resident.size() < slots && next < count) {
                const TileDesc& td = S.tiles[(size_t)L.first + (size_t)next++];
                const SnDesc& D = S.csn[td.sn];
                const int nbc = ceil_div(D.w, kTile);
                Task t{td.sn, td.row0 / kTile, td.col0 / kTile, 3, 0};
                if (t.I == 0 && t.J == 0) t.role = 0;
                else if (t.I == t.J) t.role = 2;
                else if (t.I == t.J + 1 && t.I < nbc) t.role = 1;
                if (L.fused != 0 && t.I < nbc) ++stuck;   // a rows launch holds no tile of a diagonal square
                resident.push_back(t);
                progress = true;
            }
            for (size_t q = 0; q < resident.size();) {
                Task& t = resident[q];
                const SnDesc& D = S.csn[t.sn];
                const int nbc = ceil_div(D.w, kTile);
                auto flag = [&](int I, int J) { return (size_t)D.tflag0 + (size_t)I * nbc + J; };
                bool done = false;
                if (t.role == 0) {
                    for (;;) {  // step k: diagonal tile k is published; then the two prepared tiles are needed
                        fin[flag(t.k, t.k)] = 1;
                        if (t.k + 1 >= nbc) {
                            done = true;
                            break;
                        }
                        if (!prep[flag(t.k + 1, t.k)] || !prep[flag(t.k + 1, t.k + 1)]) break;
                        fin[flag(t.k + 1, t.k)] = 1;
                        ++t.k;
                        progress = true;
                    }
                } else {
                    const int need = t.role == 2 ? t.J - 1 : t.J;
                    while (t.k < need && fin[flag(t.I, t.k)] && fin[flag(t.J, t.k)]) ++t.k;
                    if (t.k >= need) {
                        if (t.role == 3) {
                            if (fin[flag(t.J, t.J)]) {
                                fin[flag(t.I, t.J)] = 1;
                                done = true;
                            }
                        } else {
                            prep[flag(t.I, t.J)] = 1;
                            done = true;
                        }
                    }
                }
                if (done) {
                    resident[q] = resident.back();
                    resident.pop_back();
                    progress = true;
                } else {
                    ++q;
                }
            }
        }
        stuck += (count - next) + (int64_t)resident.size();
    }
    return stuck;
}

// The launches of both solves: every active supernode of one block column is solved exactly once in each direction
// (subtree runs in index order forward, reversed backward; a member's ancestors inside the run), every 256-row
// chunk / block column of an active wide supernode appears once, the backward chain's groups cover a supernode's
// block columns from the last one down, a level's launches come after (forward) / before (backward) the levels
// below it.
static void check_solve_launches(const Schedule& S, const std::function<void(const std::string&)>& fail) {
    const int ns = S.nsuper;
    std::vector<int> level_of(ns, 0);
    for (int l = 0; l < S.nlevels; ++l)
        for (int q = S.levelPtr[l]; q < S.levelPtr[l + 1]; ++q) level_of[S.levelSet[q]] = l;
    // ---- forward
    {
        std::vector<int> seen(ns, 0);
        std::vector<int64_t> chunks(ns, 0);
        int last_level = -1;
        for (const Launch& l : S.solve) {
            if (l.kind == kLaunchSolveSmall) {
                if (l.fused == 2) {
                    for (int b = l.first; b < l.first + l.count; ++b) {
                        const int32_t q0 = S.solve_small_ranges[2 * (size_t)b], q1 = S.solve_small_ranges[2 * (size_t)b + 1];
                        if (q0 < 0 || q1 <= q0 || q1 > (int32_t)S.solve_small_list.size()) {
                            fail("forward subtree launch: bad range " + std::to_string(b));
                            continue;
                        }
                        for (int32_t q = q0; q < q1; ++q) {
                            const int t = S.solve_small_list[q];
                            seen[t]++;
                            if (q > q0 && t <= S.solve_small_list[q - 1]) fail("forward subtree run is not in index order");
                            if (S.sn[t].w > l.jb) fail("forward subtree launch: supernode wider than the launch's width class");
                        }
                    }
                } else {
                    if (l.level < last_level) fail("forward solve launches go down a level");
                    last_level = l.level;
                    for (int q = l.first; q < l.first + l.count; ++q) {
                        const int t = S.solve_small_list[q];
                        seen[t]++;
                        if (level_of[t] != l.level) fail("forward launch holds a supernode of another level");
                        if (S.sn[t].w > l.jb || S.sn[t].w > kTile) fail("forward launch: supernode wider than its width class");
                    }
                }
            } else if (l.kind == kLaunchSolvePanel) {
                if (l.level < last_level) fail("forward solve launches go down a level");
                last_level = l.level;
                for (int q = l.first; q < l.first + l.count; ++q) {
                    const PanelDesc& pd = S.solve_panels[q];
                    if (pd.sn < 0 || pd.sn >= ns || S.sn[pd.sn].w <= kTile || level_of[pd.sn] != l.level)
                        fail("forward chain launch: bad chunk descriptor " + std::to_string(q));
                    else if (l.fused) {
                        if (pd.row0 != pd.jb * kSolveRows || pd.row0 >= S.sn[pd.sn].r) fail("forward chain launch: bad chunk rows");
                        chunks[pd.sn]++;
                    }
                }
                if (l.fused) {
                    // the same launch for many right-hand sides: per supernode its block columns in ascending order BEFORE
                    // anything that waits for them (tickets go in list order), every row below its columns exactly once
                    if (l.lds_bytes < 0 || l.wait_level < 0 || (size_t)l.lds_bytes + (size_t)l.wait_level > S.solve_mtasks.size())
                        fail("forward chain launch: bad task range (many right-hand sides)");
                    else {
                        std::vector<int> next_block(ns, 0);
                        std::vector<int64_t> next_row(ns, -1);
                        for (int q = l.lds_bytes; q < l.lds_bytes + l.wait_level; ++q) {
                            const PanelDesc& pd = S.solve_mtasks[(size_t)q];
                            if (pd.sn < 0 || pd.sn >= ns || S.sn[pd.sn].w <= kTile || level_of[pd.sn] != l.level) {
                                fail("forward chain launch: bad task " + std::to_string(q));
                                continue;
                            }
                            const SnDesc& T = S.sn[pd.sn];
                            if (pd.row0 < 0) {
                                if (pd.jb != next_block[pd.sn]++ || pd.jb * kTile >= T.w) fail("forward chain launch: block columns out of order");
                                if (next_row[pd.sn] >= 0) fail("forward chain launch: a block column after the row chunks that wait for it");
                            } else {
                                if (next_block[pd.sn] != ceil_div(T.w, kTile)) fail("forward chain launch: row chunk before the supernode's block columns");
                                if (next_row[pd.sn] < 0) next_row[pd.sn] = T.w;
                                if (pd.row0 != next_row[pd.sn] || pd.row0 >= T.r) fail("forward chain launch: row chunks do not tile the rows below");
                                next_row[pd.sn] = pd.row0 + kSolveRowsMrhs;
                            }
                        }
                        for (int q = l.first; q < l.first + l.count; ++q) {
                            const int t = S.solve_panels[q].sn;
                            if (t < 0 || t >= ns) continue;
                            const SnDesc& T = S.sn[t];
                            if (next_block[t] != ceil_div(T.w, kTile) || (T.r > T.w && next_row[t] < T.r) || (T.r == T.w && next_row[t] >= 0))
                                fail("forward chain launch: supernode " + std::to_string(t) + " is not covered by the many-right-hand-side tasks");
                        }
                    }
                }
            }
        }
        for (int t = 0; t < ns; ++t) {
            const bool narrow = S.sn[t].w <= kTile;
            if (narrow && seen[t] != (S.active[t] ? 1 : 0))
                fail("supernode " + std::to_string(t) + " is in " + std::to_string(seen[t]) + " forward launches");
            if (!narrow && S.solve_fix_list.empty() && chunks[t] != (S.active[t] ? ceil_div(S.sn[t].r, kSolveRows) : 0))
                fail("wide supernode " + std::to_string(t) + ": " + std::to_string(chunks[t]) + " forward chunks");
        }
    }
    // ---- backward
    {
        std::vector<int> seen(ns, 0);
        std::vector<int64_t> blocks(ns, 0), grouped(ns, 0);
        {
            // k_bsolve_below: every slot written exactly once, by the task of its (supernode, block column, chunk); a
            // chain group that names slots finds all of its supernode's there, and the launch that fills them comes first
            std::vector<uint8_t> slot_seen((size_t)S.n_bpart_slots, 0);
            for (const PanelDesc& pd : S.bsolve_below) {
                const SnDesc& T = S.sn[(size_t)pd.sn];
                if (pd.pad < 0 || pd.pad >= S.n_bpart_slots || slot_seen[(size_t)pd.pad]++) fail("backward solve: a partial-sum slot is written twice or out of range");
                if (pd.row0 < T.w || pd.row0 >= T.r || (pd.row0 - T.w) % kBelowRows || pd.jb < 0 || pd.jb * kTile >= T.w)
                    fail("backward solve: a k_bsolve_below task has a bad chunk");
            }
            for (uint8_t v : slot_seen)
                if (v != 1) fail("backward solve: a partial-sum slot is never written");
            int64_t below_done = 0;
            for (const Launch& l : S.bsolve) {
                if (l.kind == kLaunchBackBelow) below_done += l.count;
                if (l.kind != kLaunchBackBlock || l.fused != 1) continue;
                for (int g = l.lds_bytes; g < l.lds_bytes + l.wait_level; ++g) {
                    const PanelDesc& pd = S.bsolve_pairs[(size_t)g];
                    if (pd.pad <= 0) continue;
                    const SnDesc& T = S.sn[(size_t)pd.sn];
                    const int nch = ceil_div(T.r - T.w, kBelowRows), nbc = ceil_div(T.w, kTile);
                    const int64_t last = (int64_t)(pd.pad - 1) + (int64_t)nbc * nch;
                    if (T.r <= T.w || last > S.n_bpart_slots) fail("backward solve: a chain group names partial sums that do not exist");
                    // the tasks of this supernode are the slots [pad - 1, last): all before this launch
                    if (last > 0 && below_done < (int64_t)S.bsolve_below.size()) {
                        bool found = false;
                        for (int64_t q = 0; q < below_done && !found; ++q) found = S.bsolve_below[(size_t)q].pad == last - 1;
                        if (!found) fail("backward solve: a chain launch runs before the launch that forms its partial sums");
                    }
                }
            }
        }
        for (const Launch& l : S.bsolve) {
            if (l.kind != kLaunchBackBlock) continue;
            if (l.fused == 2) {
                for (int b = l.first; b < l.first + l.count; ++b) {
                    const int32_t q0 = S.bsolve_ranges[2 * (size_t)b], q1 = S.bsolve_ranges[2 * (size_t)b + 1];
                    if (q0 < 0 || q1 <= q0 || q1 > (int32_t)S.bsolve_blocks.size()) {
                        fail("backward subtree launch: bad range " + std::to_string(b));
                        continue;
                    }
                    for (int32_t q = q0; q < q1; ++q) {
                        const int t = S.bsolve_blocks[q].sn;
                        seen[t]++;
                        if (q > q0 && t >= S.bsolve_blocks[q - 1].sn) fail("backward subtree run is not in reverse index order");
                        if (S.sn[t].w > kTinyWidth) fail("backward subtree launch: supernode wider than the one-wave kernel");
                    }
                }
                continue;
            }
            for (int q = l.first; q < l.first + l.count; ++q) {
                const PanelDesc& pd = S.bsolve_blocks[q];
                if (pd.sn < 0 || pd.sn >= ns || pd.jb < 0 || pd.jb * kTile >= S.sn[pd.sn].w || level_of[pd.sn] != l.level) {
                    fail("backward launch: bad block descriptor " + std::to_string(q));
                    continue;
                }
                if (S.sn[pd.sn].w <= kTile) {
                    seen[pd.sn]++;
                    if (l.early == 1 && S.sn[pd.sn].w > kTinyWidth) fail("backward launch: supernode wider than its width class");
                    if (l.early == 2 && S.sn[pd.sn].w > kTinyWidth2) fail("backward launch: supernode wider than its width class");
                } else {
                    blocks[pd.sn]++;
                }
            }
            if (l.fused == 1) {
                // the groups of the one-right-hand-side kernel: the same block columns, last one first
                int prev_sn = -1, next_jb = -1;
                for (int g = l.lds_bytes; g < l.lds_bytes + l.wait_level; ++g) {
                    const PanelDesc& pd = S.bsolve_pairs[(size_t)g];
                    if (pd.sn != prev_sn) {
                        if (prev_sn >= 0 && next_jb != -1) fail("backward groups of supernode " + std::to_string(prev_sn) + " stop early");
                        prev_sn = pd.sn;
                        next_jb = ceil_div(S.sn[pd.sn].w, kTile) - 1;
                    }
                    if (pd.jb != next_jb || pd.row0 < 1 || pd.row0 > kBackGroup || pd.row0 > pd.jb + 1)
                        fail("backward group " + std::to_string(g) + " does not continue its supernode's block columns");
                    grouped[pd.sn] += pd.row0;
                    next_jb = pd.jb - pd.row0;
                }
                if (prev_sn >= 0 && next_jb != -1) fail("backward groups of supernode " + std::to_string(prev_sn) + " stop early");
            }
        }
        for (int t = 0; t < ns; ++t) {
            const int nbc = ceil_div(S.sn[t].w, kTile);
            if (nbc == 1 && seen[t] != (S.active[t] ? 1 : 0))
                fail("supernode " + std::to_string(t) + " is in " + std::to_string(seen[t]) + " backward launches");
            if (nbc > 1 && blocks[t] != (S.active[t] ? nbc : 0))
                fail("wide supernode " + std::to_string(t) + ": " + std::to_string(blocks[t]) + " backward blocks");
            if (nbc > 1 && S.active[t] && !S.bsolve_pairs.empty() && grouped[t] != nbc)
                fail("wide supernode " + std::to_string(t) + ": groups cover " + std::to_string(grouped[t]) + " block columns");
        }
    }
}

// The ONE-launch solves: the members are all supernodes or -- with subtree launches -- the ones outside the direction's
// subtree launch; the blocks tile the members in ticket order; every row below a block's columns has a slot of its own and
// is gathered exactly once, by the block that owns its column, at the right column, from a block with an earlier ticket
// (the backward solve takes the same blocks in reverse order).
template <class Fail>
static void check_one_lists(const Schedule& S, const Schedule::OneLists& O, const std::vector<int32_t>* subtree, const char* dir,
                            Fail&& fail) {
    const int ns = S.nsuper, nb = (int)O.sn.size();
    const std::string who = std::string("one-launch ") + dir + " solve: ";
    if ((int)O.member.size() != ns || (int)O.pull_ptr.size() != nb + 1 || (int)O.slot0.size() != nb || (int)O.wleft.size() != nb ||
        O.pull_slot.size() != O.pull_pos.size() || (int64_t)O.pull_slot.size() != O.nslots ||
        O.pull_ptr[(size_t)nb] != (int32_t)O.pull_slot.size()) {
        fail(who + "lists of the wrong length");
        return;
    }
    for (int t = 0; t < ns; ++t)
        if ((O.member[(size_t)t] != 0) != (subtree ? (*subtree)[(size_t)t] < 0 : true))
            fail(who + "supernode " + std::to_string(t) + " is on the wrong side of the subtree launch");
    std::vector<int32_t> blk_of_col((size_t)S.n, -1), slot_blk((size_t)O.nslots, -1), slot_col((size_t)O.nslots, -1);
    {
        int p = 0;
        int64_t slots = 0;
        for (int q = 0; q < ns; ++q) {
            const int t = S.levelSet[(size_t)q];
            if (!O.member[(size_t)t]) continue;
            const SnDesc& T = S.sn[t];
            for (int cb = 0; cb < T.w && p < nb; cb += kTile, ++p) {
                const SnDesc& B = O.sn[(size_t)p];
                const int wbk = std::min(kTile, T.w - cb);
                if (B.c0 != T.c0 + cb || B.w != wbk || B.r != T.r - cb || B.ld != T.r || B.px != T.px + (int64_t)cb * T.r + cb ||
                    B.pi != T.pi + cb || O.slot0[(size_t)p] != slots || O.wleft[(size_t)p] != T.w - cb)
                    fail(who + "block " + std::to_string(p) + " is not a block column of its supernode");
                for (int kk = cb + wbk; kk < T.r; ++kk) {
                    slot_blk[(size_t)(slots + (kk - cb - wbk))] = p;
                    slot_col[(size_t)(slots + (kk - cb - wbk))] = kk < T.w ? T.c0 + kk : S.rows[(size_t)T.pi + kk];
                }
                slots += B.r - B.w;
                for (int c = B.c0; c < B.c0 + B.w; ++c) blk_of_col[(size_t)c] = p;
            }
        }
        if (p != nb || slots != O.nslots) {
            fail(who + "the blocks do not tile the supernodes");
            return;
        }
    }
    std::vector<uint8_t> seen((size_t)O.nslots, 0);
    for (int p = 0; p < nb; ++p)
        for (int32_t e = O.pull_ptr[(size_t)p]; e < O.pull_ptr[(size_t)p + 1]; ++e) {
            const int32_t slot = O.pull_slot[(size_t)e], pos = O.pull_pos[(size_t)e];
            const bool in = slot >= 0 && slot < O.nslots;
            if (!in || seen[(size_t)slot] || pos < 0 || pos >= O.sn[(size_t)p].w || slot_col[(size_t)slot] != O.sn[(size_t)p].c0 + pos ||
                slot_blk[(size_t)slot] >= p) {
                fail(who + "bad entry " + std::to_string(e) + " in the gather list of block " + std::to_string(p));
                continue;
            }
            seen[(size_t)slot] = 1;
        }
    for (int64_t sl = 0; sl < O.nslots; ++sl)
        if (!seen[(size_t)sl]) fail(who + "slot " + std::to_string(sl) + " is never gathered");
}

template <class Fail>
static void check_solve_one(const Schedule& S, Fail&& fail) {
    if (!S.solve_one && !S.solve_one_back) return;
    if ((int)S.levelSet.size() != S.nsuper) {
        fail("one-launch solve: no level sets");
        return;
    }
    if (S.solve_one) check_one_lists(S, S.one_f, S.one_subtrees ? &S.solve_subtree : nullptr, "forward", fail);
    if (!S.solve_one_back) return;
    if (!S.one_b.sn.empty()) check_one_lists(S, S.one_b, S.one_subtrees ? &S.bsolve_subtree : nullptr, "backward", fail);
    else if (!S.solve_one) fail("one-launch backward solve: no lists");
    else if (S.one_subtrees)
        for (int t = 0; t < S.nsuper; ++t)
            if ((S.solve_subtree[t] < 0) != (S.bsolve_subtree[t] < 0)) {
                fail("one-launch backward solve: shares the forward lists although the subtree launches differ");
                break;
            }
    if (S.one_subtrees) {
        if (S.n_solve_subtrees > 0 && (S.solve.empty() || S.solve.front().fused != 2))
            fail("one-launch forward solve: the subtree launch is not the first launch");
        if (S.n_bsolve_subtrees > 0 && (S.bsolve.empty() || S.bsolve.back().fused != 2))
            fail("one-launch backward solve: the subtree launch is not the last launch");
    }
}

// The subtree / band launches of the many-right-hand-side solves (SubTier): every member a 16-column block of an active
// supernode with the panel window that goes with it; members of a tree in index order; every supernode in at most one tree,
// whole (all its blocks, left to right); the subtrees of tier 0 are those of the forward solve's subtree launch; every
// active supernode of a level <= sub_cover_level is in a tier and no launch list is needed for it; slots inside the tree's
// range, a member's own columns on the stack below its rows' targets; outside rows sorted and not columns of the tree.
template <class Fail>
static void check_sub_tiers(const Schedule& S, Fail&& fail) {
    if (S.sub_tiers.empty()) return;
    const int ns = S.nsuper;
    std::vector<int32_t> sn_of_col((size_t)S.n, -1);
    for (int t = 0; t < ns; ++t)
        for (int c = 0; c < S.sn[t].w; ++c) sn_of_col[(size_t)S.sn[t].c0 + c] = t;
    std::vector<int> level((size_t)ns, 0);
    for (int l = 0; l < S.nlevels; ++l)
        for (int q = S.levelPtr[l]; q < S.levelPtr[l + 1]; ++q) level[(size_t)S.levelSet[q]] = l;
    std::vector<int32_t> tree_of((size_t)ns, -1);
    int32_t next_tree = 0;
    for (size_t k = 0; k < S.sub_tiers.size(); ++k) {
        const SubTier& R = S.sub_tiers[k];
        if (R.tree0 != next_tree || R.ntrees <= 0 || (size_t)(R.tree0 + R.ntrees) > S.sub_trees.size())
            return fail("sub tiers: tiers do not tile the trees");
        next_tree = R.tree0 + R.ntrees;
        if (k == 0 && (size_t)R.ntrees != S.solve_small_ranges.size() / 2) return fail("sub tiers: tier 0 is not the subtree launch");
        for (int32_t b = R.tree0; b < R.tree0 + R.ntrees; ++b) {
            const SubTree& T = S.sub_trees[(size_t)b];
            if (T.m0 >= T.m1 || (size_t)T.m1 > S.sub_members.size() || T.ncols + T.nout + 1 > R.max_slots ||
                (size_t)(T.out0 + T.nout) > S.sub_out_rows.size())
                return fail("sub tiers: a tree's ranges");
            int32_t prev_sn = -1, expect_c = -1;
            for (int32_t q = T.m0; q < T.m1; ++q) {
                const SubMember& M = S.sub_members[(size_t)q];
                if (M.c0 < 0 || M.c0 >= S.n || M.w < 1 || M.w > kTinyWidth) return fail("sub tiers: a member's columns");
                const int t = sn_of_col[(size_t)M.c0];
                const SnDesc& D = S.sn[t];
                const int j0 = M.c0 - D.c0;
                if (!S.active[t] || j0 % kTinyWidth != 0 || M.w != std::min(kTinyWidth, D.w - j0) || M.r != D.r - j0 || M.ld != D.r ||
                    M.px != D.px + (int64_t)j0 * D.r + j0)
                    return fail("sub tiers: a member is not a 16-column block of an active supernode");
                if (j0 == 0) {
                    if (tree_of[(size_t)t] >= 0) return fail("sub tiers: a supernode in two trees");
                    if (t <= prev_sn) return fail("sub tiers: members out of index order");
                    if (k == 0 ? S.solve_subtree[t] < 0 : (level[(size_t)t] > R.top_level || S.sn[t].w > kSubTierMaxWidth))
                        return fail("sub tiers: a supernode that does not belong to its tier");
                    tree_of[(size_t)t] = b;
                    prev_sn = t;
                } else if (t != prev_sn || M.c0 != expect_c) {
                    return fail("sub tiers: the blocks of a supernode are not in a row");
                }
                expect_c = M.c0 + M.w;
                if (M.slot0 < 0 || M.slot0 + M.w > T.ncols) return fail("sub tiers: a member's column slots");
                const int nch = (M.r - M.w + 15) / 16;
                if ((size_t)(M.so + nch) * 16 > S.sub_slots.size()) return fail("sub tiers: a member's slot words");
                for (int ch = 0; ch < nch; ++ch)
                    for (int e = 0; e < 16; ++e) {
                        const int k2 = M.w + 16 * ch + 4 * (e & 3) + (e >> 2), slot = S.sub_slots[(size_t)(M.so + ch) * 16 + e];
                        if (k2 >= M.r) {
                            if (slot != T.ncols + T.nout) return fail("sub tiers: a padding row not in the padding slot");
                            continue;
                        }
                        const int32_t row = j0 + k2 < D.w ? D.c0 + j0 + k2 : S.rows[(size_t)D.pi + j0 + k2];
                        if (slot < T.ncols) {   // a column of a member of this tree that comes later
                            const int tt = sn_of_col[(size_t)row];
                            bool found = false;
                            for (int32_t q2 = q + 1; q2 < T.m1 && !found; ++q2) {
                                const SubMember& M2 = S.sub_members[(size_t)q2];
                                found = row >= M2.c0 && row < M2.c0 + M2.w && slot == M2.slot0 + row - M2.c0;
                            }
                            if (!found || tt < 0) return fail("sub tiers: a row's column slot does not belong to a later member");
                        } else if (slot >= T.ncols + T.nout || S.sub_out_rows[(size_t)T.out0 + slot - T.ncols] != row) {
                            return fail("sub tiers: a row's outside slot");
                        }
                    }
            }
            for (int32_t j = 0; j < T.nout; ++j) {
                const int32_t row = S.sub_out_rows[(size_t)T.out0 + j];
                if (j > 0 && row <= S.sub_out_rows[(size_t)T.out0 + j - 1]) return fail("sub tiers: outside rows not sorted");
                if (tree_of[(size_t)sn_of_col[(size_t)row]] == b) return fail("sub tiers: an outside row is a column of the tree");
            }
        }
    }
    for (int t = 0; t < ns; ++t) {
        if (S.solve_subtree[t] >= 0 && S.active[t] && tree_of[(size_t)t] < 0) return fail("sub tiers: a supernode of the subtree launch is missing");
        if (S.active[t] && level[(size_t)t] <= S.sub_cover_level && tree_of[(size_t)t] < 0) return fail("sub tiers: a covered level has a supernode outside the tiers");
    }
}

int64_t check_schedule(const Schedule& S, std::string& what) {
    int64_t bad = 0;
    auto fail = [&](const std::string& msg) {
        if (bad++ == 0) what = msg;
    };
    check_solve_launches(S, fail);
    check_solve_one(S, fail);
    check_sub_tiers(S, fail);
    if (S.solve_only) return bad;
    const int nc = (int)S.csn.size();
    // ---- pieces tile their supernode; levels respect the chain-extended etree
    for (int t = 0; t < S.nsuper; ++t) {
        int cols = 0;
        for (int p = S.piece0[t]; p < S.piece0[t + 1]; ++p) {
            const SnDesc& C = S.csn[p];
            if (C.c0 != S.sn[t].c0 + cols || C.rbias != cols || C.r != S.sn[t].r - cols || C.ld != S.sn[t].r)
                fail("piece " + std::to_string(p) + " is not a window of its supernode's panel");
            if (p > S.piece0[t] && S.level_of[p] != S.level_of[p - 1] + 1) fail("pieces of a supernode are not on consecutive levels");
            cols += C.w;
        }
        if (cols != S.sn[t].w) fail("pieces of supernode " + std::to_string(t) + " do not add up to its width");
    }
    // ---- every update comes from a lower level; wave + BIG entries cover it exactly (flop identity)
    std::vector<double> covered(S.upd.size(), 0.0);
    auto entry_flops = [](const WaveEntry& E, bool same_window) {
        const double K = E.K, mi = E.mn & 255, nj = (E.mn >> 8) & 255;
        return same_window ? K * nj * (nj + 1) + 2.0 * K * (mi - nj) * nj : 2.0 * K * mi * nj;
    };
    // a BIG entry is any block of the source's rows x rows: the pairs (i, j), i >= j, of it
    auto block_flops = [](const WaveEntry& E) {
        const int mi = E.mn & 255, nj = (E.mn >> 8) & 255;
        auto pairs_of = [](int ia, int m, int ja, int n) {
            double pairs = 0;
            for (int j = ja; j < ja + n; ++j) pairs += std::max(0, ia + m - std::max(ia, j));
            return pairs;
        };
        // (+ the strips that ride with a dense block: rows behind it x its columns, its rows x columns beside it)
        return 2.0 * E.K * (pairs_of(E.ia, mi, E.ja, nj) + pairs_of(E.ia + mi, big_strip_rows(E), E.ja, nj) +
                            pairs_of(E.ia, mi, E.ja + nj, big_strip_cols(E)));
    };
    for (int t = 0; t < nc; ++t) {
        const SnDesc& T = S.csn[t];
        for (int64_t u = T.upd0; u < T.upd0 + T.nupd; ++u)
            if (S.level_of[S.upd_src[u]] >= S.level_of[t]) fail("update " + std::to_string(u) + " comes from a level that is not below its target");
        if (is_small(T) || T.nupd == 0) continue;
        // map an entry back to its update: (src, K) is unique among the updates of one target
        auto find_upd = [&](const WaveEntry& E) -> int64_t {
            for (int64_t u = T.upd0; u < T.upd0 + T.nupd; ++u)
                if (S.upd[u].src == E.src && S.upd[u].K == E.K) return u;
            return -1;
        };
        if (S.sn_wp0[t] >= 0) {
            const int nbc = ceil_div(T.w, kTile), nbr = ceil_div(T.r, kTile);
            const int64_t* wp = &S.wave_ptr[S.sn_wp0[t]];
            for (size_t key = 0; key < (size_t)nbc * nbr * 8; ++key)
                for (int64_t e = wp[key]; e < wp[key + 1]; ++e) {
                    const WaveEntry& E = S.wave_entries[(size_t)e];
                    const int64_t u = find_upd(E);
                    const int mi = E.mn & 255, nj = (E.mn >> 8) & 255;
                    if (u < 0 || mi < 1 || mi > kSub || nj < 1 || nj > kSub || E.ia + mi > S.upd[u].m || E.ja + nj > S.upd[u].n1) {
                        fail("wave entry " + std::to_string(e) + " of piece " + std::to_string(t) + " has a bad window");
                        continue;
                    }
                    covered[(size_t)u] += entry_flops(E, E.ia == E.ja);
                }
        }
    }
    for (const Schedule::BigTask& b : S.big_all) {
        const SnDesc& T = S.csn[b.sn];
        if (b.src_level >= S.level_of[b.sn] || (b.next != 0) != (b.src_level == S.level_of[b.sn] - 1))
            fail("BIG task of piece " + std::to_string(b.sn) + " is filed under the wrong source level");
        const int win_r = kBigTile * b.sr, win_c = kBigTile * b.sc;
        if (b.sr < 1 || b.sc < 1 || b.row0 % win_r || b.col0 % win_c || b.row0 + win_r <= b.col0 || b.row0 >= T.r || b.col0 >= T.w)
            fail("BIG task of piece " + std::to_string(b.sn) + " has a bad tile origin");
        for (int64_t e = b.e0; e < b.e1; ++e) {
            const WaveEntry& E = S.big_entries[(size_t)e];
            int64_t u = -1;
            for (int64_t q = T.upd0; q < T.upd0 + T.nupd; ++q)
                if (S.upd[q].src == E.src && S.upd[q].K == E.K) u = q;
            const int mi = E.mn & 255, nj = (E.mn >> 8) & 255;
            const bool ident = big_ident(E);
            const int ms = big_strip_rows(E), ns = big_strip_cols(E);
            if ((ms || ns) && (e >= b.em || mi != kBigTile || nj != kBigTile || ms > kStripMax || ns > kStripMax))
                fail("BIG entry " + std::to_string(e) + " carries a strip but is not a full dense block");
            if (u < 0 || mi < 1 || mi > kBigTile || nj < 1 || nj > kBigTile || E.ia + mi + ms > S.upd[u].m || E.ja + nj + ns > S.upd[u].n1 ||
                ident != (S.upd[u].rel < 0) || S.level_of[S.upd_src[u]] != b.src_level) {
                fail("BIG entry " + std::to_string(e) + " of piece " + std::to_string(b.sn) + " has a bad window or source");
                continue;
            }
            // the rows it names land in the task's tile
            auto rel_at = [&](int k) { return ident ? k : S.relpos[(size_t)S.upd[u].rel + k] - T.rbias; };
            if (rel_at(E.ia) / win_r != b.row0 / win_r || rel_at(E.ia + mi + ms - 1) / win_r != b.row0 / win_r ||
                rel_at(E.ja) / win_c != b.col0 / win_c || rel_at(E.ja + nj + ns - 1) / win_c != b.col0 / win_c)
                fail("BIG entry " + std::to_string(e) + " names rows outside its task's tile");
            covered[(size_t)u] += block_flops(E);
        }
    }
    {
        // the BIG / DENSE launches hold every non-empty part (dense: [e0, em), ragged: [em, e1)) of every task of an
        // active target exactly once, under its (source level, kind), the dense part in a launch enqueued BEFORE the
        // one that holds the ragged part; padding tasks (XCD sequences of unequal length) have no entries
        std::vector<int64_t> starts;
        std::vector<int64_t> dense_pos(S.big_all.size(), -1), ragged_pos(S.big_all.size(), -1);
        int64_t li = 0;
        for (const Launch& l : S.chol) {
            ++li;
            if (l.kind != kLaunchBig && l.kind != kLaunchDense) continue;
            const bool dense = l.kind == kLaunchDense;
            const bool side = l.side != 0;
            auto find_task = [&](int64_t e) -> const Schedule::BigTask* {   // the task whose entry range holds big_entries[e]
                auto it = std::upper_bound(S.big_all.begin(), S.big_all.end(), e,
                                           [](int64_t v, const Schedule::BigTask& b) { return v < b.e0; });
                if (it == S.big_all.begin()) return nullptr;
                --it;
                return e < it->e1 ? &*it : nullptr;
            };
            auto right_launch = [&](const Schedule::BigTask& b) {
                return side != (b.next != 0) && (side ? l.wait_level : l.level - 1) == b.src_level;
            };
            for (int q = l.first; q < l.first + l.count; ++q) {
                const TileDesc& td = S.big_tasks[(size_t)q];
                if (td.wp >= td.sp) {
                    if (td.wp != td.sp) fail("BIG launch task " + std::to_string(q) + " has a negative entry range");
                    continue;
                }
                const Schedule::BigTask* it = find_task(td.wp);
                const bool ok = it && (dense ? (it->e0 == td.wp && it->em == td.sp && (td.part & ~kDenseStripTask) == it->dchunks &&
                                                ((td.part & kDenseStripTask) != 0) == (it->strips > 0))
                                             : (it->em == td.wp && it->e1 == td.sp));
                if (!ok || it->sn != td.sn || it->row0 != td.row0 || it->col0 != td.col0) {
                    fail("BIG launch task " + std::to_string(q) + " is not a task of the plan");
                    continue;
                }
                if (!right_launch(*it)) fail("BIG launch task " + std::to_string(q) + " runs in the launch of another source level");
                (dense ? dense_pos : ragged_pos)[(size_t)(it - S.big_all.data())] = li;
                starts.push_back(td.wp);
            }
        }
        std::sort(starts.begin(), starts.end());
        if (std::adjacent_find(starts.begin(), starts.end()) != starts.end()) fail("a BIG task is launched twice");
        int64_t want = 0;
        for (size_t k = 0; k < S.big_all.size(); ++k) {
            const Schedule::BigTask& b = S.big_all[k];
            if (b.em < b.e0 || b.em > b.e1) fail("BIG task " + std::to_string(k) + " has a bad dense / ragged split");
            int64_t dch = 0;
            int32_t nst = 0;
            for (int64_t e = b.e0; e < b.em; ++e) {
                const WaveEntry& E = S.big_entries[(size_t)e];
                nst += big_strip_rows(E) || big_strip_cols(E);
                if (E.K < kDenseChunk) fail("BIG entry " + std::to_string(e) + " is filed as dense but is narrower than a chunk");
                dch += ceil_div(E.K, kDenseChunk);
            }
            if (dch != b.dchunks) fail("BIG task " + std::to_string(k) + " has a wrong dense chunk count");
            if (nst != b.strips) fail("BIG task " + std::to_string(k) + " has a wrong count of entries with strips");
            if (!S.active_piece[b.sn]) continue;
            want += (b.em > b.e0) + (b.e1 > b.em);
            if (b.em > b.e0 && b.e1 > b.em && !(dense_pos[k] > 0 && ragged_pos[k] > dense_pos[k]))
                fail("BIG task " + std::to_string(k) + ": the ragged part is not launched after the dense part");
        }
        if ((int64_t)starts.size() != want)
            fail("BIG launches hold " + std::to_string(starts.size()) + " task parts, the active targets have " + std::to_string(want));
    }
    for (int t = 0; t < nc; ++t) {
        const SnDesc& T = S.csn[t];
        if (is_small(T)) continue;
        for (int64_t u = T.upd0; u < T.upd0 + T.nupd; ++u) {
            const UpdDesc& U = S.upd[u];
            const double want = (double)U.K * U.n1 * (U.n1 + 1) + 2.0 * U.K * (double)(U.m - U.n1) * U.n1;
            if (std::abs(covered[(size_t)u] - want) > 1e-9 * want + 0.5)
                fail("update " + std::to_string(u) + " of piece " + std::to_string(t) + " is not covered exactly once by its entries");
        }
    }
    // ---- subtree launches: what a member depends on is in the same subtree; a workgroup's run is in index
    // order; every active SMALL supernode is factored by exactly one workgroup of one launch
    for (int t = 0; t < nc; ++t) {
        const int real = S.csn_real[t], st = S.chol_subtree[real];
        if (st < 0) continue;
        if (!is_small(S.csn[t])) fail("supernode " + std::to_string(real) + " of a subtree is not a SMALL one");
        for (int64_t u = S.csn[t].upd0; u < S.csn[t].upd0 + S.csn[t].nupd; ++u)
            if (S.chol_subtree[S.csn_real[S.upd_src[u]]] != st)
                fail("supernode " + std::to_string(real) + " of a subtree is updated from outside the subtree");
    }
    {
        std::vector<int> seen(nc, 0);
        for (const Launch& l : S.chol) {
            if (l.kind != kLaunchSmall) continue;
            if (l.fused == 2) {
                for (int b = l.first; b < l.first + l.count; ++b) {
                    const int32_t q0 = S.small_ranges[2 * (size_t)b], q1 = S.small_ranges[2 * (size_t)b + 1];
                    if (q0 < 0 || q1 <= q0 || q1 > (int32_t)S.small_list.size()) {
                        fail("subtree launch: bad range " + std::to_string(b));
                        continue;
                    }
                    for (int32_t q = q0; q < q1; ++q) {
                        seen[S.small_list[q]]++;
                        if (q > q0 && S.small_list[q] <= S.small_list[q - 1]) fail("subtree launch: a run is not in index order");
                        if (S.chol_subtree[S.csn_real[S.small_list[q]]] != S.chol_subtree[S.csn_real[S.small_list[q0]]])
                            fail("subtree launch: a run mixes subtrees");
                    }
                }
            } else {
                for (int q = l.first; q < l.first + l.count; ++q) seen[S.small_list[q]]++;
            }
        }
        for (int t = 0; t < nc; ++t)
            if (seen[t] != ((is_small(S.csn[t]) && S.active_piece[t]) ? 1 : 0))
                fail("SMALL supernode " + std::to_string(t) + " is in " + std::to_string(seen[t]) + " launches");
    }
    // ---- launch sequence: a side launch comes before the main-stream launches of the level that waits for it and
    // after every main-stream launch of the levels it waits for; the main stream never goes down a level
    int main_level = -1;
    for (size_t i = 0; i < S.chol.size(); ++i) {
        const Launch& l = S.chol[i];
        if (l.side) {
            if (l.wait_level >= 0 && main_level > l.wait_level + 0 && main_level >= l.level)
                fail("side launch " + std::to_string(i) + " is enqueued after the level that waits for it");
            for (size_t j = i + 1; j < S.chol.size(); ++j)
                if (!S.chol[j].side && S.chol[j].level <= l.wait_level)
                    fail("side launch " + std::to_string(i) + " precedes a main-stream launch of a level it waits for");
        } else {
            if (l.level < main_level) fail("main-stream launches go down a level");
            main_level = l.level;
        }
    }
    return bad;
}

}  // namespace parsy
