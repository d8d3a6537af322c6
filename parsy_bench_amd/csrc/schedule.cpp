// Host-side construction of the device schedule (see schedule.hpp).
#include "schedule.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstdint>
#include <stdexcept>
#include <string>

namespace parsy {

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

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
    }

    // --- update lists, relative indices, A scatter map ---------------------------
    std::vector<int64_t> uptr(ns + 1, 0);
    std::vector<int> usn, ulb, uub;
    if (!S.solve_only) build_update_lists(P, uptr, usn, ulb, uub);
    S.upd.resize(usn.size());
    S.a_dst.assign((size_t)S.nnzA, 0);
    std::vector<int> map(n, -1), stamp(n, -1);
    for (int t = 0; t < ns; ++t) {
        SnDesc& T = S.sn[t];
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
        T.upd0 = uptr[t];
        T.nupd = (int)(uptr[t + 1] - uptr[t]);
        for (int64_t u = uptr[t]; u < uptr[t + 1]; ++u) {
            const SnDesc& D = S.sn[usn[u]];
            UpdDesc& U = S.upd[u];
            U.src = D.px + ulb[u];
            U.ld = D.r;
            U.K = D.w;
            U.m = D.r - ulb[u];
            U.n1 = uub[u] - ulb[u] + 1;
            U.rel = (int64_t)S.relpos.size();
            for (int k = ulb[u]; k < D.r; ++k) {
                const int row = S.rows[D.pi + k];
                if (stamp[row] != t)
                    throw std::runtime_error("schedule: descendant row missing from the target's pattern");
                S.relpos.push_back(map[row]);
            }
            S.update_flops += (double)U.K * U.n1 * (U.n1 + 1) + 2.0 * U.K * (double)(U.m - U.n1) * U.n1;
            S.reread_bytes += 8.0 * (double)U.K * U.m;
        }
    }

    if (S.relpos.size() > 0x7fffffffULL) throw std::runtime_error("schedule: relative index array exceeds int32");
    std::vector<int> tree(P.sparent, P.sparent + ns);
    level_sets(tree, S.levelPtr, S.levelSet);
    S.nlevels = (int)S.levelPtr.size() - 1;
    S.level_of.assign(ns, 0);
    for (int l = 0; l < S.nlevels; ++l)
        for (int q = S.levelPtr[l]; q < S.levelPtr[l + 1]; ++q) S.level_of[S.levelSet[q]] = l;

    // --- tiled supernodes: scratch slots and the per-wave update streams ---------------
    // Every (target, descendant) update is cut along the 32-row windows of the target's tiles:
    // the descendant rows that fall into row window I32 times those (among its first n1 rows)
    // that fall into column window J32 are one WaveEntry of sub-tile (I32, J32), I32 >= J32.
    // Lists keep the reference's update order, so the sum order per entry of L is fixed.
    S.sn_wp0.assign(ns, -1);
    S.sn_tw0.assign(ns, -1);
    struct Group { int32_t win, first, len; };
    std::vector<Group> groups;
    std::vector<int64_t> cursor;
    for (int t = 0; t < ns; ++t) {
        SnDesc& T = S.sn[t];
        if (is_small(T)) {
            S.n_small++;
            continue;
        }
        S.n_big++;
        const int nbc = ceil_div(T.w, kTile), nbr = ceil_div(T.r, kTile);
        T.dslot = (int32_t)S.n_dslots;
        S.n_dslots += nbc;
        if (S.n_tflags + (int64_t)nbc * nbr > 0x7fffffffLL) throw std::runtime_error("schedule: too many tiles");
        T.tflag0 = (int32_t)S.n_tflags;
        S.n_tflags += (int64_t)nbc * nbr;
        for (int jb = 1; jb < nbc; ++jb) {
            const double K = (double)jb * kTile, wb = std::min(kTile, T.w - jb * kTile);
            S.inner_flops += K * wb * (wb + 1) + 2.0 * K * (double)(T.r - jb * kTile - wb) * wb;
        }
        // key of a list: ((J * nbr + I) * 2 + phase) * 4 + wave
        const size_t nkeys = (size_t)nbc * nbr * 8;
        auto for_each_entry = [&](auto&& fn) {
            for (int64_t u = T.upd0; u < T.upd0 + T.nupd; ++u) {
                const UpdDesc& U = S.upd[u];
                // early: the descendant is complete before the level below the target even starts
                const int phase = (S.level_of[usn[u]] <= S.level_of[t] - 2) ? 0 : 1;
                const int32_t* rel = &S.relpos[U.rel];
                groups.clear();
                for (int k = 0; k < U.m;) {
                    const int win = rel[k] / kSub;
                    int k1 = k + 1;
                    while (k1 < U.m && rel[k1] / kSub == win) ++k1;
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
        for (int64_t u = T.upd0; u < T.upd0 + T.nupd; ++u) {
            const UpdDesc& U = S.upd[u];
            S.tile_update_flops += (double)U.K * U.n1 * (U.n1 + 1) + 2.0 * U.K * (double)(U.m - U.n1) * U.n1;
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
            if ((tp & 1) == 0 && tw[tp] > kSplitChunks) {
                const int nparts = std::min<int>(kSplitMaxParts, ceil_div(tw[tp], kSplitTarget));
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

    build_launches(S, active);
}

void build_launches(Schedule& S, const uint8_t* active) {
    const int ns = S.nsuper;
    // PARSY_FORCE_UNFUSED=1 schedules the solve's fallback form everywhere (per-block-column
    // launches): the path taken when a chain launch would not be resident.  Used by the tests.
    const char* fu = std::getenv("PARSY_FORCE_UNFUSED");
    const bool force_unfused = fu && fu[0] == '1';
    const int max_chain = force_unfused ? -1 : kMaxChainWorkgroups;
    S.active.assign(ns, 1);
    if (active) S.active.assign(active, active + ns);
    S.small_list.clear();
    S.tiles.clear();
    S.n_chain_launches = 0;
    S.chol.clear();
    S.solve_small_list.clear();
    S.solve_panels.clear();
    S.solve_fix_list.clear();
    S.solve_wide_list.clear();
    S.solve_wide_max_blocks = 0;
    S.solve.clear();
    S.n_solve_wide = 0;

    // TILES_EARLY(lev) is enqueued right before level lev-1's launches, so that it runs on the side
    // stream while the main stream works through that level's block-column chain.
    std::vector<Launch> early_launches;
    std::vector<size_t> level_begin;  // index in S.chol where each level's launches start
    std::vector<int> bigs, sbigs;
    for (int lev = 0; lev < S.nlevels; ++lev) {
        level_begin.push_back(S.chol.size());
        bigs.clear();
        sbigs.clear();
        // ---- Cholesky -------------------------------------------------------------
        if (!S.solve_only) {
            Launch L{kLaunchSmall, (int32_t)S.small_list.size(), 0, lev, 0, 0, 0, 0, -1, 0};
            for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
                const int t = S.levelSet[q];
                if (!S.active[t]) continue;
                const SnDesc& T = S.sn[t];
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
        if (!S.solve_only && !bigs.empty()) {
            // ---- TILES: the early part of the external updates, longest streams first ------
            {
                Launch Lt{kLaunchTiles, (int32_t)S.tiles.size(), 0, lev, 0, 0, 0, 1, lev - 2, 1};
                std::vector<std::pair<int32_t, TileDesc>> wt;  // (weight, tile)
                for (int t : bigs) {
                    const SnDesc& T = S.sn[t];
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
            Launch Lc{kLaunchChain, (int32_t)S.tiles.size(), 0, lev, S.n_chain_launches++, 0, 0, 0, -1, 0};
            // Walkers stay resident for their whole chain, so the block-column-major interleaving is
            // done per batch of at most walker_batch supernodes: what a walker waits for then lies at most
            // one batch of tiles ahead of it in ticket order, and the walkers of the batches in flight
            // never take more than a fraction of the resident workgroups.
            for (size_t b0 = 0; b0 < bigs.size(); b0 += (size_t)S.walker_batch) {
              const size_t b1 = std::min(bigs.size(), b0 + (size_t)S.walker_batch);
              int maxnb = 0;
              for (size_t q = b0; q < b1; ++q) maxnb = std::max(maxnb, ceil_div(S.sn[bigs[q]].w, kTile));
              for (int J = 0; J < maxnb; ++J)
                for (size_t q = b0; q < b1; ++q) {
                    const int t = bigs[q];
                    const SnDesc& T = S.sn[t];
                    const int nbc = ceil_div(T.w, kTile), nbr = ceil_div(T.r, kTile);
                    if (J >= nbc) continue;
                    auto push = [&](int I, int Jc) {
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
            S.chol.push_back(Lc);
        }
        // ---- forward solve ----------------------------------------------------------
        {
            Launch L{kLaunchSolveSmall, (int32_t)S.solve_small_list.size(), 0, lev, 0, 0, 0, 0, -1, 0};
            for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
                const int t = S.levelSet[q];
                if (!S.active[t]) continue;
                if (S.sn[t].w <= kTile) S.solve_small_list.push_back(t);
                else sbigs.push_back(t);
            }
            L.count = (int32_t)S.solve_small_list.size() - L.first;
            if (L.count > 0) S.solve.push_back(L);
        }
        if (!sbigs.empty()) {
            S.n_solve_wide += (int)sbigs.size();
            int chain_wgs = 0;
            for (int t : sbigs) chain_wgs += ceil_div(S.sn[t].r, kSolveRows);
            if (chain_wgs <= max_chain) {
                // one launch: every 256-row chunk of every wide supernode of the level
                Launch Lc{kLaunchSolvePanel, (int32_t)S.solve_panels.size(), 0, lev, 0, 0, 1, 0, -1, 0};
                for (int t : sbigs)
                    for (int c = 0; c * kSolveRows < S.sn[t].r; ++c)
                        S.solve_panels.push_back(PanelDesc{t, c, c * kSolveRows, 0});
                Lc.count = (int32_t)S.solve_panels.size() - Lc.first;
                S.solve.push_back(Lc);
                for (int t : sbigs) {
                    S.solve_wide_list.push_back(t);
                    S.solve_wide_max_blocks = std::max(S.solve_wide_max_blocks, ceil_div(S.sn[t].w, kTile));
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
    // splice the early launches in front of the level before their target level
    if (!early_launches.empty()) {
        std::vector<Launch> merged;
        size_t e = 0;
        std::sort(early_launches.begin(), early_launches.end(),
                  [](const Launch& a, const Launch& b) { return a.level < b.level; });
        for (int lev = 0; lev < S.nlevels; ++lev) {
            while (e < early_launches.size() && early_launches[e].level - 1 <= lev) merged.push_back(early_launches[e++]);
            const size_t b0 = level_begin[lev], b1 = lev + 1 < S.nlevels ? level_begin[lev + 1] : S.chol.size();
            merged.insert(merged.end(), S.chol.begin() + b0, S.chol.begin() + b1);
        }
        S.chol.swap(merged);
    }
    // ---- backward solve: root level first.  Per level one chain launch for the block columns of
    // the wide supernodes (last block column first: block jb waits for the published x of blocks
    // jb+1.. of its supernode; every workgroup of the launch must be resident) and one launch for the
    // supernodes of a single block; when the chain would not be resident, one launch per block-column
    // index from the last one down.
    S.bsolve_blocks.clear();
    S.bsolve.clear();
    for (int lev = S.nlevels - 1; lev >= 0; --lev) {
        int maxnb = 0, wide_blocks = 0;
        for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
            const int t = S.levelSet[q];
            if (!S.active[t]) continue;
            const int nbc = ceil_div(S.sn[t].w, kTile);
            maxnb = std::max(maxnb, nbc);
            if (nbc > 1) wide_blocks += nbc;
        }
        if (wide_blocks > 0 && wide_blocks <= max_chain) {
            Launch Lc{kLaunchBackBlock, (int32_t)S.bsolve_blocks.size(), 0, lev, 0, 0, 1, 0, -1, 0};
            for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
                const int t = S.levelSet[q];
                const int nbc = ceil_div(S.sn[t].w, kTile);
                if (!S.active[t] || nbc < 2) continue;
                for (int jb = nbc - 1; jb >= 0; --jb) S.bsolve_blocks.push_back(PanelDesc{t, jb, 0, 0});
            }
            Lc.count = (int32_t)S.bsolve_blocks.size() - Lc.first;
            S.bsolve.push_back(Lc);
            Launch Ln{kLaunchBackBlock, (int32_t)S.bsolve_blocks.size(), 0, lev, 0, 0, 0, 0, -1, 0};
            for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
                const int t = S.levelSet[q];
                if (S.active[t] && ceil_div(S.sn[t].w, kTile) == 1) S.bsolve_blocks.push_back(PanelDesc{t, 0, 0, 0});
            }
            Ln.count = (int32_t)S.bsolve_blocks.size() - Ln.first;
            if (Ln.count > 0) S.bsolve.push_back(Ln);
            continue;
        }
        for (int jb = maxnb - 1; jb >= 0; --jb) {
            Launch Lb{kLaunchBackBlock, (int32_t)S.bsolve_blocks.size(), 0, lev, jb, 0, 0, 0, -1, 0};
            for (int q = S.levelPtr[lev]; q < S.levelPtr[lev + 1]; ++q) {
                const int t = S.levelSet[q];
                if (S.active[t] && ceil_div(S.sn[t].w, kTile) > jb) S.bsolve_blocks.push_back(PanelDesc{t, jb, 0, 0});
            }
            Lb.count = (int32_t)S.bsolve_blocks.size() - Lb.first;
            if (Lb.count > 0) S.bsolve.push_back(Lb);
        }
    }
    if (!S.solve_fix_list.empty())
        S.solve.push_back(Launch{kLaunchSolveFixup, 0, (int32_t)S.solve_fix_list.size(), S.nlevels, 0, 0, 0, 0, -1, 0});
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
                const SnDesc& D = S.sn[td.sn];
                const int nbc = ceil_div(D.w, kTile);
                Task t{td.sn, td.row0 / kTile, td.col0 / kTile, 3, 0};
                if (t.I == 0 && t.J == 0) t.role = 0;
                else if (t.I == t.J) t.role = 2;
                else if (t.I == t.J + 1 && t.I < nbc) t.role = 1;
                resident.push_back(t);
                progress = true;
            }
            for (size_t q = 0; q < resident.size();) {
                Task& t = resident[q];
                const SnDesc& D = S.sn[t.sn];
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

}  // namespace parsy
