// Plan construction, upload and launch sequencing (host side of the HIP executor).
#include "executor.hpp"

#include <climits>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "errors.hpp"
#include "plan_fwd.hpp"

namespace parsy {

const Schedule& plan_schedule(const parsy_plan* plan) { return plan->S; }

#define PARSY_HIP(call)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            set_last_error(std::string(#call) + ": " + hipGetErrorString(e_));               \
            return -1;                                                                       \
        }                                                                                    \
    } while (0)

template <typename T>
static int upload(parsy_plan* pl, const std::vector<T>& v, const T*& dptr, bool launch_array) {
    dptr = nullptr;
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    void* d = nullptr;
    PARSY_HIP(hipMalloc(&d, bytes));
    (launch_array ? pl->launch_owned : pl->owned).push_back(d);
    if (!v.empty()) PARSY_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    pl->device_bytes += (int64_t)bytes;
    dptr = (const T*)d;
    return 0;
}

int plan_upload_launches(parsy_plan* pl) {
    if (pl->device < 0) return 0;
    PARSY_HIP(hipSetDevice(pl->device));
    for (void* d : pl->launch_owned) (void)hipFree(d);
    pl->launch_owned.clear();
    pl->levels_open = false;
    const Schedule& S = pl->S;
    if (upload(pl, S.small_list, pl->dp.small_list, true)) return -1;
    if (upload(pl, S.small_ranges, pl->dp.small_ranges, true)) return -1;
    if (upload(pl, S.solve_small_ranges, pl->dp.solve_small_ranges, true)) return -1;
    if (upload(pl, S.bsolve_ranges, pl->dp.bsolve_ranges, true)) return -1;
    if (upload(pl, S.tiles, pl->dp.tiles, true)) return -1;
    if (upload(pl, S.big_tasks, pl->dp.big_tasks, true)) return -1;
    if (upload(pl, S.solve_small_list, pl->dp.solve_small_list, true)) return -1;
    if (upload(pl, S.solve_panels, pl->dp.solve_panels, true)) return -1;
    if (upload(pl, S.solve_mtasks, pl->dp.solve_mtasks, true)) return -1;
    if (upload(pl, S.solve_fix_list, pl->dp.solve_fix_list, true)) return -1;
    if (upload(pl, S.solve_wide_list, pl->dp.solve_wide_list, true)) return -1;
    if (upload(pl, S.bsolve_blocks, pl->dp.bsolve_blocks, true)) return -1;
    if (upload(pl, S.bsolve_below, pl->dp.bsolve_below, true)) return -1;
    if (pl->dp.bpart) {   // (sized by the launch lists)
        (void)hipFree(pl->dp.bpart);
        pl->dp.bpart = nullptr;
    }
    if (upload(pl, S.bsolve_pairs, pl->dp.bsolve_pairs, true)) return -1;
    if (upload(pl, S.sub_members, pl->dp.sub_members, true) || upload(pl, S.sub_trees, pl->dp.sub_trees, true) ||
        upload(pl, S.sub_slots, pl->dp.sub_slots, true) || upload(pl, S.sub_out_rows, pl->dp.sub_out_rows, true))
        return -1;
    // (trees whose workgroup would not get its LDS: the subtree launches keep the level kernels' form)
    pl->dp.sub_ntiers = (!S.sub_tiers.empty() && solve_sub_prepare(S.sub_max_slots) == 0) ? (int)S.sub_tiers.size() : 0;
    {
        auto up = [&](const Schedule::OneLists& O, DevicePattern::OneDev& D) {
            D = DevicePattern::OneDev();
            if (upload(pl, O.sn, D.sn, true) || upload(pl, O.slot0, D.slot0, true) || upload(pl, O.wleft, D.wleft, true) ||
                upload(pl, O.pull_ptr, D.pull_ptr, true) || upload(pl, O.pull_slot, D.pull_slot, true) ||
                upload(pl, O.pull_pos, D.pull_pos, true))
                return -1;
            D.nblocks = (int)O.sn.size();
            D.nslots = std::max<int64_t>(O.nslots, 1);
            return 0;
        };
        if (up(S.one_f, pl->dp.one_f)) return -1;
        if (S.one_b.sn.empty()) pl->dp.one_b = pl->dp.one_f;
        else if (up(S.one_b, pl->dp.one_b)) return -1;
        // (the hand-off buffers are sized by the lists: made again by the next ONE-launch solve; the status word of the
        // last such solve lived in them)
        for (int d = 0; d < 2; ++d) {
            if (pl->one_y[d]) {
                (void)hipFree(pl->one_y[d]);
                pl->device_bytes -= pl->one_bytes[d];
            }
            pl->one_y[d] = nullptr;
            pl->one_bytes[d] = 0;
            pl->one_cap[d] = 0;
            pl->one_calls[d] = 0;
        }
        pl->solve_status_word = nullptr;
    }
    {
        void* d = nullptr;
        PARSY_HIP(hipMalloc(&d, (size_t)std::max(S.n_chain_launches, 1) * sizeof(int)));
        pl->launch_owned.push_back(d);
        pl->dp.tickets = (int*)d;
        PARSY_HIP(hipMalloc(&d, (size_t)std::max(S.n_solve_chain_launches, 1) * sizeof(int)));
        pl->launch_owned.push_back(d);
        pl->dp.stickets = (int*)d;
    }
    // hipMemcpy from pageable memory returns when the data is staged, hipMemset when it is enqueued:
    // both are only ordered against the NULL stream.  The caller's stream may be a non-blocking one,
    // so everything must have landed before the plan is handed out.
    PARSY_HIP(hipDeviceSynchronize());
    return 0;
}

static int plan_upload(parsy_plan* pl) {
    PARSY_HIP(hipSetDevice(pl->device));
    const Schedule& S = pl->S;
    if (upload(pl, S.sn, pl->dp.sn, false)) return -1;
    if (upload(pl, S.csn, pl->dp.csn, false)) return -1;
    if (upload(pl, S.big_entries, pl->dp.big_entries, false)) return -1;
    if (upload(pl, S.upd, pl->dp.upd, false)) return -1;
    if (upload(pl, S.relpos, pl->dp.relpos, false)) return -1;
    if (upload(pl, S.a_dst, pl->dp.a_dst, false)) return -1;
    if (upload(pl, S.rows, pl->dp.rows, false)) return -1;
    if (upload(pl, S.wave_entries, pl->dp.wave_entries, false)) return -1;
    if (upload(pl, S.wave_ptr, pl->dp.wave_ptr, false)) return -1;
    if (upload(pl, S.split_ranges, pl->dp.split_ranges, false)) return -1;
    {
        void* d = nullptr;
        const size_t sbytes = (size_t)std::max<int64_t>(S.n_split_doubles, 1) * sizeof(double);
        PARSY_HIP(hipMalloc(&d, sbytes));
        pl->owned.push_back(d);
        pl->dp.tile_scratch = (double*)d;
        pl->device_bytes += (int64_t)sbytes;
    }
    {
        void* d = nullptr;
        PARSY_HIP(hipMalloc(&d, sizeof(int)));
        pl->owned.push_back(d);
        pl->dp.info = (int*)d;
        PARSY_HIP(hipMemset(d, 0x7f, sizeof(int)));
        PARSY_HIP(hipMalloc(&d, sizeof(int)));
        pl->owned.push_back(d);
        pl->dp.sinfo = (int*)d;
        PARSY_HIP(hipMemset(d, 0, sizeof(int)));
        pl->n_flags = std::max<int64_t>(S.n_dslots, 1) * kPassLanes;
        const size_t fbytes = std::max<int64_t>(S.n_dslots, 1) * kPassLanes * sizeof(int);
        pl->dp.flag_stride = (int)std::max<int64_t>(S.n_dslots, 1);
        PARSY_HIP(hipMalloc(&d, fbytes));
        pl->owned.push_back(d);
        pl->dp.flags = (int*)d;
        PARSY_HIP(hipMemset(d, 0, fbytes));
        const size_t tbytes = 2 * std::max<int64_t>(S.n_tflags, 1) * sizeof(int);
        pl->dp.n_tflags = (int)S.n_tflags;
        PARSY_HIP(hipMalloc(&d, tbytes));
        pl->owned.push_back(d);
        pl->dp.tflags = (int*)d;
        PARSY_HIP(hipMemset(d, 0, tbytes));
        pl->device_bytes += (int64_t)(fbytes + tbytes);
    }
    {
        // lowest priority: the side stream only fills what the main stream's chain leaves idle
        int prio_least = 0, prio_greatest = 0;
        PARSY_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        PARSY_HIP(hipStreamCreateWithPriority(&pl->side_stream, hipStreamNonBlocking, prio_least));
    }
    PARSY_HIP(hipEventCreateWithFlags(&pl->ev_init, hipEventDisableTiming));
    pl->ev_level_done.resize(std::max(S.nlevels, S.cnlevels) + 1);
    pl->ev_early_done.resize(std::max(S.nlevels, S.cnlevels) + 1);
    for (auto& e : pl->ev_level_done) PARSY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : pl->ev_early_done) PARSY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    {
        const char* no = std::getenv("PARSY_NO_OVERLAP");
        pl->overlap = !(no && no[0] == '1');
    }
    PARSY_HIP(hipEventCreate(&pl->ev_f0));
    PARSY_HIP(hipEventCreate(&pl->ev_f1));
    PARSY_HIP(hipEventCreate(&pl->ev_s0));
    PARSY_HIP(hipEventCreate(&pl->ev_s1));
    return plan_upload_launches(pl);
}

parsy_plan* plan_build(const PatternRef& P, const size_t* lC, const int* A2p, const int* A2i,
                       int device) {
    parsy_plan* pl = new parsy_plan;
    int cus = 0;
    if (device >= 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
        // the schedule sizes its walker batches for 2 resident chain workgroups per CU: if the runtime
        // admits fewer (another register / LDS budget), count the CUs accordingly
        (void)hipSetDevice(device);
        const int per_cu = chain_workgroups_per_cu();
        if (per_cu == 1) cus = cus / 2;
    }
    try {
        build_schedule(P, lC, A2p, A2i, nullptr, pl->S, cus);
    } catch (const std::exception& e) {
        set_last_error(std::string("plan: ") + e.what());
        delete pl;
        return nullptr;
    }
    pl->solve_only = pl->S.solve_only;
    pl->device = device;
    if (device >= 0 && plan_upload(pl) != 0) {
        plan_free(pl);
        return nullptr;
    }
    return pl;
}

void plan_free(parsy_plan* pl) {
    if (!pl) return;
    if (pl->device >= 0) {
        (void)hipSetDevice(pl->device);
        (void)hipDeviceSynchronize();
        for (void* d : pl->owned) (void)hipFree(d);
        for (void* d : pl->launch_owned) (void)hipFree(d);
        if (pl->xscratch) (void)hipFree(pl->xscratch);
        for (double*& q : pl->one_y) {   // (the state words live in the same allocations)
            if (q) (void)hipFree(q);
            q = nullptr;
        }
        if (pl->xt) (void)hipFree(pl->xt);
        if (pl->dinv) (void)hipFree(pl->dinv);
        if (pl->dp.bpart) (void)hipFree(pl->dp.bpart);
        if (pl->h_values_dev) (void)hipFree(pl->h_values_dev);
        if (pl->h_L_dev) (void)hipFree(pl->h_L_dev);
        if (pl->h_x_dev) (void)hipFree(pl->h_x_dev);
        for (hipEvent_t e : {pl->ev_f0, pl->ev_f1, pl->ev_s0, pl->ev_s1})
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : pl->pev) (void)hipEventDestroy(e);
        for (hipEvent_t e : pl->ev_level_done) (void)hipEventDestroy(e);
        for (hipEvent_t e : pl->ev_early_done) (void)hipEventDestroy(e);
        if (pl->ev_init) (void)hipEventDestroy(pl->ev_init);
        if (pl->side_stream) (void)hipStreamDestroy(pl->side_stream);
        if (pl->h_stream) (void)hipStreamDestroy(pl->h_stream);
        if (pl->h_copy) (void)hipStreamDestroy(pl->h_copy);
        for (hipEvent_t e : pl->h_band_ev) (void)hipEventDestroy(e);
    }
    delete pl;
}

static void profile_mark(parsy_plan* pl, int kind, hipStream_t stream, size_t& cursor, int level = -1, int side = 0,
                         int count = 0) {
    if (!pl->profile) return;
    if (cursor >= pl->pev_count.size()) pl->pev_count.resize(cursor + 1, 0);
    pl->pev_count[cursor] = count;
    if (cursor >= pl->pev_level.size()) pl->pev_level.resize(cursor + 1, 0);
    pl->pev_level[cursor] = level >= 0 ? (level << 1) | (side & 1) : -1;
    if (cursor >= pl->pev.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        pl->pev.push_back(e);
    }
    if (cursor >= pl->pev_kind.size()) pl->pev_kind.push_back(kind);
    else pl->pev_kind[cursor] = kind;
    (void)hipEventRecord(pl->pev[cursor], stream);
    ++cursor;
}

// The subtree launches of a solve with this many right-hand sides take the form of one wave per subtree and 16 right-
// hand sides (trsv_sub_kernels.hip; PARSY_SUB_MRHS_MIN, 0: never)
// (the kernels address x by a 32-bit byte offset from a wave-uniform base: 16 rows or right-hand sides of either stride)
static bool sub_tiers_usable(const parsy_plan* pl, int nrhs, int ldx) {
    const int m = solve_sub_mrhs_min();
    return pl->dp.sub_ntiers > 0 && m > 0 && nrhs >= m && ldx < (1 << 24) && nrhs < (1 << 20);
}

// Enqueue the launches seq[i0, i1).  The state that orders the two streams (which level-completion events have been
// recorded, which side launches are in flight, the profiling cursor) lives in the plan, so that a factorization can
// be enqueued level by level (parsy_factor_level) with the caller's own work -- the exchange step of a
// multi-device run -- in between: run_begin() resets it, run_end() closes the profiling record.
static void run_begin(parsy_plan* pl) {
    pl->run_cursor = 0;
    pl->run_next_level_event = 0;
    pl->run_early_seen.assign(pl->ev_early_done.size(), 0);
}

static void run_range(parsy_plan* pl, const std::vector<Launch>& seq, size_t i0, size_t i1, double* L,
                      const double* Lc, double* x, int nrhs, int ldx, hipStream_t stream) {
    size_t& cursor = pl->run_cursor;
    // TILES_EARLY launches go to the side stream (unless profiling / disabled): they wait for the
    // completion event of the level two below their targets and run beside the main stream's
    // block-column chain; the matching TILES (late) launch waits for them.
    const bool overlap = pl->overlap && !pl->profile && pl->side_stream != nullptr;
    int& next_level_event = pl->run_next_level_event;  // ev_level_done[k] recorded for all k < next_level_event
    std::vector<char>& early_seen = pl->run_early_seen;
    auto record_levels_below = [&](int level) {
        for (; next_level_event < level && next_level_event < (int)pl->ev_level_done.size(); ++next_level_event)
            (void)hipEventRecord(pl->ev_level_done[next_level_event], stream);
    };
    for (size_t li = i0; li < i1; ++li) {
        const Launch& l = seq[li];
        const bool on_side = overlap && l.side;
        if (!on_side) {
            // everything enqueued so far on the main stream belongs to levels < l.level
            if (overlap && is_chol_launch(l.kind)) record_levels_below(l.level);
            // A main-stream launch that touches level t's tiles (NEXT, CHAIN) comes after every side launch
            // enqueued so far whose targets can be at level t: those with a level field <= t (a PUSH writes
            // into EVERY level from its field upwards).  The side stream runs in order, so the latest of them
            // covers the earlier ones; PUSH(t - 1) / TILES(t + 1) (field t + 1) stay free to overlap.
            if (overlap && (l.kind == kLaunchChain || l.kind == kLaunchBig || l.kind == kLaunchDense)) {
                int lw = std::min<int>(l.level, (int)early_seen.size() - 1);
                while (lw >= 0 && !early_seen[lw]) --lw;
                if (lw >= 0) (void)hipStreamWaitEvent(stream, pl->ev_early_done[lw], 0);
            }
        }
        profile_mark(pl, l.kind, stream, cursor, l.level, l.side, l.count);
        switch (l.kind) {
            case kLaunchSmall: launch_chol_small(pl->dp, l.first, l.count, l.lds_bytes, l.jb, l.fused == 2, L, stream); break;
            case kLaunchTiles:
            case kLaunchBig:
            case kLaunchDense:
                if (on_side) {
                    record_levels_below(l.wait_level + 1);
                    (void)hipStreamWaitEvent(pl->side_stream,
                                             l.wait_level >= 0 ? pl->ev_level_done[l.wait_level] : pl->ev_init, 0);
                    if (l.kind == kLaunchBig) launch_chol_big(pl->dp, l.first, l.count, L, pl->side_stream);
                    else if (l.kind == kLaunchDense) launch_chol_dense(pl->dp, l.first, l.count, L, pl->side_stream);
                    else launch_chol_tiles(pl->dp, l.first, l.count, L, pl->side_stream);
                    (void)hipEventRecord(pl->ev_early_done[l.level], pl->side_stream);
                    early_seen[l.level] = 1;
                } else if (l.kind == kLaunchBig) {
                    launch_chol_big(pl->dp, l.first, l.count, L, stream);
                } else if (l.kind == kLaunchDense) {
                    launch_chol_dense(pl->dp, l.first, l.count, L, stream);
                } else {
                    launch_chol_tiles(pl->dp, l.first, l.count, L, stream);
                }
                break;
            case kLaunchChain: launch_chol_chain(pl->dp, l.first, l.count, l.jb, pl->epoch, l.fused != 0, L, stream); break;
            case kLaunchSolveSmall:
                // (the subtree launch with many right-hand sides: a wave per subtree and 16 right-hand sides, traffic in LDS)
                if (l.fused == 2 && sub_tiers_usable(pl, nrhs, ldx) && pl->S.sub_tiers[0].ntrees == l.count)
                    launch_solve_sub_mrhs(pl->dp, pl->S.sub_tiers[0], Lc, x, nrhs, ldx, pl->solve_ldq, stream);
                else
                    launch_solve_small(pl->dp, l.first, l.count, l.jb, l.fused == 2, Lc, x, nrhs, ldx, pl->solve_ldq, stream);
                break;
            case kLaunchSolvePanel:
                if (l.fused && nrhs >= solve_mrhs_min() && !pl->old_mrhs_chain)
                    launch_solve_blocks_mrhs(pl->dp, l.lds_bytes, l.wait_level, Lc, pl->dinv, x, pl->xscratch, nrhs, ldx,
                                             pl->solve_ldq, l.jb, pl->solve_wait_bias, stream);
                else if (l.fused)
                    launch_solve_chain(pl->dp, l.first, l.count, Lc, pl->dinv, x, pl->xscratch, nrhs, ldx,
                                       pl->epoch, l.jb, pl->solve_wait_bias, stream);
                else
                    launch_solve_panel(pl->dp, l.first, l.count, Lc, x, pl->xscratch, nrhs, ldx, stream);
                break;
            case kLaunchSolveFixup:
                launch_solve_fixup(pl->dp, l.first, l.count, x, pl->xscratch, nrhs, ldx, stream);
                break;
            case kLaunchBackBelow:
                if (nrhs == 1) launch_bsolve_below(pl->dp, l.first, l.count, Lc, x, stream);
                break;
            case kLaunchBackBlock:
                if (l.fused == 1 && nrhs == 1) {   // one right-hand side: the wave dataflow over block-column pairs
                    launch_bsolve_chain_w(pl->dp, l.lds_bytes, l.wait_level, Lc, pl->dinv, x, pl->xscratch, l.jb,
                                          pl->solve_wait_bias, stream);
                    break;
                }
                if (l.fused == 2 && sub_tiers_usable(pl, nrhs, ldx) && pl->S.sub_tiers[0].ntrees == l.count) {
                    launch_bsolve_sub_mrhs(pl->dp, pl->S.sub_tiers[0], Lc, x, nrhs, ldx, stream);
                    break;
                }
                launch_bsolve_block(pl->dp, l.first, l.count, Lc, pl->dinv, x, pl->xscratch, nrhs, ldx, l.fused,
                                    l.early,
                                    l.fused == 1 ? l.jb : 0, pl->solve_wait_bias, stream);
                break;
        }
    }
}

static void run_end(parsy_plan* pl, hipStream_t stream) {
    profile_mark(pl, -1, stream, pl->run_cursor);
    if (pl->profile) pl->pev_kind.resize(pl->run_cursor);
}

static void run_launches(parsy_plan* pl, const std::vector<Launch>& seq, double* L, const double* Lc,
                         double* x, int nrhs, int ldx, hipStream_t stream) {
    run_begin(pl);
    run_range(pl, seq, 0, seq.size(), L, Lc, x, nrhs, ldx, stream);
    run_end(pl, stream);
}

// Start of a forward / backward solve: a fresh epoch range for its passes (flags of earlier solves go
// stale; on wrap-around the flags are cleared first, so that no old value can pass for a new one), its own
// status word and ticket counters zeroed.
// (one: a ONE-launch solve -- its counters follow the status word; no flags, epochs or chain tickets)
// (how many right-hand sides: 8 where the launch has at most kOneSmallBlocks blocks -- ex15-class, 8: 0.130 -> 0.087 ms --, else
// 4 -- from 6 on the level launches use the matrix cores: nd24k-class, 8: 0.75 vs 1.24 ms this way; 4: 0.81 -> 0.60 --, and 1 for
// the backward solve beside a subtree launch -- parabolic_fem-class, 4: 0.79 vs 0.86 ms -- and for the much larger plans)
static bool solve_takes_one_launch(const parsy_plan* pl, int nrhs, bool backward) {
    if (!(backward ? pl->S.solve_one_back : pl->S.solve_one)) return false;
    if (backward ? pl->one_off_back : pl->one_off) return false;   // (its buffers could not be allocated)
    const size_t nblocks = backward ? pl->S.one_back().sn.size() : pl->S.one_f.sn.size();
    // (round 5, tools/gate_sweep.py: forward blocks of 5-8 right-hand sides through the ONE launch on factors of fewer than
    // 400 entries per row -- 2-D grids of 3 400 .. 14 000 supernodes 0.26 -> 0.18, 0.33 -> 0.24, 0.35 -> 0.31 ms; 3-D grids of
    // 580-670 entries per row lose 5 %, the backward solve loses wherever the plan is not small)
    // -- and up to 4 096 blocks: the parabolic_fem-class plan, 6 546 blocks, takes 8 right-hand sides through its subtree and
    // band launches in 0.51 ms against 0.60; likewise the backward solve beside a subtree launch: 4 right-hand sides per
    // ONE launch up to that size (grids of 2 000 .. 4 000 blocks: 0.44 -> 0.34, 1.11 -> 0.93, 0.81 -> 0.68, 0.40 -> 0.26 ms), one
    // beyond it (parabolic_fem-class, 4: 0.79 vs 0.86 ms)
    const bool mid = nblocks <= (size_t)kOneMidBlocks && !pl->S.one_big;
    const bool f8 = !backward && mid && pl->S.xsize < (int64_t)kOneRhs8Density * pl->S.n;
    const int max_rhs = (pl->S.one_forced || nblocks <= (size_t)kOneSmallBlocks || f8)         ? kOneMaxRhs
                        : (pl->S.one_big || (backward && pl->S.one_subtrees && !mid))          ? 1
                                                                                               : 4;
    return nrhs <= max_rhs;
}

// The buffers of the ONE-launch solves, made by the first of them: per direction two hand-off buffers (forward: one
// slot per entry of the row-id array, backward: one per unknown; kOneMaxRhs right-hand sides), all armed, and two
// {status, ticket} pairs, zero.  Every such solve then works through buffer / pair (its direction's count & 1) and
// leaves the other one armed and zeroed for the next solve of its kind (k_solve_one, k_bsolve_one) -- one
// enqueue per solve, no memset.
static int one_begin(parsy_plan* pl, bool backward, int nrhs, hipStream_t stream, double*& y, double*& y_next, int*& st,
                     int*& st_next) {
    // (a direction's buffers hold 1, 4 or 8 right-hand sides -- nd24k-class: 12.5 MB per right-hand side for the forward
    // slots -- and are made again, larger, when a wider block comes: earlier solves on the stream are complete by then)
    const int d = backward ? 1 : 0;
    const int want = nrhs == 1 ? 1 : nrhs <= 4 ? 4 : kOneMaxRhs;
    if (want > pl->one_cap[d] && pl->one_y[d]) {
        PARSY_HIP(hipStreamSynchronize(stream));
        (void)hipFree(pl->one_y[d]);
        pl->one_y[d] = nullptr;
        pl->device_bytes -= pl->one_bytes[d];
        pl->one_bytes[d] = 0;
        if (pl->solve_status_word) pl->solve_status_word = nullptr;
    }
    if (!pl->one_y[d]) {
        pl->one_cap[d] = std::max(pl->one_cap[d], want);
        const size_t len = (size_t)(backward ? pl->S.n : pl->dp.one_f.nslots) * pl->one_cap[d];
        // (the two {status, ticket} pairs live behind the two buffers: one allocation)
        const size_t bytes = 2 * len * sizeof(double) + 4 * sizeof(int);
        if (hipMalloc((void**)&pl->one_y[d], bytes) != hipSuccess) {
            // no room for the hand-off buffers: this plan's solves of that direction go by the level launches from now on
            (void)hipGetLastError();
            pl->one_y[d] = nullptr;
            pl->one_cap[d] = 0;
            (backward ? pl->one_off_back : pl->one_off) = true;
            return 1;
        }
        pl->one_bytes[d] = (int64_t)bytes;
        pl->device_bytes += (int64_t)bytes;
        PARSY_HIP(solve_arm_handoff(pl->one_y[d], (int64_t)(2 * len), stream));
        PARSY_HIP(hipMemsetAsync(pl->one_y[d] + 2 * len, 0, 4 * sizeof(int), stream));
        pl->one_calls[d] = 0;
    }
    const size_t len = (size_t)(backward ? pl->S.n : pl->dp.one_f.nslots) * pl->one_cap[d];
    int* state = reinterpret_cast<int*>(pl->one_y[d] + 2 * len);
    const unsigned k = pl->one_calls[d]++ & 1u;
    y = pl->one_y[d] + k * len;
    y_next = pl->one_y[d] + (k ^ 1u) * len;
    st = state + 2 * k;
    st_next = state + 2 * (k ^ 1u);
    pl->solve_status_word = st;
    const char* stall = std::getenv("PARSY_DEBUG_SOLVE_STALL");
    pl->solve_wait_bias = (stall && stall[0] == '1') ? (1 << 20) : 0;
    return 0;
}

// (zero_words = false: the caller's k_solve_arm_wide launch zeroes the status word and the ticket counters)
static int solve_begin(parsy_plan* pl, int passes, hipStream_t stream, bool zero_words = true) {
    pl->solve_status_word = nullptr;
    if (pl->epoch > INT_MAX - 2 * passes - 2) {
        PARSY_HIP(hipMemsetAsync(pl->dp.flags, 0, (size_t)pl->n_flags * sizeof(int), stream));
        PARSY_HIP(hipMemsetAsync(pl->dp.tflags, 0, 2 * (size_t)std::max(pl->dp.n_tflags, 1) * sizeof(int), stream));
        pl->epoch = 0;
    }
    pl->epoch += 1;  // first pass uses this value; the kernels add the pass index
    if (zero_words) {
        PARSY_HIP(hipMemsetAsync(pl->dp.sinfo, 0, sizeof(int), stream));
        PARSY_HIP(hipMemsetAsync(pl->dp.stickets, 0, (size_t)std::max(pl->S.n_solve_chain_launches, 1) * sizeof(int),
                                 stream));
    }
    // diagnostic: PARSY_DEBUG_SOLVE_STALL=1 makes every waiter of the chain launches wait for an epoch that
    // is never published -- the timeout path of the hand-offs, exercised by the tests
    const char* st = std::getenv("PARSY_DEBUG_SOLVE_STALL");
    pl->solve_wait_bias = (st && st[0] == '1') ? (1 << 20) : 0;
    return 0;
}

int plan_backsolve(parsy_plan* pl, const double* d_L, double* d_x, int nrhs, int ldx, hipStream_t stream) {
    if (pl->device < 0) {
        set_last_error("parsy_backsolve: plan was built without a device (device < 0)");
        return -1;
    }
    if (nrhs < 1 || ldx < pl->S.n) {
        set_last_error("parsy_backsolve: need nrhs >= 1 and ldx >= n");
        return -1;
    }
    pl->levels_open = false;
    const int64_t need = (int64_t)ldx * nrhs;
    double *y = nullptr, *y_next = nullptr;
    int *st = nullptr, *st_next = nullptr;
    int one_rc = solve_takes_one_launch(pl, nrhs, true) ? one_begin(pl, true, nrhs, stream, y, y_next, st, st_next) : 1;
    if (one_rc < 0) return -1;
    if (one_rc == 0) {
        // the whole solve -- or everything above the subtree launch, which follows -- is ONE launch (k_bsolve_one)
        PARSY_HIP(hipEventRecord(pl->ev_s0, stream));
        run_begin(pl);
        profile_mark(pl, kLaunchBackBlock, stream, pl->run_cursor, 0, 0, pl->dp.one_b.nblocks);
        launch_bsolve_one(pl->dp, pl->S.n, d_L, d_x, nrhs, ldx, y, y_next, st, st_next, pl->solve_wait_bias, pl->one_cap[1], stream);
        if (pl->S.one_subtrees && pl->S.n_bsolve_subtrees > 0)
            run_range(pl, pl->S.bsolve, pl->S.bsolve.size() - 1, pl->S.bsolve.size(), nullptr, d_L, d_x, nrhs, ldx, stream);
        run_end(pl, stream);
        PARSY_HIP(hipGetLastError());
        PARSY_HIP(hipEventRecord(pl->ev_s1, stream));
        pl->have_s = true;
        return 0;
    }
    // one epoch per pass of right-hand sides (what the chain launches publish / wait for)
    const int passes = (nrhs + 3) / 4;
    // (several right-hand sides and wide supernodes: one prologue launch instead of memsets + a fill of n x nrhs entries)
    const bool arm_wide = nrhs > 1 && !pl->S.solve_wide_list.empty() && !pl->old_mrhs_chain;
    if (solve_begin(pl, passes, stream, !arm_wide) != 0) return -1;
    if (pl->S.max_width > kTile && pl->xscratch_len < need) {
        if (pl->xscratch) PARSY_HIP(hipFree(pl->xscratch));
        pl->xscratch = nullptr;
        PARSY_HIP(hipMalloc((void**)&pl->xscratch, (size_t)need * sizeof(double)));
        pl->xscratch_len = need;
    }
    if (!pl->S.solve_wide_list.empty() && !pl->dinv) {
        const size_t bytes = (size_t)std::max<int64_t>(pl->S.n_dslots, 1) * kTile * kTile * sizeof(double);
        PARSY_HIP(hipMalloc((void**)&pl->dinv, bytes));
        pl->device_bytes += (int64_t)bytes;
    }
    if (nrhs == 1 && pl->S.n_bpart_slots > 0 && !pl->dp.bpart) {
        const size_t bytes = (size_t)pl->S.n_bpart_slots * kTile * sizeof(double);
        PARSY_HIP(hipMalloc((void**)&pl->dp.bpart, bytes));
        pl->device_bytes += (int64_t)bytes;
    }
    PARSY_HIP(hipEventRecord(pl->ev_s0, stream));
    // the chain launches hand x over through xscratch itself (armed: see solve_arm_handoff) and finish a block
    // column with a product with its inverse diagonal block
    if (!pl->S.solve_wide_list.empty()) {
        if (arm_wide)
            launch_solve_arm_wide(pl->dp, (int)pl->S.solve_wide_list.size() / 2, pl->xscratch, nrhs, ldx, 0,
                                  std::max(pl->S.n_solve_chain_launches, 1), stream);
        else
            PARSY_HIP(solve_arm_handoff(pl->xscratch, need, stream));
        launch_diag_inverse(pl->dp, (int)pl->S.solve_wide_list.size() / 2, d_L, pl->dinv, stream);
    }
    if (sub_tiers_usable(pl, nrhs, ldx) && pl->S.sub_cover_level >= 0) {
        run_begin(pl);
        for (size_t li = 0; li < pl->S.bsolve.size(); ++li) {
            const Launch& l = pl->S.bsolve[li];
            if (l.fused == 2 || l.level <= pl->S.sub_cover_level) continue;
            run_range(pl, pl->S.bsolve, li, li + 1, nullptr, d_L, d_x, nrhs, ldx, stream);
        }
        for (size_t k = pl->S.sub_tiers.size(); k-- > 0;) {
            profile_mark(pl, kLaunchBackBlock, stream, pl->run_cursor, 0, 0, pl->S.sub_tiers[k].ntrees);
            launch_bsolve_sub_mrhs(pl->dp, pl->S.sub_tiers[k], d_L, d_x, nrhs, ldx, stream);
        }
        run_end(pl, stream);
    } else {
        run_launches(pl, pl->S.bsolve, nullptr, d_L, d_x, nrhs, ldx, stream);
    }
    PARSY_HIP(hipGetLastError());
    PARSY_HIP(hipEventRecord(pl->ev_s1, stream));
    pl->epoch += passes;
    pl->have_s = true;
    return 0;
}

// A solve in steps of etree levels (the exchange steps of a solve that is distributed above the cut go in between,
// as parsy_factor_level's do for the factorization): the level launches of the plan's active supernodes whose level lies
// in [lev0, lev1).  flags: bit 0 = the first step of a solve (status word, tickets, hand-off buffer, inverse diagonal
// blocks), bit 1 = the last one, bit 2 = backward (levels then go DOWN from step to step).  Level launches only: the
// caller's layout of X, no ONE launch, no bands (a rank's share of the supernodes has neither).
int plan_solve_levels(parsy_plan* pl, const double* d_L, double* d_x, int nrhs, int ldx, hipStream_t stream, int lev0,
                      int lev1, int flags) {
    if (pl->device < 0) {
        set_last_error("parsy_solve_levels: plan was built without a device (device < 0)");
        return -1;
    }
    if (nrhs < 1 || ldx < pl->S.n || lev0 > lev1) {
        set_last_error("parsy_solve_levels: need nrhs >= 1, ldx >= n and level_begin <= level_end");
        return -1;
    }
    const bool first = flags & 1, last = flags & 2, backward = flags & 4;
    if (!first && (!pl->levels_open || pl->levels_backward != backward || pl->levels_nrhs != nrhs)) {
        set_last_error("parsy_solve_levels: a step without PARSY_SOLVE_FIRST needs an open solve of the same direction and width");
        return -1;
    }
    const int passes = backward ? (nrhs + 3) / 4 : (nrhs + 7) / 8;
    const int64_t need = (int64_t)ldx * nrhs;
    if (first) {
        if (solve_begin(pl, passes, stream) != 0) return -1;
        {
            const char* e = std::getenv("PARSY_OLD_MRHS_CHAIN");
            pl->old_mrhs_chain = e && e[0] == '1';
        }
        if ((pl->S.n_solve_wide > 0 || pl->S.max_width > kTile) && pl->xscratch_len < need) {
            if (pl->xscratch) PARSY_HIP(hipFree(pl->xscratch));
            pl->xscratch = nullptr;
            PARSY_HIP(hipMalloc((void**)&pl->xscratch, (size_t)need * sizeof(double)));
            pl->xscratch_len = need;
        }
        if (!pl->S.solve_wide_list.empty() && !pl->dinv) {
            const size_t bytes = (size_t)std::max<int64_t>(pl->S.n_dslots, 1) * kTile * kTile * sizeof(double);
            PARSY_HIP(hipMalloc((void**)&pl->dinv, bytes));
            pl->device_bytes += (int64_t)bytes;
        }
        if (backward && nrhs == 1 && pl->S.n_bpart_slots > 0 && !pl->dp.bpart) {
            const size_t bytes = (size_t)pl->S.n_bpart_slots * kTile * sizeof(double);
            PARSY_HIP(hipMalloc((void**)&pl->dp.bpart, bytes));
            pl->device_bytes += (int64_t)bytes;
        }
        PARSY_HIP(hipEventRecord(pl->ev_s0, stream));
        if (pl->xscratch && (pl->S.n_solve_wide > 0 || !pl->S.solve_wide_list.empty()))
            PARSY_HIP(solve_arm_handoff(pl->xscratch, (!backward && nrhs == 1) ? (int64_t)ldx : need, stream));
        if (!pl->S.solve_wide_list.empty())
            launch_diag_inverse(pl->dp, (int)pl->S.solve_wide_list.size() / 2, d_L, pl->dinv, stream);
        pl->solve_ldq = 0;
        run_begin(pl);
        pl->levels_open = true;
        pl->levels_backward = backward;
        pl->levels_nrhs = nrhs;
    }
    const std::vector<Launch>& seq = backward ? pl->S.bsolve : pl->S.solve;
    for (size_t li = 0; li < seq.size(); ++li)
        if ((seq[li].level >= lev0 && seq[li].level < lev1) || (last && seq[li].kind == kLaunchSolveFixup))
            run_range(pl, seq, li, li + 1, nullptr, d_L, d_x, nrhs, ldx, stream);
    if (last) {
        run_end(pl, stream);
        PARSY_HIP(hipGetLastError());
        PARSY_HIP(hipEventRecord(pl->ev_s1, stream));
        pl->epoch += passes;
        pl->have_s = true;
        pl->levels_open = false;
    }
    return 0;
}

int plan_collect_profile(parsy_plan* pl) {
    // after a synchronised profiled run: add the elapsed time of each launch to its kind
    if (!pl->profile || pl->pev_kind.size() < 2) return -1;
    for (size_t i = 0; i + 1 < pl->pev_kind.size(); ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pl->pev[i], pl->pev[i + 1]) != hipSuccess) return -1;
        const int k = pl->pev_kind[i];
        if (pl->pev_ms.size() <= i) pl->pev_ms.resize(i + 1, 0.f);
        pl->pev_ms[i] = ms;
        if (k >= 0 && k < 10) {
            pl->kind_ms[k] += ms;
            pl->kind_launches[k] += 1;
        }
        // factorization launches also per level of the Cholesky view and stream (main / side)
        if (is_chol_launch(k) && i < pl->pev_level.size() && pl->pev_level[i] >= 0) {
            const size_t slot = (size_t)pl->pev_level[i];
            if (pl->level_ms.size() <= slot) pl->level_ms.resize(slot + 1, 0.0);
            pl->level_ms[slot] += ms;
        }
    }
    pl->profiled_runs += 1;
    return 0;
}

int plan_factor_begin(parsy_plan* pl, const double* d_values, double* d_L, hipStream_t stream, bool init) {
    if (pl->device < 0) {
        set_last_error("parsy_factor: plan was built without a device (device < 0)");
        return -1;
    }
    if (pl->solve_only) {
        set_last_error("parsy_factor: plan was built from L's pattern only (solve-only)");
        return -1;
    }
    const Schedule& S = pl->S;
    // flags of earlier factorizations go stale; on wrap-around they are cleared, so that a value left by an
    // early factorization cannot pass for a new one
    if (pl->epoch >= INT_MAX - 4) {
        PARSY_HIP(hipMemsetAsync(pl->dp.flags, 0, (size_t)pl->n_flags * sizeof(int), stream));
        PARSY_HIP(hipMemsetAsync(pl->dp.tflags, 0, 2 * (size_t)std::max(pl->dp.n_tflags, 1) * sizeof(int), stream));
        pl->epoch = 0;
    }
    pl->epoch += 1;
    PARSY_HIP(hipEventRecord(pl->ev_f0, stream));
    if (init) PARSY_HIP(hipMemsetAsync(d_L, 0, (size_t)S.xsize * sizeof(double), stream));
    // "no failed pivot" = 0x7f7f7f7f (kernels atomicMin the 1-based failing column into it)
    PARSY_HIP(hipMemsetAsync(pl->dp.info, 0x7f, sizeof(int), stream));
    PARSY_HIP(hipMemsetAsync(pl->dp.tickets, 0, (size_t)std::max(S.n_chain_launches, 1) * sizeof(int), stream));
    if (init) launch_scatter_a(d_values, pl->dp.a_dst, d_L, S.nnzA, stream);
    PARSY_HIP(hipEventRecord(pl->ev_init, stream));
    run_begin(pl);
    pl->factor_open = true;
    pl->factor_next_level = 0;
    return 0;
}

// The launches of levels [level0, level1) of the Cholesky view (levels must be passed in ascending order and
// without gaps: each step may enqueue side-stream launches that wait for the completion of the step before it).
int plan_factor_levels(parsy_plan* pl, int level0, int level1, double* d_L, hipStream_t stream) {
    const Schedule& S = pl->S;
    if (!pl->factor_open) {
        set_last_error("parsy_factor_level: no factorization is open (call parsy_factor_begin first)");
        return -1;
    }
    if (level0 != pl->factor_next_level || level1 < level0 || level1 > S.cnlevels) {
        set_last_error("parsy_factor_level: levels must be enqueued in ascending order without gaps");
        return -1;
    }
    if (level1 > level0)
        run_range(pl, S.chol, S.chol_level_begin[(size_t)level0], S.chol_level_begin[(size_t)level1], d_L, d_L, nullptr,
                  0, 0, stream);
    // (profiling: what the caller does between two steps is not a launch's time)
    profile_mark(pl, -1, stream, pl->run_cursor);
    pl->factor_next_level = level1;
    PARSY_HIP(hipGetLastError());
    return 0;
}

int plan_factor_end(parsy_plan* pl, hipStream_t stream) {
    if (!pl->factor_open) {
        set_last_error("parsy_factor_end: no factorization is open");
        return -1;
    }
    if (pl->factor_next_level != pl->S.cnlevels) {
        set_last_error("parsy_factor_end: not every level has been enqueued");
        return -1;
    }
    run_end(pl, stream);
    // the side stream's last launches belong to this factorization: the caller's stream waits for them
    if (pl->overlap && !pl->profile && pl->side_stream != nullptr) {
        for (size_t k = pl->run_early_seen.size(); k-- > 0;)
            if (pl->run_early_seen[k]) {
                (void)hipStreamWaitEvent(stream, pl->ev_early_done[k], 0);
                break;
            }
    }
    PARSY_HIP(hipGetLastError());
    PARSY_HIP(hipEventRecord(pl->ev_f1, stream));
    pl->have_f = true;
    pl->factor_open = false;
    return 0;
}

// Error path of a caller that enqueues a factorization step by step: whatever was enqueued is allowed to finish
// (caller's stream and the plan's side stream), then the plan is closed again so that a later factorization starts
// clean.  The factor in d_L is partial.
void plan_factor_abort(parsy_plan* pl, hipStream_t stream) {
    if (!pl) return;
    (void)hipStreamSynchronize(stream);
    if (pl->side_stream != nullptr) (void)hipStreamSynchronize(pl->side_stream);
    (void)hipGetLastError();
    if (pl->factor_open) run_end(pl, stream);
    pl->factor_open = false;
    pl->factor_next_level = 0;
}

int plan_factor(parsy_plan* pl, const double* d_values, double* d_L, hipStream_t stream, bool init) {
    if (plan_factor_begin(pl, d_values, d_L, stream, init) != 0) return -1;
    if (plan_factor_levels(pl, 0, pl->S.cnlevels, d_L, stream) != 0) return -1;
    return plan_factor_end(pl, stream);
}

int plan_solve(parsy_plan* pl, const double* d_L, double* d_x, int nrhs, int ldx,
               hipStream_t stream) {
    if (pl->device < 0) {
        set_last_error("parsy_solve: plan was built without a device (device < 0)");
        return -1;
    }
    if (nrhs < 1 || ldx < pl->S.n) {
        set_last_error("parsy_solve: need nrhs >= 1 and ldx >= n");
        return -1;
    }
    pl->levels_open = false;   // (a solve in steps of levels that was never finished is abandoned)
    double *y = nullptr, *y_next = nullptr;
    int *st = nullptr, *st_next = nullptr;
    int one_rc = solve_takes_one_launch(pl, nrhs, false) ? one_begin(pl, false, nrhs, stream, y, y_next, st, st_next) : 1;
    if (one_rc < 0) return -1;
    if (one_rc == 0) {
        // the whole solve -- or everything above the subtree launch, which comes first -- is ONE launch (k_solve_one)
        PARSY_HIP(hipEventRecord(pl->ev_s0, stream));
        run_begin(pl);
        pl->solve_ldq = 0;
        if (pl->S.one_subtrees && pl->S.n_solve_subtrees > 0) run_range(pl, pl->S.solve, 0, 1, nullptr, d_L, d_x, nrhs, ldx, stream);
        profile_mark(pl, kLaunchSolveSmall, stream, pl->run_cursor, 0, 0, pl->dp.one_f.nblocks);
        launch_solve_one(pl->dp, d_L, d_x, nrhs, ldx, y, y_next, st, st_next, pl->solve_wait_bias, pl->one_cap[0], stream);
        run_end(pl, stream);
        PARSY_HIP(hipGetLastError());
        PARSY_HIP(hipEventRecord(pl->ev_s1, stream));
        pl->have_s = true;
        return 0;
    }
    // one epoch per pass of right-hand sides (what the chain kernel publishes / waits for)
    const int passes = (nrhs + 7) / 8;
    {
        const char* e = std::getenv("PARSY_OLD_MRHS_CHAIN");
        pl->old_mrhs_chain = e && e[0] == '1';
    }
    // (several right-hand sides on the armed-buffer chain launches: one prologue launch, below)
    const bool arm_wide = nrhs >= solve_mrhs_min() && !pl->old_mrhs_chain && pl->S.n_solve_wide > 0 &&
                          !pl->S.solve_wide_list.empty() && pl->S.solve_fix_list.empty();
    if (solve_begin(pl, passes, stream, !arm_wide) != 0) return -1;
    // (many right-hand sides: k_solve_blocks_mrhs on the armed buffer; PARSY_OLD_MRHS_CHAIN=1: the flag protocol of rounds 1-2)
    {
        const char* e = std::getenv("PARSY_OLD_MRHS_CHAIN");
        pl->old_mrhs_chain = e && e[0] == '1';
    }
    // From 16 right-hand sides on the solve works on X with the right-hand sides of a row contiguous (transposed in
    // and out: both kernels a solve of that many takes -- k_solve_small_mrhs, k_solve_blocks_mrhs -- read and write
    // whole rows then; a supernode's x block was 64 eight-byte pieces per column).  PARSY_XT_MIN=k (0: never).
    const char* xt_env = std::getenv("PARSY_XT_MIN");
    const int xt_min = xt_env && *xt_env ? std::atoi(xt_env) : 16;
    // (measured, 64 right-hand sides: Flan-class 23.1 -> 20.4 ms, nd24k-class 1.31 -> 1.19 ms; parabolic_fem-class 1.72 ->
    // 1.75 ms -- its factor has 67 entries per row and the two transposes, 0.3 ms, cost what the kernels gain: the layout
    // is taken from 200 entries of L per row on)
    const bool use_xt = xt_min > 0 && nrhs >= xt_min && nrhs >= solve_small_mrhs_min() && nrhs >= solve_mrhs_min() &&
                        !pl->old_mrhs_chain && pl->S.solve_fix_list.empty() &&
                        (pl->S.xsize >= (int64_t)150 * pl->S.n || (xt_env && *xt_env));
    const int ldq = use_xt ? (nrhs + 15) & ~15 : 0;
    const int64_t need = use_xt ? (int64_t)pl->S.n * ldq : (int64_t)ldx * nrhs;
    if (pl->S.n_solve_wide > 0 && pl->xscratch_len < need) {
        // grows only when a larger right-hand-side block shows up (not per call)
        if (pl->xscratch) PARSY_HIP(hipFree(pl->xscratch));
        pl->xscratch = nullptr;
        PARSY_HIP(hipMalloc((void**)&pl->xscratch, (size_t)need * sizeof(double)));
        pl->xscratch_len = need;
    }
    if (use_xt && pl->xt_len < need) {
        if (pl->xt) PARSY_HIP(hipFree(pl->xt));
        pl->xt = nullptr;
        PARSY_HIP(hipMalloc((void**)&pl->xt, (size_t)need * sizeof(double)));
        pl->xt_len = need;
    }
    if (!pl->S.solve_wide_list.empty() && !pl->dinv) {
        const size_t bytes = (size_t)std::max<int64_t>(pl->S.n_dslots, 1) * kTile * kTile * sizeof(double);
        PARSY_HIP(hipMalloc((void**)&pl->dinv, bytes));
        pl->device_bytes += (int64_t)bytes;
    }
    PARSY_HIP(hipEventRecord(pl->ev_s0, stream));
    // the chain launches hand x over through xscratch itself (one right-hand side: k_solve_chain_w; many:
    // k_solve_blocks_mrhs): every entry holds the armed pattern when the solve starts
    if (nrhs == 1 && pl->S.n_solve_wide > 0) PARSY_HIP(solve_arm_handoff(pl->xscratch, ldx, stream));
    else if (arm_wide)
        launch_solve_arm_wide(pl->dp, (int)pl->S.solve_wide_list.size() / 2, pl->xscratch, nrhs, ldx, ldq,
                              std::max(pl->S.n_solve_chain_launches, 1), stream);
    else if (nrhs >= solve_mrhs_min() && !pl->old_mrhs_chain && pl->S.n_solve_wide > 0)
        PARSY_HIP(solve_arm_handoff(pl->xscratch, need, stream));
    launch_diag_inverse(pl->dp, (int)pl->S.solve_wide_list.size() / 2, d_L, pl->dinv, stream);
    pl->solve_ldq = ldq;
    if (use_xt) launch_transpose_x(d_x, ldx, pl->xt, ldq, pl->S.n, nrhs, true, stream);
    if (sub_tiers_usable(pl, nrhs, ldx) && pl->S.sub_cover_level >= 0) {
        // the bands of levels that are one launch each (tiers), then the level launches above them
        double* xw = use_xt ? pl->xt : d_x;
        run_begin(pl);
        for (const SubTier& T : pl->S.sub_tiers) {
            profile_mark(pl, kLaunchSolveSmall, stream, pl->run_cursor, 0, 0, T.ntrees);
            launch_solve_sub_mrhs(pl->dp, T, d_L, xw, nrhs, ldx, ldq, stream);
        }
        for (size_t li = 0; li < pl->S.solve.size(); ++li) {
            const Launch& l = pl->S.solve[li];
            if (l.fused == 2 || l.level <= pl->S.sub_cover_level) continue;
            run_range(pl, pl->S.solve, li, li + 1, nullptr, d_L, xw, nrhs, ldx, stream);
        }
        run_end(pl, stream);
    } else {
        run_launches(pl, pl->S.solve, nullptr, d_L, use_xt ? pl->xt : d_x, nrhs, ldx, stream);
    }
    if (use_xt) launch_transpose_x(d_x, ldx, pl->xt, ldq, pl->S.n, nrhs, false, stream);
    pl->solve_ldq = 0;
    PARSY_HIP(hipGetLastError());
    PARSY_HIP(hipEventRecord(pl->ev_s1, stream));
    pl->epoch += passes;
    pl->have_s = true;
    return 0;
}

}  // namespace parsy
