// Device-resident plan: pattern + schedule in HBM, launch sequencing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "schedule.hpp"

struct parsy_plan {
    parsy::Schedule S;
    int device = -1;          // < 0: host schedule only
    std::mutex use_mu;        // the drop-in operators hold it across a call: flags, tickets, status and the
                              // host-convenience buffers of a cached plan belong to one call at a time
    bool solve_only = false;  // built from L's pattern alone (no A, no update lists)
    bool sn_mask_set = false, piece_mask_set = false;  // parsy_plan_set_active / _set_active_pieces are in force

    // pattern arrays (uploaded once) and launch arrays (re-uploaded by set_active)
    std::vector<void*> owned;        // every hipMalloc'd block, for destroy()
    std::vector<void*> launch_owned; // the launch-array blocks
    parsy::DevicePattern dp;
    int64_t device_bytes = 0;
    int epoch = 0;            // factorization counter (value the chain launches publish / wait for)
    int64_t n_flags = 1;      // entries of dp.flags
    int solve_wait_bias = 0;  // diagnostic (PARSY_DEBUG_SOLVE_STALL): added to the epoch the solve's waiters wait for

    double* dinv = nullptr;       // inverse 64x64 diagonal blocks of the wide supernodes (solve)
    double* xscratch = nullptr;
    int64_t xscratch_len = 0;
    // ONE-launch solves (Schedule::solve_one): per direction two hand-off buffers (a solve works through the one its
    // predecessor armed and arms the other: no memset between solves) and two {status, ticket} pairs, used in turn
    double* one_y[2] = {nullptr, nullptr};   // forward, backward: two buffers + the two state pairs behind them
    unsigned one_calls[2] = {0, 0};
    int one_cap[2] = {0, 0};                 // right-hand sides the buffers are made for (1, 4 or 8: grown on demand)
    int64_t one_bytes[2] = {0, 0};           // ... and their share of device_bytes
    bool one_off = false, one_off_back = false;   // a direction whose buffers could not be allocated: level launches from then on
    const int* solve_status_word = nullptr;   // where the last solve left its status (null: dp.sinfo)

    // buffers of the host-convenience calls
    double* h_values_dev = nullptr;
    double* h_L_dev = nullptr;
    double* h_x_dev = nullptr;
    // host-buffer factorization with the download behind the kernels (capi_exec.hip, parsy_factor_host): its own
    // streams, one event per band of levels, the (offset, length) runs of lValues that are final after each band
    hipStream_t h_stream = nullptr, h_copy = nullptr;
    std::vector<hipEvent_t> h_band_ev;
    bool h_ready = false;                    // the pipelined download's streams, bands and events all exist
    bool levels_open = false, levels_backward = false;   // a solve in steps of levels (plan_solve_levels) is under way
    int levels_nrhs = 0;
    std::vector<int> h_band_level;                                    // last level of every band
    std::vector<std::vector<std::pair<int64_t, int64_t>>> h_band_runs;
    int64_t h_x_len = 0;

    // side stream of the TILES_EARLY launches + the events that order it against the main stream
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_init = nullptr;
    std::vector<hipEvent_t> ev_level_done, ev_early_done;
    bool overlap = true;      // PARSY_NO_OVERLAP=1 (or profiling) runs everything on the caller's stream

    // state of the launch sequence being enqueued (executor.hip run_range): persistent so that a factorization
    // can be enqueued level by level
    size_t run_cursor = 0;
    int run_next_level_event = 0;
    std::vector<char> run_early_seen;
    bool factor_open = false;
    int factor_next_level = 0;

    hipEvent_t ev_f0 = nullptr, ev_f1 = nullptr, ev_s0 = nullptr, ev_s1 = nullptr;
    bool have_f = false, have_s = false;

    // optional per-launch profiling (hipEvents on the launch stream)
    bool profile = false;
    double* xt = nullptr;           // X with the right-hand sides of a row contiguous (forward solves with many of them)
    int64_t xt_len = 0;
    int solve_ldq = 0;              // > 0: the running solve works on xt with this row stride
    bool old_mrhs_chain = false;    // PARSY_OLD_MRHS_CHAIN=1: forward chain launches with many right-hand sides by flags
    std::vector<hipEvent_t> pev;
    std::vector<int> pev_kind;
    std::vector<int> pev_level;     // per mark: level << 1 | side of a factorization launch (-1: other)
    std::vector<int> pev_count;     // per mark: work items of the launch
    std::vector<float> pev_ms;      // per mark: elapsed time in the last collected run (diagnostics)
    std::vector<double> level_ms;   // accumulated ms per (level << 1 | side)
    double kind_ms[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int kind_launches[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int profiled_runs = 0;
};

namespace parsy {

// Build a plan. A2p/A2i may be null => solve-only plan. device < 0 => no HIP calls.
parsy_plan* plan_build(const PatternRef& P, const size_t* lC, const int* A2p, const int* A2i,
                       int device);
void plan_free(parsy_plan* plan);
int plan_upload_launches(parsy_plan* plan);
int plan_factor(parsy_plan* plan, const double* d_values, double* d_L, hipStream_t stream,
                bool init = true);
int plan_factor_begin(parsy_plan* plan, const double* d_values, double* d_L, hipStream_t stream, bool init);
int plan_factor_levels(parsy_plan* plan, int level0, int level1, double* d_L, hipStream_t stream);
int plan_factor_end(parsy_plan* plan, hipStream_t stream);
void plan_factor_abort(parsy_plan* plan, hipStream_t stream);   // error path: drain, close the open factorization
int plan_solve(parsy_plan* plan, const double* d_L, double* d_x, int nrhs, int ldx,
               hipStream_t stream);
int plan_backsolve(parsy_plan* plan, const double* d_L, double* d_x, int nrhs, int ldx, hipStream_t stream);
int plan_solve_levels(parsy_plan* plan, const double* d_L, double* d_x, int nrhs, int ldx, hipStream_t stream, int lev0,
                      int lev1, int flags);
int plan_collect_profile(parsy_plan* plan);

}  // namespace parsy
