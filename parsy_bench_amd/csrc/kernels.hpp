// Launch wrappers of the HIP kernels (chol_kernels.hip, trsv_kernels.hip).
// All pointers are device pointers; launches are asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "schedule.hpp"

namespace parsy {

constexpr int kPassLanes = 8;    // passes over the right-hand sides of a solve that run side by side

struct DevicePattern {           // device copies of Schedule arrays
    const SnDesc* sn = nullptr;           // the supernodes (solve kernels)
    const SnDesc* csn = nullptr;          // Cholesky view (pieces of the very wide ones): Cholesky kernels
    const UpdDesc* upd = nullptr;
    const int32_t* relpos = nullptr;
    const int64_t* a_dst = nullptr;
    const int32_t* rows = nullptr;
    const WaveEntry* wave_entries = nullptr;  // update streams of the tile kernel
    const int64_t* wave_ptr = nullptr;
    const int64_t* split_ranges = nullptr;     // shares of the wave lists of tiles whose early stream is split
    double* tile_scratch = nullptr;            // partial tiles of the split streams
    const int32_t* small_list = nullptr;
    const int32_t* small_ranges = nullptr;        // subtree launches: (begin, end) pairs into the lists,
    const int32_t* solve_small_ranges = nullptr;  // one per workgroup
    const int32_t* bsolve_ranges = nullptr;
    const TileDesc* tiles = nullptr;
    const WaveEntry* big_entries = nullptr;   // BIG launches: (source, row window, column window) per task
    const TileDesc* big_tasks = nullptr;
    const int32_t* solve_small_list = nullptr;
    const PanelDesc* solve_panels = nullptr;
    const PanelDesc* solve_mtasks = nullptr;
    const int32_t* solve_fix_list = nullptr;
    const int32_t* solve_wide_list = nullptr;   // (supernode, block column) pairs: diagonal blocks to invert
    const PanelDesc* bsolve_pairs = nullptr;   // backward chain launches, one right-hand side: block-column pairs
    const PanelDesc* bsolve_below = nullptr;   // k_bsolve_below tasks
    double* bpart = nullptr;                   // their partial sums (64 doubles per slot; allocated by the first backward solve)
    const PanelDesc* bsolve_blocks = nullptr;  // backward solve: (supernode, block column) per workgroup  // supernodes solved by SOLVE_CHAIN (need inverse blocks)
    // the solves' subtree launch for many right-hand sides (Schedule::sub_*; sub_ntrees == 0: not available)
    const SubMember* sub_members = nullptr;
    const SubTree* sub_trees = nullptr;
    const uint16_t* sub_slots = nullptr;
    const int32_t* sub_out_rows = nullptr;
    int sub_ntiers = 0;
    int* info = nullptr;         // first failed pivot column + 1 (0x7f7f7f7f = none, < 0: wait timed out)
    int* flags = nullptr;        // solve chain: per block column, epoch of the pass that published it
    int* tflags = nullptr;       // Cholesky chain: per tile, epoch of the factorization that published it
                                 // ([0, n_tflags): finished tiles, [n_tflags, 2 n_tflags): tiles prepared for the walker)
    int n_tflags = 0;
    int* tickets = nullptr;      // one counter per CHAIN launch (zeroed at the start of a factorization)
    int flag_stride = 0;         // flags holds kPassLanes sets of this many entries (one per lane of passes)
    int* sinfo = nullptr;        // status of the last solve: 0 ok, < 0 a hand-off wait timed out (own word: a solve
                                 // never touches the factorization's status)
    int* stickets = nullptr;     // one counter per chain launch of the forward / backward solve
    // ONE-launch solves (Schedule::OneLists; one_b = one_f where the backward solve shares the forward lists)
    struct OneDev {
        const SnDesc* sn = nullptr;          // the block columns (<= 64 columns each) in ticket order
        const int64_t* slot0 = nullptr;      // the first hand-off slot of each
        const int32_t* wleft = nullptr;      // the columns of its supernode from its first column on
        const int32_t* pull_ptr = nullptr;   // per block its gather list [ptr[p], ptr[p + 1]) of
        const int32_t* pull_slot = nullptr;  // ... (slot of the hand-off buffer,
        const int32_t* pull_pos = nullptr;   //      column of the block)
        int nblocks = 0;
        int64_t nslots = 1;
    } one_f, one_b;
};

// lValues[a_dst[q]] = values[q]
void launch_scatter_a(const double* values, const int64_t* a_dst, double* L, int64_t nnz,
                      hipStream_t stream);
void launch_chol_small(const DevicePattern& P, int first, int count, int lds_bytes, int stage_cap, bool subtrees,
                       double* L, hipStream_t stream);
void launch_chol_big(const DevicePattern& P, int first, int count, double* L, hipStream_t stream);
void launch_chol_dense(const DevicePattern& P, int first, int count, double* L, hipStream_t stream);
void launch_chol_tiles(const DevicePattern& P, int first, int count, double* L, hipStream_t stream);
int chain_workgroups_per_cu();
void launch_chol_chain(const DevicePattern& P, int first, int count, int ticket, int epoch, bool rows, double* L,
                       hipStream_t stream);

void launch_solve_small(const DevicePattern& P, int first, int count, int wmax, bool subtrees, const double* L,
                        double* x, int nrhs, int ldx, int ldq, hipStream_t stream);
// (y: the hand-off buffer armed for this solve -- forward: nslots x cap, backward: n x cap values, cap = 1, 4 or 8 right-hand sides --,
// y_next: the one this solve arms for the next of its kind; state / state_next: {status, ticket} likewise)
void launch_solve_one(const DevicePattern& P, const double* L, double* x, int nrhs, int ldx,
                      double* y, double* y_next, int* state, int* state_next, int wait_bias, int cap, hipStream_t stream);
void launch_bsolve_one(const DevicePattern& P, int n, const double* L, double* x, int nrhs, int ldx,
                       double* y, double* y_next, int* state, int* state_next, int wait_bias, int cap, hipStream_t stream);
void launch_solve_panel(const DevicePattern& P, int first, int count, const double* L, double* x,
                        double* xscratch, int nrhs, int ldx, hipStream_t stream);
void launch_solve_chain(const DevicePattern& P, int first, int count, const double* L, const double* dinv,
                        double* x, double* xscratch, int nrhs, int ldx, int epoch0, int ticket, int wait_bias,
                        hipStream_t stream);
void launch_solve_blocks_mrhs(const DevicePattern& P, int first, int count, const double* L, const double* dinv,
                              double* x, double* xscratch, int nrhs, int ldx, int ldq, int ticket, int wait_bias,
                              hipStream_t stream);
void launch_transpose_x(double* x, int64_t ldx, double* xt, int64_t ldq, int n, int nrhs, bool to_rows, hipStream_t stream);
// the subtree launch of a solve with many right-hand sides: one wave per (subtree, 16 right-hand sides), the subtree's
// traffic in LDS (trsv_sub_kernels.hip); ldq > 0: X row-major with that row stride
void launch_solve_sub_mrhs(const DevicePattern& P, const SubTier& T, const double* L, double* x, int nrhs, int ldx, int ldq,
                           hipStream_t stream);
void launch_bsolve_sub_mrhs(const DevicePattern& P, const SubTier& T, const double* L, double* x, int nrhs, int ldx,
                            hipStream_t stream);
int solve_sub_prepare(int max_slots);   // once per plan: LDS beyond 64 KB per workgroup needs the kernels' attribute (-1: refused)
int solve_sub_mrhs_min();   // right-hand sides from which those kernels take the subtree launches (PARSY_SUB_MRHS_MIN; 0: never)
int solve_mrhs_min();
int solve_small_mrhs_min();
hipError_t solve_arm_handoff(double* xscratch, int64_t n, hipStream_t stream);
// several right-hand sides: status word + ticket counters zeroed and the hand-off buffer armed at the wide supernodes' columns
void launch_solve_arm_wide(const DevicePattern& P, int npairs, double* xscratch, int nrhs, int ldx, int ldq, int ntickets,
                           hipStream_t stream);
void launch_diag_inverse(const DevicePattern& P, int count, const double* L, double* dinv,
                         hipStream_t stream);
void launch_bsolve_chain_w(const DevicePattern& P, int first, int count, const double* L, const double* dinv,
                           double* x, double* xscratch, int ticket, int wait_bias, hipStream_t stream);
void launch_bsolve_below(const DevicePattern& P, int first, int count, const double* L, const double* x, hipStream_t stream);
void launch_bsolve_block(const DevicePattern& P, int first, int count, const double* L, const double* dinv,
                         double* x, double* xscratch, int nrhs, int ldx, int mode, int tiny, int ticket,
                         int wait_bias, hipStream_t stream);
void launch_rhs_ones(const DevicePattern& P, int nsuper, int max_rows, const double* L, double* b,
                     hipStream_t stream);
void launch_copy_segments(double* dst, const double* src, const int64_t* dst_off, const int64_t* src_off,
                          const int32_t* len, int64_t nseg, hipStream_t stream);
void launch_solve_fixup(const DevicePattern& P, int first, int count, double* x,
                        const double* xscratch, int nrhs, int ldx, hipStream_t stream);

}  // namespace parsy
