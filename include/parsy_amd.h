/*
 * parsy_amd.h -- C ABI of the MI355X-native supernodal Cholesky + BCSC
 * triangular-solve executor (libparsy_amd.so).
 *
 * Section 1 are DROP-IN replacements: same names, argument lists, ownership
 * and return values as the reference's executors, so the reference's drivers
 * (examples/choleskyTest01.cpp, choleskyTest03.cpp, triangularTest02.cpp) can
 * link this library instead of including the OpenMP headers.  All pointers
 * are caller-owned HOST pointers; the library uploads the pattern once per
 * distinct pointer set (cached plan) and runs every numeric step in HIP
 * kernels on the selected device.  There is no CPU fallback: when no HIP
 * device is usable the functions fail loudly (stderr + false / 0).
 *
 * Section 2 is the plan (handle) API the drop-ins are built on: pattern
 * resident in HBM, device pointers in / out, caller-supplied stream.
 *
 * Section 3/4 are host-side helpers (inspector, synthetic matrices) so the
 * path can be exercised without the reference's inspector.
 *
 * Plain C: pointers and sizes only, no C++ or torch types.
 */
#ifndef PARSY_AMD_H
#define PARSY_AMD_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* 1. Drop-in operators                                                      */
/* ------------------------------------------------------------------------ */

/* Replaces cholesky_left_par_05, reference cholesky/parallel_PB_Cholesky_05.h:27-39.
 *   n            matrix order
 *   c, r, values CSC of the LOWER triangle of P A P' (reference's A2)
 *   lC           value offset of every column of L (size n+1)          [L->p]
 *   lR           sorted row ids of every supernode (size ssize)        [L->s]
 *   Li_ptr       offset into lR for every column (size n+1)            [L->i_ptr]
 *   lValues      OUT: BCSC values (size lC[n]); must be zeroed by the caller
 *   blockSet     first column of every supernode (size supNo+1)        [L->super]
 *   timing       OUT: [0] seconds of the level-parallel phase, [1] seconds of
 *                the root phase (the GPU schedule has no separate root phase:
 *                [1] = 0), [2] device seconds of the numeric kernels only
 *   aTree        supernodal etree (-1 root)                            [L->sParent]
 *   cT, rT       CSC pattern of the UPPER triangle of P A P' (reference's A1)
 *   col2Sup      column -> supernode
 *   nLevels, levelPtr, levelSet, nPar, parPtr, partition: H-level schedule of the
 *                reference's inspector.  The factor is schedule independent; the
 *                GPU executor validates the partition (every supernode exactly
 *                once) and schedules by etree level itself.
 *   chunk, threads, super_max, col_max, nodCost: accepted for signature
 *                compatibility, unused (the reference ignores chunk/nodCost too).
 * Returns false iff a diagonal block is not positive definite (lValues is then
 * partial) or the device path could not run. */
bool cholesky_left_par_05(int n, int* c, int* r, double* values, size_t* lC, int* lR,
                          size_t* Li_ptr, double* lValues, int* blockSet, int supNo,
                          double* timing, int* aTree, int* cT, int* rT, int* col2Sup,
                          int nLevels, int* levelPtr, int* levelSet, int nPar, int* parPtr,
                          int* partition, int chunk, int threads, int super_max, int col_max,
                          double* nodCost);

/* The reference's PRUNE build of the same operator (parallel_PB_Cholesky_05.h:27-39 with
 * `#define PRUNE`, :30-34 and :120-123): the update lists are passed instead of the etree and the
 * upper pattern -- prunePtr (supNo+1) / pruneSet from getBlockedPruneSet
 * (cholesky/Inspection_Prune.h).  A C ABI cannot overload, hence the suffix. */
bool cholesky_left_par_05_prune(int n, int* c, int* r, double* values, size_t* lC, int* lR,
                                size_t* Li_ptr, double* lValues, int* blockSet, int supNo,
                                double* timing, int* prunePtr, int* pruneSet, int nLevels,
                                int* levelPtr, int* levelSet, int nPar, int* parPtr, int* partition,
                                int chunk, int threads, int super_max, int col_max, double* nodCost);

/* Replaces cholesky_left_par_waveFront, reference
 * cholesky/Parallel_PB_Cholesky_wavefront.h:10-20.  levelPtr/levelSet are etree
 * level sets (common/TreeUtils.h:119).  Like the reference it reports `true`
 * unless the device path could not run; a non-positive pivot is reported on
 * stderr only (the reference ignores LAPACK's info here, :137). */
bool cholesky_left_par_waveFront(int n, int* c, int* r, double* values, size_t* lC, int* lR,
                                 size_t* Li_ptr, double* lValues, int* blockSet, int supNo,
                                 double* timing, int* aTree, int* cT, int* rT, int* col2Sup,
                                 int nLevels, int* levelPtr, int* levelSet, int chunk,
                                 int threads, int super_max, int col_max);

/* Replace the BCSC forward solves L x = b (x in/out, one right-hand side),
 * reference triangularSolve/Triangular_BCSC.h:14 / :115 / :171 / :238.
 * Return 1 on success, 0 when Lp, Li or x is NULL (as the reference) or when
 * the device path could not run.  NNZ, col2sup and chunk are unused, as in the
 * reference.  All four run the same level-scheduled HIP solve; the level /
 * partition arrays are validated, not interpreted. */
int blockedLsolve(int n, size_t* Lp, int* Li, double* Lx, int NNZ, size_t* Li_ptr,
                  int* col2sup, int* sup2col, int supNo, double* x);
int leveledBlockedLsolve(int n, size_t* Lp, int* Li, double* Lx, int NNZ, size_t* Li_ptr,
                         int* col2sup, int* sup2col, int supNo, double* x, int levels,
                         int* levelPtr, int* levelSet, int chunk);
int H2LeveledBlockedLsolve(int n, size_t* Lp, int* Li, double* Lx, int NNZ, size_t* Li_ptr,
                           int* col2sup, int* sup2col, int supNo, double* x, int levels,
                           int* levelPtr, int* levelSet, int parts, int* parPtr,
                           int* partition, int chunk);
int H2LeveledBlockedLsolve_Peeled(int n, size_t* Lp, int* Li, double* Lx, int NNZ,
                                  size_t* Li_ptr, int* col2sup, int* sup2col, int supNo,
                                  double* x, int levels, int* levelPtr, int* levelSet,
                                  int parts, int* parPtr, int* partition, int chunk,
                                  int threads);

/* Drop the plans cached by the drop-in operators (frees device memory). */
void parsy_dropin_reset(void);

/* ------------------------------------------------------------------------ */
/* 2. Plan API (pattern resident on the device)                              */
/* ------------------------------------------------------------------------ */

typedef struct parsy_plan parsy_plan;
/* A plan owns scratch that its launches share (tile flags and ticket counters of the factorization; inverse
 * diagonal blocks, the hand-off vector and ticket counters of the solves): calls on ONE plan must be issued on
 * one stream, or be separated by the caller (events / synchronisation) -- one factorization or solve of a plan in
 * flight at a time.  Independent work in flight takes one plan each (plans of the same pattern are cheap next to
 * the factor: DESIGN.md, `throughput_in_flight`).  The drop-in operators of section 1 hold a per-plan mutex across
 * a call; the plan API itself is not thread-safe per plan (two host threads must not use one plan at once). */

/* Sizes and work counts of a plan (all exact, from the pattern). */
typedef struct parsy_plan_info {
    int32_t n, nsuper, nlevels, max_width, max_rows;
    int32_t n_small, n_big;        /* supernodes on the LDS-resident / tiled path */
    int32_t chol_launches;         /* kernel launches per factorization */
    int32_t solve_launches;        /* kernel launches per forward solve */
    int64_t nnzA, ssize, xsize, nnzL;
    int64_t n_updates;             /* (target, descendant) pairs */
    int64_t relpos_len;            /* relative-index entries */
    int64_t device_bytes;          /* pattern + schedule bytes resident in HBM */
    double flops_stored;           /* executed flops on the stored structure */
    double update_flops;           /* flops in the SYRK/GEMM update kernels */
    double reread_bytes;           /* left-looking re-read traffic 8*sum K*nSupRs */
    double inner_flops;            /* in-supernode SYRK/GEMM flops of the tiled path */
    double tile_update_flops;      /* external-update flops applied by the tile kernel's wave streams */
    double big_flops;              /* update flops applied by the BIG (LDS-staged GEMM) launches */
    int64_t big_entries;           /* (tile, source) pairs of the BIG launches */
    int32_t big_tasks;             /* workgroups of the BIG launches per factorization */
    int32_t n_pieces;              /* supernodes of the Cholesky view (very wide ones cut into pieces) */
    int32_t chol_levels;           /* levels of the Cholesky view's (chain-extended) etree */
    int32_t piece_width, big_min_k;
    /* subtree launches: etree subtrees of narrow supernodes walked by one workgroup each (the reference's
     * w-partitions, parallel_PB_Cholesky_05.h:66-84 / Triangular_BCSC.h:171-232) and the supernodes in them */
    int32_t chol_subtrees, chol_subtree_supernodes;
    int32_t solve_subtrees, solve_subtree_supernodes;
    int32_t backsolve_launches;    /* kernel launches per backward solve */
    int32_t dense_tasks;           /* workgroups of the DENSE launches (k_chol_dense) per factorization */
    double dense_flops;            /* part of big_flops in dense entries (full 128 x 128 blocks below the diagonal): k_chol_dense */
    int64_t dense_entries;
    int32_t solve_one;             /* a small plan -- blocks of at most 8 right-hand sides are solved in ONE launch instead of
                                    * one per level (k_solve_one / k_bsolve_one): bit 0 the forward, bit 1 the backward solve;
                                    * bit 2: the subtree launch of the narrow supernodes stays beside it */
    int32_t solve_one_blocks;      /* block columns (workgroups) of the forward ONE launch */
    int32_t sub_mrhs_trees;        /* subtrees of the solves' subtree launch for many right-hand sides (one wave per subtree and
                                    * 16 right-hand sides, the subtree's traffic in LDS); 0: that form is not available */
    int32_t sub_mrhs_slots;        /* ... and the LDS slots (16 right-hand sides each) of the largest */
    int32_t sub_mrhs_tiers;        /* launches of that form per solve: the subtrees at the bottom of the etree, then bands of levels */
    int32_t sub_mrhs_cover_level;  /* ... which replace every level launch up to this etree level (-1: only the subtree launch) */
    int64_t dense_strip_entries;   /* dense entries that carry the remainder of their source's row / column run (<= 16 rows behind
                                    * or beside the block), multiplied with the block's staged operands */
} parsy_plan_info;

/* Build a plan from the reference-shaped symbolic arrays (host pointers, copied).
 * device < 0 builds the host schedule only (no HIP call; for CPU-side tests).
 * Returns NULL on error (see parsy_last_error). */
parsy_plan* parsy_plan_create(int n, int supNo, const int* blockSet, const size_t* lC,
                              const size_t* Li_ptr, const int* lR, const int* aTree,
                              const int* col2Sup, const int* cT, const int* rT, const int* c,
                              const int* r, int device);
void parsy_plan_destroy(parsy_plan* plan);
int parsy_plan_get_info(const parsy_plan* plan, parsy_plan_info* info);

/* Restrict the work of the NEXT factor/solve calls to a set of supernodes
 * (multi-GPU: the subtrees this rank owns, or the root part).  mask has one
 * byte per supernode (non-zero = process).  NULL restores "all". Rebuilds the
 * launch schedule; not meant for the timed region. */
int parsy_plan_set_active(parsy_plan* plan, const uint8_t* mask);
/* Host-side dry run of the factorization's in-launch hand-offs (tiles of the wide supernodes are
 * passed between workgroups inside one launch per etree level) with `slots` resident workgroups:
 * returns the number of tiles that would never be finished -- 0 means the schedule cannot
 * deadlock at that residency -- or -1 for a bad argument.  The kernel runs 2 workgroups per CU. */
long long parsy_plan_chain_check(const parsy_plan* plan, int slots);
/* Host-side consistency check of the plan's schedules -- factorization: pieces, levels, windows of the update
 * entries, exact cover of every (target, descendant) update, subtree launches, launch order; solves: every active
 * supernode / chunk / block column solved exactly once in each direction, subtree runs in order, width classes,
 * the backward chain's groups --: number of violations, 0 = consistent (parsy_last_error describes the first
 * one).  For tests and for callers that build plans from their own arrays. */
long long parsy_plan_check(const parsy_plan* plan);

/* Numeric factorization, everything on the device.
 *   d_values  device, nnz(A2) doubles (same order as the host `values`)
 *   d_lValues device, xsize doubles; zeroed and overwritten
 *   stream    hipStream_t (NULL = default stream); the call is asynchronous
 * Returns 0 when the launches were enqueued, <0 on a HIP error. */
int parsy_factor_device(parsy_plan* plan, const double* d_values, double* d_lValues,
                        void* stream);
/* As parsy_factor_device with flags: PARSY_FACTOR_NO_INIT skips the zero-fill and the
 * A scatter (multi-GPU: the root-part pass on a buffer that already holds A and the
 * gathered subtree panels). */
#define PARSY_FACTOR_NO_INIT 1
int parsy_factor_device_ex(parsy_plan* plan, const double* d_values, double* d_lValues,
                           void* stream, int flags);
/* After the stream has been synchronised: 0 = factor ok, k > 0 = first
 * non-positive pivot seen at (1-based) column k, as LAPACK's dpotrf info. */
int parsy_factor_status(parsy_plan* plan);

/* ---- Level-by-level factorization and the pieces of the Cholesky view -------------------------------
 * The factorization runs on the "Cholesky view" of the pattern: the supernodes, with the very wide ones
 * cut into pieces (column ranges of one panel) that are factored one after the other; its etree levels are the
 * wavefronts of cholesky_left_par_waveFront (Parallel_PB_Cholesky_wavefront.h:28-45: one level at a time).
 * A caller that has work of its own between two levels -- the exchange step of a multi-device run -- enqueues
 * the levels one by one:  parsy_factor_begin, parsy_factor_level(0 .. chol_levels-1, ascending, no gaps),
 * parsy_factor_end; parsy_factor_device is exactly that sequence.  Everything is asynchronous on `stream`. */
int parsy_factor_begin(parsy_plan* plan, const double* d_values, double* d_lValues, void* stream, int flags);
int parsy_factor_level(parsy_plan* plan, int level, double* d_lValues, void* stream);
int parsy_factor_end(parsy_plan* plan, void* stream);
/* The pieces (n_pieces of parsy_plan_info; any output array may be NULL): supernode, level in the Cholesky
 * view, first column, width, rows (from the piece's diagonal down), and the range [value_begin, value_end) of
 * lValues that holds the piece's columns (column-major with the leading dimension of its supernode). Returns
 * the number of pieces. */
int parsy_plan_pieces(const parsy_plan* plan, int32_t* supernode, int32_t* level, int32_t* col0, int32_t* width,
                      int32_t* rows, int64_t* value_begin, int64_t* value_end);
/* As parsy_plan_set_active for the factorization, at the granularity of pieces (one byte per piece); the
 * solves keep the supernode mask.  NULL restores "all". */
int parsy_plan_set_active_pieces(parsy_plan* plan, const uint8_t* piece_mask);

/* ---- Distribution of one factorization over the devices of a node (host logic) -----------------------
 * Below a cut whole etree subtrees go to one rank each (independent: a target only reads its descendants,
 * common/Reach.h:122-135 -- the independence of the reference's w-partitions, InspectionLevel_06.h:196-217;
 * packed heaviest first, cf. common/TreeUtils.h:217-255); above it the pieces are dealt over the ranks level
 * by level.  The owner of a piece applies every update into it; after a level is complete each of its pieces
 * travels to the ranks that own a target it updates (only the rows those targets read).  The messages are
 * lists of (offset, length) segments of lValues, the same on the sending and the receiving side. */
typedef struct parsy_dist parsy_dist;
typedef struct parsy_dist_info {
    int32_t nranks, nlevels, n_pieces, n_subtrees, n_root_pieces, n_messages;
    int64_t exchange_elements;        /* doubles that travel per factorization, all messages */
    double total_cost, root_cost;     /* flops: whole job / pieces above the cut */
    double max_rank_cost;             /* the most loaded rank */
    double lockstep_cost;             /* sum over levels of the most loaded rank of that level */
} parsy_dist_info;
/* block: consecutive pieces of one split supernode that stay on one rank (<= 0: the default, 2). The plan may be
 * a host-only one (device < 0). */
parsy_dist* parsy_dist_create(const parsy_plan* plan, int nranks, int block);
void parsy_dist_destroy(parsy_dist* dist);
int parsy_dist_get_info(const parsy_dist* dist, parsy_dist_info* info);
/* owner[n_pieces]; in_subtree[n_pieces] (1: the piece lies below the cut, in a subtree that went to one rank as a
 * whole; 0: above it); rank_cost[nranks]; level_cost[nlevels * nranks] (any may be NULL) */
int parsy_dist_get(const parsy_dist* dist, int32_t* owner, uint8_t* in_subtree, double* rank_cost,
                   double* level_cost);
/* Messages that follow level `level`: their number; message `index` of them (borrowed pointers, valid until
 * the dist is destroyed): sender, receiver, number of segments, elements of the packed buffer, and per segment
 * its offset in lValues, its length and its offset in the packed buffer. */
int parsy_dist_level_messages(const parsy_dist* dist, int level);
int parsy_dist_message(const parsy_dist* dist, int level, int index, int32_t* src, int32_t* dst, int64_t* nseg,
                       int64_t* total, const int64_t** off, const int32_t** len, const int64_t** packed);
/* Host-side consistency check (every update across ranks finds its source rows delivered right after the
 * source's level; subtrees walked by one workgroup stay on one rank): number of violations. */
long long parsy_dist_check(const parsy_plan* plan, const parsy_dist* dist);

/* Forward solve L X = B in place, X is n x nrhs column-major with leading
 * dimension ldx (device pointers). Asynchronous on `stream`. */
int parsy_solve_device(parsy_plan* plan, const double* d_lValues, double* d_x, int nrhs,
                       int ldx, void* stream);

/* After the stream has been synchronised: 0 = the last forward / backward solve on this plan completed,
 * -1 = a hand-off wait inside one of its chain launches timed out (internal error; x is not the solution).
 * The host conveniences and the drop-in solves check it themselves and fail loudly. */
int parsy_solve_status(parsy_plan* plan);

/* Backward solve L' X = Y in place (same layout as parsy_solve_device).  Not part of the
 * reference (it only has the forward solve): forward + backward solve A x = b
 * for the permuted system, x = P' L'^-1 L^-1 P b. Asynchronous on `stream`. */
int parsy_backsolve_device(parsy_plan* plan, const double* d_lValues, double* d_x, int nrhs, int ldx,
                           void* stream);
/* A forward / backward solve in STEPS of etree levels (triangularSolve/Triangular_BCSC.h:115-164: one level set at a time):
 * the level launches of the plan's active supernodes (parsy_plan_set_active) whose etree level lies in
 * [level_begin, level_end).  A solve that is distributed above the cut puts its exchange steps between two calls, as
 * parsy_factor_level does for the factorization.  flags: */
/* The etree level of every supernode (nsuper entries; NULL: only the count) as the solves' launches go by it; returns the
 * number of levels.  A supernode's rows below its own columns belong to supernodes of higher levels. */
int parsy_plan_solve_levels(const parsy_plan* plan, int32_t* level);
#define PARSY_SOLVE_FIRST 1     /* the first step of a solve: status word, hand-off buffer, inverse diagonal blocks */
#define PARSY_SOLVE_LAST 2      /* the last step */
#define PARSY_SOLVE_BACKWARD 4  /* L' x = y (the steps then go from the root level down) */
/* X keeps the caller's layout (n x nrhs, leading dimension ldx); asynchronous on `stream`; parsy_solve_status as usual. */
int parsy_solve_levels_device(parsy_plan* plan, const double* d_lValues, double* d_x, int nrhs, int ldx, void* stream,
                              int level_begin, int level_end, int flags);
/* d_b = L * 1 on the stored structure (device pointers, n doubles, overwritten): the right-hand side the
 * reference's triangularTest solves (rhsInitBlocked, common/Util.h:277-288), so that L x = b has x = 1. */
int parsy_rhs_ones_device(parsy_plan* plan, const double* d_lValues, double* d_b, void* stream);
/* Device helper of the multi-GPU exchange: d_dst[dst_off[q] + i] = d_src[src_off[q] + i], i < len[q], for nseg
 * contiguous runs (all pointers device pointers).  Packs the panel rows of a subtree that the root part
 * reads into one send buffer, and unpacks them on the receiving rank. Asynchronous on `stream`. */
int parsy_copy_segments_device(double* d_dst, const double* d_src, const int64_t* d_dst_off,
                               const int64_t* d_src_off, const int32_t* d_len, int64_t nseg, void* stream);
/* Host convenience: forward (if `forward` != 0) then backward solve on host buffers. */
int parsy_solve2_host(parsy_plan* plan, const double* lValues, double* x, int nrhs, int ldx,
                      int forward, double* seconds);

/* Host-buffer conveniences (H2D + kernels + D2H, synchronous). `seconds`, if
 * non-NULL, receives the device time of the numeric kernels alone. */
int parsy_factor_host(parsy_plan* plan, const double* values, double* lValues, double* seconds);
int parsy_solve_host(parsy_plan* plan, const double* lValues, double* x, int nrhs, int ldx,
                     double* seconds);

/* Device time (ms, hipEvents on the launch stream) of the last factor / solve
 * enqueued through the *_device calls, after synchronisation; <0 if none. */
double parsy_last_factor_ms(parsy_plan* plan);
double parsy_last_solve_ms(parsy_plan* plan);

/* Per-launch timing with hipEvents on the launch stream (adds two event records per
 * kernel launch, so keep it out of throughput measurements).
 *   parsy_plan_profile(plan, 1) on, (plan, 0) off, (plan, 2) on + reset accumulators.
 *   parsy_plan_profile_collect(plan): after the stream is synchronised, add the
 *     elapsed time of every launch of the last factor/solve to its kernel kind.
 *   parsy_plan_profile_get: accumulated ms and launch counts per kind (PARSY_PROFILE_KINDS = 10 entries:
 *     0 SMALL, 1 TILES, 2 CHAIN, 3 BIG, 4 unused, 5 SOLVE_SMALL, 6 SOLVE_PANEL, 7 SOLVE_FIXUP,
 *     8 BACK_BLOCK, 9 unused; the arrays passed must hold 10 entries) and the number of collected runs. */
#define PARSY_PROFILE_KINDS 10
int parsy_plan_profile(parsy_plan* plan, int enable);
int parsy_plan_profile_collect(parsy_plan* plan);
int parsy_plan_profile_get(parsy_plan* plan, double* kind_ms, int* kind_launches, int* runs);
/* The factorization launches of the collected runs per level of the Cholesky view: accumulated ms of the launches
 * that run on the caller's stream (main_ms[chol_levels]) and of those that belong to the plan's side stream
 * (side_ms[level] = the side launches whose targets start at that level).  Returns chol_levels. */
int parsy_plan_profile_levels(parsy_plan* plan, double* main_ms, double* side_ms);

/* Thread-local message of the last failing call. */
const char* parsy_last_error(void);

/* Number of visible HIP devices (0 when none / runtime unusable). */
int parsy_device_count(void);

/* ------------------------------------------------------------------------ */
/* 3. Inspector (host)                                                       */
/* ------------------------------------------------------------------------ */

typedef struct parsy_symbolic parsy_symbolic;

/* Borrowed pointers into a parsy_symbolic (valid until it is freed).
 * Names follow the reference's BCSC (common/def.h:117-204). */
typedef struct parsy_symbolic_view {
    int32_t n, nsuper, nlevels, maxSupWid, maxCol;
    int64_t ssize, xsize, nnzL, nnzA, n_updates;
    double flops_colcount, flops_stored;
    const int* Perm;       /* n, new -> old */
    const int* Parent;     /* n, column etree */
    const int* ColCount;   /* n */
    const int* super;      /* nsuper+1 */
    const int* col2Sup;    /* n */
    const int* sParent;    /* nsuper */
    const size_t* p;       /* n+1 */
    const size_t* i_ptr;   /* n+1 */
    const int* s;          /* ssize */
    const int* A1p; const int* A1i;                       /* upper PAP' pattern */
    const int* A2p; const int* A2i; const double* A2x;    /* lower PAP' */
    const int* A2src;      /* nnzA: position of each A2 entry in the input Ax */
    const int* levelPtr;   /* nlevels+1 */
    const int* levelSet;   /* nsuper */
    const int64_t* updPtr; /* nsuper+1 */
    const int* updSn; const int* updLb; const int* updUb; /* n_updates each */
} parsy_symbolic_view;

/* Ap/Ai/Ax: lower triangle, CSC, sorted (what common/Util.h:77 reads).
 * perm: n entries new->old or NULL.  nrelax/zrelax: 3 entries each or NULL for
 * the reference driver's defaults {4,16,48} / {0.8,0.1,0.05}. */
parsy_symbolic* parsy_analyze(int n, const int* Ap, const int* Ai, const double* Ax,
                              const int* perm, const int* nrelax, const double* zrelax);
void parsy_symbolic_free(parsy_symbolic* sym);
int parsy_symbolic_get(const parsy_symbolic* sym, parsy_symbolic_view* view);
/* Plan straight from an inspector result. */
parsy_plan* parsy_plan_from_symbolic(const parsy_symbolic* sym, int device);

/* ------------------------------------------------------------------------ */
/* 4. Synthetic SPD matrices / orderings (host)                              */
/* ------------------------------------------------------------------------ */

/* Lower triangle of a grid stencil matrix (5/9-point 2-D with nz = 1, 7/27-point
 * 3-D): off-diagonals -1, diagonal = degree + shift.  Call with Ai = Ax = NULL to
 * get the entry count; Ap needs nx*ny*nz+1 ints. Returns nnz or <0. */
int64_t parsy_grid_spd_lower(int nx, int ny, int nz, int stencil, double shift, int* Ap,
                             int* Ai, double* Ax);
/* Geometric nested dissection, perm[new] = old, nx*ny*nz entries. */
int parsy_grid_nested_dissection(int nx, int ny, int nz, int leaf, int* perm);
/* Fill-reducing ordering of an arbitrary symmetric pattern (lower triangle, CSC): nested dissection
 * on the graph with level-structure separators (the reference calls METIS, cholesky/LSparsity.h:
 * not in this build).  leaf <= 0 picks the default (64).  perm[new] = old, n entries.  0 on success. */
int parsy_order_nd(int n, const int* Ap, const int* Ai, int leaf, int* perm);

/* ------------------------------------------------------------------------ */
/* 5. One process, several devices: the distributed factorization            */
/* ------------------------------------------------------------------------ */

/* The distribution of section 2 run by ONE process that drives every device (the drivers' path; bench.py runs the
 * same distribution with one process per GPU and RCCL point-to-point messages).
 * `devices` lists one HIP device per rank; a device may appear more than once (several ranks share it: how the
 * path is exercised on a node with fewer GPUs).  Every rank keeps its own lValues; finished pieces travel as
 * device-to-device copies (peer access over xGMI between different devices) of exactly the segments of the
 * parsy_dist messages, ordered by events.  values: host, nnz(A2) doubles.
 * STATUS: verified bit for bit with several ranks sharing ONE device (tests/test_gpu_parity.py::test_distributed_*).
 * The distinct-device branch -- hipDeviceEnablePeerAccess, hipStreamWaitEvent on another device's event, the copy
 * kernel reading a peer's buffer -- has NOT run on hardware yet: no node with two GPUs was available to the builder
 * (tests/test_multigpu_gpu.py::test_two_distinct_devices_* run it wherever two devices are visible). */
typedef struct parsy_mg parsy_mg;
parsy_mg* parsy_mg_create(const parsy_symbolic* sym, int nranks, const int* devices, int block);
void parsy_mg_destroy(parsy_mg* mg);
int parsy_mg_set_values(parsy_mg* mg, const double* values);
/* Enqueue one distributed factorization on every rank's stream and wait for it; seconds (may be NULL) = wall
 * time from the first enqueue to the last rank's completion.  Returns 0, or the factorization status of the
 * first rank that reports one (> 0: non-positive pivot column, < 0: error). */
int parsy_mg_factor(parsy_mg* mg, double* seconds);
/* One factorization with every launch timed by itself: the ranks take turns (a rank's step of a level runs alone
 * on its device, also when ranks share one), per-launch hipEvents as parsy_plan_profile.  main_ms / side_ms:
 * nranks x chol_levels (row = rank) ms of the main-stream and side-stream launches per level; copy_ms: nranks x
 * chol_levels ms of the copies a rank receives after a level.  What a rank would spend on a device of its own,
 * for the model of DESIGN.md section 6.  Returns as parsy_mg_factor. */
int parsy_mg_profile(parsy_mg* mg, double* main_ms, double* side_ms, double* copy_ms);
/* Per rank: device milliseconds of its last factorization (hipEvents on its stream). rank_ms[nranks]. */
int parsy_mg_rank_ms(parsy_mg* mg, double* rank_ms);
/* Collect the factor on the host: every piece from its owner (lValues: xsize doubles). */
int parsy_mg_gather_host(parsy_mg* mg, double* lValues);
const parsy_dist* parsy_mg_dist(const parsy_mg* mg);
parsy_plan* parsy_mg_plan(parsy_mg* mg, int rank);

#ifdef __cplusplus
}
#endif
#endif /* PARSY_AMD_H */
