// HIP kernels of the supernodal left-looking Cholesky for gfx950 (MI355X).
//
// Numeric contract (reference cholesky/parallel_PB_Cholesky_05.h:96-219): for a
// target supernode, panel := A(:, cols); for every descendant d in update order:
// panel(rel(i), rel(j)) -= sum_k Ld(i,k) Ld(j,k) for j < ndrow1, i >= j; then
// POTRF on the diagonal block and TRSM on the rows below it.  FP64 throughout.
//
// Data layout in HBM: lValues is the reference's BCSC value array (one
// column-major rows x width panel per supernode, leading dimension = rows);
// descriptors are the flat arrays of schedule.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>

#include <climits>

#include "kernels.hpp"

namespace parsy {

typedef double double4_t __attribute__((ext_vector_type(4)));

static constexpr int kThreads = 256;
static constexpr int kLdSub = kSub + 1;   // padded leading dimension of a wave's sub-tile in LDS
static constexpr int kLdDiag = kTile + 1; // padded leading dimension of a diagonal block in LDS

#ifdef PARSY_BIGSTAMPS
// diagnostic build only (tools/big_timeline.py): shader-clock stamps of the phases of k_chol_big's chunk loop, for
// workgroup g_bigstamp_cfg[1] of the launch whose grid has g_bigstamp_cfg[0] workgroups: per chunk 8 stamps of waves 0
// and 5 (loop top, staged, fetched, multiplied, tile updated, barrier passed, fragments, k of the entry)
__device__ unsigned long long g_bigtrace[2 * 8 * 1024];
__device__ int g_bigstamp_cfg[2];
#define BIGSTAMP(i) do { if (st_on) g_bigtrace[((size_t)st_slot * 1024 + st_n) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define BIGSTAMP(i) do { } while (0)
#endif

#ifdef PARSY_STAMPS
// diagnostic build only: phase stamps (100 MHz wall clock) of the last workgroup that factored
// the last diagonal tile of a supernode; read back with parsy_debug_stamps().
__device__ unsigned long long g_stamps[32];
#define STAMP(i) do { if (threadIdx.x == 0) g_stamps[i] = wall_clock64(); } while (0)
// per block column J of the last supernode that ran: times of the diagonal tile (J,J) and the tile below it
__device__ unsigned long long g_trace[16 * 512];
__device__ unsigned long long g_probe[8];
#define TRACE(J, i) do { if (threadIdx.x == 0 && (J) < 512) g_trace[(J) * 16 + (i)] = wall_clock64(); } while (0)
// sums over the ordinary tiles of the chain launches (100 MHz ticks): phases 0..5 = tile load + descendants' stream,
// own block columns, wait for the diagonal tile, its load, TRSM, write + publish; [8] = tiles, [9] = block columns
__device__ unsigned long long g_tilephase[16];
#define TPHASE(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); \
                       atomicAdd(&g_tilephase[i], now_ - tp_last); tp_last = now_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#define TRACE(J, i) do { } while (0)
#define TPHASE(i) do { } while (0)
#endif

// ---------------------------------------------------------------------------
// A -> L scatter (reference :104-112, hoisted: the destination of every entry
// is precomputed).  HBM-bound, 16 B read + 8 B written per entry.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_scatter_a(const double* __restrict__ values,
                                                        const int64_t* __restrict__ a_dst,
                                                        double* __restrict__ L, int64_t nnz) {
    for (int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x; q < nnz;
         q += (int64_t)gridDim.x * kThreads)
        L[a_dst[q]] = values[q];
}

void launch_scatter_a(const double* values, const int64_t* a_dst, double* L, int64_t nnz,
                      hipStream_t stream) {
    if (nnz <= 0) return;
    int64_t blocks = (nnz + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_scatter_a, dim3((unsigned)blocks), dim3(kThreads), 0, stream, values, a_dst,
                       L, nnz);
}

// ---------------------------------------------------------------------------
// POTRF of a 64x64 diagonal block by one 256-thread workgroup, register-blocked: thread
// (ti, tj) = (tid & 15, tid >> 4) keeps the 4x4 block rows 4ti.., cols 4tj.. in registers, so wave q
// holds block columns 4q..4q+3 (all 64 rows).  Four columns per step tjj:
//   (1) the wave that holds block column tjj takes the 4x4 diagonal micro-block from its owner lane
//       (v_readlane: the values become wave-uniform) and factors it (4 pivots, redundantly);
//   (2) the threads of block column tjj solve their own 4x4 block against it and publish the 4 new
//       columns in LDS (two buffers, alternating: ONE workgroup barrier per step);
//   (3) everybody applies the rank-4 update to its block -- the wave that holds the next block column
//       then goes straight on to (1) while the others are still updating.
// The published columns are laid out so that the 16 lanes along ti read / write consecutive 16-byte
// pairs (rows 4t+2h, 4t+2h+1 of column k at pair index (2k + h) * 16 + t): no bank conflicts.
// inv_out (64 doubles of LDS, optional) receives the reciprocals of the factor's diagonal.
// Entries outside the block (nb < 64) must be an identity so the loop is uniform; strictly-upper
// entries pick up garbage that is never stored.  `scr` is kPotrfScratch doubles of LDS, 16-byte
// aligned.  On return a[][] holds the factor; `bad` receives (1-based) the first column whose pivot
// was not positive.  Pivots use rsqrt (1/sqrt(d) directly: the divide-after-sqrt chain is the
// critical path of every block step).
// ---------------------------------------------------------------------------
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) double lds_f64;
static constexpr int kPotrfScratch = 2 * 4 * kTile;
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane),
                            __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}
__device__ __forceinline__ void potrf64_regs(double (&a)[4][4], double* __restrict__ scr, int ti, int tj,
                                             int nb, int& bad, double* __restrict__ inv_out = nullptr) {
    const bool lower = ti >= tj;
    bad = 0;
#ifdef PARSY_STAMPS
#define PPROBE(i) do { if (tjj == 4 && ti == 6 && tj == 4) g_probe[i] = clock64(); } while (0)
#else
#define PPROBE(i) do { } while (0)
#endif
    const int nsteps = (nb + 3) / 4;  // panels beyond the block are an identity: nothing to do
    for (int tjj = 0; tjj < nsteps; ++tjj) {
        double2_t* __restrict__ pub = reinterpret_cast<double2_t*>(scr) + (tjj & 1) * (2 * kTile);
        PPROBE(0);
        if ((tj >> 2) == (tjj >> 2)) {  // wave-uniform: the wave of block column tjj
            const int src = __builtin_amdgcn_readfirstlane((tjj & 3) * 16 + tjj);  // lane of thread (tjj, tjj)
            double l[4][4], inv[4];
            int nonpos = 0;  // pivots of this step that were not positive (bit jj)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                double d = readlane_f64(a[jj][jj], src);
#pragma unroll
                for (int k = 0; k < jj; ++k) d = fma(-l[jj][k], l[jj][k], d);
                nonpos |= (d > 0.0) ? 0 : (1 << jj);
                // 1/sqrt(d): hardware estimate + one correction step (the library's sequence without its
                // fix-ups for zero / infinite arguments: such a pivot is reported, its factor is not used)
                const double y0 = __builtin_amdgcn_rsq(d);
                const double e = fma(-d * y0, y0, 1.0);
                inv[jj] = fma(y0 * e, fma(e, 0.375, 0.5), y0);
                l[jj][jj] = d * inv[jj];
#pragma unroll
                for (int ii = jj + 1; ii < 4; ++ii) {
                    double v = readlane_f64(a[ii][jj], src);
#pragma unroll
                    for (int k = 0; k < jj; ++k) v = fma(-l[ii][k], l[jj][k], v);
                    l[ii][jj] = v * inv[jj];
                }
            }
            PPROBE(1);
            if (nonpos && bad == 0 && tj == tjj && lower) {
                const int jj = __builtin_ctz(nonpos);
                if (4 * tjj + jj < nb) bad = 4 * tjj + jj + 1;
            }
            if (tj == tjj && lower) {
                if (ti == tjj) {
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                        for (int ci = 0; ci <= ri; ++ci) a[ri][ci] = l[ri][ci];
                    if (inv_out) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) inv_out[4 * tjj + jj] = (4 * tjj + jj < nb) ? inv[jj] : 1.0;
                    }
                } else {
                    // X l' = A  (4x4, row by row)
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            double v = a[ri][jj];
#pragma unroll
                            for (int k = 0; k < jj; ++k) v = fma(-a[ri][k], l[jj][k], v);
                            a[ri][jj] = v * inv[jj];
                        }
                }
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int h = 0; h < 2; ++h) pub[(2 * ci + h) * 16 + ti] = double2_t{a[2 * h][ci], a[2 * h + 1][ci]};
            }
        }
        PPROBE(2);
        __syncthreads();
        PPROBE(3);
        if (lower && tj > tjj) {
            double li[4][4], lc[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double2_t x = pub[(2 * k + h) * 16 + ti], y = pub[(2 * k + h) * 16 + tj];
                    li[2 * h][k] = x[0];
                    li[2 * h + 1][k] = x[1];
                    lc[2 * h][k] = y[0];
                    lc[2 * h + 1][k] = y[1];
                }
#pragma unroll
            for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[ri][ci] = fma(-li[ri][k], lc[ci][k], a[ri][ci]);
        }
        PPROBE(4);
    }
}

// Step (a) of potrf64_panel, by ONE wave, out of line: the sweep keeps ~30 wave-uniform multipliers in scalar
// registers at a time, which the walker around it has none to spare for (inlined, the compiler parked them in
// vector lanes: 1 000 extra lane moves per panel).  Returns the first column (1-based, within the tile) of the panel
// whose pivot was not positive, or 0.
__device__ __noinline__ int potrf64_panel_sweep(lds_f64* __restrict__ Cb, int c0, int nb, lds_f64* __restrict__ inv_out,
                                                lds_f64* __restrict__ scr) {
    const int lane = threadIdx.x & 63;
    // lane l holds row c0 + l of the tile (the rows above the panel have nothing in it), so that the pivot of
    // column j sits in lane j and the multiplier of column c in lane c: constant lane numbers
    const int row = c0 + lane;
    const bool live = row < nb;          // (rows past nb: an identity, never stored)
    // column c0 + j of this lane's row: j * kLdSub doubles from `mine` (a panel lies inside one 32-column half)
    const int rowc = min(row, kTile - 1);
    lds_f64* __restrict__ mine = Cb + ((rowc >> 5) * 2 + (c0 >> 5)) * (kSub * kLdSub) + (c0 & 31) * kLdSub + (rowc & 31);
    double x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double v = mine[j * kLdSub];
        x[j] = (lane >= j && live && c0 + j < nb) ? v : ((lane == j) ? 1.0 : 0.0);
    }
    double mydiag = 1.0, myinv = 1.0;    // lane j: L[j][j] and its reciprocal
    // Column j: pivot and the multiplier of column j + 1 -- the critical chain -- reach the lanes as wave-uniform
    // values (v_readlane); the multipliers of the later columns go through 16 doubles of LDS (lanes 0..15 park the
    // scaled column, everybody reads the ones it needs two at a time as broadcasts): a third of the instructions of
    // reading every multiplier lane by lane, and a single wave is bound by the instructions it issues.
    typedef __attribute__((address_space(3))) double2_t lds_f64x2;
    lds_f64x2* __restrict__ bc = (lds_f64x2*)scr;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double d = readlane_f64(x[j], j);
        // 1/sqrt(d): hardware estimate + one correction step (as potrf64_regs); a pivot that is not positive
        // leaves a NaN on the diagonal, which is how it is found below
        const double y0 = __builtin_amdgcn_rsq(d);
        const double e = fma(-d * y0, y0, 1.0);
        const double inv = fma(y0 * e, fma(e, 0.375, 0.5), y0);
        const double ljj = d * inv;
        mydiag = (lane == j) ? ljj : mydiag;
        myinv = (lane == j) ? inv : myinv;
        x[j] = (lane == j) ? ljj : x[j] * inv;
        if (j + 1 < 16) {
            if (j + 2 < 16 && lane < 16) scr[lane] = x[j];
            const double l1 = readlane_f64(x[j], j + 1);
            x[j + 1] = fma(-x[j], l1, x[j + 1]);
            // (pairs (c, c + 1) from an even c on: 16-byte broadcast reads)
#pragma unroll
            for (int c = (j + 2) & ~1; c < 16; c += 2) {
                const double2_t l = bc[c >> 1];
                if (c >= j + 2) x[c] = fma(-x[j], l[0], x[c]);
                x[c + 1] = fma(-x[j], l[1], x[c + 1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // (columns one after the other: hoisted reads cost registers the caller must save)
    }
    const bool in_block = lane < 16 && row < nb;
    const unsigned long long badmask = __ballot(in_block && !(mydiag > 0.0));
    if (lane < 16) inv_out[row] = in_block ? myinv : 1.0;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (lane >= j && live && c0 + j < nb) mine[j * kLdSub] = x[j];
    return badmask != 0 ? c0 + (int)__builtin_ctzll(badmask) + 1 : 0;
}

// ---------------------------------------------------------------------------
// POTRF of a 64x64 diagonal tile held in LDS (sub-tile layout of the tile kernel: 4 x 32x33, lower triangle valid), in
// place, by 16-column panels (the walker of the chain launch; reference dpotrf at parallel_PB_Cholesky_05.h:204):
//   (a) wave 0 factors the panel with ONE ROW PER LANE: lane i keeps its 16 entries of the panel in registers; per
//       column the pivot and the multipliers of the later columns reach all lanes as wave-uniform values
//       (v_readlane), so a column costs one pivot chain + (15 - j) independent multiply-adds per lane, and the rows
//       below the diagonal block are solved by the same instructions (the TRSM of the panel comes for free) -- no
//       barrier, no LDS traffic inside a panel;
//   (b) the panel goes back to LDS, one barrier;
//   (c) the four waves subtract P_i P_j' from the 16x16 blocks to the right of the panel (v_mfma_f64_16x16x4_f64,
//       K = 16: four products per block, at most two blocks per wave), one barrier.
// 4 barrier pairs and 64 pivots in sequence instead of 16 steps of (4 pivots + 4x4 solve + publish + barrier + rank-4
// update) over the 256 threads' register blocks (potrf64_regs above, still the SMALL kernel's).  Rows / columns past
// nb are treated as an identity.  `bad` receives (1-based, lane 0 of wave 0 only) the first column whose pivot was
// not positive; inv_out (64 doubles of LDS) the reciprocals of the factor's diagonal; scr: 16 doubles of LDS, 16-byte
// aligned.  A workgroup barrier must precede the call; one ends it.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void potrf64_panel(double* __restrict__ Cb, int nb, int& bad, double* __restrict__ inv_out,
                                              double* __restrict__ scr) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    auto cell = [&](int i, int c) -> double& {
        return Cb[((i >> 5) * 2 + (c >> 5)) * (kSub * kLdSub) + (c & 31) * kLdSub + (i & 31)];
    };
    bad = 0;
    const int np = (nb + 15) >> 4;
    for (int p = 0; p < np; ++p) {
        const int c0 = 16 * p;
        if (wave == 0) {
            const int pb = potrf64_panel_sweep((lds_f64*)Cb, c0, nb, (lds_f64*)inv_out, (lds_f64*)scr);
            if (lane == 0 && bad == 0) bad = pb;   // (the first of the whole block: later panels see its NaNs)
        }
        __syncthreads();
        // blocks (bi, bj), p < bj <= bi < np, in the order (p+1,p+1), (p+2,p+1), (p+2,p+2), ...: block q to wave q % 4
        int q = 0;
        for (int bi = p + 1; bi < np; ++bi)
            for (int bj = p + 1; bj <= bi; ++bj, ++q) {
                if ((q & 3) != wave) continue;
                double4_t acc = {0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double av = cell(16 * bi + l15, c0 + kq + 4 * u);
                    const double bv = cell(16 * bj + l15, c0 + kq + 4 * u);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 16 * bi + kq + 4 * v, c = 16 * bj + l15;
                    if (i >= c && i < nb) cell(i, c) -= acc[v];   // (rows past nb may be rows of the panel below the block)
                }
            }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// SMALL: one workgroup per supernode, whole panel resident in LDS.  The updates are a
// stream of (descendant, k-slice) blocks -- the rows of the descendant from `lb` down, at
// most kSmallStage doubles -- pumped through a double-buffered LDS stage by all threads
// (coalesced along rows, two blocks in flight in registers) while the pair products of the
// staged block are subtracted from the panel; one barrier per block, fixed order.
// ---------------------------------------------------------------------------
static constexpr int kSmallStage = 2048;                       // doubles per staged block
static constexpr int kSmallPerThread = kSmallStage / kThreads;  // 8
static constexpr int kSmallRelCap = 256;                       // rows of a descendant block staged with indices

__global__ __launch_bounds__(kThreads) void k_chol_small(const SnDesc* __restrict__ sn,
                                                         const UpdDesc* __restrict__ upd,
                                                         const int32_t* __restrict__ relpos,
                                                         const int32_t* __restrict__ list,
                                                         const int32_t* __restrict__ ranges,
                                                         double* __restrict__ L,
                                                         int* __restrict__ info, int stage_cap) {
    // stage_cap (<= kSmallStage, host-chosen per launch): doubles per staged block
    // ranges != null (subtree launch): the workgroup factors list[ranges[2b] .. ranges[2b+1]) one after the
    // other -- a subtree of the etree in index order, so every descendant of a supernode has been stored by
    // this same workgroup (same CU, same L1: ordered by the barrier that ends each supernode)
    extern __shared__ __attribute__((aligned(16))) double P[];  // panel, then 2 stages, then 2 index rings
    const int tid = threadIdx.x;
    const int q_begin = ranges ? ranges[2 * blockIdx.x] : (int)blockIdx.x;
    const int q_end = ranges ? ranges[2 * blockIdx.x + 1] : q_begin + 1;
  for (int qsn = q_begin; qsn < q_end; ++qsn) {
    const SnDesc D = sn[list[qsn]];
    const int r = D.r, w = D.w, total = r * w;
    double* __restrict__ G = L + D.px;
#ifdef PARSY_STAMPS
#define STRACE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_trace[(256 + D.nupd) * 16 + (i)] = wall_clock64(); g_trace[(256 + D.nupd) * 16 + 8] = w; g_trace[(256 + D.nupd) * 16 + 9] = r; } } while (0)
#else
#define STRACE(i) do { } while (0)
#endif
    STRACE(0);
    const int total_pad = (total + 1) & ~1;
    double* __restrict__ S0 = P + total_pad;
    int32_t* __restrict__ R0 = reinterpret_cast<int32_t*>(S0 + 2 * stage_cap);

    // panel: all loads of a thread are issued before its LDS stores
    for (int base = 0; base < total; base += 8 * kThreads) {
        double tv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = base + q * kThreads + tid;
            tv[q] = e < total ? G[e] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = base + q * kThreads + tid;
            if (e < total) P[e] = tv[q];
        }
    }

    STRACE(1);
    // ---- stream state: block = (update u, k0, kc) with m * kc <= kSmallStage -----------------
    struct Blk {
        double v[kSmallPerThread];
        int32_t rel;    // relative index of row `tid` of the block (first block of an update only)
        int32_t m, kc, n1, first, u;  // m == 0: nothing; u: update number (index ring slot u & 3)
    };
    int lu = 0, lk = 0;  // loader position
    UpdDesc LU;
    int l_kstep = 1;
    auto loader_enter = [&]() {
        if (lu < D.nupd) {
            LU = upd[D.upd0 + lu];
            l_kstep = max(1, stage_cap / max(LU.m, 1));
        }
    };
    loader_enter();
    auto issue = [&](Blk& b) {
        b.m = 0;
        if (lu >= D.nupd) return;
        const int m = LU.m;
        // a descendant block taller than a stage is not staged: its update is applied straight
        // from global memory by the consumer (one pseudo-block covering all of K)
        const bool direct = m > stage_cap;
        const int kc = direct ? LU.K : min(l_kstep, LU.K - lk);
        b.m = direct ? -1 : m;
        b.kc = kc;
        b.n1 = LU.n1;
        b.first = lk == 0;
        b.u = lu;
        if (direct) {
            lk = 0;
            ++lu;
            loader_enter();
            return;
        }
        const double* __restrict__ src = L + LU.src + (int64_t)lk * LU.ld;
        const int cnt = m * kc;
#pragma unroll
        for (int q = 0; q < kSmallPerThread; ++q) {
            const int e = q * kThreads + tid;
            double v = 0.0;
            if (e < cnt) {
                const int kk = e / m, i = e - kk * m;
                v = src[i + (int64_t)kk * LU.ld];
            }
            b.v[q] = v;
        }
        b.rel = (lk == 0 && tid < m && tid < kSmallRelCap) ? relpos[LU.rel + tid] : 0;
        lk += kc;
        if (lk >= LU.K) {
            lk = 0;
            ++lu;
            loader_enter();
        }
    };
    auto store = [&](const Blk& b, int stage) {
        if (b.m <= 0) return;
        double* __restrict__ S = S0 + stage * stage_cap;
#pragma unroll
        for (int q = 0; q < kSmallPerThread; ++q) {
            const int e = q * kThreads + tid;
            if (e < b.m * b.kc) S[e] = b.v[q];
        }
        if (b.first && tid < b.m && tid < kSmallRelCap) R0[(b.u & 3) * kSmallRelCap + tid] = b.rel;
    };
    // consumer: needs m, kc, n1 of the block in `stage` and the index ring of its update
    int cu = 0, ck = 0;
    UpdDesc CU;
    int c_kstep = 1;
    auto consumer_enter = [&]() {
        if (cu < D.nupd) {
            CU = upd[D.upd0 + cu];
            c_kstep = max(1, stage_cap / max(CU.m, 1));
        }
    };
    consumer_enter();
    auto consume = [&](int stage) {
        const int m = CU.m, n1 = CU.n1;
        if (m > stage_cap) {  // not staged (see issue): straight from the descendant's panel
            const double* __restrict__ src = L + CU.src;
            const int32_t* __restrict__ relg = relpos + CU.rel;
            for (int j = tid >> 6; j < n1; j += kThreads / 64)
                for (int i = j + (tid & 63); i < m; i += 64) {
                    double acc = 0.0;
                    for (int kk = 0; kk < CU.K; ++kk)
                        acc = fma(src[i + (int64_t)kk * CU.ld], src[j + (int64_t)kk * CU.ld], acc);
                    P[relg[j] * r + relg[i]] -= acc;
                }
            ck = 0;
            ++cu;
            consumer_enter();
            return;
        }
        const int kc = min(c_kstep, CU.K - ck);
        const double* __restrict__ S = S0 + stage * stage_cap;
        const int32_t* __restrict__ rel = R0 + (cu & 3) * kSmallRelCap;  // came with the update's first block
        const int32_t* __restrict__ relg = relpos + CU.rel;  // rows beyond the staged indices (rare)
        // one wave per group of four columns j0..j0+3 of the pair block, lanes along the rows
        // i >= j0: the row operand is read once per k for four products, the column operands
        // are LDS broadcasts (no integer division anywhere)
        for (int j0 = 4 * (tid >> 6); j0 < n1; j0 += 4 * (kThreads / 64)) {
            int rj[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = min(j0 + c, n1 - 1);
                rj[c] = j < kSmallRelCap ? rel[j] : relg[j];
            }
            const int jc1 = min(j0 + 1, m - 1), jc2 = min(j0 + 2, m - 1), jc3 = min(j0 + 3, m - 1);
            for (int i = j0 + (tid & 63); i < m; i += 64) {
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                for (int kk = 0; kk < kc; ++kk) {
                    const double* __restrict__ Sk = S + kk * m;
                    const double a = Sk[i];
                    acc[0] = fma(a, Sk[j0], acc[0]);
                    acc[1] = fma(a, Sk[jc1], acc[1]);
                    acc[2] = fma(a, Sk[jc2], acc[2]);
                    acc[3] = fma(a, Sk[jc3], acc[3]);
                }
                const int ri = i < kSmallRelCap ? rel[i] : relg[i];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (j0 + c < n1 && i >= j0 + c) P[rj[c] * r + ri] -= acc[c];
            }
        }
        ck += kc;
        if (ck >= CU.K) {
            ck = 0;
            ++cu;
            consumer_enter();
        }
    };

    Blk b0, b1, b2;
    issue(b0);
    issue(b1);
    issue(b2);
    store(b0, 0);
    __syncthreads();
    int p = 0;
    while (cu < D.nupd) {
        store(b1, (p + 1) & 1);
        issue(b0);
        consume(p & 1);
        __syncthreads();
        ++p;
        if (cu >= D.nupd) break;
        store(b2, (p + 1) & 1);
        issue(b1);
        consume(p & 1);
        __syncthreads();
        ++p;
        if (cu >= D.nupd) break;
        store(b0, (p + 1) & 1);
        issue(b2);
        consume(p & 1);
        __syncthreads();
        ++p;
    }

    STRACE(2);
    // POTRF of the w x w diagonal block (register-blocked, four columns per step), then the rows
    // below it: X L' = B by forward substitution, one row per thread, L and its reciprocal
    // diagonal read from LDS as broadcasts (= POTRF followed by TRSM 'R','L','T','N',
    // reference :204,:218).
    double* __restrict__ scratch = reinterpret_cast<double*>(R0 + 4 * kSmallRelCap);  // kPotrfScratch + 64 doubles
    double* __restrict__ invd = scratch + kPotrfScratch;
    {
        const int ti = tid & 15, tj = tid >> 4;
        double a[4][4];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int i = 4 * ti + ri, c = 4 * tj + ci;
                double v = (i == c) ? 1.0 : 0.0;
                if (c < w && i < w && i >= c) v = P[c * r + i];
                a[ri][ci] = v;
            }
        int bad;
        potrf64_regs(a, scratch, ti, tj, w, bad, invd);
        if (bad) atomicMin(info, D.c0 + bad);  // only threads that saw a bad pivot
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int i = 4 * ti + ri, c = 4 * tj + ci;
                if (c < w && i < w && i >= c) P[c * r + i] = a[ri][ci];
            }
    }
    __syncthreads();
    STRACE(3);
    for (int i = w + tid; i < r; i += kThreads) {
        // four columns at a time: one read of x_k serves four products
        for (int j0 = 0; j0 < w; j0 += 4) {
            const int j1 = min(j0 + 1, w - 1), j2 = min(j0 + 2, w - 1), j3 = min(j0 + 3, w - 1);
            double x0 = P[j0 * r + i], x1 = P[j1 * r + i], x2 = P[j2 * r + i], x3 = P[j3 * r + i];
            for (int k = 0; k < j0; ++k) {
                const double* __restrict__ Pk = P + k * r;
                const double xk = Pk[i];
                x0 = fma(-xk, Pk[j0], x0);
                x1 = fma(-xk, Pk[j1], x1);
                x2 = fma(-xk, Pk[j2], x2);
                x3 = fma(-xk, Pk[j3], x3);
            }
            x0 *= invd[j0];
            x1 = fma(-x0, P[j0 * r + j1], x1) * invd[j1];
            x2 = fma(-x1, P[j1 * r + j2], fma(-x0, P[j0 * r + j2], x2)) * invd[j2];
            x3 = fma(-x2, P[j2 * r + j3], fma(-x1, P[j1 * r + j3], fma(-x0, P[j0 * r + j3], x3))) * invd[j3];
            P[j0 * r + i] = x0;
            if (j0 + 1 < w) P[j1 * r + i] = x1;
            if (j0 + 2 < w) P[j2 * r + i] = x2;
            if (j0 + 3 < w) P[j3 * r + i] = x3;
        }
    }
    __syncthreads();

    STRACE(4);
    for (int e = tid; e < total; e += kThreads) G[e] = P[e];
    STRACE(5);
    __syncthreads();  // subtree launch: the panel is stored (and LDS free) before the next supernode starts
  }
}

void launch_chol_small(const DevicePattern& P, int first, int count, int lds_bytes, int stage_cap, bool subtrees,
                       double* L, hipStream_t stream) {
    if (count <= 0) return;
    // panel (padded to 16 B) + two staged blocks + four index-ring slots + the POTRF's scratch
    stage_cap = min(max(stage_cap, 2), kSmallStage);
    const size_t lds = (size_t)((lds_bytes + 15) & ~15) + 2 * (size_t)stage_cap * sizeof(double) +
                       4 * kSmallRelCap * sizeof(int32_t) + (kPotrfScratch + kTile) * sizeof(double);
    // subtree launch: `first` counts (begin, end) pairs of small_ranges, which index the whole list
    hipLaunchKernelGGL(k_chol_small, dim3(count), dim3(kThreads), lds, stream, P.csn, P.upd, P.relpos,
                       subtrees ? P.small_list : P.small_list + first,
                       subtrees ? P.small_ranges + 2 * first : nullptr, L, P.info, stage_cap);
}

// ---------------------------------------------------------------------------
// TILES / CHAIN: one workgroup per 64x64 tile of a panel, one wave per 32x32
// sub-tile (see tile_task below).  The sub-tile lives in LDS (each wave owns its own, so the
// update loop needs no barrier and the summation order is fixed: update order, then k).
// ---------------------------------------------------------------------------
// Hand-off accesses of the chain launch (tiles published inside a launch): 8-byte agent-scope
// relaxed atomics = global_load/store_dwordx2 sc1 -- write-through stores, loads that bypass the
// CU's L1 -- on both sides, so no release/acquire fence is needed
// (MI355X_MICROARCH.md, inter-workgroup visibility, valid forms).
__device__ __forceinline__ double ld_sc1(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one LDS-DMA instruction of the chain launch: 16 bytes per lane from the lane's own global address to lds + 16 * lane,
// agent scope (sc1) like ld_sc1: the block columns it reads are published inside the launch
__device__ __forceinline__ void glds16_sc1(const double* g, double* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 16);
}

__device__ __forceinline__ void lds_sub(double* p, double v) {
    __hip_atomic_fetch_add(p, -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void lower_bound3(const int32_t* __restrict__ a, int n, int k0, int k1,
                                             int k2, int& r0, int& r1, int& r2) {
    int l0 = 0, l1 = 0, l2 = 0;
    if (n > 0) {
        for (int step = 1 << (31 - __clz(n)); step > 0; step >>= 1) {
            const int p0 = l0 + step, p1 = l1 + step, p2 = l2 + step;
            const int v0 = (p0 <= n) ? a[p0 - 1] : INT_MAX;
            const int v1 = (p1 <= n) ? a[p1 - 1] : INT_MAX;
            const int v2 = (p2 <= n) ? a[p2 - 1] : INT_MAX;
            if (v0 < k0) l0 = p0;
            if (v1 < k1) l1 = p1;
            if (v2 < k2) l2 = p2;
        }
    }
    r0 = l0;
    r1 = l1;
    r2 = l2;
}


// X := B inv(Ljj') on the 64 rows of the LDS tile `tile` (sub-tile layout of the tile kernel: 4 x
// 32x33); Dg holds Ljj (Dg[c * 65 + i], zeros above the diagonal), invd the reciprocals of its
// diagonal.  One workgroup barrier is expected before the call, one ends it.  The walker inlines
// it (values in flight across it); the other two places share one out-of-line copy, which keeps
// their register allocation apart from the stream's.
// inverses of the four 16x16 diagonal sub-blocks of Ljj, one column per thread, written
// transposed into the (unused) strict upper triangle of the same sub-block:
// Dg[(16b+r)*ld + 16b+c] = inv(L_bb)[r][c] for r > c.  A TRSM against Ljj is then all products
// (what a blocked dtrsm does): X_b = (B_b - sum_{p<b} X_p L_bp') inv(L_bb)'.  Ends with a workgroup barrier.
// (round 5) One WAVE per 16 x 16 sub-block, one ENTRY per lane, by halves -- inv [A 0; B C] = [inv A, 0; -inv(C) B inv(A),
// inv C]: the four 4 x 4 diagonal blocks by substitution (lane = (block, row, column)), then the two 4 x 4 and the one
// 8 x 8 off-diagonal blocks as two small products each (T = B inv(A), then -inv(C) T; T goes through 64 doubles of scratch).
// Five LDS round trips of a few reads each: about 0.5 us.  The form before (kept below, PARSY_INVERT16_COLUMNS) gave a
// column to a thread and walked its 16 rows one after the other, every second multiply-add waiting for its own LDS
// read: 3.7 us of the walker's 28.6-us step and ~2 us of every other tile's TRSM.  Scratch: a 16 x 16 block of Dg ABOVE the
// block diagonal (rows of an earlier sub-block, columns of a later one: zeros that nothing reads -- the TRSM takes
// L(b, p), p < b, from below the diagonal and the inverses from the upper triangles of the diagonal sub-blocks).
// Divisions: none (invd holds the reciprocals of the diagonal).
__device__ __forceinline__ void invert_diag_blocks(lds_f64* __restrict__ Dg, lds_f64* __restrict__ invd) {
#ifndef PARSY_INVERT16_COLUMNS
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = __builtin_amdgcn_readfirstlane(tid >> 6);   // sub-block = wave
    lds_f64* __restrict__ B0 = Dg + (16 * b) * kLdDiag + 16 * b;      // L[i][c] = B0[c * ld + i]; W[r][c] (r > c) -> B0[r * ld + c]
    const lds_f64* __restrict__ dv = invd + 16 * b;
    // scratch block (row block, column block): (0,1) (0,2) (0,3) (1,3)
    lds_f64* __restrict__ Sb = Dg + (16 * (b < 3 ? b + 1 : 3)) * kLdDiag + (b < 3 ? 0 : 16);
    auto SC = [&](int e) -> lds_f64& { return Sb[(e >> 4) * kLdDiag + (e & 15)]; };
    // (every LDS read below is unconditional -- positions that are not part of the inverse yet are read and replaced by a
    // select -- so that a phase's reads are in flight together: with the reads inside the selects each one was waited for)
    {   // 4 x 4 diagonal blocks: lane = (d, i', k'): column k' of inv(L_dd) by substitution, entry i'
        const int d4 = 4 * (lane >> 4), ip = (lane >> 2) & 3, kp = lane & 3;
        const lds_f64* __restrict__ T = B0 + d4 * kLdDiag + d4;
        const double d0 = dv[d4], d1 = dv[d4 + 1], d2 = dv[d4 + 2], d3 = dv[d4 + 3];
        const double t10 = T[1], t20 = T[2], t30 = T[3], t21 = T[kLdDiag + 2], t31 = T[kLdDiag + 3], t32 = T[2 * kLdDiag + 3];
        const double y0 = kp == 0 ? d0 : 0.0;
        const double y1 = kp == 1 ? d1 : -d1 * (t10 * y0);
        const double y2 = kp == 2 ? d2 : -d2 * fma(t21, y1, t20 * y0);
        const double y3 = kp == 3 ? d3 : -d3 * fma(t32, y2, fma(t31, y1, t30 * y0));
        const double y = ip == 1 ? y1 : ip == 2 ? y2 : y3;
        if (ip > kp) B0[(d4 + ip) * kLdDiag + d4 + kp] = y;
    }
    __builtin_amdgcn_wave_barrier();
    {   // the 4 x 4 blocks below the diagonal of the two 8 x 8 blocks: lanes 0..31 = (e, i, j)
        const int e8 = 8 * ((lane >> 4) & 1), i = (lane >> 2) & 3, j = lane & 3;
        double lr[4], wa[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            lr[m] = B0[(e8 + m) * kLdDiag + e8 + 4 + i];   // L[e8 + 4 + i][e8 + m]
            wa[m] = B0[(e8 + m) * kLdDiag + e8 + j];       // inv(A)[m][j] where m > j
        }
        const double dj = dv[e8 + j], di = dv[e8 + 4 + i];
        double t = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) t = fma(lr[m], m > j ? wa[m] : (m == j ? dj : 0.0), t);
        if (lane < 32) SC(lane) = t;
        __builtin_amdgcn_wave_barrier();
        double wc[4], tt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            wc[m] = B0[(e8 + 4 + i) * kLdDiag + e8 + 4 + m];   // inv(C)[i][m] where i > m
            tt[m] = SC((lane & 16) + 4 * m + j);
        }
        double w = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) w = fma(i > m ? wc[m] : (i == m ? di : 0.0), tt[m], w);
        if (lane < 32) B0[(e8 + 4 + i) * kLdDiag + e8 + j] = -w;
    }
    __builtin_amdgcn_wave_barrier();
    {   // the 8 x 8 block below the diagonal: lane = (i, j)
        const int i = lane >> 3, j = lane & 7;
        double lr[8], wa[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            lr[m] = B0[m * kLdDiag + 8 + i];   // L[8 + i][m]
            wa[m] = B0[m * kLdDiag + j];       // inv(A)[m][j] where m > j
        }
        const double dj = dv[j], di = dv[8 + i];
        double t = 0.0;
#pragma unroll
        for (int m = 0; m < 8; ++m) t = fma(lr[m], m > j ? wa[m] : (m == j ? dj : 0.0), t);
        SC(lane) = t;   // (the scratch reads of the step before were issued earlier: LDS operations of a wave execute in order)
        __builtin_amdgcn_wave_barrier();
        double wc[8], tt[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            wc[m] = B0[(8 + i) * kLdDiag + 8 + m];   // inv(C)[i][m] where i > m
            tt[m] = SC(8 * m + j);
        }
        double w = 0.0;
#pragma unroll
        for (int m = 0; m < 8; ++m) w = fma(i > m ? wc[m] : (i == m ? di : 0.0), tt[m], w);
        B0[(8 + i) * kLdDiag + j] = -w;
    }
#else
    const int tid = threadIdx.x;
    if (tid < kTile) {
        const int b16 = (tid >> 4) * 16, c = tid & 15;
        double y[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = (k == c) ? invd[b16 + k] : 0.0;
#pragma unroll
        for (int rr = 1; rr < 16; ++rr) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < rr; ++k) sacc = fma(Dg[(b16 + k) * kLdDiag + b16 + rr], y[k], sacc);
            y[rr] = (rr > c) ? -sacc * invd[b16 + rr] : y[rr];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all reads of the sub-block precede the in-place writes
#pragma unroll
        for (int rr = 1; rr < 16; ++rr)
            if (rr > c) Dg[(b16 + rr) * kLdDiag + b16 + c] = y[rr];
    }
#endif
    // a barrier for LDS alone: vector-memory loads the caller has in flight (the walker's next diagonal tile) stay
    // in flight (__syncthreads() would wait for them: 1.5 us of the walker's step)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <bool INVERT = true>
__device__ __forceinline__ void invert_and_trsm_inline(lds_f64* __restrict__ tile, lds_f64* __restrict__ Dg,
                                                       lds_f64* __restrict__ invd, int nb) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    auto cell = [&](int i, int c) -> lds_f64& {
        return tile[((i >> 5) * 2 + (c >> 5)) * (kSub * kLdSub) + (c & 31) * kLdSub + (i & 31)];
    };
    if (INVERT) invert_diag_blocks(Dg, invd);   // (false: the caller has done it)
    // each wave owns 16 rows of the tile for the whole solve: no barrier between blocks
    const int rbase = 16 * wave;
    for (int b16 = 0; b16 < nb; b16 += 16) {
        double4_t acc = {0, 0, 0, 0};
        for (int p16 = 0; p16 < b16; p16 += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = p16 + kq + 4 * u;
                const double av = cell(rbase + l15, k);                // X[row][k]
                const double bv = Dg[k * kLdDiag + b16 + l15];         // L[b16 + j][k]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) cell(rbase + kq + 4 * v, b16 + l15) -= acc[v];
        double4_t acc2 = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kq + 4 * u, j = l15;
            const double av = cell(rbase + l15, b16 + k);              // R[row][k]
            double wv = 0.0;                                            // inv(L_bb)'[k][j] = inv(L_bb)[j][k]
            if (j > k) wv = Dg[(b16 + j) * kLdDiag + b16 + k];
            else if (j == k) wv = invd[b16 + k];
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, wv, acc2, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) cell(rbase + kq + 4 * v, b16 + l15) = acc2[v];
    }
    __syncthreads();
}
__device__ __noinline__ void invert_and_trsm(lds_f64* __restrict__ tile, lds_f64* __restrict__ Dg,
                                             lds_f64* __restrict__ invd, int nb) {
    invert_and_trsm_inline(tile, Dg, invd, nb);
}

static constexpr int kKC = 16;        // k extent of one chunk of an update stream
static constexpr int kInFlightChain = 4;  // chunks in flight per wave (operands prefetched into registers)
static constexpr int kInFlightTiles = 3;  // ... in the TILES launch: fewer registers, three workgroups per CU
static constexpr unsigned long long kSpinTicks = 200000000ull;  // 2 s of the 100 MHz wall clock

// One workgroup per 64x64 tile of a panel, one wave per 32x32 sub-tile, and the four waves run
// their update streams independently (no workgroup barrier until the stream is finished).  A
// wave's stream is cut in 16-wide k chunks.  The MFMA operands of a chunk go straight from the
// source panel (column-major: 16 consecutive rows per k are one 128-B segment) into registers
// (v_mfma_f64_16x16x4_f64: lane = (row & 15, k >> 2 group)), kInFlight chunks ahead of the
// multiply; at the end of an entry the product is scatter-subtracted into the wave's private
// LDS sub-tile through the relative indices (ds_add_f64 without return; the LDS operations of
// one wave execute in order, so the summation order is fixed).
//
// CHAIN = false (TILES launch): the stream is the early WaveEntry list of the tile (built on the
//   host: every descendant with rows in the sub-tile's row AND column window, in update order);
//   the tile is written back.
// CHAIN = true: the workgroup takes its tile (I,J) from the launch's ticket counter.  Stream:
//   the late WaveEntry list, then block columns k = 0..J-1 of the tile's own supernode
//   (left-looking inside the supernode; identity row map), each as soon as tiles (I,k) and (J,k)
//   are published.  Then the tile is finished -- POTRF (I == J), or TRSM against the published
//   diagonal tile -- written to the panel write-through and published (flag = epoch; every read of
//   a published tile is an sc1 load: cdna_hip_programming.md Guideline 16).  Tickets are handed out in
//   start order and tiles are listed producers-first, so a workgroup only ever waits for
//   workgroups that have started.  Every wait is bounded; a timeout (or any failure flag) makes
//   all waiters give up and parsy_factor_status() report < 0.
struct TileLds {  // LDS of one workgroup of the tile kernel
    double T[4][kSub * kLdSub];       // the tile, one 32x33 sub-tile per wave
    alignas(16) double colbuf[kPotrfScratch];  // POTRF scratch
    double dgbuf[4 * kSub * kLdSub];  // diagonal block + its 16x16 inverses (TRSM) / the walker's next diagonal tile
    double s_invd[kTile];             // reciprocals of the diagonal of the block being solved against
    int32_t s_ok, s_task, s_cnt, s_cnt2;
};
// one staged operand block of the chain's ring (see tile_task, "finishes in REGISTERS"): 8 DMA instructions of 2 k
// columns x 64 rows, every second one 16 doubles further: k and k + 1 of an operand read in opposite bank halves
static constexpr int kRingOp = 8 * 128 + 4 * 16;
static constexpr int kRingSlot = 2 * kRingOp;   // a chunk: the tile's rows and its columns, 16 k each
// LDS of a workgroup of the chain's second launch of a level (the rows below the diagonal squares: no walker, no
// prepared tile): the tile while the descendants' stream runs, then the ring of THREE chunks, then the diagonal
// block -- one after the other in the same 52 KB, so that three workgroups share a compute unit
struct RowsLds {
    double T[4][kSub * kLdSub];
    double ring_rest[3 * kRingSlot - 4 * kSub * kLdSub];
    double s_invd[kTile];
    int32_t s_ok, s_task, s_cnt, s_cnt2;
};
static_assert(offsetof(TileLds, dgbuf) == sizeof(double) * (4 * kSub * kLdSub + kPotrfScratch) &&
                  offsetof(TileLds, s_invd) == sizeof(double) * (8 * kSub * kLdSub + kPotrfScratch) &&
                  4 * kRingSlot <= 8 * kSub * kLdSub + kPotrfScratch,
              "TileLds: T, colbuf, dgbuf are one contiguous area that holds a ring of four chunks");
static_assert(offsetof(RowsLds, s_invd) == sizeof(double) * 3 * kRingSlot && kTile * kLdDiag <= 4 * kSub * kLdSub &&
                  sizeof(RowsLds) <= 160 * 1024 / 3, "RowsLds: three chunks, three workgroups per compute unit");
__device__ __forceinline__ double* lds_colbuf(TileLds& S) { return S.colbuf; }
__device__ __forceinline__ double* lds_colbuf(RowsLds&) { return nullptr; }
__device__ __forceinline__ double* lds_dgbuf(TileLds& S) { return S.dgbuf; }
__device__ __forceinline__ double* lds_dgbuf(RowsLds& S) { return &S.T[0][0]; }   // (the tile is in registers by then)

template <bool CHAIN, bool ROWS = false, class LdsT = TileLds>
__device__ __forceinline__ void tile_task(LdsT& S, const int task, const SnDesc* __restrict__ sn,
                                          const int32_t* __restrict__ relpos,
                                          const WaveEntry* __restrict__ wents,
                                          const int64_t* __restrict__ wptr,
                                          const int64_t* __restrict__ split_ranges,
                                          double* __restrict__ tile_scratch,
                                          const TileDesc* __restrict__ tiles, double* __restrict__ L,
                                          int* __restrict__ info, int* __restrict__ tflags,
                                          const int nflags_arg, const int epoch) {
    constexpr int kInFlight = (CHAIN && !ROWS) ? kInFlightChain : kInFlightTiles;
    double (&T)[4][kSub * kLdSub] = S.T;
    double* const colbuf = lds_colbuf(S);
    double* const dgbuf = lds_dgbuf(S);
    double (&s_invd)[kTile] = S.s_invd;
    int32_t &s_ok = S.s_ok, &s_cnt = S.s_cnt, &s_cnt2 = S.s_cnt2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileDesc td = tiles[task];
    const SnDesc D = sn[td.sn];
    const int r = D.r, w = D.w, ld = D.ld;  // ld > r: a piece of a split supernode (window of its panel)
    double* __restrict__ G = L + D.px;
    const int tI = td.row0 / kTile, tJ = td.col0 / kTile, nbc = (w + kTile - 1) / kTile;
    const bool diag_tile = td.row0 == td.col0;
    // roles of the chain launch (see the walker below)
    // (ROWS: the launch holds none of them)
    const bool walker = CHAIN && !ROWS && tI == 0 && tJ == 0;
    const bool prep_c = CHAIN && !ROWS && diag_tile && tJ > 0;           // diagonal tile (J,J), J >= 1
    const bool prep_b = CHAIN && !ROWS && tI == tJ + 1 && tI < nbc;      // tile (J+1,J) left of a diagonal tile
    // the chain's waves go first where they share a SIMD with the side stream's TILES waves (priority 0)
    if (CHAIN) {
        if (walker || prep_b || prep_c) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(1);
    }

#ifdef PARSY_STAMPS
    unsigned long long tp_last = wall_clock64();
    const bool tp_on = CHAIN && !walker && !prep_b && !prep_c;
#endif
    const int wa = wave >> 1, wb = wave & 1;
    const int subrow0 = td.row0 + kSub * wa, subcol0 = td.col0 + kSub * wb;
    const bool wave_on = subrow0 < r && subcol0 < w && subrow0 >= subcol0;
    const int nrows = min(kSub, r - subrow0), ncols = min(kSub, w - subcol0);
    double* __restrict__ Tw = T[wave];
    const int l15 = lane & 15, kq = lane >> 4, l31 = lane & 31;
    const bool diag_sub = subrow0 == subcol0;

    // the sub-tile's current values: loads first, LDS stores after the stream's first loads
    // A tile whose early stream is split over several TILES workgroups: part 0 works on the tile in
    // place, parts 1.. on partial tiles (start from 0, dense 64x64 in the tile scratch) that the chain
    // launch adds, in part order, when it loads the tile.
    const int split_part = CHAIN ? 0 : (td.part & 255);
    const int split_n = CHAIN ? td.part : (td.part >> 8);
    // A tile of the chain launch with no update list at all (most tiles of the pieces of a top separator: their
    // descendants went through the BIG and TILES launches) never enters LDS: the panel -> accumulators, below.
    const bool direct = CHAIN && !walker && split_n <= 1 && wptr[td.wp] == wptr[td.wp + 4];
    double tv[kSub * kSub / 64];
    if (wave_on && !direct) {
#pragma unroll
        for (int q = 0; q < kSub * kSub / 64; ++q) {
            const int e = q * 64 + lane;
            const int cc = e >> 5, rr = e & 31;
            const bool in = rr < nrows && cc < ncols && (subrow0 + rr >= subcol0 + cc);
            tv[q] = (in && split_part == 0) ? G[(int64_t)(subcol0 + cc) * ld + subrow0 + rr] : 0.0;
        }
        if (CHAIN && split_n > 1) {
            for (int part = 1; part < split_n; ++part) {
                const double* __restrict__ PT = tile_scratch + td.sp + (int64_t)(part - 1) * (kTile * kTile);
#pragma unroll
                for (int q = 0; q < kSub * kSub / 64; ++q) {
                    const int e = q * 64 + lane;
                    const int cc = e >> 5, rr = e & 31;
                    const bool in = rr < nrows && cc < ncols && (subrow0 + rr >= subcol0 + cc);
                    if (in) tv[q] += PT[(kSub * wb + cc) * kTile + kSub * wa + rr];
                }
            }
        }
    }
    auto store_subtile = [&]() {
#pragma unroll
        for (int q = 0; q < kSub * kSub / 64; ++q) {
            const int e = q * 64 + lane;
            Tw[(e >> 5) * kLdSub + (e & 31)] = tv[q];
        }
    };

    // ---- this wave's update stream
    int64_t le = 0, e_end = 0;       // next entry / end of the list
    if (wave_on && !direct) {
        if (!CHAIN && split_n > 1) {
            le = split_ranges[td.wp + 2 * wave];
            e_end = split_ranges[td.wp + 2 * wave + 1];
        } else {
            le = wptr[td.wp + wave];
            e_end = wptr[td.wp + wave + 1];
        }
    }
    if (le < e_end) {
        struct Chunk {
            double a0[4], a1[4], b0[4], b1[4];  // MFMA operands of the four k steps
            int32_t rel;                         // this lane's relative index (used with the last chunk)
            int32_t kend, last, mn;              // wave-uniform: valid k in the chunk (0: padding of the
                                                 // stream), last chunk of its entry, window sizes
        };
        // loader state (wave-uniform except the lane offsets).  Every chunk issues the same 16
        // operand loads (+ the relative index): fragments a narrow entry does not have re-read
        // its last row, k steps past a ragged end re-read column K-1 (masked in the multiply),
        // chunks behind the end of the stream re-read the last one.  A fixed number of loads
        // per chunk keeps the s_waitcnt of the multiply exactly kInFlight-1 chunks behind.
        const double* l_p = G;                   // source panel at row lb, column = chunk start
        int l_K = 0, l_k = 0, l_ld = 0, l_mn = 0;
        unsigned l_oA0 = 0, l_oA1 = 0, l_oB0 = 0, l_oB1 = 0, l_orel = 0;  // row offsets of the lane's fragment rows
        WaveEntry l_next = {};
        if (le < e_end) l_next = wents[le];
        bool l_live = true;
        auto loader_enter = [&]() {
            WaveEntry E;
            E = l_next;  // fetched one entry ahead
            if (le + 1 < e_end) l_next = wents[le + 1];
            ++le;
            const int mi = E.mn & 255, nj = (E.mn >> 8) & 255;
            l_p = L + E.src;
            l_K = E.K;
            l_k = 0;
            l_ld = E.ld;
            l_mn = E.mn;
            l_oA0 = E.ia + min(l15, mi - 1);
            l_oA1 = E.ia + min(16 + l15, mi - 1);
            l_oB0 = E.ja + min(l15, nj - 1);
            l_oB1 = E.ja + min(16 + l15, nj - 1);
            l_orel = E.rel + (lane < 32 ? E.ia + min(l31, mi - 1) : E.ja + min(l31, nj - 1));
        };
        loader_enter();
        auto issue = [&](Chunk& c) {
            const int kend = l_live ? min(kKC, l_K - l_k) : 0;
            c.kend = kend;
            c.mn = l_mn;
            c.last = l_live && (l_k + kKC >= l_K);
#ifdef PARSY_ABL_NOLOAD
#pragma unroll
            for (int u = 0; u < 4; ++u) c.a0[u] = c.a1[u] = c.b0[u] = c.b1[u] = (double)(kend + u);
#else
            // One block of loads for every kind of chunk.  In the chain launch they are all sc1 loads
            // (block columns of the tile's own supernode are published during the launch; descendants'
            // panels read the same way cost nothing extra: sc1 only bypasses the CU's L1).
            unsigned ko[4];
            if (kend == kKC) {
#pragma unroll
                for (int u = 0; u < 4; ++u) ko[u] = (4 * u + kq) * l_ld;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) ko[u] = min(4 * u + kq, max(kend, 1) - 1) * l_ld;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (CHAIN) {
                    c.a0[u] = ld_sc1(l_p + l_oA0 + ko[u]);
                    c.b0[u] = ld_sc1(l_p + l_oB0 + ko[u]);
                    c.a1[u] = ld_sc1(l_p + l_oA1 + ko[u]);
                    c.b1[u] = ld_sc1(l_p + l_oB1 + ko[u]);
                } else {
                    c.a0[u] = l_p[l_oA0 + ko[u]];
                    c.b0[u] = l_p[l_oB0 + ko[u]];
                    c.a1[u] = l_p[l_oA1 + ko[u]];
                    c.b1[u] = l_p[l_oB1 + ko[u]];
                }
            }
#endif
            c.rel = relpos[l_orel];
            if (c.last) {
                if (le < e_end) loader_enter();
                else l_live = false;
            } else if (l_live) {
                l_k += kKC;
                l_p += (int64_t)kKC * l_ld;
            }
        };

        double4_t c00 = {0, 0, 0, 0}, c01 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};
        auto consume = [&](const Chunk& c) {
            const int mi = c.mn & 255, nj = (c.mn >> 8) & 255;
            const bool two_r = mi > 16, two_c = nj > 16;
            const bool up = two_c && !diag_sub;  // rows 0..15 x columns 16..31: strictly upper in a diagonal sub-tile
            if (c.kend > 0) {
                // k past a ragged end contributes 0 through the A operand (B holds finite panel values)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool kin = 4 * u + kq < c.kend;
                    const double a0 = kin ? c.a0[u] : 0.0, a1 = kin ? c.a1[u] : 0.0;
#ifdef PARSY_ABL_NOMFMA
                    c00[0] += a0 * c.b0[u] + a1 * c.b1[u];
                    continue;
#endif
                    c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, c.b0[u], c00, 0, 0, 0);
                    if (up) c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, c.b1[u], c01, 0, 0, 0);
                    if (two_r) c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, c.b0[u], c10, 0, 0, 0);
                    if (two_r && two_c) c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, c.b1[u], c11, 0, 0, 0);
                }
            }
#ifdef PARSY_ABL_NOSCATTER
            if (false) {
#else
            if (c.last) {
#endif
                // scatter-subtract through the relative indices (C/D layout of
                // v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg).  Lanes 0..31
                // hold the sub-tile row of source row `lane` of the row window, lanes 32..63 the
                // sub-tile column of source row `lane - 32` of the column window (-1: outside).
                const bool ident = ((c.mn >> 16) & 1) != 0;
                const int relv = (l31 < (lane < 32 ? mi : nj))
                                     ? (ident ? l31 : c.rel - D.rbias - (lane < 32 ? subrow0 : subcol0)) : -1;
                const int C0 = __builtin_amdgcn_ds_bpermute((32 + l15) * 4, relv);
                const int C1 = __builtin_amdgcn_ds_bpermute((48 + l15) * 4, relv);
                // Cells outside the update (padding of the 16x16 fragments, the strict upper triangle
                // of a diagonal sub-tile) subtract 0.0 from a padding element of the sub-tile (row 32
                // of column lane & 31): branch-free, and the dummies do not pile onto one address.
                const int dummy = l31 * kLdSub + kSub;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int R0 = __builtin_amdgcn_ds_bpermute((kq + 4 * v) * 4, relv);
                    const bool ok00 = R0 >= 0 && C0 >= 0 && (!diag_sub || R0 >= C0);
                    lds_sub(&Tw[ok00 ? C0 * kLdSub + R0 : dummy], ok00 ? c00[v] : 0.0);
                    if (up) {
                        const bool ok01 = R0 >= 0 && C1 >= 0;
                        lds_sub(&Tw[ok01 ? C1 * kLdSub + R0 : dummy], ok01 ? c01[v] : 0.0);
                    }
                    if (two_r) {
                        const int R1 = __builtin_amdgcn_ds_bpermute((16 + kq + 4 * v) * 4, relv);
                        const bool ok10 = R1 >= 0 && C0 >= 0 && (!diag_sub || R1 >= C0);
                        lds_sub(&Tw[ok10 ? C0 * kLdSub + R1 : dummy], ok10 ? c10[v] : 0.0);
                        if (two_c) {
                            const bool ok11 = R1 >= 0 && C1 >= 0 && (!diag_sub || R1 >= C1);
                            lds_sub(&Tw[ok11 ? C1 * kLdSub + R1 : dummy], ok11 ? c11[v] : 0.0);
                        }
                    }
                }
                c00 = {0, 0, 0, 0};
                c01 = {0, 0, 0, 0};
                c10 = {0, 0, 0, 0};
                c11 = {0, 0, 0, 0};
            }
        };

        // rounds of kInFlight chunks, one back edge; the stream is done when a round issued
        // nothing but padding (kend == 0: multiplies nothing, scatters nothing)
        Chunk q[kInFlight];
#pragma unroll
        for (int i = 0; i < kInFlight; ++i) issue(q[i]);
        store_subtile();
        bool more = true;
        while (more) {
            more = false;
#pragma unroll
            for (int sidx = 0; sidx < kInFlight; ++sidx) {
                consume(q[sidx]);
                issue(q[sidx]);
                more = more || q[sidx].kend != 0;
            }
        }
    } else if (wave_on && !direct) {
        store_subtile();
    }
    if (!direct) __syncthreads();
#ifdef PARSY_STAMPS
    if (tp_on) TPHASE(0);
#endif

    // ---------------------------------------------------------------------------------------
    // After the stream.  Tiles live in LDS in the sub-tile layout of T (4 x 32x33); the tile
    // buffer and the buffer of the diagonal block (64x65, with its 16x16 inverses) are T and
    // dgbuf in either order.
    // ---------------------------------------------------------------------------------------
    double* const Tflat = &T[0][0];
    auto cell = [&](double* buf, int i, int c) -> double& {
        return buf[((i >> 5) * 2 + (c >> 5)) * (kSub * kLdSub) + (c & 31) * kLdSub + (i & 31)];
    };
    // this wave's quadrant of the tile at (row0, col0): rows >= min_row of the tile go to the panel
    auto write_tile = [&](double* buf, int row0, int col0, int min_row) {
        const int sr = row0 + kSub * wa, sc = col0 + kSub * wb;
        const int nr = min(kSub, r - sr), nc = min(kSub, w - sc);
        if (sr < sc || nr <= 0 || nc <= 0) return;
        const double* __restrict__ Q = buf + wave * (kSub * kLdSub);
        for (int e = lane; e < kSub * kSub; e += 64) {
            const int cc = e >> 5, rr = e & 31;
            if (rr < nr && cc < nc && (sr + rr >= sc + cc) && kSub * wa + rr >= min_row) {
                double* dst = &G[(int64_t)(sc + cc) * ld + sr + rr];
                if (CHAIN) st_sc1(dst, Q[cc * kLdSub + rr]);
                else *dst = Q[cc * kLdSub + rr];
            }
        }
    };
    if (!CHAIN) {
        if (split_part == 0) {
            write_tile(Tflat, td.row0, td.col0, 0);
        } else if (wave_on) {
            double* __restrict__ PT = tile_scratch + td.sp + (int64_t)(split_part - 1) * (kTile * kTile);
            for (int e = lane; e < kSub * kSub; e += 64) {
                const int cc = e >> 5, rr = e & 31;
                PT[(kSub * wb + cc) * kTile + kSub * wa + rr] = Tw[cc * kLdSub + rr];
            }
        }
        return;
    }
    const int nflags = nflags_arg;
    auto publish = [&](int flag_index) {
        // the tile was stored write-through (sc1): every storing wave drains its stores, the
        // workgroup meets, one lane raises the flag (Guideline 16, R1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&tflags[flag_index], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto wait_flags = [&](int f0, int f1) -> bool {  // workgroup-wide bounded wait for two flags
        if (tid == 0) {
            const unsigned long long t0 = wall_clock64();
            int ok = 1, spins = 0;
            while (__hip_atomic_load(&tflags[f0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch ||
                   __hip_atomic_load(&tflags[f1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                if ((++spins & 15) == 0 &&
                    (wall_clock64() - t0 > kSpinTicks ||
                     __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok && tid == 0) atomicMin(info, -1);  // status < 0: a wait timed out / was abandoned
        return s_ok != 0;
    };
    double* __restrict__ invd = s_invd;
    const int my_flag = D.tflag0 + tI * nbc + tJ;

    if (!walker) {
        // -----------------------------------------------------------------------------------
        // Every tile of the chain launch but the walker's first one finishes in REGISTERS.  Wave w takes
        // rows 16 w .. 16 w + 15 of the tile, all 64 columns: four accumulators of v_mfma_f64_16x16x4_f64 in the
        // "transposed" form (lane & 15 = row of the tile, (lane >> 4) + 4 reg = column inside a 16-column block),
        // in which register u of a block IS the operand of k step u of a later product -- the TRSM below never
        // goes back to LDS with the tile.
        //   1. the updates by the earlier block columns of the tile's OWN supernode (the reference's DSYRK/DGEMM
        //      inside a supernode; parallel_PB_Cholesky_05.h:160,173 per 64-column block): tile - L(I,0..) L(J,0..)',
        //      identity row map, products negated by the instruction and accumulated across ALL block columns
        //      (tile - p0 - p1 - ... in k order).  The two operand blocks (64 rows x 16 k each) are staged once per
        //      workgroup by LDS-DMA into a ring of four chunks that takes over the LDS (the tile is in registers):
        //      8 flop per fetched byte in 16-byte lanes instead of 4 in 8-byte lanes -- the per-wave streams of
        //      round 2 ran these 7e11 flops of the Flan-class input at the L2's request rate (11 TFLOP/s).  Rows
        //      past the panel's end re-read its last row (one element beyond it: a later column of the same panel
        //      follows) and land in cells that are never stored.
        //   2. prepared tiles (walker's inputs) are stored and announced; the others wait for the diagonal tile
        //      of their block column, invert its four 16 x 16 diagonal blocks and run the blocked TRSM (what a
        //      blocked dtrsm does: X_b = (B_b - sum_{p<b} X_p L_bp') inv(L_bb)', reference :218) on the
        //      accumulators, then store and publish.
        // -----------------------------------------------------------------------------------
        constexpr int kOp = kRingOp, kSlot = kRingSlot, kSlots = ROWS ? 3 : 4;
        const int n_int = prep_c ? tJ - 1 : tJ;  // block column J-1 reaches a diagonal tile through the walker
        const int nb = min(kTile, w - td.col0);
        const int ng = diag_tile ? min(wave + 1, (nb + 15) >> 4) : (nb + 15) >> 4;   // 16-column blocks this wave holds
        const bool rows_on = td.row0 + 16 * wave < r;
        double4_t acc[4];
        if (direct) {
            const int row = td.row0 + 16 * wave + l15;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int col = td.col0 + 16 * g + kq + 4 * v;
                    acc[g][v] = (row < r && col < w && row >= col) ? G[(int64_t)col * ld + row] : 0.0;
                }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[g][v] = cell(Tflat, 16 * wave + l15, 16 * g + kq + 4 * v);
        }
        // ROWS: the diagonal tile of the block column was published by the level's first launch -- no flag to wait
        // for, and its values are fetched now, behind everything that follows
        double dtmp[kTile * kTile / kThreads];
        double dinv_pre = 1.0;
        auto fetch_diag = [&]() {
            const double* DB = G + (int64_t)td.col0 * ld + td.col0;  // factored diagonal block
#pragma unroll
            for (int q = 0; q < kTile * kTile / kThreads; ++q) {
                const int e = q * kThreads + tid;
                const int c = e >> 6, i = e & 63;
                dtmp[q] = (c < nb && i < nb && i >= c) ? ld_sc1(&DB[(int64_t)c * ld + i]) : 0.0;
            }
            if (tid < kTile) dinv_pre = (tid < nb) ? ld_sc1(&DB[(int64_t)tid * ld + tid]) : 1.0;
        };
        if (ROWS) fetch_diag();
        bool gave_up = false;
        if (n_int > 0) {
            double* const stg = Tflat;
            __syncthreads();  // the tile is in registers: the LDS is the ring now
            // kready = number of leading block columns whose tiles (I,k) and (J,k) are known to be published
            int kready = 0;
            const int fI = D.tflag0 + tI * nbc, fJ = D.tflag0 + tJ * nbc;  // flags of tiles (I,0..), (J,0..)
            auto extend_ready = [&]() {  // take every further published column
                while (kready < n_int) {
                    const int k = kready + lane;
                    bool ok = false;
                    if (k < n_int) {
                        ok = __hip_atomic_load(&tflags[fI + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
                        if (!diag_tile)
                            ok = ok && __hip_atomic_load(&tflags[fJ + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
                    }
                    const unsigned long long miss = ~__ballot(ok);
                    const int adv = miss ? __builtin_ctzll(miss) : 64;
                    kready += adv;
                    if (adv < 64) break;
                }
                // every read of a published tile is an sc1 access: no cache to invalidate; this only keeps
                // the compiler from moving those reads above the poll
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            };
            auto ensure_ready = [&](int k) {  // block (bounded) until block column k can be read
                const unsigned long long t0 = wall_clock64();
                int spins = 0;
                while (kready <= k) {
                    extend_ready();
                    if (kready > k) break;
                    if ((spins & 15) == 15 &&
                        (wall_clock64() - t0 > kSpinTicks ||
                         __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
                        gave_up = true;     // status < 0 below; go on with whatever is there so that no barrier is missed
                        kready = n_int;
                        break;
                    }
                    // the last block column of a tile the walker is waiting for is on the critical path:
                    // poll it tightly (few such tiles at any time); everything else polls lazily
                    if (spins < 8 || ((prep_b || prep_c) && k == n_int - 1)) __builtin_amdgcn_s_sleep(4);
                    else __builtin_amdgcn_s_sleep(48);
                    ++spins;
                }
            };
            // DMA: wave w moves k columns 4w .. 4w+3 of both operand blocks of a chunk; instruction 2w takes columns
            // 4w (lanes 0..31, rows 2 lane, 2 lane + 1) and 4w + 2 (lanes 32..63), instruction 2w + 1 columns 4w + 1
            // and 4w + 3, so that the k and k + 1 of one operand read (lanes 0..15 / 16..31) come from different
            // instructions
            const int dj = lane & 31, dk = 4 * wave + 2 * (lane >> 5);
            const double* gA = G + (int64_t)dk * ld + min(td.row0 + 2 * dj, r - 1);
            const double* gB = G + (int64_t)dk * ld + min(td.col0 + 2 * dj, r - 1);
            double* const ldsA = stg + 272 * wave;
            const int nch = 4 * n_int;
            int issued = 0, islot = 0, cslot = 0;   // chunks started / the slot of the next one / of the chunk being multiplied
            auto dma = [&]() {  // chunk `issued` -> slot issued % kSlots
                double* dst = ldsA + islot * kSlot;
                islot = islot + 1 == kSlots ? 0 : islot + 1;
                glds16_sc1(gA, dst);
                glds16_sc1(gA + ld, dst + 144);
                if (!diag_tile) {
                    glds16_sc1(gB, dst + kOp);
                    glds16_sc1(gB + ld, dst + kOp + 144);
                }
                gA += (int64_t)kKC * ld;
                gB += (int64_t)kKC * ld;
                ++issued;
            };
            const int oR = (kq & 1) * 144 + (kq >> 1) * 64 + 16 * wave + l15;               // the tile's rows: L(I,k)
            const int oC = (kq & 1) * 144 + (kq >> 1) * 64 + l15 + (diag_tile ? 0 : kOp);   // its columns: L(J,k)
            for (int c = 0; c < nch; ++c) {
                if (issued <= c) {
                    ensure_ready(c >> 2);
                    dma();
                }
                // this wave's part of chunk c has landed when only the DMA of the later chunks is outstanding
                const int later = issued - c - 1;
                if (diag_tile) {
                    if (kSlots > 3 && later >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else if (later == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (later == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    if (kSlots > 3 && later >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                // everybody's part of chunk c is there, and everybody is done with chunk c - 1: its slot takes chunk
                // c + kSlots - 1
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const int lim = min(c + kSlots, nch);
                while (issued < lim) {
                    if ((issued >> 2) >= kready) {
                        extend_ready();
                        if ((issued >> 2) >= kready) break;
                    }
                    dma();
                }
                if (rows_on) {
                    const double* __restrict__ Ss = stg + cslot * kSlot;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double rv = Ss[272 * u + oR];
                        double cv[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) cv[g] = Ss[272 * u + oC + 16 * g];
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            if (g < ng) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv[g], rv, acc[g], 0, 0, 1);
                    }
                }
                cslot = cslot + 1 == kSlots ? 0 : cslot + 1;
            }
        }
        // the wave's rows go to the panel straight from the accumulators (lanes along the rows: 128-byte segments)
        auto store_regs = [&]() {
            const int row = td.row0 + 16 * wave + l15;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int col = td.col0 + 16 * g + kq + 4 * v;
                    if (row < r && col < w && row >= col) st_sc1(&G[(int64_t)col * ld + row], acc[g][v]);
                }
        };
#ifdef PARSY_STAMPS
        if (tp_on) {
            TPHASE(1);
            if (tid == 0) {
                atomicAdd(&g_tilephase[8], 1ull);
                atomicAdd(&g_tilephase[9], (unsigned long long)n_int);
            }
        }
#endif
        if (gave_up) atomicMin(info, -1);
        if (prep_b || prep_c) {
            // prepared for the walker: every update except the walker's own is in; it goes to the panel
            // and is announced with the tile's PREP flag
            store_regs();
            publish(nflags + my_flag);
            return;
        }
        // ---- a tile below the diagonal: wait for the diagonal tile of its block column, TRSM, publish
        const int fd = D.tflag0 + tJ * nbc + tJ;
        if (ROWS) {
            __syncthreads();   // the ring is read
        } else {
            if (!wait_flags(fd, fd)) {   // (its barrier: the ring is read)
                store_regs();
                publish(my_flag);  // (so that nobody else waits for this tile)
                return;
            }
            TPHASE(2);
            fetch_diag();
        }
        {
#pragma unroll
            for (int q = 0; q < kTile * kTile / kThreads; ++q) {
                const int e = q * kThreads + tid;
                dgbuf[(e >> 6) * kLdDiag + (e & 63)] = dtmp[q];
            }
            if (tid < kTile) invd[tid] = 1.0 / dinv_pre;
        }
        __syncthreads();
        TPHASE(3);
        invert_diag_blocks((lds_f64*)dgbuf, (lds_f64*)invd);
        if (rows_on) {
            const lds_f64* __restrict__ Dg = (const lds_f64*)dgbuf;
            const lds_f64* __restrict__ iv = (const lds_f64*)invd;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (16 * b < nb) {
#pragma unroll
                    for (int p = 0; p < b; ++p)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int k = 16 * p + kq + 4 * u;
                            const double lv = Dg[k * kLdDiag + 16 * b + l15];         // L[16 b + j][k]
                            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, acc[p][u], acc[b], 0, 0, 1);
                        }
                    double4_t x = {0, 0, 0, 0};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = kq + 4 * u, j = l15;
                        double wv = 0.0;                                            // inv(L_bb)'[k][j] = inv(L_bb)[j][k]
                        if (j > k) wv = Dg[(16 * b + j) * kLdDiag + 16 * b + k];
                        else if (j == k) wv = iv[16 * b + k];
                        x = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, acc[b][u], x, 0, 0, 0);
                    }
                    acc[b] = x;
                }
            }
        }
        TPHASE(4);
        store_regs();
        publish(my_flag);
        TPHASE(5);
        return;
    }

    // ---------------------------------------------------------------------------------------
    // The walker: ONE workgroup per tiled supernode runs down its diagonal, so that the critical
    // path POTRF(J) -> TRSM of tile (J+1,J) -> SYRK into tile (J+1,J+1) -> POTRF(J+1) never
    // leaves this workgroup's LDS.  Other workgroups prepare tiles (J+1,J) and (J+1,J+1) (all
    // updates by block columns < J) and announce them with PREP flags; the walker publishes the
    // finished diagonal tiles and the tiles (J+1,J) with the ordinary flags.
    // X tile in T, diagonal block / next diagonal tile in dgbuf.
    // ---------------------------------------------------------------------------------------
    {
        double* Cb = Tflat;  // buffer that holds the current diagonal tile
        const int ti = tid & 15, tj = tid >> 4;
        if (tid < kTile) s_invd[tid] = 1.0;  // (entries past a narrow last block are never written by the POTRF)
        for (int J = 0; J < nbc; ++J) {
            const int col0 = J * kTile, nb = min(kTile, w - col0);
            TRACE(J, 0);
            const bool has_next = J + 1 < nbc;
            const bool tail_rows = !has_next && nb < kTile && col0 + nb < r;
            const int f_diag = D.tflag0 + J * nbc + J;
            const int row1 = col0 + kTile;  // tiles (J+1,J) and (J+1,J+1)
            const int f_b = D.tflag0 + (J + 1) * nbc + J, f_c = f_b + 1;
            // prepared tile (J+1,J) -> registers -> T, prepared diagonal tile (J+1,J+1) -> registers
            double bv[kSub * kSub / 64], cpart[kSub * kSub / 64];
            auto load_b = [&]() {  // tile (J+1,J), all four quadrants
                const int sr = row1 + kSub * wa, nr = min(kSub, r - sr), scb = col0 + kSub * wb;
#pragma unroll
                for (int q = 0; q < kSub * kSub / 64; ++q) {
                    const int e = q * 64 + lane;
                    const int cc = e >> 5, rr = e & 31;
                    bv[q] = (rr < nr) ? ld_sc1(&G[(int64_t)(scb + cc) * ld + sr + rr]) : 0.0;
                }
            };
            auto load_c = [&]() {  // diagonal tile (J+1,J+1), lower part
                const int sr = row1 + kSub * wa, nr = min(kSub, r - sr);
                const int scc = row1 + kSub * wb, ncc = min(kSub, w - scc);
#pragma unroll
                for (int q = 0; q < kSub * kSub / 64; ++q) {
                    const int e = q * 64 + lane;
                    const int cc = e >> 5, rr = e & 31;
                    const bool inc = rr < nr && cc < ncc && (sr + rr >= scc + cc);
                    cpart[q] = inc ? ld_sc1(&G[(int64_t)(scc + cc) * ld + sr + rr]) : 0.0;
                }
            };
            if (tid == 0) {
                s_cnt = 0;  // arrival counters of this step's two barrier-free publications
                s_cnt2 = 0;
            }
            int bad;
#ifdef PARSY_STAMPS
            if (tid == 0 && J < 512) g_trace[J * 16 + 8] = clock64();
#endif
#ifndef PARSY_WALKER_POTRF_PANEL   // the 4-columns-per-step form over the 256 threads' register blocks (default)
            double a[4][4];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    const int i = 4 * ti + ri, c = 4 * tj + ci;
                    double v = (i == c) ? 1.0 : 0.0;
                    if (c < nb && i < nb && i >= c) v = cell(Cb, i, c);
                    a[ri][ci] = v;
                }
            potrf64_regs(a, colbuf, ti, tj, nb, bad, s_invd);
#else
            // (the 16-column panel form: built and measured in round 3 -- 14.8 vs 14.5 us per 64 x 64 block on the
            // nd24k-class input: one wave is bound by the instructions it issues, not by the pivots' chain -- not the default)
            potrf64_panel(Cb, nb, bad, s_invd, colbuf);
            double a[4][4];   // thread (ti, tj)'s 4x4 block of the factor, as the rest of the step takes it
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    const int i = 4 * ti + ri, c = 4 * tj + ci;
                    a[ri][ci] = (c < nb && i < nb && i >= c) ? cell(Cb, i, c) : 0.0;
                }
            __syncthreads();   // (the buffer that held the tile may be the one Ljj is laid out in below)
#endif
#ifdef PARSY_STAMPS
            if (tid == 0 && J < 512) g_trace[J * 16 + 9] = clock64();
#endif
            TRACE(J, 2);
            if (bad) atomicMin(info, D.c0 + col0 + bad);  // only threads that saw a bad pivot
            // the factored block goes to the panel (the part above the diagonal is zero since the
            // panel was assembled and nobody writes there)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    const int i = 4 * ti + ri, c = 4 * tj + ci;
                    if (c < nb && i < nb && i >= c) st_sc1(&G[(int64_t)(col0 + c) * ld + col0 + i], a[ri][ci]);
                }
            if (!has_next && !tail_rows) {
                publish(f_diag);
                TRACE(J, 3);
                break;
            }
            // Ljj into the block buffer straight from the registers
            double* Dg = (tail_rows && Cb != Tflat) ? Tflat : dgbuf;
            auto fill_dg = [&]() {
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri) {
                        const int i = 4 * ti + ri, c = 4 * tj + ci;
                        Dg[c * kLdDiag + i] = (c < nb && i < nb && i >= c) ? a[ri][ci] : 0.0;
                    }
            };
            if (tail_rows) {
                // a last block column narrower than the tile leaves rows of the panel below the diagonal
                // block inside this very tile: solve them here, then publish
                fill_dg();
                __syncthreads();
                invert_and_trsm((lds_f64*)Cb, (lds_f64*)Dg, (lds_f64*)invd, nb);
                write_tile(Cb, col0, col0, nb);
                publish(f_diag);
                TRACE(J, 3);
                break;
            }
            fill_dg();  // (dgbuf: the diagonal tile was gathered before the POTRF's barriers)
            {
                // every wave waits for the two prepared tiles itself (bounded) and then loads its
                // quadrants: no workgroup barrier between the poll and the loads
                const unsigned long long t0 = wall_clock64();
                int spins = 0;
                while (__hip_atomic_load(&tflags[nflags + f_b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch ||
                       __hip_atomic_load(&tflags[nflags + f_c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                    if ((++spins & 15) == 0 &&
                        (wall_clock64() - t0 > kSpinTicks ||
                         __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
                        atomicMin(info, -1);  // status < 0; go on (with whatever is there) so that nobody hangs
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#ifndef PARSY_WALKER_LDS_TRSM
            // (round 5) tile (J+1,J) straight into the accumulator layout of the TRSM below -- wave q: rows 16 q + l15 of the
            // tile, xa[g][v] = column 16 g + kq + 4 v (lanes along the rows: 128-byte segments)
            double4_t xa[4];
            {
                const int row = row1 + 16 * wave + l15;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        xa[g][v] = row < r ? ld_sc1(&G[(int64_t)(col0 + 16 * g + kq + 4 * v) * ld + row]) : 0.0;
            }
#else
            load_b();
#endif
            TRACE(J, 4);
            // Diagonal tile J is published as soon as its stores have landed, without a workgroup
            // barrier: every wave drains its own stores (behind the loads it has to wait for anyway)
            // and counts itself in LDS; the last one raises the flag.  The tiles of block column J
            // can start their TRSM.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                const int arrived = __hip_atomic_fetch_add(&s_cnt2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (arrived == kThreads / 64 - 1)
                    __hip_atomic_store(&tflags[f_diag], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#ifdef PARSY_WALKER_LDS_TRSM
#pragma unroll
            for (int q = 0; q < kSub * kSub / 64; ++q) {
                const int e = q * 64 + lane;
                Tw[(e >> 5) * kLdSub + (e & 31)] = bv[q];
            }
#endif
            __syncthreads();  // Ljj (and, LDS form, the tile) are in LDS
            TRACE(J, 3);
            load_c();  // lands behind the TRSM
            invert_diag_blocks((lds_f64*)dgbuf, (lds_f64*)invd);
            TRACE(J, 10);
#ifndef PARSY_WALKER_LDS_TRSM
            {
                // The blocked TRSM on the accumulators, as the other tiles do it ("finishes in REGISTERS"): in the transposed
                // form X_b' = inv(L_bb) (B_b' - sum_{p<b} L_bp X_p') register u of a 16-column block IS the operand of k step u
                // of the next product -- no LDS round trip between the four blocks (the LDS form: 3.2 us of the step) --, then X
                // goes to the panel from the registers and, for the SYRK, into the tile buffer.
                const lds_f64* __restrict__ Dgl = (const lds_f64*)dgbuf;
                const lds_f64* __restrict__ iv = (const lds_f64*)invd;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int p = 0; p < b; ++p)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int k = 16 * p + kq + 4 * u;
                            const double lv = Dgl[k * kLdDiag + 16 * b + l15];         // L[16 b + j][k]
                            xa[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, xa[p][u], xa[b], 0, 0, 1);
                        }
                    double4_t x = {0, 0, 0, 0};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = kq + 4 * u, j = l15;
                        double wv = 0.0;                                            // inv(L_bb)'[k][j] = inv(L_bb)[j][k]
                        if (j > k) wv = Dgl[(16 * b + j) * kLdDiag + 16 * b + k];
                        else if (j == k) wv = iv[16 * b + k];
                        x = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, xa[b][u], x, 0, 0, 0);
                    }
                    xa[b] = x;
                }
                TRACE(J, 5);
                const int row = row1 + 16 * wave + l15;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int c = 16 * g + kq + 4 * v;
                        if (row < r) st_sc1(&G[(int64_t)(col0 + c) * ld + row], xa[g][v]);
                        cell(Tflat, 16 * wave + l15, c) = xa[g][v];
                    }
                __syncthreads();   // X is in the tile buffer: the SYRK reads other waves' rows
            }
#else
            invert_and_trsm_inline<false>((lds_f64*)Tflat, (lds_f64*)dgbuf, (lds_f64*)invd, kTile);
            TRACE(J, 5);
#endif
#ifdef PARSY_WALKER_LDS_TRSM
            {   // X = final tile (J+1,J): this wave's quadrant, all LDS reads first, then the stores
                const int sr = row1 + kSub * wa, nr = min(kSub, r - sr), scx = col0 + kSub * wb;
                double xq[kSub * kSub / 64];
#pragma unroll
                for (int q = 0; q < kSub * kSub / 64; ++q) {
                    const int e = q * 64 + lane;
                    xq[q] = Tw[(e >> 5) * kLdSub + (e & 31)];
                }
#pragma unroll
                for (int q = 0; q < kSub * kSub / 64; ++q) {
                    const int e = q * 64 + lane;
                    const int cc = e >> 5, rr = e & 31;
                    if (rr < nr) st_sc1(&G[(int64_t)(scx + cc) * ld + sr + rr], xq[q]);
                }
            }
#endif
            TRACE(J, 11);
            // X is published the same way (drain half way through the SYRK below).
            auto publish_x_wave = [&]() {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    const int arrived = __hip_atomic_fetch_add(&s_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (arrived == kThreads / 64 - 1)
                        __hip_atomic_store(&tflags[f_b], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            // next diagonal tile: C' - X X' on this wave's quadrant (dgbuf is free again)
            {
                double* __restrict__ Cq = dgbuf + wave * (kSub * kLdSub);
#pragma unroll
                for (int q = 0; q < kSub * kSub / 64; ++q) {
                    const int e = q * 64 + lane;
                    Cq[(e >> 5) * kLdSub + (e & 31)] = cpart[q];
                }
                if (wa >= wb) {
                    double4_t s00 = {0, 0, 0, 0}, s01 = {0, 0, 0, 0}, s10 = {0, 0, 0, 0}, s11 = {0, 0, 0, 0};
                    for (int half = 0; half < 2; ++half) {
#pragma unroll 2
                        for (int ks = half * (kTile / 8); ks < (half + 1) * (kTile / 8); ++ks) {
                            const int k = 4 * ks + kq;
                            const double x0 = cell(Tflat, kSub * wa + l15, k), x1 = cell(Tflat, kSub * wa + 16 + l15, k);
                            const double y0 = cell(Tflat, kSub * wb + l15, k), y1 = cell(Tflat, kSub * wb + 16 + l15, k);
                            s00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y0, s00, 0, 0, 0);
                            if (wa != wb) s01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y1, s01, 0, 0, 0);
                            s10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y0, s10, 0, 0, 0);
                            s11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y1, s11, 0, 0, 0);
                        }
                        if (half == 0) TRACE(J, 12);
                    }
                    TRACE(J, 13);
                    publish_x_wave();   // (after the products: the stores of X have landed by now, the drain costs nothing)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int r0 = kq + 4 * v;
                        Cq[l15 * kLdSub + r0] -= s00[v];
                        if (wa != wb) Cq[(16 + l15) * kLdSub + r0] -= s01[v];
                        Cq[l15 * kLdSub + 16 + r0] -= s10[v];
                        Cq[(16 + l15) * kLdSub + 16 + r0] -= s11[v];
                    }
                } else {
                    publish_x_wave();
                }
            }
            __syncthreads();  // the next diagonal tile is complete in dgbuf
            TRACE(J, 6);
            Cb = dgbuf;
        }
    }
}

// TILES: one workgroup per tile (or per part of a tile with a split stream).
__global__ __launch_bounds__(kThreads, 3) void k_chol_tiles(const SnDesc* __restrict__ sn,
                                                            const int32_t* __restrict__ relpos,
                                                            const WaveEntry* __restrict__ wents,
                                                            const int64_t* __restrict__ wptr,
                                                            const int64_t* __restrict__ split_ranges,
                                                            double* __restrict__ tile_scratch,
                                                            const TileDesc* __restrict__ tiles,
                                                            double* __restrict__ L) {
    // only the tile itself lives in LDS here (the T member comes first in TileLds; nothing else of it
    // is touched when CHAIN == false): 33 KB and <= 168 registers, so three workgroups fit a CU --
    // or two beside a workgroup of the chain launch
    __shared__ double Tonly[4 * kSub * kLdSub];
    TileLds& S = *reinterpret_cast<TileLds*>(Tonly);
    tile_task<false>(S, blockIdx.x, sn, relpos, wents, wptr, split_ranges, tile_scratch, tiles, L, nullptr, nullptr, 0,
                     0);
}

// CHAIN: one workgroup per tile of the launch's list; which one is decided by a ticket taken when the
// workgroup starts (see tile_task).  (Persistent workers -- fewer workgroups than tiles, each looping
// over tickets, to leave the side stream more room -- were slower: the levels with many supernodes
// want every slot.)
__global__ __launch_bounds__(kThreads, 2) void k_chol_chain(const SnDesc* __restrict__ sn,
                                                            const int32_t* __restrict__ relpos,
                                                            const WaveEntry* __restrict__ wents,
                                                            const int64_t* __restrict__ wptr,
                                                            const int64_t* __restrict__ split_ranges,
                                                            double* __restrict__ tile_scratch,
                                                            const TileDesc* __restrict__ tiles,
                                                            double* __restrict__ L, int* __restrict__ info,
                                                            int* __restrict__ tflags, int nflags,
                                                            int* __restrict__ ticket, int epoch) {
    __shared__ TileLds S;
    if (threadIdx.x == 0) S.s_task = atomicAdd(ticket, 1);
    __syncthreads();
    tile_task<true>(S, S.s_task, sn, relpos, wents, wptr, split_ranges, tile_scratch, tiles, L, info, tflags, nflags,
                    epoch);
}
// CHAIN, second launch of a level (Launch::fused = 1): the tiles of the rows below the diagonal squares -- every
// diagonal tile they wait for was published by the first launch.  Three workgroups per compute unit (52 KB of LDS,
// <= 168 registers): what a tile spends waiting for memory (its own load, the ring's first chunk, the diagonal
// block, its stores) runs behind the products of two others instead of one.
__global__ __launch_bounds__(kThreads, 3) void k_chol_chain_rows(const SnDesc* __restrict__ sn,
                                                                 const int32_t* __restrict__ relpos,
                                                                 const WaveEntry* __restrict__ wents,
                                                                 const int64_t* __restrict__ wptr,
                                                                 const int64_t* __restrict__ split_ranges,
                                                                 double* __restrict__ tile_scratch,
                                                                 const TileDesc* __restrict__ tiles,
                                                                 double* __restrict__ L, int* __restrict__ info,
                                                                 int* __restrict__ tflags, int nflags,
                                                                 int* __restrict__ ticket, int epoch) {
    __shared__ RowsLds S;
    if (threadIdx.x == 0) S.s_task = atomicAdd(ticket, 1);
    __syncthreads();
    tile_task<true, true>(S, S.s_task, sn, relpos, wents, wptr, split_ranges, tile_scratch, tiles, L, info, tflags,
                          nflags, epoch);
}

// ---------------------------------------------------------------------------
// BIG: the updates from wide descendants (reference DSYRK + DGEMM at
// parallel_PB_Cholesky_05.h:160,173 with K in the hundreds or thousands), and the updates
// between the pieces of a split supernode.  One workgroup per task = a super-tile of the target's
// panel and a list of entries (source, <= 128 of its rows for the tile's rows: R, <= 128 for the
// tile's columns: C, consecutive rows of its column-major panel).  Per entry R and C are staged
// through LDS in 16-wide k chunks by LDS-DMA (global_load_lds_dwordx4: no register on the way, no
// LDS store instruction), double-buffered, shared by the 8 waves (2 x 4; the 16 x 16 fragments of
// an entry are dealt evenly over them, at most 4 x 2 each: 8 accumulators of
// v_mfma_f64_16x16x4_f64); 16 flop per byte fetched against 4 for the per-wave streams.  The
// product is formed as C x R' (lanes along the target's ROWS: 128-B segments), negated, and at
// the end of a source it is added to the tile in the panel through the relative indices -- the
// tile belongs to this workgroup alone within the launch, sources in list order: fixed summation
// order.
// ---------------------------------------------------------------------------
#ifndef PARSY_BK
#define PARSY_BK 16
#endif
static constexpr int kBK = PARSY_BK;           // k extent of a staged chunk
static constexpr int kBLd = kBigTile + 16;     // k stride of a staged chunk in LDS: lanes 16..31 (k + 1) of an
                                               // operand read hit the other half of the banks
#ifndef PARSY_BIG_WC
#define PARSY_BIG_WC 4
#endif
static constexpr int kBigWC = PARSY_BIG_WC;    // waves along the tile's columns: 4 (8 waves: 2 x 4, up to 64 x 32 outputs
                                               // each; two workgroups per CU = 4 waves per SIMD, so that the start of one
                                               // task hides behind the multiplies of the others) or 2 (4 waves of 64 x 64)
static constexpr int kBigWCols = kBigTile / kBigWC;   // columns of a wave's block (32 / 64)
static constexpr int kBigNfc = kBigWCols / 16;        // 16-column fragments of it (2 / 4)
#ifndef PARSY_BIG_WR
#define PARSY_BIG_WR 2
#endif
static constexpr int kBigWR = PARSY_BIG_WR;           // waves along the tile's rows: 2, or 4 (16 waves of up to 32 x 32
                                                      // outputs: one workgroup per compute unit with PARSY_BK = 32)
static constexpr int kBigNfr = kBigTile / kBigWR / 16;   // 16-row fragments of a wave's block (4 / 2)
static constexpr int kBigWaves = kBigWR * kBigWC;
static constexpr int kBigThreads = 64 * kBigWaves;
static_assert(kBK % 4 == 0 && kBK % kBigWaves == 0 && kBigWaves >= 4, "k_chol_big: a wave stages kBK / waves columns per operand");
struct BigLds {
    double R[2][kBK * kBLd];
    double C[2][kBK * kBLd];
};

// one LDS-DMA instruction: 16 bytes per lane from the lane's own global address to lds + 16 * lane
#ifndef PARSY_GLDS_AUX
#define PARSY_GLDS_AUX 0     // (cache-policy bits of the staging loads: 2 = nt, 16 = sc1; measured: no gain, see DESIGN)
#endif
__device__ __forceinline__ void glds16(const double* g, double* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, PARSY_GLDS_AUX);
}

__global__ __launch_bounds__(kBigThreads, kBigThreads / 64 >= 16 ? 4 : kBigWC) void k_chol_big(const SnDesc* __restrict__ sn,
                                                             const int32_t* __restrict__ relpos,
                                                             const WaveEntry* __restrict__ ents,
                                                             const TileDesc* __restrict__ tasks,
                                                             double* __restrict__ L) {
    __shared__ __attribute__((aligned(16))) BigLds S;
#ifndef PARSY_BIG_NOAGPR
    // The accumulators live in AGPRs (64 of the wave's 128 registers): one inline-asm operand of class "a" makes the
    // compiler select the AGPR form of the matrix instructions (it still places every wait state itself); the
    // atomic adds of the tile update take their data straight from there.
    {
        int agpr_hint = 0;
        asm volatile("; accumulators in AGPRs %0" ::"a"(agpr_hint));
    }
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / kBigWC, wc = wave % kBigWC, l15 = lane & 15, kq = lane >> 4;
    const TileDesc td = tasks[blockIdx.x];
    const SnDesc D = sn[td.sn];
    double* __restrict__ G = L + D.px;
    const int ld = D.ld;
    const int64_t e_begin = td.wp, e_end = td.sp;
    if (e_begin >= e_end) return;
#ifdef PARSY_BIGABL_ZEROOPS
    const long long zmask = td.part == 12345 ? -1ll : 0ll;   // (0 at run time; the compiler cannot know)
#endif
#ifdef PARSY_BIGSTAMPS
    const bool st_on = (int)gridDim.x == g_bigstamp_cfg[0] && (int)blockIdx.x == g_bigstamp_cfg[1] && lane == 0 &&
                       (wave == 0 || wave == 5);
    const int st_slot = wave == 0 ? 0 : 1;
    int st_n = 0;
#endif

    // ---- loader: wave w moves columns k = w, w + 8 of both staged blocks of a chunk, lane l rows 2l and 2l + 1 of
    // a column (16 bytes: 1 KiB per instruction, one whole column of the block).  Rows past a ragged window re-read
    // its last row -- at most one row beyond it: the rows of a panel column are followed by the next column, and a
    // source's last column by the next panel of lValues (a source is never the last one) -- and land in positions
    // whose products are never stored.
    int64_t le = e_begin;
    int lk = 0;                       // k position inside entry le
    WaveEntry LE = ents[le];
    int offR = 0, offC = 0;           // this lane's first row in the two windows of entry le
    auto enter = [&]() {
        const int mi = LE.mn & 255, nj = (LE.mn >> 8) & 255;
        offR = min(2 * lane, mi - 1);
        offC = min(2 * lane, nj - 1);
    };
    enter();
    // start the DMA of the loader's next chunk into buffer b; returns its k extent (0: the task has no more chunks)
    auto fetch = [&](int b) -> int {
        if (le >= e_end) return 0;
        const int kend = min(kBK, LE.K - lk);
#ifndef PARSY_BIGABL_NOLOAD   // (diagnostic build: every chunk re-reads the entry's first one -- cache hits)
        const double* __restrict__ base = L + LE.src + (int64_t)lk * LE.ld;
#else
        const double* __restrict__ base = L + LE.src;
#endif
#pragma unroll
        for (int h = 0; h < kBK / kBigWaves; ++h) {
            const int k = wave + kBigWaves * h;
            if (k < kend) {
                const double* __restrict__ col = base + (int64_t)k * LE.ld;
                glds16(col + LE.ia + offR, &S.R[b][k * kBLd]);
                glds16(col + LE.ja + offC, &S.C[b][k * kBLd]);
            }
        }
        // ragged end of a source: the columns up to the next multiple of four are zero (k steps go by four)
        if (kend < kBK) {
            const int kz = kend + wave;
            if (kz < ((kend + 3) & ~3)) {
                S.R[b][kz * kBLd + 2 * lane] = 0.0;
                S.R[b][kz * kBLd + 2 * lane + 1] = 0.0;
                S.C[b][kz * kBLd + 2 * lane] = 0.0;
                S.C[b][kz * kBLd + 2 * lane + 1] = 0.0;
            }
        }
        lk += kBK;
        if (lk >= LE.K) {
            lk = 0;
            ++le;
            if (le < e_end) {
                LE = ents[le];
                enter();
            }
        }
        return kend;
    };

    // ---- consumer state: entry ce, progress ck
    int64_t ce = e_begin;
    int ck = 0;
    WaveEntry CE = LE;
    double4_t acc[kBigNfc][kBigNfr];  // [16-row fragment of the column window][... of the row window]
#pragma unroll
    for (int a = 0; a < kBigNfc; ++a)
#pragma unroll
        for (int b = 0; b < kBigNfr; ++b) acc[a][b] = double4_t{0, 0, 0, 0};

    // The wave's block of an entry: the 16-row fragments the source has in the two windows (NR x NC, <= 8 x 8) are
    // dealt EVENLY over the kBigWR x kBigWC waves -- ceil(NR / kBigWR) x ceil(NC / kBigWC) fragments each, from fragment
    // (fr0, fc0) on -- not in fixed 64 x 32 blocks: on the Flan-class input two thirds of the chunks have ragged
    // windows, and with fixed blocks the busiest wave of a workgroup multiplies 7.3 fragments per k step while the
    // average wave has 4.5 (tools/big_stats.py); the others wait for it at the chunk barrier.  Dealt evenly the
    // busiest wave has 5.1.  Which wave forms a product changes nothing in its value: the factor stays bit for bit
    // the same.
    auto frags = [&](const WaveEntry& E, int& nfr, int& nfc, int& r0, int& c0) {
        const int mi = E.mn & 255, nj = (E.mn >> 8) & 255;
        const int NR = (mi + 15) >> 4, NC = (nj + 15) >> 4;
        // The grid the waves form over the entry's fragments is chosen PER ENTRY among the shapes a wave's 4 x 2
        // accumulators allow -- 2 x 4 (any entry), 1 x 8 (<= 4 fragment rows), 4 x 2 (<= 4 fragment columns), 8 x 1
        // (<= 2) -- for the fewest fragments in the busiest wave: 5 x 8 fragments are 3 x 2 = 6 per wave as 2 x 4 but
        // 5 x 1 = 5 as 1 x 8; over the Flan-class input the busiest waves issue 4.2 % fewer products
        // (tools/big_stats.py; dealing single fragments round-robin would make it 6.9 % but needs every accumulator's
        // operands addressed on their own: registers the kernel does not have).
        int wrs = 1;                                   // log2 of the waves along the rows: 2 x 4
        int best = ((NR + 1) >> 1) * ((NC + 3) >> 2);
        if (kBigWR == 2 && kBigWC == 4) {
            if (NR <= 4 && NR * ((NC + 7) >> 3) < best) {            // 1 x 8
                best = NR * ((NC + 7) >> 3);
                wrs = 0;
            }
            if (NC <= 4 && ((NR + 3) >> 2) * ((NC + 1) >> 1) < best) {   // 4 x 2
                best = ((NR + 3) >> 2) * ((NC + 1) >> 1);
                wrs = 2;
            }
            if (NC <= 2 && ((NR + 7) >> 3) * NC < best) {            // 8 x 1
                best = ((NR + 7) >> 3) * NC;
                wrs = 3;
            }
        }
        static_assert(kBigWaves == 8 || (kBigWR != 2 || kBigWC != 4), "k_chol_big: the entry grids are shapes of 8 waves");
        const int wcs = (kBigWR == 2 && kBigWC == 4) ? 3 - wrs : 0;
        const int WR = (kBigWR == 2 && kBigWC == 4) ? 1 << wrs : kBigWR, WC = (kBigWR == 2 && kBigWC == 4) ? 1 << wcs : kBigWC;
        const int gwr = (kBigWR == 2 && kBigWC == 4) ? wave >> wcs : wr;
        const int gwc = (kBigWR == 2 && kBigWC == 4) ? wave & (WC - 1) : wc;   // (for 2 x 4: wr, wc)
        const int frb = (kBigWR == 2 && kBigWC == 4) ? (NR + WR - 1) >> wrs : (NR + WR - 1) / WR;
        const int fcb = (kBigWR == 2 && kBigWC == 4) ? (NC + WC - 1) >> wcs : (NC + WC - 1) / WC;
        r0 = 16 * frb * gwr;
        c0 = 16 * fcb * gwc;
        nfr = min(frb, max(0, NR - frb * gwr));
        nfc = min(fcb, max(0, NC - fcb * gwc));
        // a block whose rows all precede its columns in the source's row order lies strictly above the diagonal of
        // the target (both windows count the source's rows from the same first row)
        if (E.ia + r0 + 16 * nfr - 1 < E.ja + c0) nfr = 0;
        if (nfr == 0 || nfc == 0) nfr = nfc = 0;
    };
    // A wave whose block of the tile has no rows of this source skips the chunk; a ragged block skips the
    // 16-row fragments it does not have (wave-uniform branches: a second, branch-free copy of the loop for
    // full blocks made the compiler spill; issuing all 8 products always lost more to the ragged windows than the
    // branches cost: 428 vs 390 ms of BIG launches).  nks = k steps of four the chunk holds.
    auto compute = [&](int b, int nks, int nfr, int nfc, int r0, int c0) {
        if (nfr == 0) return;
        const double* __restrict__ Rb = &S.R[b][kq * kBLd + r0 + l15];
        const double* __restrict__ Cb = &S.C[b][kq * kBLd + c0 + l15];
        // the multiplying waves win the issue arbitration over the waves that write back (-1.7 % of the BIG
        // launches on the Flan-class input: 375 -> 369 ms, profiles/r03_big_ablation.txt)
        __builtin_amdgcn_s_setprio(1);
        // the operands of k step ks + 1 are read from LDS before the products of k step ks are issued (363 -> 357 ms of
        // BIG launches on the Flan-class input)
        double rv[2][kBigNfr], cv[2][kBigNfc];
#pragma unroll
        for (int f = 0; f < kBigNfr; ++f) rv[0][f] = Rb[16 * f];
#pragma unroll
        for (int f = 0; f < kBigNfc; ++f) cv[0][f] = Cb[16 * f];
#pragma unroll
        for (int ks = 0; ks < kBK / 4; ++ks) {
            if (ks < nks) {
                if (ks + 1 < kBK / 4 && ks + 1 < nks) {
#pragma unroll
                    for (int f = 0; f < kBigNfr; ++f) rv[(ks + 1) & 1][f] = Rb[4 * (ks + 1) * kBLd + 16 * f];
#pragma unroll
                    for (int f = 0; f < kBigNfc; ++f) cv[(ks + 1) & 1][f] = Cb[4 * (ks + 1) * kBLd + 16 * f];
                }
#ifdef PARSY_BIGABL_NOMFMA    // (diagnostic build: operands are read from LDS and dropped)
#pragma unroll
                for (int f = 0; f < kBigNfr; ++f) asm volatile("" ::"v"(rv[ks & 1][f]));
#pragma unroll
                for (int f = 0; f < kBigNfc; ++f) asm volatile("" ::"v"(cv[ks & 1][f]));
#else
#ifdef PARSY_BIGABL_ZEROOPS   // (diagnostic build: the products are formed on all-zero operands -- what the data costs)
#pragma unroll
                for (int f = 0; f < kBigNfr; ++f)
                    rv[ks & 1][f] = __longlong_as_double(__double_as_longlong(rv[ks & 1][f]) & zmask);
#pragma unroll
                for (int f = 0; f < kBigNfc; ++f)
                    cv[ks & 1][f] = __longlong_as_double(__double_as_longlong(cv[ks & 1][f]) & zmask);
#endif
#pragma unroll
                for (int fc = 0; fc < kBigNfc; ++fc) {
                    if (fc < nfc) {
#pragma unroll
                        for (int fr = 0; fr < kBigNfr; ++fr)
                            if (fr < nfr)
                                acc[fc][fr] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv[ks & 1][fc], rv[ks & 1][fr], acc[fc][fr], 0, 0, 1);
                    }
                }
#endif
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // add the finished (negated: the instruction's neg modifier on one operand, exact) product to the tile (C/D
    // layout of v_mfma_f64_16x16x4_f64 with the operands swapped: lane & 15 = row of R, (lane >> 4) + 4 reg = row of
    // C) through the relative indices, as no-return FP64 atomic adds (performed at the memory side: fire and forget
    // -- the read-modify-write this replaces was five dependent round trips per source; 369 -> 363 ms of BIG launches
    // on the Flan-class input, the factor unchanged bit for bit: old + (-p) rounds as old - p, the tile belongs to
    // this workgroup alone within the launch, one lane's adds into one address are performed in program order, and the
    // caller drains them before the next barrier, so that the order of sums per entry of L stays the list order)
    auto epilogue = [&](const WaveEntry& E, int nfr, int nfc, int r0, int c0) {
        if (nfr == 0) return;
#ifdef PARSY_BIGABL_NOEPI     // (diagnostic build: the product is dropped -- no update of the tile)
#pragma unroll
        for (int fc = 0; fc < kBigNfc; ++fc)
#pragma unroll
            for (int fr = 0; fr < kBigNfr; ++fr) {
                asm volatile("" ::"v"(acc[fc][fr]));
                acc[fc][fr] = double4_t{0, 0, 0, 0};
            }
        return;
#endif
        const int mi = E.mn & 255, nj = (E.mn >> 8) & 255;
        const bool ident = ((E.mn >> 16) & 1) != 0;
        // (the index loads are unconditional -- rows past the window re-read its last one -- so that all twelve are
        // in flight together: guarded per lane, the compiler waited for each of them in turn; loading them before the
        // source's last chunk is multiplied, so that they land behind it, cost more in registers than it hid)
        int prow[kBigNfr], pcol[kBigNfc][4];
        if (ident) {
#pragma unroll
            for (int fr = 0; fr < kBigNfr; ++fr) prow[fr] = E.ia + r0 + 16 * fr + l15;
#pragma unroll
            for (int fc = 0; fc < kBigNfc; ++fc)
#pragma unroll
                for (int v = 0; v < 4; ++v) pcol[fc][v] = E.ja + c0 + 16 * fc + kq + 4 * v;
        } else {
            const int32_t* __restrict__ rpi = relpos + (int64_t)E.rel + E.ia;
            const int32_t* __restrict__ rpj = relpos + (int64_t)E.rel + E.ja;
#pragma unroll
            for (int fr = 0; fr < kBigNfr; ++fr) prow[fr] = rpi[min(r0 + 16 * fr + l15, mi - 1)];
#pragma unroll
            for (int fc = 0; fc < kBigNfc; ++fc)
#pragma unroll
                for (int v = 0; v < 4; ++v) pcol[fc][v] = rpj[min(c0 + 16 * fc + kq + 4 * v, nj - 1)];
#pragma unroll
            for (int fr = 0; fr < kBigNfr; ++fr) prow[fr] -= D.rbias;
#pragma unroll
            for (int fc = 0; fc < kBigNfc; ++fc)
#pragma unroll
                for (int v = 0; v < 4; ++v) pcol[fc][v] -= D.rbias;
        }
#pragma unroll
        for (int fr = 0; fr < kBigNfr; ++fr)
            if (fr >= nfr || r0 + 16 * fr + l15 >= mi) prow[fr] = -1;
#pragma unroll
        for (int fc = 0; fc < kBigNfc; ++fc)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (fc >= nfc || c0 + 16 * fc + kq + 4 * v >= nj) pcol[fc][v] = -1;
#pragma unroll
        for (int fc = 0; fc < kBigNfc; ++fc) {
            if (fc < nfc) {
#pragma unroll
                for (int fr = 0; fr < kBigNfr; ++fr)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const bool ok = prow[fr] >= 0 && pcol[fc][v] >= 0 && prow[fr] >= pcol[fc][v];
                        if (ok) unsafeAtomicAdd(&G[(int64_t)pcol[fc][v] * ld + prow[fr]], acc[fc][fr][v]);
                    }
            }
#pragma unroll
            for (int fr = 0; fr < kBigNfr; ++fr) acc[fc][fr] = double4_t{0, 0, 0, 0};
        }
    };

    // ---- pipeline: chunk n is multiplied from LDS buffer n & 1 while the DMA of chunk n + 1 fills the other buffer
    // (started right after the barrier that ended the reads of that buffer; __syncthreads() waits for this wave's
    // DMA -- it counts as vector memory traffic -- before the barrier)
    int kcur = fetch(0);
    // (explicit: the LDS-DMA of this wave has landed before the barrier -- the compiler emits the same wait for
    // __syncthreads() today, but the memory model does not oblige it to: ADVICE round 3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int nfr, nfc, r0, c0;
    frags(CE, nfr, nfc, r0, c0);
    for (int n = 0;; ++n) {
        BIGSTAMP(0);
        const int knext = fetch((n + 1) & 1);
        BIGSTAMP(2);
        compute(n & 1, (kcur + 3) >> 2, nfr, nfc, r0, c0);
        BIGSTAMP(3);
#ifdef PARSY_BIGSTAMPS
        if (st_on) {
            g_bigtrace[((size_t)st_slot * 1024 + st_n) * 8 + 6] = (unsigned long long)(nfr * nfc);
            g_bigtrace[((size_t)st_slot * 1024 + st_n) * 8 + 7] = (unsigned long long)ck | ((unsigned long long)CE.K << 32);
        }
#endif
        ck += kBK;
        if (ck >= CE.K) {
            epilogue(CE, nfr, nfc, r0, c0);
            // the adds of this source are performed before the barrier: the next source's adds into the same entries
            // of L may come from other waves (another row map), and the order of sums must not depend on timing
            // (after the task's last source the end of the kernel does it)
            if (ce + 1 < e_end) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ck = 0;
            ++ce;
            if (ce < e_end) {
                CE = ents[ce];
                frags(CE, nfr, nfc, r0, c0);
            }
        }
        BIGSTAMP(4);
        if (knext == 0) break;
        kcur = knext;
#ifndef PARSY_BIGABL_NOBARRIER   // (diagnostic build: waves race through the staged chunks -- wrong results)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of the next chunk and its tile adds
        __syncthreads();
#endif
        BIGSTAMP(5);
#ifdef PARSY_BIGSTAMPS
        if (st_n < 1023) ++st_n;
#endif
    }
}

// ---------------------------------------------------------------------------
// DENSE: the BIG entries that are full 128 x 128 blocks of a source's rows (70 % of the BIG launches' products on the
// Flan-class input; the reference hands these to DGEMM:
// parallel_PB_Cholesky_05.h:173, and to threaded BLAS-3 in its root phase, :269-411).  A kernel of its own, not a
// second path of k_chol_big: no ragged window, so no wave-uniform branch per product, no clamp per load; and a loop
// built so that a wave alone keeps its matrix pipe busy --
//   * 4 waves (256 threads), each a 64 x 64 block of the product (16 accumulators of v_mfma_f64_16x16x4_f64 in
//     AGPRs: 8 LDS operand reads per 16 products), two workgroups per compute unit;
//   * k chunks of 8 in a RING OF FOUR LDS slots (the same 72 KB as k_chol_big's two chunks of 16), filled by LDS-DMA
//     three chunks ahead; a wave waits for ITS OWN DMA of the next chunk (s_waitcnt vmcnt(4): the chunk after that may
//     stay in flight) and then meets the others at ONE s_barrier per chunk, placed between the chunk's two k steps:
//     the operands of the second k step are in registers by then, and the barrier that makes chunk n + 1 visible also
//     frees the slot of chunk n - 1 for the DMA of chunk n + 3 -- so the first operands of chunk n + 1 are read while
//     the second k step of chunk n is multiplied, and nothing waits at the chunk boundary;
//   * a source's ragged last chunk: the DMA re-reads its last column for the columns it does not have and the lanes of
//     those k positions multiply by zero (a select on the operand, no branch).
// The product is formed negated and added to the tile with no-return FP64 atomic adds as in k_chol_big (a block that
// straddles the target's diagonal is multiplied whole and its upper part dropped here).  A task's dense entries come
// first in its entry list, in update order; the launch of the ragged rest (k_chol_big) follows on the same stream, so
// the order of sums per entry of L is fixed: dense entries, then ragged ones, each in the reference's update order.
// ---------------------------------------------------------------------------
static constexpr int kDK = 8;                        // k extent of a chunk
static constexpr int kDSlots = 4;                    // ring depth
static constexpr int kDOp = kDK * kBLd;              // doubles of one operand of a chunk (k stride kBLd: see kBLd)
static constexpr int kDSlot = 2 * kDOp;              // a slot: the block's rows (R), then its columns (C)
static constexpr int kDenseThreads = 256;
static constexpr int kDStrip = kDSlots * kDSlot;     // first double of the strips' ring (256 per slot)
#ifdef PARSY_DENSESTAMPS
// (diagnostic build) shader clocks per phase of the chunk loop, summed over every wave of every dense launch
__device__ unsigned long long g_densephase[16];
// (PARSY_DENSESTAMPS is a bit mask of the probes that are compiled in: every probe is a scalar memory operation with a
// wait for everything the wave has in flight at the LDS, so few probes distort less)
#define DSTAMP(i) do { if ((PARSY_DENSESTAMPS >> (i)) & 1) { const unsigned long long now_ = __builtin_readcyclecounter(); dph[i] += now_ - dlast; dlast = now_; } } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif

// One task of k_chol_dense.  STRIPS: entries of the task carry strips (schedule.hpp: big_strip_rows / big_strip_cols) -- up
// to 16 rows of the source right behind the block's row window and / or 16 right behind its column window.  A strip is
// staged like the block's operands, by one more DMA instruction per chunk (wave 0: the row strip, wave 1: the column
// strip; [k][16 rows], 1 KiB, in a ring of its own behind the operands' -- issued ahead of the chunk's four
// instructions, so that the wait which lets those four stay in flight has seen the strip land), and every wave multiplies
// two 16 x 16 fragments of each strip with operands it holds anyway: the row strip times two of its four column
// fragments, two of its four row fragments times the column strip (waves (r0, c0) = (0, 0), (0, 64), (64, 0), (64, 64)
// take fragments 0-1 / 0-1 / 2-3 / 2-3 of their 64 columns resp. rows: together 128) -- four accumulators more, in
// VGPRs, two operand reads more per k step.
template <bool STRIPS>
__device__ __forceinline__ void chol_dense_task(double* __restrict__ S, const TileDesc& td, const SnDesc* __restrict__ sn,
                                                const int32_t* __restrict__ relpos, const WaveEntry* __restrict__ ents,
                                                double* __restrict__ L) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int r0 = 64 * (wave >> 1), c0 = 64 * (wave & 1);   // this wave's block of the 128 x 128 product
#ifdef PARSY_DENSESTAMPS
    unsigned long long dph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dlast = __builtin_readcyclecounter();
#endif
    const int64_t e_begin = td.wp, e_end = td.sp;
    const SnDesc D = sn[td.sn];
    double* __restrict__ G = L + D.px;
    const int ld = D.ld;
    const int total = td.part & ~kDenseStripTask;     // chunks of the task: sum over its entries of ceil(K / 8) (host)

    // ---- loader: wave w moves columns k = w, w + 4 of both operands of a chunk (lane l: rows 2l, 2l + 1: one
    // 1-KiB column per instruction), always four instructions per chunk, so that vmcnt counts chunks.  The four are
    // issued one by one BETWEEN the products of the chunk's second k step (issue(slot, 0..3), then advance()): a
    // vector-memory instruction that finds the memory pipeline's queue full blocks its wave, and a wave that has just
    // issued four products leaves the matrix pipe busy meanwhile.
    int64_t le = e_begin;
    int lk = 0;
    WaveEntry LE = ents[le];
    // this lane's rows in column (lk + wave) of the two windows.  (A window of fewer than 128 rows -- the few ragged
    // entries a launch hands to this kernel rather than to a launch of their own -- re-reads its last row, at most one
    // row beyond it as in k_chol_big; those products are formed and never stored.)
    auto lane_ptr = [&](const WaveEntry& E, bool cols) {
        const int rows = cols ? (E.mn >> 8) & 255 : E.mn & 255;
        return L + E.src + (cols ? E.ja : E.ia) + min(2 * lane, rows - 1) + (int64_t)wave * E.ld;
    };
    const double* __restrict__ lpR = lane_ptr(LE, false);
    const double* __restrict__ lpC = lane_ptr(LE, true);
    // (live = false: the task has no chunk n + 3 -- the instruction is issued all the same, from a valid address into
    // the slot that chunk would have had (free: everybody has read chunk n - 1, nobody reads it again), so that every
    // iteration issues exactly four and the loop is ONE path: a second copy of the product block for the last
    // iterations made the register allocator move the accumulators through scratch)
    auto issue = [&](int slot, int i, bool live) {
        double* __restrict__ dst = &S[slot * kDSlot + (i & 1) * kDOp + (wave + 4 * (i >> 1)) * kBLd];
        const double* __restrict__ src = (i & 1) ? lpC : lpR;
        // a source's ragged last chunk re-reads its last column for the columns it does not have (their products are
        // masked out): klast = last column it has, counted from lk
        const int klast = LE.K - 1 - lk;
        int64_t off = (i >> 1) ? 4 * (int64_t)LE.ld : 0;
        if (klast < kDK - 1) off = (int64_t)(min(wave + 4 * (i >> 1), klast) - wave) * LE.ld;
        if (!live) {
            src = G + 2 * lane;
            off = 0;
        }
#if defined(PARSY_DENSEABL_NODMA)        // (diagnostic builds: wrong results) no staging at all: what the loop costs without it
        (void)src; (void)off; (void)dst;
#elif defined(PARSY_DENSEABL_SAMECHUNK)  // every chunk re-reads the task's first one: staging from cache hits
        glds16(L + ents[e_begin].src + ((i & 1) ? ents[e_begin].ja : ents[e_begin].ia) + 2 * lane, dst);
#else
        glds16(src + off, dst);
#endif
    };
    // ---- strips (STRIPS only): one more DMA instruction per chunk and strip -- wave 0 the row strip's, wave 1 the column
    // strip's: lane l = rows 2 (l & 7), + 1 of k column l >> 3, 1 KiB that lands as [k][16 rows] behind the ring (kDStrip)
    auto issue_strip = [&](int slot) {
        if (wave < 2 && (LE.mn >> 17) != 0) {
            const int ext = wave ? (LE.mn >> 22) & 31 : (LE.mn >> 17) & 31;   // (a strip the entry does not have, rows past a
            const int klast = LE.K - 1 - lk;                                  // strip's last, columns past the source's last:
            const double* __restrict__ src = L + LE.src + (wave ? LE.ja : LE.ia) + kBigTile + min(2 * (lane & 7), max(ext, 1) - 1) +
                                             (int64_t)(lk + min(lane >> 3, klast)) * LE.ld;   // re-read, never stored)
            glds16(src, &S[kDStrip + slot * 256 + wave * 128]);
        }
    };
    auto advance = [&]() {
        lk += kDK;
        lpR += (int64_t)kDK * LE.ld;
        lpC += (int64_t)kDK * LE.ld;
        if (lk >= LE.K) {
            lk = 0;
            ++le;
            if (le < e_end) {
                LE = ents[le];
                lpR = lane_ptr(LE, false);
                lpC = lane_ptr(LE, true);
            }
        }
    };
    auto fetch = [&](int slot) {
        if (STRIPS) issue_strip(slot);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue(slot, i, true);
        advance();
    };

    // ---- consumer
    int64_t ce = e_begin;
    int ck = 0;
    WaveEntry CE = LE;
    double4_t acc[4][4];   // [16-row fragment of the column window (C)][... of the row window (R)]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = double4_t{0, 0, 0, 0};
    // strips: (row strip) x (column fragments 2 * hi_c, + 1 of this wave), (row fragments 2 * hi_r, + 1) x (column strip)
    double4_t rs0 = {0, 0, 0, 0}, rs1 = {0, 0, 0, 0}, cs0 = {0, 0, 0, 0}, cs1 = {0, 0, 0, 0};
    const bool hi_c = r0 != 0, hi_r = c0 != 0;
    // The operand reads and their waits are written out (inline assembly): the compiler waits for EVERY outstanding
    // LDS read whenever it needs one of them (s_waitcnt lgkmcnt(0) at the loop header and before each product block --
    // a full LDS round trip twice per chunk); here each wait names how many younger reads may stay in flight.  The
    // registers a read fills pass through the wait statement ("+v"), so that no use can be scheduled ahead of it.
    struct Ops { double2_t r01, r23, c01, c23; double rs, cs; };   // 16-row fragments 0..3 of the wave's rows (R) and columns (C), one k each; the strips' rows
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)(&S[0]);
    const unsigned roff = lds0 + 8u * (unsigned)(kq * kBLd + r0 + l15), coff = lds0 + 8u * (unsigned)(kDOp + kq * kBLd + c0 + l15);
    auto read_ops = [&](int slot, int kstep, Ops& o) {
        const unsigned so = (unsigned)(slot * kDSlot + 4 * kstep * kBLd) * 8u;
        const unsigned ar = roff + so, ac = coff + so;
        asm volatile("ds_read2_b64 %0, %1 offset1:16" : "=v"(o.r01) : "v"(ar));
        asm volatile("ds_read2_b64 %0, %1 offset0:32 offset1:48" : "=v"(o.r23) : "v"(ar));
        asm volatile("ds_read2_b64 %0, %1 offset1:16" : "=v"(o.c01) : "v"(ac));
        asm volatile("ds_read2_b64 %0, %1 offset0:32 offset1:48" : "=v"(o.c23) : "v"(ac));
        if (STRIPS) {   // (row l15 of k column 4 kstep + kq of the two strips)
            const unsigned as = lds0 + 8u * (unsigned)(kDStrip + slot * 256 + (4 * kstep + kq) * 16 + l15);
            asm volatile("ds_read_b64 %0, %1" : "=v"(o.rs) : "v"(as));
            asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(o.cs) : "v"(as));
        }
    };
    // (wait until at most the reads issued after o's are in flight: four, with strips six)
#define PARSY_DENSE_WAIT(o, n)                                                                                        \
    do {                                                                                                              \
        if (STRIPS) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(o.r01), "+v"(o.r23), "+v"(o.c01), "+v"(o.c23), "+v"(o.rs), "+v"(o.cs)); \
        else asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(o.r01), "+v"(o.r23), "+v"(o.c01), "+v"(o.c23));         \
    } while (0)
    // a source's ragged last chunk: lanes of k positions it does not have multiply by zero
    auto mask_ops = [&](Ops& o, bool ok) {
        o.r01 = ok ? o.r01 : double2_t{0, 0};
        o.r23 = ok ? o.r23 : double2_t{0, 0};
        o.c01 = ok ? o.c01 : double2_t{0, 0};
        o.c23 = ok ? o.c23 : double2_t{0, 0};
    };
    // the four products of column fragment fc
    auto products4 = [&](const Ops& o, int fc) {
        const double rv[4] = {o.r01[0], o.r01[1], o.r23[0], o.r23[1]};
        const double cv[4] = {o.c01[0], o.c01[1], o.c23[0], o.c23[1]};
#pragma unroll
        for (int fr = 0; fr < 4; ++fr)
            acc[fc][fr] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv[fc], rv[fr], acc[fc][fr], 0, 0, 1);
    };
    auto products = [&](const Ops& o) {
#pragma unroll
        for (int fc = 0; fc < 4; ++fc) products4(o, fc);
    };
    // the strips of the consumer's entry (wave-uniform branches: an entry has a strip or not)
    auto strip_products = [&](const Ops& o, bool rows, bool cols) {
        // (written out: these four accumulators live in VGPRs -- the 128 AGPRs a wave has at two workgroups per
        // compute unit hold the block's sixteen, and the compiler gives every matrix instruction of a function the same
        // accumulator file.  Same instruction, same negated first operand as the builtin's blgp = 1.)
        if (rows) {
            const double ca = hi_c ? o.c23[0] : o.c01[0], cb = hi_c ? o.c23[1] : o.c01[1];
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0 neg:[1,0,0]" : "+v"(rs0) : "v"(ca), "v"(o.rs));
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0 neg:[1,0,0]" : "+v"(rs1) : "v"(cb), "v"(o.rs));
        }
        if (cols) {
            const double ra = hi_r ? o.r23[0] : o.r01[0], rb = hi_r ? o.r23[1] : o.r01[1];
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0 neg:[1,0,0]" : "+v"(cs0) : "v"(o.cs), "v"(ra));
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0 neg:[1,0,0]" : "+v"(cs1) : "v"(o.cs), "v"(rb));
        }
    };
    auto epilogue = [&](const WaveEntry& E) {
        const bool ident = ((E.mn >> 16) & 1) != 0;
        const int mi = E.mn & 255, nj = (E.mn >> 8) & 255;
        int prow[4], pcol[4][4];
        if (ident) {
#pragma unroll
            for (int fr = 0; fr < 4; ++fr) prow[fr] = E.ia + r0 + 16 * fr + l15;
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int v = 0; v < 4; ++v) pcol[fc][v] = E.ja + c0 + 16 * fc + kq + 4 * v;
        } else {
            // (unconditional loads, rows past a ragged window re-read its last index: all twenty in flight together)
            const int32_t* __restrict__ rpi = relpos + (int64_t)E.rel + E.ia;
            const int32_t* __restrict__ rpj = relpos + (int64_t)E.rel + E.ja;
#pragma unroll
            for (int fr = 0; fr < 4; ++fr) prow[fr] = rpi[min(r0 + 16 * fr + l15, mi - 1)];
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int v = 0; v < 4; ++v) pcol[fc][v] = rpj[min(c0 + 16 * fc + kq + 4 * v, nj - 1)];
#pragma unroll
            for (int fr = 0; fr < 4; ++fr) prow[fr] -= D.rbias;
#pragma unroll
            for (int fc = 0; fc < 4; ++fc)
#pragma unroll
                for (int v = 0; v < 4; ++v) pcol[fc][v] -= D.rbias;
        }
        // rows / columns the window does not have; the part of a block above the target's diagonal
#pragma unroll
        for (int fr = 0; fr < 4; ++fr)
            if (r0 + 16 * fr + l15 >= mi) prow[fr] = -1;
#pragma unroll
        for (int fc = 0; fc < 4; ++fc)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (c0 + 16 * fc + kq + 4 * v >= nj) pcol[fc][v] = 0x7fffffff;
#pragma unroll
        for (int fc = 0; fc < 4; ++fc) {
#pragma unroll
            for (int fr = 0; fr < 4; ++fr) {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (prow[fr] >= pcol[fc][v])
                        unsafeAtomicAdd(&G[(int64_t)pcol[fc][v] * ld + prow[fr]], acc[fc][fr][v]);
                acc[fc][fr] = double4_t{0, 0, 0, 0};
            }
        }
        if (STRIPS) {
            const int ms = (E.mn >> 17) & 31, ns = (E.mn >> 22) & 31;
            // (the strips' accumulators were written by matrix instructions the compiler has not seen: the wait states
            // before a vector instruction may read them -- long over after the block's own updates, but not its to know)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(rs0), "+v"(rs1), "+v"(cs0), "+v"(cs1));
            if (ms) {   // rows ia + 128 + l15 (< ms) x this wave's column fragments 2 * hi_c, + 1
                int ps = ident ? E.ia + kBigTile + l15 : relpos[(int64_t)E.rel + E.ia + kBigTile + min(l15, ms - 1)] - D.rbias;
                if (l15 >= ms) ps = -1;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int pa = hi_c ? pcol[2][v] : pcol[0][v], pb = hi_c ? pcol[3][v] : pcol[1][v];
                    if (ps >= pa) unsafeAtomicAdd(&G[(int64_t)pa * ld + ps], rs0[v]);
                    if (ps >= pb) unsafeAtomicAdd(&G[(int64_t)pb * ld + ps], rs1[v]);
                }
                rs0 = rs1 = double4_t{0, 0, 0, 0};
            }
            if (ns) {   // this wave's row fragments 2 * hi_r, + 1 x columns ja + 128 + kq + 4 v (< ns)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int cc = kq + 4 * v;
                    int pc = ident ? E.ja + kBigTile + cc : relpos[(int64_t)E.rel + E.ja + kBigTile + min(cc, ns - 1)] - D.rbias;
                    if (cc >= ns) pc = 0x7fffffff;
                    const int pa = hi_r ? prow[2] : prow[0], pb = hi_r ? prow[3] : prow[1];
                    if (pa >= pc) unsafeAtomicAdd(&G[(int64_t)pc * ld + pa], cs0[v]);
                    if (pb >= pc) unsafeAtomicAdd(&G[(int64_t)pc * ld + pb], cs1[v]);
                }
                cs0 = cs1 = double4_t{0, 0, 0, 0};
            }
        }
    };

    // ---- prologue: chunks 0, 1, 2 on their way; chunk 0 landed and visible; the operands of its two k steps read
    fetch(0);
    if (total > 1) fetch(1);
    else
        for (int i = 0; i < 4; ++i) issue(1, i, false);
    if (total > 2) fetch(2);
    else
        for (int i = 0; i < 4; ++i) issue(2, i, false);
    // (waves 0 / 1 of a task with strips have up to three more instructions in flight, each AHEAD of its chunk's four: the
    // first seven -- chunk 0 and its strip -- have landed all the same)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    Ops A, B;
    read_ops(0, 0, A);
    read_ops(0, 1, B);
    PARSY_DENSE_WAIT(A, 4);
    if (CE.K < 4) mask_ops(A, kq < CE.K);
    DSTAMP(10);   // (entry -> first products: descriptors, the first chunk's round trip)
    // Software pipeline over the k steps: every LDS read is issued one whole k step (16 products) before its
    // operands are needed, into the registers the products issued just before have finished reading, so that a wave
    // never waits for the LDS.  State at the top of iteration n: A = the operands of chunk n's first k step, landed
    // and masked; B = those of its second k step, on their way.
    for (int n = 0; n < total; ++n) {
        const int kend = CE.K - ck;            // columns the source has from this chunk on (>= 1)
        const int nslot = (n + 1) & (kDSlots - 1);   // (after the last chunk: reads of a stale slot, never used)
        DSTAMP(0);
        __builtin_amdgcn_s_setprio(1);   // (the multiplying waves win the issue arbitration, as in k_chol_big)
        products(A);
        if (STRIPS) strip_products(A, ((CE.mn >> 17) & 31) != 0, ((CE.mn >> 22) & 31) != 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        DSTAMP(1);
        // chunk n + 1: this wave's part has landed (the four instructions of chunk n + 2 -- or their placeholders --
        // may stay in flight), then everybody's
        // (a strip's instruction is issued ahead of its chunk's four: the same count holds for the waves that move strips)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        DSTAMP(2);
#ifndef PARSY_DENSEABL_NOBARRIER   // (diagnostic build: waves race through the ring -- wrong results)
        asm volatile("s_barrier" ::: "memory");
#endif
        DSTAMP(3);
        DSTAMP(4);
        read_ops(nslot, 0, A);
        PARSY_DENSE_WAIT(B, 4);
        __builtin_amdgcn_sched_barrier(0);
        DSTAMP(5);
        if (kend < kDK) mask_ops(B, kq + 4 < kend);   // (only a source's last chunk can be ragged)
        {
            const bool live = n + 3 < total;
            const int fslot = (n + 3) & (kDSlots - 1);
            if (STRIPS) {   // the strips of chunk n + 3, ahead of its four
                if (live) issue_strip(fslot);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                products4(B, g);
                __builtin_amdgcn_sched_barrier(0);
                issue(fslot, g, live);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (live) advance();
        }
        if (STRIPS) strip_products(B, ((CE.mn >> 17) & 31) != 0, ((CE.mn >> 22) & 31) != 0);
        __builtin_amdgcn_sched_barrier(0);
        DSTAMP(6);
        ck += kDK;
        if (ck >= CE.K) {
#ifndef PARSY_DENSEABL_NOEPI        // (diagnostic build: the products are dropped)
            epilogue(CE);
#else
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    asm volatile("" ::"v"(acc[a][b]));
                    acc[a][b] = double4_t{0, 0, 0, 0};
                }
#endif
            // the adds of this source are performed before the next barrier: the next source's adds into the same
            // entries of L may come from other waves (another row map), and the order of sums must not depend on timing
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ck = 0;
            ++ce;
            if (ce < e_end) {
                CE = ents[ce];
                // (waited for HERE: a scalar load still pending at the loop header would make the compiler wait for
                // every outstanding LDS read there)
                asm volatile("" ::"s"(CE.K), "s"(CE.mn), "s"(CE.ia), "s"(CE.ja), "s"(CE.rel));
            }
            DSTAMP(7);
        }
        read_ops(nslot, 1, B);
        PARSY_DENSE_WAIT(A, 4);
        __builtin_amdgcn_sched_barrier(0);
        if (CE.K - ck < 4) mask_ops(A, kq < CE.K - ck);
        DSTAMP(8);
    }
#ifdef PARSY_DENSESTAMPS
    if (lane == 0) {
        for (int i = 0; i < 9; ++i) atomicAdd(&g_densephase[i], dph[i]);
        atomicAdd(&g_densephase[9], (unsigned long long)total);
        atomicAdd(&g_densephase[10], dph[10]);
        atomicAdd(&g_densephase[11], 1ull);   // (waves x tasks)
    }
#endif
#undef PARSY_DENSE_WAIT
}

__global__ __launch_bounds__(kDenseThreads, 2) void k_chol_dense(const SnDesc* __restrict__ sn,
                                                                const int32_t* __restrict__ relpos,
                                                                const WaveEntry* __restrict__ ents,
                                                                const TileDesc* __restrict__ tasks,
                                                                double* __restrict__ L) {
    // (the ring of four chunks, then the strips' ring: 4 x (row strip, column strip) of 16 rows x 8 k -- 80 KiB, two
    // workgroups fill a compute unit's LDS exactly)
    __shared__ __attribute__((aligned(16))) double S[kDStrip + kDSlots * 256];
    {
        int agpr_hint = 0;
        asm volatile("; accumulators in AGPRs %0" ::"a"(agpr_hint));
    }
    const TileDesc td = tasks[blockIdx.x];
    if (td.wp >= td.sp) return;
    // (two bodies, chosen per task: the tasks without strips -- nearly all -- run the loop they always ran)
    if (td.part & kDenseStripTask) chol_dense_task<true>(S, td, sn, relpos, ents, L);
    else chol_dense_task<false>(S, td, sn, relpos, ents, L);
}

void launch_chol_dense(const DevicePattern& P, int first, int count, double* L, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_chol_dense, dim3(count), dim3(kDenseThreads), 0, stream, P.csn, P.relpos, P.big_entries,
                       P.big_tasks + first, L);
}

#ifdef PARSY_DENSESTAMPS
extern "C" void parsy_debug_densephase(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_densephase), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_densephase), z, sizeof(z));
    }
}
#endif

#ifdef PARSY_BIGSTAMPS
extern "C" void parsy_debug_bigstamp_cfg(int grid, int block) {
    const int v[2] = {grid, block};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bigstamp_cfg), v, sizeof(v));
}
extern "C" void parsy_debug_bigtrace(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bigtrace), sizeof(unsigned long long) * 2 * 8 * 1024);
}
#endif

#ifdef PARSY_STAMPS
extern "C" void parsy_debug_stamps(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32);
}
extern "C" void parsy_debug_probe(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), sizeof(unsigned long long) * 8);
}
extern "C" void parsy_debug_trace(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 16 * 512);
}
extern "C" void parsy_debug_tilephase(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tilephase), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tilephase), z, sizeof(z));
    }
}
#endif

void launch_chol_big(const DevicePattern& P, int first, int count, double* L, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_chol_big, dim3(count), dim3(kBigThreads), 0, stream, P.csn, P.relpos, P.big_entries,
                       P.big_tasks + first, L);
}

void launch_chol_tiles(const DevicePattern& P, int first, int count, double* L, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_chol_tiles, dim3(count), dim3(kThreads), 0, stream, P.csn, P.relpos, P.wave_entries,
                       P.wave_ptr, P.split_ranges, P.tile_scratch, P.tiles + first, L);
}

// Resident workgroups of the chain kernel per CU as the runtime sees them (registers, LDS); 0 on error.
int chain_workgroups_per_cu() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_chol_chain, kThreads, 0) != hipSuccess) return 0;
    return nb;
}

void launch_chol_chain(const DevicePattern& P, int first, int count, int ticket, int epoch, bool rows, double* L,
                       hipStream_t stream) {
    if (count <= 0) return;
    if (rows)
        hipLaunchKernelGGL(k_chol_chain_rows, dim3(count), dim3(kThreads), 0, stream, P.csn, P.relpos, P.wave_entries,
                           P.wave_ptr, P.split_ranges, P.tile_scratch, P.tiles + first, L, P.info, P.tflags,
                           P.n_tflags, P.tickets + ticket, epoch);
    else
        hipLaunchKernelGGL(k_chol_chain, dim3(count), dim3(kThreads), 0, stream, P.csn, P.relpos, P.wave_entries,
                           P.wave_ptr, P.split_ranges, P.tile_scratch, P.tiles + first, L, P.info, P.tflags,
                           P.n_tflags, P.tickets + ticket, epoch);
}

}  // namespace parsy
