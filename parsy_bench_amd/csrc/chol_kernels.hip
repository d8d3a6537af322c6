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

#include <climits>

#include "kernels.hpp"

namespace parsy {

typedef double double4_t __attribute__((ext_vector_type(4)));

static constexpr int kThreads = 256;
static constexpr int kLdSub = kSub + 1;   // padded leading dimension of a wave's sub-tile in LDS
static constexpr int kLdDiag = kTile + 1; // padded leading dimension of a diagonal block in LDS

#ifdef PARSY_STAMPS
// diagnostic build only: phase stamps (100 MHz wall clock) of the last PANEL workgroup 0
// and the last potrf-ing tile workgroup; read back with parsy_debug_stamps().
__device__ unsigned long long g_stamps[32];
#define STAMP(i) do { if (threadIdx.x == 0) g_stamps[i] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
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
                                                         double* __restrict__ L,
                                                         int* __restrict__ info, int stage_cap) {
    // stage_cap (<= kSmallStage, host-chosen per launch): doubles per staged block
    extern __shared__ __attribute__((aligned(16))) double P[];  // panel, then 2 stages, then 2 index rings
    const int tid = threadIdx.x;
    const SnDesc D = sn[list[blockIdx.x]];
    const int r = D.r, w = D.w, total = r * w;
    double* __restrict__ G = L + D.px;
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
            for (int e = tid; e < m * n1; e += kThreads) {
                const int j = e / m, i = e - j * m;
                if (i >= j) {
                    double acc = 0.0;
                    for (int kk = 0; kk < CU.K; ++kk)
                        acc = fma(src[i + (int64_t)kk * CU.ld], src[j + (int64_t)kk * CU.ld], acc);
                    P[relg[j] * r + relg[i]] -= acc;
                }
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
        const int pairs = m * n1;
        for (int e = tid; e < pairs; e += kThreads) {
            const int j = e / m, i = e - j * m;
            if (i >= j) {
                double acc = 0.0;
                for (int kk = 0; kk < kc; ++kk) acc = fma(S[kk * m + i], S[kk * m + j], acc);
                const int ri = i < kSmallRelCap ? rel[i] : relg[i];
                const int rj = j < kSmallRelCap ? rel[j] : relg[j];
                P[rj * r + ri] -= acc;
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

    // right-looking POTRF on the w diagonal rows with the r-w rows below carried
    // along (= POTRF followed by TRSM 'R','L','T','N', reference :204,:218)
    for (int j = 0; j < w; ++j) {
        const double d = P[j * r + j];
        const double s = sqrt(d);
        const double inv = 1.0 / s;
        if (tid == 0 && !(d > 0.0)) atomicMin(info, D.c0 + j + 1);
        __syncthreads();
        for (int i = j + tid; i < r; i += kThreads) P[j * r + i] = (i == j) ? s : P[j * r + i] * inv;
        __syncthreads();
        const int nc = w - j - 1, nr = r - j - 1;
        for (int e = tid; e < nc * nr; e += kThreads) {
            const int cc = e / nr;
            const int c = j + 1 + cc, i = j + 1 + (e - cc * nr);
            if (i >= c) P[c * r + i] = fma(-P[j * r + i], P[j * r + c], P[c * r + i]);
        }
        __syncthreads();
    }

    for (int e = tid; e < total; e += kThreads) G[e] = P[e];
}

void launch_chol_small(const DevicePattern& P, int first, int count, int lds_bytes, int stage_cap, double* L,
                       hipStream_t stream) {
    if (count <= 0) return;
    // panel (padded to 16 B) + two staged blocks + four index-ring slots
    stage_cap = min(max(stage_cap, 2), kSmallStage);
    const size_t lds = (size_t)((lds_bytes + 15) & ~15) + 2 * (size_t)stage_cap * sizeof(double) +
                       4 * kSmallRelCap * sizeof(int32_t);
    hipLaunchKernelGGL(k_chol_small, dim3(count), dim3(kThreads), lds, stream, P.sn,
                       P.upd, P.relpos, P.small_list + first, L, P.info, stage_cap);
}

// ---------------------------------------------------------------------------
// TILES / INNER: one workgroup per 64x64 tile of a panel, one wave per 32x32
// sub-tile.  The sub-tile lives in LDS (each wave owns its own, so the update
// loop needs no barrier and the summation order is fixed: update order, then
// k).  The dense product of an update is formed in descendant coordinates with
// v_mfma_f64_16x16x4_f64 straight from the descendant's column-major panel
// (rows of a 16-row fragment are contiguous: 128-B segments per k), then
// scatter-subtracted through the relative indices.
// ---------------------------------------------------------------------------
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

// ---------------------------------------------------------------------------
// POTRF of a 64x64 diagonal block by one 256-thread workgroup, register-blocked: thread
// (ti, tj) keeps the 4x4 block rows 4ti.., cols 4tj.. in registers.  Four columns per step:
//   (1) the owner of the 4x4 diagonal micro-block broadcasts it through LDS,
//   (2) every thread of that block column factors the micro-block itself (4 pivots in
//       registers) and solves its own 4x4 block against it, then publishes the 4 new columns,
//   (3) everybody applies the rank-4 update to its block.
// Two barriers per 4 columns.  Entries outside the block (nb < 64) must be an identity so the
// loop is uniform; strictly-upper entries pick up garbage that is never stored.  `scr` is
// 16 + 64*5 doubles of LDS.  On return a[][] holds the factor; `bad` receives (1-based) the
// first column whose pivot was not positive.  Pivots use rsqrt (1/sqrt(d) directly: the
// divide-after-sqrt chain is the critical path of every block step).
// ---------------------------------------------------------------------------
static constexpr int kPotrfScratch = 16 + kTile * 5;
__device__ __forceinline__ void potrf64_regs(double (&a)[4][4], double* __restrict__ scr, int ti, int tj,
                                             int nb, int& bad) {
    double* __restrict__ bufD = scr;        // 4x4 micro-block, row-major
    double* __restrict__ bufP = scr + 16;   // [64 rows][4 cols], ld 5
    const bool lower = ti >= tj;
    bad = 0;
    for (int tjj = 0; tjj < kTile / 4; ++tjj) {
        if (ti == tjj && tj == tjj) {
#pragma unroll
            for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) bufD[ri * 4 + ci] = a[ri][ci];
        }
        __syncthreads();
        if (tj == tjj && lower) {
            double m[4][4], l[4][4], inv[4];
#pragma unroll
            for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) m[ri][ci] = bufD[ri * 4 + ci];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                double d = m[jj][jj];
#pragma unroll
                for (int k = 0; k < jj; ++k) d = fma(-l[jj][k], l[jj][k], d);
                if (!(d > 0.0) && bad == 0 && 4 * tjj + jj < nb) bad = 4 * tjj + jj + 1;
                inv[jj] = rsqrt(d);
                l[jj][jj] = d * inv[jj];
#pragma unroll
                for (int ii = jj + 1; ii < 4; ++ii) {
                    double v = m[ii][jj];
#pragma unroll
                    for (int k = 0; k < jj; ++k) v = fma(-l[ii][k], l[jj][k], v);
                    l[ii][jj] = v * inv[jj];
                }
            }
            if (ti == tjj) {
#pragma unroll
                for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                    for (int ci = 0; ci <= ri; ++ci) a[ri][ci] = l[ri][ci];
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
            for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) bufP[(4 * ti + ri) * 5 + ci] = a[ri][ci];
        }
        __syncthreads();
        if (lower && tj > tjj) {
            double li[4][4], lc[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    li[q][k] = bufP[(4 * ti + q) * 5 + k];
                    lc[q][k] = bufP[(4 * tj + q) * 5 + k];
                }
#pragma unroll
            for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[ri][ci] = fma(-li[ri][k], lc[ci][k], a[ri][ci]);
        }
    }
}

static constexpr int kKC = 16;        // k extent of one chunk of an update stream
static constexpr int kInFlight = 4;   // chunks in flight per wave (operands prefetched into registers)

// One workgroup per 64x64 tile of a panel, one wave per 32x32 sub-tile, and the four waves run
// their update streams independently (no workgroup barrier until the stream is finished).  A
// wave's stream -- WaveEntry list built on the host: every descendant with rows in the
// sub-tile's row AND column window, in update order -- is cut in 16-wide k chunks.  The MFMA
// operands of a chunk go straight from the descendant's column-major panel into registers
// (v_mfma_f64_16x16x4_f64: lane = (row & 15, k >> 2 group), 16 consecutive rows per k are one
// 128-B segment), kInFlight chunks ahead of the multiply; at the end of a descendant the
// product is scatter-subtracted into the wave's private LDS sub-tile through the relative
// indices (ds_add_f64 without return; the LDS operations of one wave execute in order, so the
// summation order is fixed).
template <bool INNER>
__global__ __launch_bounds__(kThreads, 2) void k_chol_tiles(const SnDesc* __restrict__ sn,
                                                            const int32_t* __restrict__ relpos,
                                                            const WaveEntry* __restrict__ wents,
                                                            const int64_t* __restrict__ wptr,
                                                            const TileDesc* __restrict__ tiles, int jb,
                                                            double* __restrict__ L,
                                                            double* __restrict__ dscratch,
                                                            int* __restrict__ info,
                                                            int* __restrict__ flags, int epoch,
                                                            int fused, int finalize) {
    __shared__ double T[4][kSub * kLdSub];
    __shared__ double colbuf[kPotrfScratch];
    __shared__ double dgbuf[kTile * kLdDiag];  // diagonal block + its 16x16 inverses (TRSM)
    __shared__ int32_t s_ok;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileDesc td = tiles[blockIdx.x];
    const SnDesc D = sn[td.sn];
    const int r = D.r, w = D.w;
    double* __restrict__ G = L + D.px;
    const bool stamp_wg = td.row0 == td.col0 && (INNER ? td.col0 == (jb + 1) * kTile : td.col0 == 0);
    if (stamp_wg) STAMP(8);

    const int wa = wave >> 1, wb = wave & 1;
    const int subrow0 = td.row0 + kSub * wa, subcol0 = td.col0 + kSub * wb;
    const bool wave_on = subrow0 < r && subcol0 < w && subrow0 >= subcol0;
    const int nrows = min(kSub, r - subrow0), ncols = min(kSub, w - subcol0);
    double* __restrict__ Tw = T[wave];

    if (wave_on) {
        // all 16 loads of the lane are issued before the first LDS store
        double tv[kSub * kSub / 64];
#pragma unroll
        for (int q = 0; q < kSub * kSub / 64; ++q) {
            const int e = q * 64 + lane;
            const int cc = e >> 5, rr = e & 31;
            const bool in = rr < nrows && cc < ncols && (subrow0 + rr >= subcol0 + cc);
            tv[q] = in ? G[(int64_t)(subcol0 + cc) * r + subrow0 + rr] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < kSub * kSub / 64; ++q) {
            const int e = q * 64 + lane;
            Tw[(e >> 5) * kLdSub + (e & 31)] = tv[q];
        }
    }

    const int l15 = lane & 15, kq = lane >> 4, l31 = lane & 31;
    const bool diag_sub = subrow0 == subcol0;

    // ---- this wave's update stream
    int64_t le = 0, e_end = 0;  // next entry the loader enters / end of the list
    if (wave_on) {
        if (INNER) {
            e_end = 1;
        } else {
            le = wptr[td.wp + wave];
            e_end = wptr[td.wp + wave + 1];
        }
    }
    if (le < e_end) {
        struct Chunk {
            double a0[4], a1[4], b0[4], b1[4];  // MFMA operands of the four k steps
            int32_t rel;                         // last chunk of an entry: this lane's relative index
            int32_t kend, last, mn;              // wave-uniform: valid k in the chunk (0: padding of the
                                                 // stream), last chunk of its entry, window sizes
        };
        // loader state (wave-uniform except the lane offsets).  Every chunk issues the same 16
        // operand loads (+ the relative index): fragments a narrow entry does not have re-read
        // its last row, k steps past a ragged end re-read column K-1 (masked in the multiply),
        // chunks behind the end of the stream re-read the last one.  A fixed number of loads
        // per chunk keeps the s_waitcnt of the multiply exactly kInFlight-1 chunks behind.
        const double* l_p = G;                   // descendant panel at row lb, column = chunk start
        int l_K = 0, l_k = 0, l_ld = 0, l_mn = 0;
        int l_oA0 = 0, l_oA1 = 0, l_oB0 = 0, l_oB1 = 0, l_orel = 0;
        WaveEntry l_next = {};
        if (!INNER) l_next = wents[le];
        auto loader_enter = [&]() {
            WaveEntry E;
            if (INNER) {
                E.src = D.px + (int64_t)jb * kTile * r;  // block column jb of the same panel, identity map
                E.rel = 0;
                E.ld = r;
                E.K = min(kTile, w - jb * kTile);
                E.ia = subrow0;
                E.ja = subcol0;
                E.mn = nrows | (ncols << 8);
            } else {
                E = l_next;  // fetched one entry ahead
                if (le + 1 < e_end) l_next = wents[le + 1];
            }
            const int mi = E.mn & 255, nj = E.mn >> 8;
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
            ++le;
        };
        bool l_live = true;
        loader_enter();
        auto issue = [&](Chunk& c) {
            const int kend = l_live ? min(kKC, l_K - l_k) : 0;
            c.kend = kend;
            c.mn = l_mn;
            c.last = l_live && (l_k + kKC >= l_K);
            int ko[4];
            if (kend == kKC) {
#pragma unroll
                for (int u = 0; u < 4; ++u) ko[u] = (4 * u + kq) * l_ld;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) ko[u] = min(4 * u + kq, max(kend, 1) - 1) * l_ld;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#ifdef PARSY_ABL_NOLOAD
                c.a0[u] = c.a1[u] = c.b0[u] = c.b1[u] = (double)(ko[u]);
#else
                c.a0[u] = l_p[l_oA0 + ko[u]];
                c.b0[u] = l_p[l_oB0 + ko[u]];
                c.a1[u] = l_p[l_oA1 + ko[u]];
                c.b1[u] = l_p[l_oB1 + ko[u]];
#endif
            }
            if (!INNER) c.rel = relpos[l_orel];
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
            const int mi = c.mn & 255, nj = c.mn >> 8;
            const bool two_r = mi > 16, two_c = nj > 16;
            const bool up = two_c && !diag_sub;  // rows 0..15 x columns 16..31: strictly upper in a diagonal sub-tile
            // straight-line on purpose (no early exit on a ragged or padding chunk): k past the
            // end contributes 0 through the A operand, B holds finite panel values
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
#ifdef PARSY_ABL_NOSCATTER
            if (false) {
#else
            if (c.last) {
#endif
                // scatter-subtract through the relative indices (C/D layout of
                // v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg).  Lanes 0..31
                // hold the sub-tile row of descendant row `lane` of the row window, lanes 32..63 the
                // sub-tile column of descendant row `lane - 32` of the column window (-1: outside).
                const int relv = (l31 < (lane < 32 ? mi : nj)) ? (INNER ? l31 : c.rel - (lane < 32 ? subrow0 : subcol0)) : -1;
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

        // rounds of kInFlight chunks, one back edge: the stream is done when the last chunk of
        // a round is padding (kend == 0; padding multiplies by zero and scatters nothing)
        Chunk q[kInFlight];
#pragma unroll
        for (int i = 0; i < kInFlight; ++i) issue(q[i]);
        bool more = true;
        while (more) {
#pragma unroll
            for (int sidx = 0; sidx < kInFlight; ++sidx) {
                consume(q[sidx]);
                if (sidx == kInFlight - 1) more = q[sidx].kend != 0;
                issue(q[sidx]);
            }
        }
    }
    __syncthreads();

    if (stamp_wg) STAMP(9);
    // The block column that has just received its last update is finished here: its diagonal
    // tile is factored on the spot and parked in a scratch slot (FIXUP copies it into the
    // panel); in a fused launch the tiles below it wait for that block and do their TRSM
    // straight out of LDS, otherwise PANEL does it in the next launch.
    const bool col_final = finalize && (INNER ? td.col0 == (jb + 1) * kTile : td.col0 == 0);
    const bool diag_tile = td.row0 == td.col0;
    const bool trsm_here = fused && col_final && !diag_tile;
    auto write_back = [&](int min_row) {  // min_row: first row of the TILE that is written
        if (wave_on) {
            for (int e = lane; e < kSub * kSub; e += 64) {
                const int cc = e >> 5, rr = e & 31;
                if (rr < nrows && cc < ncols && (subrow0 + rr >= subcol0 + cc) && kSub * wa + rr >= min_row)
                    G[(int64_t)(subcol0 + cc) * r + subrow0 + rr] = Tw[cc * kLdSub + rr];
            }
        }
    };
    if (!trsm_here) write_back(0);
    if (!col_final) return;
    const int nb = min(kTile, w - td.col0);
    int trsm_min_row = 0;
    const int slot_id = D.dslot + td.col0 / kTile;
    double* __restrict__ slot = dscratch + (int64_t)slot_id * (kTile * kTile);

    if (diag_tile) {
        __syncthreads();
        STAMP(10);
        const int ti = tid & 15, tj = tid >> 4;
        double a[4][4];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int i = 4 * ti + ri, c = 4 * tj + ci;
                double v = (i == c) ? 1.0 : 0.0;
                if (c < nb && i < nb && i >= c) v = T[(i >> 5) * 2 + (c >> 5)][(c & 31) * kLdSub + (i & 31)];
                a[ri][ci] = v;
            }
        int bad;
        STAMP(11);
        potrf64_regs(a, colbuf, ti, tj, nb, bad);
        STAMP(12);
        if (bad) atomicMin(info, D.c0 + td.col0 + bad);  // only threads that saw a bad pivot
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int i = 4 * ti + ri, c = 4 * tj + ci;
                slot[c * kTile + i] = (c < nb && i < nb && i >= c) ? a[ri][ci] : 0.0;
            }
        if (fused) {
            // publish (agent-scope release, cdna_hip_programming.md Guideline 16): every storing
            // wave drains its stores, the workgroup meets, one lane releases and raises the flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&flags[slot_id], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        STAMP(13);
        // a block column narrower than the tile leaves rows of the panel below the diagonal
        // block inside this very tile: solve them here (PANEL does it when not fused)
        if (!(fused && nb < kTile && td.row0 + nb < r)) return;
        trsm_min_row = nb;
        __syncthreads();  // the parked block (global) is visible to the whole workgroup
    } else if (!trsm_here) {
        return;
    } else if (tid == 0) {
        if (td.row0 == td.col0 + kTile) STAMP(24);
        // ---- wait for the parked diagonal block (bounded)
        const unsigned long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(&flags[slot_id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            if (wall_clock64() - t0 > 20000000ull) {  // 0.2 s at 100 MHz: give up, report
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ok = ok;
    }
    if (!diag_tile) {
        __syncthreads();
        if (!s_ok) {
            if (tid == 0) atomicMin(info, -1);  // status < 0: a fused wait timed out
            write_back(0);
            return;
        }
    }
    if (td.row0 == td.col0 + kTile) STAMP(25);
    // ---- X := B inv(Ljj') on the rows of the LDS tile
    double* __restrict__ Dg = dgbuf;
    double* __restrict__ invd = colbuf;
    {
        double dtmp[kTile * kTile / kThreads];
#pragma unroll
        for (int q = 0; q < kTile * kTile / kThreads; ++q) dtmp[q] = slot[q * kThreads + tid];
#pragma unroll
        for (int q = 0; q < kTile * kTile / kThreads; ++q) {
            const int e = q * kThreads + tid;
            Dg[(e >> 6) * kLdDiag + (e & 63)] = dtmp[q];
        }
        if (tid < kTile) invd[tid] = (tid < nb) ? 1.0 / slot[tid * kTile + tid] : 1.0;
    }
    __syncthreads();
    if (td.row0 == td.col0 + kTile) STAMP(26);
    // inverses of the four 16x16 diagonal sub-blocks of Ljj, one column per thread, written
    // transposed into the (unused) strict upper triangle of the same sub-block:
    // Dg[(16b+r)*ld + 16b+c] = inv(L_bb)[r][c] for r > c.  The TRSM below is then all products
    // (what a blocked dtrsm does): X_b = (B_b - sum_{p<b} X_p L_bp') inv(L_bb)'.
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
        __builtin_amdgcn_s_waitcnt(0);  // all reads of the sub-block precede the in-place writes
#pragma unroll
        for (int rr = 1; rr < 16; ++rr)
            if (rr > c) Dg[(b16 + rr) * kLdDiag + b16 + c] = y[rr];
    }
    __syncthreads();
    {
        // each wave owns 16 rows of the tile for the whole solve: no barrier between blocks
        const int rbase = 16 * wave;
        auto TT = [&](int i, int c) -> double& {
            return T[(i >> 5) * 2 + (c >> 5)][(c & 31) * kLdSub + (i & 31)];
        };
        for (int b16 = 0; b16 < nb; b16 += 16) {
            double4_t acc = {0, 0, 0, 0};
            for (int p16 = 0; p16 < b16; p16 += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = p16 + kq + 4 * u;
                    const double av = TT(rbase + l15, k);                  // X[row][k]
                    const double bv = Dg[k * kLdDiag + b16 + l15];         // L[b16 + j][k]
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) TT(rbase + kq + 4 * v, b16 + l15) -= acc[v];
            double4_t acc2 = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kq + 4 * u, j = l15;
                const double av = TT(rbase + l15, b16 + k);                // R[row][k]
                double wv = 0.0;                                            // inv(L_bb)'[k][j] = inv(L_bb)[j][k]
                if (j > k) wv = Dg[(b16 + j) * kLdDiag + b16 + k];
                else if (j == k) wv = invd[b16 + k];
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, wv, acc2, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) TT(rbase + kq + 4 * v, b16 + l15) = acc2[v];
        }
    }
    __syncthreads();
    if (td.row0 == td.col0 + kTile) STAMP(27);
    write_back(trsm_min_row);
    if (td.row0 == td.col0 + kTile) STAMP(28);
}

#ifdef PARSY_STAMPS
extern "C" void parsy_debug_stamps(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32);
}
#endif

void launch_chol_tiles(const DevicePattern& P, int first, int count, bool inner, int jb, int fused,
                       int finalize, int epoch, double* L, hipStream_t stream) {
    if (count <= 0) return;
    if (inner)
        hipLaunchKernelGGL(k_chol_tiles<true>, dim3(count), dim3(kThreads), 0, stream, P.sn, P.relpos,
                           P.wave_entries, P.wave_ptr, P.tiles + first, jb, L, P.dscratch, P.info, P.flags, epoch,
                           fused, finalize);
    else
        hipLaunchKernelGGL(k_chol_tiles<false>, dim3(count), dim3(kThreads), 0, stream, P.sn, P.relpos,
                           P.wave_entries, P.wave_ptr, P.tiles + first, jb, L, P.dscratch, P.info, P.flags, epoch,
                           fused, finalize);
}

// ---------------------------------------------------------------------------
// PANEL: TRSM of one 128-row chunk below diagonal block jb, staged in LDS.  The
// factored diagonal block is read from its scratch slot, where the tile kernel parked
// it (nobody rewrites the slot during this launch).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_chol_panel(const SnDesc* __restrict__ sn,
                                                         const PanelDesc* __restrict__ pds,
                                                         double* __restrict__ L,
                                                         const double* __restrict__ dscratch) {
    __shared__ double Dg[kTile * kLdDiag];
    __shared__ double invd[kTile];
    __shared__ double Bs[kTile][kPanelRows + 1];
    const int tid = threadIdx.x;
    const PanelDesc pd = pds[blockIdx.x];
    const SnDesc D = sn[pd.sn];
    const int r = D.r, cb = pd.jb * kTile, wbk = min(kTile, D.w - cb);
    double* __restrict__ G = L + D.px;
    if (blockIdx.x == 0) STAMP(0);

    // all loads of the chunk are issued before the first LDS store (one latency, not 32)
    const double* __restrict__ slot = dscratch + (int64_t)(D.dslot + pd.jb) * (kTile * kTile);
    {
        constexpr int kPer = kTile * kPanelRows / kThreads;  // 32
        double tmp[kPer], dtmp[kTile * kTile / kThreads];
        const int rr = tid & (kPanelRows - 1), chalf = tid >> 7;
        const int row = pd.row0 + rr;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int c = 2 * q + chalf;
            tmp[q] = (c < wbk && row < r) ? G[(int64_t)(cb + c) * r + row] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < kTile * kTile / kThreads; ++q) dtmp[q] = slot[q * kThreads + tid];
#pragma unroll
        for (int q = 0; q < kPer; ++q) Bs[2 * q + chalf][rr] = tmp[q];
#pragma unroll
        for (int q = 0; q < kTile * kTile / kThreads; ++q) {
            const int e = q * kThreads + tid;
            Dg[(e >> 6) * kLdDiag + (e & 63)] = dtmp[q];
        }
    }
    if (tid < kTile) invd[tid] = (tid < wbk) ? 1.0 / slot[tid * kTile + tid] : 1.0;
    __syncthreads();
    if (blockIdx.x == 0) STAMP(1);

    // X := B * inv(Ljj') by forward substitution, 16 columns at a time: the low half of
    // the workgroup solves the 16x16 triangle of its row, then both halves apply the
    // rank-16 update to the remaining columns (odd / even columns).
    const int row = tid & (kPanelRows - 1), half = tid >> 7;
    for (int c0 = 0; c0 < wbk; c0 += 16) {
        if (half == 0) {
            double xb[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) xb[j] = Bs[c0 + j][row];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double acc = xb[j];
#pragma unroll
                for (int i = 0; i < j; ++i) acc = fma(-xb[i], Dg[(c0 + i) * kLdDiag + c0 + j], acc);
                xb[j] = acc * invd[c0 + j];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) Bs[c0 + j][row] = xb[j];
        }
        __syncthreads();
        if (c0 + 16 < wbk) {
            double xk[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) xk[k] = Bs[c0 + k][row];
            for (int j = c0 + 16 + half; j < wbk; j += 2) {
                double acc = Bs[j][row];
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = fma(-xk[k], Dg[(c0 + k) * kLdDiag + j], acc);
                Bs[j][row] = acc;
            }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) STAMP(2);
    for (int e = tid; e < kTile * kPanelRows; e += kThreads) {
        const int c = e >> 7, rr = e & (kPanelRows - 1);
        const int grow = pd.row0 + rr;
        if (c < wbk && grow < r) G[(int64_t)(cb + c) * r + grow] = Bs[c][rr];
    }
    if (blockIdx.x == 0) STAMP(3);
}

void launch_chol_panel(const DevicePattern& P, int first, int count, double* L, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_chol_panel, dim3(count), dim3(kThreads), 0, stream, P.sn, P.panels + first, L,
                       P.dscratch);
}

// FIXUP: copy parked diagonal blocks into their panels (lower triangle only).
__global__ __launch_bounds__(kThreads) void k_chol_fixup(const SnDesc* __restrict__ sn,
                                                         const int32_t* __restrict__ list,
                                                         double* __restrict__ L,
                                                         const double* __restrict__ dscratch) {
    const SnDesc D = sn[list[blockIdx.x]];
    const int r = D.r, nbc = (D.w + kTile - 1) / kTile;
    double* __restrict__ G = L + D.px;
    for (int jb = blockIdx.y; jb < nbc; jb += gridDim.y) {
        const int cb = jb * kTile, wbk = min(kTile, D.w - cb);
        const double* __restrict__ slot = dscratch + (int64_t)(D.dslot + jb) * (kTile * kTile);
        for (int e = threadIdx.x; e < kTile * kTile; e += kThreads) {
            const int c = e >> 6, i = e & 63;
            if (c < wbk && i < wbk && i >= c) G[(int64_t)(cb + c) * r + cb + i] = slot[e];
        }
    }
}

void launch_chol_fixup(const DevicePattern& P, int first, int count, double* L, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_chol_fixup, dim3(count, 32), dim3(kThreads), 0, stream, P.sn,
                       P.fix_list + first, L, P.dscratch);
}

}  // namespace parsy
