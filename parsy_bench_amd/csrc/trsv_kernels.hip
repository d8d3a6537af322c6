// HIP kernels of the level-scheduled BCSC forward solve L X = B for gfx950.
//
// Numeric contract (reference triangularSolve/Triangular_BCSC.h:139-157): per
// supernode, dense forward solve of x[cols] with the diagonal block (non-unit
// diagonal, division as triangularSolve/BLAS.h:8), tmp = L21 * x[cols], then
// x[Li[l]] -= tmp[k] with an atomic (the reference uses `omp atomic`, so its
// rounding order is schedule-dependent too).  X is n x nrhs column-major.
// HBM-bound: every stored L value is read once per pass of kRhs right-hand sides.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"

namespace parsy {

static constexpr int kThreads = 256;
static constexpr unsigned long long kSolveSpinTicks = 200000000ull;  // 2 s of the 100 MHz wall clock (as the factorization's waits)
static constexpr int kLdDiag = kTile + 1;
static constexpr int kRhs = 8;  // right-hand sides carried per pass over a panel
// Passes over the right-hand sides are independent of each other: up to kPassLanes of them run side by
// side (workgroups of their own; the chain launches keep one set of flags per lane), the rest follow in
// rounds.  A block of 64 right-hand sides then costs about one pass of latency instead of eight.

// Forward solve of the staged block xs[c][q] (c < w <= 64, q < nq) with the lower-triangular
// block Dg (column-major, ld kLdDiag; entries outside w x w must be an identity).  Blocked by
// 16: the four 16x16 diagonal sub-blocks are inverted once (one column per thread, written
// transposed into the unused strict upper triangle of the sub-block), then per sub-block
//   y_b = inv(L_bb) x_b   and   x_rest -= L(rest, b) y_b
// -- 2 barriers per 16 columns instead of 2 per column.  All threads participate; ends
// synchronised.  (triangularSolve/BLAS.h:8 divides by the diagonal; so does the inversion.)
__device__ __forceinline__ void block_solve_apply16(const double* Dg, const double* invd,
                                                    double (*xs)[kRhs], int w, int nq, int tid);

__device__ __forceinline__ void block_invert16(double* Dg, double* invd, int w, int tid);
__device__ __forceinline__ void block_solve_inv16(double* Dg, double* invd, double (*xs)[kRhs], int w,
                                                  int nq, int tid) {
    block_invert16(Dg, invd, w, tid);
    block_solve_apply16(Dg, invd, xs, w, nq, tid);
}
// (the inversions alone: they do not depend on x -- k_solve_one does them before it waits for its x; ends synchronised)
// (round 5: one wave per 16 x 16 sub-block, one entry per lane, by halves -- inv [A 0; B C] = [inv A, 0; -inv(C) B inv(A),
// inv C] --, as chol_kernels.hip's invert_diag_blocks: the form before gave a column to a thread and walked its 16 rows one
// after the other (3.7 us).  Scratch: a 16 x 16 block of Dg above the block diagonal -- zeros that nothing reads.  Needs the
// whole workgroup of kThreads = 256: wave b = sub-block b.)
__device__ __forceinline__ void block_invert16(double* Dg, double* invd, int w, int tid) {
    if (tid < kTile) invd[tid] = 1.0 / Dg[tid * kLdDiag + tid];
    __syncthreads();
    const int lane = tid & 63;
    const int b = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (16 * b < w) {
        double* __restrict__ B0 = Dg + (16 * b) * kLdDiag + 16 * b;      // L[i][c] = B0[c * ld + i]; W[r][c] (r > c) -> B0[r * ld + c]
        const double* __restrict__ dv = invd + 16 * b;
        double* __restrict__ Sb = Dg + (16 * (b < 3 ? b + 1 : 3)) * kLdDiag + (b < 3 ? 0 : 16);
        auto SC = [&](int e) -> double& { return Sb[(e >> 4) * kLdDiag + (e & 15)]; };
        {   // 4 x 4 diagonal blocks: lane = (d, i', k'): column k' of inv(L_dd) by substitution, entry i'
            const int d4 = 4 * (lane >> 4), ip = (lane >> 2) & 3, kp = lane & 3;
            const double* __restrict__ T = B0 + d4 * kLdDiag + d4;
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
                lr[m] = B0[(e8 + m) * kLdDiag + e8 + 4 + i];
                wa[m] = B0[(e8 + m) * kLdDiag + e8 + j];
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
                wc[m] = B0[(e8 + 4 + i) * kLdDiag + e8 + 4 + m];
                tt[m] = SC((lane & 16) + 4 * m + j);
            }
            double wv = 0.0;
#pragma unroll
            for (int m = 0; m < 4; ++m) wv = fma(i > m ? wc[m] : (i == m ? di : 0.0), tt[m], wv);
            if (lane < 32) B0[(e8 + 4 + i) * kLdDiag + e8 + j] = -wv;
        }
        __builtin_amdgcn_wave_barrier();
        {   // the 8 x 8 block below the diagonal: lane = (i, j)
            const int i = lane >> 3, j = lane & 7;
            double lr[8], wa[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                lr[m] = B0[m * kLdDiag + 8 + i];
                wa[m] = B0[m * kLdDiag + j];
            }
            const double dj = dv[j], di = dv[8 + i];
            double t = 0.0;
#pragma unroll
            for (int m = 0; m < 8; ++m) t = fma(lr[m], m > j ? wa[m] : (m == j ? dj : 0.0), t);
            SC(lane) = t;
            __builtin_amdgcn_wave_barrier();
            double wc[8], tt[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                wc[m] = B0[(8 + i) * kLdDiag + 8 + m];
                tt[m] = SC(8 * m + j);
            }
            double wv = 0.0;
#pragma unroll
            for (int m = 0; m < 8; ++m) wv = fma(i > m ? wc[m] : (i == m ? di : 0.0), tt[m], wv);
            B0[(8 + i) * kLdDiag + j] = -wv;
        }
    }
    __syncthreads();
}

// The substitution itself, given the inverted sub-blocks (re-used for every pass of
// right-hand sides over the same diagonal block).
__device__ __forceinline__ void block_solve_apply16(const double* Dg, const double* invd,
                                                    double (*xs)[kRhs], int w, int nq, int tid) {
    for (int b16 = 0; b16 < w; b16 += 16) {
        // y = inv(L_bb) x_b : thread (i, q), i < 16
        double yv = 0.0;
        const int i = tid & 15, q = tid >> 4;
        const bool act = q < nq;
        if (act) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                double lv = 0.0;
                if (k < i) lv = Dg[(b16 + i) * kLdDiag + b16 + k];  // inv(L_bb)[i][k], stored transposed
                else if (k == i) lv = invd[b16 + i];
                yv = fma(lv, xs[b16 + k][q], yv);
            }
        }
        __syncthreads();
        if (act) xs[b16 + i][q] = yv;
        __syncthreads();
        // rows below the sub-block: x[r][q] -= sum_k L[r][b16+k] y[k][q]
        const int rem = w - b16 - 16;
        for (int e = tid; e < rem * nq; e += kThreads) {
            const int qq = e / rem, rr = b16 + 16 + (e - qq * rem);
            double acc = xs[rr][qq];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = fma(-Dg[(b16 + k) * kLdDiag + rr], xs[b16 + k][qq], acc);
            xs[rr][qq] = acc;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane),
                            __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}

// An entry of x that earlier supernodes of the same launch may have updated with atomics (subtree launches: the
// same workgroup, ordered by a fence + barrier): read at the L2, where the atomics were performed.
__device__ __forceinline__ double ld_x(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// SOLVE_SMALL: one workgroup per supernode of width <= 64 -- or, ranges != null (subtree launch), per subtree
// of such supernodes: list[ranges[2b] .. ranges[2b+1]) one after the other (index order: descendants first;
// what they subtract from the x of the later ones is complete when those start).
__global__ __launch_bounds__(kThreads) void k_solve_small(const SnDesc* __restrict__ sn,
                                                          const int32_t* __restrict__ list,
                                                          const int32_t* __restrict__ ranges,
                                                          const int32_t* __restrict__ rows,
                                                          const double* __restrict__ L,
                                                          double* __restrict__ x, int nrhs, int ldx) {
    __shared__ double Dg[kTile * kLdDiag];
    __shared__ double invd[kTile];
    __shared__ double xs[kTile][kRhs];
    const int tid = threadIdx.x;
    const int q_begin = ranges ? ranges[2 * blockIdx.x] : (int)blockIdx.x;
    const int q_end = ranges ? ranges[2 * blockIdx.x + 1] : q_begin + 1;
  for (int qsn = q_begin; qsn < q_end; ++qsn) {
    if (qsn > q_begin) {
        // this thread's atomics are performed (they execute at the device's coherence point: waiting for their
        // completion is all it takes -- no L2 write-back as an agent-scope fence would do) ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // ... and so are everybody's; Dg / xs are free again
    }
    const SnDesc D = sn[list[qsn]];
    const int r = D.r, w = D.w;
    const double* __restrict__ G = L + D.px;
    const int32_t* __restrict__ ri = rows + D.pi;

    const int wpad = (w + 15) & ~15;
    for (int e = tid; e < wpad * wpad; e += kThreads) {
        const int c = e / wpad, i = e - c * wpad;
        double v = (i == c) ? 1.0 : 0.0;
        if (i >= c && i < w && c < w) v = G[(int64_t)c * r + i];
        Dg[c * kLdDiag + i] = v;
    }
    bool first = true;
    for (int q0 = blockIdx.y * kRhs; q0 < nrhs; q0 += gridDim.y * kRhs) {
        const int nq = min(kRhs, nrhs - q0);
        __syncthreads();
        for (int e = tid; e < wpad * nq; e += kThreads) {
            const int q = e / wpad, c = e - q * wpad;
            xs[c][q] = (c < w) ? ld_x(&x[(int64_t)(q0 + q) * ldx + D.c0 + c]) : 0.0;
        }
        __syncthreads();
        const bool was_first = first;
        first = false;
        if (was_first) block_solve_inv16(Dg, invd, xs, w, nq, tid);
        else block_solve_apply16(Dg, invd, xs, w, nq, tid);
        for (int e = tid; e < w * nq; e += kThreads) {
            const int q = e / w, c = e - q * w;
            x[(int64_t)(q0 + q) * ldx + D.c0 + c] = xs[c][q];
        }
        for (int k = w + tid; k < r; k += kThreads) {
            double acc[kRhs];
#pragma unroll
            for (int q = 0; q < kRhs; ++q) acc[q] = 0.0;
            for (int c = 0; c < w; ++c) {
                const double lv = G[(int64_t)c * r + k];
#pragma unroll
                for (int q = 0; q < kRhs; ++q) acc[q] = fma(lv, xs[c][q], acc[q]);
            }
            const int row = ri[k];
#pragma unroll
            for (int q = 0; q < kRhs; ++q)
                if (q < nq) atomicAdd(&x[(int64_t)(q0 + q) * ldx + row], -acc[q]);
        }
    }
  }
}

// SOLVE_SMALL for supernodes of width <= 16 -- most supernodes of a nested-dissection ordering -- : one WAVE per
// supernode (or per subtree of them: ranges != null, as k_solve_small), no LDS and no barrier.  Lane i holds row i
// of the diagonal block; the substitution goes column by column with lane broadcasts (v_readlane: x_c becomes
// wave-uniform); the rows below the block are one row per lane, their first 64 loaded before the substitution
// starts, and are subtracted from x with atomics.  All loads of a supernode are in flight together: its latency
// is one memory round trip instead of the three of the workgroup kernel.
template <int kTinyW>   // width class of the launch (kTinyWidth; the backward kernel also has kTinyWidth2)
__global__ __launch_bounds__(64) void k_solve_tiny(const SnDesc* __restrict__ sn, const int32_t* __restrict__ list,
                                                   const int32_t* __restrict__ ranges,
                                                   const int32_t* __restrict__ rows, const double* __restrict__ L,
                                                   double* __restrict__ x, int nrhs, int ldx) {
    const int lane = threadIdx.x;
    const int q_begin = ranges ? ranges[2 * blockIdx.x] : (int)blockIdx.x;
    const int q_end = ranges ? ranges[2 * blockIdx.x + 1] : q_begin + 1;
    for (int qsn = q_begin; qsn < q_end; ++qsn) {
        // (subtree launch: the atomics of the supernodes before are performed before this one reads x)
        if (qsn > q_begin) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const SnDesc D = sn[list[qsn]];
        const int r = D.r, w = D.w;
        const double* __restrict__ G = L + D.px;
        const int32_t* __restrict__ ri = rows + D.pi;
        // every load is unconditional: lanes / columns past the block read its last row / column (never used)
        const int il = min(lane, w - 1);
        const int kb = min(w + lane, r - 1);   // first chunk of the rows below the block
        double l[kTinyW], lb[kTinyW];
#pragma unroll
        for (int c = 0; c < kTinyW; ++c) {
            const double* __restrict__ col = G + (int64_t)min(c, w - 1) * r;
            l[c] = col[il];
            lb[c] = col[kb];
        }
        const int row_b = ri[kb];
        double diag = 1.0;
#pragma unroll
        for (int c = 0; c < kTinyW; ++c) diag = (lane == c && c < w) ? l[c] : diag;
        const double rdiag = 1.0 / diag;   // (triangularSolve/BLAS.h:8 divides by the diagonal)
        for (int q = blockIdx.y; q < nrhs; q += gridDim.y) {
            double* __restrict__ xq = x + (int64_t)q * ldx;
            double xi = lane < w ? ld_x(&xq[D.c0 + lane]) : 0.0;
            double xc[kTinyW];   // the solved block, wave-uniform
#pragma unroll
            for (int c = 0; c < kTinyW; ++c) {
                xc[c] = 0.0;
                if (c < w) {
                    xc[c] = readlane_f64(xi * rdiag, c);
                    xi = (lane > c) ? fma(-l[c], xc[c], xi) : xi;
                }
            }
            double xfin = 0.0;
#pragma unroll
            for (int c = 0; c < kTinyW; ++c) xfin = (lane == c) ? xc[c] : xfin;
            if (lane < w) xq[D.c0 + lane] = xfin;
            // rows below the block: the prefetched chunk, then the rest
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < kTinyW; ++c) acc = fma(lb[c], xc[c], acc);
            if (w + lane < r) atomicAdd(&xq[row_b], -acc);
            for (int k = w + 64 + lane; k < r; k += 64) {
                double a2 = 0.0;
#pragma unroll
                for (int c = 0; c < kTinyW; ++c) a2 = fma(G[(int64_t)min(c, w - 1) * r + k], xc[c], a2);
                atomicAdd(&xq[ri[k]], -a2);
            }
        }
    }
}

// SOLVE_SMALL for many right-hand sides (nrhs >= 16): up to 64 of them per pass over the panel, so that L is
// read once per 64 right-hand sides instead of once per 8, and the products run on the matrix cores
// (v_mfma_f64_16x16x4_f64).  Wave q owns right-hand sides 16q..16q+15 of the pass:
//   (A) diagonal solve, blocked by 16 with the inverted 16x16 sub-blocks (as block_solve_inv16), every product
//       an MFMA with both operands in LDS; a wave only touches its own 16 columns of xs: no barrier inside;
//   (B) rows below the diagonal block, 16 at a time, distributed over the waves: x_s (64 x 64) sits in
//       registers as the A operand, the 16 x 64 piece of L comes straight from the panel (16 consecutive rows
//       per k: 128-B segments) as the B operand -- formed as x_s' L21' so that the lanes run along the ROWS --
//       and is subtracted from x with atomics (the reference's `omp atomic`, Triangular_BCSC.h:154).
typedef double double4_s __attribute__((ext_vector_type(4)));
static constexpr int kRhsM = 64;        // right-hand sides per pass
// The many-right-hand-side kernels (64 per pass over L, matrix cores) take over from the 8-per-pass kernels
//   * for the narrow supernodes (k_solve_small_mrhs) from 6 right-hand sides on,
//   * for the wide supernodes' chain (k_solve_blocks_mrhs) from 2 on
// (parabolic_fem-class input, MI355X, forward solve: 4 right-hand sides 1.02 -> 0.85 ms, 8: 1.41 -> 0.89 ms; Flan-class,
// 8: 18.2 -> 11.6 ms).  PARSY_MRHS_MIN=k sets both thresholds to k (diagnostics, tests).
static int mrhs_env() {
    static const int v = [] {
        const char* e = std::getenv("PARSY_MRHS_MIN");
        return e && *e ? std::atoi(e) : 0;
    }();
    return v;
}
static int mrhs_min() { return mrhs_env() > 0 ? mrhs_env() : 6; }         // narrow supernodes
static int chain_mrhs_min() { return mrhs_env() > 0 ? mrhs_env() : 2; }   // wide supernodes' chain
static int bmrhs_min() {                // ... of the backward solve (PARSY_BMRHS_MIN)
    static const int v = [] {
        const char* e = std::getenv("PARSY_BMRHS_MIN");
        return e && *e ? std::atoi(e) : 16;
    }();
    return v;
}
static constexpr int kLdXs = kRhsM + 4; // row stride of xs in LDS
static constexpr int kSmallMrhsHalvesMin = 512;   // k_solve_small_mrhs<64>: launches of at least this many supernodes sweep in halves

template <int WMAX, bool HALVES = false>  // WMAX: widest supernode of the launch, rounded up to 16 / 32 / 64: sizes LDS and loops
__global__ __launch_bounds__(kThreads) void k_solve_small_mrhs(const SnDesc* __restrict__ sn,
                                                               const int32_t* __restrict__ list,
                                                               const int32_t* __restrict__ ranges,
                                                               const int32_t* __restrict__ rows,
                                                               const double* __restrict__ L,
                                                               double* __restrict__ x, int nrhs, int ldx, int ldq) {
    // X is addressed as x[row * sr + q * sq]: right-hand-side-major (the interface: sr = 1, sq = ldx) or, inside a
    // solve with many right-hand sides, row-major with the right-hand sides of a row contiguous (ldq > 0: sr = ldq,
    // sq = 1 -- a supernode's x block and its write-back are then whole 512-byte rows instead of 64 eight-byte pieces
    // per column, and the thread -> (column, right-hand side) maps of the copies follow the contiguous index)
    const bool tr = ldq > 0;
    const int64_t sr = tr ? ldq : 1, sq = tr ? 1 : ldx;
    constexpr int kLd = WMAX + 1;
    __shared__ double Dg[WMAX * kLd];
    __shared__ double invd[WMAX];
    __shared__ double xs[WMAX * kLdXs];   // xs[c * kLdXs + q]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int q_begin = ranges ? ranges[2 * blockIdx.x] : (int)blockIdx.x;   // subtree launch: as k_solve_small
    const int q_end = ranges ? ranges[2 * blockIdx.x + 1] : q_begin + 1;
  for (int qsn = q_begin; qsn < q_end; ++qsn) {
    if (qsn > q_begin) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const SnDesc D = sn[list[qsn]];
    const int r = D.r, w = D.w;
    const double* __restrict__ G = L + D.px;
    const int32_t* __restrict__ ri = rows + D.pi;
    const int wpad = (w + 15) & ~15;

    for (int e = tid; e < WMAX * WMAX; e += kThreads) {
        const int c = e / WMAX, i = e - c * WMAX;
        double v = (i == c) ? 1.0 : 0.0;
        if (i >= c && i < w && c < w) v = G[(int64_t)c * r + i];
        Dg[c * kLd + i] = v;
    }
    __syncthreads();
    // inverses of the 16x16 diagonal sub-blocks, stored transposed in the strict upper triangle (as
    // block_solve_inv16): Dg[(b+r) * ld + b + c] = inv(L_bb)[r][c], r > c; reciprocal diagonal in invd
    if (tid < WMAX) invd[tid] = 1.0 / Dg[tid * kLd + tid];
    __syncthreads();
    if (tid < WMAX && (tid & ~15) < w) {
        const int b16 = tid & ~15, c = tid & 15;
        double y[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = (k == c) ? invd[b16 + k] : 0.0;
#pragma unroll
        for (int rr = 1; rr < 16; ++rr) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < rr; ++k) sacc = fma(Dg[(b16 + k) * kLd + b16 + rr], y[k], sacc);
            y[rr] = (rr > c) ? -sacc * invd[b16 + rr] : y[rr];
        }
        __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
        for (int rr = 1; rr < 16; ++rr)
            if (rr > c) Dg[(b16 + rr) * kLd + b16 + c] = y[rr];
    }
    for (int q0 = 0; q0 < nrhs; q0 += kRhsM) {
        const int nq = min(kRhsM, nrhs - q0);
        __syncthreads();  // inverses written / xs of the previous pass consumed
        for (int e = tid; e < WMAX * kRhsM; e += kThreads) {
            const int q = tr ? e % kRhsM : e / WMAX, c = tr ? e / kRhsM : e - q * WMAX;
            xs[c * kLdXs + q] = (c < w && q < nq) ? ld_x(&x[(D.c0 + c) * sr + (q0 + q) * sq]) : 0.0;
        }
        __syncthreads();
        // ---- (A) wave `wave` solves its 16 right-hand sides
        if (16 * wave < nq) {
            const int qc = 16 * wave + l15;  // this lane's right-hand side as B operand / result column
            for (int b16 = 0; b16 < wpad; b16 += 16) {
                // y_b = inv(L_bb) x_b
                double4_s acc = {0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int k = 4 * st + kq, i = l15;
                    double av = 0.0;                       // inv(L_bb)[i][k]
                    if (k < i) av = Dg[(b16 + i) * kLd + b16 + k];
                    else if (k == i) av = invd[b16 + i];
                    const double bv = xs[(b16 + k) * kLdXs + qc];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) xs[(b16 + kq + 4 * v) * kLdXs + qc] = acc[v];
                // the sub-blocks below: x_b2 -= L(b2, b) y_b
                for (int b2 = b16 + 16; b2 < wpad; b2 += 16) {
                    double4_s a2 = {0, 0, 0, 0};
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const int k = 4 * st + kq;
                        const double av = Dg[(b16 + k) * kLd + b2 + l15];   // L[b2 + i][b16 + k]
                        const double bv = xs[(b16 + k) * kLdXs + qc];
                        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, a2, 0, 0, 0);
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) xs[(b2 + kq + 4 * v) * kLdXs + qc] -= a2[v];
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < w * nq; e += kThreads) {
            const int q = tr ? e % nq : e / w, c = tr ? e / nq : e - q * w;
            x[(D.c0 + c) * sr + (q0 + q) * sq] = xs[c * kLdXs + q];
        }
        // ---- (B) rows below the diagonal block
        const int nfrag_n = (nq + 15) >> 4;
        // (HALVES, for launches of many supernodes up to 64 wide: the right-hand sides in two halves of 32, the rows below
        // swept once per half -- with the x operands of all 64 in registers the kernel takes 311 of them and ONE workgroup
        // fits a compute unit: Flan-class, levels 3-5, 64 right-hand sides: 12 500 supernodes in 2.75 ms; in halves 214
        // registers, two workgroups: forward solve 18.3 -> 17.6 ms.  A launch of few supernodes has the units to itself
        // either way and keeps the one sweep: nd24k- / mid3d-class +1.4 / +3 % in halves)
        constexpr int kNh = HALVES ? 2 : 4;   // 16-right-hand-side groups per sweep
        if (r > w)
          for (int n0 = 0; n0 < nfrag_n; n0 += kNh) {
            constexpr int kSt = WMAX / 4;
            double xa[kNh][kSt];  // A operands: xa[n][st] = x_s[c = 4 st + kq][rhs 16 (n0 + n) + l15]
#pragma unroll
            for (int n = 0; n < kNh; ++n)
#pragma unroll
                for (int st = 0; st < kSt; ++st) xa[n][st] = xs[(4 * st + kq) * kLdXs + 16 * (n0 + n) + l15];
            for (int k0 = w + 16 * wave; k0 < r; k0 += 16 * (kThreads / 64)) {
                const int row_l = min(k0 + l15, r - 1);
                double lv[kSt];
#pragma unroll
                for (int st = 0; st < kSt; ++st) {
                    const int c = 4 * st + kq;
                    lv[st] = (c < w) ? G[(int64_t)c * r + row_l] : 0.0;
                }
                if (tr) {
                    // X row-major: the product the other way round -- L21 x_s, lanes along the RIGHT-HAND SIDES, the
                    // register index along the rows -- so that an instruction's atomics are four rows x 128 contiguous
                    // bytes (8 requests of 64 bytes with 8 adds each) instead of 16 rows x 32 bytes (16 requests with
                    // 4 adds each): they execute at the memory side at the rate of those requests
                    int xr[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) xr[v] = ri[min(k0 + kq + 4 * v, r - 1)];
#pragma unroll
                    for (int n = 0; n < kNh; ++n) {
                        if (n0 + n < nfrag_n) {
                            double4_s acc = {0, 0, 0, 0};
#pragma unroll
                            for (int st = 0; st < kSt; ++st)
                                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[st], xa[n][st], acc, 0, 0, 0);
                            const int q = 16 * (n0 + n) + l15;
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                if (k0 + kq + 4 * v < r && q < nq) atomicAdd(&x[xr[v] * sr + (q0 + q) * sq], -acc[v]);
                        }
                    }
                    continue;
                }
                const int xrow = ri[row_l];
                const bool rok = k0 + l15 < r;
#pragma unroll
                for (int n = 0; n < kNh; ++n) {
                    if (n0 + n < nfrag_n) {
                        double4_s acc = {0, 0, 0, 0};
#pragma unroll
                        for (int st = 0; st < kSt; ++st)
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[n][st], lv[st], acc, 0, 0, 0);
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int q = 16 * (n0 + n) + kq + 4 * v;
                            if (rok && q < nq) atomicAdd(&x[xrow * sr + (q0 + q) * sq], -acc[v]);
                        }
                    }
                }
            }
        }
    }
  }
}

void launch_solve_small(const DevicePattern& P, int first, int count, int wmax, bool subtrees, const double* L,
                        double* x, int nrhs, int ldx, int ldq, hipStream_t stream) {
    if (count <= 0) return;
    // subtree launch: `first` counts (begin, end) pairs of solve_small_ranges, which index the whole list
    const int32_t* list = subtrees ? P.solve_small_list : P.solve_small_list + first;
    const int32_t* ranges = subtrees ? P.solve_small_ranges + 2 * first : nullptr;
    if (nrhs < mrhs_min() && wmax <= kTinyWidth) {   // one wave per supernode (or per subtree of them)
        hipLaunchKernelGGL(k_solve_tiny<kTinyWidth>, dim3(count, std::min(kPassLanes, nrhs)), dim3(64), 0, stream, P.sn,
                           list, ranges, P.rows, L, x, nrhs, ldx);
        return;
    }
    if (nrhs >= mrhs_min()) {
        if (wmax <= 16)
            hipLaunchKernelGGL(k_solve_small_mrhs<16>, dim3(count), dim3(kThreads), 0, stream, P.sn, list, ranges,
                               P.rows, L, x, nrhs, ldx, ldq);
        else if (wmax <= 32)
            hipLaunchKernelGGL(k_solve_small_mrhs<32>, dim3(count), dim3(kThreads), 0, stream, P.sn, list, ranges,
                               P.rows, L, x, nrhs, ldx, ldq);
        else
            if (count >= kSmallMrhsHalvesMin || nrhs <= 32)   // (at most 32 right-hand sides: one sweep either way, fewer registers)
                hipLaunchKernelGGL((k_solve_small_mrhs<64, true>), dim3(count), dim3(kThreads), 0, stream, P.sn, list, ranges, P.rows, L, x,
                                   nrhs, ldx, ldq);
            else
            hipLaunchKernelGGL(k_solve_small_mrhs<64>, dim3(count), dim3(kThreads), 0, stream, P.sn, list, ranges,
                               P.rows, L, x, nrhs, ldx, ldq);
    } else {
        hipLaunchKernelGGL(k_solve_small, dim3(count, std::min(kPassLanes, (nrhs + kRhs - 1) / kRhs)),
                           dim3(kThreads), 0, stream, P.sn, list, ranges, P.rows, L, x, nrhs, ldx);
    }
}

// SOLVE_PANEL: block column jb of a wide supernode.  Every workgroup solves the
// 64-wide diagonal block for x_jb in LDS (the designated one parks the result in
// xscratch, so the in-place block stays stable while others read it), then applies
// its 256-row chunk of the block column: x[rows] -= L[rows, jb] * x_jb.
__global__ __launch_bounds__(kThreads) void k_solve_panel(const SnDesc* __restrict__ sn,
                                                          const PanelDesc* __restrict__ pds,
                                                          const int32_t* __restrict__ rows,
                                                          const double* __restrict__ L,
                                                          double* __restrict__ x,
                                                          double* __restrict__ xscratch, int nrhs,
                                                          int ldx) {
    __shared__ double Dg[kTile * kLdDiag];
    __shared__ double invd[kTile];
    __shared__ double xs[kTile][kRhs];
    const int tid = threadIdx.x;
    const PanelDesc pd = pds[blockIdx.x];
    const SnDesc D = sn[pd.sn];
    const int r = D.r, cb = pd.jb * kTile, wbk = min(kTile, D.w - cb);
    const double* __restrict__ G = L + D.px;
    const int32_t* __restrict__ ri = rows + D.pi;

    {
        double dtmp[kTile * kTile / kThreads];
#pragma unroll
        for (int t = 0; t < kTile * kTile / kThreads; ++t) {
            const int e = t * kThreads + tid;
            const int c = e >> 6, i = e & 63;
            double v = (i == c) ? 1.0 : 0.0;
            if (c < wbk && i < wbk && i >= c) v = G[(int64_t)(cb + c) * r + cb + i];
            dtmp[t] = v;
        }
#pragma unroll
        for (int t = 0; t < kTile * kTile / kThreads; ++t) {
            const int e = t * kThreads + tid;
            Dg[(e >> 6) * kLdDiag + (e & 63)] = dtmp[t];
        }
    }
    for (int q0 = 0; q0 < nrhs; q0 += kRhs) {
        const int nq = min(kRhs, nrhs - q0);
        __syncthreads();
        for (int e = tid; e < kTile * nq; e += kThreads) {
            const int q = e >> 6, c = e & 63;
            xs[c][q] = (c < wbk) ? x[(int64_t)(q0 + q) * ldx + D.c0 + cb + c] : 0.0;
        }
        __syncthreads();
        if (q0 == 0) block_solve_inv16(Dg, invd, xs, wbk, nq, tid);
        else block_solve_apply16(Dg, invd, xs, wbk, nq, tid);
        if (pd.row0 < 0) {
            for (int e = tid; e < wbk * nq; e += kThreads) {
                const int q = e / wbk, c = e - q * wbk;
                xscratch[(int64_t)(q0 + q) * ldx + D.c0 + cb + c] = xs[c][q];
            }
            continue;
        }
        const int k = pd.row0 + tid;
        if (k < r) {
            double acc[kRhs];
#pragma unroll
            for (int q = 0; q < kRhs; ++q) acc[q] = 0.0;
            for (int c0 = 0; c0 < wbk; c0 += 16) {  // 16 loads in flight, then their FMAs
                double lv[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) lv[c] = (c0 + c < wbk) ? G[(int64_t)(cb + c0 + c) * r + k] : 0.0;
#pragma unroll
                for (int c = 0; c < 16; ++c)
#pragma unroll
                    for (int q = 0; q < kRhs; ++q) acc[q] = fma(lv[c], xs[c0 + c][q], acc[q]);
            }
            const int row = ri[k];  // rows inside the supernode map to its own columns
#pragma unroll
            for (int q = 0; q < kRhs; ++q)
                if (q < nq) atomicAdd(&x[(int64_t)(q0 + q) * ldx + row], -acc[q]);
        }
    }
}

void launch_solve_panel(const DevicePattern& P, int first, int count, const double* L, double* x,
                        double* xscratch, int nrhs, int ldx, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_solve_panel, dim3(count), dim3(kThreads), 0, stream, P.sn,
                       P.solve_panels + first, P.rows, L, x, xscratch, nrhs, ldx);
}

// DIAG_INVERSE: inverse of every 64x64 diagonal block of the wide supernodes (what the chain kernels multiply
// by: x_jb = inv(L_jj) t).  One wave per block, in place in LDS, by halves:
//     inv [A 0; B C] = [inv A, 0; -inv(C) B inv(A), inv C]
// -- the four 16x16 diagonal sub-blocks by substitution (one column per lane), then the 16x16 and the 32x32
// off-diagonal blocks as products on the matrix cores (v_mfma_f64_16x16x4_f64, operands from LDS): 16 block
// products instead of a 64-step substitution whose longest column is a chain of two thousand dependent
// multiply-adds.  dinv[(dslot + jb) * 4096 + c * 64 + i] = inv(L_jj)[i][c], zeros above the diagonal; a block
// narrower than 64 is padded with an identity.
__device__ __forceinline__ double4_s mm16(double4_s acc, const double* __restrict__ X, const double* __restrict__ Y,
                                          int l15, int kq) {
    // acc += X * Y for 16x16 blocks of the LDS matrix (column-major, ld kLdDiag); result layout: lane (l15, kq)
    // holds column l15, rows kq + 4 v
#pragma unroll
    for (int st = 0; st < 4; ++st)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[(4 * st + kq) * kLdDiag + l15], Y[l15 * kLdDiag + 4 * st + kq], acc,
                                                   0, 0, 0);
    return acc;
}
__device__ __forceinline__ void put16(double* __restrict__ Z, double4_s acc, double sign, int l15, int kq) {
#pragma unroll
    for (int v = 0; v < 4; ++v) Z[l15 * kLdDiag + kq + 4 * v] = sign * acc[v];
}

__global__ __launch_bounds__(64) void k_diag_inverse(const SnDesc* __restrict__ sn,
                                                     const int32_t* __restrict__ list,
                                                     const double* __restrict__ L,
                                                     double* __restrict__ dinv) {
    __shared__ double M[kTile * kLdDiag];
    const SnDesc D = sn[list[2 * blockIdx.x]];   // one workgroup per (supernode, block column) pair of the list
    const int jb = list[2 * blockIdx.x + 1], lane = threadIdx.x;
    const int l15 = lane & 15, kq = lane >> 4;
    const int r = D.r, cb = jb * kTile, wbk = min(kTile, D.w - cb);
    const double* __restrict__ G = L + D.px;
    {
        // column cc, row `lane`: all 64 loads in flight together (unconditional: clamped to the block, then masked)
        double v[kTile];
        const double* __restrict__ src = G + (int64_t)cb * r + cb + min(lane, wbk - 1);
#pragma unroll
        for (int cc = 0; cc < kTile; ++cc) v[cc] = src[(int64_t)min(cc, wbk - 1) * r];
#pragma unroll
        for (int cc = 0; cc < kTile; ++cc)
            M[cc * kLdDiag + lane] = (cc < wbk && lane < wbk && lane >= cc) ? v[cc] : (lane == cc ? 1.0 : 0.0);
    }
    __builtin_amdgcn_wave_barrier();   // (one wave: its LDS operations complete in order)
    auto blk = [&](int bi, int bj) { return M + (16 * bj) * kLdDiag + 16 * bi; };   // 16x16 block (bi, bj)
    {
        // the four diagonal sub-blocks: lane (b = kq, c = l15) forms column c of inv(L_bb) and writes it in place
        double* __restrict__ B = blk(kq, kq);
        double y[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            double sacc = (rr == l15) ? -1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < rr; ++k) sacc = fma(B[k * kLdDiag + rr], y[k], sacc);
            y[rr] = (rr >= l15) ? -sacc / B[rr * kLdDiag + rr] : 0.0;
        }
        __builtin_amdgcn_s_waitcnt(0);   // every read of the sub-blocks precedes the in-place writes
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) B[l15 * kLdDiag + rr] = y[rr];
    }
    __builtin_amdgcn_wave_barrier();
    const double4_s zero = {0, 0, 0, 0};
    // the two 32x32 diagonal blocks: block (b+1, b) := -inv(L_{b+1,b+1}) L_{b+1,b} inv(L_bb), b = 0, 2
#pragma unroll
    for (int b = 0; b < 4; b += 2) {
        const double4_s t = mm16(zero, blk(b + 1, b), blk(b, b), l15, kq);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        put16(blk(b + 1, b), t, 1.0, l15, kq);
        __builtin_amdgcn_wave_barrier();
        const double4_s u = mm16(zero, blk(b + 1, b + 1), blk(b + 1, b), l15, kq);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        put16(blk(b + 1, b), u, -1.0, l15, kq);
        __builtin_amdgcn_wave_barrier();
    }
    // the 32x32 block below: X := -inv(C) B inv(A), A = blocks (0..1, 0..1), C = blocks (2..3, 2..3), both inverted
    {
        // T = B inv(A): T_i0 = B_i0 A00 + B_i1 A10, T_i1 = B_i1 A11  (i = 2, 3)
        double4_s t20 = mm16(mm16(zero, blk(2, 0), blk(0, 0), l15, kq), blk(2, 1), blk(1, 0), l15, kq);
        double4_s t30 = mm16(mm16(zero, blk(3, 0), blk(0, 0), l15, kq), blk(3, 1), blk(1, 0), l15, kq);
        double4_s t21 = mm16(zero, blk(2, 1), blk(1, 1), l15, kq);
        double4_s t31 = mm16(zero, blk(3, 1), blk(1, 1), l15, kq);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        put16(blk(2, 0), t20, 1.0, l15, kq);
        put16(blk(3, 0), t30, 1.0, l15, kq);
        put16(blk(2, 1), t21, 1.0, l15, kq);
        put16(blk(3, 1), t31, 1.0, l15, kq);
        __builtin_amdgcn_wave_barrier();
        // X = -inv(C) T: X_2j = -C22 T_2j, X_3j = -(C32 T_2j + C33 T_3j)
        double4_s x20 = mm16(zero, blk(2, 2), blk(2, 0), l15, kq);
        double4_s x21 = mm16(zero, blk(2, 2), blk(2, 1), l15, kq);
        double4_s x30 = mm16(mm16(zero, blk(3, 2), blk(2, 0), l15, kq), blk(3, 3), blk(3, 0), l15, kq);
        double4_s x31 = mm16(mm16(zero, blk(3, 2), blk(2, 1), l15, kq), blk(3, 3), blk(3, 1), l15, kq);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        put16(blk(2, 0), x20, -1.0, l15, kq);
        put16(blk(2, 1), x21, -1.0, l15, kq);
        put16(blk(3, 0), x30, -1.0, l15, kq);
        put16(blk(3, 1), x31, -1.0, l15, kq);
        __builtin_amdgcn_wave_barrier();
    }
    double* __restrict__ out = dinv + (int64_t)(D.dslot + jb) * (kTile * kTile);
    for (int e = lane; e < kTile * kTile; e += 64) {
        const int cc = e >> 6, i = e & 63;
        out[e] = (i >= cc) ? M[cc * kLdDiag + i] : 0.0;
    }
}

void launch_diag_inverse(const DevicePattern& P, int count, const double* L, double* dinv, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_diag_inverse, dim3(count), dim3(64), 0, stream, P.sn, P.solve_wide_list, L, dinv);
}

// SOLVE_CHAIN: the whole block-column chain of a wide supernode in ONE launch.  Workgroup c
// owns rows [256c, 256c+256) of the panel and keeps the running update of its rows in
// registers (pull form: no atomics inside the supernode).  For block column jb the owner of
// rows [64jb, 64jb+64) multiplies its up-to-date rows by the inverse diagonal block
// (DIAG_INVERSE) and publishes x_jb as 8-byte agent-scope atomics + a flag
// (cdna_hip_programming.md Guideline 16, "8-B agent atomics both sides"); workgroups with rows
// below wait for the flag (bounded), read x_jb and update their rows.  Rows below the
// supernode's own columns are scattered once, at the end, with atomics (other supernodes of
// the level update the same ancestor rows).  All workgroups of a launch are resident at
// once (the host caps their number), so the waits cannot starve the workgroup they wait for.
template <int NQ>
__global__ __launch_bounds__(kThreads) void k_solve_chain(const SnDesc* __restrict__ sn,
                                                          const PanelDesc* __restrict__ pds,
                                                          const int32_t* __restrict__ rows,
                                                          const double* __restrict__ L,
                                                          const double* __restrict__ dinv,
                                                          double* __restrict__ x,
                                                          double* __restrict__ xscratch, int nrhs,
                                                          int ldx, int* __restrict__ flags, int epoch0,
                                                          int* __restrict__ info, int* __restrict__ ticket,
                                                          int wait_bias, int nchunks, int fstride) {
    __shared__ double Di[2][kTile * kLdDiag];  // inverse diagonal blocks, double buffered
    __shared__ double xs[kTile][NQ];
    __shared__ double ts[kTile][NQ];
    __shared__ int32_t s_ok, s_task;
    const int tid = threadIdx.x;
    // chunks are listed producers first (the owner of a block column before the chunks below it) and a
    // workgroup takes its chunk from a ticket counter when it starts: whatever it waits for belongs to a
    // workgroup that has already started -- no assumption on residency or dispatch order
    if (tid == 0) s_task = atomicAdd(ticket, 1);
    __syncthreads();
    // the launch holds every chunk once per pass lane: tickets 0..nchunks-1 are lane 0, and so on
    const int plane = s_task / nchunks;
    const PanelDesc pd = pds[s_task - plane * nchunks];
    flags += (int64_t)plane * fstride;   // one set of flags per lane
    const SnDesc D = sn[pd.sn];
    const int r = D.r, w = D.w, chunk = pd.jb, row0 = pd.row0;
    const int nbc = (w + kTile - 1) / kTile;
    const double* __restrict__ G = L + D.px;
    const int k = row0 + tid;          // this thread's panel row
    const bool kv = k < r;
    const bool kdiag = kv && k < w;    // row inside the supernode's own columns
    // block columns whose diagonal block this workgroup owns: [jb_first, jb_last)
    const int jb_first = row0 / kTile, jb_last = min(nbc, (row0 + kSolveRows) / kTile);
    auto load_inv = [&](int jb, double (&regs)[kTile * kTile / kThreads]) {
        const double* __restrict__ src = dinv + (int64_t)(D.dslot + jb) * (kTile * kTile);
#pragma unroll
        for (int t = 0; t < kTile * kTile / kThreads; ++t) regs[t] = src[t * kThreads + tid];
    };
    auto store_inv = [&](int buf, const double (&regs)[kTile * kTile / kThreads]) {
#pragma unroll
        for (int t = 0; t < kTile * kTile / kThreads; ++t) {
            const int e = t * kThreads + tid;
            Di[buf][(e >> 6) * kLdDiag + (e & 63)] = regs[t];
        }
    };
    for (int pass = plane; pass * NQ < nrhs; pass += kPassLanes) {
        const int q0 = pass * NQ;
        const int nq = min(NQ, nrhs - q0);
        const int epoch = epoch0 + pass / kPassLanes;   // round of this lane
        double xv[NQ], acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            acc[q] = 0.0;
            xv[q] = (kdiag && q < nq) ? x[(int64_t)(q0 + q) * ldx + D.c0 + k] : 0.0;
        }
        double inv_regs[kTile * kTile / kThreads];
        if (jb_first < jb_last) {
            load_inv(jb_first, inv_regs);
            store_inv(jb_first & 1, inv_regs);
        }
        for (int jb = 0; jb < nbc; ++jb) {
            const int cb = jb * kTile, wbk = min(kTile, w - cb);
            const int owner = cb / kSolveRows;
            if (chunk < owner) break;  // no rows at or below this block column
            // this thread's row of L against block column jb: all 64 loads are issued here, BEFORE the
            // wait for x_jb, so that their latency hides behind the hand-off
            const bool below = kv && k >= cb + wbk;
            double lv[kTile];
            if (below) {
#pragma unroll
                for (int c = 0; c < kTile; ++c) lv[c] = (c < wbk) ? G[(int64_t)(cb + c) * r + k] : 0.0;
            }
            __syncthreads();           // xs / ts of the previous block column are free
            if (chunk == owner) {
                const int lr = k - cb;  // row inside the block for the 64 threads that hold it
                if (lr >= 0 && lr < kTile) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) ts[lr][q] = (lr < wbk) ? xv[q] - acc[q] : 0.0;
                }
                __syncthreads();
                // x_jb = inv(Ljj) t : thread (i, q-slice); the next owned inverse is fetched meanwhile
                const bool next_owned = jb + 1 < jb_last;
                if (next_owned) load_inv(jb + 1, inv_regs);
                {
                    const double* __restrict__ Dv = Di[jb & 1];
                    for (int e = tid; e < kTile * nq; e += kThreads) {
                        const int i = e & 63, q = e >> 6;
                        double sacc = 0.0;
                        for (int k2 = 0; k2 <= i; ++k2) sacc = fma(Dv[k2 * kLdDiag + i], ts[k2][q], sacc);
                        xs[i][q] = sacc;
                    }
                }
                __syncthreads();
                for (int e = tid; e < wbk * nq; e += kThreads) {
                    const int q = e / wbk, c = e - q * wbk;
                    const double v = xs[c][q];
                    __hip_atomic_store(&xscratch[(int64_t)(q0 + q) * ldx + D.c0 + cb + c], v, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    x[(int64_t)(q0 + q) * ldx + D.c0 + cb + c] = v;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0)
                    __hip_atomic_store(&flags[D.dslot + jb], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (next_owned) store_inv((jb + 1) & 1, inv_regs);
            } else {
                if (tid == 0) {
                    const unsigned long long t0 = wall_clock64();
                    int ok = 1;
                    // epochs only grow: a later pass of this solve may already have raised the flag.  The wait is
                    // bounded and watches the solve's status word: one timeout ends every wait of the solve
                    int spins = 0;
                    while (__hip_atomic_load(&flags[D.dslot + jb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                               (epoch + wait_bias) < 0) {
                        if ((++spins & 15) == 0 &&
                            (wall_clock64() - t0 > kSolveSpinTicks ||
                             __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
                            ok = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    s_ok = ok;
                }
                __syncthreads();
                if (!s_ok) {
                    if (tid == 0) atomicMin(info, -1);
                    return;
                }
                for (int e = tid; e < kTile * nq; e += kThreads) {
                    const int q = e >> 6, c = e & 63;
                    xs[c][q] = (c < wbk) ? __hip_atomic_load(&xscratch[(int64_t)(q0 + q) * ldx + D.c0 + cb + c],
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                         : 0.0;
                }
                __syncthreads();
            }
            // rows strictly below the diagonal block: acc += L[k, cb..cb+wbk) x_jb
            if (below) {
#pragma unroll
                for (int c = 0; c < kTile; ++c)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q] = fma(lv[c], xs[c][q], acc[q]);
            }
        }
        if (kv && !kdiag) {
            const int row = rows[D.pi + k];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q < nq) atomicAdd(&x[(int64_t)(q0 + q) * ldx + row], -acc[q]);
        }
    }
}

// SOLVE_CHAIN for ONE right-hand side: a dataflow of single waves, no workgroup barrier (but one when the
// workgroup starts) and no flag in memory.
// A workgroup = eight waves = one of the launch's 256-row chunks, taken from the launch's ticket counter (tickets
// follow the row order, so whatever a wave waits for belongs to a workgroup that has started).  Two waves share
// each 64-row block of the chunk (lane = row): wave p of the pair streams the block columns jb = p, p + 2, ... to
// the left of its rows -- takes x_jb (64 values), applies its 64 x 64 piece of L (all 64 loads of a lane's row
// issued BEFORE the wait) -- and keeps its part of the rows' running sum in registers.  A wave has 32 KiB of L in
// flight, so a row block streams at twice that: what bounds a long chain is how fast the rows right below the
// diagonal get through their block columns, not only the hand-off.  When the rows are diagonal block jb of the
// supernode, wave 1 hands its part to wave 0 (LDS), which forms t = x - sum, x_jb = inv(L_jj) t (inverse diagonal
// block staged in LDS when the wave starts: lower triangle, packed) and publishes the 64 values.
// The hand-off is the data itself.  Across workgroups: xscratch is armed with a signalling-NaN pattern (kXArmed)
// before the solve and a value is valid as soon as it differs from it -- every value is one 8-byte agent-scope
// atomic, so the producer neither waits for its stores nor raises a flag, and the consumer's poll (lane c polls
// value c) is the load of the data: one memory round trip per hand-off instead of three.  Inside a workgroup --
// three hand-offs of the critical path out of four, diagonal block to next diagonal block -- x_jb goes through
// LDS (one flag per row block, raised after its 64 values are written).
// Arithmetic never produces the armed pattern (results of operations on NaNs are quiet NaNs); should the data
// hold it, the wait times out and the solve's status word reports it, like any other abandoned hand-off.
// Rows below the supernode's own columns are scattered at the end with atomics, as in the other kernels.
static constexpr unsigned kXArmedWord = 0xFFF7A5A5u;   // both 32-bit halves of the armed pattern
static constexpr long long kXArmed = (long long)(((unsigned long long)kXArmedWord << 32) | kXArmedWord);
// A published value must differ from the armed pattern.  Arithmetic cannot produce it (operations on NaNs return quiet
// NaNs), only an input can carry it: it is published as a quiet NaN instead, so that the hand-off completes and the
// NaN propagates like any other (without this such a right-hand side ran into the 2 s timeout and status -1).
__device__ __forceinline__ double unarmed(double v) {
    return __double_as_longlong(v) == kXArmed ? __longlong_as_double(0x7FF8000000000000LL) : v;
}
static constexpr int kInvPacked = kTile * (kTile + 1) / 2;   // packed lower triangle of an inverse diagonal block
static constexpr int kChainThreads = 2 * kSolveRows;         // eight waves
static constexpr int kChainFewChunks = 384;                  // launches of at most this many chunks: top of the tree

// RB = 64-row blocks per workgroup (8 / RB waves share each of them): 4 where a launch has plenty of chunks, 2
// where it is the top of the tree -- few, very wide supernodes: twice the workgroups (a CU sustains about
// 45 GB/s of this access pattern, so the number of CUs that stream is what bounds a launch of few chunks) and
// twice the waves behind every row block, for half of the diagonal-to-diagonal hand-offs inside a workgroup.
template <int RB>
__global__ __launch_bounds__(kChainThreads, 1) void k_solve_chain_w(const SnDesc* __restrict__ sn,
                                                                    const PanelDesc* __restrict__ pds,
                                                                    const int32_t* __restrict__ rows,
                                                                    const double* __restrict__ L,
                                                                    const double* __restrict__ dinv,
                                                                    double* __restrict__ x, double* __restrict__ xscratch,
                                                                    int* __restrict__ info, int* __restrict__ ticket,
                                                                    int wait_bias) {
    constexpr int G = kChainThreads / 64 / RB;   // waves per row block
    constexpr int kParts = kSolveRows / kTile / RB;   // workgroups per 256-row chunk of the launch
    __shared__ double s_inv[RB][kInvPacked + 1];   // column c of the inverse from row c on, at c*64 - c(c-1)/2
                                                   // (+ one slot for the lanes above it)
    __shared__ double s_x[RB * G][kTile];          // per wave: x_jb taken from memory / t
    __shared__ double s_part[RB][G - 1][kTile];    // the other waves' parts of a diagonal block's sums, for wave 0
    __shared__ double s_pub[RB][kTile];            // x of a diagonal block, for the waves of the blocks after it
    __shared__ int s_parts_in[RB], s_ready[RB];
    __shared__ int s_task;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rb = wave / G, p = wave % G;   // row block of the workgroup, wave of its group
    if (threadIdx.x == 0) s_task = atomicAdd(ticket, 1);
    if (threadIdx.x < RB) {
        s_ready[threadIdx.x] = 0;
        s_parts_in[threadIdx.x] = 0;
    }
    __syncthreads();   // (the only one: from here on the waves go their own way)
    const PanelDesc pd = pds[s_task / kParts];
    const SnDesc D = sn[pd.sn];
    const int r = D.r, w = D.w;
    const int row0_wg = pd.row0 + (s_task % kParts) * (RB * kTile);
    const int row0 = row0_wg + kTile * rb;
    if (row0 >= r) return;   // (the last chunk of a panel may have fewer than four blocks)
    const int nbc = (w + kTile - 1) / kTile;
    const double* __restrict__ G_ = L + D.px;
    const int k = row0 + lane;          // this lane's panel row
    const bool kv = k < r;
    const bool is_diag = row0 < w;      // the rows hold diagonal block row0 / 64
    const int nblk = is_diag ? row0 / kTile : nbc;   // block columns to the left of the rows
    const int jb_wg = row0_wg / kTile;               // block columns from here on belong to this workgroup's waves
    double* __restrict__ xs = s_x[wave];
    double* __restrict__ inv = s_inv[rb];

    double lv[kTile];
    // every load is unconditional (no branch per column): rows past the panel read its last row (their sum is
    // never used), columns past the supernode read its last column again (their x is zero).  The address walks
    // from column to column by the panel's stride: one vector add per load, nothing else.
    const char* __restrict__ lane_base = reinterpret_cast<const char*>(G_ + min(k, r - 1));
    const int64_t col_stride = (int64_t)r * (int64_t)sizeof(double);
    auto load_block = [&](int jb) {
        const int cb = jb * kTile, last = w - 1 - cb;   // columns 0..last of the block exist (last >= 0)
        const char* pcol = lane_base + (int64_t)cb * col_stride;
#pragma unroll
        for (int c = 0; c < kTile; ++c) {
            lv[c] = *reinterpret_cast<const double*>(pcol);
            pcol += (c < last) ? col_stride : 0;
        }
    };
    auto inv_at = [&](int c) { return lane >= c ? c * kTile - c * (c - 1) / 2 + (lane - c) : kInvPacked; };
    auto gave_up = [&](unsigned long long t0, int& spins) {   // bounded waits: one timeout ends every wait of the solve
        if ((++spins & 15) != 0) return false;
        return wall_clock64() - t0 > kSolveSpinTicks ||
               __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0;
    };
    auto wait_lds = [&](int* flag, int want) {   // a counter of this workgroup's LDS
        const unsigned long long t0 = wall_clock64();
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want || wait_bias != 0) {
            if (gave_up(t0, spins)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        return true;
    };
    if (is_diag && p == 0) {
        // inverse diagonal block -> LDS (off the critical path: the wave starts well before its turn)
        const double* __restrict__ src = dinv + (int64_t)(D.dslot + row0 / kTile) * (kTile * kTile);
#pragma unroll
        for (int c = 0; c < kTile; ++c)
            lv[c] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(src + c * kTile) + (unsigned)lane * 8u);
#pragma unroll
        for (int c = 0; c < kTile; ++c) inv[inv_at(c)] = lv[c];
    }
    const double xv = (is_diag && p == 0 && k < w) ? x[D.c0 + k] : 0.0;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int jb = p; jb < nblk; jb += G) {
        const int cb = jb * kTile, wbk = min(kTile, w - cb);
        load_block(jb);   // all 64 loads of the lane's row are issued BEFORE the wait for x_jb
        const double* __restrict__ xb = xs;
        if (jb >= jb_wg) {
            // ---- x_jb comes from a wave of this workgroup: LDS
            if (!wait_lds(&s_ready[jb - jb_wg], 1)) {
                if (lane == 0) atomicMin(info, -1);   // (the result is wrong and reported; nobody may hang)
                return;
            }
            xb = s_pub[jb - jb_wg];
        } else {
            // ---- x_jb comes from another workgroup: lane c polls value c of the armed buffer
            const long long* __restrict__ src = reinterpret_cast<const long long*>(xscratch + D.c0 + cb) + lane;
            long long bits = 0;
            const unsigned long long t0 = wall_clock64();
            int spins = 0;
            for (;;) {
                bits = lane < wbk ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                if (__all(bits != kXArmed && wait_bias == 0)) break;
                if (gave_up(t0, spins)) {
                    if (lane == 0) atomicMin(info, -1);
                    return;
                }
                // the rows right below the diagonal wait for the newest x: the critical path; the rows further
                // down are never critical and poll at leisure
                if (is_diag && jb >= nblk - G) __builtin_amdgcn_s_sleep(1);
                else __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_wave_barrier();
            xs[lane] = __longlong_as_double(bits);
            __builtin_amdgcn_wave_barrier();   // (one wave: LDS operations complete in order)
        }
#pragma unroll
        for (int c = 0; c < kTile; c += 4) {
            // (16 columns at a time: all 64 LDS reads hoisted above the products would not fit beside lv)
            if ((c & 15) == 0) __builtin_amdgcn_sched_barrier(0);
            a0 = fma(lv[c], xb[c], a0);
            a1 = fma(lv[c + 1], xb[c + 1], a1);
            a2 = fma(lv[c + 2], xb[c + 2], a2);
            a3 = fma(lv[c + 3], xb[c + 3], a3);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    double acc = (a0 + a1) + (a2 + a3);
    if (is_diag) {
        const int wbk = min(kTile, w - row0);
        if (p != 0) {
            // this wave's part of the sums -> wave 0 of the group
            s_part[rb][p - 1][lane] = acc;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_fetch_add(&s_parts_in[rb], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (!wait_lds(&s_parts_in[rb], G - 1)) {
                if (lane == 0) atomicMin(info, -1);
                return;
            }
            double others = 0.0;
#pragma unroll
            for (int q = 0; q < G - 1; ++q) others += s_part[rb][q][lane];
            __builtin_amdgcn_wave_barrier();
            xs[lane] = (lane < wbk) ? xv - (acc + others) : 0.0;
            __builtin_amdgcn_wave_barrier();
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int c = 0; c < kTile; c += 2) {
                if ((c & 15) == 0) __builtin_amdgcn_sched_barrier(0);
                const double d0 = inv[inv_at(c)], d1 = inv[inv_at(c + 1)];
                s0 = fma(c <= lane ? d0 : 0.0, xs[c], s0);
                s1 = fma(c + 1 <= lane ? d1 : 0.0, xs[c + 1], s1);
            }
            const double xi = (lane < wbk) ? s0 + s1 : 0.0;
            // to the waves of this workgroup through LDS (flag after the data) ...
            s_pub[rb][lane] = xi;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_store(&s_ready[rb], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            // ... to everybody else through the armed buffer, and into x
            if (lane < wbk) {
                __hip_atomic_store(&xscratch[D.c0 + row0 + lane], unarmed(xi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x[D.c0 + row0 + lane] = xi;
            }
            // a last diagonal block narrower than 64: the other rows of the block lie below the supernode's columns
            // and take this block column too
            if (kv && k >= w)
                for (int c = 0; c < wbk; ++c) acc = fma(G_[(int64_t)(row0 + c) * r + k], s_pub[rb][c], acc);
        }
    }
    // rows below the supernode's columns: each wave of the group scatters its own part of the sums
    if (kv && k >= w) atomicAdd(&x[rows[D.pi + k]], -acc);
}

// SOLVE_CHAIN for many right-hand sides (nrhs >= 16): the same protocol (256-row chunks, pull form, the owner of a
// block column publishes x_jb, tickets, flags per lane of passes) with 64 right-hand sides per pass over the panel
// and every product on v_mfma_f64_16x16x4_f64, so that L is read once per 64 right-hand sides.  Wave v of a chunk
// owns its rows 64 v .. 64 v + 63; the running update of a row lives in accumulator layout as
// D[i = right-hand side][j = row] (lanes along the rows), started at -x for the rows of the supernode's own
// columns, so that the owner's t_jb = x - sum is just the negated accumulator.  x_jb = inv(L_jj) t_jb is an MFMA
// product too (t through LDS into operand layout); the rows below the supernode's columns are scattered at the end
// with atomics, as in the one-vector kernel.
static constexpr int kMrhsWideBlocks = 128;   // backward launches of at least this many blocks take 64 right-hand sides per pass
static constexpr int kBChainMinBlocks = 128;  // ... chain launches: k_bsolve_chain_mrhs from this many block columns on
static constexpr int kLdXm = kRhsM + 16;  // row stride of the staged x_jb / t_jb (doubles): conflict-free operand reads
static constexpr int kSolveRowsM = kSolveRowsMrhs;   // rows of a chunk task of k_solve_blocks_mrhs (schedule.hpp)

__global__ __launch_bounds__(kThreads, 1) void k_solve_chain_mrhs(const SnDesc* __restrict__ sn,
                                                                  const PanelDesc* __restrict__ pds,
                                                                  const int32_t* __restrict__ rows,
                                                                  const double* __restrict__ L,
                                                                  const double* __restrict__ dinv,
                                                                  double* __restrict__ x, double* __restrict__ xscratch,
                                                                  int nrhs, int ldx, int* __restrict__ flags, int epoch0,
                                                                  int* __restrict__ info, int* __restrict__ ticket,
                                                                  int wait_bias, int nchunks, int fstride) {
    __shared__ double Di[kTile * kLdDiag];   // inverse diagonal block of the block column being solved (owner)
    __shared__ double xs[kTile * kLdXm];     // x_jb: xs[c * kLdXm + q]
    __shared__ double ts[kTile * kLdXm];     // t_jb: ts[row * kLdXm + q]
    __shared__ int32_t s_ok, s_task;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    if (tid == 0) s_task = atomicAdd(ticket, 1);
    __syncthreads();
    const int plane = s_task / nchunks;
    const PanelDesc pd = pds[s_task - plane * nchunks];
    flags += (int64_t)plane * fstride;
    const SnDesc D = sn[pd.sn];
    const int r = D.r, w = D.w, chunk = pd.jb, row0 = pd.row0;
    const int nbc = (w + kTile - 1) / kTile;
    const double* __restrict__ G = L + D.px;
    const int wrow0 = row0 + 64 * wave;                 // first panel row of this wave
    int prow[4];                                        // this lane's row of each 16-row fragment (-1: past the panel)
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) prow[rf] = (wrow0 + 16 * rf + l15 < r) ? wrow0 + 16 * rf + l15 : -1;

    for (int pass = plane; pass * kRhsM < nrhs; pass += kPassLanes) {
        const int q0 = pass * kRhsM;
        const int nq = min(kRhsM, nrhs - q0);
        const int nfn = (nq + 15) >> 4;                 // 16-wide fragments of right-hand sides in use
        const int epoch = epoch0 + pass / kPassLanes;
        double4_s acc[4][4];                            // [fragment of right-hand sides][fragment of rows]
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int rf = 0; rf < 4; ++rf)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int q = 16 * nf + kq + 4 * v;
                    const bool diag = prow[rf] >= 0 && prow[rf] < w && q < nq;
                    acc[nf][rf][v] = diag ? -x[(int64_t)(q0 + q) * ldx + D.c0 + prow[rf]] : 0.0;
                }
        for (int jb = 0; jb < nbc; ++jb) {
            const int cb = jb * kTile, wbk = min(kTile, w - cb);
            const int owner = cb / kSolveRows;
            if (chunk < owner) break;
            // this wave's rows of L against block column jb, in B-operand layout (lane = (k >> 2 group, row)):
            // issued before the wait for x_jb
            const bool wave_below = wrow0 + 64 > cb + wbk && wrow0 < r;
            double lv[4][16];
            if (wave_below) {
#pragma unroll
                for (int rf = 0; rf < 4; ++rf)
#pragma unroll
                    for (int st = 0; st < 16; ++st) {
                        const int c = 4 * st + kq;
                        const bool ok = prow[rf] >= cb + wbk && c < wbk;
                        lv[rf][st] = ok ? G[(int64_t)(cb + c) * r + prow[rf]] : 0.0;
                    }
            }
            __syncthreads();  // xs / ts of the previous block column are free
            if (chunk == owner) {
                // t_jb = -(accumulator) of the block's 64 rows (one wave holds them) -> LDS, row-major
                const int wv_o = (cb - row0) >> 6;
                for (int e = tid; e < kTile * kTile; e += kThreads) {  // inverse diagonal block -> LDS
                    Di[(e >> 6) * kLdDiag + (e & 63)] = dinv[(int64_t)(D.dslot + jb) * (kTile * kTile) + e];
                }
                if (wave == wv_o) {
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                        for (int rf = 0; rf < 4; ++rf)
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                ts[(16 * rf + l15) * kLdXm + 16 * nf + kq + 4 * v] =
                                    (16 * rf + l15 < wbk) ? -acc[nf][rf][v] : 0.0;
                }
                __syncthreads();
                // x_jb = inv(L_jj) t_jb: wave n takes the right-hand sides 16 n .. 16 n + 15
                if (wave < nfn) {
                    double4_s xa[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
                    for (int st = 0; st < 16; ++st) {
                        const int kr = 4 * st + kq;
                        const double av = ts[kr * kLdXm + 16 * wave + l15];          // t[kr][q]
#pragma unroll
                        for (int rf = 0; rf < 4; ++rf) {
                            const double bv = Di[kr * kLdDiag + 16 * rf + l15];      // inv[row][kr]
                            xa[rf] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, xa[rf], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int rf = 0; rf < 4; ++rf)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int q = 16 * wave + kq + 4 * v, c = 16 * rf + l15;
                            xs[c * kLdXm + q] = xa[rf][v];
                            if (c < wbk && q < nq) {
                                __hip_atomic_store(&xscratch[(int64_t)(q0 + q) * ldx + D.c0 + cb + c], xa[rf][v],
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                x[(int64_t)(q0 + q) * ldx + D.c0 + cb + c] = xa[rf][v];
                            }
                        }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0)
                    __hip_atomic_store(&flags[D.dslot + jb], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (tid == 0) {
                    const unsigned long long t0 = wall_clock64();
                    int ok = 1, spins = 0;
                    while (__hip_atomic_load(&flags[D.dslot + jb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                               (epoch + wait_bias) < 0) {
                        if ((++spins & 15) == 0 &&
                            (wall_clock64() - t0 > kSolveSpinTicks ||
                             __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
                            ok = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    s_ok = ok;
                }
                __syncthreads();
                if (!s_ok) {
                    if (tid == 0) atomicMin(info, -1);
                    return;
                }
                for (int e = tid; e < kTile * kRhsM; e += kThreads) {
                    const int q = e >> 6, c = e & 63;
                    xs[c * kLdXm + q] = (c < wbk && q < nq)
                                            ? __hip_atomic_load(&xscratch[(int64_t)(q0 + q) * ldx + D.c0 + cb + c],
                                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                            : 0.0;
                }
                __syncthreads();
            }
            // rows below the diagonal block: accumulator += x_jb' L(rows, jb)'
            if (wave_below) {
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    if (nf < nfn) {
#pragma unroll
                        for (int st = 0; st < 16; ++st) {
                            const double av = xs[(4 * st + kq) * kLdXm + 16 * nf + l15];
#pragma unroll
                            for (int rf = 0; rf < 4; ++rf)
                                acc[nf][rf] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, lv[rf][st], acc[nf][rf], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // rows below the supernode's own columns: x[row] -= accumulated update
#pragma unroll
        for (int rf = 0; rf < 4; ++rf) {
            if (prow[rf] >= w) {
                const int xrow = rows[D.pi + prow[rf]];
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int q = 16 * nf + kq + 4 * v;
                        if (q < nq) atomicAdd(&x[(int64_t)(q0 + q) * ldx + xrow], -acc[nf][rf][v]);
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Forward chain launch for many right-hand sides, round 3 (replaces k_solve_chain_mrhs's protocol: flag + barrier +
// staged copy per block column, 35 us per block column on the parabolic_fem-class root): the armed buffer of the
// one-vector kernels -- the data is the flag -- and two kinds of workgroups, taken by ticket, producers first:
//   * one per BLOCK COLUMN jb of a wide supernode (task row0 < 0), the counterpart of k_bsolve_block_mrhs's chain form:
//       T = B_jb - sum_{k < jb} L(jb, k) X_k,     X_jb = inv(L_jj) T
//     with the COLUMNS of the earlier blocks as the contraction index: wave v takes columns 16 v .. 16 v + 15 of every
//     earlier block k as soon as X_k is there (lane (row, kk) holds L[cb + row][64 k + 16 v + 4 st + kk]: 128-byte
//     segments along the rows; lane (q, kk) holds X_k through the armed buffer), keeps its part of T in 16 accumulator
//     tiles, the four parts are subtracted from the staged B in a fixed order, the product with the inverse diagonal
//     block (DIAG_INVERSE) is on the matrix cores too; only the last k is on the chain's critical path (the loads of
//     L(jb, k) are issued before the wait);
//   * one per 128-ROW CHUNK of the rows below the supernode's own columns (row0 >= w): the running update of its rows
//     over ALL block columns in accumulator layout, each X_jb staged through LDS as soon as it is published, one
//     atomicAdd per (row, right-hand side) at the end (reference Triangular_BCSC.h:139-157: the `omp atomic` scatter).
// 64 right-hand sides per pass over L.  Every wait is bounded and watches the solve's status word.
// ---------------------------------------------------------------------------
#ifdef PARSY_BLKSTAMPS
// (diagnostic build, tools/blk_stamps.py) the block-column tasks of the LAST launch leave, per block column jb: 100-MHz
// clock when X_(jb-1) was seen whole, after the products with it, after the product with the inverse block, after the stores
__device__ unsigned long long g_blkstamp[512 * 8];
#define BLK_STAMP(i) do { if (tid == 0 && plane == 0 && jb < 512) g_blkstamp[jb * 8 + (i)] = wall_clock64(); } while (0)
extern "C" void parsy_debug_blkstamps(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blkstamp), sizeof(unsigned long long) * 512 * 8);
}
#else
#define BLK_STAMP(i) do { } while (0)
#endif
#define TSF(c, q) ts[(c) * kLdXm + (q)]
template <bool NARROW>   // at most 16 right-hand sides in all: the block-column tasks' waves share the rows of one group
__global__ __launch_bounds__(kThreads, 1) void k_solve_blocks_mrhs(const SnDesc* __restrict__ sn,
                                                                   const PanelDesc* __restrict__ pds,
                                                                   const int32_t* __restrict__ rows,
                                                                   const double* __restrict__ L,
                                                                   const double* __restrict__ dinv,
                                                                   double* __restrict__ x, double* __restrict__ xscratch,
                                                                   int nrhs, int ldx, int ldq, int* __restrict__ info,
                                                                   int* __restrict__ ticket, int wait_bias, int ntasks) {
    // (x and the armed buffer as x[row * sr + q * sq]: k_solve_small_mrhs)
    const bool tr = ldq > 0;
    const int64_t sr = tr ? ldq : 1, sq = tr ? 1 : ldx;
    __shared__ double Dg[kTile * kLdDiag];   // block task: inverse diagonal block, Dg[k][row] = inv(L_jj)[row][k]
    __shared__ double ts[kTile * kLdXm];     // block task: M; chunk task: the staged X_jb as [col][q]
    __shared__ double Tx[kTile * 17];        // block task, at most 16 right-hand sides: T' as [row][q]
    __shared__ int32_t s_task, s_ok;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    if (tid == 0) {
        s_task = atomicAdd(ticket, 1);
        s_ok = 1;
    }
    __syncthreads();
    const int plane = s_task / ntasks;
    const PanelDesc pd = pds[s_task - plane * ntasks];
    const SnDesc D = sn[pd.sn];
    const int r = D.r, w = D.w;
    const int nbc = (w + kTile - 1) / kTile;
    const double* __restrict__ G = L + D.px;
    // all lanes of a wave: true when every value of `vals` differs from the armed pattern
    auto give_up = [&](unsigned long long t0, int& spins) {
        return (++spins & 15) == 0 && (wall_clock64() - t0 > kSolveSpinTicks ||
                                       __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0);
    };

    if (pd.row0 < 0) {
        // ================= a block column of the triangle =================
        // Every wave works for itself on 16 right-hand sides and ALL 64 rows of the block -- no LDS traffic and no
        // barrier on the chain's path: lane (q, kk) keeps T[4 st + kk][q], st = 0..15, in registers (the accumulator
        // layout of v_mfma_f64_16x16x4_f64 -- lane (q, kk) holds rows kk + 4 v of a 16-row group -- IS the B-operand
        // layout of the product with the inverse block: k step st = 4 rg + v), the rows of L(jb, k) come straight from
        // the panel as A operands (each wave reads all of them: L2 hits for three of the four).
        const int jb = pd.jb, cb = jb * kTile, wbk = min(kTile, w - cb);
        {   // inverse diagonal block -> LDS (the only shared data; one barrier per task)
            const double* __restrict__ inv_blk = dinv + (int64_t)(D.dslot + jb) * (kTile * kTile);
            double dtmp[kTile * kTile / kThreads];
#pragma unroll
            for (int t = 0; t < kTile * kTile / kThreads; ++t) dtmp[t] = inv_blk[t * kThreads + tid];
#pragma unroll
            for (int t = 0; t < kTile * kTile / kThreads; ++t) {
                const int e = t * kThreads + tid;
                Dg[(e >> 6) * kLdDiag + (e & 63)] = dtmp[t];
            }
        }
        __syncthreads();
        // The chain's step, block column jb - 1 -> jb, was 64 products with X_(jb-1), 40 with the inverse block and the
        // stores, one after the other (tools/blk_stamps.py: 2.4 + 1.4 + 0.8 of a 5.7-us hop, the hand-off itself 1.0).  With
        //     M = inv(L_jj) L(jb, jb-1)   (64 x 64, needs no X: formed here, wave v its columns 16 v .. 16 v + 15, -> LDS)
        //     P = inv(L_jj) (B_jb - sum_{k < jb-1} L(jb, k) X_k)   (formed while X_(jb-1) is still on its way)
        // what is left behind the wait is X_jb = P - M X_(jb-1): the 64 products alone, in two independent sets of
        // accumulators.
        double* __restrict__ Ms = ts;   // Ms[k][row] = M[row][k], ld kLdDiag (the layout of Dg: the same operand reads)
        if (jb > 0) {
            const int cbp = cb - kTile;
            double4_s am[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) am[rg] = double4_s{0, 0, 0, 0};
            // B operand: lane (j = l15, kk = kq) holds L(jb, jb-1)[4 st + kk][16 v + j]
            const double* __restrict__ bcol = G + (int64_t)(cbp + 16 * wave + l15) * r + cb;
            double lb[16];
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int i = 4 * st + kq;
                const double v = bcol[min(i, wbk - 1)];
                lb[st] = i < wbk ? v : 0.0;
            }
#pragma unroll
            for (int st = 0; st < 16; ++st)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    if (st < 4 * rg + 4)
                        am[rg] = __builtin_amdgcn_mfma_f64_16x16x4f64(Dg[(4 * st + kq) * kLdDiag + 16 * rg + l15], lb[st], am[rg], 0, 0, 0);
            // accumulator (lane (j, kk): rows 16 rg + kk + 4 v of column 16 wave + j) -> Ms[column][row]
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                for (int v = 0; v < 4; ++v) Ms[(16 * wave + l15) * kLdDiag + 16 * rg + kq + 4 * v] = am[rg][v];
            __syncthreads();
        }
        // this lane's rows of the block as A operand: row 16 rg + l15 (clamped into the block)
        const double* __restrict__ arow[4];
        bool aok[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            aok[rg] = 16 * rg + l15 < wbk;
            arow[rg] = G + cb + min(16 * rg + l15, wbk - 1);
        }
        for (int pass = plane; pass * kRhsM < nrhs; pass += kPassLanes) {
            const int q0 = pass * kRhsM;
            const int nq = min(kRhsM, nrhs - q0);
            if constexpr (NARROW) {
                // ---- at most 16 right-hand sides: ONE group, and the four waves share its ROWS -- wave v the rows
                // 16 v .. 16 v + 15 of the block -- instead of three of them standing by: 16 products per earlier block and
                // 16 behind the last wait (64 on one wave before: parabolic_fem-class, 8 right-hand sides, the chains 305 us of
                // a 511-us solve).  P = inv(L_jj) T' needs the rows above a wave's own: T' goes through LDS once, before the wait.
                const bool qok = l15 < nq;
                const int64_t qoff = (q0 + min(l15, nq - 1)) * sq + D.c0 * sr;
                const double* __restrict__ xq = xscratch + qoff;
                const bool aokw = 16 * wave + l15 < wbk;
                const double* __restrict__ ar = G + cb + min(16 * wave + l15, wbk - 1);
                double tvo[4];   // this wave's rows of B_jb: rows 16 wave + 4 v + kq (the accumulator layout)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = 16 * wave + 4 * v + kq;
                    const double t = x[qoff + (cb + min(c, wbk - 1)) * sr];
                    tvo[v] = (qok && c < wbk) ? t : 0.0;
                }
                bool ok = true;
                auto take_x = [&](int k, bool lazy, double (&bv)[16]) __attribute__((always_inline)) {
                    const unsigned long long t0 = wall_clock64();
                    int spins = 0;
                    if (lazy) {
                        const long long* __restrict__ watch = reinterpret_cast<const long long*>(xq + (k * kTile + 63) * sr);
                        while (__hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kXArmed || wait_bias != 0) {
                            if (give_up(t0, spins)) {
                                ok = false;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(48);
                        }
                    }
                    while (ok) {
                        bool in = true;
#pragma unroll
                        for (int st = 0; st < 16; ++st) {
                            const long long b = __hip_atomic_load(reinterpret_cast<const long long*>(xq + (k * kTile + 4 * st + kq) * sr),
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            in = in && b != kXArmed;
                            bv[st] = qok ? __longlong_as_double(b) : 0.0;
                        }
                        if (__all((in || !qok) && wait_bias == 0)) break;
                        if (give_up(t0, spins)) ok = false;
                        else __builtin_amdgcn_s_sleep(1);
                    }
                };
                double4_s a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
                for (int k = 0; k < jb - 1 && ok; ++k) {
                    double av[16];
#pragma unroll
                    for (int st = 0; st < 16; ++st) {
                        const double a = ar[(int64_t)(k * kTile + 4 * st + kq) * r];
                        av[st] = aokw ? a : 0.0;
                    }
                    double bv[16];
                    take_x(k, true, bv);
                    if (!ok) break;
#pragma unroll
                    for (int st = 0; st < 16; st += 2) {
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], bv[st], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st + 1], bv[st + 1], a1, 0, 0, 0);
                    }
                }
                // (a hand-off that timed out is reported; the waves leave TOGETHER, behind the barrier)
                if (!ok) {
                    if (lane == 0) atomicMin(info, -1);
                    s_ok = 0;
                }
                // T' = B - sums -> LDS as [row][q] (every wave reads the rows up to its own)
#pragma unroll
                for (int v = 0; v < 4; ++v) Tx[(16 * wave + 4 * v + kq) * 17 + l15] = tvo[v] - (a0[v] + a1[v]);
                __syncthreads();
                if (!s_ok) return;
                double4_s out = {0, 0, 0, 0};
                for (int st = 0; st < 4 * wave + 4; ++st)     // (inv(L_jj)[row][k] = 0 for k > row)
                    out = __builtin_amdgcn_mfma_f64_16x16x4f64(Dg[(4 * st + kq) * kLdDiag + 16 * wave + l15], Tx[(4 * st + kq) * 17 + l15], out, 0, 0, 0);
                if (jb > 0) {
                    double mv[16];
#pragma unroll
                    for (int st = 0; st < 16; ++st) mv[st] = Ms[(4 * st + kq) * kLdDiag + 16 * wave + l15];
                    double bv[16];
                    take_x(jb - 1, false, bv);
                    if (!ok) {
                        if (lane == 0) atomicMin(info, -1);
                        s_ok = 0;
                    }
                    double4_s m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0};
#pragma unroll
                    for (int st = 0; st < 16; st += 2) {
                        m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(mv[st], bv[st], m0, 0, 0, 0);
                        m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mv[st + 1], bv[st + 1], m1, 0, 0, 0);
                    }
                    out -= m0 + m1;
                }
                // (this block's x stays armed after a time-out: its waiters see the status word)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = 16 * wave + kq + 4 * v;
                    if (c < wbk && qok && ok)
                        __hip_atomic_store(&xscratch[qoff + (cb + c) * sr], unarmed(out[v]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = 16 * wave + kq + 4 * v;
                    if (c < wbk && qok && ok) x[qoff + (cb + c) * sr] = out[v];
                }
                __syncthreads();   // (Tx is free for the next pass)
                if (!s_ok) return;
                continue;
            } else {
            if (16 * wave >= nq) continue;             // (this wave's 16 right-hand sides are not in the pass)
            const bool qok = 16 * wave + l15 < nq;
            const int64_t qoff = (q0 + min(16 * wave + l15, nq - 1)) * sq + D.c0 * sr;   // this lane's right-hand side, row c0
            const double* __restrict__ xq = xscratch + qoff;      // ... in the armed buffer
            // B_jb (with every contribution of the levels below): rows 4 st + kq
            double tv[16];
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int c = 4 * st + kq;
                const double v = x[qoff + (cb + min(c, wbk - 1)) * sr];
                tv[st] = (qok && c < wbk) ? v : 0.0;
            }
            double4_s acc[4];   // [16 rows rg]: lane (q = l15, row = kq + 4 v)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) acc[rg] = double4_s{0, 0, 0, 0};
            bool ok = true;
            // X_k (this wave's 16 right-hand sides, rows 4 st + kq) as soon as it is there; lazy: a block that is not the
            // one right before this one first watches ONE value, then goes on to the full poll (normally satisfied at once)
            auto take_x = [&](int k, bool lazy, double (&bv)[16]) __attribute__((always_inline)) {
                const unsigned long long t0 = wall_clock64();
                int spins = 0;
                if (lazy) {
                    const long long* __restrict__ watch = reinterpret_cast<const long long*>(xq + (k * kTile + 63) * sr);
                    while (__hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kXArmed || wait_bias != 0) {
                        if (give_up(t0, spins)) {
                            ok = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(48);
                    }
                }
                while (ok) {
                    bool in = true;
#pragma unroll
                    for (int st = 0; st < 16; ++st) {
                        const long long b = __hip_atomic_load(reinterpret_cast<const long long*>(xq + (k * kTile + 4 * st + kq) * sr),
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        in = in && b != kXArmed;
                        bv[st] = qok ? __longlong_as_double(b) : 0.0;
                    }
                    if (__all((in || !qok) && wait_bias == 0)) break;
                    if (give_up(t0, spins)) ok = false;
                    else __builtin_amdgcn_s_sleep(1);
                }
            };
            for (int k = 0; k < jb - 1 && ok; ++k) {
                // L(jb, k): columns 64 k + 4 st + kq, issued before the wait for X_k
                double av[16][4];
#pragma unroll
                for (int st = 0; st < 16; ++st) {
                    const int64_t col = (int64_t)(k * kTile + 4 * st + kq) * r;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const double a = arow[rg][col];
                        av[st][rg] = aok[rg] ? a : 0.0;
                    }
                }
                double bv[16];
                take_x(k, true, bv);
                if (!ok) break;
#pragma unroll
                for (int st = 0; st < 16; ++st)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        acc[rg] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st][rg], bv[st], acc[rg], 0, 0, 0);
            }
            if (!ok) {   // a hand-off timed out: reported; this block's x stays armed, its waiters give up on the status word
                if (lane == 0) atomicMin(info, -1);
                return;
            }
            // T' = B - sum: k step st = 4 rg + v of the product with the inverse block
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                for (int v = 0; v < 4; ++v) tv[4 * rg + v] -= acc[rg][v];
            // P = inv(L_jj) T' (row group rg needs k <= 16 rg + 15) -- X_jb itself for the first block column
            double4_s out[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) out[rg] = double4_s{0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 16; ++st)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    if (st < 4 * rg + 4)
                        out[rg] = __builtin_amdgcn_mfma_f64_16x16x4f64(Dg[(4 * st + kq) * kLdDiag + 16 * rg + l15], tv[st], out[rg], 0, 0, 0);
            if (jb > 0) {
                // M as A operand (rows 16 rg + l15, columns 4 st + kq), in registers before the wait
                double mv[16][4];
#pragma unroll
                for (int st = 0; st < 16; ++st)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) mv[st][rg] = Ms[(4 * st + kq) * kLdDiag + 16 * rg + l15];
                double bv[16];
                take_x(jb - 1, false, bv);
                if (!ok) {
                    if (lane == 0) atomicMin(info, -1);
                    return;
                }
                BLK_STAMP(0);
                double4_s m0[4], m1[4];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) m0[rg] = m1[rg] = double4_s{0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 16; st += 2)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        m0[rg] = __builtin_amdgcn_mfma_f64_16x16x4f64(mv[st][rg], bv[st], m0[rg], 0, 0, 0);
                        m1[rg] = __builtin_amdgcn_mfma_f64_16x16x4f64(mv[st + 1][rg], bv[st + 1], m1[rg], 0, 0, 0);
                    }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) out[rg] -= m0[rg] + m1[rg];
            }
#ifdef PARSY_BLKSTAMPS
            asm volatile("" ::"v"(out[3][3]));
            BLK_STAMP(1);
#endif
            // straight from the accumulators to x and the armed buffer (lane (q = l15, row = kq + 4 v))
            // (the armed buffer first: it is what the next block column polls)
            BLK_STAMP(2);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = 16 * rg + kq + 4 * v;
                    if (c < wbk && qok)
                        __hip_atomic_store(&xscratch[qoff + (cb + c) * sr], unarmed(out[rg][v]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            BLK_STAMP(3);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = 16 * rg + kq + 4 * v;
                    if (c < wbk && qok) x[qoff + (cb + c) * sr] = out[rg][v];
                }
            }
        }
        return;
    }

    // ================= a 128-row chunk of the rows below the supernode's columns =================
    // wave = 32 rows; its rows of L against block column jb + 1 are loaded while block column jb is multiplied (two
    // register sets: with one set and 64 rows per wave the loads had only the hand-off to hide behind, and the one wave
    // per SIMD that fits waited for them: 16 TFLOP/s on the Flan-class input)
    const int row0 = pd.row0;
    const int wrow0 = row0 + (kSolveRowsM / 4) * wave;   // first panel row of this wave
    int prow[2];                                         // this lane's row of each 16-row fragment (-1: past the panel)
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) prow[rf] = (wrow0 + 16 * rf + l15 < r && 16 * rf + (kSolveRowsM / 4) * wave < kSolveRowsM)
                                                   ? wrow0 + 16 * rf + l15 : -1;
    const bool wave_on = wrow0 < r;
    auto load_l = [&](int jb, double (&lv)[2][16]) {
        const int cb = jb * kTile, wbk = min(kTile, w - cb);
        if (!wave_on || jb >= nbc) return;
#pragma unroll
        for (int rf = 0; rf < 2; ++rf)
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int c = 4 * st + kq;
                const bool okl = prow[rf] >= 0 && c < wbk;
                const double v = G[(int64_t)(cb + min(c, wbk - 1)) * r + max(prow[rf], 0)];
                lv[rf][st] = okl ? v : 0.0;
            }
    };
    for (int pass = plane; pass * kRhsM < nrhs; pass += kPassLanes) {
        const int q0 = pass * kRhsM;
        const int nq = min(kRhsM, nrhs - q0);
        const int nfn = (nq + 15) >> 4;                 // 16-wide fragments of right-hand sides in use
        double4_s acc[4][2];                            // [fragment of right-hand sides][fragment of rows]
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int rf = 0; rf < 2; ++rf) acc[nf][rf] = double4_s{0, 0, 0, 0};
        double lvA[2][16], lvB[2][16];
        load_l(0, lvA);
        // one block column: stage X_jb (armed buffer -> LDS), start the loads of the next one, multiply
        auto step = [&](int jb, double (&cur)[2][16], double (&nxt)[2][16]) -> bool {
            const int cb = jb * kTile, wbk = min(kTile, w - cb);
            __syncthreads();  // ts of the previous block column is free
            {   // thread (c, q) = (tid & 63, (tid >> 6) + 4 u) -- or, X row-major, ((tid >> 6) + 4 u, tid & 63): 16 values each
                const unsigned long long t0 = wall_clock64();
                int spins = 0;
                const int lo = tid & 63, hi = tid >> 6;
                double xv[kRhsM / 4];
                bool ok = true;
                const bool any = tr ? lo < nq : lo < wbk;
                if (any) {
                    // (a chunk far behind the chain first watches one value lazily)
                    const int cw = tr ? min(hi, wbk - 1) : lo, qw = tr ? lo : 0;
                    const long long* __restrict__ watch =
                        reinterpret_cast<const long long*>(xscratch + (D.c0 + cb + cw) * sr + (q0 + qw) * sq);
                    while (__hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kXArmed || wait_bias != 0) {
                        if (give_up(t0, spins)) {
                            ok = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(16);
                    }
                    while (ok) {
                        bool in = true;
#pragma unroll
                        for (int u = 0; u < kRhsM / 4; ++u) {
                            const int c = tr ? hi + 4 * u : lo, q = tr ? lo : hi + 4 * u;
                            long long b = 0;
                            if (q < nq && c < wbk)
                                b = __hip_atomic_load(reinterpret_cast<const long long*>(
                                                          xscratch + (D.c0 + cb + c) * sr + (q0 + q) * sq),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            in = in && b != kXArmed;
                            xv[u] = __longlong_as_double(b);
                        }
                        if (in) break;
                        if (give_up(t0, spins)) ok = false;
                        else __builtin_amdgcn_s_sleep(1);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < kRhsM / 4; ++u) xv[u] = 0.0;
                }
                if (!ok) {
                    atomicMin(info, -1);
                    s_ok = 0;
                }
#pragma unroll
                for (int u = 0; u < kRhsM / 4; ++u) {
                    const int c = tr ? hi + 4 * u : lo, q = tr ? lo : hi + 4 * u;
                    TSF(c, q) = xv[u];
                }
            }
            __syncthreads();
            if (!s_ok) return false;
            load_l(jb + 1, nxt);
            // accumulator += x_jb' L(rows, jb)'
            if (wave_on) {
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    if (nf < nfn) {
#pragma unroll
                        for (int st = 0; st < 16; ++st) {
                            const double av = TSF(4 * st + kq, 16 * nf + l15);
#pragma unroll
                            for (int rf = 0; rf < 2; ++rf)
                                acc[nf][rf] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, cur[rf][st], acc[nf][rf], 0, 0, 0);
                        }
                    }
                }
            }
            return true;
        };
        for (int jb = 0; jb < nbc; jb += 2) {
            if (!step(jb, lvA, lvB)) return;
            if (jb + 1 < nbc && !step(jb + 1, lvB, lvA)) return;
        }
        // x[row] -= accumulated update
#pragma unroll
        for (int rf = 0; rf < 2; ++rf) {
            if (prow[rf] >= 0) {
                const int xrow = rows[D.pi + prow[rf]];
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int q = 16 * nf + kq + 4 * v;
                        if (q < nq) atomicAdd(&x[xrow * sr + (q0 + q) * sq], -acc[nf][rf][v]);
                    }
            }
        }
    }
}
#undef TSF

void launch_solve_blocks_mrhs(const DevicePattern& P, int first, int count, const double* L, const double* dinv,
                              double* x, double* xscratch, int nrhs, int ldx, int ldq, int ticket, int wait_bias,
                              hipStream_t stream) {
    if (count <= 0) return;
    const int lanes_m = std::min(kPassLanes, (nrhs + kRhsM - 1) / kRhsM);
    if (nrhs <= 16)
        hipLaunchKernelGGL(k_solve_blocks_mrhs<true>, dim3(count * lanes_m), dim3(kThreads), 0, stream, P.sn, P.solve_mtasks + first,
                           P.rows, L, dinv, x, xscratch, nrhs, ldx, ldq, P.sinfo, P.stickets + ticket, wait_bias, count);
    else
        hipLaunchKernelGGL(k_solve_blocks_mrhs<false>, dim3(count * lanes_m), dim3(kThreads), 0, stream, P.sn, P.solve_mtasks + first,
                           P.rows, L, dinv, x, xscratch, nrhs, ldx, ldq, P.sinfo, P.stickets + ticket, wait_bias, count);
}

// X between its two layouts: right-hand-side-major a[q * lda + row] <-> row-major b[row * ldb + q] (64 x 64 tiles
// through LDS: both sides in 512-byte runs).  to_rows: a -> b, otherwise b -> a.
__global__ __launch_bounds__(kThreads) void k_transpose_x(double* __restrict__ a, int64_t lda, double* __restrict__ b,
                                                          int64_t ldb, int n, int nrhs, int to_rows) {
    __shared__ double T[64][65];
    const int tid = threadIdx.x, lo = tid & 63, hi = tid >> 6;
    const int row0 = 64 * (int)blockIdx.x, q0 = 64 * (int)blockIdx.y;
    if (to_rows) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int q = q0 + hi + 4 * u, row = row0 + lo;
            T[hi + 4 * u][lo] = (q < nrhs && row < n) ? a[(int64_t)q * lda + row] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int row = row0 + hi + 4 * u, q = q0 + lo;
            if (row < n && q < nrhs) b[(int64_t)row * ldb + q] = T[lo][hi + 4 * u];
        }
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int row = row0 + hi + 4 * u, q = q0 + lo;
            T[lo][hi + 4 * u] = (row < n && q < nrhs) ? b[(int64_t)row * ldb + q] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int q = q0 + hi + 4 * u, row = row0 + lo;
            if (q < nrhs && row < n) a[(int64_t)q * lda + row] = T[hi + 4 * u][lo];
        }
    }
}
void launch_transpose_x(double* x, int64_t ldx, double* xt, int64_t ldq, int n, int nrhs, bool to_rows, hipStream_t stream) {
    if (n <= 0 || nrhs <= 0) return;
    hipLaunchKernelGGL(k_transpose_x, dim3((n + 63) / 64, (nrhs + 63) / 64), dim3(kThreads), 0, stream, x, ldx, xt, ldq, n,
                       nrhs, to_rows ? 1 : 0);
}

// the many-right-hand-side kernels start at this many right-hand sides (the executor arms the hand-off buffer for them)
int solve_mrhs_min() { return chain_mrhs_min(); }
int solve_small_mrhs_min() { return mrhs_min(); }

// One right-hand side: the chain launches hand x over through xscratch itself (k_solve_chain_w); every entry must
// hold the armed pattern when the solve starts.
hipError_t solve_arm_handoff(double* xscratch, int64_t n, hipStream_t stream) {
    return hipMemsetD32Async((hipDeviceptr_t)xscratch, (int)kXArmedWord, (size_t)n * 2, stream);
}

// Start of a solve with several right-hand sides, ONE launch instead of three memsets and a fill of the whole hand-off
// buffer: the solve's status word and the ticket counters of its chain launches are zeroed (block 0), and the hand-off
// buffer is armed where it is polled -- the columns of the wide supernodes (one workgroup per (supernode, block column) of
// solve_wide_list: a chain launch waits only for x of block columns of its own supernode).  The fill of all n x nrhs
// entries took 39 us on the parabolic_fem-class input (269 MB at 64 right-hand sides) and the three memsets 20 us more
// with the gaps between them: 0.13 of a 1.19-ms solve.
__global__ __launch_bounds__(kThreads) void k_solve_arm_wide(const SnDesc* __restrict__ sn, const int32_t* __restrict__ wide,
                                                             int npairs, double* __restrict__ xscratch, int nrhs, int64_t sr,
                                                             int64_t sq, int* __restrict__ sinfo, int* __restrict__ stickets,
                                                             int ntickets) {
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        if (tid == 0) *sinfo = 0;
        for (int t = tid; t < ntickets; t += kThreads) stickets[t] = 0;
    }
    if ((int)blockIdx.x >= npairs) return;
    const SnDesc D = sn[wide[2 * blockIdx.x]];
    const int cb = kTile * wide[2 * blockIdx.x + 1], wb = min(kTile, D.w - cb);
    long long* __restrict__ xs = reinterpret_cast<long long*>(xscratch);
    // (lanes along the contiguous index of the layout: columns when right-hand-side-major, right-hand sides when row-major)
    if (sr == 1) {
        for (int e = tid; e < wb * nrhs; e += kThreads) {
            const int q = e / wb, c = e - q * wb;
            xs[(int64_t)(D.c0 + cb + c) + q * sq] = kXArmed;
        }
    } else {
        for (int e = tid; e < wb * nrhs; e += kThreads) {
            const int c = e / nrhs, q = e - c * nrhs;
            xs[(int64_t)(D.c0 + cb + c) * sr + q * sq] = kXArmed;
        }
    }
}

void launch_solve_arm_wide(const DevicePattern& P, int npairs, double* xscratch, int nrhs, int ldx, int ldq, int ntickets,
                           hipStream_t stream) {
    const bool tr = ldq > 0;
    hipLaunchKernelGGL(k_solve_arm_wide, dim3(std::max(npairs, 1)), dim3(kThreads), 0, stream, P.sn, P.solve_wide_list, npairs,
                       xscratch, nrhs, (int64_t)(tr ? ldq : 1), (int64_t)(tr ? 1 : ldx), P.sinfo, P.stickets, ntickets);
}

void launch_solve_chain(const DevicePattern& P, int first, int count, const double* L, const double* dinv,
                        double* x, double* xscratch, int nrhs, int ldx, int epoch0, int ticket, int wait_bias,
                        hipStream_t stream) {
    if (count <= 0) return;
    if (nrhs >= chain_mrhs_min()) {
        const int lanes_m = std::min(kPassLanes, (nrhs + kRhsM - 1) / kRhsM);
        hipLaunchKernelGGL(k_solve_chain_mrhs, dim3(count * lanes_m), dim3(kThreads), 0, stream, P.sn,
                           P.solve_panels + first, P.rows, L, dinv, x, xscratch, nrhs, ldx, P.flags, epoch0, P.sinfo,
                           P.stickets + ticket, wait_bias, count, P.flag_stride);
        return;
    }
    const int nq = nrhs == 1 ? 1 : kRhs;
    const int lanes = std::min(kPassLanes, (nrhs + nq - 1) / nq);
    if (nrhs == 1) {   // (xscratch was armed by the caller: solve_arm_handoff)
        if (count <= kChainFewChunks)
            hipLaunchKernelGGL(k_solve_chain_w<2>, dim3(2 * count), dim3(kChainThreads), 0, stream, P.sn,
                               P.solve_panels + first, P.rows, L, dinv, x, xscratch, P.sinfo, P.stickets + ticket,
                               wait_bias);
        else
            hipLaunchKernelGGL(k_solve_chain_w<4>, dim3(count), dim3(kChainThreads), 0, stream, P.sn,
                               P.solve_panels + first, P.rows, L, dinv, x, xscratch, P.sinfo, P.stickets + ticket,
                               wait_bias);
    } else
        hipLaunchKernelGGL(k_solve_chain<kRhs>, dim3(count * lanes), dim3(kThreads), 0, stream, P.sn,
                           P.solve_panels + first, P.rows, L, dinv, x, xscratch, nrhs, ldx, P.flags, epoch0,
                           P.sinfo, P.stickets + ticket, wait_bias, count, P.flag_stride);
}

// ---------------------------------------------------------------------------
// ONE-launch solves of small plans (Schedule::solve_one): a job of a few thousand supernodes is ten to twenty level
// launches of a few microseconds of work each -- the ex15-class forward solve took 0.126 ms, slower than one CPU
// thread (0.0745 ms), all of it launch latency and hand-offs through memory.  Here a solve is ONE enqueue: one
// workgroup per block column, taken by ticket in level order (a workgroup only ever waits for blocks with smaller
// tickets, which are held by workgroups that run or have finished: no deadlock at any residency), and instead of
// level barriers every value is handed over as the data itself: the hand-off buffer holds the armed pattern when the
// solve starts and a value is valid once it differs (8-byte agent-scope atomics both sides, as the chain launches'
// xscratch) -- ONE memory round trip per step of the critical path.
// Forward: the tasks are BLOCK COLUMNS (<= 64 columns of a supernode: Schedule::one_sn).  What block p subtracts from
// the x of row k below its columns -- a later column of its supernode or a row of an ancestor --, c = L[k, cols] y_p,
// is not added to x but written to a slot of its own (written once); the block that owns row k gathers its slots
// (one_pull_*: all of them polled at once, one per thread), solves its diagonal block, publishes x and then its own
// c's -- from its panel, which it staged in LDS BEFORE it waited (L does not depend on anybody).
// The same sums as the level launches, in gather order (the reference's `omp atomic` order is schedule-dependent
// as well: triangularSolve/Triangular_BCSC.h:139-157).  Earlier forms, all measured on the ex15-class input:
// atomics on x + a dependency counter per supernode (three round trips per step) 0.098 ms; a pull through the
// factorization's update lists (the top separators then stream every row inside their columns themselves, after
// their last descendant) 0.175 ms; whole supernodes as tasks (the 124-column root then spends 24 us on its two block
// columns, its second diagonal block and the rows between them fetched on the critical path) 0.093 ms.
// No memset either: a solve arms the OTHER buffer of its kind (its supernode's slots / columns) and zeroes the other
// {status, ticket} pair for the solve after it; the executor alternates.  Every wait is bounded like the chain
// launches' (2 s, status -1, nobody hangs).
__device__ __forceinline__ double one_poll(const double* p, int* info, int wait_bias, bool& ok) {
    const long long* src = reinterpret_cast<const long long*>(p);
    const unsigned long long t0 = wall_clock64();
    int spins = 0;
    for (;;) {
        const long long b = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b != kXArmed && wait_bias == 0) return __longlong_as_double(b);
        if ((++spins & 15) == 0 && (wall_clock64() - t0 > kSolveSpinTicks ||
                                    __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
            atomicMin(info, -1);   // (the result is wrong and reported)
            ok = false;
            return 0.0;
        }
        if (spins < 8) __builtin_amdgcn_s_sleep(1);
        else __builtin_amdgcn_s_sleep(8);   // (the late ones poll at leisure)
    }
}
// the values of NQ right-hand sides of one slot: the loads go out together, the polls only where a value is not there yet
// (one after the other they were NQ round trips per slot)
template <int NQ>
__device__ __forceinline__ void one_poll_n(const double* p, int64_t stride, int nrhs, double (&v)[NQ], int* info, int wait_bias,
                                           bool& ok) {
    long long b[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        b[q] = q < nrhs ? __hip_atomic_load(reinterpret_cast<const long long*>(p + (int64_t)q * stride), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT)
                        : 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        v[q] = q >= nrhs ? 0.0 : (b[q] == kXArmed || wait_bias != 0) ? one_poll(p + (int64_t)q * stride, info, wait_bias, ok)
                                                                     : __longlong_as_double(b[q]);
}
// what every workgroup of a ONE-launch solve does first: its part (entries [e0, e0 + len) per right-hand side, stride
// ldy) of the NEXT solve's buffer is armed, and the first ticket zeroes the next solve's {status, ticket}
__device__ __forceinline__ void one_arm_next(double* __restrict__ y_next, int* __restrict__ state_next, int64_t ldy, int64_t e0,
                                             int len, int task, int tid, int cap) {
    for (int e = tid; e < len * cap; e += kThreads) {   // (cap: right-hand sides the buffers are made for)
        const int q = e / len, c = e - q * len;
        reinterpret_cast<long long*>(y_next)[(int64_t)q * ldy + e0 + c] = kXArmed;
    }
    if (task == 0 && tid < 2) state_next[tid] = 0;
}

#ifdef PARSY_ONESTAMPS
// diagnostic build: per supernode the 100 MHz wall clock at five points of k_solve_one (tools/one_stamps.py)
__device__ unsigned long long g_onestamp[4096 * 8];
#define ONESTAMP(i) do { if (tid == 0 && t < 4096) g_onestamp[t * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ONESTAMP(i) do { } while (0)
#endif
static constexpr int kOneRows = 256;     // rows below a block whose x the backward solve keeps in LDS
// doubles of a block's rows below its columns staged in LDS (a larger panel waits in registers); forward, by right-hand
// sides carried: what leaves room for TWO workgroups per compute unit beside the other arrays (80 KB each)
template <int NQ> struct OneStage { static constexpr int fwd = NQ <= 4 ? 4096 : 3584, bwd = 4096; };
// The inverse M of a diagonal block (<= 64 x 64, column-major in Dg, identity padded to wpad), in place.  First the 16 x 16
// diagonal sub-blocks (block_invert16), copied dense into s_m; then, block row by block row from the bottom up and inside
// a row from right to left,
//     M_ij = -(sum_{k = j + 1}^{i} M_ik L_kj) M_jj
// which overwrites L_ij (no later block reads it: rows above use only rows <= their own of L).  One thread per entry;
// ends synchronised.  Afterwards M[c][k], k <= c, is s_m[..] inside c's 16-block and Dg[k * kLdDiag + c] left of it.
__device__ __forceinline__ void one_invert_block(double* Dg, double* invd, double* s_t, double (*s_m)[16 * 17], int w, int wpad,
                                                 int tid) {
    block_invert16(Dg, invd, w, tid);   // (ends synchronised)
    for (int e = tid; e < (wpad / 16) * 256; e += kThreads) {
        const int b = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
        s_m[b][rr * 17 + cc] = rr > cc ? Dg[(16 * b + rr) * kLdDiag + 16 * b + cc] : (rr == cc ? invd[16 * b + rr] : 0.0);
    }
    __syncthreads();
    {
        const int rr = tid >> 4, cc = tid & 15;
        for (int bi = wpad - 16; bi >= 16; bi -= 16)
            for (int bj = bi - 16; bj >= 0; bj -= 16) {
                // T = sum_k M_ik L_kj, k from j + 1 to i (M_ii from the dense copy, the others already in place)
                double tv = 0.0;
                for (int bk = bj + 16; bk < bi; bk += 16) {
#pragma unroll
                    for (int m = 0; m < 16; ++m)
                        tv = fma(Dg[(bk + m) * kLdDiag + bi + rr], Dg[(bj + cc) * kLdDiag + bk + m], tv);
                }
#pragma unroll
                for (int m = 0; m < 16; ++m) tv = fma(s_m[bi >> 4][rr * 17 + m], Dg[(bj + cc) * kLdDiag + bi + m], tv);
                s_t[rr * 17 + cc] = tv;
                __syncthreads();   // (T complete, and every read of L_ij is done: it is overwritten next)
                double mv = 0.0;
#pragma unroll
                for (int m = 0; m < 16; ++m) mv = fma(s_t[rr * 17 + m], s_m[bj >> 4][m * 17 + cc], mv);
                Dg[(bj + cc) * kLdDiag + bi + rr] = -mv;
                __syncthreads();
            }
    }
}

// NQ: right-hand sides carried (1, 4 or 8).  On the critical path -- between the last slot's arrival and the block's
// own slots going out -- there are two LDS matrix-vector products and nothing else: the INVERSE of the diagonal block
// is formed while the block waits (it had 6 - 35 us of slack on the ex15-class chain; the blocked substitution of
// k_solve_small took 1.1 - 4.8 us per step of the chain instead).
template <int NQ>
__global__ __launch_bounds__(kThreads) void k_solve_one(const SnDesc* __restrict__ blocks, const int64_t* __restrict__ slot0,
                                                        const int32_t* __restrict__ pull_ptr,
                                                        const int32_t* __restrict__ pull_slot,
                                                        const int32_t* __restrict__ pull_pos, const double* __restrict__ L,
                                                        double* __restrict__ x, int nrhs, int ldx, int64_t nslots,
                                                        double* __restrict__ y, double* __restrict__ y_next,
                                                        int* __restrict__ state, int* __restrict__ state_next,
                                                        int wait_bias, int cap) {
    __shared__ double Dg[kTile * kLdDiag];  // the diagonal block (column-major), then its inverse in place (see below);
                                            // at the end: the parts of the products below
    __shared__ double invd[kTile];
    __shared__ double s_t[16 * 17];         // a 16 x 16 block of the inverse under way
    __shared__ double s_m[kTile / 16][16 * 17];   // the inverses of the 16 x 16 diagonal sub-blocks, dense (zero above the diagonal)
    __shared__ double s_b[kTile][NQ];       // the block's right-hand side minus what the gathered slots say
    __shared__ double s_x[kTile][NQ];       // its solution
    __shared__ double s_pan[OneStage<NQ>::fwd];     // the rows below the block's columns, [c][k - w] (row stride nb | 1)
    __shared__ int s_task;
    static_assert(4 * kTile * 8 <= kTile * kLdDiag, "the products' parts reuse Dg");
    const int tid = threadIdx.x;
    if (tid == 0) s_task = atomicAdd(&state[1], 1);
    __syncthreads();
    const int t = s_task;                   // (the blocks are listed in ticket order)
    const SnDesc D = blocks[t];
    const int w = D.w, nb = D.r - w, ldp = nb | 1, ld = D.ld;
    const double* __restrict__ G = L + D.px;
    const int64_t s0 = slot0[t];
    ONESTAMP(0);
    one_arm_next(y_next, state_next, nslots, s0, nb, t, tid, cap);
    // ---- what does not depend on the other blocks: the right-hand side, the panel below the columns (small: LDS), the
    // diagonal block and its inverse
    for (int e = tid; e < kTile * NQ; e += kThreads) {
        const int q = e / kTile, c = e - q * kTile;
        s_b[c][q] = (c < w && q < nrhs) ? x[(int64_t)q * ldx + D.c0 + c] : 0.0;
    }
    const bool staged = w * ldp <= OneStage<NQ>::fwd;
    if (staged)
        for (int e = tid; e < w * nb; e += kThreads) {
            const int c = e / nb, k = e - c * nb;
            s_pan[c * ldp + k] = G[(int64_t)c * ld + w + k];
        }
    // (a larger panel: the first 256 rows below, one per thread, wait in REGISTERS -- the rows after them are streamed at the end)
    double pre[kTile];
    if (!staged && nb > 0) {
#pragma unroll
        for (int c = 0; c < kTile; ++c) pre[c] = G[(int64_t)min(c, w - 1) * ld + w + min(tid, nb - 1)];
    }
    const int wpad = (w + 15) & ~15;
    for (int e = tid; e < wpad * wpad; e += kThreads) {
        const int c = e / wpad, i = e - c * wpad;
        double v = (i == c) ? 1.0 : 0.0;
        if (i >= c && i < w && c < w) v = G[(int64_t)c * ld + i];
        Dg[c * kLdDiag + i] = v;
    }
    __syncthreads();
    one_invert_block(Dg, invd, s_t, s_m, w, wpad, tid);   // (ends synchronised)
    auto m_at = [&](int c, int k) {   // M[c][k], k <= c
        const int b = c & ~15;
        return k >= b ? s_m[b >> 4][(c - b) * 17 + (k - b)] : Dg[k * kLdDiag + c];
    };
    ONESTAMP(1);
    // ---- gather: one slot per thread and right-hand side, all polls in flight together
    {
        bool ok = true;
        for (int e = pull_ptr[t] + tid; e < pull_ptr[t + 1]; e += kThreads) {
            const int64_t slot = pull_slot[e];
            const int pos = pull_pos[e];
            double v[NQ];
            one_poll_n<NQ>(&y[slot], nslots, nrhs, v, state, wait_bias, ok);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q < nrhs) unsafeAtomicAdd(&s_b[pos][q], -v[q]);
        }
    }
    __syncthreads();
    ONESTAMP(2);
    // ---- x = inv(L_bb) b: four lanes per row share the sum
    {
        const int c = tid >> 2, part = tid & 3;
        double acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
        if (c < w) {
            for (int k = part; k <= c; k += 4) {
                const double iv = m_at(c, k);
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = fma(iv, s_b[k][q], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            acc[q] += __shfl_xor(acc[q], 1);
            acc[q] += __shfl_xor(acc[q], 2);
        }
        if (part == 0 && c < kTile) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                s_x[c][q] = c < w ? acc[q] : 0.0;
                if (c < w && q < nrhs) x[(int64_t)q * ldx + D.c0 + c] = acc[q];
            }
        }
    }
    __syncthreads();
    ONESTAMP(3);
    auto publish = [&](int k, int q, double v) {
        __hip_atomic_store(&y[(int64_t)q * nslots + s0 + k], unarmed(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // ---- what this block subtracts from the rows below its columns: published, one slot per row.  Up to 256 staged
    // rows: the columns are dealt over P = 4 / 2 / 1 groups of threads, the parts added up through LDS
    const int nbp = (nb + 63) & ~63;
    if (staged && nbp <= kThreads && nb > 0) {
        const int P = kThreads / nbp;                 // 4, 2 or 1 (nbp = 64, 128, 192 / 256)
        const int part = tid / nbp, k = tid - part * nbp;
        double* __restrict__ red = Dg;                // [part][q][k]: the inverse is not needed any more
        if (part < P) {
            double acc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
            if (k < nb) {
                const int c_lo = part * w / P, c_hi = (part + 1) * w / P;
                for (int c = c_lo; c < c_hi; ++c) {
                    const double lv = s_pan[c * ldp + k];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q] = fma(lv, s_x[c][q], acc[q]);
                }
            }
            if (P > 1) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) red[(part * NQ + q) * nbp + k] = acc[q];
            } else if (k < nb) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if (q < nrhs)
                        __hip_atomic_store(&y[(int64_t)q * nslots + s0 + k], unarmed(acc[q]), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (P > 1) {
            __syncthreads();
            for (int e = tid; e < nb * NQ; e += kThreads) {
                const int q = e / nb, kk = e - q * nb;
                double v = 0.0;
                for (int pp = 0; pp < P; ++pp) v += red[(pp * NQ + q) * nbp + kk];
                if (q < nrhs)
                    __hip_atomic_store(&y[(int64_t)q * nslots + s0 + kk], unarmed(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if (staged) {
        for (int k = tid; k < nb; k += kThreads) {
            double acc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
            for (int c = 0; c < w; ++c) {
                const double lv = s_pan[c * ldp + k];
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = fma(lv, s_x[c][q], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q < nrhs) publish(k, q, acc[q]);
        }
    } else {
        // a larger panel: the first 256 rows from the registers they have been waiting in ...
        if (tid < nb) {
            double acc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
#pragma unroll
            for (int c = 0; c < kTile; ++c)
                if (c < w) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q] = fma(pre[c], s_x[c][q], acc[q]);
                }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q < nrhs) publish(tid, q, acc[q]);
        }
        // ... the rows after them in chunks of 128, two register sets (as the backward kernel's far rows): wave v takes the
        // columns v, v + 4, ..., lane l the rows l and l + 64 of a chunk and sums its 16 columns in place; the four waves'
        // parts meet in LDS (the inverse's place) and one thread per row publishes.  The loads of the chunk after next go
        // out as soon as a set is multiplied.
        if (nb > kThreads) {
            constexpr int kChunk = 128, kPc = kTile / (kThreads / 64), kPr = kChunk / 64;
            constexpr int kHalves = NQ <= 4 ? 2 : 1;   // (8 right-hand sides: one buffer of parts, two barriers per chunk)
            static_assert(kHalves * (kThreads / 64) * kChunk * NQ <= kTile * kLdDiag, "the waves' parts reuse Dg");
            const int wave = tid >> 6, lane = tid & 63;
            const int nchunks = (nb - kThreads + kChunk - 1) / kChunk;
            double fs[2][kPc][kPr];
            auto fetch = [&](double (&dst)[kPc][kPr], int ch) {
                const int k0 = kThreads + ch * kChunk;
#pragma unroll
                for (int ci = 0; ci < kPc; ++ci) {
                    const double* __restrict__ col = G + (int64_t)min(wave + (kThreads / 64) * ci, w - 1) * ld + w;
#pragma unroll
                    for (int jr = 0; jr < kPr; ++jr) dst[ci][jr] = col[min(k0 + lane + 64 * jr, nb - 1)];
                }
            };
            fetch(fs[0], 0);
            if (nchunks > 1) fetch(fs[1], 1);
            double* __restrict__ part = Dg;   // [half][wave][row of the chunk][q]
            auto step = [&](double (&cur)[kPc][kPr], int ch) {
                const int k0 = kThreads + ch * kChunk, len = min(kChunk, nb - k0);
                const int half = (kHalves == 2 ? (ch & 1) : 0) * ((kThreads / 64) * kChunk * NQ);
                double pr[kPr][NQ];
#pragma unroll
                for (int jr = 0; jr < kPr; ++jr)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) pr[jr][q] = 0.0;
#pragma unroll
                for (int ci = 0; ci < kPc; ++ci) {
                    const int c = wave + (kThreads / 64) * ci;
                    if (c < w) {
#pragma unroll
                        for (int jr = 0; jr < kPr; ++jr)
#pragma unroll
                            for (int q = 0; q < NQ; ++q) pr[jr][q] = fma(cur[ci][jr], s_x[c][q], pr[jr][q]);
                    }
                }
                if (ch + 2 < nchunks) fetch(cur, ch + 2);
                if (kHalves == 1) __syncthreads();   // (the parts of the chunk before are read)
#pragma unroll
                for (int jr = 0; jr < kPr; ++jr)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) part[half + ((wave * kChunk) + lane + 64 * jr) * NQ + q] = pr[jr][q];
                __syncthreads();
                if (tid < len) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        double v = 0.0;
#pragma unroll
                        for (int wv = 0; wv < kThreads / 64; ++wv) v += part[half + (wv * kChunk + tid) * NQ + q];
                        if (q < nrhs) publish(k0 + tid, q, v);
                    }
                }
            };
            for (int ch = 0; ch < nchunks; ch += 2) {
                step(fs[0], ch);
                if (ch + 1 < nchunks) step(fs[1], ch + 1);
            }
        }
    }
    ONESTAMP(4);
}

// The backward counterpart: block column p, taken by ticket from the root down (the forward order reversed), needs the x
// of every row below its columns -- the later columns of its supernode and rows of ancestors: all of them owned by
// blocks with earlier tickets --, written once each: x itself is the hand-off (y: n values per right-hand side, armed).
//     t = y_blk - L(below, blk)' x(below),   x_blk = inv(L_bb)' t
// Before it polls, the block has its panel below in LDS and the inverse of its diagonal block (as k_solve_one); after the
// last x arrives there are two LDS matrix-vector products.  The first version ran k_bsolve_block's loops (blocked
// substitution, rows below streamed from memory after the wait): ex15-class 0.086 ms.
template <int NQ>
__global__ __launch_bounds__(kThreads) void k_bsolve_one(const SnDesc* __restrict__ blocks, const int32_t* __restrict__ rows,
                                                         const int32_t* __restrict__ blk_w0, const double* __restrict__ L,
                                                         double* __restrict__ x, int nrhs, int ldx, int n, int nblocks,
                                                         double* y, double* __restrict__ y_next,
                                                         int* __restrict__ state, int* __restrict__ state_next,
                                                         int wait_bias, int cap) {
    // (y: other workgroups publish into it while this one reads: no __restrict__, every read of it is an atomic load)
    __shared__ double Dg[kTile * kLdDiag];  // the diagonal block (column-major), then its inverse in place; at the end: parts
    __shared__ double invd[kTile];
    __shared__ double s_t[16 * 17];
    __shared__ double s_m[kTile / 16][16 * 17];
    __shared__ double s_b[kTile][NQ];       // y_blk, then t
    __shared__ double s_xb[kOneRows][NQ];   // x of the rows below
    __shared__ double s_pan[OneStage<NQ>::bwd + kTile];   // the rows below the block's columns, [c][k - w] (row stride nb | 1; a larger
                                                  // panel: its first 64 rows, stride 65)
    __shared__ int s_task;
    static_assert(4 * kTile * 8 <= kTile * kLdDiag, "the products' parts reuse Dg");
    const int tid = threadIdx.x;
    if (tid == 0) s_task = atomicAdd(&state[1], 1);
    __syncthreads();
    const int t = nblocks - 1 - s_task;     // (the forward order reversed)
    const SnDesc D = blocks[t];
    const int w = D.w, nb = D.r - w, ldp = nb | 1, ld = D.ld;
    const double* __restrict__ G = L + D.px;
    // row k >= w of the block's window: a later column of its supernode (the first w0 - w rows below) or lR
    const int w_left = blk_w0[t];           // columns of the supernode from this block's first one on
    const int32_t* __restrict__ ri = rows + D.pi;
    one_arm_next(y_next, state_next, n, D.c0, w, s_task, tid, cap);
    for (int e = tid; e < kTile * NQ; e += kThreads) {
        const int q = e / kTile, c = e - q * kTile;
        s_b[c][q] = (c < w && q < nrhs) ? x[(int64_t)q * ldx + D.c0 + c] : 0.0;
    }
    const bool staged = nb <= kOneRows && w * ldp <= OneStage<NQ>::bwd;   // the panel below and the x of its rows fit in LDS
    if (staged)
        for (int e = tid; e < w * nb; e += kThreads) {
            const int c = e / nb, k = e - c * nb;
            s_pan[c * ldp + k] = G[(int64_t)c * ld + w + k];
        }
    // A larger panel: its first 64 rows -- in a wide supernode the next block column, whose x arrives last -- in LDS all the
    // same; the rows after them in chunks of kOneRows, the farthest first (their x has been there for long): wave v takes
    // the columns v, v + 4, ..., lane l the rows l, l + 64, ... of a chunk -- 64 values of L per lane, in REGISTERS; the
    // first chunk's are asked for here, before the polls
    const int near = staged ? nb : min(nb, kTile), ldn = kTile + 1;
    if (!staged)
        for (int e = tid; e < w * near; e += kThreads) {
            const int c = e / near, k = e - c * near;
            s_pan[c * ldn + k] = G[(int64_t)c * ld + w + k];
        }
    const int wave = tid >> 6, lane = tid & 63;
    // (chunks of kOneRows / 2 rows, two register sets: the loads of the chunk after are in flight while one is multiplied)
    constexpr int kChunk = kOneRows / 2, kPc = kTile / (kThreads / 64), kPr = kChunk / 64;   // columns per wave, rows per lane
    double pre[2][kPc][kPr];
    auto preload = [&](double (&dst)[kPc][kPr], int k0) {
#pragma unroll
        for (int ci = 0; ci < kPc; ++ci) {
            const double* __restrict__ col = G + (int64_t)min(wave + (kThreads / 64) * ci, w - 1) * ld + w;
#pragma unroll
            for (int j = 0; j < kPr; ++j) dst[ci][j] = col[min(k0 + lane + 64 * j, nb - 1)];
        }
    };
    const int nfar = nb - near, nchunks = (nfar + kChunk - 1) / kChunk;   // (0 when staged)
    if (nchunks > 0) preload(pre[0], near + (nchunks - 1) * kChunk);
    const int wpad = (w + 15) & ~15;
    for (int e = tid; e < wpad * wpad; e += kThreads) {
        const int c = e / wpad, i = e - c * wpad;
        double v = (i == c) ? 1.0 : 0.0;
        if (i >= c && i < w && c < w) v = G[(int64_t)c * ld + i];
        Dg[c * kLdDiag + i] = v;
    }
    __syncthreads();
    one_invert_block(Dg, invd, s_t, s_m, w, wpad, tid);   // (ends synchronised)
    auto m_at = [&](int c, int k) {   // M[c][k], k <= c
        const int b = c & ~15;
        return k >= b ? s_m[b >> 4][(c - b) * 17 + (k - b)] : Dg[k * kLdDiag + c];
    };
    bool ok = true;
    // the x of (a chunk of) the rows below: one row per thread and right-hand side, all polls in flight together
    auto gather = [&](int k0, int len) {
        for (int k = tid; k < len; k += kThreads) {
            const int row = w + k0 + k < w_left ? D.c0 + w + k0 + k : ri[w + k0 + k];
            double v[NQ];
            one_poll_n<NQ>(&y[row], n, nrhs, v, state, wait_bias, ok);
#pragma unroll
            for (int q = 0; q < NQ; ++q) s_xb[k][q] = v[q];
        }
    };
    // far chunks, the farthest first; step i uses register set i & 1 and the half i & 1 of s_xb, and asks for the chunk
    // after it -- its part of L and, with a PLAIN load, its x -- before it multiplies (one barrier per chunk: the half
    // written next was read two steps ago).  The plain load is the fast path: far rows were published long ago and the
    // line is in the L2 for every block that reads it; it can only see the armed pattern too early (a line cached before
    // the value was stored: values of earlier solves do not survive the launch boundary), and then the poll takes over.
    double xv[2][NQ];
    double facc[kPc];
#pragma unroll
    for (int ci = 0; ci < kPc; ++ci) facc[ci] = 0.0;
    auto x_ahead = [&](double (&dst)[NQ], int ch) {
        const int k0 = near + ch * kChunk, k = min(k0 + (tid & (kChunk - 1)), nb - 1);
        const int row = w + k < w_left ? D.c0 + w + k : ri[w + k];
#pragma unroll
        for (int q = 0; q < NQ; ++q)   // (a relaxed load of wavefront scope: the same cached load, defined beside the publishers' stores)
            dst[q] = q < nrhs ? __hip_atomic_load(&y[(int64_t)q * n + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0.0;
    };
    if (nchunks > 0) x_ahead(xv[0], nchunks - 1);
    auto far_step = [&](double (&cur)[kPc][kPr], double (&nxt)[kPc][kPr], double (&xc)[NQ], double (&xn)[NQ], int step) {
        const int ch = nchunks - 1 - step, k0 = near + ch * kChunk, len = min(kChunk, nb - k0), half = (step & 1) * kChunk;
        if (ch > 0) {
            preload(nxt, near + (ch - 1) * kChunk);
            x_ahead(xn, ch - 1);
        }
        if (tid < len) {
            const int row = w + k0 + tid < w_left ? D.c0 + w + k0 + tid : ri[w + k0 + tid];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                double v = xc[q];
                if (q < nrhs && __double_as_longlong(v) == kXArmed) v = one_poll(&y[(int64_t)q * n + row], state, wait_bias, ok);
                s_xb[half + tid][q] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < kPc; ++ci) {
            const int c = wave + (kThreads / 64) * ci;
            double acc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
#pragma unroll
            for (int j = 0; j < kPr; ++j) {
                const int k = lane + 64 * j;
                if (k < len) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q] = fma(cur[ci][j], s_xb[half + k][q], acc[q]);
                }
            }
            if (NQ == 1) {
                facc[ci] += acc[0];   // (one right-hand side: the lanes' parts are added up once, after the last chunk)
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) acc[q] += __shfl_xor(acc[q], o);
                }
                if (lane == 0 && c < w) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) s_b[c][q] -= acc[q];   // (column c belongs to this wave alone)
                }
            }
        }
    };
    for (int step = 0; step < nchunks; step += 2) {
        far_step(pre[0], pre[1], xv[0], xv[1], step);
        if (step + 1 < nchunks) far_step(pre[1], pre[0], xv[1], xv[0], step + 1);
    }
    if (NQ == 1 && nchunks > 0) {
#pragma unroll
        for (int ci = 0; ci < kPc; ++ci) {
            double v = facc[ci];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            const int c = wave + (kThreads / 64) * ci;
            if (lane == 0 && c < w) s_b[c][0] -= v;
        }
    }
    if (near > 0) {
        // ---- the rows in LDS (all of a small panel): t = y_blk - panel' x(below), four lanes per column share the sum
        if (nchunks > 0) __syncthreads();
        gather(0, near);
        __syncthreads();
        const int c = tid >> 2, part = tid & 3, ldl = staged ? ldp : ldn;
        double acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
        if (c < w)
            for (int k = part; k < near; k += 4) {
                const double lv = s_pan[c * ldl + k];
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = fma(lv, s_xb[k][q], acc[q]);
            }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            acc[q] += __shfl_xor(acc[q], 1);
            acc[q] += __shfl_xor(acc[q], 2);
        }
        if (part == 0 && c < w) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) s_b[c][q] -= acc[q];
        }
    }
    __syncthreads();
    // ---- x = inv(L_bb)' t: x[c] = sum_{k >= c} M[k][c] t[k]; four lanes per column
    {
        const int c = tid >> 2, part = tid & 3;
        double acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
        if (c < w)
            for (int k = c + part; k < w; k += 4) {
                const double iv = m_at(k, c);
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = fma(iv, s_b[k][q], acc[q]);
            }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            acc[q] += __shfl_xor(acc[q], 1);
            acc[q] += __shfl_xor(acc[q], 2);
        }
        if (part == 0 && c < w) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q < nrhs) {
                    __hip_atomic_store(&y[(int64_t)q * n + D.c0 + c], unarmed(acc[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    x[(int64_t)q * ldx + D.c0 + c] = acc[q];
                }
        }
    }
}

void launch_solve_one(const DevicePattern& P, const double* L, double* x, int nrhs, int ldx, double* y, double* y_next,
                      int* state, int* state_next, int wait_bias, int cap, hipStream_t stream) {
    const DevicePattern::OneDev& O = P.one_f;
    if (O.nblocks <= 0) return;
#define PARSY_ONE_LAUNCH(NQ)                                                                                               \
    hipLaunchKernelGGL(k_solve_one<NQ>, dim3(O.nblocks), dim3(kThreads), 0, stream, O.sn, O.slot0, O.pull_ptr, O.pull_slot, \
                       O.pull_pos, L, x, nrhs, ldx, O.nslots, y, y_next, state, state_next, wait_bias, cap)
    if (nrhs == 1) PARSY_ONE_LAUNCH(1);
    else if (nrhs <= 4) PARSY_ONE_LAUNCH(4);
    else PARSY_ONE_LAUNCH(8);
#undef PARSY_ONE_LAUNCH
}

#ifdef PARSY_ONESTAMPS
extern "C" void parsy_debug_onestamps(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_onestamp), sizeof(unsigned long long) * 4096 * 8);
}
#endif

// ---------------------------------------------------------------------------
// Backward solve L' x = y (the reference only has the forward solve;
// with this the library solves A x = b end to end).  Pull form, no atomics: a block of <= 64
// columns [cb, cb+wbk) of a supernode is finished by one workgroup once every row below it
// is final (ancestors' columns: earlier levels; later block columns of the same supernode:
// earlier launches):   t = y_blk - L(below, blk)' x(below),   x_blk = inv(L_bb)' t.
// The product runs one wave per column with lanes along the (contiguous) rows.
// ---------------------------------------------------------------------------
static constexpr int kLdRedB = 65;   // row stride of a wave's reduction buffer (doubles): conflict-free both ways
template <int NQ>
__global__ __launch_bounds__(kThreads) void k_bsolve_block(const SnDesc* __restrict__ sn,
                                                           const PanelDesc* __restrict__ pds,
                                                           const int32_t* __restrict__ rows,
                                                           const double* __restrict__ L,
                                                           double* __restrict__ x, double* __restrict__ xscratch,
                                                           int nrhs, int ldx, int chain, int* __restrict__ info,
                                                           int* __restrict__ ticket, int wait_bias, int nblocks,
                                                           const int32_t* __restrict__ ranges,
                                                           const double* __restrict__ dinv) {
    // chain != 0: every block column of the wide supernodes of a level is in this launch; block jb takes the x
    // of blocks jb+1.. of its supernode as they are published: as the data itself, through the armed buffer
    // (xscratch: 8-byte agent-scope atomics both sides, a value is valid once it differs from kXArmed -- see
    // k_solve_chain_w), and finishes with a product with the inverse diagonal block (DIAG_INVERSE).
    // ranges != null (subtree launch, chain == 0): task b is the run pds[ranges[2b] .. ranges[2b+1]) -- a subtree
    // of single-block supernodes from its root down; the workgroup reads its own earlier x (same CU, plain
    // stores and loads ordered by the barrier between two supernodes).
    __shared__ double Dg[kTile * kLdDiag];
    __shared__ double invd[kTile];
    __shared__ double ts[kTile][NQ];
    __shared__ double s_red[kThreads / 64][(kTile / 4) * kLdRedB];   // per wave: the lanes' parts of its 16 column sums
    __shared__ int s_task;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // chain launch: blocks are listed last block column first (producers first) and taken by ticket
    if (tid == 0) s_task = chain ? atomicAdd(ticket, 1) : (int)(blockIdx.x + blockIdx.y * nblocks);
    __syncthreads();
    // every block once per pass lane: tasks 0..nblocks-1 are lane 0, and so on
    const int plane = s_task / nblocks;
    const int task = s_task - plane * nblocks;
    const int q_begin = ranges ? ranges[2 * task] : task;
    const int q_end = ranges ? ranges[2 * task + 1] : q_begin + 1;
  for (int qsn = q_begin; qsn < q_end; ++qsn) {
    if (qsn > q_begin) __syncthreads();  // the x of the supernode before is stored; Dg / ts are free again
    const PanelDesc pd = pds[qsn];
    const SnDesc D = sn[pd.sn];
    const int r = D.r, w = D.w, cb = pd.jb * kTile, wbk = min(kTile, w - cb);
    const double* __restrict__ G = L + D.px;
    const int32_t* __restrict__ ri = rows + D.pi;
    const int kbeg = cb + wbk;  // first panel row below the block

    {   // diagonal block (identity padded) -> LDS; chain launch: its inverse instead (same layout)
        double dtmp[kTile * kTile / kThreads];
        const double* __restrict__ inv_blk = chain ? dinv + (int64_t)(D.dslot + pd.jb) * (kTile * kTile) : nullptr;
#pragma unroll
        for (int t = 0; t < kTile * kTile / kThreads; ++t) {
            const int e = t * kThreads + tid;
            const int c = e >> 6, i = e & 63;
            double v = (i == c) ? 1.0 : 0.0;
            if (chain) v = inv_blk[e];
            else if (c < wbk && i < wbk && i >= c) v = G[(int64_t)(cb + c) * r + cb + i];
            dtmp[t] = v;
        }
#pragma unroll
        for (int t = 0; t < kTile * kTile / kThreads; ++t) {
            const int e = t * kThreads + tid;
            Dg[(e >> 6) * kLdDiag + (e & 63)] = dtmp[t];
        }
    }
    __syncthreads();
    // inverses of the 16x16 diagonal sub-blocks (stored transposed in the strict upper triangle)
    if (!chain && tid < kTile) invd[tid] = 1.0 / Dg[tid * kLdDiag + tid];
    __syncthreads();
    if (!chain && tid < kTile && (tid & ~15) < wbk) {
        const int b16 = tid & ~15, c = tid & 15;
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
        __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
        for (int rr = 1; rr < 16; ++rr)
            if (rr > c) Dg[(b16 + rr) * kLdDiag + b16 + c] = y[rr];
    }

    const int nbc = (w + kTile - 1) / kTile;
    for (int pass = plane; pass * NQ < nrhs; pass += kPassLanes) {
        const int q0 = pass * NQ;
        const int nq = min(NQ, nrhs - q0);
        __syncthreads();
        for (int e = tid; e < kTile * NQ; e += kThreads) {
            const int c = e & 63, q = e >> 6;
            ts[c][q] = (c < wbk && q < nq) ? x[(int64_t)(q0 + q) * ldx + D.c0 + cb + c] : 0.0;
        }
        __syncthreads();
        // t[c] -= sum_k L[k, cb+c] x[row(k)]: wave `wave` takes columns wave, wave+4, ...
        {
            double acc[kTile / 4][NQ];
#pragma unroll
            for (int ci = 0; ci < kTile / 4; ++ci)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[ci][q] = 0.0;
            // rows below the supernode's own columns (ancestors: final before this launch); without
            // the chain also the rows of the later blocks (solved by earlier launches)
            // (kBackUnroll row chunks per trip, every load of a trip issued before its products: the row ids, the
            // x they point to and the wave's 16 columns of L are three dependent-latency phases otherwise)
            constexpr int kBackUnroll = NQ == 1 ? 4 : 2;
            for (int k0 = (chain ? w : kbeg) + lane; k0 < r; k0 += 64 * kBackUnroll) {
                int xr[kBackUnroll];
                double lv[kBackUnroll][kTile / 4];
#pragma unroll
                for (int u = 0; u < kBackUnroll; ++u) {
                    const int k = min(k0 + 64 * u, r - 1);   // (clamped: the chunks past the panel multiply by zero)
                    xr[u] = (k < w) ? (D.c0 + k) : ri[k];
#pragma unroll
                    for (int ci = 0; ci < kTile / 4; ++ci) {
                        const int c = min(wave + 4 * ci, wbk - 1);
                        lv[u][ci] = G[(int64_t)(cb + c) * r + k];
                    }
                }
#pragma unroll
                for (int u = 0; u < kBackUnroll; ++u) {
                    double xk[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        xk[q] = (q < nq && k0 + 64 * u < r) ? x[(int64_t)(q0 + q) * ldx + xr[u]] : 0.0;
#pragma unroll
                    for (int ci = 0; ci < kTile / 4; ++ci) {
                        const double lvv = (wave + 4 * ci < wbk) ? lv[u][ci] : 0.0;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) acc[ci][q] = fma(lvv, xk[q], acc[ci][q]);
                    }
                }
            }
            if (chain) {
                // the later blocks of this supernode, last one first, each as soon as it is published
                for (int I = nbc - 1; I > pd.jb; --I) {
                    // this lane's row of block I against the wave's 16 columns: loaded BEFORE the wait
                    const int k = I * kTile + lane;
                    double lv[kTile / 4];
#pragma unroll
                    for (int ci = 0; ci < kTile / 4; ++ci) {
                        const int c = wave + 4 * ci;
                        lv[ci] = (c < wbk && k < w) ? G[(int64_t)(cb + c) * r + k] : 0.0;
                    }
                    // x of block I: every lane polls its own row's values (the data is the flag); bounded, and
                    // one timeout (status word) ends every wait of the solve
                    const unsigned long long t0 = wall_clock64();
                    bool ok = true;
                    int spins = 0;
                    double xk[NQ];
                    for (;;) {
                        bool in = true;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            long long b = 0;
                            if (q < nq && k < w)
                                b = __hip_atomic_load(reinterpret_cast<const long long*>(
                                                          &xscratch[(int64_t)(q0 + q) * ldx + D.c0 + k]),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            in = in && b != kXArmed;
                            xk[q] = __longlong_as_double(b);
                        }
                        if (__all(in && wait_bias == 0)) break;
                        if ((++spins & 15) == 0 &&
                            (wall_clock64() - t0 > kSolveSpinTicks ||
                             __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
                            ok = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!ok) {
                        if (lane == 0) atomicMin(info, -1);
                        break;  // (the result is wrong and reported; nobody may hang)
                    }
                    {
#pragma unroll
                        for (int ci = 0; ci < kTile / 4; ++ci)
#pragma unroll
                            for (int q = 0; q < NQ; ++q) acc[ci][q] = fma(lv[ci], xk[q], acc[ci][q]);
                    }
                }
            }
            // column sums across the wave through LDS (as k_bsolve_chain_w: the lanes park their 16 parts, lane
            // (ci, g) adds up 16 of them, two exchanges finish the column), one right-hand side after the other
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                double* __restrict__ red = s_red[wave];
                __builtin_amdgcn_wave_barrier();   // (the parts of the right-hand side before are consumed)
#pragma unroll
                for (int ci = 0; ci < kTile / 4; ++ci) red[ci * kLdRedB + lane] = acc[ci][q];
                __builtin_amdgcn_wave_barrier();
                const int ci = lane & 15, g = lane >> 4;
                double v0 = 0.0, v1 = 0.0;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    v0 += red[ci * kLdRedB + 16 * g + i];
                    v1 += red[ci * kLdRedB + 16 * g + i + 1];
                }
                double v = v0 + v1;
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (g == 0) ts[wave + 4 * ci][q] -= v;  // each (column, q) has one writer
            }
        }
        __syncthreads();
        if (chain) {
            // x_blk = inv(L_bb)' t as a product: x_c = sum_{k >= c} inv(L_bb)[k][c] t_k (column c of the inverse is
            // contiguous in LDS)
            double zv[(kTile * NQ + kThreads - 1) / kThreads];
#pragma unroll
            for (int u = 0; u < (kTile * NQ + kThreads - 1) / kThreads; ++u) {
                const int e = u * kThreads + tid, c = e & 63, q = e >> 6;
                double s0 = 0.0, s1 = 0.0;
                if (e < kTile * NQ && q < nq) {
                    for (int k = c; k + 1 < kTile; k += 2) {
                        s0 = fma(Dg[c * kLdDiag + k], ts[k][q], s0);
                        s1 = fma(Dg[c * kLdDiag + k + 1], ts[k + 1][q], s1);
                    }
                    if (((kTile - c) & 1) != 0) s0 = fma(Dg[c * kLdDiag + kTile - 1], ts[kTile - 1][q], s0);
                }
                zv[u] = s0 + s1;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < (kTile * NQ + kThreads - 1) / kThreads; ++u) {
                const int e = u * kThreads + tid;
                if (e < kTile * NQ && (e >> 6) < nq) ts[e & 63][e >> 6] = zv[u];
            }
            __syncthreads();
        }
        // x_blk = inv(L_bb)' t, 16 columns at a time from the last sub-block up
        for (int b16 = chain ? -1 : ((wbk - 1) & ~15); b16 >= 0; b16 -= 16) {
            const int i = tid & 15, q = tid >> 4;
            const bool act = q < nq;
            double zv = 0.0;
            if (act) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    double lv = 0.0;  // inv(L_bb)'[i][k] = inv(L_bb)[k][i]
                    if (k > i) lv = Dg[(b16 + k) * kLdDiag + b16 + i];
                    else if (k == i) lv = invd[b16 + i];
                    zv = fma(lv, ts[b16 + k][q], zv);
                }
            }
            __syncthreads();
            if (act) ts[b16 + i][q] = zv;
            __syncthreads();
            // columns above the sub-block: t[a] -= sum_k L[b16+k][a] z[k]
            for (int e = tid; e < b16 * nq; e += kThreads) {
                const int qq = e / b16, a2 = e - qq * b16;
                double accv = ts[a2][qq];
#pragma unroll
                for (int k = 0; k < 16; ++k) accv = fma(-Dg[a2 * kLdDiag + b16 + k], ts[b16 + k][qq], accv);
                ts[a2][qq] = accv;
            }
            __syncthreads();
        }
        for (int e = tid; e < wbk * nq; e += kThreads) {
            const int q = e / wbk, c = e - q * wbk;
            x[(int64_t)(q0 + q) * ldx + D.c0 + cb + c] = ts[c][q];
            if (chain)
                __hip_atomic_store(&xscratch[(int64_t)(q0 + q) * ldx + D.c0 + cb + c], unarmed(ts[c][q]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        // (chain launch: nothing else to do -- the values stored in the armed buffer are the publication)
    }
  }
}

void launch_bsolve_one(const DevicePattern& P, int n, const double* L, double* x, int nrhs, int ldx, double* y, double* y_next,
                       int* state, int* state_next, int wait_bias, int cap, hipStream_t stream) {
    const DevicePattern::OneDev& O = P.one_b;
    if (O.nblocks <= 0) return;
#define PARSY_ONE_LAUNCH(NQ)                                                                                              \
    hipLaunchKernelGGL(k_bsolve_one<NQ>, dim3(O.nblocks), dim3(kThreads), 0, stream, O.sn, P.rows, O.wleft, L, x, nrhs,    \
                       ldx, n, O.nblocks, y, y_next, state, state_next, wait_bias, cap)
    if (nrhs == 1) PARSY_ONE_LAUNCH(1);
    else if (nrhs <= 4) PARSY_ONE_LAUNCH(4);
    else PARSY_ONE_LAUNCH(8);
#undef PARSY_ONE_LAUNCH
}

// ---------------------------------------------------------------------------
// Backward solve, many right-hand sides (from mrhs_min() on): the counterpart of k_solve_small_mrhs /
// k_solve_chain_mrhs -- 64 right-hand sides per pass over L instead of 4, every product on the matrix cores.
// One workgroup (4 waves) per block of <= 64 columns, as k_bsolve_block:
//     T = Y_blk - L(below, blk)' X(below)        (64 columns x 64 right-hand sides)
// with the panel ROWS as the contraction index of v_mfma_f64_16x16x4_f64: lane (c, kk) holds L[k0 + kk][cb + c]
// (A operand: 16 columns x 4 consecutive rows -- 32-byte runs of 16 panel columns per load, every 128-byte line is
// used up by four consecutive k steps) and lane (kk, q) holds X[row(k0 + kk)][q] (B operand, gathered through the
// row ids).  The 64-row chunks below the block are dealt over the four waves, each keeps all 16 tiles of T (128
// accumulator registers; one workgroup per CU); chain launches then take the later blocks of the supernode as their
// X is published (armed buffer: the data is the flag), 16 rows per wave.  The waves' parts are subtracted from the
// staged Y one after the other (fixed order: reproducible), then X_blk = inv(L_bb)' T -- chain launches: a product
// with the inverse diagonal block (DIAG_INVERSE) on the matrix cores; otherwise the blocked substitution of
// k_bsolve_block.  Reference: the backward solve is an extension; the forward kernel it mirrors
// replaces Triangular_BCSC.h:139-157.
// ---------------------------------------------------------------------------
#define TSM(c, q) ts[(c) * kLdXm + (q)]
template <int QG>   // 16-right-hand-side groups per pass: 4 (64 per pass over L) or 1 (small launches: more workgroups)
__global__ __launch_bounds__(kThreads, QG == 4 ? 1 : 2) void k_bsolve_block_mrhs(const SnDesc* __restrict__ sn,
                                                                   const PanelDesc* __restrict__ pds,
                                                                   const int32_t* __restrict__ rows,
                                                                   const double* __restrict__ L,
                                                                   double* __restrict__ x, double* __restrict__ xscratch,
                                                                   int nrhs, int ldx, int chain, int* __restrict__ info,
                                                                   int* __restrict__ ticket, int wait_bias, int nblocks,
                                                                   const int32_t* __restrict__ ranges,
                                                                   const double* __restrict__ dinv) {
    __shared__ double Dg[kTile * kLdDiag];
    __shared__ double invd[kTile];
    __shared__ double ts[kTile * kLdXm];
    __shared__ int s_task;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    if (tid == 0) s_task = chain ? atomicAdd(ticket, 1) : (int)(blockIdx.x + blockIdx.y * nblocks);
    __syncthreads();
    const int plane = s_task / nblocks;
    const int task = s_task - plane * nblocks;
    const int q_begin = ranges ? ranges[2 * task] : task;
    const int q_end = ranges ? ranges[2 * task + 1] : q_begin + 1;
  for (int qsn = q_begin; qsn < q_end; ++qsn) {
    if (qsn > q_begin) __syncthreads();
    const PanelDesc pd = pds[qsn];
    const SnDesc D = sn[pd.sn];
    const int r = D.r, w = D.w, cb = pd.jb * kTile, wbk = min(kTile, w - cb);
    const double* __restrict__ G = L + D.px;
    const int32_t* __restrict__ ri = rows + D.pi;
    const int kbeg = cb + wbk;
    {   // diagonal block (identity padded) -> LDS; chain launch: its inverse instead (same layout)
        double dtmp[kTile * kTile / kThreads];
        const double* __restrict__ inv_blk = chain ? dinv + (int64_t)(D.dslot + pd.jb) * (kTile * kTile) : nullptr;
#pragma unroll
        for (int t = 0; t < kTile * kTile / kThreads; ++t) {
            const int e = t * kThreads + tid;
            const int c = e >> 6, i = e & 63;
            double v = (i == c) ? 1.0 : 0.0;
            if (chain) v = inv_blk[e];
            else if (c < wbk && i < wbk && i >= c) v = G[(int64_t)(cb + c) * r + cb + i];
            dtmp[t] = v;
        }
#pragma unroll
        for (int t = 0; t < kTile * kTile / kThreads; ++t) {
            const int e = t * kThreads + tid;
            Dg[(e >> 6) * kLdDiag + (e & 63)] = dtmp[t];
        }
    }
    __syncthreads();
    if (!chain && tid < kTile) invd[tid] = 1.0 / Dg[tid * kLdDiag + tid];
    __syncthreads();
    if (!chain && tid < kTile && (tid & ~15) < wbk) {   // inverses of the 16x16 diagonal sub-blocks (as k_bsolve_block)
        const int b16 = tid & ~15, c = tid & 15;
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
        __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
        for (int rr = 1; rr < 16; ++rr)
            if (rr > c) Dg[(b16 + rr) * kLdDiag + b16 + c] = y[rr];
    }
    const int nbc = (w + kTile - 1) / kTile;
    constexpr int kPassRhs = 16 * QG;
    for (int pass = plane; pass * kPassRhs < nrhs; pass += kPassLanes) {
        const int q0 = pass * kPassRhs;
        const int nq = min(kPassRhs, nrhs - q0);
        __syncthreads();
        for (int e = tid; e < kTile * kPassRhs; e += kThreads) {
            const int c = e & 63, q = e >> 6;
            TSM(c, q) = (c < wbk && q < nq) ? x[(int64_t)(q0 + q) * ldx + D.c0 + cb + c] : 0.0;
        }
        double4_s acc[4][QG];   // [16 columns][16 right-hand sides]: lane (q = l15, c = kq + 4 v)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < QG; ++b) acc[a][b] = double4_s{0, 0, 0, 0};
        // operand pointers of this lane: column c = 16 cg + l15 of the block (clamped), right-hand side 16 qg + l15
        const double* __restrict__ acol[4];
        const double* __restrict__ xcol[QG];
        bool aok[4], bok[QG];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            aok[g] = 16 * g + l15 < wbk;
            acol[g] = G + (int64_t)(cb + min(16 * g + l15, wbk - 1)) * r;
        }
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            bok[g] = 16 * g + l15 < nq;
            xcol[g] = x + (int64_t)(q0 + min(16 * g + l15, nq - 1)) * ldx;
        }
        // ---- the rows below (chain: below the supernode's own columns -- ancestors, final; otherwise everything
        // below the block: the later blocks were solved by earlier launches): 64-row chunks over the waves, 16 rows
        // (4 k steps) of loads in flight ahead of their 64 products
        // Software-pipelined over the k steps (4 rows each) of all of this wave's chunks: the operands of step t + 3 are
        // loaded while the 4 x QG products of step t are issued (a ring of four operand sets); the row ids of a chunk are
        // ONE load per lane, issued a chunk and a half ahead and handed to the lanes that need them by ds_bpermute.
        // (The plain form -- ids, then operands, then 64 products, per 16 rows -- waited two dependent round trips per 64
        // products: Flan-class input, 64 right-hand sides: 45 -> 37 ms per backward solve.)
        {
            const int kstart = (chain ? w : kbeg) + 64 * wave;
            constexpr int kStride = 64 * (kThreads / 64);
            const int nsteps = kstart < r ? 16 * ((r - kstart + kStride - 1) / kStride) : 0;
            auto row_ids = [&](int j) {   // lane l: the row id of row l of this wave's chunk j (clamped into the panel)
                const int kr = min(kstart + kStride * j + lane, r - 1);
                return (kr < w) ? (D.c0 + kr) : ri[kr];
            };
            int ridA = row_ids(0), ridB = row_ids(1);
            double ra[4][4], rb[4][QG];
            auto load = [&](int t, double (&A)[4], double (&B)[QG]) {
                const int j = t >> 4, u = t & 15;
                const int k = kstart + kStride * j + 4 * u + kq;
                const bool kin = k < r && t < nsteps;
                const int kc = min(k, r - 1);
                const int rid = __builtin_amdgcn_ds_bpermute((4 * u + kq) * 4, (j & 1) ? ridB : ridA);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const double a = acol[g][kc];
                    A[g] = (kin && aok[g]) ? a : 0.0;
                }
#pragma unroll
                for (int g = 0; g < QG; ++g) {
                    const double b2 = xcol[g][rid];
                    B[g] = (kin && bok[g]) ? b2 : 0.0;
                }
            };
            if (nsteps > 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t) load(t, ra[t], rb[t]);
            }
            for (int t = 0; t < nsteps; t += 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tt = t + i;
                    load(tt + 3, ra[(i + 3) & 3], rb[(i + 3) & 3]);
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                        for (int qg = 0; qg < QG; ++qg)
                            acc[cg][qg] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[i][cg], rb[i][qg], acc[cg][qg], 0, 0, 0);
                    if ((tt & 15) == 13) {   // the loads of this chunk's steps are all issued: its id register takes chunk j + 2
                        const int j = tt >> 4;
                        const int v = row_ids(j + 2);
                        if (j & 1) ridB = v;
                        else ridA = v;
                    }
                }
            }
        }
        if (chain) {
            // ---- the later blocks of this supernode, last one first, each as soon as its X is published: rows
            // 64 I + 16 wave .. + 15 for this wave
            bool ok = true;
            for (int I = nbc - 1; I > pd.jb && ok; --I) {
                double av[4][4], bv[4][QG];
                int kk[4];
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    kk[st] = I * kTile + 16 * wave + 4 * st + kq;
                    const int kc = min(kk[st], w - 1);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const double a = acol[g][kc];
                        av[st][g] = (kk[st] < w && aok[g]) ? a : 0.0;
                    }
                }
                const unsigned long long t0 = wall_clock64();
                int spins = 0;
                // Only the block right above this one is on the critical path of the chain.  A workgroup further up
                // would poll for a long time, with 16 loads per lane and round, next to hundreds of others on the
                // same lines (the top separator: 313 workgroups): it first watches ONE value of the block lazily
                // (the last column this wave reads: one load per wave and round), and goes on to the full poll --
                // normally satisfied at once -- when that value is there.
                if (I - pd.jb > 1) {
                    const int kw = min(I * kTile + 16 * wave + 15, w - 1);
                    const long long* __restrict__ watch =
                        reinterpret_cast<const long long*>(xscratch + (int64_t)q0 * ldx + D.c0 + kw);
                    while (__hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kXArmed ||
                           wait_bias != 0) {
                        if ((++spins & 15) == 0 &&
                            (wall_clock64() - t0 > kSolveSpinTicks ||
                             __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
                            ok = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(48);
                    }
                    if (!ok) {
                        if (lane == 0) atomicMin(info, -1);
                        break;
                    }
                }
                for (;;) {
                    bool in = true;
#pragma unroll
                    for (int st = 0; st < 4; ++st)
#pragma unroll
                        for (int g = 0; g < QG; ++g) {
                            long long b = 0;
                            if (kk[st] < w && bok[g])
                                b = __hip_atomic_load(reinterpret_cast<const long long*>(
                                                          xscratch + (xcol[g] - x) + D.c0 + kk[st]),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            in = in && b != kXArmed;
                            bv[st][g] = __longlong_as_double(b);
                        }
                    if (__all(in && wait_bias == 0)) break;
                    if ((++spins & 15) == 0 &&
                        (wall_clock64() - t0 > kSolveSpinTicks ||
                         __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok) {
                    if (lane == 0) atomicMin(info, -1);
                    break;   // (the result is wrong and reported; nobody may hang)
                }
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                        for (int qg = 0; qg < QG; ++qg)
                            acc[cg][qg] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st][cg], bv[st][qg], acc[cg][qg], 0, 0, 0);
            }
        }
        // ---- T = Y - (the waves' parts, one wave after the other: a fixed order of sums; waves that had no rows --
        // fewer than four 64-row chunks below the block and no chain -- are skipped)
        const int nparts = chain ? kThreads / 64 : min(kThreads / 64, (r - kbeg + 63) / 64);
        for (int wv = 0; wv < nparts; ++wv) {
            __syncthreads();
            if (wave == wv) {
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg)
#pragma unroll
                        for (int v = 0; v < 4; ++v) TSM(16 * cg + kq + 4 * v, 16 * qg + l15) -= acc[cg][qg][v];
            }
        }
        __syncthreads();
        if (chain) {
            // X_blk = inv(L_bb)' T on the matrix cores: X[c][q] = sum_k inv(L_bb)[k][c] T[k][q]; Dg[c][k] holds
            // inv(L_bb)[k][c] (zero for k < c).  Wave = 16 columns; the result replaces T behind a barrier.
            double4_s out[QG];
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) out[qg] = double4_s{0, 0, 0, 0};
            for (int st = 4 * wave; st < kTile / 4; ++st) {   // (k < 16 wave: the inverse is zero there)
                const double a = Dg[(16 * wave + l15) * kLdDiag + 4 * st + kq];
#pragma unroll
                for (int qg = 0; qg < QG; ++qg)
                    out[qg] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, TSM(4 * st + kq, 16 * qg + l15), out[qg], 0, 0, 0);
            }
            __syncthreads();
#pragma unroll
            for (int qg = 0; qg < QG; ++qg)
#pragma unroll
                for (int v = 0; v < 4; ++v) TSM(16 * wave + kq + 4 * v, 16 * qg + l15) = out[qg][v];
            __syncthreads();
        }
        // otherwise: x_blk = inv(L_bb)' t, 16 columns at a time from the last sub-block up (as k_bsolve_block)
        for (int b16 = chain ? -1 : ((wbk - 1) & ~15); b16 >= 0; b16 -= 16) {
            const int i = tid & 15;
            double zv[QG];
#pragma unroll
            for (int u = 0; u < QG; ++u) {
                const int q = (tid >> 4) + 16 * u;
                double z = 0.0;
                if (q < nq) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        double lv = 0.0;  // inv(L_bb)'[i][k] = inv(L_bb)[k][i]
                        if (k > i) lv = Dg[(b16 + k) * kLdDiag + b16 + i];
                        else if (k == i) lv = invd[b16 + i];
                        z = fma(lv, TSM(b16 + k, q), z);
                    }
                }
                zv[u] = z;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < QG; ++u) {
                const int q = (tid >> 4) + 16 * u;
                if (q < nq) TSM(b16 + i, q) = zv[u];
            }
            __syncthreads();
            // columns above the sub-block: t[a] -= sum_k L[b16+k][a] z[k]
            for (int e = tid; e < b16 * nq; e += kThreads) {
                const int qq = e / b16, a2 = e - qq * b16;
                double accv = TSM(a2, qq);
#pragma unroll
                for (int k = 0; k < 16; ++k) accv = fma(-Dg[a2 * kLdDiag + b16 + k], TSM(b16 + k, qq), accv);
                TSM(a2, qq) = accv;
            }
            __syncthreads();
        }
        for (int e = tid; e < kTile * kPassRhs; e += kThreads) {
            const int c = e & 63, q = e >> 6;
            if (c < wbk && q < nq) {
                x[(int64_t)(q0 + q) * ldx + D.c0 + cb + c] = TSM(c, q);
                if (chain)
                    __hip_atomic_store(&xscratch[(int64_t)(q0 + q) * ldx + D.c0 + cb + c], unarmed(TSM(c, q)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
  }
}
// ---------------------------------------------------------------------------
// The chain launches of the backward solve with many right-hand sides, 64 per pass (round 5): a kernel of its own -- inside
// k_bsolve_block_mrhs, beside that kernel's other forms, the compiler spilled 235 registers.  One workgroup per block column
// of a wide supernode, taken by ticket (last block column first).  First the rows below the supernode's own columns (x final
// there), 64-row chunks over the four waves as in k_bsolve_block_mrhs, reduced ONCE into the staged Y; then the chain, every
// wave for itself on 16 right-hand sides and all 64 columns (the comment inside).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 1) void k_bsolve_chain_mrhs(const SnDesc* __restrict__ sn, const PanelDesc* __restrict__ pds,
                                                                  const int32_t* __restrict__ rows, const double* __restrict__ L,
                                                                  double* __restrict__ x, double* __restrict__ xscratch, int nrhs,
                                                                  int ldx, int* __restrict__ info, int* __restrict__ ticket,
                                                                  int wait_bias, int nblocks, const double* __restrict__ dinv) {
    constexpr int QG = 4;
    __shared__ double Dg[kTile * kLdDiag];
    __shared__ double ts[kTile * kLdXm];
    __shared__ double Ms[kTile * kLdDiag];   // Ms[k][c] = M[c][k]
    __shared__ int s_task;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    if (tid == 0) s_task = atomicAdd(ticket, 1);
    __syncthreads();
    const int plane = s_task / nblocks;
    const PanelDesc pd = pds[s_task - plane * nblocks];
    const SnDesc D = sn[pd.sn];
    const int r = D.r, w = D.w, cb = pd.jb * kTile, wbk = min(kTile, w - cb);
    const double* __restrict__ G = L + D.px;
    const int32_t* __restrict__ ri = rows + D.pi;
    {   // the inverse diagonal block (DIAG_INVERSE; identity outside the block) -> LDS: Dg[c][k] = inv(L_jj)[k][c]
        double dtmp[kTile * kTile / kThreads];
        const double* __restrict__ inv_blk = dinv + (int64_t)(D.dslot + pd.jb) * (kTile * kTile);
#pragma unroll
        for (int t = 0; t < kTile * kTile / kThreads; ++t) dtmp[t] = inv_blk[t * kThreads + tid];
#pragma unroll
        for (int t = 0; t < kTile * kTile / kThreads; ++t) {
            const int e = t * kThreads + tid;
            Dg[(e >> 6) * kLdDiag + (e & 63)] = dtmp[t];
        }
    }
    const int nbc = (w + kTile - 1) / kTile;
    const bool has_up = pd.jb + 1 < nbc;
    __syncthreads();
    if (has_up) {
        // M[c'][k] = sum_c inv(L_jj)[c][c'] L[64 (jb+1) + k][cb + c]: wave v the k's 16 v .. 16 v + 15
        const int k0 = (pd.jb + 1) * kTile + 16 * wave + l15;      // this lane's row of block jb + 1 (B operand: j = l15)
        const bool kin = k0 < w;
        const double* __restrict__ brow = G + (int64_t)cb * r + min(k0, w - 1);
        double lb[16];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int c = 4 * st + kq;
            const double v = brow[(int64_t)min(c, wbk - 1) * r];
            lb[st] = (kin && c < wbk) ? v : 0.0;
        }
        double4_s am[4];
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) am[cg] = double4_s{0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < 16; ++st)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
                if (st >= 4 * cg)     // (inv(L_jj)[c][c'] = 0 for c < c')
                    am[cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(Dg[(16 * cg + l15) * kLdDiag + 4 * st + kq], lb[st], am[cg], 0, 0, 0);
#pragma unroll
        for (int cg = 0; cg < 4; ++cg)
#pragma unroll
            for (int v = 0; v < 4; ++v) Ms[(16 * wave + l15) * kLdDiag + 16 * cg + kq + 4 * v] = am[cg][v];
    }
    constexpr int kPassRhs = 16 * QG;
    for (int pass = plane; pass * kPassRhs < nrhs; pass += kPassLanes) {
        const int q0 = pass * kPassRhs;
        const int nq = min(kPassRhs, nrhs - q0);
        __syncthreads();
        for (int e = tid; e < kTile * kPassRhs; e += kThreads) {
            const int c = e & 63, q = e >> 6;
            TSM(c, q) = (c < wbk && q < nq) ? x[(int64_t)(q0 + q) * ldx + D.c0 + cb + c] : 0.0;
        }
        double4_s acc[4][QG];   // [16 columns][16 right-hand sides]: lane (q = l15, c = kq + 4 v)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < QG; ++b) acc[a][b] = double4_s{0, 0, 0, 0};
        // operand pointers of this lane: column c = 16 cg + l15 of the block (clamped), right-hand side 16 qg + l15
        const double* __restrict__ acol[4];
        const double* __restrict__ xcol[QG];
        bool aok[4], bok[QG];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            aok[g] = 16 * g + l15 < wbk;
            acol[g] = G + (int64_t)(cb + min(16 * g + l15, wbk - 1)) * r;
        }
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            bok[g] = 16 * g + l15 < nq;
            xcol[g] = x + (int64_t)(q0 + min(16 * g + l15, nq - 1)) * ldx;
        }
        // ---- the rows below (chain: below the supernode's own columns -- ancestors, final; otherwise everything
        // below the block: the later blocks were solved by earlier launches): 64-row chunks over the waves, 16 rows
        // (4 k steps) of loads in flight ahead of their 64 products
        // Software-pipelined over the k steps (4 rows each) of all of this wave's chunks: the operands of step t + 3 are
        // loaded while the 4 x QG products of step t are issued (a ring of four operand sets); the row ids of a chunk are
        // ONE load per lane, issued a chunk and a half ahead and handed to the lanes that need them by ds_bpermute.
        // (The plain form -- ids, then operands, then 64 products, per 16 rows -- waited two dependent round trips per 64
        // products: Flan-class input, 64 right-hand sides: 45 -> 37 ms per backward solve.)
        {
            const int kstart = w + 64 * wave;
            constexpr int kStride = 64 * (kThreads / 64);
            const int nsteps = kstart < r ? 16 * ((r - kstart + kStride - 1) / kStride) : 0;
            auto row_ids = [&](int j) {   // lane l: the row id of row l of this wave's chunk j (clamped into the panel)
                const int kr = min(kstart + kStride * j + lane, r - 1);
                return (kr < w) ? (D.c0 + kr) : ri[kr];
            };
            int ridA = row_ids(0), ridB = row_ids(1);
            double ra[4][4], rb[4][QG];
            auto load = [&](int t, double (&A)[4], double (&B)[QG]) {
                const int j = t >> 4, u = t & 15;
                const int k = kstart + kStride * j + 4 * u + kq;
                const bool kin = k < r && t < nsteps;
                const int kc = min(k, r - 1);
                const int rid = __builtin_amdgcn_ds_bpermute((4 * u + kq) * 4, (j & 1) ? ridB : ridA);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const double a = acol[g][kc];
                    A[g] = (kin && aok[g]) ? a : 0.0;
                }
#pragma unroll
                for (int g = 0; g < QG; ++g) {
                    const double b2 = xcol[g][rid];
                    B[g] = (kin && bok[g]) ? b2 : 0.0;
                }
            };
            if (nsteps > 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t) load(t, ra[t], rb[t]);
            }
            for (int t = 0; t < nsteps; t += 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tt = t + i;
                    load(tt + 3, ra[(i + 3) & 3], rb[(i + 3) & 3]);
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                        for (int qg = 0; qg < QG; ++qg)
                            acc[cg][qg] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[i][cg], rb[i][qg], acc[cg][qg], 0, 0, 0);
                    if ((tt & 15) == 13) {   // the loads of this chunk's steps are all issued: its id register takes chunk j + 2
                        const int j = tt >> 4;
                        const int v = row_ids(j + 2);
                        if (j & 1) ridB = v;
                        else ridA = v;
                    }
                }
            }
        }
        {
            // ---- the later blocks of this supernode but the one right above, last one first, each as soon as its X is
            // published: rows 64 I + 16 wave .. + 15 for this wave (every 128-byte line of L read once per workgroup: with
            // all 64 rows per wave -- 16 lines per load instruction, four times over -- the Flan-class solve took 45-49 ms)
            bool ok = true;
            for (int I = nbc - 1; I > pd.jb + 1 && ok; --I) {
                double av[4][4], bv[4][QG];
                int kk[4];
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    kk[st] = I * kTile + 16 * wave + 4 * st + kq;
                    const int kc = min(kk[st], w - 1);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const double a = acol[g][kc];
                        av[st][g] = (kk[st] < w && aok[g]) ? a : 0.0;
                    }
                }
                const unsigned long long t0 = wall_clock64();
                int spins = 0;
                // Only the block right above this one is on the critical path of the chain.  A workgroup further up
                // would poll for a long time, with 16 loads per lane and round, next to hundreds of others on the
                // same lines (the top separator: 313 workgroups): it first watches ONE value of the block lazily
                // (the last column this wave reads: one load per wave and round), and goes on to the full poll --
                // normally satisfied at once -- when that value is there.
                if (I - pd.jb > 1) {
                    const int kw = min(I * kTile + 16 * wave + 15, w - 1);
                    const long long* __restrict__ watch =
                        reinterpret_cast<const long long*>(xscratch + (int64_t)q0 * ldx + D.c0 + kw);
                    while (__hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kXArmed ||
                           wait_bias != 0) {
                        if ((++spins & 15) == 0 &&
                            (wall_clock64() - t0 > kSolveSpinTicks ||
                             __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
                            ok = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(48);
                    }
                    if (!ok) {
                        if (lane == 0) atomicMin(info, -1);
                        break;
                    }
                }
                for (;;) {
                    bool in = true;
#pragma unroll
                    for (int st = 0; st < 4; ++st)
#pragma unroll
                        for (int g = 0; g < QG; ++g) {
                            long long b = 0;
                            if (kk[st] < w && bok[g])
                                b = __hip_atomic_load(reinterpret_cast<const long long*>(
                                                          xscratch + (xcol[g] - x) + D.c0 + kk[st]),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            in = in && b != kXArmed;
                            bv[st][g] = __longlong_as_double(b);
                        }
                    if (__all(in && wait_bias == 0)) break;
                    if ((++spins & 15) == 0 &&
                        (wall_clock64() - t0 > kSolveSpinTicks ||
                         __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok) {
                    if (lane == 0) atomicMin(info, -1);
                    break;   // (the result is wrong and reported; nobody may hang)
                }
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                        for (int qg = 0; qg < QG; ++qg)
                            acc[cg][qg] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st][cg], bv[st][qg], acc[cg][qg], 0, 0, 0);
            }
        }
        {
            // ---- The chain (round 5: the mirror of k_solve_blocks_mrhs's block-column task).  Its step used to be: 16 rows of
            // block I per wave (64 products), the four waves' parts subtracted from the staged Y one after the other behind
            // barriers, the product with the inverse block by 16 columns per wave behind two more, the stores through LDS -- 19 us
            // per block column on the Flan-class top separator.  Now the parts are reduced ONCE, after the block second next
            // (T in LDS), and what follows is every wave's own, on 16 right-hand sides and all 64 columns, no barrier:
            //     P = inv(L_jj)' T,     M = inv(L_jj)' L(jb + 1, jb)'   (64 x 64, formed when the pass starts)
            //     X_jb = P - M X_(jb+1)                                   (all that is left behind the last wait)
            // lane (q = l15, kk = kq) holds T[4 st + kk][q] -- the accumulator layout is the B-operand layout of k step
            // st = 4 cg + v --, results go straight from the accumulators to the armed buffer and x.
#ifdef PARSY_BLKSTAMPS
            { const int jb = pd.jb; BLK_STAMP(4); }
#endif
            // (the reduction in four rounds, all waves at once: in round rd wave v subtracts its part of the columns
            // 16 ((v + rd) & 3) .. + 15 -- disjoint quarters of T, a fixed order of sums per entry; one wave after the other
            // with all of its sixteen tiles took 4 us of the step)
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                __syncthreads();
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
                    if (cg == ((wave + rd) & 3)) {
#pragma unroll
                        for (int qg = 0; qg < QG; ++qg)
#pragma unroll
                            for (int v = 0; v < 4; ++v) TSM(16 * cg + kq + 4 * v, 16 * qg + l15) -= acc[cg][qg][v];
                    }
            }
            __syncthreads();
#ifdef PARSY_BLKSTAMPS
            const int jb = pd.jb;   // (BLK_STAMP's index)
            BLK_STAMP(5);
#endif
            const bool won = 16 * wave < nq;                 // this wave's 16 right-hand sides are in the pass
            const bool qok = 16 * wave + l15 < nq;
            double tv[16];
#pragma unroll
            for (int st = 0; st < 16; ++st) tv[st] = TSM(4 * st + kq, 16 * wave + l15);
            if (won) {
                const int64_t qoff = (int64_t)(q0 + min(16 * wave + l15, nq - 1)) * ldx + D.c0;   // this lane's right-hand side, row c0
                const double* __restrict__ xq = xscratch + qoff;
                bool ok = true;
                auto take_x = [&](int I, bool lazy, double (&bv)[16]) __attribute__((always_inline)) {
                    const unsigned long long t0 = wall_clock64();
                    int spins = 0;
                    auto give_up = [&]() {
                        return (++spins & 15) == 0 && (wall_clock64() - t0 > kSolveSpinTicks ||
                                                       __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0);
                    };
                    if (lazy) {   // (a block further up: watch ONE value first -- the top separator has 313 workgroups polling)
                        const long long* __restrict__ watch = reinterpret_cast<const long long*>(xq + min(I * kTile + 63, w - 1));
                        while (__hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kXArmed || wait_bias != 0) {
                            if (give_up()) {
                                ok = false;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(48);
                        }
                    }
                    while (ok) {
                        bool in = true;
#pragma unroll
                        for (int st = 0; st < 16; ++st) {
                            const int k = I * kTile + 4 * st + kq;
                            long long b = 0;
                            if (k < w && qok)
                                b = __hip_atomic_load(reinterpret_cast<const long long*>(xq + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            in = in && b != kXArmed;
                            bv[st] = __longlong_as_double(b);
                        }
                        if (__all(in && wait_bias == 0)) break;
                        if (give_up()) ok = false;
                        else __builtin_amdgcn_s_sleep(1);
                    }
                };
                // P = inv(L_jj)' T: X[c'][q] = sum_{c >= c'} inv(L_jj)[c][c'] T[c][q]
                double4_s out[4];
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) out[cg] = double4_s{0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 16; ++st)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg)
                        if (st >= 4 * cg)
                            out[cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(Dg[(16 * cg + l15) * kLdDiag + 4 * st + kq], tv[st], out[cg], 0, 0, 0);
                if (has_up) {
                    // (M's operands take the registers of the last block's: not before that block is multiplied)
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    double mv[16][4];
#pragma unroll
                    for (int st = 0; st < 16; ++st)
#pragma unroll
                        for (int cg = 0; cg < 4; ++cg) mv[st][cg] = Ms[(4 * st + kq) * kLdDiag + 16 * cg + l15];
                    double bv[16];
                    BLK_STAMP(6);
                    take_x(pd.jb + 1, false, bv);
                    if (!ok) {
                        if (lane == 0) atomicMin(info, -1);
                        return;
                    }
                    BLK_STAMP(0);
                    double4_s m0[4], m1[4];
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) m0[cg] = m1[cg] = double4_s{0, 0, 0, 0};
#pragma unroll
                    for (int st = 0; st < 16; st += 2)
#pragma unroll
                        for (int cg = 0; cg < 4; ++cg) {
                            m0[cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(mv[st][cg], bv[st], m0[cg], 0, 0, 0);
                            m1[cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(mv[st + 1][cg], bv[st + 1], m1[cg], 0, 0, 0);
                        }
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) out[cg] -= m0[cg] + m1[cg];
                }
#ifdef PARSY_BLKSTAMPS
                asm volatile("" ::"v"(out[3][3]));
                BLK_STAMP(1);
#endif
                // straight from the accumulators (lane (q = l15, c = 16 cg + kq + 4 v)): the armed buffer first, then x
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int c = 16 * cg + kq + 4 * v;
                        if (c < wbk && qok)
                            __hip_atomic_store(&xscratch[qoff + cb + c], unarmed(out[cg][v]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                BLK_STAMP(3);
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int c = 16 * cg + kq + 4 * v;
                        if (c < wbk && qok) x[qoff + cb + c] = out[cg][v];
                    }
            }
        }
    }
}

#undef TSM

// Backward solve of supernodes of width <= 16: one WAVE per supernode, or per subtree of them from its root down
// (ranges != null), no LDS and no barrier -- the counterpart of k_solve_tiny.  The rows below the diagonal block
// are one row per lane (their x is final: ancestors), every lane keeps its part of t_c = sum_k L[k][c] x_k for
// the <= 16 columns and the parts are summed across the wave once, at the end (xor butterfly); lane c then holds
// column c of the diagonal block and the transposed substitution goes from the last column up with lane
// broadcasts (v_readlane).
template <int kTinyW>   // width class of the launch: kTinyWidth (16) or kTinyWidth2 (32)
__global__ __launch_bounds__(64) void k_bsolve_tiny(const SnDesc* __restrict__ sn, const PanelDesc* __restrict__ pds,
                                                    const int32_t* __restrict__ ranges,
                                                    const int32_t* __restrict__ rows, const double* __restrict__ L,
                                                    double* __restrict__ x, int nrhs, int ldx) {
    __shared__ double s_red[kTinyW * kLdRedB];   // the lanes' parts of the column sums
    const int lane = threadIdx.x;
    const int q_begin = ranges ? ranges[2 * blockIdx.x] : (int)blockIdx.x;
    const int q_end = ranges ? ranges[2 * blockIdx.x + 1] : q_begin + 1;
    for (int qsn = q_begin; qsn < q_end; ++qsn) {
        // (subtree launch: the x this wave stored for the supernode before -- an ancestor -- is read back below)
        if (qsn > q_begin) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const SnDesc D = sn[pds[qsn].sn];
        const int r = D.r, w = D.w;
        const double* __restrict__ G = L + D.px;
        const int32_t* __restrict__ ri = rows + D.pi;
        // lane c: column c of the diagonal block (rows 0..w-1); every lane: its row of the first 64 rows below
        // (unconditional loads, clamped into the panel)
        const int cl = min(lane, w - 1);
        const int kb = min(w + lane, r - 1);
        double lc[kTinyW], lb[kTinyW];
#pragma unroll
        for (int k = 0; k < kTinyW; ++k) {
            lc[k] = G[(int64_t)cl * r + min(k, w - 1)];
            lb[k] = G[(int64_t)min(k, w - 1) * r + kb];
        }
        const int row_b = ri[kb];
        double diag = 1.0;
#pragma unroll
        for (int k = 0; k < kTinyW; ++k) diag = (lane == k && k < w) ? lc[k] : diag;
        const double rdiag = 1.0 / diag;
        for (int q = blockIdx.y; q < nrhs; q += gridDim.y) {
            double* __restrict__ xq = x + (int64_t)q * ldx;
            // parts of t over this lane's rows
            double p[kTinyW];
            const double xb = (w + lane < r) ? xq[row_b] : 0.0;
#pragma unroll
            for (int c = 0; c < kTinyW; ++c) p[c] = lb[c] * xb;
            for (int k = w + 64 + lane; k < r; k += 64) {
                const double xk = xq[ri[k]];
#pragma unroll
                for (int c = 0; c < kTinyW; ++c) p[c] = fma(G[(int64_t)min(c, w - 1) * r + k], xk, p[c]);
            }
            double t = lane < w ? xq[D.c0 + lane] : 0.0;
            {
                // column sums across the wave through LDS: lane (c, g) adds up 64 / kGroups of the lanes' parts of
                // column c, the exchanges over g finish it
                constexpr int kGroups = 64 / kTinyW, kPer = 64 / kGroups;
                __builtin_amdgcn_wave_barrier();   // (the parts of the right-hand side / supernode before are consumed)
#pragma unroll
                for (int c = 0; c < kTinyW; ++c) s_red[c * kLdRedB + lane] = p[c];
                __builtin_amdgcn_wave_barrier();
                const int c = lane % kTinyW, g = lane / kTinyW;
                double v0 = 0.0, v1 = 0.0;
#pragma unroll
                for (int i = 0; i < kPer; i += 2) {
                    v0 += s_red[c * kLdRedB + kPer * g + i];
                    v1 += s_red[c * kLdRedB + kPer * g + i + 1];
                }
                double v = v0 + v1;
#pragma unroll
                for (int off = kTinyW; off < 64; off <<= 1) v += __shfl_xor(v, off);
                t = (lane < w) ? t - v : t;   // (lane c < kTinyW holds column c's sum: g == 0)
            }
            // x_k = (t_k - sum_{j > k} L[j][k] x_j) / L[k][k], from the last column up: lane c holds L[k][c] = lc[k]
            double xfin = 0.0;
#pragma unroll
            for (int k = kTinyW - 1; k >= 0; --k) {
                if (k < w) {
                    const double xk = readlane_f64(t * rdiag, k);
                    t = (lane < k) ? fma(-lc[k], xk, t) : t;
                    xfin = (lane == k) ? xk : xfin;
                }
            }
            if (lane < w) xq[D.c0 + lane] = xfin;
        }
    }
}

// Backward solve of supernodes of width <= 16, many right-hand sides: one WAVE per supernode (or per subtree of
// them from its root down), 64 right-hand sides per pass, products on the matrix cores.  T = Y - L21' X(below) as in
// k_bsolve_block_mrhs (panel rows = contraction index; 4 tiles of 16 columns x 16 right-hand sides); the result
// layout of v_mfma_f64_16x16x4_f64 (lane (q, c = kq + 4 v)) is the B-operand layout of k step v, so
// X_blk = inv(L_bb)' T is four more products per tile with T straight from the accumulators -- inv(L_bb) by
// substitution, one column per lane, through 4 KB of LDS.
__global__ __launch_bounds__(64, 4) void k_bsolve_tiny_mrhs(const SnDesc* __restrict__ sn, const PanelDesc* __restrict__ pds,
                                                         const int32_t* __restrict__ ranges,
                                                         const int32_t* __restrict__ rows, const double* __restrict__ L,
                                                         double* __restrict__ x, int nrhs, int ldx) {
    constexpr int W = kTinyWidth, kLd = W + 1;
    __shared__ double Db[W * kLd];    // diagonal block, column-major: Db[c * kLd + i] = L[i][c]
    __shared__ double Iv[W * kLd];    // its inverse:                  Iv[c * kLd + i] = inv(L_bb)[i][c]
    const int lane = threadIdx.x, l15 = lane & 15, kq = lane >> 4;
    const int q_begin = ranges ? ranges[2 * blockIdx.x] : (int)blockIdx.x;
    const int q_end = ranges ? ranges[2 * blockIdx.x + 1] : q_begin + 1;
    for (int qsn = q_begin; qsn < q_end; ++qsn) {
        // (subtree launch: the x this wave stored for the supernode before -- an ancestor -- is read back below)
        if (qsn > q_begin) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const SnDesc D = sn[pds[qsn].sn];
        const int r = D.r, w = D.w;
        const double* __restrict__ G = L + D.px;
        const int32_t* __restrict__ ri = rows + D.pi;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < W * W / 64; ++t) {
            const int e = t * 64 + lane, c = e >> 4, i = e & 15;
            double v = (i == c) ? 1.0 : 0.0;
            if (c < w && i < w && i >= c) v = G[(int64_t)c * r + i];
            Db[c * kLd + i] = v;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < W) {   // column `lane` of inv(L_bb): y_c = 1 / d_c, y_rr = -(sum_{k < rr} L[rr][k] y_k) / d_rr
            // (y lives in LDS, one row of the block per step: a register array indexed by the step would go to scratch)
            const int c = lane;
            double* __restrict__ y = &Iv[c * kLd];
#pragma unroll
            for (int k = 0; k < W; ++k) y[k] = (k == c) ? 1.0 / Db[k * kLd + k] : 0.0;
#pragma unroll 1
            for (int rr = 1; rr < W; ++rr) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int k = 0; k < W - 1; k += 2) {
                    const double l0 = Db[k * kLd + rr], l1 = Db[(k + 1) * kLd + rr];   // (zero above the diagonal: k > rr)
                    s0 = fma(k < rr ? l0 : 0.0, y[k], s0);
                    s1 = fma(k + 1 < rr ? l1 : 0.0, y[k + 1], s1);
                }
                if (rr > c) y[rr] = -(s0 + s1) / Db[rr * kLd + rr];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // A operands of the diagonal product: lane (c_out = l15, k = 4 st + kq) <- inv(L_bb)[k][c_out]
        double ainv[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) ainv[st] = Iv[l15 * kLd + 4 * st + kq];
        const bool aok = l15 < w;
        const double* __restrict__ acol = G + (int64_t)min(l15, w - 1) * r;
        for (int q0 = 64 * (int)blockIdx.y; q0 < nrhs; q0 += 64 * (int)gridDim.y) {
            const int nq = min(64, nrhs - q0), nqg = (nq + 15) >> 4;
            double4_s acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = double4_s{0, 0, 0, 0};
            const double* __restrict__ xcol[4];
            bool bok[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bok[g] = 16 * g + l15 < nq;
                xcol[g] = x + (int64_t)(q0 + min(16 * g + l15, nq - 1)) * ldx;
            }
            constexpr int kSt = 2;   // k steps (4 rows each) whose loads are in flight together
#pragma unroll 1
            for (int k0 = w; k0 < r; k0 += 4 * kSt) {
                double av[kSt], bv[kSt][4];
#pragma unroll
                for (int st = 0; st < kSt; ++st) {
                    const int k = k0 + 4 * st + kq;
                    const bool kin = k < r;
                    const int kc = min(k, r - 1);
                    const int rid = ri[kc];
                    const double a = acol[kc];
                    av[st] = (kin && aok) ? a : 0.0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        double b = 0.0;
                        if (g < nqg) b = xcol[g][rid];
                        bv[st][g] = (kin && bok[g]) ? b : 0.0;
                    }
                }
#pragma unroll
                for (int st = 0; st < kSt; ++st)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (g < nqg) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], bv[st][g], acc[g], 0, 0, 0);
            }
            // T = Y - acc in the accumulator layout (lane (q = l15, c = kq + 4 v)), then X_blk = inv(L_bb)' T
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g >= nqg) continue;
                double4_s t;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = kq + 4 * v;
                    const double y = xcol[g][D.c0 + min(c, w - 1)];
                    t[v] = (c < w && bok[g]) ? y - acc[g][v] : 0.0;
                }
                double4_s out = {0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 4; ++st) out = __builtin_amdgcn_mfma_f64_16x16x4f64(ainv[st], t[st], out, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int c = kq + 4 * v;
                    if (c < w && bok[g]) x[(int64_t)(q0 + 16 * g + l15) * ldx + D.c0 + c] = out[v];
                }
            }
        }
    }
}

// Backward solve, one right-hand side: the sums of a wide supernode's block column over the rows BELOW the supernode's
// own columns (x of the ancestors: final when the level starts), for the chain launches of few, tall supernodes.
// One workgroup (4 waves) per (supernode, block column, chunk of kBelowRows rows): wave q carries columns 16 q .. 16 q + 15
// of the block, lanes along the rows (coalesced), the lanes' parts added up through LDS as in k_bsolve_chain_w; the 64
// sums of the chunk go to its slot of `part`.  The chain adds the slots of a block column in chunk order.
__global__ __launch_bounds__(kThreads) void k_bsolve_below(const SnDesc* __restrict__ sn, const PanelDesc* __restrict__ tasks,
                                                           const int32_t* __restrict__ rows, const double* __restrict__ L,
                                                           const double* __restrict__ x, double* __restrict__ part) {
    constexpr int kCols = 16, kLdR = 65;
    __shared__ double s_red[kThreads / 64][kCols * kLdR];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const PanelDesc pd = tasks[blockIdx.x];
    const SnDesc D = sn[pd.sn];
    const int r = D.r, w = D.w;
    const int cb = pd.jb * kTile, wbk = min(kTile, w - cb);
    const double* __restrict__ G = L + D.px;
    const int32_t* __restrict__ ri = rows + D.pi;
    const double* __restrict__ colp[kCols];
#pragma unroll
    for (int ci = 0; ci < kCols; ++ci) colp[ci] = G + (int64_t)(cb + min(kCols * wave + ci, wbk - 1)) * r;
    double acc[kCols];
#pragma unroll
    for (int ci = 0; ci < kCols; ++ci) acc[ci] = 0.0;
    const int k1 = min(pd.row0 + kBelowRows, r);
    for (int k0 = pd.row0 + lane; k0 < k1; k0 += 128) {   // two 64-row pieces in flight
        const int ka = k0, kb = min(k0 + 64, r - 1);
        const int xa = ri[ka], xb = ri[kb];
        double la[kCols], lb[kCols];
#pragma unroll
        for (int ci = 0; ci < kCols; ++ci) la[ci] = colp[ci][ka];
#pragma unroll
        for (int ci = 0; ci < kCols; ++ci) lb[ci] = colp[ci][kb];
        const double va = x[xa], vb = (k0 + 64 < k1) ? x[xb] : 0.0;
#pragma unroll
        for (int ci = 0; ci < kCols; ++ci) acc[ci] = fma(la[ci], va, acc[ci]);
#pragma unroll
        for (int ci = 0; ci < kCols; ++ci) acc[ci] = fma(lb[ci], vb, acc[ci]);
    }
    double* __restrict__ red = s_red[wave];
#pragma unroll
    for (int ci = 0; ci < kCols; ++ci) red[ci * kLdR + lane] = acc[ci];
    __builtin_amdgcn_wave_barrier();
    const int c = lane % kCols, g = lane / kCols;   // 4 row groups of 16 parts per column
    double v0 = 0.0, v1 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        v0 += red[c * kLdR + 16 * g + i];
        v1 += red[c * kLdR + 16 * g + i + 1];
    }
    double v = v0 + v1;
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (g == 0) part[(int64_t)pd.pad * kTile + kCols * wave + c] = v;
}

void launch_bsolve_below(const DevicePattern& P, int first, int count, const double* L, const double* x, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_bsolve_below, dim3(count), dim3(kThreads), 0, stream, P.sn, P.bsolve_below + first, P.rows, L, x,
                       P.bpart);
}

// Backward chain for ONE right-hand side: the counterpart of k_solve_chain_w.  A workgroup = eight waves = up to
// kBackGroup (2) consecutive block columns of a wide supernode (the highest one first), taken by ticket from a list
// that runs from the last block column of a supernode to its first; four waves share a block column, 16 columns
// each, lanes along the rows (coalesced), 16 running sums per lane:
//     t_c = sum_k L[k][c] x_k   over the rows k below the block.
// First the rows below the supernode's own columns (x final: ancestors), then the supernode's later block
// columns from the last one up, each as soon as its x is there -- through the armed buffer (the data is the
// flag: lane k polls the value of row k), or through LDS when it is a higher block of this workgroup; the loads
// of four 64-row pieces are in flight before the first wait.  (More block columns per workgroup would keep more of
// the diagonal-to-diagonal hand-offs in LDS, but a block column's rows are streamed by its own waves only, and
// the number of CUs that stream is what bounds the launches of few, wide supernodes: four per workgroup measured
// 16.1 vs 7.6 ms on the Flan-class input.)  Then the sums are added up across the wave, the waves' columns meet
// in LDS, and the first wave of the
// block forms x_blk = inv(L_bb)' (y - t) as a product with the inverse diagonal block (DIAG_INVERSE; staged in LDS
// when the wave starts) and publishes it (LDS for the blocks below it in this workgroup, armed buffer + x for
// everybody else).  No workgroup barrier after the start, every wait bounded.
static constexpr int kBackBlocks = kBackGroup;            // block columns per workgroup
static constexpr int kBackWaves = 8 / kBackBlocks;        // waves per block column
static constexpr int kBackCols = kTile / kBackWaves;      // columns per wave
static constexpr int kBackAhead = 64 / kBackCols;         // 64-row pieces whose loads are issued before the first wait
static constexpr int kLdRed = 65;                         // row stride of a wave's reduction buffer (doubles)
__global__ __launch_bounds__(kChainThreads, 1) void k_bsolve_chain_w(const SnDesc* __restrict__ sn,
                                                                     const PanelDesc* __restrict__ groups,
                                                                     const int32_t* __restrict__ rows,
                                                                     const double* __restrict__ L,
                                                                     const double* __restrict__ dinv,
                                                                     double* __restrict__ x, double* __restrict__ xscratch,
                                                                     int* __restrict__ info, int* __restrict__ ticket,
                                                                     int wait_bias, const double* __restrict__ part) {
    __shared__ double s_inv[kBackBlocks][kInvPacked + 1];   // column c of the inverse from row c on (as k_solve_chain_w)
    __shared__ double s_t[kBackBlocks][kTile];              // t of a block: its waves' column sums
    __shared__ double s_pub[kBackBlocks][kTile];            // x of a block, for the blocks below it in this workgroup
    __shared__ double s_red[kChainThreads / 64][kBackCols * kLdRed];   // per wave: the lanes' parts of the column sums
    __shared__ int s_sums_in[kBackBlocks], s_ready[kBackBlocks];
    __shared__ int s_task;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = wave / kBackWaves, qw = wave % kBackWaves;   // block of the group (0: the highest), wave of the block
    if (threadIdx.x == 0) s_task = atomicAdd(ticket, 1);
    if (threadIdx.x < kBackBlocks) {
        s_sums_in[threadIdx.x] = 0;
        s_ready[threadIdx.x] = 0;
    }
    __syncthreads();   // (the only one)
    const PanelDesc pd = groups[s_task];    // jb: the highest block column; row0: blocks of this workgroup (1..4)
    if (b >= pd.row0) return;
    const SnDesc D = sn[pd.sn];
    const int r = D.r, w = D.w;
    const int jb = pd.jb - b, cb = jb * kTile, wbk = min(kTile, w - cb);
    const int nbc = (w + kTile - 1) / kTile;
    const double* __restrict__ G = L + D.px;
    const int32_t* __restrict__ ri = rows + D.pi;
    auto gave_up = [&](unsigned long long t0, int& spins) {
        if ((++spins & 15) != 0) return false;
        return wall_clock64() - t0 > kSolveSpinTicks ||
               __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0;
    };
    auto wait_lds = [&](int* flag, int want) {
        const unsigned long long t0 = wall_clock64();
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want || wait_bias != 0) {
            if (gave_up(t0, spins)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        return true;
    };
    auto inv_at = [&](int c) { return lane >= c ? c * kTile - c * (c - 1) / 2 + (lane - c) : kInvPacked; };
    if (qw == 0) {
        // inverse diagonal block -> LDS: lane = row, column c of the inverse from row c on
        const double* __restrict__ src = dinv + (int64_t)(D.dslot + jb) * (kTile * kTile);
        double tmp[kTile];
#pragma unroll
        for (int c = 0; c < kTile; ++c) tmp[c] = src[c * kTile + lane];
#pragma unroll
        for (int c = 0; c < kTile; ++c) s_inv[b][inv_at(c)] = tmp[c];
    }
    // this wave's columns (clamped into the block: the sums of columns past it are not used); wave-uniform pointers,
    // every load unconditional
    const double* __restrict__ colp[kBackCols];
#pragma unroll
    for (int ci = 0; ci < kBackCols; ++ci) colp[ci] = G + (int64_t)(cb + min(kBackCols * qw + ci, wbk - 1)) * r;
    double acc[kBackCols];
#pragma unroll
    for (int ci = 0; ci < kBackCols; ++ci) acc[ci] = 0.0;
    double lv[kBackAhead][kBackCols];
    auto load_piece = [&](int u, int k) {
#pragma unroll
        for (int ci = 0; ci < kBackCols; ++ci) lv[u][ci] = colp[ci][k];
    };
    // ---- rows below the supernode's own columns: x is final.  Summed by k_bsolve_below before this launch (pd.pad =
    // the supernode's first slot + 1: one slot of 64 sums per block column and 512-row chunk) or streamed here.
    // Lane (c, g) of the final reduction adds the chunks g, g + 4, .. of its column, in that order: loaded now, used then.
    double below = 0.0;
    if (pd.pad > 0) {
        const int nch = (r - w + kBelowRows - 1) / kBelowRows;
        const double* __restrict__ mine = part + ((int64_t)(pd.pad - 1) + (int64_t)jb * nch) * kTile + kBackCols * qw + lane % kBackCols;
        for (int ch = lane / kBackCols; ch < nch; ch += 64 / kBackCols) below += mine[(int64_t)ch * kTile];
    }
    for (int k0 = w + lane; k0 < r && pd.pad <= 0; k0 += 64 * kBackAhead) {
        int xr[kBackAhead];
#pragma unroll
        for (int u = 0; u < kBackAhead; ++u) {
            const int k = min(k0 + 64 * u, r - 1);
            xr[u] = ri[k];
            load_piece(u, k);
        }
#pragma unroll
        for (int u = 0; u < kBackAhead; ++u) {
            const double xk = (k0 + 64 * u < r) ? x[xr[u]] : 0.0;
#pragma unroll
            for (int ci = 0; ci < kBackCols; ++ci) acc[ci] = fma(lv[u][ci], xk, acc[ci]);
        }
    }
    // ---- the later block columns of the supernode, last one first
    for (int I0 = nbc - 1; I0 > jb; I0 -= kBackAhead) {
#pragma unroll
        for (int u = 0; u < kBackAhead; ++u)   // (pieces past jb + 1: loaded again, not used)
            load_piece(u, min(max(I0 - u, jb + 1) * kTile + lane, w - 1));
#pragma unroll
        for (int u = 0; u < kBackAhead; ++u) {
            const int I = I0 - u;
            if (I <= jb) break;
            const int k = I * kTile + lane;
            double xk = 0.0;
            if (I <= pd.jb) {
                // a higher block of this workgroup: LDS
                if (!wait_lds(&s_ready[pd.jb - I], 1)) {
                    if (lane == 0) atomicMin(info, -1);   // (the result is wrong and reported; nobody may hang)
                    return;
                }
                xk = s_pub[pd.jb - I][lane];
            } else {
                const long long* __restrict__ src = reinterpret_cast<const long long*>(xscratch + D.c0) + k;
                const unsigned long long t0 = wall_clock64();
                int spins = 0;
                long long bits = 0;
                for (;;) {
                    bits = k < w ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                    if (__all(bits != kXArmed && wait_bias == 0)) break;
                    if (gave_up(t0, spins)) {
                        if (lane == 0) atomicMin(info, -1);
                        return;
                    }
                    if (I <= pd.jb + 2) __builtin_amdgcn_s_sleep(1);   // (the blocks right above: the critical path)
                    else __builtin_amdgcn_s_sleep(8);
                }
                xk = __longlong_as_double(bits);
            }
            if (k >= w) xk = 0.0;
#pragma unroll
            for (int ci = 0; ci < kBackCols; ++ci) acc[ci] = fma(lv[u][ci], xk, acc[ci]);
        }
    }
    // ---- column sums across the wave, through LDS: every lane parks its kBackCols parts, lane (c, g) adds up the
    // parts of 64 / (64 / kBackCols) rows for column c, two exchanges finish the column (16 xor butterflies of six
    // steps each took about 1 us of every step of the chain; a butterfly that halves the columns a lane carries
    // per step -- 17 exchanges -- measured slower still); then the block's waves meet in LDS
    {
        constexpr int kGroups = 64 / kBackCols;      // row groups a column's 64 parts are split into
        constexpr int kPer = 64 / kGroups;           // parts per group (= kBackCols)
        double* __restrict__ red = s_red[wave];
#pragma unroll
        for (int ci = 0; ci < kBackCols; ++ci) red[ci * kLdRed + lane] = acc[ci];
        __builtin_amdgcn_wave_barrier();
        const int c = lane % kBackCols, g = lane / kBackCols;
        double v0 = 0.0, v1 = 0.0;
#pragma unroll
        for (int i = 0; i < kPer; i += 2) {
            v0 += red[c * kLdRed + kPer * g + i];
            v1 += red[c * kLdRed + kPer * g + i + 1];
        }
        double v = v0 + v1 + below;
#pragma unroll
        for (int off = kBackCols; off < 64; off <<= 1) v += __shfl_xor(v, off);
        if (g == 0) s_t[b][kBackCols * qw + c] = v;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_fetch_add(&s_sums_in[b], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (qw != 0) return;
    if (!wait_lds(&s_sums_in[b], kTile / kBackCols)) {
        if (lane == 0) atomicMin(info, -1);
        return;
    }
    // ---- x_blk = inv(L_bb)' (y - t): x_c = sum_{k >= c} inv(L_bb)[k][c] (y_k - t_k); lane c walks its column of
    // the packed inverse
    const double tk = (lane < wbk) ? x[D.c0 + cb + lane] - s_t[b][lane] : 0.0;
    __builtin_amdgcn_wave_barrier();
    s_t[b][lane] = tk;
    __builtin_amdgcn_wave_barrier();
    double s0 = 0.0, s1 = 0.0;
    {
        // column `lane` of the inverse: rows lane..63 are contiguous from lane*64 - lane(lane-1)/2
        const double* __restrict__ col = s_inv[b] + lane * kTile - lane * (lane - 1) / 2 - lane;   // col[k], k >= lane
        for (int k = lane; k + 1 < kTile; k += 2) {
            s0 = fma(col[k], s_t[b][k], s0);
            s1 = fma(col[k + 1], s_t[b][k + 1], s1);
        }
        if (((kTile - lane) & 1) != 0) s0 = fma(col[kTile - 1], s_t[b][kTile - 1], s0);
    }
    const double xi = (lane < wbk) ? s0 + s1 : 0.0;
    if (b + 1 < pd.row0) {
        s_pub[b][lane] = xi;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(&s_ready[b], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (lane < wbk) {
        __hip_atomic_store(&xscratch[D.c0 + cb + lane], unarmed(xi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x[D.c0 + cb + lane] = xi;
    }
}

void launch_bsolve_chain_w(const DevicePattern& P, int first, int count, const double* L, const double* dinv,
                           double* x, double* xscratch, int ticket, int wait_bias, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_bsolve_chain_w, dim3(count), dim3(kChainThreads), 0, stream, P.sn, P.bsolve_pairs + first,
                       P.rows, L, dinv, x, xscratch, P.sinfo, P.stickets + ticket, wait_bias, P.bpart);
}

// mode: Launch::fused -- 0: one workgroup per block, 1: chain launch (tickets), 2: subtree launch (`first` counts
// (begin, end) pairs of bsolve_ranges, which index the whole block list); tiny: Launch::early (0 | width class 1 | 2)
void launch_bsolve_block(const DevicePattern& P, int first, int count, const double* L, const double* dinv,
                         double* x, double* xscratch, int nrhs, int ldx, int mode, int tiny, int ticket,
                         int wait_bias, hipStream_t stream) {
    if (count <= 0) return;
    const int chain = mode == 1;
    const PanelDesc* pds = mode == 2 ? P.bsolve_blocks : P.bsolve_blocks + first;
    const int32_t* ranges = mode == 2 ? P.bsolve_ranges + 2 * first : nullptr;
    if (tiny == 1 && nrhs >= bmrhs_min()) {   // many right-hand sides: 64 per pass, matrix cores, one wave per supernode
        const dim3 grid(count, std::min(kPassLanes, (nrhs + kRhsM - 1) / kRhsM));
        hipLaunchKernelGGL(k_bsolve_tiny_mrhs, grid, dim3(64), 0, stream, P.sn, pds, ranges, P.rows, L, x, nrhs, ldx);
        return;
    }
    if (tiny && nrhs < bmrhs_min()) {   // width class kTinyWidth (1) or kTinyWidth2 (2), a subtree launch or a level's launch: one wave each
        const dim3 grid(count, std::min(kPassLanes, nrhs));
        if (tiny == 1)
            hipLaunchKernelGGL(k_bsolve_tiny<kTinyWidth>, grid, dim3(64), 0, stream, P.sn, pds, ranges, P.rows, L, x, nrhs,
                               ldx);
        else
            hipLaunchKernelGGL(k_bsolve_tiny<kTinyWidth2>, grid, dim3(64), 0, stream, P.sn, pds, ranges, P.rows, L, x, nrhs,
                               ldx);
        return;
    }
    // launches of few blocks (the top of the tree, small inputs: a chain of hand-offs, not a stream of L) keep the
    // light kernel -- 4 right-hand sides per workgroup, up to 8 workgroups per block side by side (nd24k-class, 16
    // right-hand sides: 1.02 ms against 2.06 with the kernel below everywhere)
    static const bool wide_only = [] {
        const char* e = std::getenv("PARSY_BMRHS_WIDE_ONLY");
        return !(e && e[0] == '0');
    }();
    // (round 5) chain launches of more than 16 right-hand sides: k_bsolve_chain_mrhs from kBChainMinBlocks block columns on
    // (PARSY_BCHAIN_MIN_BLOCKS), 64 right-hand sides per pass
    static const int bchain_min = [] {
        const char* e = std::getenv("PARSY_BCHAIN_MIN_BLOCKS");
        return e && *e ? std::atoi(e) : kBChainMinBlocks;
    }();
    if (chain && nrhs > 16 && nrhs >= bmrhs_min() && count >= bchain_min) {
        const int mlanes = std::min(kPassLanes, (nrhs + kRhsM - 1) / kRhsM);
        hipLaunchKernelGGL(k_bsolve_chain_mrhs, dim3(count * mlanes), dim3(kThreads), 0, stream, P.sn, pds, P.rows, L, x, xscratch, nrhs,
                           ldx, P.sinfo, P.stickets + ticket, wait_bias, count, dinv);
        return;
    }
    if (nrhs >= bmrhs_min() && (!wide_only || count >= kMrhsWideBlocks)) {   // 64 right-hand sides per pass over L, products on the matrix cores
        // launches of few blocks (the top of the tree, small inputs) take 16 right-hand sides per pass in up to 8
        // workgroups per block side by side; the others 64 per pass (L read once per 64)
        const bool wide = count >= kMrhsWideBlocks && nrhs > 16;
        const int per = wide ? kRhsM : 16;
        const int mlanes = std::min(kPassLanes, (nrhs + per - 1) / per);
        const dim3 mgrid = chain ? dim3(count * mlanes) : dim3(count, mlanes);
        if (wide)
            hipLaunchKernelGGL(k_bsolve_block_mrhs<4>, mgrid, dim3(kThreads), 0, stream, P.sn, pds, P.rows, L, x, xscratch,
                               nrhs, ldx, chain, P.sinfo, P.stickets + ticket, wait_bias, count, ranges, dinv);
        else
            hipLaunchKernelGGL(k_bsolve_block_mrhs<1>, mgrid, dim3(kThreads), 0, stream, P.sn, pds, P.rows, L, x, xscratch,
                               nrhs, ldx, chain, P.sinfo, P.stickets + ticket, wait_bias, count, ranges, dinv);
        return;
    }
    const int lanes = nrhs == 1 ? 1 : std::min(kPassLanes, (nrhs + 3) / 4);
    const dim3 grid = chain ? dim3(count * lanes) : dim3(count, lanes);
    if (nrhs == 1)
        hipLaunchKernelGGL(k_bsolve_block<1>, grid, dim3(kThreads), 0, stream, P.sn, pds, P.rows, L, x, xscratch,
                           nrhs, ldx, chain, P.sinfo, P.stickets + ticket, wait_bias, count, ranges, dinv);
    else
        hipLaunchKernelGGL(k_bsolve_block<4>, grid, dim3(kThreads), 0, stream, P.sn, pds, P.rows, L, x, xscratch,
                           nrhs, ldx, chain, P.sinfo, P.stickets + ticket, wait_bias, count, ranges, dinv);
}

// SOLVE_FIXUP: solved blocks of the wide supernodes go from scratch into x.
__global__ __launch_bounds__(kThreads) void k_solve_fixup(const SnDesc* __restrict__ sn,
                                                          const int32_t* __restrict__ list,
                                                          double* __restrict__ x,
                                                          const double* __restrict__ xscratch,
                                                          int nrhs, int ldx) {
    const SnDesc D = sn[list[blockIdx.x]];
    for (int q = blockIdx.y; q < nrhs; q += gridDim.y)
        for (int c = threadIdx.x; c < D.w; c += kThreads)
            x[(int64_t)q * ldx + D.c0 + c] = xscratch[(int64_t)q * ldx + D.c0 + c];
}

void launch_solve_fixup(const DevicePattern& P, int first, int count, double* x,
                        const double* xscratch, int nrhs, int ldx, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_solve_fixup, dim3(count, min(nrhs, 64)), dim3(kThreads), 0, stream, P.sn,
                       P.solve_fix_list + first, x, xscratch, nrhs, ldx);
}

// b = L * 1 on the stored structure (the reference's rhsInitBlocked, common/Util.h:277-288: the right-hand
// side its triangularTest solves, so that x = 1): every stored entry of a panel row is added to b[row].
// One thread per panel row (lanes along the rows: coalesced column by column), rows of a tall panel
// spread over gridDim.y workgroups.  b must be zero on entry.
__global__ __launch_bounds__(kThreads) void k_rhs_ones(const SnDesc* __restrict__ sn,
                                                       const int32_t* __restrict__ rows,
                                                       const double* __restrict__ L, double* __restrict__ b) {
    const SnDesc D = sn[blockIdx.x];
    const double* __restrict__ G = L + D.px;
    for (int i = blockIdx.y * kThreads + threadIdx.x; i < D.r; i += gridDim.y * kThreads) {
        const int cmax = min(D.w, i + 1);  // the diagonal block is lower triangular
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int c = 0;
        for (; c + 4 <= cmax; c += 4) {
            s0 += G[(int64_t)c * D.r + i];
            s1 += G[(int64_t)(c + 1) * D.r + i];
            s2 += G[(int64_t)(c + 2) * D.r + i];
            s3 += G[(int64_t)(c + 3) * D.r + i];
        }
        for (; c < cmax; ++c) s0 += G[(int64_t)c * D.r + i];
        atomicAdd(&b[rows[D.pi + i]], (s0 + s1) + (s2 + s3));
    }
}

void launch_rhs_ones(const DevicePattern& P, int nsuper, int max_rows, const double* L, double* b,
                     hipStream_t stream) {
    if (nsuper <= 0) return;
    const int ny = std::max(1, std::min(16, (max_rows + kThreads - 1) / kThreads));
    hipLaunchKernelGGL(k_rhs_ones, dim3(nsuper, ny), dim3(kThreads), 0, stream, P.sn, P.rows, L, b);
}

// Copy contiguous runs of doubles between two device buffers: dst[dst_off[q] + i] = src[src_off[q] + i],
// i < len[q].  The multi-GPU exchange packs with it the rows of a subtree's panels that the root part
// reads (the tail of every panel column) and unpacks them on the receiving rank.
__global__ __launch_bounds__(64) void k_copy_segments(double* __restrict__ dst, const double* __restrict__ src,
                                                      const int64_t* __restrict__ dst_off,
                                                      const int64_t* __restrict__ src_off,
                                                      const int32_t* __restrict__ len, int64_t nseg) {
    for (int64_t q = blockIdx.x; q < nseg; q += gridDim.x) {
        const double* __restrict__ s = src + src_off[q];
        double* __restrict__ d = dst + dst_off[q];
        const int n = len[q];
        for (int i = threadIdx.x; i < n; i += 64) d[i] = s[i];
    }
}

void launch_copy_segments(double* dst, const double* src, const int64_t* dst_off, const int64_t* src_off,
                          const int32_t* len, int64_t nseg, hipStream_t stream) {
    if (nseg <= 0) return;
    const unsigned grid = (unsigned)std::min<int64_t>(nseg, 1 << 20);
    hipLaunchKernelGGL(k_copy_segments, dim3(grid), dim3(64), 0, stream, dst, src, dst_off, src_off, len, nseg);
}

}  // namespace parsy
