// HIP kernels of the level-scheduled BCSC forward solve L X = B for gfx950.
//
// Numeric contract (reference triangularSolve/Triangular_BCSC.h:139-157): per
// supernode, dense forward solve of x[cols] with the diagonal block (non-unit
// diagonal, division as triangularSolve/BLAS.h:8), tmp = L21 * x[cols], then
// x[Li[l]] -= tmp[k] with an atomic (the reference uses `omp atomic`, so its
// rounding order is schedule-dependent too).  X is n x nrhs column-major.
// HBM-bound: every stored L value is read once per pass of kRhs right-hand sides.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace parsy {

static constexpr int kThreads = 256;
static constexpr int kLdDiag = kTile + 1;
static constexpr int kRhs = 8;  // right-hand sides carried per pass over a panel

// Forward solve of the staged block xs[c][q] (c < w <= 64, q < nq) with the lower-triangular
// block Dg (column-major, ld kLdDiag; entries outside w x w must be an identity).  Blocked by
// 16: the four 16x16 diagonal sub-blocks are inverted once (one column per thread, written
// transposed into the unused strict upper triangle of the sub-block), then per sub-block
//   y_b = inv(L_bb) x_b   and   x_rest -= L(rest, b) y_b
// -- 2 barriers per 16 columns instead of 2 per column.  All threads participate; ends
// synchronised.  (triangularSolve/BLAS.h:8 divides by the diagonal; so does the inversion.)
__device__ __forceinline__ void block_solve_apply16(const double* Dg, const double* invd,
                                                    double (*xs)[kRhs], int w, int nq, int tid);

__device__ __forceinline__ void block_solve_inv16(double* Dg, double* invd, double (*xs)[kRhs], int w,
                                                  int nq, int tid) {
    if (tid < kTile) invd[tid] = 1.0 / Dg[tid * kLdDiag + tid];
    __syncthreads();
    if (tid < kTile && (tid & ~15) < w) {
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
        __builtin_amdgcn_s_waitcnt(0);  // every read of the sub-block precedes the in-place writes
#pragma unroll
        for (int rr = 1; rr < 16; ++rr)
            if (rr > c) Dg[(b16 + rr) * kLdDiag + b16 + c] = y[rr];
    }
    __syncthreads();
    block_solve_apply16(Dg, invd, xs, w, nq, tid);
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

// SOLVE_SMALL: one workgroup per supernode of width <= 64.
__global__ __launch_bounds__(kThreads) void k_solve_small(const SnDesc* __restrict__ sn,
                                                          const int32_t* __restrict__ list,
                                                          const int32_t* __restrict__ rows,
                                                          const double* __restrict__ L,
                                                          double* __restrict__ x, int nrhs, int ldx) {
    __shared__ double Dg[kTile * kLdDiag];
    __shared__ double invd[kTile];
    __shared__ double xs[kTile][kRhs];
    const int tid = threadIdx.x;
    const SnDesc D = sn[list[blockIdx.x]];
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
    for (int q0 = 0; q0 < nrhs; q0 += kRhs) {
        const int nq = min(kRhs, nrhs - q0);
        __syncthreads();
        for (int e = tid; e < wpad * nq; e += kThreads) {
            const int q = e / wpad, c = e - q * wpad;
            xs[c][q] = (c < w) ? x[(int64_t)(q0 + q) * ldx + D.c0 + c] : 0.0;
        }
        __syncthreads();
        if (q0 == 0) block_solve_inv16(Dg, invd, xs, w, nq, tid);
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

void launch_solve_small(const DevicePattern& P, int first, int count, const double* L, double* x,
                        int nrhs, int ldx, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_solve_small, dim3(count), dim3(kThreads), 0, stream, P.sn,
                       P.solve_small_list + first, P.rows, L, x, nrhs, ldx);
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

}  // namespace parsy
