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

// Forward substitution of the staged block: xs[c][q], c < w, q < nq, with the
// lower-triangular block Dg (column-major, ld kLdDiag). All threads of the
// workgroup participate; ends synchronised.
__device__ __forceinline__ void block_forward_solve(const double* Dg, double (*xs)[kRhs], int w,
                                                    int nq, int tid) {
    for (int c = 0; c < w; ++c) {
        if (tid < nq) xs[c][tid] = xs[c][tid] / Dg[c * kLdDiag + c];
        __syncthreads();
        const int rem = w - c - 1;
        for (int e = tid; e < rem * nq; e += kThreads) {
            const int q = e / rem, i = c + 1 + (e - q * rem);
            xs[i][q] = fma(-Dg[c * kLdDiag + i], xs[c][q], xs[i][q]);
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
    __shared__ double xs[kTile][kRhs];
    const int tid = threadIdx.x;
    const SnDesc D = sn[list[blockIdx.x]];
    const int r = D.r, w = D.w;
    const double* __restrict__ G = L + D.px;
    const int32_t* __restrict__ ri = rows + D.pi;

    for (int e = tid; e < w * w; e += kThreads) {
        const int c = e / w, i = e - c * w;
        if (i >= c) Dg[c * kLdDiag + i] = G[(int64_t)c * r + i];
    }
    for (int q0 = 0; q0 < nrhs; q0 += kRhs) {
        const int nq = min(kRhs, nrhs - q0);
        __syncthreads();
        for (int e = tid; e < w * nq; e += kThreads) {
            const int q = e / w, c = e - q * w;
            xs[c][q] = x[(int64_t)(q0 + q) * ldx + D.c0 + c];
        }
        __syncthreads();
        block_forward_solve(Dg, xs, w, nq, tid);
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
    __shared__ double xs[kTile][kRhs];
    const int tid = threadIdx.x;
    const PanelDesc pd = pds[blockIdx.x];
    const SnDesc D = sn[pd.sn];
    const int r = D.r, cb = pd.jb * kTile, wbk = min(kTile, D.w - cb);
    const double* __restrict__ G = L + D.px;
    const int32_t* __restrict__ ri = rows + D.pi;

    for (int e = tid; e < kTile * kTile; e += kThreads) {
        const int c = e >> 6, i = e & 63;
        if (c < wbk && i < wbk && i >= c) Dg[c * kLdDiag + i] = G[(int64_t)(cb + c) * r + cb + i];
    }
    for (int q0 = 0; q0 < nrhs; q0 += kRhs) {
        const int nq = min(kRhs, nrhs - q0);
        __syncthreads();
        for (int e = tid; e < wbk * nq; e += kThreads) {
            const int q = e / wbk, c = e - q * wbk;
            xs[c][q] = x[(int64_t)(q0 + q) * ldx + D.c0 + cb + c];
        }
        __syncthreads();
        block_forward_solve(Dg, xs, wbk, nq, tid);
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
            for (int c = 0; c < wbk; ++c) {
                const double lv = G[(int64_t)(cb + c) * r + k];
#pragma unroll
                for (int q = 0; q < kRhs; ++q) acc[q] = fma(lv, xs[c][q], acc[q]);
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
