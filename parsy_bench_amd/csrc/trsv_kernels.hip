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

// SOLVE_CHAIN: the whole block-column chain of a wide supernode in ONE launch.  Workgroup c
// owns rows [256c, 256c+256) of the panel and keeps the running update of its rows in
// registers (pull form: no atomics inside the supernode).  For block column jb the owner of
// rows [64jb, 64jb+64) solves the diagonal block on its up-to-date rows and publishes x_jb
// (agent-scope release + flag, cdna_hip_programming.md Guideline 16); workgroups with rows
// below wait for the flag (bounded), read x_jb and update their rows.  Rows below the
// supernode's own columns are scattered once, at the end, with atomics (other supernodes of
// the level update the same ancestor rows).  All workgroups of a launch are resident at
// once (the host caps their number), so the waits cannot starve the workgroup they wait for.
__global__ __launch_bounds__(kThreads) void k_solve_chain(const SnDesc* __restrict__ sn,
                                                          const PanelDesc* __restrict__ pds,
                                                          const int32_t* __restrict__ rows,
                                                          const double* __restrict__ L,
                                                          double* __restrict__ x,
                                                          double* __restrict__ xscratch, int nrhs,
                                                          int ldx, int* __restrict__ flags, int epoch0,
                                                          int* __restrict__ info) {
    __shared__ double Dg[kTile * kLdDiag];
    __shared__ double invd[kTile];
    __shared__ double xs[kTile][kRhs];
    __shared__ int32_t s_ok;
    const int tid = threadIdx.x;
    const PanelDesc pd = pds[blockIdx.x];
    const SnDesc D = sn[pd.sn];
    const int r = D.r, w = D.w, chunk = pd.jb, row0 = pd.row0;
    const int nbc = (w + kTile - 1) / kTile;
    const double* __restrict__ G = L + D.px;
    const int k = row0 + tid;          // this thread's panel row
    const bool kv = k < r;
    const bool kdiag = kv && k < w;    // row inside the supernode's own columns
    int pass = 0;
    for (int q0 = 0; q0 < nrhs; q0 += kRhs, ++pass) {
        const int nq = min(kRhs, nrhs - q0);
        const int epoch = epoch0 + pass;
        double xv[kRhs], acc[kRhs];
#pragma unroll
        for (int q = 0; q < kRhs; ++q) {
            acc[q] = 0.0;
            xv[q] = (kdiag && q < nq) ? x[(int64_t)(q0 + q) * ldx + D.c0 + k] : 0.0;
        }
        for (int jb = 0; jb < nbc; ++jb) {
            const int cb = jb * kTile, wbk = min(kTile, w - cb);
            const int owner = cb / kSolveRows;
            if (chunk < owner) break;  // no rows at or below this block column
            __syncthreads();           // xs / Dg of the previous block column are free
            if (chunk == owner) {
                // diagonal block (identity padded) -> LDS, up-to-date rows of the block -> xs
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
                const int lr = k - cb;  // row inside the block (0..63) for the 64 threads that hold it
                if (lr >= 0 && lr < kTile) {
#pragma unroll
                    for (int q = 0; q < kRhs; ++q) xs[lr][q] = (lr < wbk) ? xv[q] - acc[q] : 0.0;
                }
                __syncthreads();
                block_solve_inv16(Dg, invd, xs, wbk, nq, tid);
                // publish x_jb.  The hand-off payload is small, so it travels as 8-byte agent-scope
                // atomics on both sides (a valid form of Guideline 16 that needs neither the L2
                // write-back of a release fence nor the L1 invalidate of an acquire); x itself gets
                // the final value with plain stores (nobody reads it inside this launch).
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
            } else {
                if (tid == 0) {
                    const unsigned long long t0 = wall_clock64();
                    int ok = 1;
                    // epochs only grow: a later pass of this solve may already have raised the flag
                    while (__hip_atomic_load(&flags[D.dslot + jb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch < 0) {
                        if (wall_clock64() - t0 > 20000000ull) {  // 0.2 s at 100 MHz: give up, report
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
            // (all 64 loads of the row in flight at once: one memory latency per block column)
            if (kv && k >= cb + wbk) {
                double lv[kTile];
#pragma unroll
                for (int c = 0; c < kTile; ++c) lv[c] = (c < wbk) ? G[(int64_t)(cb + c) * r + k] : 0.0;
#pragma unroll
                for (int c = 0; c < kTile; ++c)
#pragma unroll
                    for (int q = 0; q < kRhs; ++q) acc[q] = fma(lv[c], xs[c][q], acc[q]);
            }
        }
        if (kv && !kdiag) {
            const int row = rows[D.pi + k];
#pragma unroll
            for (int q = 0; q < kRhs; ++q)
                if (q < nq) atomicAdd(&x[(int64_t)(q0 + q) * ldx + row], -acc[q]);
        }
    }
}

void launch_solve_chain(const DevicePattern& P, int first, int count, const double* L, double* x,
                        double* xscratch, int nrhs, int ldx, int epoch0, hipStream_t stream) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_solve_chain, dim3(count), dim3(kThreads), 0, stream, P.sn, P.solve_panels + first,
                       P.rows, L, x, xscratch, nrhs, ldx, P.flags, epoch0, P.info);
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
