// Diagnostic (not part of the product): throughput of the forward solve's x[rows] -= updates for 64 right-hand sides
// in the two candidate layouts of X (n = 524 288 rows, 64 right-hand sides, 53 M atomics = parabolic_fem-class):
//   mode 0  X column-major (x[q * n + row]): one wave instruction = 16 consecutive rows x 4 right-hand sides
//           (the result layout of v_mfma_f64_16x16x4_f64: what k_solve_small_mrhs issues)
//   mode 1  X right-hand-side-contiguous (x[row * 64 + q]): one wave instruction = one row x 64 right-hand sides
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_layout_bench.hip -o tools/atomic_layout_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kN = 1 << 19, kRhs = 64;
constexpr long kTotal = 53L << 20;   // lane-level atomics per run

template <int MODE>
__global__ __launch_bounds__(256) void k_atomics(double* __restrict__ x, int iters) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned h = wid * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        if (MODE == 0) {
            const int row = (int)((h >> 8) % (kN / 16)) * 16 + l15;   // < kN
            const int q = (int)((h >> 4) & 15) * 4 + kq;               // < 64
            atomicAdd(&x[(long)q * kN + row], 1.0);
        } else {
            const int row = (int)((h >> 8) % kN);                      // < kN
            atomicAdd(&x[(long)row * kRhs + lane], 1.0);
        }
    }
}

int main() {
    double* x = nullptr;
    const size_t bytes = (size_t)kN * kRhs * sizeof(double);
    if (hipMalloc(&x, bytes) != hipSuccess || hipMemset(x, 0, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int wgs = 4096, waves = wgs * 4;
    const int iters = (int)(kTotal / 64 / waves);
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k_atomics<0>, dim3(wgs), dim3(256), 0, 0, x, iters);
            else hipLaunchKernelGGL(k_atomics<1>, dim3(wgs), dim3(256), 0, 0, x, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("mode %d: %ld lane atomics (%d wave instructions of 64) in %.3f ms = %.1f G atomics/s\n", mode,
               (long)iters * waves * 64, iters * waves, best, (double)iters * waves * 64 / best / 1e6);
        if (hipGetLastError() != hipSuccess) return 2;
    }
    return 0;
}
