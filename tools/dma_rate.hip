// Microbenchmark: how fast ONE compute unit pulls data that misses its L2 (a stream through a buffer far larger than
// the caches) into LDS -- by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, `depth` instructions in
// flight per wave) and by register loads (global_load_dwordx4, same depth, + ds_write_b128) -- for 2 / 4 / 8 / 16
// waves per CU.  What bounds k_chol_dense / k_chol_big is this rate per CU, not the device's aggregate bandwidth.
// Build: hipcc --offload-arch=gfx950 -O3 tools/dma_rate.hip -o tools/dma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void glds16(const double* g, double* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
// every wave reads `n` 1-KiB columns, `stride` doubles apart, starting at its own offset
template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void k_rate(const double* __restrict__ src, double* out, int n, long stride, long wave_span, int active_mod) {
    extern __shared__ double S[];
    if (blockIdx.x % active_mod) return;   // (few active CUs: what ONE compute unit can pull when the device is idle)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    double* mine = S + wave * (DEPTH * 128);
    const double* p = src + ((long)blockIdx.x * nw + wave) * wave_span + 2 * lane;
    double2 acc = {0, 0};
    if (MODE == 0) {
        for (int i = 0; i < n; i += DEPTH) {
#pragma unroll
            for (int c = 0; c < DEPTH; ++c) glds16(p + (long)(i + c) * stride, mine + c * 128);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc.x = mine[lane];
    } else {
        double2 v[DEPTH];
#pragma unroll
        for (int c = 0; c < DEPTH; ++c) v[c] = *reinterpret_cast<const double2*>(p + (long)c * stride);
        for (int i = DEPTH; i < n; i += DEPTH) {
#pragma unroll
            for (int c = 0; c < DEPTH; ++c) {
                *reinterpret_cast<double2*>(mine + c * 128 + 2 * lane) = v[c];
                v[c] = *reinterpret_cast<const double2*>(p + (long)(i + c) * stride);
            }
        }
#pragma unroll
        for (int c = 0; c < DEPTH; ++c) acc.x += v[c].x + v[c].y;
        acc.x += mine[lane];
    }
    if (acc.x == 12345.678) out[0] = acc.x;
}
template <int MODE, int DEPTH>
static void run(const double* src, double* out, int waves, int n, long stride, long span, hipEvent_t e0, hipEvent_t e1, const char* what, int active_mod) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rate<MODE, DEPTH>), dim3(256), dim3(64 * waves), waves * DEPTH * 1024, 0, src, out, n, stride, span, active_mod);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ncu = 256.0 / active_mod;
    const double bytes = ncu * waves * n * 1024.0;
    printf("%3.0f CUs  %-18s depth %2d  %2d waves/CU: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", ncu, what, DEPTH, waves, best,
           bytes / best / 1e9, bytes / ncu / (best * 1e-3 * 2.4e9));
}
int main() {
    // 256 CUs x 16 waves x 2048 columns of 1 KiB, columns 8 KiB apart (a panel with 1024 rows): 64 GiB of address space
    // would be needed for distinct data per wave -- instead every wave walks its own 16-MiB window (2048 columns x
    // 8 KiB): 256 x 16 x 16 MiB = 64 GiB ... too much; use 4-MiB windows (512 columns) and n = 512: 16 GiB.
    const int n = 504;   // (a multiple of the depths)
    const long stride = 1024, span = (long)n * stride;        // doubles
    const size_t total = (size_t)256 * 16 * span + 4096;
    double* src; double* out;
    if (hipMalloc(&src, total * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&out, 64);
    (void)hipMemset(src, 0, total * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mod : {32, 4, 1})
        for (int waves : {2, 4, 8}) {
            run<0, 4>(src, out, waves, n, stride, span, e0, e1, "LDS-DMA dwordx4", mod);
            run<0, 12>(src, out, waves, n, stride, span, e0, e1, "LDS-DMA dwordx4", mod);
            run<1, 4>(src, out, waves, n, stride, span, e0, e1, "load x4 + ds_write", mod);
            run<1, 12>(src, out, waves, n, stride, span, e0, e1, "load x4 + ds_write", mod);
        }
    return 0;
}
