// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (and v_fma_f64) on gfx950.
// The microarchitecture guide has no FP64 row; this measures the number bench.py's
// roofline is priced against.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products per instruction (512 flops per wave)
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(double* out, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_fma(double* out, int iters) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = threadIdx.x * 1e-9 + 1.0, b = blockIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    double* d;
    hipMalloc(&d, 4096 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    auto run = [&](const char* name, auto launch, double flops_per_thread_iter, int blocks) {
        launch(blocks, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch(blocks, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double fl = flops_per_thread_iter * iters * 256.0 * blocks;
        printf("%-28s blocks=%4d  %.3f ms  %.2f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
    };
    // one MFMA 16x16x4 = 2*16*16*4 = 2048 flops per wave = 32 flops per lane
    for (int blocks : {256, 512, 1024, 2048}) {
        run("mfma_f64 1 acc", [&](int b, int it) { hipLaunchKernelGGL(k_mfma<1>, dim3(b), dim3(256), 0, 0, d, it); }, 32.0 * 1, blocks);
        run("mfma_f64 4 acc", [&](int b, int it) { hipLaunchKernelGGL(k_mfma<4>, dim3(b), dim3(256), 0, 0, d, it); }, 32.0 * 4, blocks);
        run("mfma_f64 8 acc", [&](int b, int it) { hipLaunchKernelGGL(k_mfma<8>, dim3(b), dim3(256), 0, 0, d, it); }, 32.0 * 8, blocks);
        run("mfma_f64_4x4x4 8 acc", [&](int b, int it) { hipLaunchKernelGGL(k_mfma4<8>, dim3(b), dim3(256), 0, 0, d, it); }, 8.0 * 8, blocks);
        run("mfma_f64_4x4x4 16 acc", [&](int b, int it) { hipLaunchKernelGGL(k_mfma4<16>, dim3(b), dim3(256), 0, 0, d, it); }, 8.0 * 16, blocks);
        run("v_fma_f64 16 chains", [&](int b, int it) { hipLaunchKernelGGL(k_fma, dim3(b), dim3(256), 0, 0, d, it); }, 2.0 * 16, blocks);
    }
    return 0;
}
