// Microbenchmark: what the inner loop of k_chol_big can sustain with v_mfma_f64_4x4x4_4b_f64 at 4 waves per SIMD
// (512-thread workgroups, 2 per CU).  MODE 0: one a / b register; 1: 4 a x 8 b registers, 32 accumulators (the
// kernel's k step); 2: + the DPP rotations of the b operands every step; 3: + the operands read from LDS every step;
// 4: as 3 with the 16x16x4 form (8 MFMAs per step).  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma4_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double ror(double v, int ctrl) {
    const long long bits = __builtin_bit_cast(long long, v);
    int lo = (int)bits, hi = (int)(bits >> 32);
    if (ctrl == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x124, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x124, 0xf, 0xf, false); }
    if (ctrl == 2) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xf, 0xf, false); }
    if (ctrl == 3) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x12c, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x12c, 0xf, 0xf, false); }
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int MODE>
__global__ __launch_bounds__(512, 4) void k_loop(double* out, int iters) {
    __shared__ double lds[2 * 16 * 144];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 2 * 16 * 144; i += 512) lds[i] = 1e-3 * i;
    __syncthreads();
    double acc[2][4][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) acc[a][b][c] = 0;
    double4_t acc16[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) acc16[a][b] = double4_t{0, 0, 0, 0};
    double rv[4], cv[2];
    for (int f = 0; f < 4; ++f) rv[f] = 1.0 + tid * 1e-3 + f;
    for (int f = 0; f < 2; ++f) cv[f] = 2.0 + tid * 1e-3 + f;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 3) {
            const double* Rb = &lds[((it & 3) * 4 + kq) * 144 + l15];
            const double* Cb = &lds[16 * 144 + ((it & 3) * 4 + kq) * 144 + l15];
#pragma unroll
            for (int f = 0; f < 4; ++f) rv[f] = Rb[16 * f];
#pragma unroll
            for (int f = 0; f < 2; ++f) cv[f] = Cb[16 * f];
        }
        if (MODE == 4) {
#pragma unroll
            for (int fc = 0; fc < 2; ++fc)
#pragma unroll
                for (int fr = 0; fr < 4; ++fr)
                    acc16[fc][fr] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv[fc], rv[fr], acc16[fc][fr], 0, 0, 0);
            continue;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int fc = 0; fc < 2; ++fc) {
                double cq = MODE == 0 ? cv[0] : (MODE == 1 ? cv[fc] + 0.0 : ror(cv[fc], q));
#pragma unroll
                for (int fr = 0; fr < 4; ++fr)
                    acc[fc][fr][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(MODE == 0 ? rv[0] : rv[fr], cq, acc[fc][fr][q], 0, 0, 0);
            }
    }
    double s = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) s += acc[a][b][c] + acc16[a][b][c];
    out[blockIdx.x * 512 + tid] = s;
}
int main() {
    double* d;
    (void)hipMalloc(&d, 4096 * 512 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    int iters = 20000;
    auto run = [&](const char* name, auto kern, int blocks) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = 16384.0 * iters * 8.0 * blocks;  // 16384 flops per wave and step (64 x 32 x 4 x 2)
        printf("%-34s blocks=%4d  %.3f ms  %.2f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
    };
    for (int blocks : {512, 1024}) {
        run("4x4x4 same registers", k_loop<0>, blocks);
        run("4x4x4 4a x 8b registers", k_loop<1>, blocks);
        run("4x4x4 + DPP rotations", k_loop<2>, blocks);
        run("4x4x4 + DPP + LDS operands", k_loop<3>, blocks);
        run("16x16x4 + LDS operands", k_loop<4>, blocks);
    }
    // sustained: ~0.3 s per measurement (clock management shows)
    iters = 400000;
    run("SUSTAINED 4x4x4 4a x 8b registers", k_loop<1>, 512);
    run("SUSTAINED 4x4x4 + DPP + LDS", k_loop<3>, 512);
    run("SUSTAINED 16x16x4 + LDS operands", k_loop<4>, 512);
    run("SUSTAINED 16x16x4 + LDS operands", k_loop<4>, 512);
    return 0;
}
