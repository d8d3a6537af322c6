// Diagnostic (not part of the product): bandwidth of reading a column-major panel in register tiles of 64 doubles
// per lane, for three tile shapes (what a wave of the solve kernels has in flight):
//   mode 0: 64 rows x 64 columns  (one 512-byte run per load instruction)
//   mode 1: 128 rows x 32 columns (one 1-KiB run per load instruction, 16 bytes per lane)
//   mode 2: 256 rows x 16 columns (two 1-KiB runs per column)
// and for several numbers of workgroups (how many CUs it takes to reach the bandwidth).
// Build: hipcc --offload-arch=gfx950 -O3 tools/panel_read_bench.hip -o tools/panel_read_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr long R = 16384, C = 8192;          // 1 GiB of doubles
constexpr long kTiles = (R / 64) * (C / 64);  // 32768 tiles of 4096 doubles in every mode

template <int MODE>
__global__ __launch_bounds__(256) void k_read(const double* __restrict__ A, double* __restrict__ out, int tiles_per_wave,
                                              int nwaves) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);   // < nwaves by construction of the grid
    double s0 = 0, s1 = 0;
    for (int t = 0; t < tiles_per_wave; ++t) {
        const long tile = (long)wid + (long)t * nwaves;
        if (tile >= kTiles) break;
        if (MODE == 0) {
            const long nrb = R / 64, rb = tile % nrb, cb = tile / nrb;
            double v[64];
#pragma unroll
            for (int c = 0; c < 64; ++c) v[c] = A[(cb * 64 + c) * R + rb * 64 + lane];
#pragma unroll
            for (int c = 0; c < 64; c += 2) { s0 += v[c]; s1 += v[c + 1]; }
        } else if (MODE == 1) {
            const long nrb = R / 128, rb = tile % nrb, cb = tile / nrb;
            const double2_t* A2 = reinterpret_cast<const double2_t*>(A);
            double2_t v[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) v[c] = A2[((cb * 32 + c) * R + rb * 128) / 2 + lane];
#pragma unroll
            for (int c = 0; c < 32; ++c) { s0 += v[c][0]; s1 += v[c][1]; }
        } else {
            const long nrb = R / 256, rb = tile % nrb, cb = tile / nrb;
            const double2_t* A2 = reinterpret_cast<const double2_t*>(A);
            double2_t v[32];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                v[2 * c] = A2[((cb * 16 + c) * R + rb * 256) / 2 + lane];
                v[2 * c + 1] = A2[((cb * 16 + c) * R + rb * 256) / 2 + 64 + lane];
            }
#pragma unroll
            for (int c = 0; c < 32; ++c) { s0 += v[c][0]; s1 += v[c][1]; }
        }
    }
    out[(long)wid * 64 + lane] = s0 + s1;
}

int main() {
    double *A = nullptr, *out = nullptr;
    const size_t bytes = (size_t)R * C * sizeof(double);
    if (hipMalloc(&A, bytes) != hipSuccess) return 1;
    if (hipMemset(A, 0, bytes) != hipSuccess) return 1;
    const int max_wgs = 8192;
    if (hipMalloc(&out, (size_t)max_wgs * 4 * 64 * sizeof(double)) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int wgs_list[] = {64, 128, 256, 512, 1024, 2048, 8192};
    for (int mode = 0; mode < 3; ++mode)
        for (int wgs : wgs_list) {
            const int nwaves = wgs * 4;
            const int tpw = (int)((kTiles + nwaves - 1) / nwaves);
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k_read<0>, dim3(wgs), dim3(256), 0, 0, A, out, tpw, nwaves);
                else if (mode == 1) hipLaunchKernelGGL(k_read<1>, dim3(wgs), dim3(256), 0, 0, A, out, tpw, nwaves);
                else hipLaunchKernelGGL(k_read<2>, dim3(wgs), dim3(256), 0, 0, A, out, tpw, nwaves);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("mode %d  workgroups %5d  tiles/wave %4d  %8.3f ms  %7.1f GB/s\n", mode, wgs, tpw, best,
                   (double)bytes / best / 1e6);
            if (hipGetLastError() != hipSuccess) return 2;
        }
    return 0;
}
