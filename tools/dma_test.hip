// LDS-DMA test: 16 B per lane from 8-byte-aligned (not 16-byte-aligned) global addresses, per-lane clamped rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* __restrict__ g, double* __restrict__ out, int off, int mi, int ld) {
    __shared__ double S[2][144 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave w loads column w (128 rows) of the window starting at row `off`
    const int r = min(2 * lane, mi - 1);
    const double* src = g + off + r + (long)wave * ld;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)&S[0][wave * 144], 16, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 128; i += blockDim.x) out[i] = S[0][(i >> 7) * 144 + (i & 127)];
}
int main() {
    const int ld = 1000, n = ld * 8;
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) h[i] = i;
    double *g, *o;
    hipMalloc(&g, n * 8); hipMalloc(&o, 512 * 8);
    hipMemcpy(g, h.data(), n * 8, hipMemcpyHostToDevice);
    int bad = 0;
    for (int off : {0, 1, 7, 13, 14, 15, 16, 17}) for (int mi : {128, 127, 65, 2, 1}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, g, o, off, mi, ld);
        std::vector<double> r(512);
        hipMemcpy(r.data(), o, 512 * 8, hipMemcpyDeviceToHost);
        for (int c = 0; c < 4; ++c) for (int i = 0; i < mi; ++i)
            if (r[c * 128 + i] != (double)(off + i + c * ld)) { if (bad < 10) printf("off %d mi %d col %d row %d got %g want %d\n", off, mi, c, i, r[c*128+i], off + i + c*ld); ++bad; }
    }
    printf("bad %d\n", bad);
    return bad != 0;
}
