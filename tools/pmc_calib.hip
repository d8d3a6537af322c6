// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters for the access shapes of the tile kernel
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern").  Three kernels over a buffer far larger than the 256-MiB Infinity Cache:
//   calib_read8_mfma   every wave instruction reads 4 x 128-B segments (16 rows x 8 B at 4 k columns, ld apart)
//                      = the MFMA operand pattern of k_chol_tiles; reads exactly `bytes`
//   calib_read16       16 B per lane, contiguous (the guide's reference shape)
//   calib_write8_sc1   8-B write-through stores (the chain's tile publication)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); tools/pmc_traffic.py
// turns counter value / known bytes into the correction factors.
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib.bin
#include <hip/hip_runtime.h>

#include <cstdio>

static constexpr size_t kBytes = 2ull << 30;  // 2 GiB
static constexpr int kLd = 2604;              // leading dimension of the panel (doubles)

// panel of kLd rows x ncols columns; every wave reads blocks of 16 rows x 16 columns (4 k steps of 4 columns)
__global__ __launch_bounds__(256) void calib_read8_mfma(const double* __restrict__ p, double* __restrict__ out,
                                                        long ncols) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const long row_blocks = kLd / 16, col_blocks = ncols / 16;
    double acc = 0;
    for (long b = wave; b < row_blocks * col_blocks; b += nwaves) {
        const long rb = b % row_blocks, cb = b / row_blocks;
        const double* q = p + (cb * 16 + kq) * kLd + rb * 16 + l15;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += q[(long)(4 * u) * kLd];
    }
    if (acc == 12345.678) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_read16(const double2* __restrict__ p, double* __restrict__ out, long n2) {
    double acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) {
        const double2 v = p[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_write8_sc1(double* __restrict__ p, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        __hip_atomic_store(p + i, (double)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 8 B per lane, 64 lanes contiguous (512 B per wave instruction): the staging loads of k_chol_big and the
// row loads of the solve kernels
__global__ __launch_bounds__(256) void calib_read8(const double* __restrict__ p, double* __restrict__ out, long n) {
    double acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += p[i];
    if (acc == 12345.678) out[0] = acc;
}

// plain 8-B stores, contiguous lanes (k_chol_big's read-modify-write of the target tile)
__global__ __launch_bounds__(256) void calib_write8(double* __restrict__ p, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = (double)i;
}

// LDS-DMA, 16 B per lane, 64 lanes contiguous (1 KiB per wave instruction) from an address that is 8- but not
// 16-byte aligned: the staging of k_chol_big (round 3: global_load_lds_dwordx4)
__global__ __launch_bounds__(256) void calib_dma16(const double* __restrict__ p, double* __restrict__ out, long n2) {
    __shared__ __attribute__((aligned(16))) double S[4][128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double acc = 0;
    for (long i = (long)blockIdx.x * 256 + wave * 64; i + 64 <= n2 - 1; i += (long)gridDim.x * 256) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 1 + 2 * (i + lane)),
                                         (__attribute__((address_space(3))) void*)&S[wave][0], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += S[wave][lane];
    }
    if (acc == 12345.678) out[0] = acc;
}

int main() {
    double *buf, *out;
    if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(buf, 0, kBytes);
    const long n = (long)(kBytes / 8);
    const long ncols = (n / kLd) / 16 * 16;
    const long read8_bytes = (kLd / 16) * 16 * ncols * 8;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_read8_mfma, dim3(4096), dim3(256), 0, 0, buf, out, ncols);
        hipLaunchKernelGGL(calib_read16, dim3(4096), dim3(256), 0, 0, (const double2*)buf, out, n / 2);
        hipLaunchKernelGGL(calib_write8_sc1, dim3(4096), dim3(256), 0, 0, buf, n);
        hipLaunchKernelGGL(calib_read8, dim3(4096), dim3(256), 0, 0, buf, out, n);
        hipLaunchKernelGGL(calib_write8, dim3(4096), dim3(256), 0, 0, buf, n);
        hipLaunchKernelGGL(calib_dma16, dim3(4096), dim3(256), 0, 0, buf, out, n / 2);
    }
    hipDeviceSynchronize();
    printf("{\"calib_read8_mfma_bytes\": %ld, \"calib_read16_bytes\": %ld, \"calib_write8_sc1_bytes\": %ld, "
           "\"calib_read8_bytes\": %ld, \"calib_write8_bytes\": %ld, \"calib_dma16_bytes\": %ld}\n",
           read8_bytes, (long)kBytes, (long)kBytes, (long)kBytes, (long)kBytes, (long)((n / 2 - 1) / 64 * 64 * 16));
    return 0;
}
