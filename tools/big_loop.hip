// Microbenchmark: the steady-state chunk loop of k_chol_big built up step by step (512 threads, 2 workgroups per
// CU, v_mfma_f64_16x16x4_f64, 8 MFMAs per k step, 4 k steps per chunk):
//   A  MFMAs with operands from LDS          B  + one workgroup barrier per chunk
//   C  + the staging stores of the next chunk (8 ds_write_b64 per thread)
//   D  + the global loads of the chunk after that (8 x 8 B per thread, streaming a buffer larger than the caches)
// Build: hipcc --offload-arch=gfx950 -O3 tools/big_loop.hip -o tools/big_loop.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int kLd = 144;
template <int MODE>
__global__ __launch_bounds__(512, 4) void k_loop(const double* __restrict__ src, double* out, int chunks, long stride,
                                                 int nfr, int nfc, double* __restrict__ tiles) {
    __shared__ double R[2][16 * kLd], C[2][16 * kLd];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, lrow = tid & 127, lkh = tid >> 7;
    for (int i = tid; i < 2 * 16 * kLd; i += 512) { (&R[0][0])[i] = 1e-3 * i; (&C[0][0])[i] = 2e-3 * i; }
    __syncthreads();
    double4_t acc[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = double4_t{0, 0, 0, 0};
    const double* p = src + (long)(blockIdx.x & 255) * stride + lrow;  // (the host sizes src for 256 streams)
    double vR[4] = {0, 0, 0, 0}, vC[4] = {0, 0, 0, 0};
    if (MODE >= 5) {  // the start of a task: first chunk fetched and staged, second fetched, then the barrier
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vR[q] = p[(long)(lkh + 4 * q) * 20000];
            vC[q] = p[(long)(lkh + 4 * q) * 20000 + 128];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            R[0][(lkh + 4 * q) * kLd + lrow] = vR[q];
            C[0][(lkh + 4 * q) * kLd + lrow] = vC[q];
        }
        p += 16 * 20000L;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vR[q] = p[(long)(lkh + 4 * q) * 20000];
            vC[q] = p[(long)(lkh + 4 * q) * 20000 + 128];
        }
        p += 16 * 20000L;
        __syncthreads();
    }
    for (int n = 0; n < chunks; ++n) {
        const int b = n & 1;
        if (MODE >= 2) {  // stage chunk n + 1
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                R[b ^ 1][(lkh + 4 * q) * kLd + lrow] = vR[q];
                C[b ^ 1][(lkh + 4 * q) * kLd + lrow] = vC[q];
            }
        }
        if (MODE >= 3) {  // fetch chunk n + 2
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                vR[q] = p[(long)(lkh + 4 * q) * 20000];
                vC[q] = p[(long)(lkh + 4 * q) * 20000 + 128];
            }
            p += 16 * 20000L;
        }
        const double* Rb = &R[b][64 * wr + l15];
        const double* Cb = &C[b][32 * wc + l15];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            double rv[4], cv[2];
#pragma unroll
            for (int f = 0; f < 4; ++f) rv[f] = Rb[(4 * ks + kq) * kLd + 16 * f];
#pragma unroll
            for (int f = 0; f < 2; ++f) cv[f] = Cb[(4 * ks + kq) * kLd + 16 * f];
#pragma unroll
            for (int fc = 0; fc < 2; ++fc) {
                if (MODE < 4 || fc < nfc) {
#pragma unroll
                    for (int fr = 0; fr < 4; ++fr)
                        if (MODE < 4 || fr < nfr)
                            acc[fc][fr] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv[fc], rv[fr], acc[fc][fr], 0, 0, 0);
                }
            }
        }
        if (MODE >= 1) __syncthreads();
    }
    if (MODE >= 5) {  // the end of a source: read-modify-write of the 128 x 128 tile, 8 loads of a lane at a time
        // 4096 tiles side by side in a panel of ld = 4096 * 128 rows: element (r, c) of a tile at c * ld + r
        double* __restrict__ T = tiles + (size_t)(blockIdx.x & 4095) * 128;
        const long tld = 4096L * 128;
#pragma unroll
        for (int fc = 0; fc < 2; ++fc)
#pragma unroll
            for (int fh = 0; fh < 4; fh += 2) {
                double old[2][4];
#pragma unroll
                for (int fr = fh; fr < fh + 2; ++fr)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        old[fr - fh][v] = T[(long)(32 * wc + 16 * fc + kq + 4 * v) * tld + 64 * wr + 16 * fr + l15];
#pragma unroll
                for (int fr = fh; fr < fh + 2; ++fr)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        T[(long)(32 * wc + 16 * fc + kq + 4 * v) * tld + 64 * wr + 16 * fr + l15] = old[fr - fh][v] - acc[fc][fr][v];
                asm volatile("" ::: "memory");
            }
    }
    double s = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) s += acc[a][b][c];
    out[(blockIdx.x & 4095) * 512 + tid] = s;  // (out holds 4096 x 512 doubles)
}
__global__ void k_fill(double* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        p[i] = ((double)(h & 0xfffffffffffffull) / 4503599627370496.0 - 0.5) * 1e-2;  // random, |v| < 5e-3
    }
}

int main() {
    double *d, *src, *tiles;
    const long stride = 4L << 20;  // doubles between the streams of two workgroups (32 MB)
    (void)hipMalloc(&d, 4096 * 512 * 8);
    const size_t src_doubles = (size_t)(256 * stride + 20000L * 16 * 2100 + 1024);  // 256 streams, <= 2000 chunks each
    if (hipMalloc(&src, src_doubles * 8) != hipSuccess) return 1;
    const bool zeros = getenv("BIG_LOOP_ZEROS") != nullptr;
    if (zeros) (void)hipMemset(src, 0, src_doubles * 8);
    else hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, src, src_doubles);
    printf("operands: %s\n", zeros ? "zeros" : "random");
    if (hipMalloc(&tiles, (size_t)4096 * 128 * 128 * 8) != hipSuccess) return 1;  // 4096 tiles side by side (512 MB)
    (void)hipMemset(tiles, 0, (size_t)4096 * 128 * 128 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int blocks, int chunks) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, src, d, 10, stride, 4, 2, tiles);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, src, d, chunks, stride, 4, 2, tiles);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * 128 * 128 * 16 * (double)chunks * blocks;
        printf("%-44s blocks=%4d chunks=%5d  %.3f ms  %.2f TFLOP/s\n", name, blocks, chunks, ms, fl / ms / 1e9);
    };
    for (int blocks : {512, 2048}) {
        run("A mfma + LDS operands", k_loop<0>, blocks, 2000);
        run("B + barrier per chunk", k_loop<1>, blocks, 2000);
        run("C + staging stores", k_loop<2>, blocks, 2000);
        run("D + global loads (streaming)", k_loop<3>, blocks, 2000);
        run("D with 32-chunk tasks (K = 512)", k_loop<3>, blocks * 16, 32);
        run("E = D + wave-uniform branch per MFMA", k_loop<4>, blocks, 2000);
        run("E with 32-chunk tasks (K = 512)", k_loop<4>, blocks * 16, 32);
        run("F = E + task start + tile RMW, K = 512", k_loop<5>, blocks * 16, 32);
        run("F with K = 2048", k_loop<5>, blocks * 4, 128);
    }
    return 0;
}
