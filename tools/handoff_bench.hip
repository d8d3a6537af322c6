// Microbenchmark of the in-launch tile hand-off used by the chain kernel: a producer workgroup writes a
// 64x64 tile of a column-major panel (ld = 2604) and raises a flag; a consumer workgroup polls the flag
// and reads the tile the way the tile kernel reads MFMA operands.  Prints, per variant, the consumer's
// time from "flag seen" to "all loads returned" and the producer's store+drain time (100 MHz ticks -> us).
//   hipcc --offload-arch=gfx950 -O3 tools/handoff_bench.hip -o tools/handoff_bench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

static constexpr int kLd = 2604;

__device__ __forceinline__ double ld_sc1(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// variant: bit0 = stores sc1 (else plain + release fence); bit1 = loads sc1 (else acquire fence + plain);
// bit2 = consumer reads 16 B per lane contiguous along rows instead of the MFMA operand pattern
__global__ __launch_bounds__(256) void k_handoff(double* __restrict__ buf, int* __restrict__ flag,
                                                 unsigned long long* __restrict__ out, int variant, int consumer_block,
                                                 int epoch, int delay) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* tile = buf + 64 * 7;  // a tile somewhere inside the panel, not line-aligned on purpose
    if (blockIdx.x == 0) {
        // let the consumer get to its poll loop first
        for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(64);
        const unsigned long long t0 = wall_clock64();
        const int wa = wave >> 1, wb = wave & 1;
        for (int e = lane; e < 32 * 32; e += 64) {
            const int cc = e >> 5, rr = e & 31;
            double* dst = &tile[(int64_t)(32 * wb + cc) * kLd + 32 * wa + rr];
            const double v = (double)(epoch * 10000 + (32 * wb + cc) * 64 + 32 * wa + rr);
            if (variant & 1) st_sc1(dst, v);
            else *dst = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (!(variant & 1)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[0] = wall_clock64() - t0;
            out[3] = wall_clock64();
        }
    } else if ((int)blockIdx.x == consumer_block) {
        // pre-read the lines (L1/L2 warm with the OLD contents), as a tile of the chain would not -- worst case
        double warm = 0;
        for (int e = tid; e < 64 * 64; e += 256) warm += tile[(int64_t)(e >> 6) * kLd + (e & 63)];
        unsigned long long t0 = wall_clock64();
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            if (wall_clock64() - t0 > 100000000ull) break;
            __builtin_amdgcn_s_sleep(4);
            ++spins;
        }
        const unsigned long long t1 = wall_clock64();
        if (!(variant & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        double acc = 0;
        const int l15 = lane & 15, kq = lane >> 4;
        const int wa = wave >> 1, wb = wave & 1;
        if (variant & 4) {
            // 16 B per lane, contiguous: a wave instruction covers 2 columns x 64 rows
            for (int i = 0; i < 8; ++i) {
                const int col = wave * 16 + i * 2 + (lane >> 5), row = (lane & 31) * 2;
                const double* p = &tile[(int64_t)col * kLd + row];
                if (variant & 2) {
                    acc += ld_sc1(p) + ld_sc1(p + 1);
                } else {
                    acc += p[0] + p[1];
                }
            }
        } else {
            double a[16][4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double* p = tile + (int64_t)(16 * c + 4 * u + kq) * kLd;
                    if (variant & 2) {
                        a[c * 4 + u][0] = ld_sc1(p + 32 * wa + l15);
                        a[c * 4 + u][1] = ld_sc1(p + 32 * wb + l15);
                        a[c * 4 + u][2] = ld_sc1(p + 32 * wa + 16 + l15);
                        a[c * 4 + u][3] = ld_sc1(p + 32 * wb + 16 + l15);
                    } else {
                        a[c * 4 + u][0] = p[32 * wa + l15];
                        a[c * 4 + u][1] = p[32 * wb + l15];
                        a[c * 4 + u][2] = p[32 * wa + 16 + l15];
                        a[c * 4 + u][3] = p[32 * wb + 16 + l15];
                    }
                }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += a[i][0] + a[i][1] + a[i][2] + a[i][3];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = wall_clock64();
        if (tid == 0) {
            out[1] = t2 - t1;
            out[2] = spins;
            out[4] = t1;
        }
        // checksum so that nothing is optimised away and staleness shows: every value must carry the epoch
        buf[(int64_t)70 * kLd + tid] = acc + warm * 1e-300;
    }
}

int main() {
    double* buf;
    int* flag;
    unsigned long long* out;
    hipMalloc(&buf, sizeof(double) * kLd * 80);
    hipMalloc(&flag, 64);
    hipMalloc(&out, 64);
    hipMemset(buf, 0, sizeof(double) * kLd * 80);
    hipMemset(flag, 0, 64);
    int epoch = 0;
    const char* names[8] = {"plain st + release | acquire + plain ld, mfma pattern", "sc1 st | acquire + plain ld, mfma pattern",
                            "plain st + release | sc1 ld, mfma pattern", "sc1 st | sc1 ld, mfma pattern",
                            "plain st + release | acquire + plain ld, 16B rows", "sc1 st | acquire + plain ld, 16B rows",
                            "plain st + release | sc1 ld, 16B rows", "sc1 st | sc1 ld, 16B rows"};
    for (int consumer : {1, 8}) {
        printf("consumer block %d (%s XCD as the producer, if blocks are dealt round-robin)\n", consumer,
               consumer % 8 == 0 ? "same" : "another");
        for (int v = 0; v < 8; ++v) {
            double cons = 0, prod = 0, lag = 0;
            const int reps = 20;
            for (int rep = 0; rep < reps + 2; ++rep) {
                ++epoch;
                hipLaunchKernelGGL(k_handoff, dim3(16), dim3(256), 0, 0, buf, flag, out, v, consumer, epoch, 40);
                hipDeviceSynchronize();
                unsigned long long h[8];
                hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
                if (rep >= 2) {
                    prod += h[0] / 100.0;
                    cons += h[1] / 100.0;
                    lag += ((double)h[4] - (double)h[3]) / 100.0;
                }
            }
            printf("  %-58s producer store+publish %6.2f us   consumer flag->data %6.2f us   flag latency %6.2f us\n", names[v],
                   prod / reps, cons / reps, lag / reps);
        }
    }
    return 0;
}
