// Probe of v_mfma_f64_4x4x4_4b_f64's operand lanes (with the A-block broadcast CBSZ = 2, ABID = q): wave (la, lb)
// sets a = 1 on lane la, b = 1 on lane lb and records which lanes of d become 1.  Output: one line per
// (abid, la, lb) with a non-empty result: the d lanes.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int ABID, int CBSZ>
__global__ void probe(double* out) {
    const int lane = threadIdx.x, la = blockIdx.x, lb = blockIdx.y;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
    out[((size_t)la * 64 + lb) * 64 + lane] = d;
}
int main() {
    double* d;
    (void)hipMalloc(&d, 64 * 64 * 64 * 8);
    std::vector<double> h(64 * 64 * 64);
    auto run = [&](auto kern, int abid, int cbsz) {
        (void)hipMemset(d, 0, 64 * 64 * 64 * 8);
        hipLaunchKernelGGL(kern, dim3(64, 64), dim3(64), 0, 0, d);
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        for (int la = 0; la < 64; ++la)
            for (int lb = 0; lb < 64; ++lb) {
                bool any = false;
                for (int l = 0; l < 64; ++l) any = any || h[((size_t)la * 64 + lb) * 64 + l] != 0.0;
                if (!any) continue;
                printf("cbsz %d abid %d la %2d lb %2d ->", cbsz, abid, la, lb);
                for (int l = 0; l < 64; ++l)
                    if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" %d", l);
                printf("\n");
            }
    };
    run(probe<0, 0>, 0, 0);
    run(probe<0, 2>, 0, 2);
    run(probe<1, 2>, 1, 2);
    run(probe<2, 2>, 2, 2);
    run(probe<3, 2>, 3, 2);
    return 0;
}
