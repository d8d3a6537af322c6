// Subtree launches of the BCSC solves with many right-hand sides (gfx950).
//
// Numeric contract as trsv_kernels.hip (reference triangularSolve/Triangular_BCSC.h:139-157: per supernode a dense
// solve with the diagonal block, tmp = L21 * x[cols], x[Li[l]] -= tmp[k]); the backward solve L' x = y is the
// extension of SURVEY 8(f).
//
// The bottom of the etree is cut into subtrees of supernodes of at most 16 columns (schedule.hpp: SubMember, SubTree).
// ONE WAVE walks a subtree for 16 right-hand sides, supernode by supernode in index order (descendants first; the
// backward solve in reverse).  Everything the members hand to each other stays in the wave's LDS -- one slot of 16
// doubles per column of a path to the subtree's root (the column slots are a stack: schedule.cpp, build_sub_lists) and
// per outside row a member touches:
//   forward   slots accumulate L21 * y; a member's x block is its right-hand side minus its slots; the outside slots
//             leave the wave once, at the end, as atomic adds (the level kernels: one atomic per supernode, row and
//             right-hand side -- on the parabolic_fem-class input 39 M of them in this launch, executed at the memory
//             side at the rate of its 64-byte requests; now 14.5 M);
//   backward  slots hold x: the outside rows gathered once, the members' blocks as they are solved.
// Nothing a member needs from global memory depends on x, so the operands of the NEXT member (diagonal block, right-
// hand side, the first chunks of its rows below, their slot numbers) are loaded while the current one is worked on;
// no barrier, no flag, no wait for an atomic to be performed.  All products run on the matrix cores:
//   * v_mfma_f64_16x16x4_f64 with the 16 right-hand sides as the N index; the result layout (lane (rhs, kq), register v
//     = row 4 v + kq) is the B-operand layout of k step v, so x stays in four registers through the whole member;
//   * the triangular solve with the 16 x 16 diagonal block goes by 4-column blocks: y_b = inv(L_bb) x_b, then
//     x_rest -= L(rest, b) y_b -- two products per block; the 4 x 4 inverses by substitution (one entry per lane).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"

namespace parsy {

typedef double double4_s __attribute__((ext_vector_type(4)));

// Ablation build (tools/build_variant.sh subabl -DPARSY_SUBABL; wrong results): PARSY_SUB_ABL=mask drops parts of a
// member's work -- 1: the stores of x, 2: the outside rows (flush / gather), 4: the products of the rows below and their
// LDS traffic, 8: the 4 x 4 inverses, 16: the products of the diagonal solve, 32: the loads of the rows below, 64: the
// loads of the right-hand side.
// Stamps build (tools/build_variant.sh substamps -DPARSY_SUBSTAMPS, tools/sub_stamps.py): wave 0 of the first 2048
// workgroups of the forward kernel writes the shader clock at the marks below (8 per member, the first 7 members).
#ifdef PARSY_SUBSTAMPS
__device__ unsigned long long g_substamp[2048 * 64];
#define SUB_STAMP(i) do { if (lane == 0 && g == 0 && blockIdx.x < 2048 && (i) < 64) g_substamp[blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" void parsy_debug_substamps(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_substamp), sizeof(unsigned long long) * 2048 * 64);
}
#else
#define SUB_STAMP(i) do { } while (0)
#endif
#ifdef PARSY_SUBABL
#define SUB_ABL(bit) ((abl & (bit)) != 0)
#else
#define SUB_ABL(bit) false
#endif

static constexpr int kSubLd = 17;      // doubles per slot: 16 right-hand sides + 1 (lanes along the slots: conflict-free)
static constexpr int kSubFixed = 16 * 17 + 64;   // doubles of LDS besides the slots: the staged diagonal block, its 4 x 4 inverses
static constexpr int kSubPre = 3;      // 16-row chunks of a member's rows below that are loaded a member ahead

struct SubPre {                  // what is loaded a member ahead -- raw: rows / columns past the panel's are replaced at use
    double d[4];                 // diagonal block: forward lane (i = l15, kq): L[i][4 st + kq]; backward: L[4 st + kq][l15]
    double b[4];                 // right-hand side of the member's columns, result layout: row 4 v + kq, rhs l15
    double a[kSubPre][4];        // rows below, chunk ch: forward L21[16 ch + l15][4 st + kq]; backward L21[16 ch + 4 st + kq][l15]
    uint2 sw[kSubPre];           // their slots: 4 x 16 bits, v-th = slot of row 16 ch + 4 v + kq
};

// Every load is unconditional and nothing is done with the values here (addresses past the panel are clamped into it):
// the wave must not wait for any of them before the member in hand is finished.  What a clamped load brings -- an entry
// of L: finite -- meets a zero of x in every product (columns past the member's width, the slot of the padding rows).
// panel entry at a 32-bit BYTE offset from a wave-uniform base: the load takes the base from scalar registers and one
// 32-bit offset register per lane (no 64-bit address arithmetic per load)
__device__ __forceinline__ double sub_ldg(const double* __restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void sub_stg(double* __restrict__ base, unsigned byte_off, double v) {
    *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

template <bool BACK>
__device__ __forceinline__ void sub_load(const SubMember& M, const double* __restrict__ L, const double* __restrict__ xq,
                                         unsigned lq8, const uint2* __restrict__ slots, unsigned sr, int l15, int kq, SubPre& P,
                                         int abl = 0) {
    const double* __restrict__ G = L + M.px;
    const int w = M.w, r = M.r;
    const unsigned ld8 = 8u * (unsigned)M.ld;
    const unsigned il = (unsigned)min(l15, w - 1);
    // forward: column 4 st + kq of the block, rows from l15; backward: column l15, rows from 4 st + kq
    unsigned col[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) col[st] = BACK ? il * ld8 : (unsigned)min(4 * st + kq, w - 1) * ld8;
#pragma unroll
    for (int st = 0; st < 4; ++st) P.d[st] = BACK ? sub_ldg(G, col[0] + 8u * (unsigned)min(4 * st + kq, w - 1)) : sub_ldg(G, col[st] + 8u * il);
#pragma unroll
    for (int v = 0; v < 4; ++v)
        P.b[v] = SUB_ABL(64) ? 1.0 : sub_ldg(xq + (uint64_t)(unsigned)M.c0 * sr, lq8 + 8u * sr * (unsigned)min(4 * v + kq, w - 1));
    const int nch = SUB_ABL(32) ? 0 : (r - w + 15) >> 4;
#pragma unroll
    for (int ch = 0; ch < kSubPre; ++ch) {
        if (ch < nch) {
            const int k0 = w + 16 * ch;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                if (BACK) P.a[ch][st] = sub_ldg(G, col[0] + 8u * (unsigned)min(k0 + 4 * st + kq, r - 1));
                else P.a[ch][st] = sub_ldg(G, col[st] + 8u * (unsigned)min(k0 + l15, r - 1));
            }
            P.sw[ch] = slots[(int64_t)(M.so + ch) * 4 + kq];
        }
    }
}

// The operands loaded a member ahead are claimed BEFORE the next member's loads are issued: the compiler counts the
// loads in flight per path (the chunks a member has, stores behind exec branches), and a wait placed after the new
// loads would -- counted for the shortest path -- also wait for most of them.
__device__ __forceinline__ void sub_arrived(SubPre& P) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        asm volatile("" : "+v"(P.d[i]));
        asm volatile("" : "+v"(P.b[i]));
    }
#pragma unroll
    for (int ch = 0; ch < kSubPre; ++ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(P.a[ch][i]));
        asm volatile("" : "+v"(P.sw[ch].x), "+v"(P.sw[ch].y));
    }
}

// 1 / t to double precision (v_rcp_f64 and two Newton steps; the reference divides, triangularSolve/BLAS.h:8 -- the
// parity bound is a tolerance)
__device__ __forceinline__ double sub_rcp(double t) {
    double d = __builtin_amdgcn_rcp(t);
    d = fma(fma(-t, d, 1.0), d, d);
    d = fma(fma(-t, d, 1.0), d, d);
    return d;
}

// Inverses of the four 4 x 4 diagonal blocks of the staged block Ls[c * 17 + i] = L[i][c]: lane (b = kq, i' = l15 >> 2,
// k' = l15 & 3) forms column k' of inv(L_bb) by substitution (division by the diagonal as triangularSolve/BLAS.h:8) and
// stores entry i': Iv[16 b + 4 i' + k'].
__device__ __forceinline__ void sub_inv4(const double* Ls, double* Iv, int l15, int kq) {
    const double* T = Ls + 4 * kq * kSubLd + 4 * kq;
    const int ip = l15 >> 2, kp = l15 & 3;
    const double d0 = sub_rcp(T[0]), d1 = sub_rcp(T[kSubLd + 1]), d2 = sub_rcp(T[2 * kSubLd + 2]), d3 = sub_rcp(T[3 * kSubLd + 3]);
    const double t10 = T[1], t20 = T[2], t30 = T[3], t21 = T[kSubLd + 2], t31 = T[kSubLd + 3], t32 = T[2 * kSubLd + 3];
    const double y0 = kp == 0 ? d0 : 0.0;
    const double y1 = kp == 1 ? d1 : -d1 * (t10 * y0);
    const double y2 = kp == 2 ? d2 : -d2 * fma(t21, y1, t20 * y0);
    const double y3 = kp == 3 ? d3 : -d3 * fma(t32, y2, fma(t31, y1, t30 * y0));
    Iv[16 * kq + 4 * ip + kp] = ip == 0 ? y0 : ip == 1 ? y1 : ip == 2 ? y2 : y3;
}

__device__ __forceinline__ void sub_lds_add(double* p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_f64, no return
}

__device__ __forceinline__ int sub_slot(uint2 sw, int v) {
    const unsigned word = v < 2 ? sw.x : sw.y;
    return (int)((word >> (16 * (v & 1))) & 0xffffu);
}

// A workgroup = the (up to four) groups of 16 right-hand sides of ONE subtree, one wave each: the waves share nothing
// but the compute unit -- they run the same loads over the subtree's panels a few instructions apart, so three of four
// find the lines in its vector cache (as waves of their own anywhere on the XCD the launch fetched L from L2 once per
// group: at ten waves per compute unit its time did not fall with the occupancy -- bound by the lines a compute unit
// can fetch).  Wave-private LDS: per_wave doubles each.
__device__ __forceinline__ bool sub_block(int tree0, int ngroups, int per_wave, double*& smem, int& tree, int& g) {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    tree = tree0 + (int)blockIdx.x;
    g = (int)blockIdx.y * ((int)blockDim.x >> 6) + wave;
    smem += (size_t)wave * per_wave;
    return g < ngroups;
}

// The member's diagonal block with the identity beyond its width (forward: lane (i = l15, kq), st: L[i][4 st + kq];
// backward: L[4 st + kq][l15]) from the raw loads
template <bool BACK>
__device__ __forceinline__ void sub_diag(const SubPre& P, int w, int l15, int kq, double (&d)[4]) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int c = 4 * st + kq;
        const bool in = BACK ? (l15 < w && c < w && c >= l15) : (c < w && l15 < w && l15 >= c);
        d[st] = in ? P.d[st] : (c == l15 ? 1.0 : 0.0);
    }
}

__global__ __launch_bounds__(256) void k_solve_sub_mrhs(const SubMember* __restrict__ members, const SubTree* __restrict__ trees,
                                                       const uint2* __restrict__ slots, const int32_t* __restrict__ out_rows,
                                                       int tree0, int per_wave, int ngroups, const double* __restrict__ L,
                                                       double* __restrict__ x, int nrhs, unsigned sr, int64_t sq, int tr, int abl) {
    extern __shared__ double smem_all[];
    double* smem = smem_all;
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    int tree, g;
    if (!sub_block(tree0, ngroups, per_wave, smem, tree, g)) return;
    double* Ls = smem;
    double* Iv = smem + 16 * kSubLd;
    double* acc = smem + kSubFixed;
    SUB_STAMP(0);
    const SubTree T = trees[tree];
    const int q0 = 16 * g;
    const bool qok = q0 + l15 < nrhs;
    // x of (row, this lane's right-hand side): a wave-uniform base + a 32-bit byte offset per lane (the launch checks
    // that 16 rows / right-hand sides of either stride stay below 4 GB)
    double* __restrict__ xq = x + (int64_t)q0 * sq;
    const unsigned lq8 = 8u * (unsigned)((qok ? l15 : 0) * sq);
    const int trash = T.ncols + T.nout, nsl = trash + 1;
    SubPre cur, nxt;
    double4_s Xp = {0, 0, 0, 0};   // the member before: its x is stored a member late (below)
    int pc0 = 0, pw = 0;
    SubMember Mc = members[T.m0];
    SubMember Mn = members[min(T.m0 + 1, T.m1 - 1)];   // (descriptors: two members ahead, operands: one)
    sub_load<false>(Mc, L, xq, lq8, slots, sr, l15, kq, cur, abl);
    for (int e = lane; e < nsl * kSubLd; e += 64) acc[e] = 0.0;
    __builtin_amdgcn_wave_barrier();
    SUB_STAMP(1);
    // (the LDS operations of one wave are executed in order: no barrier anywhere)
    for (int m = T.m0; m < T.m1; ++m) {
        const SubMember Mnn = members[min(m + 2, T.m1 - 1)];
        sub_arrived(cur);
        SUB_STAMP(8 + 8 * (m - T.m0) + 0);
        // the x of the member before goes out here, AHEAD of the next loads: the wait at the top of the next member
        // then finds stores that have had a whole member's time (stored right away they were the youngest operations
        // in flight at that wait, and it waited for their acknowledgements)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int c = 4 * v + kq;
            if (c < pw && qok && !SUB_ABL(1)) sub_stg(xq + (uint64_t)(unsigned)pc0 * sr, lq8 + 8u * sr * (unsigned)c, Xp[v]);
        }
        if (m + 1 < T.m1) sub_load<false>(Mn, L, xq, lq8, slots, sr, l15, kq, nxt, abl);
        SUB_STAMP(8 + 8 * (m - T.m0) + 1);
        const int w = Mc.w, r = Mc.r;
        // the diagonal block's 4 x 4 inverses
        double d[4];
        sub_diag<false>(cur, w, l15, kq, d);
#pragma unroll
        for (int st = 0; st < 4; ++st) Ls[(4 * st + kq) * kSubLd + l15] = d[st];
        __builtin_amdgcn_wave_barrier();   // (for the compiler: other lanes' entries are read next)
        if (!SUB_ABL(8)) sub_inv4(Ls, Iv, l15, kq);
        __builtin_amdgcn_wave_barrier();
        const double ainv = SUB_ABL(8) ? 1.0 : Iv[16 * (l15 >> 2) + 4 * (l15 & 3) + kq];
#ifdef PARSY_SUBSTAMPS
        asm volatile("" ::"v"(ainv));
#endif
        SUB_STAMP(8 + 8 * (m - T.m0) + 2);
        // x_s = b_s - (what the members before subtracted); the slots go back to the stack as zeros (a sibling's subtree
        // uses them next) -- lanes of rows past the member's width write to the padding slot
        double4_s X;
        {
            double* sl[4];
            double a[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int c = 4 * v + kq;
                sl[v] = &acc[(c < w ? Mc.slot0 + c : trash) * kSubLd + l15];
                a[v] = *sl[v];
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) *sl[v] = 0.0;
#pragma unroll
            for (int v = 0; v < 4; ++v) X[v] = (4 * v + kq < w && qok) ? cur.b[v] - a[v] : 0.0;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (SUB_ABL(16)) break;
            const double a1 = (l15 >> 2) == b ? ainv : 0.0;
            const double a2 = (l15 >> 2) > b ? -d[b] : 0.0;
            const double t = X[b];
            X[b] = 0.0;
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, t, X, 0, 0, 0);
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, X[b], X, 0, 0, 0);
        }
#ifdef PARSY_SUBSTAMPS
        asm volatile("" ::"v"(X[0]), "v"(X[3]));
#endif
        SUB_STAMP(8 + 8 * (m - T.m0) + 3);
        Xp = X;
        pc0 = Mc.c0;
        pw = w;
        SUB_STAMP(8 + 8 * (m - T.m0) + 4);
        // rows below: slots += L21 y (rows past the panel's go to the padding slot; columns past w meet y = 0)
        const int nch = SUB_ABL(4 | 32) ? 0 : (r - w + 15) >> 4;
#pragma unroll
        for (int ch = 0; ch < kSubPre; ++ch) {
            if (ch < nch) {
                double4_s D = {0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 4; ++st) D = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.a[ch][st], X[st], D, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) sub_lds_add(&acc[sub_slot(cur.sw[ch], v) * kSubLd + l15], D[v]);
            }
        }
        if (nch > kSubPre) {   // (taller members: the chunks beyond those loaded ahead, each loaded while the one before is multiplied)
            const double* __restrict__ G = L + Mc.px;
            const unsigned ld8 = 8u * (unsigned)Mc.ld;
            double a[4], an[4];
            uint2 sw, swn;
            {
                const unsigned k8 = 8u * (unsigned)min(w + 16 * kSubPre + l15, r - 1);
#pragma unroll
                for (int st = 0; st < 4; ++st) a[st] = sub_ldg(G, (unsigned)min(4 * st + kq, w - 1) * ld8 + k8);
                sw = slots[(int64_t)(Mc.so + kSubPre) * 4 + kq];
            }
#pragma unroll 1
            for (int ch = kSubPre; ch < nch; ++ch) {
                const int chn = min(ch + 1, nch - 1);
                const unsigned k8 = 8u * (unsigned)min(w + 16 * chn + l15, r - 1);
#pragma unroll
                for (int st = 0; st < 4; ++st) an[st] = sub_ldg(G, (unsigned)min(4 * st + kq, w - 1) * ld8 + k8);
                swn = slots[(int64_t)(Mc.so + chn) * 4 + kq];
                double4_s D = {0, 0, 0, 0};
#pragma unroll
                for (int st = 0; st < 4; ++st) D = __builtin_amdgcn_mfma_f64_16x16x4f64(a[st], X[st], D, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) sub_lds_add(&acc[sub_slot(sw, v) * kSubLd + l15], D[v]);
#pragma unroll
                for (int st = 0; st < 4; ++st) a[st] = an[st];
                sw = swn;
            }
        }
        __builtin_amdgcn_wave_barrier();
        SUB_STAMP(8 + 8 * (m - T.m0) + 5);
        Mc = Mn;
        Mn = Mnn;
        cur = nxt;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int c = 4 * v + kq;
        if (c < pw && qok && !SUB_ABL(1)) sub_stg(xq + (uint64_t)(unsigned)pc0 * sr, lq8 + 8u * sr * (unsigned)c, Xp[v]);
    }
    SUB_STAMP(2);
    // the outside rows leave the wave: x[row] -= slot (reference Triangular_BCSC.h:154: omp atomic)
    const int32_t* __restrict__ orow = out_rows + T.out0;
    if (SUB_ABL(2)) return;
    if (tr) {   // X row-major: lanes along the right-hand sides (128-byte runs), four rows per instruction
        for (int j0 = 0; j0 < T.nout; j0 += 4) {
            const int j = j0 + kq;
            if (j < T.nout && qok)
                atomicAdd(reinterpret_cast<double*>(reinterpret_cast<char*>(xq + (uint64_t)(unsigned)orow[j] * sr) + lq8),
                          -acc[(T.ncols + j) * kSubLd + l15]);
        }
    } else {    // X right-hand-side-major (sr = 1): lanes along the rows
        const int nq = min(16, nrhs - q0);
        for (int j0 = 0; j0 < T.nout; j0 += 64) {
            const int j = j0 + lane;
            if (j < T.nout) {
                double* __restrict__ xr = x + orow[j] + (int64_t)q0 * sq;
                for (int q = 0; q < nq; ++q) atomicAdd(&xr[(int64_t)q * sq], -acc[(T.ncols + j) * kSubLd + q]);
            }
        }
    }
    SUB_STAMP(3);
}

__global__ __launch_bounds__(256) void k_bsolve_sub_mrhs(const SubMember* __restrict__ members, const SubTree* __restrict__ trees,
                                                        const uint2* __restrict__ slots, const int32_t* __restrict__ out_rows,
                                                        int tree0, int per_wave, int ngroups, const double* __restrict__ L,
                                                        double* __restrict__ x, int nrhs, unsigned sr, int64_t sq, int abl) {
    extern __shared__ double smem_all[];
    double* smem = smem_all;
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    int tree, g;
    if (!sub_block(tree0, ngroups, per_wave, smem, tree, g)) return;
    double* Ls = smem;
    double* Iv = smem + 16 * kSubLd;
    double* Xs = smem + kSubFixed;
    const SubTree T = trees[tree];
    const int q0 = 16 * g;
    const bool qok = q0 + l15 < nrhs;
    double* __restrict__ xq = x + (int64_t)q0 * sq;
    const unsigned lq8 = 8u * (unsigned)((qok ? l15 : 0) * sq);
    SubPre cur, nxt;
    double4_s Xp = {0, 0, 0, 0};   // (stored a member late: as in the forward kernel)
    int pc0 = 0, pw = 0;
    SubMember Mc = members[T.m1 - 1];
    SubMember Mn = members[max(T.m1 - 2, T.m0)];
    sub_load<true>(Mc, L, xq, lq8, slots, sr, l15, kq, cur, abl);
    // x of the outside rows (final: their supernodes were solved by the launches before); the slot of the padding rows = 0
    const int32_t* __restrict__ orow = out_rows + T.out0;
    for (int j0 = 0; j0 < T.nout; j0 += 4) {
        const int j = j0 + kq;
        if (j < T.nout && !SUB_ABL(2)) {
            const double v = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(xq + (uint64_t)(unsigned)orow[j] * sr) + lq8);
            Xs[(T.ncols + j) * kSubLd + l15] = qok ? v : 0.0;
        }
    }
    if (lane < kSubLd) Xs[(T.ncols + T.nout) * kSubLd + lane] = 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int m = T.m1 - 1; m >= T.m0; --m) {
        const SubMember Mnn = members[max(m - 2, T.m0)];
        sub_arrived(cur);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int c = 4 * v + kq;
            if (c < pw && qok && !SUB_ABL(1)) sub_stg(xq + (uint64_t)(unsigned)pc0 * sr, lq8 + 8u * sr * (unsigned)c, Xp[v]);
        }
        if (m > T.m0) sub_load<true>(Mn, L, xq, lq8, slots, sr, l15, kq, nxt, abl);
        const int w = Mc.w, r = Mc.r;
        // t = y_s - L21' x(below)  (rows past the panel's read the padding slot: 0; columns past w are dropped below)
        double4_s A = {0, 0, 0, 0};
        const int nch = SUB_ABL(4 | 32) ? 0 : (r - w + 15) >> 4;
#pragma unroll
        for (int ch = 0; ch < kSubPre; ++ch) {
            if (ch < nch) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const double bv = Xs[sub_slot(cur.sw[ch], st) * kSubLd + l15];
                    A = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.a[ch][st], bv, A, 0, 0, 0);
                }
            }
        }
        if (nch > kSubPre) {
            const double* __restrict__ G = L + Mc.px;
            const unsigned cb = (unsigned)min(l15, w - 1) * 8u * (unsigned)Mc.ld;
            double a[4], an[4];
            uint2 sw, swn;
#pragma unroll
            for (int st = 0; st < 4; ++st) a[st] = sub_ldg(G, cb + 8u * (unsigned)min(w + 16 * kSubPre + 4 * st + kq, r - 1));
            sw = slots[(int64_t)(Mc.so + kSubPre) * 4 + kq];
#pragma unroll 1
            for (int ch = kSubPre; ch < nch; ++ch) {
                const int chn = min(ch + 1, nch - 1);
#pragma unroll
                for (int st = 0; st < 4; ++st) an[st] = sub_ldg(G, cb + 8u * (unsigned)min(w + 16 * chn + 4 * st + kq, r - 1));
                swn = slots[(int64_t)(Mc.so + chn) * 4 + kq];
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const double bv = Xs[sub_slot(sw, st) * kSubLd + l15];
                    A = __builtin_amdgcn_mfma_f64_16x16x4f64(a[st], bv, A, 0, 0, 0);
                }
#pragma unroll
                for (int st = 0; st < 4; ++st) a[st] = an[st];
                sw = swn;
            }
        }
        double d[4];
        sub_diag<true>(cur, w, l15, kq, d);
#pragma unroll
        for (int st = 0; st < 4; ++st) Ls[l15 * kSubLd + 4 * st + kq] = d[st];
        __builtin_amdgcn_wave_barrier();
        if (!SUB_ABL(8)) sub_inv4(Ls, Iv, l15, kq);
        __builtin_amdgcn_wave_barrier();
        const double ainv = SUB_ABL(8) ? 1.0 : Iv[16 * (l15 >> 2) + 4 * kq + (l15 & 3)];   // inv(L_bb)'[i][k] = inv(L_bb)[k][i]
        double4_s X;
#pragma unroll
        for (int v = 0; v < 4; ++v) X[v] = (4 * v + kq < w && qok) ? cur.b[v] - A[v] : 0.0;
        // x_s = inv(L11') t, from the last 4-column block up
#pragma unroll
        for (int b = 3; b >= 0; --b) {
            if (SUB_ABL(16)) break;
            const double a1 = (l15 >> 2) == b ? ainv : 0.0;
            const double a2 = (l15 >> 2) < b ? -d[b] : 0.0;
            const double t = X[b];
            X[b] = 0.0;
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, t, X, 0, 0, 0);
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, X[b], X, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int c = 4 * v + kq;
            if (c < w) Xs[(Mc.slot0 + c) * kSubLd + l15] = X[v];
        }
        Xp = X;
        pc0 = Mc.c0;
        pw = w;
        __builtin_amdgcn_wave_barrier();
        Mc = Mn;
        Mn = Mnn;
        cur = nxt;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int c = 4 * v + kq;
        if (c < pw && qok && !SUB_ABL(1)) sub_stg(xq + (uint64_t)(unsigned)pc0 * sr, lq8 + 8u * sr * (unsigned)c, Xp[v]);
    }
}

int solve_sub_mrhs_min() {   // (read per solve: the tests switch between the two forms of the subtree launch)
    const char* e = std::getenv("PARSY_SUB_MRHS_MIN");
    return e && *e ? std::atoi(e) : 6;
}

static int sub_abl() {
    const char* e = std::getenv("PARSY_SUB_ABL");
    return e && *e ? std::atoi(e) : 0;
}

// LDS a workgroup of these kernels may ask for: 64 KB, or what solve_sub_prepare was granted
static int g_sub_lds_cap = 64 * 1024;

static void sub_grid(const SubTier& T, int nrhs, int& ngroups, dim3& grid, dim3& block, int& per_wave, size_t& lds) {
    ngroups = (nrhs + 15) / 16;
    per_wave = kSubFixed + T.max_slots * kSubLd;
    // (up to four groups of 16 right-hand sides per workgroup -- fewer where the trees' slots are many)
    const int gpb = std::max(1, std::min({ngroups, 4, g_sub_lds_cap / (per_wave * (int)sizeof(double))}));
    grid = dim3(T.ntrees, (ngroups + gpb - 1) / gpb);
    block = dim3(64 * gpb);
    lds = (size_t)per_wave * gpb * sizeof(double);
}

void launch_solve_sub_mrhs(const DevicePattern& P, const SubTier& T, const double* L, double* x, int nrhs, int ldx, int ldq,
                           hipStream_t stream) {
    if (T.ntrees <= 0) return;
    int ngroups, per_wave;
    dim3 grid, block;
    size_t lds;
    sub_grid(T, nrhs, ngroups, grid, block, per_wave, lds);
    const bool tr = ldq > 0;
    hipLaunchKernelGGL(k_solve_sub_mrhs, grid, block, lds, stream, P.sub_members, P.sub_trees,
                       reinterpret_cast<const uint2*>(P.sub_slots), P.sub_out_rows, T.tree0, per_wave, ngroups, L, x, nrhs,
                       (unsigned)(tr ? ldq : 1), (int64_t)(tr ? 1 : ldx), tr ? 1 : 0, sub_abl());
}

void launch_bsolve_sub_mrhs(const DevicePattern& P, const SubTier& T, const double* L, double* x, int nrhs, int ldx,
                            hipStream_t stream) {
    if (T.ntrees <= 0) return;
    int ngroups, per_wave;
    dim3 grid, block;
    size_t lds;
    sub_grid(T, nrhs, ngroups, grid, block, per_wave, lds);
    hipLaunchKernelGGL(k_bsolve_sub_mrhs, grid, block, lds, stream, P.sub_members, P.sub_trees,
                       reinterpret_cast<const uint2*>(P.sub_slots), P.sub_out_rows, T.tree0, per_wave, ngroups, L, x, nrhs, 1u,
                       (int64_t)ldx, sub_abl());
}

// Once per plan: a workgroup of four waves on the largest trees may need more than 64 KB of LDS -- ask for it (up to
// 152 KB); refused: fewer waves per workgroup.  -1: one wave alone would not fit.
int solve_sub_prepare(int max_slots) {
    const int per_wave = (kSubFixed + max_slots * kSubLd) * (int)sizeof(double);
    if (per_wave > 64 * 1024) return -1;
    const int want = std::min(4 * per_wave, 152 * 1024);
    if (want > g_sub_lds_cap) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_solve_sub_mrhs), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_bsolve_sub_mrhs), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess)
            g_sub_lds_cap = want;
        else
            (void)hipGetLastError();
    }
    return 0;
}

}  // namespace parsy
