// Deterministic synthetic SPD matrices and geometric nested-dissection orderings.
//
// SuiteSparse files (ex15, nd24k, Flan_1565, parabolic_fem; reference
// scripts/dlMat.sh:5-21) cannot be fetched offline, and METIS (reference
// cholesky/LSparsity.h:597) is absent, so the benchmark classes are stood in for
// by grid stencils ordered with geometric nested dissection.
// Matrices are emitted the way the reference's reader expects them
// (common/Util.h:77-179): lower triangle, CSC, rows sorted within columns.
#include "gen.hpp"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

namespace parsy {

static inline bool stencil_has(int stencil, int dx, int dy, int dz) {
    const int nzc = (dx != 0) + (dy != 0) + (dz != 0);
    if (nzc == 0) return false;
    switch (stencil) {
        case 5:   // 2-D 5-point
        case 7:   // 3-D 7-point
            return nzc == 1;
        case 9:   // 2-D 9-point
        case 27:  // 3-D 27-point
            return true;
        default:
            throw std::invalid_argument("stencil must be 5, 7, 9 or 27");
    }
}

void grid_spd_lower(int nx, int ny, int nz, int stencil, double shift, std::vector<int>& Ap,
                    std::vector<int>& Ai, std::vector<double>& Ax) {
    if ((stencil == 5 || stencil == 9) && nz != 1) throw std::invalid_argument("2-D stencil needs nz == 1");
    const int64_t n64 = (int64_t)nx * ny * nz;
    if (n64 > 0x7fffffff / 16) throw std::invalid_argument("grid too large for int32 CSC");
    const int n = (int)n64;
    auto id = [&](int x, int y, int z) { return (z * ny + y) * nx + x; };
    Ap.assign(n + 1, 0);
    Ai.clear();
    Ax.clear();
    Ai.reserve((size_t)n * (stencil / 2 + 1));
    Ax.reserve((size_t)n * (stencil / 2 + 1));
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                const int j = id(x, y, z);
                int deg = 0;
                const size_t dpos = Ai.size();
                Ai.push_back(j);
                Ax.push_back(0.0);
                // neighbours with larger linear index, visited in ascending index order
                for (int dz = -1; dz <= 1; ++dz)
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            if (!stencil_has(stencil, dx, dy, dz)) continue;
                            const int X = x + dx, Y = y + dy, Z = z + dz;
                            if (X < 0 || X >= nx || Y < 0 || Y >= ny || Z < 0 || Z >= nz) continue;
                            ++deg;
                            const int i = id(X, Y, Z);
                            if (i > j) {
                                Ai.push_back(i);
                                Ax.push_back(-1.0);
                            }
                        }
                Ax[dpos] = (double)deg + shift;
                Ap[j + 1] = (int)Ai.size();
            }
}

namespace {
struct Box {
    int x0, x1, y0, y1, z0, z1;  // half-open
};
}  // namespace

void grid_nested_dissection(int nx, int ny, int nz, int leaf, std::vector<int>& perm) {
    const int n = nx * ny * nz;
    perm.clear();
    perm.reserve(n);
    auto id = [&](int x, int y, int z) { return (z * ny + y) * nx + x; };
    auto emit = [&](const Box& b) {
        for (int z = b.z0; z < b.z1; ++z)
            for (int y = b.y0; y < b.y1; ++y)
                for (int x = b.x0; x < b.x1; ++x) perm.push_back(id(x, y, z));
    };
    // explicit stack; each frame is visited twice (descend, then emit separator)
    struct Frame {
        Box box;
        Box sep;
        bool expanded;
    };
    std::vector<Frame> st;
    st.push_back({{0, nx, 0, ny, 0, nz}, {}, false});
    while (!st.empty()) {
        Frame f = st.back();
        st.pop_back();
        if (f.expanded) {
            emit(f.sep);
            continue;
        }
        const Box& b = f.box;
        const int lx = b.x1 - b.x0, ly = b.y1 - b.y0, lz = b.z1 - b.z0;
        if (lx <= 0 || ly <= 0 || lz <= 0) continue;
        if ((int64_t)lx * ly * lz <= leaf || std::max(lx, std::max(ly, lz)) < 3) {
            emit(b);
            continue;
        }
        Box lo = b, hi = b, sep = b;
        if (lx >= ly && lx >= lz) {
            const int m = b.x0 + lx / 2;
            lo.x1 = m; sep.x0 = m; sep.x1 = m + 1; hi.x0 = m + 1;
        } else if (ly >= lz) {
            const int m = b.y0 + ly / 2;
            lo.y1 = m; sep.y0 = m; sep.y1 = m + 1; hi.y0 = m + 1;
        } else {
            const int m = b.z0 + lz / 2;
            lo.z1 = m; sep.z0 = m; sep.z1 = m + 1; hi.z0 = m + 1;
        }
        // order: lo subtree, hi subtree, separator  (stack is LIFO)
        st.push_back({b, sep, true});
        st.push_back({hi, {}, false});
        st.push_back({lo, {}, false});
    }
    if ((int)perm.size() != n) throw std::runtime_error("nested dissection lost points");
}

}  // namespace parsy
