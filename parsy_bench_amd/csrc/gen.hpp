// Synthetic SPD test matrices + geometric nested dissection (see gen.cpp).
#pragma once
#include <cstdint>
#include <vector>

namespace parsy {

// Lower triangle (CSC, sorted) of a grid stencil matrix: off-diagonals -1,
// diagonal = (#neighbours) + shift  => strictly diagonally dominant, SPD.
// stencil: 5 / 9 (2-D, nz must be 1), 7 / 27 (3-D).
void grid_spd_lower(int nx, int ny, int nz, int stencil, double shift, std::vector<int>& Ap,
                    std::vector<int>& Ai, std::vector<double>& Ax);

// Geometric nested dissection of the nx*ny*nz grid: split the longest axis at
// its midpoint plane, order the two halves, then the separator plane; boxes
// with <= leaf points are ordered naturally. perm[new] = old.
void grid_nested_dissection(int nx, int ny, int nz, int leaf, std::vector<int>& perm);

}  // namespace parsy
