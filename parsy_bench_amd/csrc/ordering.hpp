// Fill-reducing ordering for matrices that come without one (the reference calls METIS,
// cholesky/LSparsity.h; METIS is not in this build).
#pragma once
#include <vector>

namespace parsy {

// Nested dissection on the graph of A (lower triangle, CSC, sorted; the diagonal is ignored):
// every connected piece larger than `leaf` is cut by a level of the breadth-first level structure
// rooted at a pseudo-peripheral vertex (George's automatic nested dissection), thinned to the
// vertices that really touch the far side; the two sides are ordered first, the separator last.
// Pieces of at most `leaf` vertices are ordered by reverse Cuthill-McKee.  perm[new] = old.
void order_nested_dissection(int n, const int* Ap, const int* Ai, int leaf, std::vector<int>& perm);

}  // namespace parsy
