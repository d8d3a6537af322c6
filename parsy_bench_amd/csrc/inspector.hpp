// Host-side inspector for the MI355X supernodal Cholesky / BCSC-solve executor.
//
// Produces, for a lower-triangular SPD matrix and a fill-reducing permutation,
// the same symbolic objects the reference's inspector hands to its executors
// (reference: cholesky/LSparsity.h:256-842 `analyze_p2`, GIVEN-ordering path):
//   Perm (weighted-postordered), column etree, ColCount, relaxed supernodes
//   (`super`), supernodal etree (`sParent`), row patterns (`s`), value / row
//   pointers (`p`, `i_ptr`), col2Sup, A1 (upper PAP') and A2 (lower PAP').
// It additionally hoists to inspect time what the reference recomputes inside
// its numeric loop: the ordered update lists (common/Reach.h:112-143
// `ereach_sn`; cholesky/Inspection_Prune.h:25-60) and the per-(target,
// descendant) overlap (lb, ub) scan (cholesky/parallel_PB_Cholesky_05.h:137-152).
//
// Everything here is plain host C++ (no HIP); the device executor consumes
// the arrays through `parsy::Symbolic`.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace parsy {

struct CscPattern {
    int n = 0;
    std::vector<int> p;      // n+1
    std::vector<int> i;      // nnz, sorted within columns
    std::vector<double> x;   // nnz (may be empty for pattern-only)
    // for permuted matrices: position of each entry in the caller's value array
    std::vector<int> src;    // nnz (may be empty)
};

struct Symbolic {
    int n = 0;
    int nsuper = 0;
    int64_t ssize = 0, xsize = 0;
    int64_t nnzL = 0;             // stored lower-trapezoid entries, sum(w*r - w(w-1)/2)
    double flops_colcount = 0;    // F = sum_j ColCount_j^2  
    double flops_stored = 0;      // sum_sn sum_t (r-t)^2
    int maxSupWid = 0, maxCol = 0;
    std::vector<int> perm;        // new -> old, after weighted postorder
    std::vector<int> parent;      // column etree in final labels
    std::vector<int> colcount;    // nnz(L(:,j)) incl. diagonal, final labels
    std::vector<int> super;       // nsuper+1   (reference: blockSet / sup2col)
    std::vector<int> col2sup;     // n
    std::vector<int> sparent;     // nsuper     (reference: aTree)
    std::vector<size_t> pi;       // nsuper+1   row-pattern pointer per supernode
    std::vector<size_t> p;        // n+1        (reference: lC / Lp)
    std::vector<size_t> i_ptr;    // n+1        (reference: Li_ptr)
    std::vector<int> s;           // ssize      (reference: lR / Li)
    CscPattern A1;                // upper triangle of PAP' (columns hold rows <= col)
    CscPattern A2;                // lower triangle of PAP' (values + `src` map)
    // wavefront level sets on the supernodal etree (common/TreeUtils.h:119-169)
    std::vector<int> levelPtr, levelSet;
    // ordered update lists: for target t, descendants upd_sn[upd_ptr[t]..upd_ptr[t+1])
    std::vector<int64_t> upd_ptr; // nsuper+1
    std::vector<int> upd_sn;      // descendant supernode id
    std::vector<int> upd_lb;      // first row index (within descendant) >= first col of target
    std::vector<int> upd_ub;      // last row index (within descendant) < last col of target + 1
};

// Symmetric permutation of a lower-triangular CSC matrix: returns the upper
// (`up`) and lower (`lo`) triangles of P A P' in sorted CSC form; `src` maps each
// entry back to its position in the input value array.
void permute_sym(int n, const int* Ap, const int* Ai, const double* Ax,
                 const int* perm, CscPattern& up, CscPattern& lo);

void etree_upper(const CscPattern& up, std::vector<int>& parent);
void postorder(const std::vector<int>& parent, const int* weight, std::vector<int>& post);
void col_counts(const CscPattern& lo, const std::vector<int>& parent,
                const std::vector<int>& post, std::vector<int>& colcount);
void level_sets(const std::vector<int>& tree, std::vector<int>& levelPtr,
                std::vector<int>& levelSet);

// Full analysis. `perm` may be null (identity). nrelax/zrelax as the reference
// drivers pass them (examples/choleskyTest01.cpp:111-112).
void analyze(int n, const int* Ap, const int* Ai, const double* Ax, const int* perm,
             const int nrelax[3], const double zrelax[3], Symbolic& out);

// Borrowed view of the reference-shaped symbolic arrays (what the executors'
// argument lists carry): enough to rebuild update lists from raw pointers.
struct PatternRef {
    int n = 0, nsuper = 0;
    const int* super = nullptr;     // nsuper+1
    const int* col2sup = nullptr;   // n
    const int* sparent = nullptr;   // nsuper
    const size_t* i_ptr = nullptr;  // n+1, per column
    const int* s = nullptr;         // row ids
    const int* A1p = nullptr;       // upper-triangle pattern
    const int* A1i = nullptr;
    // the reference's PRUNE build passes the update lists themselves (getBlockedPruneSet,
    // cholesky/Inspection_Prune.h) instead of aTree / A1: descendants of supernode t in update order
    // are pruneSet[prunePtr[t] .. prunePtr[t+1])
    const int* prunePtr = nullptr;
    const int* pruneSet = nullptr;
};
PatternRef pattern_ref(const Symbolic& S);

// Ordered descendant list of one target supernode (semantics of
// common/Reach.h:112-143); `mark` is a zeroed workspace of size nsuper.
int ereach_supernodal(const PatternRef& S, int target, std::vector<int>& out,
                      std::vector<char>& mark, std::vector<int>& tmp);

// Update lists + (lb, ub) overlap of every (target, descendant) pair
// (cholesky/Inspection_Prune.h:25-60, parallel_PB_Cholesky_05.h:137-152).
void build_update_lists(const PatternRef& S, std::vector<int64_t>& ptr, std::vector<int>& sn,
                        std::vector<int>& lb, std::vector<int>& ub);

}  // namespace parsy
