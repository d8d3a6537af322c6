// ref_harness.cpp -- builds oracle/_ref/libparsy_ref.so (TEST INFRASTRUCTURE).
//
// This file is OUR glue; it #includes, where they lie under /root/reference, the
// reference headers that compile from their own sources with no missing
// dependency, and exports them through a flat C ABI so tests can pin the oracle
// and the product's inspector against the real reference code:
//   common/Reach.h        ereach_sn
//   common/TreeUtils.h    getLevelSet
//   triangularSolve/BLAS.h   dlsolve_blas_nonUnit, dmatvec_blas
//   cholesky/MyBLAS.h     Cholesky_col, lSolve_dense_col (the reference's own
//                         readable POTRF/TRSM, used under -DMYBLAS)
//   common/Etree.h, PostOrder.h, cholesky/Transpose.h, ColumnCount.h,
//   Inspection_BlockC.h, performanceModel.h, InspectionLevel_06.h,
//   Inspection_Prune.h    the inspector pieces analyze_p2 strings together
//
// NOT includable here (they need mkl.h / amd.h, which this image lacks, and we
// do not write stand-ins): cholesky/parallel_PB_Cholesky_05.h,
// Parallel_PB_Cholesky_wavefront.h, triangularSolve/Triangular_BCSC.h,
// cholesky/LSparsity.h, common/Util.h.  ref_analyze() below therefore restates
// the ~60 lines of glue of analyze_p2's GIVEN-ordering path
// (cholesky/LSparsity.h:256-842) around the real reference functions.
//
// Only built when /root/reference exists (this container); the GPU box uses the
// committed vectors in tests/golden/ instead.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <iostream>
#include <chrono>

#include "common/def.h"
#include "common/Reach.h"
#include "common/TreeUtils.h"
#include "common/Etree.h"
#include "common/PostOrder.h"
#include "cholesky/Transpose.h"
#include "cholesky/ColumnCount.h"
#include "cholesky/Inspection_BlockC.h"
#include "cholesky/InspectionLevel_06.h"
#include "cholesky/performanceModel.h"
#include "cholesky/Inspection_Prune.h"
#include "cholesky/MyBLAS.h"
#include "triangularSolve/BLAS.h"

#define REF_API extern "C" __attribute__((visibility("default")))

REF_API int ref_ereach_sn(int n, int* Ap, int* Ai, int col1, int col2, int* col2sup,
                          const int* parent, int* s, int* w) {
    return ereach_sn(n, Ap, Ai, col1, col2, col2sup, parent, s, w);
}

REF_API int ref_getLevelSet(size_t n, const int* inTree, int* levelPtr, int* levelSet) {
    return getLevelSet(n, inTree, levelPtr, levelSet);
}

REF_API void ref_dlsolve_blas_nonUnit(int ldm, int ncol, double* M, double* rhs) {
    dlsolve_blas_nonUnit(ldm, ncol, M, rhs);
}

REF_API void ref_dmatvec_blas(int ldm, int nrow, int ncol, double* M, double* vec, double* Mxvec) {
    dmatvec_blas(ldm, nrow, ncol, M, vec, Mxvec);
}

REF_API void ref_Cholesky_col(int n, int dim, double* a) { Cholesky_col(n, dim, a); }

REF_API int ref_lSolve_dense_col(int colSize, int col, double* M, double* rhs) {
    return lSolve_dense_col(colSize, col, M, rhs);
}

// ---------------------------------------------------------------------------
// Inspector: analyze_p2's GIVEN path around the reference's own functions.
// ---------------------------------------------------------------------------
struct ref_symbolic {
    int n = 0, nsuper = 0, maxSupWid = 0, maxCol = 0;
    long long ssize = 0, xsize = 0;
    int *Perm = nullptr, *Parent = nullptr, *ColCount = nullptr;
    int *super = nullptr, *col2Sup = nullptr, *sParent = nullptr, *s = nullptr;
    size_t *p = nullptr, *i_ptr = nullptr, *pi = nullptr;
    CSC *A1 = nullptr, *A2 = nullptr;  // upper / lower of PAP' with values
    int nLevels = 0, nPar = 0;
    int *levelPtr = nullptr, *parPtr = nullptr, *partition = nullptr;
    int *prunePtr = nullptr, *pruneSet = nullptr;
    int wfLevels = 0;
    int *wfLevelPtr = nullptr, *wfLevelSet = nullptr;
};

REF_API ref_symbolic* ref_analyze(int n, int* Ap, int* Ai, double* Ax, const int* inPerm,
                                  int* nrelax, double* zrelax, int costParam, int levelParam,
                                  int finalSeqNodes) {
    int status = 0;
    ref_symbolic* R = new ref_symbolic;
    R->n = n;
    CSC* A = new CSC;
    A->nzmax = Ap[n];
    A->ncol = A->nrow = n;
    A->stype = -1;
    A->xtype = CHOLMOD_REAL;
    A->packed = TRUE;
    A->p = Ap;
    A->i = Ai;
    A->x = Ax;
    A->nz = NULL;
    A->sorted = TRUE;
    A->z = NULL;

    int* Lperm = new int[n];
    int* Lcolcount = new int[n]();
    int* Lparent = new int[n]();
    int* Post = new int[n]();
    int* First = new int[n]();
    int* Level = new int[n]();
    for (int l = 0; l < n; ++l) Lperm[l] = inPerm ? inPerm[l] : l;

    {   // analyze_ordering (cholesky/LSparsity.h:167-247), permuted lower case
        CSC* S = ptranspose(A, 0, Lperm, NULL, 0, status);   // S = tril(A(p,p))'
        CSC* F = ptranspose(S, 0, NULL, NULL, 0, status);    // F = S'
        etreeC(S, Lparent, status);
        if (postOrderC(Lparent, n, NULL, Post, status) != n) return nullptr;
        int fl = 0, aatfl = 0, lnz = 0;
        rowcolcounts(F, NULL, 0, Lparent, Post, NULL, Lcolcount, First, Level, fl, aatfl, lnz,
                     status);
    }
    // weighted postorder folded into the permutation (cholesky/LSparsity.h:675-723)
    if (postOrderC(Lparent, n, Lcolcount, Post, status) != n) return nullptr;
    {
        int *Wi = First, *InvPost = Level;
        for (int k = 0; k < n; k++) Wi[k] = Lperm[Post[k]];
        for (int k = 0; k < n; k++) Lperm[k] = Wi[k];
        for (int k = 0; k < n; k++) Wi[k] = Lcolcount[Post[k]];
        for (int k = 0; k < n; k++) Lcolcount[k] = Wi[k];
        for (int k = 0; k < n; k++) InvPost[Post[k]] = k;
        for (int nc = 0; nc < n; nc++) {
            int op = Lparent[Post[nc]];
            Wi[nc] = (op == EMPTY) ? EMPTY : InvPost[op];
        }
        for (int k = 0; k < n; k++) Lparent[k] = Wi[k];
    }
    BCSC* L = new BCSC;
    L->n = n;
    L->Perm = Lperm;
    L->ColCount = Lcolcount;
    L->ordering = CHOLMOD_METIS;
    int* Sparent = new int[n]();
    CSC* S = ptranspose(A, 0, Lperm, NULL, 0, status);
    CSC* F = ptranspose(S, 0, NULL, NULL, 0, status);
    super_symbolic2(1, S, F, Lparent, L, nrelax, zrelax, Sparent, status);

    int* col2Sup = new int[n];
    int maxSupWid = 0, maxCol = 0, colLength = 0;
    for (int i = 0; i < (int)L->nsuper; ++i) {
        int k1 = L->super[i], k2 = L->super[i + 1];
        if (maxSupWid < k2 - k1) maxSupWid = k2 - k1;
        for (int j = k1; j < k2; ++j) col2Sup[j] = i;
    }
    for (int j = 0; j < (int)L->nsuper; ++j) {  // cholesky/LSparsity.h:767-785
        int curCol = L->super[j], nxtCol = L->super[j + 1];
        colLength = (int)(L->pi[j + 1] - L->pi[j]);
        if (colLength > maxCol) maxCol = colLength;
        for (int i = curCol + 1; i < nxtCol + 1; ++i) {
            L->i_ptr[i - 1] = L->pi[j];
            L->p[i] = L->p[curCol] + (size_t)(i - curCol) * colLength;
        }
    }
    L->i_ptr[n] = L->pi[L->nsuper];
    L->p[n] = L->p[n - 1] + colLength;

    double* nodeCost = new double[L->nsuper];
    int* xi = new int[2 * n]();
    for (int s = 1; s <= (int)L->nsuper; ++s)  // cholesky/LSparsity.h:790-799 (column etree, as there)
        nodeCost[s - 1] = computeCostperBlock(n, L->super[s - 1], L->super[s], Lparent, F->p, F->i,
                                              col2Sup, L->super, L->s, L->i_ptr, xi);
    delete[] xi;
    R->prunePtr = new int[L->nsuper + 1]();
    R->pruneSet = new int[L->ssize > 0 ? L->ssize : 1];
    getBlockedPruneSet((int)L->nsuper, S->p, S->i, col2Sup, Sparent, L->super, R->prunePtr,
                       R->pruneSet);

    int *levelSet = nullptr;
    getCoarseLevelSet_6(L->nsuper, Sparent, L->super, R->nLevels, R->levelPtr, levelSet, R->nPar,
                        R->parPtr, R->partition, costParam, levelParam, finalSeqNodes, nodeCost);
    delete[] nodeCost;

    R->wfLevelPtr = new int[L->nsuper + 1]();
    R->wfLevelSet = new int[L->nsuper]();
    R->wfLevels = getLevelSet(L->nsuper, Sparent, R->wfLevelPtr, R->wfLevelSet);

    // examples/choleskyTest01.cpp:190-191
    R->A1 = ptranspose(A, 2, Lperm, NULL, 0, status);
    R->A2 = ptranspose(R->A1, 2, NULL, NULL, 0, status);

    R->nsuper = (int)L->nsuper;
    R->ssize = (long long)L->ssize;
    R->xsize = (long long)L->xsize;
    R->maxSupWid = maxSupWid;
    R->maxCol = maxCol;
    R->Perm = Lperm;
    R->Parent = Lparent;
    R->ColCount = Lcolcount;
    R->super = L->super;
    R->col2Sup = col2Sup;
    R->sParent = Sparent;
    R->s = L->s;
    R->p = L->p;
    R->i_ptr = L->i_ptr;
    R->pi = L->pi;
    return R;
}

// flat getters (kind selects the array; returns its length, copies when out != NULL)
REF_API long long ref_get_int(ref_symbolic* R, const char* name, int* out) {
    const int n = R->n, ns = R->nsuper;
    const int* src = nullptr;
    long long len = 0;
    int totalParts = R->levelPtr ? R->levelPtr[R->nLevels] : 0;
#define PICK(nm, ptr, l) if (!strcmp(name, nm)) { src = (ptr); len = (l); }
    PICK("Perm", R->Perm, n)
    PICK("Parent", R->Parent, n)
    PICK("ColCount", R->ColCount, n)
    PICK("super", R->super, ns + 1)
    PICK("col2Sup", R->col2Sup, n)
    PICK("sParent", R->sParent, ns)
    PICK("s", R->s, R->ssize)
    PICK("A1p", R->A1->p, n + 1)
    PICK("A1i", R->A1->i, R->A1->p[n])
    PICK("A2p", R->A2->p, n + 1)
    PICK("A2i", R->A2->i, R->A2->p[n])
    PICK("levelPtr", R->levelPtr, R->nLevels + 1)
    PICK("parPtr", R->parPtr, totalParts + 1)
    PICK("partition", R->partition, ns)
    PICK("prunePtr", R->prunePtr, ns + 1)
    PICK("pruneSet", R->pruneSet, R->prunePtr[ns])
    PICK("wfLevelPtr", R->wfLevelPtr, R->wfLevels + 1)
    PICK("wfLevelSet", R->wfLevelSet, ns)
#undef PICK
    if (!src) return -1;
    if (out) memcpy(out, src, sizeof(int) * (size_t)len);
    return len;
}

REF_API long long ref_get_size(ref_symbolic* R, const char* name, unsigned long long* out) {
    const size_t* src = nullptr;
    long long len = 0;
    if (!strcmp(name, "p")) { src = R->p; len = R->n + 1; }
    if (!strcmp(name, "i_ptr")) { src = R->i_ptr; len = R->n + 1; }
    if (!strcmp(name, "pi")) { src = R->pi; len = R->nsuper + 1; }
    if (!src) return -1;
    if (out) for (long long k = 0; k < len; ++k) out[k] = (unsigned long long)src[k];
    return len;
}

REF_API long long ref_get_double(ref_symbolic* R, const char* name, double* out) {
    const double* src = nullptr;
    long long len = 0;
    if (!strcmp(name, "A1x")) { src = R->A1->x; len = R->A1->p[R->n]; }
    if (!strcmp(name, "A2x")) { src = R->A2->x; len = R->A2->p[R->n]; }
    if (!src) return -1;
    if (out) memcpy(out, src, sizeof(double) * (size_t)len);
    return len;
}

REF_API void ref_get_scalars(ref_symbolic* R, long long* out) {
    out[0] = R->n; out[1] = R->nsuper; out[2] = R->ssize; out[3] = R->xsize;
    out[4] = R->maxSupWid; out[5] = R->maxCol; out[6] = R->nLevels; out[7] = R->nPar;
    out[8] = R->wfLevels;
}
