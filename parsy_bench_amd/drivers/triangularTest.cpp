// triangularTest -- same command line and output as the reference's triangularTest_chol
// (examples/triangularTest02.cpp:37-271), running on the MI355X executor:
//
//   triangularTest <lower.mtx> numThread chunk costParam levelParam blasThreads finalSeqNode [orderFile]
//
// factor once (cholesky_left_par_05), then 5 timed runs each of blockedLsolve,
// leveledBlockedLsolve (H1), H2LeveledBlockedLsolve and H2LeveledBlockedLsolve_Peeled on
// b = L*1 (rhsInitBlocked, common/Util.h:277), each checked with the reference's
// one-sided testTriangular (common/Util.h:294); prints
//   file,levelParam,finalSeqNode,n,etreeHeight,nBlocks,nnz, t,t,t,t,t,*: t,...,*: ...
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/parsy_amd.h"
#include "mtx_io.hpp"

static void rhsInitBlocked(size_t n, const size_t* Ap, const int* Ai, const size_t* AiP, const double* Ax,
                           double* b) {
    for (size_t j = 0; j < n; ++j) b[j] = 0;
    for (size_t c = 0; c < n; ++c) {
        size_t j = 0;
        for (size_t cc = Ap[c]; cc < Ap[c + 1]; ++cc, ++j) b[Ai[AiP[c] + j]] += Ax[cc];
    }
}

static bool testTriangular(size_t n, const double* x) {  // one-sided, as the reference
    size_t ok = 0;
    for (size_t i = 0; i < n; ++i)
        if (1 - x[i] < 0.001) ok++;
    return ok == n;
}

int main(int argc, char* argv[]) {
    if (argc < 8) {
        std::printf("input args are missing\n"
                    "usage: %s <lower.mtx> numThread chunk costParam levelParam blasThreads finalSeqNode [orderFile]\n",
                    argv[0]);
        return -1;
    }
    const std::string f1 = argv[1];
    const int numThread = std::atoi(argv[2]), chunk = std::atoi(argv[3]);
    const int levelParam = std::atoi(argv[5]), finalSeqNode = std::atoi(argv[7]);
    int n = 0;
    std::vector<int> Ap, Ai, perm;
    std::vector<double> Ax;
    if (!parsy_io::read_lower_mtx(f1, n, Ap, Ai, Ax)) return -1;
    if (argc > 8) {
        if (!parsy_io::read_ordering(argv[8], n, perm)) return -1;
    } else {
        perm.resize(n);  // (the reference orders with METIS; this build with its own graph nested dissection)
        if (parsy_order_nd(n, Ap.data(), Ai.data(), 0, perm.data()) != 0) return -1;
    }
    const int nrelax[3] = {4, 16, 48};
    const double zrelax[3] = {0.8, 0.1, 0.05};
    parsy_symbolic* sym = parsy_analyze(n, Ap.data(), Ai.data(), Ax.data(), perm.empty() ? nullptr : perm.data(),
                                        nrelax, zrelax);
    if (!sym) {
        std::cerr << "analysis failed: " << parsy_last_error() << "\n";
        return -1;
    }
    parsy_symbolic_view v;
    parsy_symbolic_get(sym, &v);
    std::vector<int> levelPtr(v.levelPtr, v.levelPtr + v.nlevels + 1), parPtr(v.nsuper + 1), partition(v.nsuper);
    for (int k = 0; k <= v.nsuper; ++k) parPtr[k] = k;
    for (int k = 0; k < v.nsuper; ++k) partition[k] = v.levelSet[k];
    const int nLevels = v.nlevels, nPar = v.nsuper;

    std::vector<double> valL((size_t)v.xsize, 0.0), timing(8 + numThread, 0.0), x(n);
    if (!cholesky_left_par_05(n, (int*)v.A2p, (int*)v.A2i, (double*)v.A2x, (size_t*)v.p, (int*)v.s,
                              (size_t*)v.i_ptr, valL.data(), (int*)v.super, v.nsuper, timing.data(), (int*)v.sParent,
                              (int*)v.A1p, (int*)v.A1i, (int*)v.col2Sup, nLevels, levelPtr.data(), nullptr, 0,
                              parPtr.data(), partition.data(), chunk, numThread, v.maxSupWid + 1, v.maxCol + 1,
                              nullptr))
        return -1;

    size_t* newCol = (size_t*)v.p;
    int* newRow = (int*)v.s;
    size_t* rowP = (size_t*)v.i_ptr;
    int* col2sup = (int*)v.col2Sup;
    int* sup2col = (int*)v.super;
    const int nBlocks = v.nsuper;
    const int nnz = (int)v.xsize;
    std::cout << f1 << "," << levelParam << "," << finalSeqNode << "," << n << "," << v.nlevels << "," << nBlocks
              << "," << nnz << ",";
    const int iterno = 5;
    auto run = [&](const char* fail, auto&& call) {
        for (int j = 0; j < iterno; ++j) {
            rhsInitBlocked(n, newCol, newRow, rowP, valL.data(), x.data());
            auto s = std::chrono::system_clock::now();
            const int rc = call();
            const double dt = std::chrono::duration<double>(std::chrono::system_clock::now() - s).count();
            if (rc == 1 && (fail == nullptr || testTriangular(n, x.data()))) std::cout << dt << ",";
            else std::cout << (fail ? fail : "failed") << ",";
        }
        std::cout << "*:";
    };
    run(nullptr, [&] { return blockedLsolve(n, newCol, newRow, valL.data(), nnz, rowP, col2sup, sup2col, nBlocks, x.data()); });
    run("H1 failed", [&] {
        return leveledBlockedLsolve(n, newCol, newRow, valL.data(), nnz, rowP, col2sup, sup2col, nBlocks, x.data(),
                                    v.nlevels, (int*)v.levelPtr, (int*)v.levelSet, chunk);
    });
    run("H2 failed", [&] {
        return H2LeveledBlockedLsolve(n, newCol, newRow, valL.data(), nnz, rowP, col2sup, sup2col, nBlocks, x.data(),
                                      nLevels, levelPtr.data(), nullptr, nPar, parPtr.data(), partition.data(), chunk);
    });
    run("H2 failed", [&] {
        return H2LeveledBlockedLsolve_Peeled(n, newCol, newRow, valL.data(), nnz, rowP, col2sup, sup2col, nBlocks,
                                             x.data(), nLevels, levelPtr.data(), nullptr, nPar, parPtr.data(),
                                             partition.data(), chunk, numThread);
    });
    std::cout << "\n";
    parsy_dropin_reset();
    parsy_symbolic_free(sym);
    return 0;
}
