// choleskyTest03 -- same command line and CSV as the reference's wavefront driver
// (examples/choleskyTest03.cpp:46-240), running on the MI355X executor:
//
//   choleskyTest03 <lower.mtx> numThread chunk costParam levelParam blasThreads finalSeqNode [orderFile]
//
// read -> inspect with nrelax = {4,16,0} (:107: a different supernode blocking from choleskyTest) ->
// etree level sets of the supernodal etree (getLevelSet, :135-141; level 0 = leaves) ->
// 5 x { zero valL; cholesky_left_par_waveFront(...) } (:200-209) -> the five times are sorted in
// descending order and the middle one is printed (:228-236):
//   file,numThread,chunk,costParam,levelParam,blasThreads,finalSeqNode,total_s,symbolic_s,ordering_s,
// The inspector is this library's (parsy_analyze; its level sets are the reference's getLevelSet
// bit for bit: tests/test_oracle.py).  METIS is not available here: without an order file the
// matrix is ordered by parsy_order_nd (graph nested dissection).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/parsy_amd.h"
#include "mtx_io.hpp"
#include "verify.hpp"

int main(int argc, char* argv[]) {
    if (argc < 8) {
        std::printf("input args are missing\n"
                    "usage: %s <lower.mtx> numThread chunk costParam levelParam blasThreads finalSeqNode [orderFile]\n",
                    argv[0]);
        return -1;
    }
    const std::string f1 = argv[1];
    const int numThread = std::atoi(argv[2]), chunk = std::atoi(argv[3]), costParam = std::atoi(argv[4]);
    const int levelParam = std::atoi(argv[5]), blasThreads = std::atoi(argv[6]), finalSeqNode = std::atoi(argv[7]);
    int n = 0;
    std::vector<int> Ap, Ai, perm;
    std::vector<double> Ax;
    if (!parsy_io::read_lower_mtx(f1, n, Ap, Ai, Ax)) return -1;
    auto t0 = std::chrono::system_clock::now();
    if (argc > 8) {
        if (!parsy_io::read_ordering(argv[8], n, perm)) return -1;
    } else {
        perm.resize(n);
        if (parsy_order_nd(n, Ap.data(), Ai.data(), 0, perm.data()) != 0) {
            std::cerr << "[choleskyTest03] ordering failed: " << parsy_last_error() << "\n";
            return -1;
        }
    }
    const double orderingTime = std::chrono::duration<double>(std::chrono::system_clock::now() - t0).count();

    const int nrelax[3] = {4, 16, 0};            // examples/choleskyTest03.cpp:107
    const double zrelax[3] = {0.8, 0.1, 0.05};   // :108
    t0 = std::chrono::system_clock::now();
    parsy_symbolic* sym = parsy_analyze(n, Ap.data(), Ai.data(), Ax.data(), perm.empty() ? nullptr : perm.data(),
                                        nrelax, zrelax);
    if (!sym) {
        std::cerr << "analysis failed: " << parsy_last_error() << "\n";
        return -1;
    }
    parsy_symbolic_view v;
    parsy_symbolic_get(sym, &v);
    const double durationSym = std::chrono::duration<double>(std::chrono::system_clock::now() - t0).count();

    // wavefront schedule: the etree level sets (the reference builds them with getLevelSet)
    std::vector<int> waveFrontPtr(v.levelPtr, v.levelPtr + v.nlevels + 1), waveFrontSet(v.levelSet, v.levelSet + v.nsuper);
    const int nLevels = v.nlevels;

    std::vector<double> valL((size_t)v.xsize);
    std::vector<double> timingChol(4 + (numThread > 0 ? numThread : 1));
    std::vector<double> timeArray;
    const int iterNo = 5;
    for (int k = 0; k < iterNo; ++k) {
        std::fill(valL.begin(), valL.end(), 0.0);
        std::fill(timingChol.begin(), timingChol.end(), 0.0);
        auto s = std::chrono::system_clock::now();
        const bool ok = cholesky_left_par_waveFront(
            n, (int*)v.A2p, (int*)v.A2i, (double*)v.A2x, (size_t*)v.p, (int*)v.s, (size_t*)v.i_ptr, valL.data(),
            (int*)v.super, v.nsuper, timingChol.data(), (int*)v.sParent, (int*)v.A1p, (int*)v.A1i, (int*)v.col2Sup,
            nLevels, waveFrontPtr.data(), waveFrontSet.data(), chunk, numThread, v.maxSupWid + 1, v.maxCol + 1);
        const double dt = std::chrono::duration<double>(std::chrono::system_clock::now() - s).count();
        if (!ok) return -1;
        timeArray.push_back(dt);
    }
    std::sort(timeArray.begin(), timeArray.end(), [](double a, double b) { return a > b; });  // :228-231
    const int mid = iterNo == 1 ? 0 : iterNo / 2;
    std::cout << f1 << "," << numThread << "," << chunk << "," << costParam << "," << levelParam << ","
              << blasThreads << "," << finalSeqNode << ",";
    std::cout << timeArray[mid] << ",";
    std::cout << durationSym << "," << orderingTime << ",";
    std::cout << "\n";
    std::cerr << "[choleskyTest03] n=" << n << " nsuper=" << v.nsuper << " levels=" << nLevels << " nnz(L)=" << v.nnzL
              << " F=" << v.flops_colcount << " device_s(last)=" << timingChol[2] << "\n";
    int rc = 0;   // PARSY_VERIFY=1: check the factor (drivers/verify.hpp; verdict on stderr, exit code -2 on failure)
    if (parsy_io::verify_requested() && !parsy_io::verify_and_report("choleskyTest03", sym, v, valL.data(), 0)) rc = -2;
    parsy_dropin_reset();
    parsy_symbolic_free(sym);
    return rc;
}
