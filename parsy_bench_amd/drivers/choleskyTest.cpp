// choleskyTest -- same command line and CSV as the reference's driver
// (examples/choleskyTest01.cpp:43-277), running on the MI355X executor:
//
//   choleskyTest <lower.mtx> numThread chunk costParam levelParam blasThreads finalSeqNode [orderFile]
//
// read -> inspect -> 5 x { zero valL; cholesky_left_par_05(...) } -> one CSV line:
//   file,numThread,chunk,costParam,levelParam,blasThreads,finalSeqNode,total_s,parallel_s,root_s,symbolic_s,ordering_s,
// (iteration #3 of 5 is reported, unsorted, as the reference does: :266-277).
// The inspector is this library's (parsy_analyze).  METIS is not available here, so
// without an order file the matrix is ordered by parsy_order_nd (graph nested dissection).
// The H-level arrays handed to the executor are the etree level sets with one
// supernode per w-partition; numThread / chunk / costParam / levelParam / blasThreads /
// finalSeqNode are accepted and echoed (the GPU executor schedules by etree level).
//
// Environment (the reference has no slot for either):
//   PARSY_VERIFY=1       check the factor after the timed iterations (drivers/verify.hpp: the reference's VERIFY build
//                        compares with CHOLMOD, examples/choleskyTest01.cpp:459-546); verdict on stderr, exit code -2 on
//                        failure
//   PARSY_DEVICES=0,1,.. one factorization over several devices (one rank per entry; parsy_mg_*, include/parsy_amd.h
//                        section 5): subtrees below a cut on one device each, the pieces of the separators above it
//                        dealt over all devices.  With it the CSV's total_s / parallel_s columns are HOST WALL TIME of
//                        the call (enqueue + run) and root_s is 0; device seconds per rank go to stderr
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
        // the reference orders with METIS here (cholesky/LSparsity.h); this build has its own graph
        // nested dissection instead
        perm.resize(n);
        if (parsy_order_nd(n, Ap.data(), Ai.data(), 0, perm.data()) != 0) {
            std::cerr << "[choleskyTest] ordering failed: " << parsy_last_error() << "\n";
            return -1;
        }
    }
    const double orderingTime = std::chrono::duration<double>(std::chrono::system_clock::now() - t0).count();

    const int nrelax[3] = {4, 16, 48};           // examples/choleskyTest01.cpp:111
    const double zrelax[3] = {0.8, 0.1, 0.05};   // :112
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

    // H-level arrays: etree levels, one supernode per w-partition
    std::vector<int> levelPtr(v.levelPtr, v.levelPtr + v.nlevels + 1), parPtr(v.nsuper + 1), partition(v.nsuper);
    for (int k = 0; k <= v.nsuper; ++k) parPtr[k] = k;
    for (int k = 0; k < v.nsuper; ++k) partition[k] = v.levelSet[k];

    std::vector<double> valL((size_t)v.xsize);
    std::vector<double> timingChol(4 + (numThread > 0 ? numThread : 1));
    struct CholTime { double alltogether, parallel, rootNodes; };
    std::vector<CholTime> timeArray;
    const int iterNo = 5;
    bool devices_ok = true;
    const std::vector<int> devices = parsy_io::device_list(&devices_ok);
    if (!devices_ok) {
        std::cerr << "[choleskyTest] PARSY_DEVICES must be a comma-separated list of device ordinals, got '"
                  << std::getenv("PARSY_DEVICES") << "'\n";
        parsy_symbolic_free(sym);
        return -1;
    }
    if (!devices.empty()) {
        // several devices: the distributed factorization (every rank keeps its lValues on its device; the factor is
        // collected once, after the timed iterations)
        parsy_mg* mg = parsy_mg_create(sym, (int)devices.size(), devices.data(), /*block: default*/ 0);
        if (!mg || parsy_mg_set_values(mg, v.A2x) != 0) {
            std::cerr << "[choleskyTest] multi-device setup failed: " << parsy_last_error() << "\n";
            parsy_mg_destroy(mg);
            parsy_symbolic_free(sym);
            return -1;
        }
        for (int k = 0; k < iterNo; ++k) {
            double sec = 0;
            const int st = parsy_mg_factor(mg, &sec);
            if (st != 0) {
                std::cerr << "[choleskyTest] " << parsy_last_error() << "\n";
                parsy_mg_destroy(mg);
                parsy_symbolic_free(sym);
                return -1;
            }
            timeArray.push_back({sec, sec, 0.0});
            if (k == iterNo / 2) {   // device seconds of the reported iteration: the slowest rank
                std::vector<double> ms(devices.size(), 0.0);
                if (parsy_mg_rank_ms(mg, ms.data()) == 0)
                    for (double m : ms) timingChol[2] = std::max(timingChol[2], m * 1e-3);
            }
        }
        std::vector<double> rank_ms(devices.size());
        parsy_mg_rank_ms(mg, rank_ms.data());
        std::cerr << "[choleskyTest] " << devices.size() << " ranks, device ms of the last iteration:";
        for (double ms : rank_ms) std::cerr << " " << ms;
        std::cerr << "\n";
        const int grc = parsy_mg_gather_host(mg, valL.data());
        parsy_mg_destroy(mg);
        if (grc != 0) {
            std::cerr << "[choleskyTest] " << parsy_last_error() << "\n";
            parsy_symbolic_free(sym);
            return -1;
        }
    }
    for (int k = 0; k < iterNo && devices.empty(); ++k) {
        std::fill(valL.begin(), valL.end(), 0.0);
        std::fill(timingChol.begin(), timingChol.end(), 0.0);
        auto s = std::chrono::system_clock::now();
        const bool ok = cholesky_left_par_05(
            n, (int*)v.A2p, (int*)v.A2i, (double*)v.A2x, (size_t*)v.p, (int*)v.s, (size_t*)v.i_ptr, valL.data(),
            (int*)v.super, v.nsuper, timingChol.data(), (int*)v.sParent, (int*)v.A1p, (int*)v.A1i, (int*)v.col2Sup,
            v.nlevels, levelPtr.data(), nullptr, 0, parPtr.data(), partition.data(), chunk, numThread,
            v.maxSupWid + 1, v.maxCol + 1, nullptr);
        const double dt = std::chrono::duration<double>(std::chrono::system_clock::now() - s).count();
        if (!ok) return -1;
        timeArray.push_back({dt, timingChol[0], timingChol[1]});
    }
    const int mid = iterNo == 1 ? 0 : iterNo / 2;
    std::cout << f1 << "," << numThread << "," << chunk << "," << costParam << "," << levelParam << ","
              << blasThreads << "," << finalSeqNode << ",";
    std::cout << timeArray[mid].alltogether << "," << timeArray[mid].parallel << "," << timeArray[mid].rootNodes << ",";
    std::cout << durationSym << "," << orderingTime << ",";
    std::cout << "\n";
    std::cerr << "[choleskyTest] n=" << n << " nsuper=" << v.nsuper << " nnz(L)=" << v.nnzL << " F=" << v.flops_colcount
              << " device_s(iter3)=" << timingChol[2] << "\n";
    int rc = 0;
    if (parsy_io::verify_requested() &&
        !parsy_io::verify_and_report("choleskyTest", sym, v, valL.data(), devices.empty() ? 0 : devices[0]))
        rc = -2;
    parsy_dropin_reset();
    parsy_symbolic_free(sym);
    return rc;
}
