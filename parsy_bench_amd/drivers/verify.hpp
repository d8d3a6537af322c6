// Shared by the drivers: the check behind PARSY_VERIFY=1 and the device list of PARSY_DEVICES.
//
// The reference's VERIFY build (examples/choleskyTest01.cpp:459-546) compares valL entry by entry with CHOLMOD's
// supernodal factor and runs testTriangular; CHOLMOD is not in this build, so the factor is checked through the
// system it must solve: c = (P A P') 1 from the matrix itself, then L L' x = c with the library's forward and
// backward solves must give x = 1 (max|x - 1| <= tol, the reference's testTriangular threshold is 1e-3; the
// drivers use 1e-9).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/parsy_amd.h"

namespace parsy_io {

inline bool verify_requested() {
    const char* e = std::getenv("PARSY_VERIFY");
    return e && e[0] == '1';
}

// One HIP device per rank from PARSY_DEVICES ("0,1,2,3"; a device may repeat); empty: single-device run.
// PARSY_DEVICES=0,1,..: device ordinals, non-negative decimal integers separated by commas.  `ok` (if given) is
// cleared when an entry is anything else ("a,b" must not silently become devices 0,0).
inline std::vector<int> device_list(bool* ok = nullptr) {
    std::vector<int> d;
    if (ok) *ok = true;
    const char* e = std::getenv("PARSY_DEVICES");
    if (!e) return d;
    std::string s(e);
    size_t p = 0;
    while (p < s.size()) {
        size_t q = s.find(',', p);
        if (q == std::string::npos) q = s.size();
        if (q > p) {
            const std::string tok = s.substr(p, q - p);
            const bool numeric = tok.find_first_not_of("0123456789") == std::string::npos && tok.size() <= 4;
            if (!numeric) {
                if (ok) *ok = false;
                return {};
            }
            d.push_back(std::atoi(tok.c_str()));
        }
        p = q + 1;
    }
    return d;
}

// max|x - 1| of L L' x = (P A P') 1 with the factor valL (host); < 0 on an error of the library.
inline double verify_factor(const parsy_symbolic* sym, const parsy_symbolic_view& v, const double* valL, int device) {
    std::vector<double> c((size_t)v.n, 0.0);
    for (int j = 0; j < v.n; ++j)
        for (int q = v.A2p[j]; q < v.A2p[j + 1]; ++q) {
            const int i = v.A2i[q];
            c[(size_t)i] += v.A2x[q];
            if (i != j) c[(size_t)j] += v.A2x[q];
        }
    parsy_plan* plan = parsy_plan_from_symbolic(sym, device);
    if (!plan) return -1.0;
    double err = -1.0;
    if (parsy_solve2_host(plan, valL, c.data(), 1, v.n, 1, nullptr) == 0) {
        err = 0.0;
        for (int i = 0; i < v.n; ++i) err = std::fmax(err, std::fabs(c[(size_t)i] - 1.0));
    }
    parsy_plan_destroy(plan);
    return err;
}

// Prints the verdict on stderr (the CSV on stdout stays the reference's); returns true when the factor passes.
inline bool verify_and_report(const char* who, const parsy_symbolic* sym, const parsy_symbolic_view& v,
                              const double* valL, int device, double tol = 1e-9) {
    const double err = verify_factor(sym, v, valL, device);
    const bool ok = err >= 0.0 && err <= tol;
    std::fprintf(stderr, "[%s] verify: max|x - 1| of L L' x = (P A P') 1 is %.3e (tolerance %.1e): %s\n", who, err, tol,
                 ok ? "ok" : "FAILED");
    return ok;
}

}  // namespace parsy_io
