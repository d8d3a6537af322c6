// C ABI, host-only part: inspector + synthetic matrices (include/parsy_amd.h §3, §4).
#include <cstring>
#include <exception>
#include <string>

#include "../../include/parsy_amd.h"
#include "errors.hpp"
#include "gen.hpp"
#include "ordering.hpp"
#include "inspector.hpp"

namespace parsy {
static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const std::string& last_error() { return g_last_error; }
}  // namespace parsy

struct parsy_symbolic {
    parsy::Symbolic S;
};

extern "C" {

const char* parsy_last_error(void) { return parsy::last_error().c_str(); }

parsy_symbolic* parsy_analyze(int n, const int* Ap, const int* Ai, const double* Ax,
                              const int* perm, const int* nrelax, const double* zrelax) {
    static const int def_nrelax[3] = {4, 16, 48};          // examples/choleskyTest01.cpp:111
    static const double def_zrelax[3] = {0.8, 0.1, 0.05};  // examples/choleskyTest01.cpp:112
    if (n < 0 || !Ap || (n > 0 && !Ai)) {
        parsy::set_last_error("parsy_analyze: null matrix");
        return nullptr;
    }
    try {
        auto* h = new parsy_symbolic;
        parsy::analyze(n, Ap, Ai, Ax, perm, nrelax ? nrelax : def_nrelax,
                       zrelax ? zrelax : def_zrelax, h->S);
        return h;
    } catch (const std::exception& e) {
        parsy::set_last_error(std::string("parsy_analyze: ") + e.what());
        return nullptr;
    }
}

void parsy_symbolic_free(parsy_symbolic* sym) { delete sym; }

int parsy_symbolic_get(const parsy_symbolic* sym, parsy_symbolic_view* v) {
    if (!sym || !v) return -1;
    const parsy::Symbolic& S = sym->S;
    std::memset(v, 0, sizeof(*v));
    v->n = S.n;
    v->nsuper = S.nsuper;
    v->nlevels = (int)S.levelPtr.size() - 1;
    v->maxSupWid = S.maxSupWid;
    v->maxCol = S.maxCol;
    v->ssize = S.ssize;
    v->xsize = S.xsize;
    v->nnzL = S.nnzL;
    v->nnzA = (int64_t)S.A2.i.size();
    v->n_updates = (int64_t)S.upd_sn.size();
    v->flops_colcount = S.flops_colcount;
    v->flops_stored = S.flops_stored;
    v->Perm = S.perm.data();
    v->Parent = S.parent.data();
    v->ColCount = S.colcount.data();
    v->super = S.super.data();
    v->col2Sup = S.col2sup.data();
    v->sParent = S.sparent.data();
    v->p = S.p.data();
    v->i_ptr = S.i_ptr.data();
    v->s = S.s.data();
    v->A1p = S.A1.p.data();
    v->A1i = S.A1.i.data();
    v->A2p = S.A2.p.data();
    v->A2i = S.A2.i.data();
    v->A2x = S.A2.x.empty() ? nullptr : S.A2.x.data();
    v->A2src = S.A2.src.data();
    v->levelPtr = S.levelPtr.data();
    v->levelSet = S.levelSet.data();
    v->updPtr = S.upd_ptr.data();
    v->updSn = S.upd_sn.data();
    v->updLb = S.upd_lb.data();
    v->updUb = S.upd_ub.data();
    return 0;
}

int64_t parsy_grid_spd_lower(int nx, int ny, int nz, int stencil, double shift, int* Ap, int* Ai,
                             double* Ax) {
    try {
        std::vector<int> p, i;
        std::vector<double> x;
        parsy::grid_spd_lower(nx, ny, nz, stencil, shift, p, i, x);
        if (Ap) std::memcpy(Ap, p.data(), p.size() * sizeof(int));
        if (Ai) std::memcpy(Ai, i.data(), i.size() * sizeof(int));
        if (Ax) std::memcpy(Ax, x.data(), x.size() * sizeof(double));
        return (int64_t)i.size();
    } catch (const std::exception& e) {
        parsy::set_last_error(std::string("parsy_grid_spd_lower: ") + e.what());
        return -1;
    }
}

int parsy_order_nd(int n, const int* Ap, const int* Ai, int leaf, int* perm) {
    if (!perm) {
        parsy::set_last_error("parsy_order_nd: perm is NULL");
        return -1;
    }
    try {
        std::vector<int> p;
        parsy::order_nested_dissection(n, Ap, Ai, leaf, p);
        std::copy(p.begin(), p.end(), perm);
        return 0;
    } catch (const std::exception& e) {
        parsy::set_last_error(std::string("parsy_order_nd: ") + e.what());
        return -1;
    }
}

int parsy_grid_nested_dissection(int nx, int ny, int nz, int leaf, int* perm) {
    try {
        std::vector<int> p;
        parsy::grid_nested_dissection(nx, ny, nz, leaf, p);
        std::memcpy(perm, p.data(), p.size() * sizeof(int));
        return 0;
    } catch (const std::exception& e) {
        parsy::set_last_error(std::string("parsy_grid_nested_dissection: ") + e.what());
        return -1;
    }
}

}  // extern "C"

const parsy::Symbolic* parsy_symbolic_cxx(const parsy_symbolic* s) { return s ? &s->S : nullptr; }
