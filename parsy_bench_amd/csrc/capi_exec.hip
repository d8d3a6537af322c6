// C ABI, device part: plan API and the drop-in operators (include/parsy_amd.h §1, §2).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/parsy_amd.h"
#include "errors.hpp"
#include "dist.hpp"
#include "executor.hpp"
#include "inspector.hpp"

using parsy::set_last_error;

namespace {

#define CAPI_HIP(call, ret)                                                          \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            set_last_error(std::string(#call) + ": " + hipGetErrorString(e_));       \
            return ret;                                                              \
        }                                                                            \
    } while (0)

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int pick_device() {
    // PARSY_DEVICE selects the HIP device of the drop-in operators (default 0)
    const char* e = std::getenv("PARSY_DEVICE");
    return e ? std::atoi(e) : 0;
}

// ---- plan cache of the drop-in operators -------------------------------------
// Pattern hash of the drop-in plan cache: 8 bytes per step in four independent lanes (the Flan-class pattern is
// hundreds of MB per call; a byte-at-a-time FNV was a measurable part of the call), folded FNV-style.
uint64_t fnv(uint64_t h, const void* p, size_t bytes) {
    const unsigned char* c = (const unsigned char*)p;
    uint64_t lane[4] = {h, h ^ 0x9E3779B97F4A7C15ULL, h ^ 0xC2B2AE3D27D4EB4FULL, h ^ 0x165667B19E3779F9ULL};
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32)
        for (int k = 0; k < 4; ++k) {
            uint64_t w;
            std::memcpy(&w, c + i + 8 * k, 8);
            lane[k] = (lane[k] ^ w) * 1099511628211ULL;
            lane[k] ^= lane[k] >> 29;
        }
    for (int k = 0; k < 4; ++k) h = (h ^ lane[k]) * 1099511628211ULL;
    for (; i < bytes; ++i) {
        h ^= c[i];
        h *= 1099511628211ULL;
    }
    return h ^ (uint64_t)bytes;
}

struct CacheKey {
    const void* a;
    const void* b;
    const void* c;
    uint64_t hash;
    int n, supNo, kind;
    bool operator<(const CacheKey& o) const {
        return std::tie(a, b, c, hash, n, supNo, kind) < std::tie(o.a, o.b, o.c, o.hash, o.n, o.supNo, o.kind);
    }
};

std::mutex g_mu;
std::map<CacheKey, parsy_plan*> g_plans;

bool validate_partition(int supNo, int nLevels, const int* levelPtr, const int* parPtr,
                        const int* partition, const char* who) {
    if (!levelPtr || !parPtr || !partition || nLevels < 1) {
        set_last_error(std::string(who) + ": null H-level schedule");
        return false;
    }
    std::vector<char> seen(supNo, 0);
    const int nparts = levelPtr[nLevels];
    int cnt = 0;
    for (int q = 0; q < nparts; ++q)
        for (int k = parPtr[q]; k < parPtr[q + 1]; ++k) {
            const int s = partition[k];
            if (s < 0 || s >= supNo || seen[s]) {
                set_last_error(std::string(who) + ": H-level partition is not a permutation of the supernodes");
                return false;
            }
            seen[s] = 1;
            ++cnt;
        }
    if (cnt != supNo) {
        set_last_error(std::string(who) + ": H-level partition does not cover every supernode");
        return false;
    }
    return true;
}

bool validate_levelset(int supNo, int nLevels, const int* levelPtr, const int* levelSet, const char* who) {
    if (!levelPtr || !levelSet || nLevels < 1 || levelPtr[nLevels] != supNo) {
        set_last_error(std::string(who) + ": level set does not cover every supernode");
        return false;
    }
    std::vector<char> seen(supNo, 0);
    for (int k = 0; k < supNo; ++k) {
        const int s = levelSet[k];
        if (s < 0 || s >= supNo || seen[s]) {
            set_last_error(std::string(who) + ": level set is not a permutation of the supernodes");
            return false;
        }
        seen[s] = 1;
    }
    return true;
}

void loud(const char* who) {
    std::fprintf(stderr, "[parsy_amd] %s failed: %s\n", who, parsy_last_error());
}

parsy_plan* cached_chol_plan(int n, int supNo, const int* blockSet, const size_t* lC,
                             const size_t* Li_ptr, const int* lR, const int* aTree,
                             const int* col2Sup, const int* cT, const int* rT, const int* c,
                             const int* r) {
    uint64_t h = 1469598103934665603ULL;
    h = fnv(h, blockSet, sizeof(int) * (supNo + 1));
    h = fnv(h, lR, sizeof(int) * Li_ptr[n]);
    h = fnv(h, c, sizeof(int) * (n + 1));
    h = fnv(h, r, sizeof(int) * c[n]);            // A's rows and L's row pointers decide where A lands in L
    h = fnv(h, Li_ptr, sizeof(size_t) * (n + 1));
    h = fnv(h, lC, sizeof(size_t) * (n + 1));
    h = fnv(h, aTree, sizeof(int) * supNo);
    CacheKey key{blockSet, lR, c, h, n, supNo, 0};
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    parsy_plan* pl = parsy_plan_create(n, supNo, blockSet, lC, Li_ptr, lR, aTree, col2Sup, cT, rT, c, r,
                                       pick_device());
    if (pl) g_plans[key] = pl;
    return pl;
}

// the PRUNE operator: no etree, no upper pattern; the supernodal etree comes from L's pattern
parsy_plan* cached_chol_plan_prune(int n, int supNo, const int* blockSet, const size_t* lC, const size_t* Li_ptr,
                                   const int* lR, const int* prunePtr, const int* pruneSet, const int* c,
                                   const int* r) {
    uint64_t h = 1469598103934665603ULL;
    h = fnv(h, blockSet, sizeof(int) * (supNo + 1));
    h = fnv(h, lR, sizeof(int) * Li_ptr[n]);
    h = fnv(h, c, sizeof(int) * (n + 1));
    h = fnv(h, r, sizeof(int) * c[n]);
    h = fnv(h, Li_ptr, sizeof(size_t) * (n + 1));
    h = fnv(h, lC, sizeof(size_t) * (n + 1));
    h = fnv(h, prunePtr, sizeof(int) * (supNo + 1));
    h = fnv(h, pruneSet, sizeof(int) * prunePtr[supNo]);
    CacheKey key{blockSet, lR, c, h, n, supNo, 2};
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    const int dev = pick_device();
    if (dev >= parsy_device_count()) {
        set_last_error("cholesky_left_par_05_prune: no usable HIP device " + std::to_string(dev) +
                       " (this library has no CPU fallback)");
        return nullptr;
    }
    std::vector<int> col2sup(n), sparent(supNo, -1);
    for (int s = 0; s < supNo; ++s)
        for (int k = blockSet[s]; k < blockSet[s + 1]; ++k) col2sup[k] = s;
    for (int s = 0; s < supNo; ++s) {
        const int w = blockSet[s + 1] - blockSet[s];
        const size_t b = Li_ptr[blockSet[s]], e = Li_ptr[blockSet[s + 1]];
        if (e - b > (size_t)w) sparent[s] = col2sup[lR[b + w]];
    }
    parsy::PatternRef P;
    P.n = n;
    P.nsuper = supNo;
    P.super = blockSet;
    P.col2sup = col2sup.data();
    P.sparent = sparent.data();
    P.i_ptr = Li_ptr;
    P.s = lR;
    P.prunePtr = prunePtr;
    P.pruneSet = pruneSet;
    parsy_plan* pl = parsy::plan_build(P, lC, c, r, dev);
    if (pl) g_plans[key] = pl;
    return pl;
}

parsy_plan* cached_solve_plan(int n, int supNo, const size_t* Lp, const int* Li, const size_t* Li_ptr,
                              const int* sup2col) {
    uint64_t h = 1469598103934665603ULL;
    h = fnv(h, sup2col, sizeof(int) * (supNo + 1));
    h = fnv(h, Li, sizeof(int) * Li_ptr[n]);
    h = fnv(h, Li_ptr, sizeof(size_t) * (n + 1));
    h = fnv(h, Lp, sizeof(size_t) * (n + 1));
    CacheKey key{sup2col, Li, Lp, h, n, supNo, 1};
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    // supernodal etree straight from L's pattern: the parent of a supernode is the
    // supernode of its first below-diagonal row.
    std::vector<int> col2sup(n), sparent(supNo, -1);
    for (int s = 0; s < supNo; ++s)
        for (int k = sup2col[s]; k < sup2col[s + 1]; ++k) col2sup[k] = s;
    for (int s = 0; s < supNo; ++s) {
        const int w = sup2col[s + 1] - sup2col[s];
        const size_t b = Li_ptr[sup2col[s]], e = Li_ptr[sup2col[s + 1]];
        if (e - b > (size_t)w) sparent[s] = col2sup[Li[b + w]];
    }
    parsy::PatternRef P;
    P.n = n;
    P.nsuper = supNo;
    P.super = sup2col;
    P.col2sup = col2sup.data();
    P.sparent = sparent.data();
    P.i_ptr = Li_ptr;
    P.s = Li;
    if (parsy_device_count() < 1) {
        set_last_error("no usable HIP device: the BCSC solve has no CPU fallback in this library");
        return nullptr;
    }
    parsy_plan* pl = parsy::plan_build(P, Lp, nullptr, nullptr, pick_device());
    if (pl) g_plans[key] = pl;
    return pl;
}

int dropin_solve(const char* who, int n, size_t* Lp, int* Li, double* Lx, size_t* Li_ptr,
                 int* sup2col, int supNo, double* x) {
    if (!Lp || !Li || !x) return 0;  // reference: Triangular_BCSC.h:24
    parsy_plan* pl = cached_solve_plan(n, supNo, Lp, Li, Li_ptr, sup2col);
    std::unique_lock<std::mutex> use;
    if (pl) use = std::unique_lock<std::mutex>(pl->use_mu);
    if (!pl || parsy_solve_host(pl, Lx, x, 1, n, nullptr) != 0) {
        loud(who);
        return 0;
    }
    return 1;
}

}  // namespace

extern "C" {

int parsy_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

parsy_plan* parsy_plan_create(int n, int supNo, const int* blockSet, const size_t* lC,
                              const size_t* Li_ptr, const int* lR, const int* aTree,
                              const int* col2Sup, const int* cT, const int* rT, const int* c,
                              const int* r, int device) {
    if (!blockSet || !lC || !Li_ptr || !lR || !aTree || !col2Sup || !cT || !rT || !c || !r || n < 0 ||
        supNo < 0) {
        set_last_error("parsy_plan_create: null or negative argument");
        return nullptr;
    }
    if (device >= 0 && device >= parsy_device_count()) {
        set_last_error("parsy_plan_create: no usable HIP device " + std::to_string(device) +
                       " (this library has no CPU fallback; pass device < 0 for a host-only schedule)");
        return nullptr;
    }
    parsy::PatternRef P;
    P.n = n;
    P.nsuper = supNo;
    P.super = blockSet;
    P.col2sup = col2Sup;
    P.sparent = aTree;
    P.i_ptr = Li_ptr;
    P.s = lR;
    P.A1p = cT;
    P.A1i = rT;
    return parsy::plan_build(P, lC, c, r, device);
}

parsy_plan* parsy_plan_from_symbolic(const parsy_symbolic* sym, int device) {
    const parsy::Symbolic* S = parsy_symbolic_cxx(sym);
    if (!S) {
        set_last_error("parsy_plan_from_symbolic: null symbolic");
        return nullptr;
    }
    return parsy_plan_create(S->n, S->nsuper, S->super.data(), S->p.data(), S->i_ptr.data(), S->s.data(),
                             S->sparent.data(), S->col2sup.data(), S->A1.p.data(), S->A1.i.data(),
                             S->A2.p.data(), S->A2.i.data(), device);
}

void parsy_plan_destroy(parsy_plan* plan) { parsy::plan_free(plan); }

int parsy_plan_get_info(const parsy_plan* pl, parsy_plan_info* o) {
    if (!pl || !o) return -1;
    const parsy::Schedule& S = pl->S;
    std::memset(o, 0, sizeof(*o));
    o->n = S.n;
    o->nsuper = S.nsuper;
    o->nlevels = S.nlevels;
    o->max_width = S.max_width;
    o->max_rows = S.max_rows;
    o->n_small = S.n_small;
    o->n_big = S.n_big;
    o->chol_launches = (int32_t)S.chol.size() + 1;  // + the A scatter
    o->solve_launches = (int32_t)S.solve.size();
    o->nnzA = S.nnzA;
    o->ssize = S.ssize;
    o->xsize = S.xsize;
    o->nnzL = S.nnzL;
    o->n_updates = (int64_t)S.upd.size();
    o->relpos_len = (int64_t)S.relpos.size();
    o->device_bytes = pl->device_bytes;
    o->flops_stored = S.flops_stored;
    o->update_flops = S.update_flops;
    o->reread_bytes = S.reread_bytes;
    o->inner_flops = S.inner_flops;
    o->tile_update_flops = S.tile_update_flops;
    o->big_flops = S.big_flops;
    o->big_entries = (int64_t)S.big_entries.size();
    o->big_tasks = (int32_t)S.big_tasks.size();
    o->n_pieces = (int32_t)S.csn.size();
    o->chol_levels = S.cnlevels;
    o->piece_width = S.piece_width;
    o->big_min_k = S.big_min_k;
    // as launched (active supernodes only)
    o->chol_subtrees = (int32_t)S.small_ranges.size() / 2;
    for (size_t b = 0; b + 1 < S.small_ranges.size(); b += 2) o->chol_subtree_supernodes += S.small_ranges[b + 1] - S.small_ranges[b];
    o->solve_subtrees = (int32_t)S.solve_small_ranges.size() / 2;
    for (size_t b = 0; b + 1 < S.solve_small_ranges.size(); b += 2)
        o->solve_subtree_supernodes += S.solve_small_ranges[b + 1] - S.solve_small_ranges[b];
    o->backsolve_launches = (int32_t)S.bsolve.size();
    for (const parsy::Launch& l : S.chol)
        if (l.kind == parsy::kLaunchDense) o->dense_tasks += l.count;
    o->dense_flops = S.dense_flops;
    o->dense_entries = S.n_dense_entries;
    o->solve_one = (S.solve_one ? 1 : 0) | (S.solve_one_back ? 2 : 0) | ((S.solve_one || S.solve_one_back) && S.one_subtrees ? 4 : 0);
    o->solve_one_blocks = S.solve_one ? (int32_t)S.one_f.sn.size() : S.solve_one_back ? (int32_t)S.one_back().sn.size() : 0;
    o->sub_mrhs_trees = (int32_t)S.sub_trees.size();
    o->sub_mrhs_slots = S.sub_max_slots;
    o->sub_mrhs_tiers = (int32_t)S.sub_tiers.size();
    o->sub_mrhs_cover_level = S.sub_cover_level;
    o->dense_strip_entries = S.n_strip_entries;
    return 0;
}

long long parsy_plan_chain_check(const parsy_plan* pl, int slots) {
    if (!pl || slots < 1) {
        set_last_error("parsy_plan_chain_check: null plan or slots < 1");
        return -1;
    }
    return (long long)simulate_chain(pl->S, slots);
}

long long parsy_plan_check(const parsy_plan* pl) {
    if (!pl) {
        set_last_error("parsy_plan_check: null plan");
        return -1;
    }
    std::string what;
    const long long bad = (long long)parsy::check_schedule(pl->S, what);
    if (bad) set_last_error("parsy_plan_check: " + what);
    return bad;
}

// The two masks are independent: the supernode mask restricts the solves (and, while no piece mask is set, the
// factorization: a piece follows its supernode); the piece mask restricts the factorization alone.
int parsy_plan_set_active(parsy_plan* pl, const uint8_t* mask) {
    if (!pl) return -1;
    std::vector<uint8_t> pieces;
    if (pl->piece_mask_set) pieces = pl->S.active_piece;
    parsy::build_launches(pl->S, mask, pl->piece_mask_set ? pieces.data() : nullptr);
    pl->sn_mask_set = mask != nullptr;
    return parsy::plan_upload_launches(pl);
}

int parsy_plan_set_active_pieces(parsy_plan* pl, const uint8_t* piece_mask) {
    if (!pl) return -1;
    std::vector<uint8_t> sn_mask;
    if (pl->sn_mask_set) sn_mask = pl->S.active;
    parsy::build_launches(pl->S, pl->sn_mask_set ? sn_mask.data() : nullptr, piece_mask);
    pl->piece_mask_set = piece_mask != nullptr;
    return parsy::plan_upload_launches(pl);
}

int parsy_plan_pieces(const parsy_plan* pl, int32_t* supernode, int32_t* level, int32_t* col0, int32_t* width,
                      int32_t* rows, int64_t* value_begin, int64_t* value_end) {
    if (!pl) return -1;
    const parsy::Schedule& S = pl->S;
    const int nc = (int)S.csn.size();
    for (int p = 0; p < nc; ++p) {
        const parsy::SnDesc& C = S.csn[(size_t)p];
        const parsy::SnDesc& R = S.sn[(size_t)S.csn_real[(size_t)p]];
        if (supernode) supernode[p] = S.csn_real[(size_t)p];
        if (level) level[p] = S.level_of.empty() ? 0 : S.level_of[(size_t)p];
        if (col0) col0[p] = C.c0;
        if (width) width[p] = C.w;
        if (rows) rows[p] = C.r;
        if (value_begin) value_begin[p] = R.px + (int64_t)C.rbias * R.r;
        if (value_end) value_end[p] = R.px + (int64_t)(C.rbias + C.w) * R.r;
    }
    return nc;
}

int parsy_plan_solve_levels(const parsy_plan* pl, int32_t* level) {
    if (!pl) return -1;
    const parsy::Schedule& S = pl->S;
    if (level)
        for (int l = 0; l < S.nlevels; ++l)
            for (int q = S.levelPtr[(size_t)l]; q < S.levelPtr[(size_t)l + 1]; ++q) level[S.levelSet[(size_t)q]] = l;
    return S.nlevels;
}

// Diagnostics (tools/big_stats.py; not part of the public header): the BIG tasks of the plan as rows of
// (task, launch = 2 * source level + push, K, rows in the row window, rows in the column window, identity map,
// first row of the row window - first row of the column window); returns the number of entries (out may be null).
int64_t parsy_debug_big_entries(const parsy_plan* pl, int32_t* out, int64_t cap) {
    if (!pl) return -1;
    const parsy::Schedule& S = pl->S;
    int64_t n = 0, t = 0;
    for (const parsy::Schedule::BigTask& b : S.big_all) {
        for (int64_t e = b.e0; e < b.e1; ++e, ++n) {
            if (!out || n >= cap) continue;
            const parsy::WaveEntry& E = S.big_entries[(size_t)e];
            int32_t* o = out + 7 * n;
            o[0] = (int32_t)t;
            o[1] = b.src_level * 2 + (b.next ? 0 : 1);
            o[2] = E.K;
            o[3] = E.mn & 255;
            o[4] = (E.mn >> 8) & 255;
            o[5] = ((E.mn >> 16) & 1) != 0;
            o[6] = E.ia - E.ja;
        }
        ++t;
    }
    return n;
}

// Diagnostics (tools/pair_stats.py): as parsy_debug_big_entries with the windows themselves -- rows of (task, launch, K,
// rows, columns, first row of the row window, first row of the column window, source (its rank among the panel
// offsets is all a reader needs: low 32 bits of src / 8 ... high bits), 1 = a dense entry of its task).
int64_t parsy_debug_big_windows(const parsy_plan* pl, int64_t* out, int64_t cap) {
    if (!pl) return -1;
    const parsy::Schedule& S = pl->S;
    int64_t n = 0, t = 0;
    for (const parsy::Schedule::BigTask& b : S.big_all) {
        for (int64_t e = b.e0; e < b.e1; ++e, ++n) {
            if (!out || n >= cap) continue;
            const parsy::WaveEntry& E = S.big_entries[(size_t)e];
            int64_t* o = out + 9 * n;
            o[0] = t;
            o[1] = b.src_level * 2 + (b.next ? 0 : 1);
            o[2] = E.K;
            o[3] = E.mn & 255;
            o[4] = (E.mn >> 8) & 255;
            o[5] = E.ia;
            o[6] = E.ja;
            o[7] = E.src;
            o[8] = e < b.em;
        }
        ++t;
    }
    return n;
}

// Diagnostics: the tasks of the DENSE launches in launch order as rows of (launch index, 8-wide k chunks).
int64_t parsy_debug_dense_tasks(const parsy_plan* pl, int32_t* out, int64_t cap) {
    if (!pl) return -1;
    const parsy::Schedule& S = pl->S;
    int64_t n = 0;
    int li = 0;
    for (const parsy::Launch& l : S.chol) {
        if (l.kind != parsy::kLaunchDense) continue;
        for (int q = l.first; q < l.first + l.count; ++q, ++n)
            if (out && n < cap) {
                out[2 * n] = li;
                out[2 * n + 1] = S.big_tasks[(size_t)q].part;
            }
        ++li;
    }
    return n;
}

// Diagnostics: the launches of the last collected profiled run as rows of (kind, level << 1 | side, work items, ms).
int64_t parsy_debug_launch_times(const parsy_plan* pl, double* out, int64_t cap) {
    if (!pl) return -1;
    const int64_t n = (int64_t)std::min(pl->pev_ms.size(), pl->pev_kind.size());
    for (int64_t i = 0; out && i < n && i < cap; ++i) {
        out[4 * i] = pl->pev_kind[(size_t)i];
        out[4 * i + 1] = (size_t)i < pl->pev_level.size() ? pl->pev_level[(size_t)i] : -1;
        out[4 * i + 2] = (size_t)i < pl->pev_count.size() ? pl->pev_count[(size_t)i] : 0;
        out[4 * i + 3] = pl->pev_ms[(size_t)i];
    }
    return n;
}

int parsy_factor_begin(parsy_plan* pl, const double* d_values, double* d_lValues, void* stream, int flags) {
    if (!pl || !d_values || !d_lValues) {
        set_last_error("parsy_factor_begin: null argument");
        return -1;
    }
    return parsy::plan_factor_begin(pl, d_values, d_lValues, (hipStream_t)stream, (flags & PARSY_FACTOR_NO_INIT) == 0);
}

int parsy_factor_level(parsy_plan* pl, int level, double* d_lValues, void* stream) {
    if (!pl || !d_lValues) {
        set_last_error("parsy_factor_level: null argument");
        return -1;
    }
    return parsy::plan_factor_levels(pl, level, level + 1, d_lValues, (hipStream_t)stream);
}

int parsy_factor_end(parsy_plan* pl, void* stream) {
    if (!pl) {
        set_last_error("parsy_factor_end: null plan");
        return -1;
    }
    return parsy::plan_factor_end(pl, (hipStream_t)stream);
}

int parsy_factor_device(parsy_plan* pl, const double* d_values, double* d_lValues, void* stream) {
    if (!pl || !d_values || !d_lValues) {
        set_last_error("parsy_factor_device: null argument");
        return -1;
    }
    return parsy::plan_factor(pl, d_values, d_lValues, (hipStream_t)stream);
}

int parsy_factor_device_ex(parsy_plan* pl, const double* d_values, double* d_lValues, void* stream,
                           int flags) {
    if (!pl || !d_values || !d_lValues) {
        set_last_error("parsy_factor_device_ex: null argument");
        return -1;
    }
    return parsy::plan_factor(pl, d_values, d_lValues, (hipStream_t)stream,
                              (flags & PARSY_FACTOR_NO_INIT) == 0);
}

int parsy_factor_status(parsy_plan* pl) {
    if (!pl || pl->device < 0) return -1;
    int v = 0;
    if (hipMemcpy(&v, pl->dp.info, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v >= 0x7f7f7f7f ? 0 : v;  // < 0: an in-launch wait timed out (internal error)
}

int parsy_solve_status(parsy_plan* pl) {
    if (!pl || pl->device < 0) return -1;
    int v = 0;
    const int* word = pl->solve_status_word ? pl->solve_status_word : pl->dp.sinfo;
    if (hipMemcpy(&v, word, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (v < 0) set_last_error("solve: a hand-off wait inside a chain launch timed out; x is not the solution");
    return v < 0 ? -1 : 0;
}

int parsy_solve_device(parsy_plan* pl, const double* d_lValues, double* d_x, int nrhs, int ldx,
                       void* stream) {
    if (!pl || !d_lValues || !d_x) {
        set_last_error("parsy_solve_device: null argument");
        return -1;
    }
    return parsy::plan_solve(pl, d_lValues, d_x, nrhs, ldx, (hipStream_t)stream);
}

double parsy_last_factor_ms(parsy_plan* pl) {
    float ms = -1;
    if (!pl || !pl->have_f) return -1;
    if (hipEventSynchronize(pl->ev_f1) != hipSuccess) return -1;
    if (hipEventElapsedTime(&ms, pl->ev_f0, pl->ev_f1) != hipSuccess) return -1;
    return ms;
}

double parsy_last_solve_ms(parsy_plan* pl) {
    float ms = -1;
    if (!pl || !pl->have_s) return -1;
    if (hipEventSynchronize(pl->ev_s1) != hipSuccess) return -1;
    if (hipEventElapsedTime(&ms, pl->ev_s0, pl->ev_s1) != hipSuccess) return -1;
    return ms;
}

int parsy_plan_profile(parsy_plan* pl, int enable) {
    if (!pl) return -1;
    pl->profile = enable != 0;
    if (enable == 2) {  // reset the accumulators
        for (int k = 0; k < 10; ++k) pl->kind_ms[k] = 0, pl->kind_launches[k] = 0;
        pl->profiled_runs = 0;
        pl->level_ms.clear();
    }
    return 0;
}

int parsy_plan_profile_collect(parsy_plan* pl) { return pl ? parsy::plan_collect_profile(pl) : -1; }

int parsy_plan_profile_get(parsy_plan* pl, double* kind_ms, int* kind_launches, int* runs) {
    if (!pl) return -1;
    for (int k = 0; k < 10; ++k) {  // PARSY_PROFILE_KINDS entries each
        if (kind_ms) kind_ms[k] = pl->kind_ms[k];
        if (kind_launches) kind_launches[k] = pl->kind_launches[k];
    }
    if (runs) *runs = pl->profiled_runs;
    return 0;
}

int parsy_plan_profile_levels(parsy_plan* pl, double* main_ms, double* side_ms) {
    if (!pl) return -1;
    const int nl = pl->S.cnlevels;
    for (int l = 0; l < nl; ++l) {
        const size_t a = (size_t)l << 1, b = a | 1;
        if (main_ms) main_ms[l] = a < pl->level_ms.size() ? pl->level_ms[a] : 0.0;
        if (side_ms) side_ms[l] = b < pl->level_ms.size() ? pl->level_ms[b] : 0.0;
    }
    return nl;
}

// The bands of levels of the pipelined host factorization and, per band, the runs of lValues that are final once
// the band is complete (a piece is final after the chain launch of its own level: everything that updates it comes
// from lower levels and is applied before that launch).  Pieces are in column order = lValues order, so consecutive
// pieces of one band are one run; runs separated by less than 128 K doubles are merged (the gap is copied early and
// again with its own band: harmless).  Built once per plan.
static void build_download_bands(parsy_plan* pl) {
    const parsy::Schedule& S = pl->S;
    const int nl = S.cnlevels, np = (int)S.csn.size();
    // band boundaries: a band ends with the level at which another eighth of the factor's bytes has become final
    // (few bands = few, long runs: every copy from device to pageable host memory has a fixed cost), the last band
    // with the last level
    {
        std::vector<double> bytes((size_t)nl, 0.0);
        for (int p = 0; p < np; ++p) bytes[(size_t)S.level_of[(size_t)p]] += 8.0 * S.csn[(size_t)p].w * S.csn[(size_t)p].ld;   // (ld = rows of the supernode)
        const double total = 8.0 * (double)S.xsize;
        pl->h_band_level.clear();
        double run = 0, next = total / 8;
        for (int l = 0; l < nl; ++l) {
            run += bytes[(size_t)l];
            if (l == nl - 1 || run >= next) {
                pl->h_band_level.push_back(l);
                while (next <= run) next += total / 8;
            }
        }
    }
    const size_t nb = pl->h_band_level.size();
    pl->h_band_runs.assign(nb, {});
    std::vector<int> band_of((size_t)nl, 0);
    for (size_t b = 0, l = 0; b < nb; ++b)
        for (; (int)l <= pl->h_band_level[b]; ++l) band_of[l] = (int)b;
    const int64_t gap = 131072;
    for (int p = 0; p < np; ++p) {
        const parsy::SnDesc& C = S.csn[(size_t)p];
        const parsy::SnDesc& R = S.sn[(size_t)S.csn_real[(size_t)p]];
        // (a piece's columns are whole columns of its supernode's panel)
        const int64_t a = R.px + (int64_t)C.rbias * R.r, e = R.px + (int64_t)(C.rbias + C.w) * R.r;
        auto& runs = pl->h_band_runs[(size_t)band_of[(size_t)S.level_of[(size_t)p]]];
        if (!runs.empty() && a - (runs.back().first + runs.back().second) <= gap && a >= runs.back().first)
            runs.back().second = std::max(runs.back().second, e - runs.back().first);
        else
            runs.push_back({a, e - a});
    }
}

int parsy_factor_host(parsy_plan* pl, const double* values, double* lValues, double* seconds) {
    if (!pl || !values || !lValues) {
        set_last_error("parsy_factor_host: null argument");
        return -1;
    }
    if (pl->device < 0) {
        set_last_error("parsy_factor_host: plan has no device");
        return -1;
    }
    const parsy::Schedule& S = pl->S;
    CAPI_HIP(hipSetDevice(pl->device), -1);
    if (!pl->h_values_dev) CAPI_HIP(hipMalloc((void**)&pl->h_values_dev, std::max<int64_t>(S.nnzA, 1) * 8), -1);
    if (!pl->h_L_dev) CAPI_HIP(hipMalloc((void**)&pl->h_L_dev, std::max<int64_t>(S.xsize, 1) * 8), -1);
    CAPI_HIP(hipMemcpy(pl->h_values_dev, values, (size_t)S.nnzA * 8, hipMemcpyHostToDevice), -1);
    // Small factors: kernels, then one download.  Large ones (PARSY_HOST_PIPELINE=0: never): the download of every
    // band of levels runs BEHIND the kernels of the levels above it -- a worker thread copies the runs of lValues
    // that a band has made final while this thread's stream goes on (Flan-class: 19.4 GB at PCIe speed take as long
    // as the kernels; one after the other the call was 0.78 s).
    // (read per call: the tests switch it; PARSY_HOST_PIPELINE=2 takes the pipelined path whatever the size)
    const char* pe = std::getenv("PARSY_HOST_PIPELINE");
    const bool pipeline_on = !(pe && pe[0] == '0'), pipeline_forced = pe && pe[0] == '2';
    if (!pipeline_on || (!pipeline_forced && S.xsize * 8 < (int64_t)256 << 20) || pl->profile || S.cnlevels < 1) {
        if (parsy::plan_factor(pl, pl->h_values_dev, pl->h_L_dev, nullptr) != 0) return -1;
        CAPI_HIP(hipDeviceSynchronize(), -1);
        if (seconds) *seconds = parsy_last_factor_ms(pl) * 1e-3;
        CAPI_HIP(hipMemcpy(lValues, pl->h_L_dev, (size_t)S.xsize * 8, hipMemcpyDeviceToHost), -1);
        return 0;
    }
    if (!pl->h_ready) {
        // streams, bands and events are made into locals and handed to the plan only when all of them exist: a setup
        // that failed half-way must not leave a plan that "pipelines" over no band at all (and downloads nothing)
        hipStream_t hs = nullptr, hc = nullptr;
        std::vector<hipEvent_t> evs;
        auto undo = [&] {
            for (hipEvent_t e : evs) (void)hipEventDestroy(e);
            if (hs) (void)hipStreamDestroy(hs);
            if (hc) (void)hipStreamDestroy(hc);
            pl->h_band_level.clear();
            pl->h_band_runs.clear();
            (void)hipGetLastError();
        };
        bool ok = hipStreamCreateWithFlags(&hs, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&hc, hipStreamNonBlocking) == hipSuccess;
        if (ok) {
            build_download_bands(pl);
            evs.resize(pl->h_band_level.size(), nullptr);
            for (hipEvent_t& e : evs)
                if (ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                    e = nullptr;
                    ok = false;
                }
            ok = ok && !evs.empty();
        }
        if (!ok) {   // the plain form: kernels, then one download
            evs.erase(std::remove(evs.begin(), evs.end(), (hipEvent_t) nullptr), evs.end());
            undo();
            if (parsy::plan_factor(pl, pl->h_values_dev, pl->h_L_dev, nullptr) != 0) return -1;
            CAPI_HIP(hipDeviceSynchronize(), -1);
            if (seconds) *seconds = parsy_last_factor_ms(pl) * 1e-3;
            CAPI_HIP(hipMemcpy(lValues, pl->h_L_dev, (size_t)S.xsize * 8, hipMemcpyDeviceToHost), -1);
            return 0;
        }
        pl->h_stream = hs;
        pl->h_copy = hc;
        pl->h_band_ev = evs;
        pl->h_ready = true;
    }
    const size_t nb = pl->h_band_level.size();
    std::atomic<int> recorded{0};
    std::atomic<int> failed{0};
    double* const dL = pl->h_L_dev;
    const int device = pl->device;
    std::thread worker([&, dL, device] {
        if (hipSetDevice(device) != hipSuccess) {
            failed = 1;
            return;
        }
        for (size_t b = 0; b < nb; ++b) {
            while (recorded.load(std::memory_order_acquire) <= (int)b && !failed.load()) std::this_thread::yield();
            if (failed.load()) return;
            if (hipEventSynchronize(pl->h_band_ev[b]) != hipSuccess) {
                failed = 1;
                return;
            }
            for (const auto& r : pl->h_band_runs[b])
                if (hipMemcpyAsync(lValues + r.first, dL + r.first, (size_t)r.second * 8, hipMemcpyDeviceToHost, pl->h_copy) !=
                    hipSuccess) {
                    failed = 1;
                    return;
                }
            if (hipStreamSynchronize(pl->h_copy) != hipSuccess) {
                failed = 1;
                return;
            }
        }
    });
    int rc = parsy::plan_factor_begin(pl, pl->h_values_dev, dL, pl->h_stream, true);
    size_t b = 0;
    for (int lev = 0; rc == 0 && lev < S.cnlevels; ++lev) {
        rc = parsy::plan_factor_levels(pl, lev, lev + 1, dL, pl->h_stream);
        if (rc == 0 && b < nb && pl->h_band_level[b] == lev) {
            if (hipEventRecord(pl->h_band_ev[b], pl->h_stream) != hipSuccess) rc = -1;
            ++b;
            recorded.store((int)b, std::memory_order_release);
        }
    }
    if (rc == 0) rc = parsy::plan_factor_end(pl, pl->h_stream);
    if (rc != 0) {
        failed = 1;
        worker.join();
        parsy::plan_factor_abort(pl, pl->h_stream);
        return -1;
    }
    const hipError_t es = hipStreamSynchronize(pl->h_stream);
    worker.join();
    if (es != hipSuccess || failed.load()) {
        set_last_error("parsy_factor_host: the pipelined download failed");
        return -1;
    }
    if (seconds) *seconds = parsy_last_factor_ms(pl) * 1e-3;
    return 0;
}

int parsy_solve_host(parsy_plan* pl, const double* lValues, double* x, int nrhs, int ldx,
                     double* seconds) {
    if (!pl || !lValues || !x) {
        set_last_error("parsy_solve_host: null argument");
        return -1;
    }
    if (pl->device < 0) {
        set_last_error("parsy_solve_host: plan has no device");
        return -1;
    }
    const parsy::Schedule& S = pl->S;
    CAPI_HIP(hipSetDevice(pl->device), -1);
    if (!pl->h_L_dev) CAPI_HIP(hipMalloc((void**)&pl->h_L_dev, std::max<int64_t>(S.xsize, 1) * 8), -1);
    const int64_t need = (int64_t)ldx * nrhs;
    if (pl->h_x_len < need) {
        if (pl->h_x_dev) (void)hipFree(pl->h_x_dev);
        pl->h_x_dev = nullptr;
        CAPI_HIP(hipMalloc((void**)&pl->h_x_dev, (size_t)need * 8), -1);
        pl->h_x_len = need;
    }
    CAPI_HIP(hipMemcpy(pl->h_L_dev, lValues, (size_t)S.xsize * 8, hipMemcpyHostToDevice), -1);
    CAPI_HIP(hipMemcpy(pl->h_x_dev, x, (size_t)need * 8, hipMemcpyHostToDevice), -1);
    if (parsy::plan_solve(pl, pl->h_L_dev, pl->h_x_dev, nrhs, ldx, nullptr) != 0) return -1;
    CAPI_HIP(hipDeviceSynchronize(), -1);
    if (parsy_solve_status(pl) != 0) return -1;  // (x stays untouched: it would not be the solution)
    if (seconds) *seconds = parsy_last_solve_ms(pl) * 1e-3;
    CAPI_HIP(hipMemcpy(x, pl->h_x_dev, (size_t)need * 8, hipMemcpyDeviceToHost), -1);
    return 0;
}

int parsy_backsolve_device(parsy_plan* pl, const double* d_lValues, double* d_x, int nrhs, int ldx,
                           void* stream) {
    if (!pl || !d_lValues || !d_x) {
        set_last_error("parsy_backsolve_device: null argument");
        return -1;
    }
    return parsy::plan_backsolve(pl, d_lValues, d_x, nrhs, ldx, (hipStream_t)stream);
}

int parsy_solve_levels_device(parsy_plan* pl, const double* d_lValues, double* d_x, int nrhs, int ldx, void* stream,
                              int level_begin, int level_end, int flags) {
    if (!pl || !d_lValues || !d_x) {
        set_last_error("parsy_solve_levels_device: null argument");
        return -1;
    }
    return parsy::plan_solve_levels(pl, d_lValues, d_x, nrhs, ldx, (hipStream_t)stream, level_begin, level_end, flags);
}

int parsy_copy_segments_device(double* d_dst, const double* d_src, const int64_t* d_dst_off,
                               const int64_t* d_src_off, const int32_t* d_len, int64_t nseg, void* stream) {
    if (nseg < 0 || (nseg > 0 && (!d_dst || !d_src || !d_dst_off || !d_src_off || !d_len))) {
        set_last_error("parsy_copy_segments_device: null argument");
        return -1;
    }
    parsy::launch_copy_segments(d_dst, d_src, d_dst_off, d_src_off, d_len, nseg, (hipStream_t)stream);
    CAPI_HIP(hipGetLastError(), -1);
    return 0;
}

int parsy_rhs_ones_device(parsy_plan* pl, const double* d_lValues, double* d_b, void* stream) {
    if (!pl || !d_lValues || !d_b || pl->device < 0) {
        set_last_error("parsy_rhs_ones_device: null argument or plan without a device");
        return -1;
    }
    CAPI_HIP(hipSetDevice(pl->device), -1);
    CAPI_HIP(hipMemsetAsync(d_b, 0, (size_t)pl->S.n * sizeof(double), (hipStream_t)stream), -1);
    parsy::launch_rhs_ones(pl->dp, pl->S.nsuper, pl->S.max_rows, d_lValues, d_b, (hipStream_t)stream);
    CAPI_HIP(hipGetLastError(), -1);
    return 0;
}

int parsy_solve2_host(parsy_plan* pl, const double* lValues, double* x, int nrhs, int ldx, int forward,
                      double* seconds) {
    if (!pl || !lValues || !x) {
        set_last_error("parsy_solve2_host: null argument");
        return -1;
    }
    if (pl->device < 0) {
        set_last_error("parsy_solve2_host: plan has no device");
        return -1;
    }
    const parsy::Schedule& S = pl->S;
    CAPI_HIP(hipSetDevice(pl->device), -1);
    if (!pl->h_L_dev) CAPI_HIP(hipMalloc((void**)&pl->h_L_dev, std::max<int64_t>(S.xsize, 1) * 8), -1);
    const int64_t need = (int64_t)ldx * nrhs;
    if (pl->h_x_len < need) {
        if (pl->h_x_dev) (void)hipFree(pl->h_x_dev);
        pl->h_x_dev = nullptr;
        CAPI_HIP(hipMalloc((void**)&pl->h_x_dev, (size_t)need * 8), -1);
        pl->h_x_len = need;
    }
    CAPI_HIP(hipMemcpy(pl->h_L_dev, lValues, (size_t)S.xsize * 8, hipMemcpyHostToDevice), -1);
    CAPI_HIP(hipMemcpy(pl->h_x_dev, x, (size_t)need * 8, hipMemcpyHostToDevice), -1);
    double sec = 0;
    if (forward) {
        if (parsy::plan_solve(pl, pl->h_L_dev, pl->h_x_dev, nrhs, ldx, nullptr) != 0) return -1;
        CAPI_HIP(hipDeviceSynchronize(), -1);
        if (parsy_solve_status(pl) != 0) return -1;
        sec += parsy_last_solve_ms(pl) * 1e-3;
    }
    if (parsy::plan_backsolve(pl, pl->h_L_dev, pl->h_x_dev, nrhs, ldx, nullptr) != 0) return -1;
    CAPI_HIP(hipDeviceSynchronize(), -1);
    if (parsy_solve_status(pl) != 0) return -1;
    sec += parsy_last_solve_ms(pl) * 1e-3;
    if (seconds) *seconds = sec;
    CAPI_HIP(hipMemcpy(x, pl->h_x_dev, (size_t)need * 8, hipMemcpyDeviceToHost), -1);
    return 0;
}

void parsy_dropin_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_plans) parsy::plan_free(kv.second);
    g_plans.clear();
}

// ---- drop-in operators ----------------------------------------------------------

bool cholesky_left_par_05(int n, int* c, int* r, double* values, size_t* lC, int* lR, size_t* Li_ptr,
                          double* lValues, int* blockSet, int supNo, double* timing, int* aTree,
                          int* cT, int* rT, int* col2Sup, int nLevels, int* levelPtr, int* levelSet,
                          int nPar, int* parPtr, int* partition, int chunk, int threads,
                          int super_max, int col_max, double* nodCost) {
    (void)levelSet; (void)nPar; (void)chunk; (void)threads; (void)super_max; (void)col_max; (void)nodCost;
    const char* who = "cholesky_left_par_05";
    const double t0 = now_s();
    if (!validate_partition(supNo, nLevels, levelPtr, parPtr, partition, who)) {
        loud(who);
        return false;
    }
    parsy_plan* pl = cached_chol_plan(n, supNo, blockSet, lC, Li_ptr, lR, aTree, col2Sup, cT, rT, c, r);
    double dev_s = 0;
    std::unique_lock<std::mutex> use;
    if (pl) use = std::unique_lock<std::mutex>(pl->use_mu);
    if (!pl || parsy_factor_host(pl, values, lValues, &dev_s) != 0) {
        loud(who);
        return false;
    }
    if (timing) {
        timing[0] = now_s() - t0;
        timing[1] = 0.0;
        timing[2] = dev_s;
    }
    return parsy_factor_status(pl) == 0;
}

bool cholesky_left_par_05_prune(int n, int* c, int* r, double* values, size_t* lC, int* lR, size_t* Li_ptr,
                                double* lValues, int* blockSet, int supNo, double* timing, int* prunePtr,
                                int* pruneSet, int nLevels, int* levelPtr, int* levelSet, int nPar, int* parPtr,
                                int* partition, int chunk, int threads, int super_max, int col_max,
                                double* nodCost) {
    (void)levelSet; (void)nPar; (void)chunk; (void)threads; (void)super_max; (void)col_max; (void)nodCost;
    const char* who = "cholesky_left_par_05_prune";
    const double t0 = now_s();
    if (!c || !r || !values || !lC || !lR || !Li_ptr || !lValues || !blockSet || !prunePtr || !pruneSet || n < 0 ||
        supNo < 0) {
        set_last_error(std::string(who) + ": null or negative argument");
        loud(who);
        return false;
    }
    if (!validate_partition(supNo, nLevels, levelPtr, parPtr, partition, who)) {
        loud(who);
        return false;
    }
    parsy_plan* pl = cached_chol_plan_prune(n, supNo, blockSet, lC, Li_ptr, lR, prunePtr, pruneSet, c, r);
    double dev_s = 0;
    std::unique_lock<std::mutex> use;
    if (pl) use = std::unique_lock<std::mutex>(pl->use_mu);
    if (!pl || parsy_factor_host(pl, values, lValues, &dev_s) != 0) {
        loud(who);
        return false;
    }
    if (timing) {
        timing[0] = now_s() - t0;
        timing[1] = 0.0;
        timing[2] = dev_s;
    }
    return parsy_factor_status(pl) == 0;
}

bool cholesky_left_par_waveFront(int n, int* c, int* r, double* values, size_t* lC, int* lR,
                                 size_t* Li_ptr, double* lValues, int* blockSet, int supNo,
                                 double* timing, int* aTree, int* cT, int* rT, int* col2Sup,
                                 int nLevels, int* levelPtr, int* levelSet, int chunk, int threads,
                                 int super_max, int col_max) {
    (void)chunk; (void)threads; (void)super_max; (void)col_max;
    const char* who = "cholesky_left_par_waveFront";
    const double t0 = now_s();
    if (!validate_levelset(supNo, nLevels, levelPtr, levelSet, who)) {
        loud(who);
        return false;
    }
    parsy_plan* pl = cached_chol_plan(n, supNo, blockSet, lC, Li_ptr, lR, aTree, col2Sup, cT, rT, c, r);
    double dev_s = 0;
    std::unique_lock<std::mutex> use;
    if (pl) use = std::unique_lock<std::mutex>(pl->use_mu);
    if (!pl || parsy_factor_host(pl, values, lValues, &dev_s) != 0) {
        loud(who);
        return false;
    }
    if (timing) {
        timing[0] = now_s() - t0;
        timing[1] = 0.0;
        timing[2] = dev_s;
    }
    const int st = parsy_factor_status(pl);
    if (st != 0)
        std::fprintf(stderr, "[parsy_amd] %s: non-positive pivot at column %d (the reference ignores LAPACK's info here)\n",
                     who, st);
    return true;
}

int blockedLsolve(int n, size_t* Lp, int* Li, double* Lx, int NNZ, size_t* Li_ptr, int* col2sup,
                  int* sup2col, int supNo, double* x) {
    (void)NNZ; (void)col2sup;
    return dropin_solve("blockedLsolve", n, Lp, Li, Lx, Li_ptr, sup2col, supNo, x);
}

int leveledBlockedLsolve(int n, size_t* Lp, int* Li, double* Lx, int NNZ, size_t* Li_ptr,
                         int* col2sup, int* sup2col, int supNo, double* x, int levels,
                         int* levelPtr, int* levelSet, int chunk) {
    (void)NNZ; (void)col2sup; (void)chunk;
    if (!Lp || !Li || !x) return 0;
    if (!validate_levelset(supNo, levels, levelPtr, levelSet, "leveledBlockedLsolve")) {
        loud("leveledBlockedLsolve");
        return 0;
    }
    return dropin_solve("leveledBlockedLsolve", n, Lp, Li, Lx, Li_ptr, sup2col, supNo, x);
}

int H2LeveledBlockedLsolve(int n, size_t* Lp, int* Li, double* Lx, int NNZ, size_t* Li_ptr,
                           int* col2sup, int* sup2col, int supNo, double* x, int levels,
                           int* levelPtr, int* levelSet, int parts, int* parPtr, int* partition,
                           int chunk) {
    (void)NNZ; (void)col2sup; (void)chunk; (void)levelSet; (void)parts;
    if (!Lp || !Li || !x) return 0;
    if (!validate_partition(supNo, levels, levelPtr, parPtr, partition, "H2LeveledBlockedLsolve")) {
        loud("H2LeveledBlockedLsolve");
        return 0;
    }
    return dropin_solve("H2LeveledBlockedLsolve", n, Lp, Li, Lx, Li_ptr, sup2col, supNo, x);
}

int H2LeveledBlockedLsolve_Peeled(int n, size_t* Lp, int* Li, double* Lx, int NNZ, size_t* Li_ptr,
                                  int* col2sup, int* sup2col, int supNo, double* x, int levels,
                                  int* levelPtr, int* levelSet, int parts, int* parPtr,
                                  int* partition, int chunk, int threads) {
    (void)NNZ; (void)col2sup; (void)chunk; (void)levelSet; (void)parts; (void)threads;
    if (!Lp || !Li || !x) return 0;
    if (!validate_partition(supNo, levels, levelPtr, parPtr, partition, "H2LeveledBlockedLsolve_Peeled")) {
        loud("H2LeveledBlockedLsolve_Peeled");
        return 0;
    }
    return dropin_solve("H2LeveledBlockedLsolve_Peeled", n, Lp, Li, Lx, Li_ptr, sup2col, supNo, x);
}

}  // extern "C"
