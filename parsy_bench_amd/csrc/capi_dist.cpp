// C ABI of the distribution module (include/parsy_amd.h, "Distribution of one factorization ..."): host only.
#include <algorithm>
#include <exception>
#include <string>

#include "../../include/parsy_amd.h"
#include "dist.hpp"
#include "errors.hpp"
#include "plan_fwd.hpp"

struct parsy_dist {
    parsy::Dist D;
};

using parsy::set_last_error;

extern "C" {

parsy_dist* parsy_dist_create(const parsy_plan* plan, int nranks, int block) {
    if (!plan || nranks < 1) {
        set_last_error("parsy_dist_create: null plan or nranks < 1");
        return nullptr;
    }
    parsy_dist* d = new parsy_dist;
    try {
        parsy::build_dist(parsy::plan_schedule(plan), nranks, block <= 0 ? parsy::kDistBlock : block, d->D);
    } catch (const std::exception& e) {
        set_last_error(std::string("parsy_dist_create: ") + e.what());
        delete d;
        return nullptr;
    }
    return d;
}

void parsy_dist_destroy(parsy_dist* dist) { delete dist; }

int parsy_dist_get_info(const parsy_dist* dist, parsy_dist_info* o) {
    if (!dist || !o) return -1;
    const parsy::Dist& D = dist->D;
    o->nranks = D.nranks;
    o->nlevels = D.nlevels;
    o->n_pieces = D.npieces;
    o->n_subtrees = D.n_subtrees;
    o->n_root_pieces = D.n_root_pieces;
    o->n_messages = (int32_t)D.msgs.size();
    o->exchange_elements = D.exchange_elements;
    o->total_cost = D.total_cost;
    o->root_cost = D.root_cost;
    o->max_rank_cost = D.rank_cost.empty() ? 0.0 : *std::max_element(D.rank_cost.begin(), D.rank_cost.end());
    o->lockstep_cost = 0;
    for (int l = 0; l < D.nlevels; ++l)
        o->lockstep_cost += *std::max_element(D.level_cost.begin() + (size_t)l * D.nranks,
                                              D.level_cost.begin() + (size_t)(l + 1) * D.nranks);
    return 0;
}

int parsy_dist_get(const parsy_dist* dist, int32_t* owner, uint8_t* in_subtree, double* rank_cost,
                   double* level_cost) {
    if (!dist) return -1;
    const parsy::Dist& D = dist->D;
    if (owner) std::copy(D.owner.begin(), D.owner.end(), owner);
    if (in_subtree) std::copy(D.in_subtree.begin(), D.in_subtree.end(), in_subtree);
    if (rank_cost) std::copy(D.rank_cost.begin(), D.rank_cost.end(), rank_cost);
    if (level_cost) std::copy(D.level_cost.begin(), D.level_cost.end(), level_cost);
    return 0;
}

int parsy_dist_level_messages(const parsy_dist* dist, int level) {
    if (!dist || level < 0 || level >= dist->D.nlevels) return -1;
    return (int)(dist->D.level_msg0[(size_t)level + 1] - dist->D.level_msg0[(size_t)level]);
}

int parsy_dist_message(const parsy_dist* dist, int level, int index, int32_t* src, int32_t* dst, int64_t* nseg,
                       int64_t* total, const int64_t** off, const int32_t** len, const int64_t** packed) {
    const int count = parsy_dist_level_messages(dist, level);
    if (count < 0 || index < 0 || index >= count) {
        set_last_error("parsy_dist_message: no such message");
        return -1;
    }
    const parsy::DistMessage& M = dist->D.msgs[(size_t)(dist->D.level_msg0[(size_t)level] + index)];
    if (src) *src = M.src;
    if (dst) *dst = M.dst;
    if (nseg) *nseg = (int64_t)M.off.size();
    if (total) *total = M.total;
    if (off) *off = M.off.data();
    if (len) *len = M.len.data();
    if (packed) *packed = M.packed.data();
    return 0;
}

long long parsy_dist_check(const parsy_plan* plan, const parsy_dist* dist) {
    if (!plan || !dist) {
        set_last_error("parsy_dist_check: null argument");
        return -1;
    }
    std::string what;
    const long long bad = (long long)parsy::check_dist(parsy::plan_schedule(plan), dist->D, what);
    if (bad) set_last_error("parsy_dist_check: " + what);
    return bad;
}

}  // extern "C"

const parsy::Dist& parsy_dist_cxx(const parsy_dist* d) { return d->D; }
