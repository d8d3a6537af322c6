// One process, several devices: the distributed factorization of dist.hpp behind the C ABI
// (include/parsy_amd.h section 5).  One plan, one lValues buffer and one stream per rank; after every level of
// the Cholesky view each finished piece is pulled by the ranks that need it -- a copy kernel on the RECEIVING
// rank's stream that reads the owner's buffer (peer access over xGMI between different devices; a plain
// device-to-device copy when ranks share a device) and writes the same segments of its own buffer, ordered
// behind the owner's level by an event.  No host synchronisation inside a factorization.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/parsy_amd.h"
#include "dist.hpp"
#include "errors.hpp"
#include "executor.hpp"
#include "inspector.hpp"
#include "plan_fwd.hpp"

using parsy::set_last_error;

#define MG_HIP(call, ret)                                                            \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            set_last_error(std::string(#call) + ": " + hipGetErrorString(e_));       \
            return ret;                                                              \
        }                                                                            \
    } while (0)

struct parsy_mg {
    int nranks = 0;
    std::vector<int> device;
    std::vector<parsy_plan*> plan;
    parsy_dist* dist = nullptr;
    std::vector<double*> L, values;          // per rank, on its device
    std::vector<hipStream_t> stream;
    std::vector<std::vector<hipEvent_t>> level_done;   // [rank][level]
    std::vector<hipEvent_t> t0, t1;
    struct DevMsg { const int64_t* off = nullptr; const int32_t* len = nullptr; };
    std::vector<DevMsg> dmsg;                // per message, on the RECEIVER's device
    std::vector<void*> owned;                // (device memory, freed with the matching device current)
    std::vector<int> owned_dev;
    int64_t xsize = 0, nnzA = 0;
    bool have_values = false;
};

namespace {

int upload(parsy_mg* mg, int dev, const void* host, size_t bytes, void** out) {
    MG_HIP(hipSetDevice(dev), -1);
    void* d = nullptr;
    MG_HIP(hipMalloc(&d, std::max<size_t>(bytes, 8)), -1);
    mg->owned.push_back(d);
    mg->owned_dev.push_back(dev);
    if (bytes) MG_HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice), -1);
    *out = d;
    return 0;
}

}  // namespace

extern "C" {

parsy_mg* parsy_mg_create(const parsy_symbolic* sym, int nranks, const int* devices, int block) {
    if (!sym || nranks < 1 || !devices) {
        set_last_error("parsy_mg_create: null argument or nranks < 1");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_last_error("parsy_mg_create: no HIP device is usable (the executor has no CPU fallback)");
        return nullptr;
    }
    for (int r = 0; r < nranks; ++r)
        if (devices[r] < 0 || devices[r] >= ndev) {
            set_last_error("parsy_mg_create: device " + std::to_string(devices[r]) + " does not exist");
            return nullptr;
        }
    parsy_mg* mg = new parsy_mg;
    mg->nranks = nranks;
    mg->device.assign(devices, devices + nranks);
    auto fail = [&]() -> parsy_mg* {
        parsy_mg_destroy(mg);
        return nullptr;
    };
    for (int r = 0; r < nranks; ++r) {
        parsy_plan* pl = parsy_plan_from_symbolic(sym, devices[r]);
        if (!pl) return fail();
        mg->plan.push_back(pl);
    }
    mg->dist = parsy_dist_create(mg->plan[0], nranks, block);
    if (!mg->dist) return fail();
    if (parsy_dist_check(mg->plan[0], mg->dist) != 0) return fail();
    const parsy::Dist& D = parsy_dist_cxx(mg->dist);
    const parsy::Schedule& S = parsy::plan_schedule(mg->plan[0]);
    mg->xsize = S.xsize;
    mg->nnzA = S.nnzA;
    // every rank factors the pieces it owns
    {
        std::vector<uint8_t> mask((size_t)D.npieces);
        for (int r = 0; r < nranks; ++r) {
            for (int p = 0; p < D.npieces; ++p) mask[(size_t)p] = D.owner[(size_t)p] == r;
            if (parsy_plan_set_active_pieces(mg->plan[(size_t)r], mask.data()) != 0) return fail();
        }
    }
    // peer access between every pair of different devices that exchange messages
    for (const parsy::DistMessage& M : D.msgs) {
        const int a = devices[M.dst], b = devices[M.src];
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) {
            set_last_error("parsy_mg_create: device " + std::to_string(a) + " cannot access device " + std::to_string(b));
            return fail();
        }
        if (hipSetDevice(a) != hipSuccess) return fail();
        const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
            set_last_error(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            return fail();
        }
        (void)hipGetLastError();
    }
    mg->L.assign((size_t)nranks, nullptr);
    mg->values.assign((size_t)nranks, nullptr);
    mg->stream.assign((size_t)nranks, nullptr);
    mg->t0.assign((size_t)nranks, nullptr);
    mg->t1.assign((size_t)nranks, nullptr);
    mg->level_done.resize((size_t)nranks);
    for (int r = 0; r < nranks; ++r) {
        if (hipSetDevice(devices[r]) != hipSuccess) return fail();
        void* d = nullptr;
        if (hipMalloc(&d, (size_t)std::max<int64_t>(mg->xsize, 1) * sizeof(double)) != hipSuccess) {
            set_last_error("parsy_mg_create: out of device memory for the lValues of rank " + std::to_string(r));
            return fail();
        }
        mg->L[(size_t)r] = (double*)d;
        if (hipMalloc(&d, (size_t)std::max<int64_t>(mg->nnzA, 1) * sizeof(double)) != hipSuccess) return fail();
        mg->values[(size_t)r] = (double*)d;
        if (hipStreamCreateWithFlags(&mg->stream[(size_t)r], hipStreamNonBlocking) != hipSuccess) return fail();
        if (hipEventCreate(&mg->t0[(size_t)r]) != hipSuccess || hipEventCreate(&mg->t1[(size_t)r]) != hipSuccess) return fail();
        mg->level_done[(size_t)r].resize((size_t)D.nlevels);
        for (hipEvent_t& e : mg->level_done[(size_t)r])
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail();
    }
    mg->dmsg.resize(D.msgs.size());
    for (size_t m = 0; m < D.msgs.size(); ++m) {
        const parsy::DistMessage& M = D.msgs[m];
        void *o = nullptr, *l = nullptr;
        if (upload(mg, devices[M.dst], M.off.data(), M.off.size() * sizeof(int64_t), &o) != 0) return fail();
        if (upload(mg, devices[M.dst], M.len.data(), M.len.size() * sizeof(int32_t), &l) != 0) return fail();
        mg->dmsg[m].off = (const int64_t*)o;
        mg->dmsg[m].len = (const int32_t*)l;
    }
    for (int r = 0; r < nranks; ++r) {
        (void)hipSetDevice(devices[r]);
        (void)hipDeviceSynchronize();
    }
    return mg;
}

void parsy_mg_destroy(parsy_mg* mg) {
    if (!mg) return;
    for (int r = 0; r < mg->nranks; ++r) {
        (void)hipSetDevice(mg->device[(size_t)r]);
        (void)hipDeviceSynchronize();
        if ((size_t)r < mg->L.size() && mg->L[(size_t)r]) (void)hipFree(mg->L[(size_t)r]);
        if ((size_t)r < mg->values.size() && mg->values[(size_t)r]) (void)hipFree(mg->values[(size_t)r]);
        if ((size_t)r < mg->stream.size() && mg->stream[(size_t)r]) (void)hipStreamDestroy(mg->stream[(size_t)r]);
        if ((size_t)r < mg->t0.size() && mg->t0[(size_t)r]) (void)hipEventDestroy(mg->t0[(size_t)r]);
        if ((size_t)r < mg->t1.size() && mg->t1[(size_t)r]) (void)hipEventDestroy(mg->t1[(size_t)r]);
        if ((size_t)r < mg->level_done.size())
            for (hipEvent_t e : mg->level_done[(size_t)r])
                if (e) (void)hipEventDestroy(e);
    }
    for (size_t k = 0; k < mg->owned.size(); ++k) {
        (void)hipSetDevice(mg->owned_dev[k]);
        (void)hipFree(mg->owned[k]);
    }
    for (parsy_plan* pl : mg->plan) parsy_plan_destroy(pl);
    if (mg->dist) parsy_dist_destroy(mg->dist);
    delete mg;
}

int parsy_mg_set_values(parsy_mg* mg, const double* values) {
    if (!mg || !values) {
        set_last_error("parsy_mg_set_values: null argument");
        return -1;
    }
    for (int r = 0; r < mg->nranks; ++r) {
        MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
        MG_HIP(hipMemcpy(mg->values[(size_t)r], values, (size_t)mg->nnzA * sizeof(double), hipMemcpyHostToDevice), -1);
    }
    mg->have_values = true;
    return 0;
}

static int mg_factor_enqueue(parsy_mg* mg, double* seconds) {
    const parsy::Dist& D = parsy_dist_cxx(mg->dist);
    const int nr = mg->nranks;
    for (int r = 0; r < nr; ++r) {
        MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
        MG_HIP(hipStreamSynchronize(mg->stream[(size_t)r]), -1);
    }
    const auto w0 = std::chrono::steady_clock::now();
    for (int r = 0; r < nr; ++r) {
        MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
        MG_HIP(hipEventRecord(mg->t0[(size_t)r], mg->stream[(size_t)r]), -1);
        if (parsy::plan_factor_begin(mg->plan[(size_t)r], mg->values[(size_t)r], mg->L[(size_t)r], mg->stream[(size_t)r], true) != 0)
            return -1;
    }
    for (int lev = 0; lev < D.nlevels; ++lev) {
        for (int r = 0; r < nr; ++r) {
            MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
            if (parsy::plan_factor_levels(mg->plan[(size_t)r], lev, lev + 1, mg->L[(size_t)r], mg->stream[(size_t)r]) != 0)
                return -1;
        }
        const int64_t m0 = D.level_msg0[(size_t)lev], m1 = D.level_msg0[(size_t)lev + 1];
        if (m0 == m1) continue;
        std::vector<char> sends((size_t)nr, 0);
        for (int64_t m = m0; m < m1; ++m) sends[(size_t)D.msgs[(size_t)m].src] = 1;
        for (int r = 0; r < nr; ++r)
            if (sends[(size_t)r]) {
                MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
                MG_HIP(hipEventRecord(mg->level_done[(size_t)r][(size_t)lev], mg->stream[(size_t)r]), -1);
            }
        for (int64_t m = m0; m < m1; ++m) {
            const parsy::DistMessage& M = D.msgs[(size_t)m];
            MG_HIP(hipSetDevice(mg->device[(size_t)M.dst]), -1);
            MG_HIP(hipStreamWaitEvent(mg->stream[(size_t)M.dst], mg->level_done[(size_t)M.src][(size_t)lev], 0), -1);
            parsy::launch_copy_segments(mg->L[(size_t)M.dst], mg->L[(size_t)M.src], mg->dmsg[(size_t)m].off,
                                        mg->dmsg[(size_t)m].off, mg->dmsg[(size_t)m].len, (int64_t)M.off.size(),
                                        mg->stream[(size_t)M.dst]);
        }
    }
    for (int r = 0; r < nr; ++r) {
        MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
        if (parsy::plan_factor_end(mg->plan[(size_t)r], mg->stream[(size_t)r]) != 0) return -1;
        MG_HIP(hipEventRecord(mg->t1[(size_t)r], mg->stream[(size_t)r]), -1);
    }
    for (int r = 0; r < nr; ++r) {
        MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
        MG_HIP(hipStreamSynchronize(mg->stream[(size_t)r]), -1);
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    for (int r = 0; r < nr; ++r) {
        const int st = parsy_factor_status(mg->plan[(size_t)r]);
        if (st != 0) {
            set_last_error("parsy_mg_factor: rank " + std::to_string(r) + " reports factorization status " + std::to_string(st));
            return st;
        }
    }
    return 0;
}

// One cleanup path for the step-by-step drivers below: on any error every rank's streams are drained,
// an open factorization is closed, and profile mode is switched off, so that the handle stays usable.
static void mg_abort(parsy_mg* mg) {
    for (int r = 0; r < mg->nranks; ++r) {
        (void)hipSetDevice(mg->device[(size_t)r]);
        parsy::plan_factor_abort(mg->plan[(size_t)r], mg->stream[(size_t)r]);
        parsy_plan_profile(mg->plan[(size_t)r], 0);
    }
    (void)hipGetLastError();
}

int parsy_mg_factor(parsy_mg* mg, double* seconds) {
    if (!mg || !mg->have_values) {
        set_last_error("parsy_mg_factor: null handle or no values (parsy_mg_set_values first)");
        return -1;
    }
    const int rc = mg_factor_enqueue(mg, seconds);
    if (rc < 0) mg_abort(mg);   // (rc > 0: a non-positive pivot -- the run completed)
    return rc;
}

static int mg_profile_run(parsy_mg* mg, double* main_ms, double* side_ms, double* copy_ms, std::vector<hipEvent_t>& c0,
                          std::vector<hipEvent_t>& c1) {
    const parsy::Dist& D = parsy_dist_cxx(mg->dist);
    const int nr = mg->nranks, nl = D.nlevels;
    if (copy_ms) std::fill(copy_ms, copy_ms + (size_t)nr * nl, 0.0);
    for (int r = 0; r < nr; ++r) {
        MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
        MG_HIP(hipStreamSynchronize(mg->stream[(size_t)r]), -1);
        MG_HIP(hipEventCreate(&c0[(size_t)r]), -1);
        MG_HIP(hipEventCreate(&c1[(size_t)r]), -1);
        parsy_plan_profile(mg->plan[(size_t)r], 2);
        if (parsy::plan_factor_begin(mg->plan[(size_t)r], mg->values[(size_t)r], mg->L[(size_t)r], mg->stream[(size_t)r], true) != 0)
            return -1;
        MG_HIP(hipStreamSynchronize(mg->stream[(size_t)r]), -1);
    }
    for (int lev = 0; lev < nl; ++lev) {
        for (int r = 0; r < nr; ++r) {   // one rank at a time: its step has the device to itself
            MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
            if (parsy::plan_factor_levels(mg->plan[(size_t)r], lev, lev + 1, mg->L[(size_t)r], mg->stream[(size_t)r]) != 0)
                return -1;
            MG_HIP(hipStreamSynchronize(mg->stream[(size_t)r]), -1);
        }
        for (int r = 0; r < nr; ++r) {   // the copies a rank receives, timed as one block
            bool any = false;
            MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
            for (int64_t m = D.level_msg0[(size_t)lev]; m < D.level_msg0[(size_t)lev + 1]; ++m) {
                const parsy::DistMessage& M = D.msgs[(size_t)m];
                if (M.dst != r) continue;
                if (!any) MG_HIP(hipEventRecord(c0[(size_t)r], mg->stream[(size_t)r]), -1);
                any = true;
                parsy::launch_copy_segments(mg->L[(size_t)M.dst], mg->L[(size_t)M.src], mg->dmsg[(size_t)m].off,
                                            mg->dmsg[(size_t)m].off, mg->dmsg[(size_t)m].len, (int64_t)M.off.size(),
                                            mg->stream[(size_t)r]);
            }
            if (any) {
                MG_HIP(hipEventRecord(c1[(size_t)r], mg->stream[(size_t)r]), -1);
                MG_HIP(hipStreamSynchronize(mg->stream[(size_t)r]), -1);
                float ms = 0;
                MG_HIP(hipEventElapsedTime(&ms, c0[(size_t)r], c1[(size_t)r]), -1);
                if (copy_ms) copy_ms[(size_t)r * nl + lev] = ms;
            }
        }
    }
    int status = 0;
    for (int r = 0; r < nr; ++r) {
        MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
        if (parsy::plan_factor_end(mg->plan[(size_t)r], mg->stream[(size_t)r]) != 0) return -1;
        MG_HIP(hipStreamSynchronize(mg->stream[(size_t)r]), -1);
        parsy_plan_profile_collect(mg->plan[(size_t)r]);
        parsy_plan_profile_levels(mg->plan[(size_t)r], main_ms ? main_ms + (size_t)r * nl : nullptr,
                                  side_ms ? side_ms + (size_t)r * nl : nullptr);
        parsy_plan_profile(mg->plan[(size_t)r], 0);
        const int st = parsy_factor_status(mg->plan[(size_t)r]);
        if (st != 0 && status == 0) status = st;
    }
    return status;
}

int parsy_mg_profile(parsy_mg* mg, double* main_ms, double* side_ms, double* copy_ms) {
    if (!mg || !mg->have_values) {
        set_last_error("parsy_mg_profile: null handle or no values (parsy_mg_set_values first)");
        return -1;
    }
    std::vector<hipEvent_t> c0((size_t)mg->nranks, nullptr), c1((size_t)mg->nranks, nullptr);
    const int rc = mg_profile_run(mg, main_ms, side_ms, copy_ms, c0, c1);
    if (rc < 0) mg_abort(mg);
    for (int r = 0; r < mg->nranks; ++r) {
        (void)hipSetDevice(mg->device[(size_t)r]);
        if (c0[(size_t)r]) (void)hipEventDestroy(c0[(size_t)r]);
        if (c1[(size_t)r]) (void)hipEventDestroy(c1[(size_t)r]);
    }
    return rc;
}

int parsy_mg_rank_ms(parsy_mg* mg, double* rank_ms) {
    if (!mg || !rank_ms) return -1;
    for (int r = 0; r < mg->nranks; ++r) {
        float ms = -1;
        MG_HIP(hipSetDevice(mg->device[(size_t)r]), -1);
        MG_HIP(hipEventElapsedTime(&ms, mg->t0[(size_t)r], mg->t1[(size_t)r]), -1);
        rank_ms[r] = ms;
    }
    return 0;
}

int parsy_mg_gather_host(parsy_mg* mg, double* lValues) {
    if (!mg || !lValues) {
        set_last_error("parsy_mg_gather_host: null argument");
        return -1;
    }
    const parsy::Dist& D = parsy_dist_cxx(mg->dist);
    const parsy::Schedule& S = parsy::plan_schedule(mg->plan[0]);
    // runs of consecutive pieces with one owner are contiguous in lValues (pieces are in column order)
    for (int p = 0; p < D.npieces;) {
        int q = p;
        while (q + 1 < D.npieces && D.owner[(size_t)q + 1] == D.owner[(size_t)p]) ++q;
        const parsy::SnDesc& R0 = S.sn[(size_t)S.csn_real[(size_t)p]];
        const parsy::SnDesc& R1 = S.sn[(size_t)S.csn_real[(size_t)q]];
        const int64_t a = R0.px + (int64_t)S.csn[(size_t)p].rbias * R0.r;
        const int64_t b = R1.px + (int64_t)(S.csn[(size_t)q].rbias + S.csn[(size_t)q].w) * R1.r;
        const int rk = D.owner[(size_t)p];
        MG_HIP(hipSetDevice(mg->device[(size_t)rk]), -1);
        MG_HIP(hipMemcpy(lValues + a, mg->L[(size_t)rk] + a, (size_t)(b - a) * sizeof(double), hipMemcpyDeviceToHost), -1);
        p = q + 1;
    }
    return 0;
}

const parsy_dist* parsy_mg_dist(const parsy_mg* mg) { return mg ? mg->dist : nullptr; }
parsy_plan* parsy_mg_plan(parsy_mg* mg, int rank) {
    return (mg && rank >= 0 && rank < mg->nranks) ? mg->plan[(size_t)rank] : nullptr;
}

}  // extern "C"
