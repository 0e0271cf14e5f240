// Several GPUs in one process (reference: the critical-section accumulate of HaploCart.cpp:408-421, one context per GPU instead of one
// thread per core): the contexts' vectors summed onto the first -- ncclReduce over xGMI between distinct devices (RCCL loaded at run
// time), through the host otherwise.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <chrono>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "hc_ctx.h"
#include "hc_device.h"
#include "host/common.h"
#include "vgan_gpu.h"

using namespace vgan;

#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(VGAN_ENODEV, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

// ---------------------------------------------------------------------------------------------- several GPUs, one process
// RCCL is bound at run time (dlopen) and only on this path: the library carries no link-time dependency on it, and a
// process that already holds another RCCL (PyTorch ships its own) never sees two.
namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why_not; // (not ok: what was missing)
    Rccl() {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) {
            const char *e = dlerror();
            why_not = std::string("librccl could not be loaded") + (e ? std::string(": ") + e : std::string());
            return;
        }
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        Reduce = (decltype(Reduce))dlsym(h, "ncclReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Reduce;
        if (!ok) why_not = "librccl lacks one of ncclCommInitAll / ncclCommDestroy / ncclGroupStart / ncclGroupEnd / ncclReduce";
    }
};
Rccl &rccl() {
    static Rccl r;
    return r;
}
// one communicator per set of devices, for the life of the process (never destroyed: RCCL tears down with the runtime)
struct CommCache {
    std::mutex mu;
    std::map<std::vector<int>, std::vector<ncclComm_t>> m;
    double last_setup_ms = 0.0, last_reduce_ms = 0.0; // (the reduce's wall time includes a set-up made inside it)
    int n_setups = 0, last_was_rccl = 0;
    std::string last_why; // why the last reduce was summed on the host ("" when it went through RCCL)
};
CommCache &comm_cache() {
    static CommCache c;
    return c;
}
} // namespace

// Sum over contexts of final_vec (src/HaploCart.cpp:419-420, the accumulate the reference does under `omp critical`, here
// across GPUs): every context finalizes on its own device, then ONE reduce of P doubles onto the first context's device --
// ncclReduce over the contexts' streams when they sit on distinct devices (xGMI), through the host otherwise (several
// contexts on one device, RCCL not loadable).  out: host double[P].  *used_rccl (or NULL) tells which way it went.
extern "C" int vgan_hc_reduce(vgan_hc_ctx **ctxs, int n, double *out, int *used_rccl) {
    if (!ctxs || n <= 0 || !out) return fail(VGAN_EINVAL, "vgan_hc_reduce: null argument");
    for (int i = 0; i < n; ++i)
        if (!ctxs[i] || ctxs[i]->P != ctxs[0]->P) return fail(VGAN_EINVAL, "vgan_hc_reduce: contexts of different graphs");
    const uint32_t P = ctxs[0]->P;
    int rc;
    if (used_rccl) *used_rccl = 0;
    // contexts nothing was accumulated into add nothing: a short input that reached one GPU only is that context's finalize,
    // whatever the number of contexts standing by
    std::vector<vgan_hc_ctx *> live;
    for (int i = 0; i < n; ++i)
        if (ctxs[i]->touched) live.push_back(ctxs[i]);
    if (live.size() <= 1) return vgan_hc_finalize(live.empty() ? ctxs[0] : live[0], nullptr, out);
    const bool partial = (int)live.size() < n; // (a communicator is per device set: a partial set goes through the host)
    if (partial) {
        ctxs = live.data();
        n = (int)live.size();
    }
    for (int i = 0; i < n; ++i)
        if ((rc = vgan_hc_finalize(ctxs[i], nullptr, nullptr))) return rc; // final_vec on every device, asynchronously
    bool distinct = n > 1 && !partial;
    for (int i = 0; i < n && distinct; ++i)
        for (int j = 0; j < i; ++j) distinct = distinct && ctxs[i]->device != ctxs[j]->device;
    // Contexts on distinct devices reduce with ncclReduce over xGMI (BASELINE.json's north_star: reads shard across GPUs, one RCCL
    // reduce of the per-path vector): the communicator is created once per device set and kept for the life of the process
    // (its set-up time is reported: vgan_hc_reduce_info).  VGAN_HC_REDUCE=host keeps the sum on the host -- for ONE reduce of
    // 41 KB per context that is the cheaper way, a communicator costs more to set up than it saves --, contexts sharing a
    // device always take it, and so does a set for which RCCL cannot be loaded or initialised.
    const auto t_red0 = std::chrono::steady_clock::now();
    const char *how = getenv("VGAN_HC_REDUCE");
    std::string why; // the host sum is never taken silently: vgan_hc_reduce_why() says what sent the reduce there
    if (!distinct) why = partial ? "a chunk reached only some of the contexts (a communicator is per device set)" : "contexts share a device";
    else if (how && strcmp(how, "host") == 0) why = "VGAN_HC_REDUCE=host";
    else if (!rccl().ok) why = rccl().why_not;
    if (distinct && rccl().ok && !(how && strcmp(how, "host") == 0)) {
        std::vector<int> devs((size_t)n);
        for (int i = 0; i < n; ++i) devs[(size_t)i] = ctxs[i]->device;
        std::vector<ncclComm_t> *comms = nullptr;
        {
            std::lock_guard<std::mutex> lk(comm_cache().mu);
            auto it = comm_cache().m.find(devs);
            if (it != comm_cache().m.end()) {
                comms = &it->second;
            } else {
                std::vector<ncclComm_t> fresh((size_t)n, nullptr);
                const auto t0 = std::chrono::steady_clock::now();
                const ncclResult_t ir = rccl().CommInitAll(fresh.data(), n, devs.data());
                if (ir == ncclSuccess) {
                    comm_cache().last_setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    comm_cache().n_setups += 1;
                    comms = &(comm_cache().m[devs] = std::move(fresh));
                } else {
                    why = std::string("ncclCommInitAll failed: ") + (rccl().GetErrorString ? rccl().GetErrorString(ir) : "unknown error");
                }
            }
        }
        if (comms) {
            bool good = rccl().GroupStart() == ncclSuccess;
            for (int i = 0; i < n && good; ++i) {
                good = hipSetDevice(ctxs[i]->device) == hipSuccess &&
                       rccl().Reduce(ctxs[i]->final_vec.p, ctxs[i]->final_vec.p, P, ncclDouble, ncclSum, 0, (*comms)[(size_t)i], ctxs[i]->stream) == ncclSuccess;
            }
            good = rccl().GroupEnd() == ncclSuccess && good;
            for (int i = 0; i < n; ++i) {
                (void)hipSetDevice(ctxs[i]->device);
                good = hipStreamSynchronize(ctxs[i]->stream) == hipSuccess && good;
            }
            if (!good) return fail(VGAN_ENODEV, "vgan_hc_reduce: the RCCL reduce failed");
            HIPCHK(hipSetDevice(ctxs[0]->device));
            HIPCHK(hipMemcpy(out, ctxs[0]->final_vec.p, (size_t)P * 8, hipMemcpyDeviceToHost));
            if (used_rccl) *used_rccl = 1;
            std::lock_guard<std::mutex> lk(comm_cache().mu);
            comm_cache().last_reduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_red0).count();
            comm_cache().last_was_rccl = 1;
            comm_cache().last_why.clear();
            return VGAN_OK;
        }
    }
    std::vector<double> part(P);
    for (uint32_t p = 0; p < P; ++p) out[p] = 0.0;
    for (int i = 0; i < n; ++i) {
        HIPCHK(hipSetDevice(ctxs[i]->device));
        HIPCHK(hipMemcpyAsync(part.data(), ctxs[i]->final_vec.p, (size_t)P * 8, hipMemcpyDeviceToHost, ctxs[i]->stream));
        HIPCHK(hipStreamSynchronize(ctxs[i]->stream));
        for (uint32_t p = 0; p < P; ++p) out[p] += part[p];
    }
    {
        std::lock_guard<std::mutex> lk(comm_cache().mu);
        comm_cache().last_reduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_red0).count();
        comm_cache().last_was_rccl = 0;
        comm_cache().last_why = why;
    }
    return VGAN_OK;
}

extern "C" int vgan_hc_reduce_why(char *buf, int64_t cap) {
    std::lock_guard<std::mutex> lk(comm_cache().mu);
    if (buf && cap > 0) snprintf(buf, (size_t)cap, "%s", comm_cache().last_why.c_str());
    return comm_cache().last_was_rccl;
}

extern "C" int vgan_hc_reduce_last(double *reduce_ms, int *was_rccl) {
    std::lock_guard<std::mutex> lk(comm_cache().mu);
    if (reduce_ms) *reduce_ms = comm_cache().last_reduce_ms;
    if (was_rccl) *was_rccl = comm_cache().last_was_rccl;
    return VGAN_OK;
}

extern "C" int vgan_hc_reduce_info(double *last_setup_ms, int *n_setups) {
    std::lock_guard<std::mutex> lk(comm_cache().mu);
    if (last_setup_ms) *last_setup_ms = comm_cache().last_setup_ms;
    if (n_setups) *n_setups = comm_cache().n_setups;
    return VGAN_OK;
}
