// vgan soibean over the device front end's pipeline (reference: src/getLCAfromGAM.h:31-45 -- analyse_GAM's loop over the
// stream -- feeding the per-read body of :92-186): the consumer of csrc/gam_pipe.hip's pieces for soibean.  A piece's reads go through
// soibean's flatten as kernels (sb_flatten_kernels.hip) into the batch of the lane's context, which grows in HBM piece by piece; the
// reads the device flatten leaves (indels, soft clips, anything the reference would index out of bounds on) go through the host's parser
// and vgan_sb_flatten behind the slot's back and are appended at the end; then analyse_GAM's tables are made ONCE per context over its
// whole batch (vgan_sb_precompute).  Which context a read lands in and where in its batch does not change a bit of what the chains
// compute: their sums over reads are integers (csrc/sb_device.h: SbFix), the signature counts are counts.  No kernel lives here.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gam_device.h"
#include "gam_object.h"
#include "host/common.h"
#include "sb_device.h"
#include "vgan_gpu.h"

using namespace vgan;
using namespace vgan::gd;

namespace {
double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
struct SbLeftJob { // the reads of one piece that the device flatten leaves to the host
    std::thread t;
    int rc = VGAN_OK;
    std::string err;
    int lane = 0;
    int64_t piece = 0;
    vgan_sb_host_batch *hb = nullptr;
    std::vector<uint32_t> where; // the reads' places among the file's mapped reads
    vgan_sb_flatten_stats st{};
    ~SbLeftJob() { vgan_sb_host_batch_free(hb); }
};
} // namespace

struct vgan_sb_gamrun : GamConsumer {
    std::vector<int> devices;
    const void *bytes = nullptr;
    uint64_t n = 0;
    vgan_gampipe_opts opts{};
    std::thread coord;
    int rc = VGAN_OK;
    std::string err;
    vgan_gampipe_stats pst{};
    std::mutex mu;
    std::condition_variable cv;
    bool attached = false, gave_up = false, finished = false;
    std::vector<vgan_sb_ctx *> ctx;
    const vgan_graph *graph = nullptr;
    std::vector<vgan_sb_devflat *> df;
    std::deque<std::mutex> lane_mu;
    std::vector<std::shared_ptr<SbLeftJob>> bg;
    vgan_sb_gam_result res{};
    uint64_t n_host_reads = 0, n_device_reads = 0;
    double ms_wait_contexts = 0, ms_tables = 0;
    int host_threads = 2;

    void aborted() override {
        {
            std::lock_guard<std::mutex> lk(mu);
            gave_up = true;
        }
        cv.notify_all();
    }
    int consume(int lane, vgan_gamdev *g, uint64_t read_base, const uint8_t *, int64_t piece) override {
        {
            const auto t0 = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return attached || gave_up; });
            if (!attached) return fail(VGAN_ESTATE, "vgan_sb_gam: no contexts were attached");
            ms_wait_contexts = std::max(ms_wait_contexts, ms_since(t0));
        }
        uint64_t sz[8];
        (void)vgan_gamdev_sizes(g, sz, nullptr);
        const uint64_t R = sz[2];
        if (read_base + R > 0xFFFFFFF0ull) return fail(VGAN_ERANGE, "vgan_sb_gam: more than 2^32 reads");
        std::vector<uint8_t> mask((size_t)R, 0);
        int r;
        {
            std::lock_guard<std::mutex> lk(lane_mu[(size_t)lane]); // (the lane's batch and its stream: one piece at a time)
            const size_t l = (size_t)lane;
            if (!df[l]) {
                if ((r = vgan_sb_devflat_create(ctx[l], graph, &df[l])) < 0) return r;
                // the lane's share of the input over this piece: the batch's arrays are sized once
                const double pieces = (double)n / (double)std::max<uint64_t>(opts.piece_bytes, 1) / (double)devices.size();
                (void)vgan_sb_devflat_expect(df[l], std::max(1.0, pieces * 1.1 + 0.5));
            }
            vgan_sb_flatten_stats st{};
            if ((r = vgan_sb_devflat_append_gamdev(df[l], g, (uint32_t)read_base, mask.data(), &st)) < 0) return r;
            std::lock_guard<std::mutex> lk2(mu);
            n_device_reads += (uint64_t)st.n_out;
        }
        // ---- the reads left to the host: their messages down now (the object's buffers are the next piece's after this call), the rest behind
        auto job = std::make_shared<SbLeftJob>();
        for (uint64_t i = 0; i < R; ++i)
            if (mask[(size_t)i]) job->where.push_back((uint32_t)(read_base + i));
        if (job->where.empty()) return VGAN_OK;
        uint64_t nm = 0, nb = 0;
        if ((r = vgan_gamdev_pick(g, mask.data(), &nm, &nb)) < 0) return r;
        if (nm != job->where.size()) return fail(VGAN_ESTATE, "vgan_sb_gam: %llu messages picked for %zu reads", (unsigned long long)nm, job->where.size());
        auto offs = std::make_shared<std::vector<uint64_t>>((size_t)nm + 1);
        auto msgs = std::make_shared<std::vector<uint8_t>>((size_t)std::max<uint64_t>(nb, 1));
        if ((r = vgan_gamdev_picked(g, offs->data(), msgs->data())) < 0) return r;
        job->lane = lane;
        job->piece = piece;
        vgan_sb_gamrun *self = this;
        job->t = std::thread([self, job, offs, msgs] {
            vgan_alnparts *parts = nullptr;
            vgan_alnset merged;
            const size_t want = job->where.size();
            // (keep_unmapped: the parse on the device dropped identity == 0 already; every message handed back is a read)
            if ((job->rc = vgan_alnparts_from_messages(msgs->data(), offs->data(), (int64_t)want, 1, self->host_threads, &parts)) >= 0) {
                merge_alnsets(parts->parts, merged);
                vgan_alnparts_free(parts);
                if (merged.n_reads() != (int64_t)want) job->rc = fail(VGAN_ESTATE, "vgan_sb_gam: %lld reads parsed of %zu messages", (long long)merged.n_reads(), want);
            }
            if (job->rc >= 0) job->rc = vgan_sb_flatten(self->graph, &merged, 0, merged.n_reads(), self->host_threads, &job->hb, &job->st);
            if (job->rc < 0) job->err = last_error();
        });
        std::lock_guard<std::mutex> lk(mu);
        bg.push_back(job);
        n_host_reads += job->where.size();
        return VGAN_OK;
    }
};

extern "C" int vgan_sb_gam_start(const int *devices, int n_lanes, const void *bytes, uint64_t n, const vgan_gampipe_opts *opts, vgan_sb_gamrun **out) {
    if (!devices || n_lanes <= 0 || (!bytes && n) || !out) return fail(VGAN_EINVAL, "vgan_sb_gam_start: null argument");
    auto *r = new vgan_sb_gamrun();
    r->devices.assign(devices, devices + n_lanes);
    r->bytes = bytes;
    r->n = n;
    r->opts = gampipe_defaults(opts, n, n_lanes);
    r->opts.keep_unmapped = 0;   // getLCAfromGAM.h:101: identity == 0 is skipped
    r->opts.mark_duplicates = 0; // (soibean removes no duplicates)
    r->df.assign((size_t)n_lanes, nullptr);
    r->lane_mu.resize((size_t)n_lanes);
    const int cpus = r->opts.n_threads > 0 ? r->opts.n_threads : (int)usable_cpus();
    r->host_threads = std::max(1, std::min(8, cpus / std::max(1, n_lanes * r->opts.slots)));
    r->coord = std::thread([r] {
        r->rc = gampipe_run(r->bytes, r->n, r->devices, r->opts, *r, &r->pst);
        if (r->rc < 0) r->err = last_error();
    });
    *out = r;
    return VGAN_OK;
}

extern "C" int vgan_sb_gam_attach(vgan_sb_gamrun *r, vgan_sb_ctx *const *ctxs, int n_ctx, const vgan_graph *graph) {
    if (!r || !ctxs || !graph) return fail(VGAN_EINVAL, "vgan_sb_gam_attach: null argument");
    if (n_ctx != (int)r->devices.size()) return fail(VGAN_EINVAL, "vgan_sb_gam_attach: %d contexts for %zu lanes", n_ctx, r->devices.size());
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if (r->attached) return fail(VGAN_ESTATE, "vgan_sb_gam_attach: called twice");
        r->ctx.assign(ctxs, ctxs + n_ctx);
        r->graph = graph;
        r->attached = true;
    }
    r->cv.notify_all();
    return VGAN_OK;
}

extern "C" int vgan_sb_gam_finish(vgan_sb_gamrun *r, vgan_sb_gam_result *res, vgan_gampipe_stats *pstats) {
    if (!r) return fail(VGAN_EINVAL, "vgan_sb_gam_finish: null argument");
    if (res) memset(res, 0, sizeof *res);
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if (!r->attached) r->gave_up = true;
    }
    r->cv.notify_all();
    if (r->coord.joinable()) r->coord.join();
    for (auto &j : r->bg) {
        if (j->t.joinable()) j->t.join();
        if (j->rc < 0 && r->rc >= 0) {
            r->rc = j->rc;
            r->err = "the reads left to the host: " + j->err;
        }
    }
    if (r->rc >= 0 && !r->finished) {
        r->finished = true;
        const auto t0 = std::chrono::steady_clock::now();
        // the host's batches behind the device's, in the order of the pieces (any order gives the same tables' sums; this one is repeatable)
        std::stable_sort(r->bg.begin(), r->bg.end(), [](const std::shared_ptr<SbLeftJob> &a, const std::shared_ptr<SbLeftJob> &b) { return a->piece < b->piece; });
        for (auto &j : r->bg) {
            const size_t l = (size_t)j->lane;
            vgan_sb_batch hb;
            int rc = vgan_sb_host_batch_get(j->hb, &hb);
            if (rc >= 0 && hb.n_reads && !r->df[l]) rc = vgan_sb_devflat_create(r->ctx[l], r->graph, &r->df[l]);
            if (rc >= 0) rc = vgan_sb_devflat_append_host(r->df[l], &hb, j->where.data());
            if (rc < 0) {
                r->rc = rc;
                r->err = last_error();
                break;
            }
            r->res.n_bad += j->st.n_bad;
            vgan_sb_host_batch_free(j->hb);
            j->hb = nullptr;
        }
        for (size_t l = 0; l < r->ctx.size() && r->rc >= 0; ++l) { // analyse_GAM's tables, once per context (an empty share gives empty tables)
            vgan_sb_batch b{};
            int64_t bad = 0;
            int rc = r->df[l] ? vgan_sb_devflat_batch(r->df[l], &b) : VGAN_OK;
            if (rc >= 0) rc = vgan_sb_precompute(r->ctx[l], &b, &bad);
            if (rc < 0) {
                r->rc = rc;
                r->err = last_error();
                break;
            }
            r->res.n_dev_bad += bad;
            r->res.n_reads += b.n_reads;
        }
        r->ms_tables = ms_since(t0);
        for (size_t l = 0; l < r->df.size(); ++l) r->pst.device_bytes += sb_devflat_device_bytes(r->df[l]);
    }
    r->pst.n_host_reads = r->n_host_reads;
    r->pst.n_device_reads = r->n_device_reads;
    r->pst.ms_wait_contexts = r->ms_wait_contexts;
    if (pstats) *pstats = r->pst;
    if (r->rc < 0) return fail(r->rc, "%s", r->err.c_str());
    r->res.n_messages = (int64_t)r->pst.n_messages;
    r->res.n_mapped = (int64_t)r->pst.n_reads;
    r->res.ms_tables = r->ms_tables;
    if (res) *res = r->res;
    return VGAN_OK;
}

// the batch of lane `lane` as the tables were made from it (device pointers; valid until vgan_sb_gam_free): read_src = the reads' places
// among the file's mapped reads
extern "C" int vgan_sb_gam_batch(const vgan_sb_gamrun *r, int lane, vgan_sb_batch *out) {
    if (!r || !out || lane < 0 || lane >= (int)r->df.size()) return fail(VGAN_EINVAL, "vgan_sb_gam_batch: bad argument");
    if (!r->finished) return fail(VGAN_ESTATE, "vgan_sb_gam_batch: the run has not finished");
    if (!r->df[(size_t)lane]) {
        memset(out, 0, sizeof *out);
        out->on_device = 1;
        return VGAN_OK;
    }
    return vgan_sb_devflat_batch(r->df[(size_t)lane], out);
}

extern "C" void vgan_sb_gam_free(vgan_sb_gamrun *r) {
    if (!r) return;
    if (r->coord.joinable()) {
        r->finished = true; // (no tables for a run that is thrown away)
        (void)vgan_sb_gam_finish(r, nullptr, nullptr);
    }
    for (auto &j : r->bg)
        if (j->t.joinable()) j->t.join();
    r->bg.clear();
    for (auto *f : r->df) vgan_sb_devflat_free(f);
    delete r;
}
