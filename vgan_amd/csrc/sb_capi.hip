// C-ABI of the soibean device path (include/vgan_gpu.h).  No CPU fallback.
#include <hip/hip_runtime.h>

#include <thread>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <sstream>
#include <vector>

#include "host/common.h"
#include "sb_device.h"
#include "vgan_gpu.h"

using namespace vgan;

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(VGAN_ENODEV, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {
template <class T> struct Buf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap && p) return VGAN_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = n + 64;
        HIPCHK(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        return VGAN_OK;
    }
    int upload(const T *src, size_t n) {
        int rc = reserve(n);
        if (rc) return rc;
        if (n) HIPCHK(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
        return VGAN_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
} // namespace

struct vgan_sb_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    SbGraphDev g{};
    SbTablesDev t{};
    uint32_t P = 0;
    Buf<uint64_t> mask;
    Buf<uint8_t> findable;
    Buf<double> sub5p, sub3p, qscore;
    Buf<double> pm;
    Buf<uint16_t> cnt;
    Buf<double> stage_pm;    // read-major rows of one chunk of reads (sb_transpose_kernel moves them into pm / cnt)
    Buf<uint16_t> stage_cnt;
    Buf<uint8_t> ok;
    Buf<unsigned long long> n_bad, guard;
    Buf<uint32_t> s32;
    Buf<uint16_t> s16;
    Buf<uint8_t> s8;
    Buf<SbSourceDev> src;
    Buf<double> hky, out, freqs;
    Buf<SbFix> partial, out_fix; // per-workgroup sums of a refresh; the folded sums themselves (vgan_sb_*_sums)
    Buf<int32_t> best, mix_paths;
    Buf<unsigned long long> sig;
    std::vector<char> h_params; // host staging of one refresh's parameters
    // the chain driver's refresh (one state per call, launch bound) is one kernel writing into pinned host memory
    char *pin = nullptr;                 // out double[16] | guard u64[16] | sums SbFix[16]
    bool time_refresh = false;           // HIP events around the engine's refresh too (vgan_sb_time_engine)
    uint64_t refresh_seq = 0;            // refreshes launched so far (the finishing kernel leaves the number in the pinned block)
    uint32_t refresh_grid = 0;           // workgroups of the fused refresh: what the device holds at once (asked once)
    Buf<unsigned long long> ticket;      // guard counts (one per state) of the fused refresh, zero between refreshes
    // the resident refresh (sb_kernels.hip: sb_refresh_resident_kernel): a stream of its own, the mailbox (pinned), the device block
    hipStream_t res_stream = nullptr;
    void *mailbox = nullptr;
    Buf<uint8_t> resident;
    Buf<SbFix> res_partial;
    uint32_t res_grid = 0;
    bool res_on = false, res_running = false; // wanted (vgan_sb_resident / VGAN_SB_RESIDENT); a launch is out
    uint64_t res_launch = 0, res_launches = 0; // the id of the last launch; how many there were (test aid)
    unsigned long long res_busy_seen = 0, res_served_seen = 0; // the mailbox's counters at the last vgan_sb_kernel_ms
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool pending[2] = {false, false};
    double ms[2] = {0, 0};
    uint64_t launches[2] = {0, 0};
};

vgan::SbCtxInfo vgan::sb_ctx_info(const vgan_sb_ctx *c) { return SbCtxInfo{c->device, c->stream}; }

static void resolve(vgan_sb_ctx *c, int i) {
    if (!c->pending[i]) return;
    float ms = 0.f;
    if (hipEventSynchronize(c->ev[2 * i + 1]) == hipSuccess && hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]) == hipSuccess) {
        c->ms[i] += ms;
        c->launches[i] += 1;
    }
    c->pending[i] = false;
}

// Every entry point that is not the engine's refresh calls this first: the resident kernel holds the tables' addresses of its launch
// and half the device; it is told to leave and waited for (it relaunches with the next refresh).
static std::atomic<vgan_sb_ctx *> g_res_owner[64]; // per device: the context whose resident kernel is out (one at a time: each takes half the device)
static int res_quiesce(vgan_sb_ctx *c) {
    if (!c->res_running) return VGAN_OK;
    sb_mailbox_stop(c->mailbox, true);
    const hipError_t e = hipStreamSynchronize(c->res_stream);
    sb_mailbox_stop(c->mailbox, false);
    c->res_running = false;
    vgan_sb_ctx *me = c;
    (void)g_res_owner[c->device & 63].compare_exchange_strong(me, nullptr);
    if (e != hipSuccess) return fail(VGAN_ENODEV, "the resident refresh kernel failed: %s", hipGetErrorString(e));
    return VGAN_OK;
}

extern "C" int vgan_sb_create(const vgan_graph_view *gv, const vgan_damage_view *dmg, const vgan_sb_params *prm, int device,
                              vgan_sb_ctx **out) {
    if (!gv || !dmg || !prm || !out) return fail(VGAN_EINVAL, "vgan_sb_create: null argument");
    if (gv->n_paths == 0 || gv->n_paths > SB_MAX_PATHS) return fail(VGAN_ERANGE, "vgan_sb_create: 1..%u paths supported, got %u", SB_MAX_PATHS, gv->n_paths);
    if (!gv->mask || gv->max_id < 0) return fail(VGAN_EINVAL, "vgan_sb_create: graph has no path membership mask");
    if (prm->penalty <= 0) return fail(VGAN_EINVAL, "vgan_sb_create: penalty must be positive");
    if (dmg->n5 == 0 || dmg->n3 == 0) return fail(VGAN_EINVAL, "vgan_sb_create: empty damage tables");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(VGAN_ENODEV, "vgan_sb_create: no HIP device is visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(VGAN_EINVAL, "vgan_sb_create: device %d out of range", device);
    HIPCHK(hipSetDevice(device));
    auto c = new vgan_sb_ctx();
    c->device = device;
    if (const char *e = getenv("VGAN_SB_RESIDENT")) c->res_on = atoi(e) != 0; // (vgan_sb_resident sets it per context; off unless asked for)
    c->P = gv->n_paths;
    const uint32_t W = (gv->n_paths + 63) / 64, rows = (uint32_t)gv->max_id + 1;
    std::vector<uint8_t> findable(c->P, 1);
    {
        std::istringstream in(gv->path_names ? gv->path_names : "");
        std::string line;
        uint32_t p = 0;
        while (std::getline(in, line) && p < c->P) findable[p++] = line.size() > 101 ? 0 : 1; // getLCAfromGAM.h:80-88
    }
    std::vector<double> qs(100);
    for (int Q = 0; Q < 100; ++Q) qs[(size_t)Q] = Q >= 2 ? pow(10, ((-1 * Q) * 0.1)) : 0.25;
    auto bail = [&](int code) {
        vgan_sb_destroy(c);
        return code;
    };
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(VGAN_ENODEV, "stream creation failed"));
    for (auto &e : c->ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return bail(fail(VGAN_ENODEV, "event creation failed"));
    c->stream = c->own_stream;
    int rc;
    if ((rc = c->mask.upload(gv->mask, (size_t)rows * W)) || (rc = c->findable.upload(findable.data(), findable.size())) ||
        (rc = c->sub5p.upload(dmg->sub5p, (size_t)dmg->n5 * 16)) || (rc = c->sub3p.upload(dmg->sub3p, (size_t)dmg->n3 * 16)) ||
        (rc = c->qscore.upload(qs.data(), 100)) || (rc = c->n_bad.reserve(1)) || (rc = c->freqs.reserve(8)))
        return bail(rc);
    c->g.mask = c->mask.p;
    c->g.findable = c->findable.p;
    c->g.sub5p = c->sub5p.p;
    c->g.sub3p = c->sub3p.p;
    c->g.n5 = dmg->n5;
    c->g.n3 = dmg->n3;
    c->g.qscore = c->qscore.p;
    c->g.rows = rows;
    c->g.mask_words = W;
    c->g.n_paths = c->P;
    c->g.penalty = prm->penalty;
    *out = c;
    return VGAN_OK;
}

extern "C" void vgan_sb_destroy(vgan_sb_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)res_quiesce(c);
    {
        vgan_sb_ctx *me = c; // (a kernel that left by itself leaves the claim standing)
        (void)g_res_owner[c->device & 63].compare_exchange_strong(me, nullptr);
    }
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->mailbox) (void)hipHostFree(c->mailbox);
    if (c->res_stream) (void)hipStreamDestroy(c->res_stream);
    c->resident.release();
    c->res_partial.release();
    c->ticket.release();
    c->mask.release();
    c->findable.release();
    c->sub5p.release();
    c->sub3p.release();
    c->qscore.release();
    c->pm.release();
    c->cnt.release();
    c->stage_pm.release();
    c->stage_cnt.release();
    c->ok.release();
    c->n_bad.release();
    c->guard.release();
    c->s32.release();
    c->s16.release();
    c->s8.release();
    c->src.release();
    c->hky.release();
    c->partial.release();
    c->out_fix.release();
    c->out.release();
    c->freqs.release();
    c->best.release();
    c->mix_paths.release();
    c->sig.release();
    for (auto e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int vgan_sb_set_stream(vgan_sb_ctx *c, void *s) {
    if (!c) return fail(VGAN_EINVAL, "vgan_sb_set_stream: null context");
    if (int rq = res_quiesce(c)) return rq;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return VGAN_OK;
}

extern "C" int vgan_sb_precompute(vgan_sb_ctx *c, const vgan_sb_batch *b, int64_t *n_bad) {
    if (!c || !b) return fail(VGAN_EINVAL, "vgan_sb_precompute: null argument");
    HIPCHK(hipSetDevice(c->device));
    if (int rq = res_quiesce(c)) return rq;
    resolve(c, 0);
    const size_t R = b->n_reads, S = b->n_segments;
    int rc;
    if ((rc = c->pm.reserve((size_t)c->P * std::max<size_t>(R, 1))) ||
        (rc = c->cnt.reserve((size_t)c->P * SB_NCNT * 64u * std::max<size_t>(sb_cnt_tiles((uint32_t)R), 1))) || (rc = c->ok.reserve(std::max<size_t>(R, 1))))
        return rc;
    c->t.pm = c->pm.p;
    c->t.cnt = c->cnt.p;
    c->t.ok = c->ok.p;
    c->t.n_reads = (uint32_t)R;
    HIPCHK(hipMemsetAsync(c->n_bad.p, 0, 8, c->stream));
    if (R == 0) {
        if (n_bad) *n_bad = 0;
        return VGAN_OK;
    }
    SbBatchDev d{};
    d.n_reads = b->n_reads;
    if (b->on_device) {
        d.read_seg_off = b->read_seg_off;
        d.read_col_off = b->read_col_off;
        d.read_qual_off = b->read_qual_off;
        d.read_gseq_len = b->read_gseq_len;
        d.read_rseq_len = b->read_rseq_len;
        d.read_rev = b->read_rev;
        d.seg_node = b->seg_node;
        d.seg_col = b->seg_col;
        d.seg_len = b->seg_len;
        d.seg_base_ix = b->seg_base_ix;
        d.graph_seq = b->graph_seq;
        d.read_seq = b->read_seq;
        d.qual = b->qual;
    } else {
        auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
        if ((rc = c->s32.reserve(3 * up(R + 1) + up(S))) || (rc = c->s16.reserve(2 * up(R) + 3 * up(S))) ||
            (rc = c->s8.reserve(up(R) + 2 * up(b->n_cols) + up(b->n_qual))))
            return rc;
#define COPY(dst, src, n)                                                                                                \
    do {                                                                                                                 \
        if ((n) > 0) HIPCHK(hipMemcpyAsync((void *)(dst), (src), (n) * sizeof(*(src)), hipMemcpyHostToDevice, c->stream)); \
    } while (0)
        uint32_t *p32 = c->s32.p;
        uint16_t *p16 = c->s16.p;
        uint8_t *p8 = c->s8.p;
        d.read_seg_off = p32;
        COPY(p32, b->read_seg_off, R + 1);
        p32 += up(R + 1);
        d.read_col_off = p32;
        COPY(p32, b->read_col_off, R + 1);
        p32 += up(R + 1);
        d.read_qual_off = p32;
        COPY(p32, b->read_qual_off, R + 1);
        p32 += up(R + 1);
        d.seg_node = p32;
        COPY(p32, b->seg_node, S);
        d.read_gseq_len = p16;
        COPY(p16, b->read_gseq_len, R);
        p16 += up(R);
        d.read_rseq_len = p16;
        COPY(p16, b->read_rseq_len, R);
        p16 += up(R);
        d.seg_col = p16;
        COPY(p16, b->seg_col, S);
        p16 += up(S);
        d.seg_len = p16;
        COPY(p16, b->seg_len, S);
        p16 += up(S);
        d.seg_base_ix = p16;
        COPY(p16, b->seg_base_ix, S);
        d.read_rev = p8;
        COPY(p8, b->read_rev, R);
        p8 += up(R);
        d.graph_seq = p8;
        COPY(p8, b->graph_seq, (size_t)b->n_cols);
        p8 += up(b->n_cols);
        d.read_seq = p8;
        COPY(p8, b->read_seq, (size_t)b->n_cols);
        p8 += up(b->n_cols);
        d.qual = p8;
        COPY(p8, b->qual, (size_t)b->n_qual);
#undef COPY
    }
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    {
        const uint32_t chunk = (uint32_t)std::min<size_t>(std::max<size_t>(R, 1), 65536);
        if ((rc = c->stage_pm.reserve((size_t)chunk * c->P)) || (rc = c->stage_cnt.reserve((size_t)chunk * c->P * SB_NCNT))) return rc;
        launch_sb_precompute(c->g, d, c->t, c->stage_pm.p, c->stage_cnt.p, chunk, c->n_bad.p, c->stream);
    }
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    c->pending[0] = true;
    HIPCHK(hipGetLastError());
    unsigned long long nb = 0;
    HIPCHK(hipMemcpyAsync(&nb, c->n_bad.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n_bad) *n_bad = (int64_t)nb;
    return VGAN_OK;
}

extern "C" int vgan_sb_read_tables(vgan_sb_ctx *c, uint32_t r0, uint32_t r1, double *pm, uint16_t *cnt, uint8_t *ok) {
    if (!c) return fail(VGAN_EINVAL, "vgan_sb_read_tables: null context");
    if (r0 > r1 || r1 > c->t.n_reads) return fail(VGAN_EINVAL, "vgan_sb_read_tables: bad range");
    HIPCHK(hipSetDevice(c->device));
    if (int rq = res_quiesce(c)) return rq;
    const size_t n = r1 - r0, R = c->t.n_reads;
    if (n == 0) return VGAN_OK;
    // (the counts lie tiled by 64 reads on the device: the tiles covering [r0, r1) come over per path and are laid out
    // [path][pair][read] for the caller here)
    const uint32_t n_tiles = sb_cnt_tiles((uint32_t)R), t0 = r0 / 64u, t1 = (r1 + 63u) / 64u;
    std::vector<uint16_t> tiles;
    if (cnt) tiles.resize((size_t)c->P * (t1 - t0) * SB_NCNT * 64u);
    for (uint32_t p = 0; p < c->P; ++p) {
        if (pm) HIPCHK(hipMemcpyAsync(pm + (size_t)p * n, c->t.pm + (size_t)p * R + r0, n * 8, hipMemcpyDeviceToHost, c->stream));
        if (cnt)
            HIPCHK(hipMemcpyAsync(tiles.data() + (size_t)p * (t1 - t0) * SB_NCNT * 64u, c->t.cnt + sb_cnt_index(p, t0 * 64u, n_tiles),
                                  (size_t)(t1 - t0) * SB_NCNT * 64u * 2u, hipMemcpyDeviceToHost, c->stream));
    }
    if (ok) HIPCHK(hipMemcpyAsync(ok, c->t.ok + r0, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (cnt)
        for (uint32_t p = 0; p < c->P; ++p)
            for (uint32_t j = 0; j < SB_NCNT; ++j)
                for (uint32_t r = r0; r < r1; ++r)
                    cnt[((size_t)p * SB_NCNT + j) * n + (r - r0)] =
                        tiles[(((size_t)p * (t1 - t0) + (r / 64u - t0)) * SB_NCNT + j) * 64u + r % 64u];
    return VGAN_OK;
}

extern "C" int vgan_sb_best_paths(vgan_sb_ctx *c, int32_t *best, int64_t *sig_count, int64_t *n_reads_ok) {
    if (!c) return fail(VGAN_EINVAL, "vgan_sb_best_paths: null context");
    HIPCHK(hipSetDevice(c->device));
    if (int rq = res_quiesce(c)) return rq;
    const uint32_t R = c->t.n_reads;
    int rc;
    if ((rc = c->sig.reserve(c->P + 1)) || (best && (rc = c->best.reserve(R)))) return rc;
    HIPCHK(hipMemsetAsync(c->sig.p, 0, (size_t)(c->P + 1) * 8, c->stream));
    launch_sb_best_paths(c->t, c->P, best ? c->best.p : nullptr, c->sig.p, c->sig.p + c->P, c->stream);
    HIPCHK(hipGetLastError());
    std::vector<unsigned long long> h(c->P + 1);
    HIPCHK(hipMemcpyAsync(h.data(), c->sig.p, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
    if (best && R) HIPCHK(hipMemcpyAsync(best, c->best.p, (size_t)R * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (sig_count)
        for (uint32_t p = 0; p < c->P; ++p) sig_count[p] = (int64_t)h[p];
    if (n_reads_ok) *n_reads_ok = (int64_t)h[c->P];
    return VGAN_OK;
}

static_assert(sizeof(vgan_sb_sum) == sizeof(SbFix) && offsetof(vgan_sb_sum, lo) == offsetof(SbFix, lo) && offsetof(vgan_sb_sum, nf) == offsetof(SbFix, nf),
              "vgan_sb_sum is the device's SbFix");

extern "C" double vgan_sb_sum_value(const vgan_sb_sum *s) {
    if (!s) return 0.0;
    return sb_fix_value(SbFix{(long long)s->hi, (unsigned long long)s->lo, s->nf});
}
extern "C" void vgan_sb_sum_add(vgan_sb_sum *acc, const vgan_sb_sum *x) {
    if (!acc || !x) return;
    acc->hi += x->hi;
    acc->lo += x->lo;
    acc->nf += x->nf;
}

static int mixture_impl(vgan_sb_ctx *c, uint32_t n, const int32_t *paths, double log_freq, double *out, vgan_sb_sum *sum);
extern "C" int vgan_sb_mixture_loglike(vgan_sb_ctx *c, uint32_t n, const int32_t *paths, double log_freq, double *out) {
    if (!out) return fail(VGAN_EINVAL, "vgan_sb_mixture_loglike: null argument");
    return mixture_impl(c, n, paths, log_freq, out, nullptr);
}
extern "C" int vgan_sb_mixture_sums(vgan_sb_ctx *c, uint32_t n, const int32_t *paths, double log_freq, vgan_sb_sum *sum) {
    if (!sum) return fail(VGAN_EINVAL, "vgan_sb_mixture_sums: null argument");
    return mixture_impl(c, n, paths, log_freq, nullptr, sum);
}
static int mixture_impl(vgan_sb_ctx *c, uint32_t n, const int32_t *paths, double log_freq, double *out, vgan_sb_sum *sum) {
    if (!c || !paths) return fail(VGAN_EINVAL, "vgan_sb_mixture_loglike: null argument");
    if (n == 0 || n > SB_MAX_PATHS) return fail(VGAN_EINVAL, "vgan_sb_mixture_loglike: 1..%u sources, got %u", SB_MAX_PATHS, n);
    for (uint32_t i = 0; i < n; ++i)
        if (paths[i] < 0 || (uint32_t)paths[i] >= c->P) return fail(VGAN_EINVAL, "vgan_sb_mixture_loglike: path index out of range");
    HIPCHK(hipSetDevice(c->device));
    if (int rq = res_quiesce(c)) return rq;
    const uint32_t R = c->t.n_reads;
    const uint32_t n_blocks = std::max(1u, std::min(1024u, (R + 255) / 256));
    int rc;
    if ((rc = c->mix_paths.reserve(n)) || (rc = c->partial.reserve(n_blocks)) || (rc = c->out.reserve(1)) || (rc = c->out_fix.reserve(1))) return rc;
    HIPCHK(hipMemcpyAsync(c->mix_paths.p, paths, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    launch_sb_mixture(c->t, n, c->mix_paths.p, log_freq, c->partial.p, n_blocks, c->out.p, c->out_fix.p, c->stream);
    HIPCHK(hipGetLastError());
    if (out) HIPCHK(hipMemcpyAsync(out, c->out.p, 8, hipMemcpyDeviceToHost, c->stream));
    if (sum) HIPCHK(hipMemcpyAsync(sum, c->out_fix.p, sizeof(SbFix), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VGAN_OK;
}

static int loglike_impl(vgan_sb_ctx *c, uint32_t n_states, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7,
                        double *out, double *d_out, uint64_t *guard, vgan_sb_sum *sums);
extern "C" int vgan_sb_loglike(vgan_sb_ctx *c, uint32_t n_states, uint32_t k, const vgan_sb_source *src, double con,
                               const double *freqs7, double *out, double *d_out, uint64_t *guard) {
    return loglike_impl(c, n_states, k, src, con, freqs7, out, d_out, guard, nullptr);
}
extern "C" int vgan_sb_loglike_sums(vgan_sb_ctx *c, uint32_t n_states, uint32_t k, const vgan_sb_source *src, double con,
                                    const double *freqs7, vgan_sb_sum *sums, uint64_t *guard) {
    if (!sums) return fail(VGAN_EINVAL, "vgan_sb_loglike_sums: null argument");
    return loglike_impl(c, n_states, k, src, con, freqs7, nullptr, nullptr, guard, sums);
}
static int loglike_impl(vgan_sb_ctx *c, uint32_t n_states, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7,
                        double *out, double *d_out, uint64_t *guard, vgan_sb_sum *sums) {
    if (!c || !src || !freqs7) return fail(VGAN_EINVAL, "vgan_sb_loglike: null argument");
    if (n_states == 0 || k == 0) return fail(VGAN_EINVAL, "vgan_sb_loglike: need at least one state and one source");
    if ((size_t)n_states * k * 2 * SB_NCNT * 8 > 60000) return fail(VGAN_ERANGE, "vgan_sb_loglike: n_states*k too large for one launch (<= 150)");
    HIPCHK(hipSetDevice(c->device));
    if (int rq = res_quiesce(c)) return rq;
    resolve(c, 1);
    const uint32_t ne = n_states * k;
    std::vector<SbSourceDev> sd(ne);
    for (uint32_t i = 0; i < ne; ++i) {
        if (src[i].child < 0 || src[i].parent < 0 || (uint32_t)src[i].child >= c->P || (uint32_t)src[i].parent >= c->P)
            return fail(VGAN_EINVAL, "vgan_sb_loglike: path index out of range");
        double t = src[i].dist;
        if (t == 0.0) t = 0.00001; // MCMC.cpp:753-755,893-895
        sd[i].child = src[i].child;
        sd[i].parent = src[i].parent;
        sd[i].t1 = src[i].pos * t;
        sd[i].t2 = t - sd[i].t1;
        sd[i].pos = src[i].pos;
        sd[i].log_pos = log(src[i].pos);
        sd[i].log_1mpos = log((1 - src[i].pos));
        sd[i].log_theta = log(src[i].theta);
    }
    const uint32_t R = c->t.n_reads;
    const uint32_t n_blocks = std::max(1u, std::min(1024u, (R + 255) / 256));
    int rc;
    if ((rc = c->src.reserve(ne + 2)) || (rc = c->hky.reserve((size_t)ne * 2 * SB_NCNT)) || (rc = c->partial.reserve((size_t)n_states * n_blocks)) ||
        (rc = c->out.reserve(n_states)) || (rc = c->out_fix.reserve(n_states)) || (rc = c->guard.reserve(n_states)))
        return rc;
    // the sources and the seven frequencies travel in one copy: [SbSourceDev x ne][double x 7]
    static_assert(sizeof(SbSourceDev) % 8 == 0, "freqs follow the sources at an 8-byte offset");
    c->h_params.resize(ne * sizeof(SbSourceDev) + 7 * 8);
    memcpy(c->h_params.data(), sd.data(), ne * sizeof(SbSourceDev));
    memcpy(c->h_params.data() + ne * sizeof(SbSourceDev), freqs7, 7 * 8);
    HIPCHK(hipMemcpyAsync(c->src.p, c->h_params.data(), c->h_params.size(), hipMemcpyHostToDevice, c->stream));
    const double *d_freqs = reinterpret_cast<const double *>(reinterpret_cast<const char *>(c->src.p) + ne * sizeof(SbSourceDev));
    launch_sb_hky(ne, c->src.p, con, d_freqs, c->hky.p, c->guard.p, n_states, c->stream);
    HIPCHK(hipEventRecord(c->ev[2], c->stream));
    launch_sb_loglike(c->t, c->P, n_states, k, c->src.p, c->hky.p, c->partial.p, n_blocks, c->out.p, d_out, c->out_fix.p, c->guard.p, c->stream);
    HIPCHK(hipEventRecord(c->ev[3], c->stream));
    c->pending[1] = true;
    HIPCHK(hipGetLastError());
    if (out || guard || sums) {
        std::vector<unsigned long long> gd(n_states);
        if (out) HIPCHK(hipMemcpyAsync(out, c->out.p, (size_t)n_states * 8, hipMemcpyDeviceToHost, c->stream));
        if (sums) HIPCHK(hipMemcpyAsync(sums, c->out_fix.p, (size_t)n_states * sizeof(SbFix), hipMemcpyDeviceToHost, c->stream));
        if (guard) HIPCHK(hipMemcpyAsync(gd.data(), c->guard.p, (size_t)n_states * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (guard)
            for (uint32_t i = 0; i < n_states; ++i) guard[i] = gd[i];
    }
    return VGAN_OK;
}

// n_states states of k sources each, results to the host: the per-iteration call of the chain driver (the chains of one source
// count advance together) -- sb_refresh_fused_kernel with the sources as kernel arguments, its last workgroup folding the partials into
// pinned host memory; bit-identical to vgan_sb_loglike.
constexpr size_t SB_PIN_SEQ_OFF = 2 * SB_FUSED_MAX_K * 8 + SB_FUSED_MAX_K * sizeof(SbFix); // u64[16]: the refresh that wrote the entry
constexpr size_t SB_PIN_BYTES = SB_PIN_SEQ_OFF + SB_FUSED_MAX_K * 8;

// launch half: everything is queued on the context's stream, nothing is waited for.  *general: the states did not fit the
// kernel-argument staging and went through vgan_sb_loglike (already complete: results in gen_*).
struct SbPending {
    bool general = false, resident = false;
    uint64_t seq = 0; // the refresh's number: what the finishing kernel leaves in the pinned block behind its results
    std::vector<double> gen_out;
    std::vector<vgan_sb_sum> gen_sum;
    std::vector<uint64_t> gen_guard;
};
// Launches the resident kernel: `done` = the number of the last refresh served (the kernel waits for another one in the mailbox).  A device
// that cannot hold the grid as a whole leaves res_running false: the caller launches per refresh.
constexpr unsigned long long SB_RES_IDLE_TICKS = 500000; // 5 ms of the 100 MHz clock without a refresh: the kernel leaves
static int res_start(vgan_sb_ctx *c, uint64_t done, bool posted /* the mailbox holds refresh done + 1: it is left alone */) {
    int rc;
    if (!c->mailbox) {
        HIPCHK(hipHostMalloc(&c->mailbox, sb_mailbox_bytes(), hipHostMallocDefault));
        memset(c->mailbox, 0, sb_mailbox_bytes());
        HIPCHK(hipStreamCreateWithFlags(&c->res_stream, hipStreamNonBlocking));
        if ((rc = c->resident.reserve(sb_resident_bytes()))) return rc;
        HIPCHK(hipMemset(c->resident.p, 0, sb_resident_bytes()));
        c->res_grid = sb_resident_grid(c->device, 1024u);
        if (const char *e = getenv("VGAN_SB_RESIDENT_GRID")) c->res_grid = std::min<uint32_t>(std::max(1, atoi(e)), 2u * c->res_grid); // (developer aid; at most all that fits)
    }
    if (c->res_grid == 0) return VGAN_OK;
    {
        vgan_sb_ctx *none = nullptr;
        auto &owner = g_res_owner[c->device & 63];
        if (owner.load() != c && !owner.compare_exchange_strong(none, c)) return VGAN_OK; // (another context's kernel holds this device: launches here)
    }
    if (!posted) sb_mailbox_idle(c->mailbox, done);
    const uint32_t R = c->t.n_reads;
    const uint32_t n_blocks = std::max(1u, std::min(c->res_grid, (R + 255) / 256));
    if ((rc = c->res_partial.reserve((size_t)SB_FUSED_MAX_K * n_blocks))) return rc;
    if (!c->pin) {
        HIPCHK(hipHostMalloc((void **)&c->pin, SB_PIN_BYTES, hipHostMallocDefault));
        memset(c->pin, 0, SB_PIN_BYTES);
    }
    if (!c->ticket.p) {
        if ((rc = c->ticket.reserve(SB_FUSED_MAX_K + 1))) return rc; // (the guard words, then the last-to-finish count of the launched refresh)
        HIPCHK(hipMemsetAsync(c->ticket.p, 0, (SB_FUSED_MAX_K + 1) * 8, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream)); // (the tables are final; the guard words are zero)
    double *pin_out = reinterpret_cast<double *>(c->pin);
    unsigned long long *pin_guard = reinterpret_cast<unsigned long long *>(c->pin + SB_FUSED_MAX_K * 8);
    SbFix *pin_fix = reinterpret_cast<SbFix *>(c->pin + 2 * SB_FUSED_MAX_K * 8);
    c->res_launch += 1;
    c->res_launches += 1;
    launch_sb_refresh_resident(c->t, c->mailbox, c->resident.p, c->res_partial.p, n_blocks, c->ticket.p, pin_out, pin_guard, pin_fix,
                               reinterpret_cast<unsigned long long *>(c->pin + SB_PIN_SEQ_OFF), done, SB_RES_IDLE_TICKS, c->res_launch, c->res_stream);
    HIPCHK(hipGetLastError());
    c->res_running = true;
    return VGAN_OK;
}
// before a refresh is posted: a kernel is out, or one is started
static int res_ensure(vgan_sb_ctx *c) {
    if (c->res_running) {
        if (sb_mailbox_exited(c->mailbox) != c->res_launch) return VGAN_OK;
        HIPCHK(hipStreamSynchronize(c->res_stream)); // it left by itself
        c->res_running = false;
    }
    return res_start(c, c->refresh_seq, false);
}

static int refresh_launch(vgan_sb_ctx *c, uint32_t n_states, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7,
                          SbPending &pd) {
    if (!c || !src || !freqs7) return fail(VGAN_EINVAL, "vgan_sb_engine refresh: null argument");
    if (n_states == 0 || k == 0) return fail(VGAN_EINVAL, "vgan_sb_engine refresh: need at least one state and one source");
    const uint32_t ne = n_states * k;
    pd.general = ne > SB_FUSED_MAX_K;
    if (pd.general) { // beyond the kernel-argument staging: the general path (a launch per call, copies)
        pd.gen_out.assign(n_states, 0.0);
        pd.gen_sum.assign(n_states, vgan_sb_sum{0, 0, 0.0});
        pd.gen_guard.assign(n_states, 0);
        int rc = loglike_impl(c, n_states, k, src, con, freqs7, pd.gen_out.data(), nullptr, pd.gen_guard.data(), pd.gen_sum.data());
        return rc;
    }
    HIPCHK(hipSetDevice(c->device));
    SbFusedArgs a;
    for (uint32_t i = 0; i < ne; ++i) {
        if (src[i].child < 0 || src[i].parent < 0 || (uint32_t)src[i].child >= c->P || (uint32_t)src[i].parent >= c->P)
            return fail(VGAN_EINVAL, "vgan_sb_loglike: path index out of range");
        double t = src[i].dist;
        if (t == 0.0) t = 0.00001; // MCMC.cpp:753-755,893-895
        SbSourceDev &d = a.src[i];
        d.child = src[i].child;
        d.parent = src[i].parent;
        d.t1 = src[i].pos * t;
        d.t2 = t - d.t1;
        d.pos = src[i].pos;
        d.log_pos = log(src[i].pos);
        d.log_1mpos = log((1 - src[i].pos));
        d.log_theta = log(src[i].theta);
    }
    for (uint32_t i = ne; i < SB_FUSED_MAX_K; ++i) a.src[i] = a.src[0];
    memcpy(a.freqs7, freqs7, 7 * 8);
    a.con = con;
    const uint32_t R = c->t.n_reads;
    if (c->refresh_grid == 0) {
        const char *e = getenv("VGAN_SB_REFRESH_BLOCKS"); // (developer aid)
        c->refresh_grid = e && atoi(e) > 0 ? (uint32_t)atoi(e) : sb_refresh_grid(c->device);
    }
    const uint32_t n_blocks = std::max(1u, std::min(c->refresh_grid, (R + 255) / 256));
    int rc;
    if ((rc = c->partial.reserve((size_t)n_states * n_blocks))) return rc;
    if (!c->pin) {
        HIPCHK(hipHostMalloc((void **)&c->pin, SB_PIN_BYTES, hipHostMallocDefault));
        memset(c->pin, 0, SB_PIN_BYTES);
    }
    if (!c->ticket.p) {
        if ((rc = c->ticket.reserve(SB_FUSED_MAX_K + 1))) return rc; // (the guard words, then the last-to-finish count of the launched refresh)
        HIPCHK(hipMemsetAsync(c->ticket.p, 0, (SB_FUSED_MAX_K + 1) * 8, c->stream));
    }
    double *pin_out = reinterpret_cast<double *>(c->pin);
    unsigned long long *pin_guard = reinterpret_cast<unsigned long long *>(c->pin + SB_FUSED_MAX_K * 8);
    SbFix *pin_fix = reinterpret_cast<SbFix *>(c->pin + 2 * SB_FUSED_MAX_K * 8);
    if (c->res_on && !c->time_refresh && R > 0) {
        if ((rc = res_ensure(c))) return rc;
        if (c->res_running) {
            pd.seq = ++c->refresh_seq;
            pd.resident = true;
            sb_mailbox_post(c->mailbox, n_states, k, a, pd.seq);
            return VGAN_OK;
        }
    }
    if (c->time_refresh) {
        resolve(c, 1);
        HIPCHK(hipEventRecord(c->ev[2], c->stream));
    }
    pd.seq = ++c->refresh_seq;
    launch_sb_refresh_fused(c->t, n_states, k, a, c->partial.p, n_blocks, c->ticket.p, reinterpret_cast<unsigned int *>(c->ticket.p + SB_FUSED_MAX_K), pin_out,
                            pin_guard, pin_fix, reinterpret_cast<unsigned long long *>(c->pin + SB_PIN_SEQ_OFF), pd.seq, c->stream);
    if (c->time_refresh) {
        HIPCHK(hipEventRecord(c->ev[3], c->stream));
        c->pending[1] = true;
    }
    HIPCHK(hipGetLastError());
    return VGAN_OK;
}
// collect half: waits for the context's stream; out / sums / guard (each may be NULL) receive n_states entries
static int refresh_collect(vgan_sb_ctx *c, uint32_t n_states, const SbPending &pd, double *out, vgan_sb_sum *sums, uint64_t *guard) {
    if (pd.general) {
        for (uint32_t e = 0; e < n_states; ++e) {
            if (out) out[e] = pd.gen_out[e];
            if (sums) sums[e] = pd.gen_sum[e];
            if (guard) guard[e] = pd.gen_guard[e];
        }
        return VGAN_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    // The finishing kernel writes the refresh's number behind each state's results (system-scope fence in between): the host
    // watches those words instead of sleeping in hipStreamSynchronize -- an MCMC iteration is one such wait, and the wake-up
    // was a tenth of it.  The stream is asked now and then so that a failed launch ends the wait.
    if (pd.resident) {
        // the resident kernel writes the same words; what can go wrong is that it left (5 ms without a refresh) just before this one was
        // posted: it is started again on the mailbox as it stands
        const volatile uint64_t *seq = reinterpret_cast<const volatile uint64_t *>(c->pin + SB_PIN_SEQ_OFF);
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0;; ++spins) {
            bool all = true;
            for (uint32_t e = 0; e < n_states; ++e) all = all && seq[e] == pd.seq;
            if (all) break;
            if ((spins & 0x3FFu) == 0x3FFu) {
                if (sb_mailbox_exited(c->mailbox) == c->res_launch) {
                    if (hipStreamSynchronize(c->res_stream) != hipSuccess) return fail(VGAN_ENODEV, "vgan_sb_engine refresh: the resident kernel failed");
                    c->res_running = false;
                    all = true;
                    for (uint32_t e = 0; e < n_states; ++e) all = all && seq[e] == pd.seq;
                    if (all) break; // (served, then left)
                    int rc = res_start(c, pd.seq - 1, true);
                    if (rc) return rc;
                    if (!c->res_running) return fail(VGAN_ENODEV, "vgan_sb_engine refresh: the resident kernel cannot be started again");
                } else if ((spins & 0xFFFFFu) == 0xFFFFFu) {
                    const hipError_t q = hipStreamQuery(c->res_stream);
                    if (q != hipSuccess && q != hipErrorNotReady) return fail(VGAN_ENODEV, "vgan_sb_engine refresh: the resident kernel failed: %s", hipGetErrorString(q));
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) {
                        sb_mailbox_stop(c->mailbox, true);
                        return fail(VGAN_ENODEV, "vgan_sb_engine refresh: no answer from the resident kernel within 60 s");
                    }
                }
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    } else {
        const volatile uint64_t *seq = reinterpret_cast<const volatile uint64_t *>(c->pin + SB_PIN_SEQ_OFF);
        for (uint32_t spins = 0;; ++spins) {
            bool all = true;
            for (uint32_t e = 0; e < n_states; ++e) all = all && seq[e] == pd.seq;
            if (all) break;
            if ((spins & 0xFFFu) == 0xFFFu) {
                const hipError_t q = hipStreamQuery(c->stream);
                if (q == hipSuccess) break; // (everything queued has run: the words are there)
                if (q != hipErrorNotReady) return fail(VGAN_ENODEV, "vgan_sb_engine refresh: the stream failed");
            }
            if (spins >= 0x20000u) { // (~50 us of spinning did not see the words: a long refresh -- sleep on the stream instead of a core)
                if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(VGAN_ENODEV, "vgan_sb_engine refresh: the stream failed");
                break;
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    const double *pin_out = reinterpret_cast<const double *>(c->pin);
    const unsigned long long *pin_guard = reinterpret_cast<const unsigned long long *>(c->pin + SB_FUSED_MAX_K * 8);
    const SbFix *pin_fix = reinterpret_cast<const SbFix *>(c->pin + 2 * SB_FUSED_MAX_K * 8);
    for (uint32_t e = 0; e < n_states; ++e) {
        if (out) out[e] = pin_out[e];
        if (sums) sums[e] = vgan_sb_sum{(int64_t)pin_fix[e].hi, (uint64_t)pin_fix[e].lo, pin_fix[e].nf};
        if (guard) guard[e] = pin_guard[e];
    }
    return VGAN_OK;
}
static int refresh_states(vgan_sb_ctx *c, uint32_t n_states, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7,
                          double *out, uint64_t *guard) {
    if (!out) return fail(VGAN_EINVAL, "vgan_sb_engine refresh: null argument");
    SbPending pd;
    int rc = refresh_launch(c, n_states, k, src, con, freqs7, pd);
    if (rc) return rc;
    return refresh_collect(c, n_states, pd, out, nullptr, guard);
}

// the chain driver's view of this context (host/sb_chain.cpp, vgan_sb_estimate)
static int engine_refresh(void *user, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7, double *out, uint64_t *guard) {
    return refresh_states((vgan_sb_ctx *)user, 1, k, src, con, freqs7, out, guard);
}
static int engine_refresh_many(void *user, uint32_t n_states, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7,
                               double *out, uint64_t *guard) {
    return refresh_states((vgan_sb_ctx *)user, n_states, k, src, con, freqs7, out, guard);
}
static int engine_mixture(void *user, uint32_t n, const int32_t *paths, double log_freq, double *out) {
    return vgan_sb_mixture_loglike((vgan_sb_ctx *)user, n, paths, log_freq, out);
}

extern "C" int vgan_sb_time_engine(vgan_sb_ctx *c, int on) {
    if (!c) return fail(VGAN_EINVAL, "vgan_sb_time_engine: null context");
    if (int rq = res_quiesce(c)) return rq; // (HIP events bracket launches: a timed refresh is a launched one)
    c->time_refresh = on != 0;
    return VGAN_OK;
}

// The engine's refresh served by a kernel that stays on the device between refreshes (include/vgan_gpu.h); -1 asks, 0 / 1 sets.
extern "C" int vgan_sb_resident(vgan_sb_ctx *c, int on) {
    if (!c) return fail(VGAN_EINVAL, "vgan_sb_resident: null context");
    if (on < 0) return c->res_on ? 1 : 0;
    HIPCHK(hipSetDevice(c->device));
    if (int rq = res_quiesce(c)) return rq;
    c->res_on = on != 0;
    return VGAN_OK;
}
extern "C" int vgan_sb_resident_launches(const vgan_sb_ctx *c, uint64_t *launches) {
    if (!c || !launches) return fail(VGAN_EINVAL, "vgan_sb_resident_launches: null argument");
    *launches = c->res_launches;
    return VGAN_OK;
}

extern "C" int vgan_sb_engine_gpu(vgan_sb_ctx *c, vgan_sb_engine *out) {
    if (!c || !out) return fail(VGAN_EINVAL, "vgan_sb_engine_gpu: null argument");
    out->user = c;
    out->refresh = engine_refresh;
    out->mixture = engine_mixture;
    out->refresh_many = engine_refresh_many;
    return VGAN_OK;
}

// ---------------------------------------------------------------------------------------------- several contexts, one engine
// The reads of one job dealt to several contexts (one per GPU: MCMC.cpp:739 is `#pragma omp parallel for ... reduction(+:
// logLike)` over the reads, here over devices): a refresh is launched on every context, then the per-context sums -- integers,
// see SbFix -- are added on the host and turned into the log-likelihood once.  Same bits as one context holding all the reads.
struct vgan_sb_group {
    std::vector<vgan_sb_ctx *> ctxs;
};

extern "C" int vgan_sb_group_create(vgan_sb_ctx **ctxs, int n, vgan_sb_group **out) {
    if (!ctxs || n <= 0 || !out) return fail(VGAN_EINVAL, "vgan_sb_group_create: null argument");
    for (int i = 0; i < n; ++i)
        if (!ctxs[i] || ctxs[i]->P != ctxs[0]->P) return fail(VGAN_EINVAL, "vgan_sb_group_create: contexts of different graphs");
    auto g = new vgan_sb_group();
    g->ctxs.assign(ctxs, ctxs + n);
    *out = g;
    return VGAN_OK;
}
extern "C" void vgan_sb_group_free(vgan_sb_group *g) { delete g; }

static int group_refresh_many(void *user, uint32_t n_states, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7,
                              double *out, uint64_t *guard) {
    auto g = (vgan_sb_group *)user;
    if (!g || !out) return fail(VGAN_EINVAL, "vgan_sb_group refresh: null argument");
    const size_t nc = g->ctxs.size();
    std::vector<SbPending> pd(nc);
    int rc;
    for (size_t i = 0; i < nc; ++i) // every device starts before any is waited for
        if (g->ctxs[i]->t.n_reads && (rc = refresh_launch(g->ctxs[i], n_states, k, src, con, freqs7, pd[i]))) return rc;
    std::vector<vgan_sb_sum> tot(n_states, vgan_sb_sum{0, 0, 0.0}), part(n_states);
    std::vector<uint64_t> gd(n_states), gtot(n_states, 0);
    for (size_t i = 0; i < nc; ++i) {
        if (!g->ctxs[i]->t.n_reads) continue;
        if ((rc = refresh_collect(g->ctxs[i], n_states, pd[i], nullptr, part.data(), gd.data()))) return rc;
        for (uint32_t e = 0; e < n_states; ++e) {
            vgan_sb_sum_add(&tot[e], &part[e]);
            gtot[e] += gd[e];
        }
    }
    for (uint32_t e = 0; e < n_states; ++e) {
        out[e] = vgan_sb_sum_value(&tot[e]);
        if (guard) guard[e] = gtot[e];
    }
    return VGAN_OK;
}
static int group_refresh(void *user, uint32_t k, const vgan_sb_source *src, double con, const double *freqs7, double *out, uint64_t *guard) {
    return group_refresh_many(user, 1, k, src, con, freqs7, out, guard);
}
static int group_mixture(void *user, uint32_t n, const int32_t *paths, double log_freq, double *out) {
    auto g = (vgan_sb_group *)user;
    if (!g || !out) return fail(VGAN_EINVAL, "vgan_sb_group mixture: null argument");
    vgan_sb_sum tot{0, 0, 0.0}, part;
    int rc;
    for (auto c : g->ctxs) {
        if (!c->t.n_reads) continue;
        if ((rc = vgan_sb_mixture_sums(c, n, paths, log_freq, &part))) return rc;
        vgan_sb_sum_add(&tot, &part);
    }
    *out = vgan_sb_sum_value(&tot);
    return VGAN_OK;
}
extern "C" int vgan_sb_engine_group(vgan_sb_group *g, vgan_sb_engine *out) {
    if (!g || !out) return fail(VGAN_EINVAL, "vgan_sb_engine_group: null argument");
    out->user = g;
    out->refresh = group_refresh;
    out->mixture = group_mixture;
    out->refresh_many = group_refresh_many;
    return VGAN_OK;
}
// the per-path signature counts and the number of usable reads over all the group's contexts (vgan_sb_best_paths, summed)
extern "C" int vgan_sb_group_best_paths(vgan_sb_group *g, int64_t *sig_count, int64_t *n_reads_ok) {
    if (!g || !sig_count || !n_reads_ok) return fail(VGAN_EINVAL, "vgan_sb_group_best_paths: null argument");
    const uint32_t P = g->ctxs[0]->P;
    std::vector<int64_t> part(P);
    for (uint32_t p = 0; p < P; ++p) sig_count[p] = 0;
    *n_reads_ok = 0;
    int rc;
    for (auto c : g->ctxs) {
        if (!c->t.n_reads) continue;
        int64_t ok = 0;
        if ((rc = vgan_sb_best_paths(c, nullptr, part.data(), &ok))) return rc;
        for (uint32_t p = 0; p < P; ++p) sig_count[p] += part[p];
        *n_reads_ok += ok;
    }
    return VGAN_OK;
}

extern "C" int vgan_sb_kernel_ms(vgan_sb_ctx *c, double ms[2], uint64_t launches[2]) {
    if (!c) return fail(VGAN_EINVAL, "vgan_sb_kernel_ms: null context");
    HIPCHK(hipSetDevice(c->device));
    resolve(c, 0);
    resolve(c, 1);
    if (c->mailbox) { // the resident kernel's own account (it writes it when it leaves): 100 MHz ticks between seeing a refresh's number and publishing its results
        if (int rq = res_quiesce(c)) return rq;
        unsigned long long ticks = 0, served = 0, stamp[8];
        sb_mailbox_busy(c->mailbox, &ticks, &served, stamp);
        if (getenv("VGAN_TIMING") && served != c->res_served_seen)
            fprintf(stderr, "[vgan timing] soibean resident refresh, the last one (us since its number was seen): arguments over %.2f, tables built %.2f, reads done %.2f, "
                            "last workgroup in %.2f, folded %.2f, published %.2f\n",
                    stamp[0] * 1e-2, stamp[1] * 1e-2, stamp[2] * 1e-2, stamp[3] * 1e-2, stamp[4] * 1e-2, stamp[5] * 1e-2);
        c->ms[1] += (double)(ticks - c->res_busy_seen) * 1e-5;
        c->launches[1] += served - c->res_served_seen;
        c->res_busy_seen = ticks;
        c->res_served_seen = served;
    }
    for (int i = 0; i < 2; ++i) {
        if (ms) ms[i] = c->ms[i];
        if (launches) launches[i] = c->launches[i];
        c->ms[i] = 0;
        c->launches[i] = 0;
    }
    return VGAN_OK;
}
