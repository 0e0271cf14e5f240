// C-ABI of the euka device path (include/vgan_gpu.h): context, uploads, launches.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "euka_device.h"
#include "host/common.h"
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
        const size_t want = n + n / 8 + 64;
        HIPCHK(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        return VGAN_OK;
    }
    int upload(const std::vector<T> &v) {
        int rc = reserve(v.size());
        if (rc) return rc;
        if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return VGAN_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
} // namespace

struct vgan_euka_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    EukaDev d{};
    uint32_t n_clades = 0, n_bins = 0;
    int32_t ltp = 5;
    Buf<uint32_t> bp, bin_off;
    Buf<int32_t> bp_clade, bin_lo, bin_hi, node_clade;
    Buf<double> clade_dist, sub5p, sub3p, tables, dmg_pair;
    // accumulators: slices of one allocation (one launch clears them all: vgan_euka_reset)
    template <class T> struct Slice {
        T *p = nullptr;
    };
    Buf<uint8_t> acc_all;
    size_t acc_bytes = 0;
    Slice<int32_t> clade_count;
    Slice<uint32_t> baseshift;
    Slice<double> bin_cov;
    Slice<uint32_t> like_n;
    Slice<double> like_logsum;
    Slice<unsigned long long> n_bad;
    // staging (host batches) and per-read outputs
    Buf<uint32_t> s32;
    Buf<uint16_t> s16;
    Buf<uint8_t> s8;
    Buf<int32_t> smq, o_clade;
    Buf<double> o_d;
    Buf<uint8_t> o_pass;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double ms = 0;
    uint64_t launches = 0;
    bool ev_pending = false;
};

vgan::EukaCtxInfo vgan::euka_ctx_info(const vgan_euka_ctx *c) { return EukaCtxInfo{c->device, c->stream}; }

extern "C" int vgan_euka_create(const vgan_euka_db_view *db, const vgan_damage_view *dmg, const vgan_euka_params *prm,
                                int device, vgan_euka_ctx **out) {
    if (!db || !dmg || !prm || !out) return fail(VGAN_EINVAL, "vgan_euka_create: null argument");
    if (db->n_clades == 0 || !db->clade_dist || !db->bin_off) return fail(VGAN_EINVAL, "vgan_euka_create: empty clade table");
    if (db->n_clades > (1u << 20)) return fail(VGAN_ERANGE, "vgan_euka_create: %u clades (the kernel indexes the per-clade tables in 32 bits: at most 2^20)", db->n_clades);
    if (dmg->n5 == 0 || dmg->n3 == 0) return fail(VGAN_EINVAL, "vgan_euka_create: empty damage tables");
    if (prm->length_to_prof < 0 || prm->length_to_prof > 32) return fail(VGAN_EINVAL, "vgan_euka_create: length_to_prof must be 0..32");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(VGAN_ENODEV, "vgan_euka_create: no HIP device is visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(VGAN_EINVAL, "vgan_euka_create: device %d out of range", device);
    // the read kernel packs a clade's bins as {first bin: 24 bits, count: 8 bits} (euka_kernels.hip: EkBlk)
    if (db->bin_off[db->n_clades] >= (1u << 24))
        return fail(VGAN_ERANGE, "vgan_euka_create: %u coverage bins in all (the kernel addresses 2^24)", db->bin_off[db->n_clades]);
    for (uint32_t cl = 0; cl < db->n_clades; ++cl)
        if (db->bin_off[cl + 1] - db->bin_off[cl] > 255u)
            return fail(VGAN_ERANGE, "vgan_euka_create: clade %u has %u coverage bins (the kernel keeps at most 255 per clade)", cl, db->bin_off[cl + 1] - db->bin_off[cl]);
    HIPCHK(hipSetDevice(device));
    auto c = new vgan_euka_ctx();
    c->device = device;
    c->n_clades = db->n_clades;
    c->n_bins = db->bin_off[db->n_clades];
    c->ltp = prm->length_to_prof;
    // elementary intervals of the bin bounds; per interval the clade the reference's double loop ends on
    std::vector<uint32_t> bp;
    for (uint32_t j = 0; j < c->n_bins; ++j) {
        if (db->bin_lo[j] > db->bin_hi[j]) continue; // std::clamp(lo > hi) is undefined in the reference: never matches here
        bp.push_back((uint32_t)std::max(0, db->bin_lo[j]));
        bp.push_back((uint32_t)std::max(0, db->bin_hi[j]) + 1u);
    }
    std::sort(bp.begin(), bp.end());
    bp.erase(std::unique(bp.begin(), bp.end()), bp.end());
    std::vector<int32_t> bpc(bp.size(), -1);
    for (size_t i = 0; i < bp.size(); ++i) {
        const int64_t node = bp[i];
        for (uint32_t cl = 0; cl < db->n_clades; ++cl)
            for (uint32_t j = db->bin_off[cl]; j < db->bin_off[cl + 1]; ++j)
                if (node >= db->bin_lo[j] && node <= db->bin_hi[j]) bpc[i] = (int32_t)cl;
    }
    // the same lookup per node id
    std::vector<int32_t> node_clade;
    if (!bp.empty() && bp.back() < (1u << 26)) {
        node_clade.assign((size_t)bp.back() + 1, 0);
        for (size_t i = 0; i < bp.size(); ++i) {
            const size_t end = i + 1 < bp.size() ? bp[i + 1] : (size_t)bp.back() + 1;
            const int32_t cl = bpc[i] >= 0 ? bpc[i] : 0;
            for (size_t node = bp[i]; node < end; ++node) node_clade[node] = cl;
        }
    }
    // damage matrix per (5' row, 3' row) pair: row o from the end whose diagonal is smaller (damage.cpp:18-36)
    std::vector<double> pair((size_t)dmg->n5 * dmg->n3 * 20);
    for (uint32_t i5 = 0; i5 < dmg->n5; ++i5)
        for (uint32_t i3 = 0; i3 < dmg->n3; ++i3) {
            double *e = &pair[((size_t)i5 * dmg->n3 + i3) * 20];
            for (int o = 0; o < 4; ++o) {
                const double *r5 = dmg->sub5p + (size_t)i5 * 16 + 4 * o, *r3 = dmg->sub3p + (size_t)i3 * 16 + 4 * o;
                const double *row = r5[o] <= r3[o] ? r5 : r3;
                for (int b = 0; b < 4; ++b) e[4 * b + o] = row[b];
                e[16 + o] = ((row[0] + row[1]) + row[2]) + row[3];
            }
        }
    std::vector<double> tb(356);
    for (int Q = 0; Q < 100; ++Q) tb[(size_t)Q] = Q >= 2 ? pow(10, ((-1 * Q) * 0.1)) : 0.25; // Euka.cpp:38-51
    for (int Q = 0; Q < 256; ++Q) tb[(size_t)(100 + Q)] = 1 - pow(10, ((-1 * Q) * 0.1));     // miscfunc.h:215-216
    int rc = VGAN_OK;
    auto bail = [&](int code) {
        vgan_euka_destroy(c);
        return code;
    };
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev0, hipEventDisableSystemFence) != hipSuccess || hipEventCreateWithFlags(&c->ev1, hipEventDisableSystemFence) != hipSuccess)
        return bail(fail(VGAN_ENODEV, "stream/event creation failed"));
    c->stream = c->own_stream;
    if ((rc = c->bp.upload(bp)) || (rc = c->bp_clade.upload(bpc)) ||
        (rc = c->clade_dist.upload(std::vector<double>(db->clade_dist, db->clade_dist + db->n_clades))) ||
        (rc = c->bin_off.upload(std::vector<uint32_t>(db->bin_off, db->bin_off + db->n_clades + 1))) ||
        (rc = c->bin_lo.upload(std::vector<int32_t>(db->bin_lo, db->bin_lo + c->n_bins))) ||
        (rc = c->bin_hi.upload(std::vector<int32_t>(db->bin_hi, db->bin_hi + c->n_bins))) ||
        (rc = c->sub5p.upload(std::vector<double>(dmg->sub5p, dmg->sub5p + (size_t)dmg->n5 * 16))) ||
        (rc = c->sub3p.upload(std::vector<double>(dmg->sub3p, dmg->sub3p + (size_t)dmg->n3 * 16))) ||
        (rc = c->tables.upload(tb)) || (rc = c->dmg_pair.upload(pair)) ||
        (!node_clade.empty() && (rc = c->node_clade.upload(node_clade))))
        return bail(rc);
    {
        auto up256 = [](size_t n) { return (n + 255) & ~(size_t)255; };
        const size_t n_cc = up256((size_t)EUKA_REPLICAS * c->n_clades * 4), n_ln = n_cc, n_ls = up256((size_t)EUKA_REPLICAS * c->n_clades * 8),
                     n_bs = up256((size_t)EUKA_REPLICAS * c->n_clades * 2 * std::max(1, c->ltp) * 16 * 4),
                     n_bc = up256((size_t)EUKA_REPLICAS * std::max<uint32_t>(1, c->n_bins) * 8);
        c->acc_bytes = n_cc + n_ln + n_ls + n_bs + n_bc + 256;
        if ((rc = c->acc_all.reserve(c->acc_bytes))) return bail(rc);
        uint8_t *q = c->acc_all.p;
        c->bin_cov.p = reinterpret_cast<double *>(q), q += n_bc;
        c->like_logsum.p = reinterpret_cast<double *>(q), q += n_ls;
        c->n_bad.p = reinterpret_cast<unsigned long long *>(q), q += 256;
        c->baseshift.p = reinterpret_cast<uint32_t *>(q), q += n_bs;
        c->clade_count.p = reinterpret_cast<int32_t *>(q), q += n_cc;
        c->like_n.p = reinterpret_cast<uint32_t *>(q);
    }
    c->d.bp = c->bp.p;
    c->d.bp_clade = c->bp_clade.p;
    c->d.n_bp = (uint32_t)bp.size();
    c->d.node_clade = c->node_clade.p;
    c->d.n_node_clade = (uint32_t)node_clade.size();
    c->d.dmg_pair = c->dmg_pair.p;
    c->d.clade_dist = c->clade_dist.p;
    c->d.bin_off = c->bin_off.p;
    c->d.bin_lo = c->bin_lo.p;
    c->d.bin_hi = c->bin_hi.p;
    c->d.sub5p = c->sub5p.p;
    c->d.sub3p = c->sub3p.p;
    c->d.n5 = dmg->n5;
    c->d.n3 = dmg->n3;
    c->d.qscore = c->tables.p;
    c->d.mapq_ok = c->tables.p + 100;
    c->d.n_clades = c->n_clades;
    c->d.min_mapq = prm->min_mapq;
    c->d.ltp = c->ltp;
    if ((rc = vgan_euka_reset(c))) return bail(rc);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(fail(VGAN_ENODEV, "sync failed"));
    *out = c;
    return VGAN_OK;
}

extern "C" void vgan_euka_destroy(vgan_euka_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    c->bp.release();
    c->bin_off.release();
    c->bp_clade.release();
    c->node_clade.release();
    c->dmg_pair.release();
    c->bin_lo.release();
    c->bin_hi.release();
    c->clade_dist.release();
    c->sub5p.release();
    c->sub3p.release();
    c->tables.release();
    c->acc_all.release();
    c->s32.release();
    c->s16.release();
    c->s8.release();
    c->smq.release();
    c->o_clade.release();
    c->o_d.release();
    c->o_pass.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int vgan_euka_set_stream(vgan_euka_ctx *c, void *s) {
    if (!c) return fail(VGAN_EINVAL, "vgan_euka_set_stream: null context");
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return VGAN_OK;
}

extern "C" int vgan_euka_reset(vgan_euka_ctx *c) {
    if (!c) return fail(VGAN_EINVAL, "vgan_euka_reset: null context");
    HIPCHK(hipSetDevice(c->device));
    launch_euka_clear(c->acc_all.p, c->acc_bytes, c->stream);
    HIPCHK(hipGetLastError());
    return VGAN_OK;
}

static void resolve_event(vgan_euka_ctx *c) {
    if (!c->ev_pending) return;
    float ms = 0.f;
    if (hipEventSynchronize(c->ev1) == hipSuccess && hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) {
        c->ms += ms;
        c->launches += 1;
    }
    c->ev_pending = false;
}

extern "C" int vgan_euka_accumulate(vgan_euka_ctx *c, const vgan_euka_batch *b, const vgan_euka_read_out *out) {
    if (!c || !b || !out) return fail(VGAN_EINVAL, "vgan_euka_accumulate: null argument");
    if (b->n_reads == 0) return VGAN_OK;
    if (!out->clade || !out->in_lik || !out->out_lik || !out->like || !out->not_like || !out->pass)
        return fail(VGAN_EINVAL, "vgan_euka_accumulate: null output array");
    HIPCHK(hipSetDevice(c->device));
    resolve_event(c);
    const size_t R = b->n_reads;
    EukaBatchDev d{};
    EukaOutDev o{};
    d.n_reads = b->n_reads;
    int rc;
    if (b->on_device) {
        d.read_col_off = b->read_col_off;
        d.read_qual_off = b->read_qual_off;
        d.read_map_off = b->read_map_off;
        d.read_gseq_len = b->read_gseq_len;
        d.read_rseq_len = b->read_rseq_len;
        d.read_seq_len = b->read_seq_len;
        d.read_mapq = b->read_mapq;
        d.read_rev = b->read_rev;
        d.map_node = b->map_node;
        d.graph_seq = b->graph_seq;
        d.read_seq = b->read_seq;
        d.qual = b->qual;
        o.clade = out->clade;
        o.in_lik = out->in_lik;
        o.out_lik = out->out_lik;
        o.like = out->like;
        o.not_like = out->not_like;
        o.pass = out->pass;
    } else {
        // (the read kernel keeps a read's mapping count and quality length in 16 bits each: EkBlk)
        for (size_t r = 0; r < R; ++r)
            if (b->read_map_off[r + 1] - b->read_map_off[r] > 0xFFFFu || b->read_qual_off[r + 1] - b->read_qual_off[r] > 0xFFFFu)
                return fail(VGAN_ERANGE, "vgan_euka_accumulate: read %zu has more than 65535 mappings or quality bytes", r);
        auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
        if ((rc = c->s32.reserve(3 * up(R + 1) + up(b->n_maps))) || (rc = c->s16.reserve(3 * up(R))) ||
            (rc = c->s8.reserve(up(R) + 2 * up(b->n_cols) + up(b->n_qual))) || (rc = c->smq.reserve(up(R))) ||
            (rc = c->o_clade.reserve(R)) || (rc = c->o_d.reserve(4 * up(R))) || (rc = c->o_pass.reserve(R)))
            return rc;
#define COPY(dst, src, n)                                                                                                \
    do {                                                                                                                 \
        if ((n) > 0) HIPCHK(hipMemcpyAsync((void *)(dst), (src), (n) * sizeof(*(src)), hipMemcpyHostToDevice, c->stream)); \
    } while (0)
        uint32_t *p32 = c->s32.p;
        uint16_t *p16 = c->s16.p;
        uint8_t *p8 = c->s8.p;
        d.read_col_off = p32;
        COPY(p32, b->read_col_off, R + 1);
        p32 += up(R + 1);
        d.read_qual_off = p32;
        COPY(p32, b->read_qual_off, R + 1);
        p32 += up(R + 1);
        d.read_map_off = p32;
        COPY(p32, b->read_map_off, R + 1);
        p32 += up(R + 1);
        d.map_node = p32;
        COPY(p32, b->map_node, (size_t)b->n_maps);
        d.read_gseq_len = p16;
        COPY(p16, b->read_gseq_len, R);
        p16 += up(R);
        d.read_rseq_len = p16;
        COPY(p16, b->read_rseq_len, R);
        p16 += up(R);
        d.read_seq_len = p16;
        COPY(p16, b->read_seq_len, R);
        d.read_mapq = c->smq.p;
        COPY(c->smq.p, b->read_mapq, R);
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
        o.clade = c->o_clade.p;
        o.in_lik = c->o_d.p;
        o.out_lik = c->o_d.p + up(R);
        o.like = c->o_d.p + 2 * up(R);
        o.not_like = c->o_d.p + 3 * up(R);
        o.pass = c->o_pass.p;
    }
    o.clade_count = c->clade_count.p;
    o.baseshift = c->baseshift.p;
    o.bin_cov = c->bin_cov.p;
    o.like_n = c->like_n.p;
    o.like_logsum = c->like_logsum.p;
    o.n_bins = c->n_bins;
    o.n_bad = c->n_bad.p;
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    launch_euka_reads(c->d, d, o, c->stream);
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    c->ev_pending = true;
    HIPCHK(hipGetLastError());
    if (!b->on_device) {
        HIPCHK(hipMemcpyAsync(out->clade, o.clade, R * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(out->in_lik, o.in_lik, R * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(out->out_lik, o.out_lik, R * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(out->like, o.like, R * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(out->not_like, o.not_like, R * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(out->pass, o.pass, R, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return VGAN_OK;
}

extern "C" int vgan_euka_synchronize(vgan_euka_ctx *c) {
    if (!c) return fail(VGAN_EINVAL, "vgan_euka_synchronize: null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VGAN_OK;
}

extern "C" int vgan_euka_finalize(vgan_euka_ctx *c, int32_t *clade_count, uint32_t *baseshift, double *bin_cov, int64_t *n_bad) {
    if (!c) return fail(VGAN_EINVAL, "vgan_euka_finalize: null context");
    HIPCHK(hipSetDevice(c->device));
    {
        EukaOutDev o{};
        o.clade_count = c->clade_count.p;
        o.baseshift = c->baseshift.p;
        o.bin_cov = c->bin_cov.p;
        o.like_n = c->like_n.p;
        o.like_logsum = c->like_logsum.p;
        o.n_bins = c->n_bins;
        launch_euka_reduce(o, c->n_clades, c->ltp, c->stream); // fold the replicas into replica 0
        HIPCHK(hipGetLastError());
    }
    if (clade_count) HIPCHK(hipMemcpyAsync(clade_count, c->clade_count.p, (size_t)c->n_clades * 4, hipMemcpyDeviceToHost, c->stream));
    if (baseshift && c->ltp > 0)
        HIPCHK(hipMemcpyAsync(baseshift, c->baseshift.p, (size_t)c->n_clades * 2 * c->ltp * 16 * 4, hipMemcpyDeviceToHost, c->stream));
    if (bin_cov && c->n_bins) HIPCHK(hipMemcpyAsync(bin_cov, c->bin_cov.p, (size_t)c->n_bins * 8, hipMemcpyDeviceToHost, c->stream));
    unsigned long long nb = 0;
    HIPCHK(hipMemcpyAsync(&nb, c->n_bad.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n_bad) *n_bad = (int64_t)nb;
    return VGAN_OK;
}

extern "C" int vgan_euka_like_sums(vgan_euka_ctx *c, int64_t *n_like, double *sum_log_like) {
    if (!c) return fail(VGAN_EINVAL, "vgan_euka_like_sums: null context");
    HIPCHK(hipSetDevice(c->device));
    // valid after vgan_euka_finalize folded the replicas (replica 0 holds the totals, the others are zero)
    std::vector<uint32_t> n(c->n_clades);
    HIPCHK(hipMemcpyAsync(n.data(), c->like_n.p, (size_t)c->n_clades * 4, hipMemcpyDeviceToHost, c->stream));
    if (sum_log_like)
        HIPCHK(hipMemcpyAsync(sum_log_like, c->like_logsum.p, (size_t)c->n_clades * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n_like)
        for (uint32_t i = 0; i < c->n_clades; ++i) n_like[i] = n[i];
    return VGAN_OK;
}

// Several contexts (one per GPU) of one run: the per-clade accumulators of all of them, summed (counts,
// base-shift tables and likelihood counts are integers and add exactly; bin coverage and log-likelihood sums are doubles).
// The accumulators are a few hundred KB of five arrays of three types: each context finalizes on its device and the sum
// is taken on the host (the HaploCart vector, one array of doubles, goes through RCCL in vgan_hc_reduce).
extern "C" int vgan_euka_reduce(vgan_euka_ctx **ctxs, int n, int32_t *clade_count, uint32_t *baseshift, double *bin_cov,
                                int64_t *n_like, double *sum_log_like, int64_t *n_bad) {
    if (!ctxs || n <= 0) return fail(VGAN_EINVAL, "vgan_euka_reduce: null argument");
    for (int i = 0; i < n; ++i)
        if (!ctxs[i] || ctxs[i]->n_clades != ctxs[0]->n_clades || ctxs[i]->n_bins != ctxs[0]->n_bins || ctxs[i]->ltp != ctxs[0]->ltp)
            return fail(VGAN_EINVAL, "vgan_euka_reduce: contexts of different databases");
    const vgan_euka_ctx *c0 = ctxs[0];
    const size_t C_ = c0->n_clades, NB = c0->n_bins, BS = (size_t)c0->n_clades * 2 * (size_t)std::max(c0->ltp, 0) * 16;
    std::vector<int32_t> cc(C_);
    std::vector<uint32_t> bs(BS);
    std::vector<double> bc(NB), sl(C_);
    std::vector<int64_t> nl(C_);
    if (clade_count) std::fill(clade_count, clade_count + C_, 0);
    if (baseshift) std::fill(baseshift, baseshift + BS, 0u);
    if (bin_cov) std::fill(bin_cov, bin_cov + NB, 0.0);
    if (n_like) std::fill(n_like, n_like + C_, (int64_t)0);
    if (sum_log_like) std::fill(sum_log_like, sum_log_like + C_, 0.0);
    int64_t bad_total = 0;
    for (int i = 0; i < n; ++i) {
        int64_t bad = 0;
        int rc = vgan_euka_finalize(ctxs[i], cc.data(), BS ? bs.data() : nullptr, NB ? bc.data() : nullptr, &bad);
        if (rc) return rc;
        if ((rc = vgan_euka_like_sums(ctxs[i], nl.data(), sl.data()))) return rc;
        bad_total += bad;
        for (size_t k = 0; k < C_; ++k) {
            if (clade_count) clade_count[k] += cc[k];
            if (n_like) n_like[k] += nl[k];
            // a clade without reads holds 0 in a context; -inf (a read of likelihood 0) stays -inf in the sum
            if (sum_log_like) sum_log_like[k] += sl[k];
        }
        if (baseshift)
            for (size_t k = 0; k < BS; ++k) baseshift[k] += bs[k];
        if (bin_cov)
            for (size_t k = 0; k < NB; ++k) bin_cov[k] += bc[k];
    }
    if (n_bad) *n_bad = bad_total;
    return VGAN_OK;
}

extern "C" int vgan_euka_kernel_ms(vgan_euka_ctx *c, double *ms, uint64_t *launches) {
    if (!c) return fail(VGAN_EINVAL, "vgan_euka_kernel_ms: null context");
    HIPCHK(hipSetDevice(c->device));
    resolve_event(c);
    if (ms) *ms = c->ms;
    if (launches) *launches = c->launches;
    c->ms = 0;
    c->launches = 0;
    return VGAN_OK;
}
