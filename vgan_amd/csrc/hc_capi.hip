// C-ABI of the HaploCart device path (include/vgan_gpu.h): context, uploads, launches.
// There is no CPU fallback here: every compute entry point needs a HIP device.
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

vgan::HcCtxInfo vgan::hc_ctx_info(const vgan_hc_ctx *c) { return HcCtxInfo{c->device, c->stream, c->rows}; }

namespace {
struct ScopedTimer {
    vgan_hc_ctx *c;
    int slot;
    hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t get(vgan_hc_ctx *c) {
        if (!c->event_pool.empty()) {
            hipEvent_t e = c->event_pool.back();
            c->event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence); // (timing only: no system-scope fence with every record)
        return e;
    }
    ScopedTimer(vgan_hc_ctx *ctx, int s) : c(ctx), slot(s) {
        if (!c->profiling || (c->profiling == 2 && slot != VGAN_HC_K_SEGMENT)) return;
        a = get(c);
        b = get(c);
        if (a) (void)hipEventRecord(a, c->stream);
    }
    ~ScopedTimer() {
        if (!c->profiling || !a || !b) return;
        (void)hipEventRecord(b, c->stream);
        c->timed.push_back({slot, a, b});
    }
};
} // namespace

namespace {

int stage_batch(vgan_hc_ctx *c, const vgan_hc_batch *b, HcBatchDev &d) {
    d.n_reads = b->n_reads;
    d.n_segments = b->n_segments;
    if (b->on_device) {
        d.read_seg_off = b->read_seg_off;
        d.read_col_off = b->read_col_off;
        d.read_qual_off = b->read_qual_off;
        d.read_algn_len = b->read_algn_len;
        d.read_mapq = b->read_mapq;
        d.seg_node = b->seg_node;
        d.seg_start = b->seg_start;
        d.seg_len = b->seg_len;
        d.graph_seq = b->graph_seq;
        d.algnseq = b->algnseq;
        d.qual = b->qual;
        return VGAN_OK;
    }
    const size_t R = b->n_reads, S = b->n_segments;
    auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
    const size_t n32 = 3 * up(R + 1) + up(S);
    const size_t n16 = up(R) + 2 * up(S);
    const size_t n8 = up(R) + 2 * up(b->n_cols) + up(b->n_qual);
    int rc;
    if ((rc = c->s_u32.reserve(n32)) || (rc = c->s_u16.reserve(n16)) || (rc = c->s_u8.reserve(n8))) return rc;
    uint32_t *p32 = c->s_u32.p;
    uint16_t *p16 = c->s_u16.p;
    uint8_t *p8 = c->s_u8.p;
#define COPY(dst, src, n)                                                                                \
    do {                                                                                                 \
        if ((n) > 0) HIPCHK(hipMemcpyAsync((void *)(dst), (src), (n) * sizeof(*(src)), hipMemcpyHostToDevice, c->stream)); \
    } while (0)
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
    d.read_algn_len = p16;
    COPY(p16, b->read_algn_len, R);
    p16 += up(R);
    d.seg_start = p16;
    COPY(p16, b->seg_start, S);
    p16 += up(S);
    d.seg_len = p16;
    COPY(p16, b->seg_len, S);
    d.read_mapq = p8;
    COPY(p8, b->read_mapq, R);
    p8 += up(R);
    d.graph_seq = p8;
    COPY(p8, b->graph_seq, (size_t)b->n_cols);
    p8 += up(b->n_cols);
    d.algnseq = p8;
    COPY(p8, b->algnseq, (size_t)b->n_cols);
    p8 += up(b->n_cols);
    d.qual = p8;
    COPY(p8, b->qual, (size_t)b->n_qual);
#undef COPY
    return VGAN_OK;
}

int check_batch(const vgan_hc_batch *b) {
    if (!b) return fail(VGAN_EINVAL, "null batch");
    if (b->n_reads == 0) return VGAN_OK;
    if (!b->read_seg_off || !b->read_col_off || !b->read_qual_off || !b->read_algn_len || !b->read_mapq)
        return fail(VGAN_EINVAL, "batch: null per-read array");
    if (b->n_segments && (!b->seg_node || !b->seg_start || !b->seg_len)) return fail(VGAN_EINVAL, "batch: null per-segment array");
    if (b->n_cols && (!b->graph_seq || !b->algnseq)) return fail(VGAN_EINVAL, "batch: null sequence array");
    if (b->n_qual && !b->qual) return fail(VGAN_EINVAL, "batch: null quality array");
    return VGAN_OK;
}

// VGAN_HC_KERNEL=tile keeps the tileable reads on the LDS-tiled kernel (developer aid: A/B runs of the two data paths)
bool wave_kernel_enabled() {
    const char *e = getenv("VGAN_HC_KERNEL"); // (read per call: a test flips it between two accumulates)
    return !(e && strcmp(e, "tile") == 0);
}

// The layout pass over reads [0, nt) of the staged batch d into P (hc_device.h: HcPackedDev).  max_segs / max_qual: the
// largest per-read counts when the caller knows them (host arrays); otherwise (readback) they are taken from the device,
// which synchronises the stream.
int pack_into(vgan_hc_ctx *c, const vgan_hc_batch *b, const HcBatchDev &d, uint32_t nt, vgan_hc_packed &P, bool readback,
              uint32_t max_segs, uint32_t max_qual, uint32_t max_cols, uint32_t qual_excess = 0) {
    int rc;
    P.device = c->device;
    if ((rc = P.rhdr.reserve((size_t)nt + 1)) || (rc = P.srec.reserve(std::max<size_t>(1, b->n_segments))) ||
        (rc = P.crec.reserve(std::max<size_t>(1, b->n_cols))) || (rc = P.qualp.reserve((size_t)b->n_qual + 32)) ||
        (rc = P.maxima.reserve(4)))
        return rc;
    launch_hc_pack(d, nt, b->n_cols, b->n_qual, P.rhdr.p, P.srec.p, P.crec.p, P.qualp.p, readback ? P.maxima.p : nullptr, c->stream);
    HIPCHK(hipGetLastError());
    if (readback) {
        uint32_t mx[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(mx, P.maxima.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        max_segs = mx[0];
        max_qual = mx[1];
        max_cols = mx[2];
        qual_excess = mx[3];
    }
    P.d.rhdr = P.rhdr.p;
    P.d.srec = P.srec.p;
    P.d.crec = P.crec.p;
    P.d.qualp = P.qualp.p;
    P.d.n_reads = nt;
    P.d.n_segments = b->n_segments;
    P.d.n_cols = b->n_cols;
    P.d.n_qual = b->n_qual;
    P.d.max_read_segs = max_segs;
    P.d.max_read_qual = max_qual;
    P.d.max_read_cols = max_cols;
    P.d.qual_excess = qual_excess;
    return VGAN_OK;
}

int ensure_work_queue(vgan_hc_ctx *c);

// D_m per segment (segD) and / or W[node] += D_m (nodeW) and the totals for every read of the batch: the tileable reads
// through the wave kernel on their packed form (the batch's companion, or the context's scratch filled here), the others
// through the general kernel.  `staged`: d holds the batch's arrays on the device already.
int run_segments(vgan_hc_ctx *c, const vgan_hc_batch *b, double *segD, double *nodeW, double *totals, HcBatchDev *staged = nullptr) {
    int rc;
    const uint32_t nt = std::min(b->n_tileable, b->n_reads);
    if (nodeW || totals) c->touched = true;
    const uint32_t mean_cols = (uint32_t)(b->n_cols / std::max<uint32_t>(1, b->n_reads));
    const uint32_t mean_segs = b->n_segments / std::max<uint32_t>(1, b->n_reads);
    const vgan_hc_packed *pk = nullptr;
    if (nt && b->packed) {
        if (b->packed->device != c->device || b->packed->d.n_reads != nt || b->packed->d.n_segments != b->n_segments)
            return fail(VGAN_EINVAL, "batch: the packed companion belongs to another batch or device");
        if (hc_wave_kernel_fits(b->packed->d.max_read_segs, b->packed->d.max_read_qual, b->packed->d.max_read_cols, mean_segs, mean_cols))
            pk = b->packed;
    }
    HcBatchDev d{};
    const bool need_arrays = staged || !pk || nt < b->n_reads || !wave_kernel_enabled();
    if (need_arrays && (rc = stage_batch(c, b, d))) return rc;
    if (staged) *staged = d;
    if (nt && !pk && !b->on_device && wave_kernel_enabled()) {
        // host arrays: the per-read maxima that select the kernel variant cost one pass over the offsets here
        uint32_t ms = 0, mq = 0, mc = 0, mx = 0;
        for (uint32_t r = 0; r < nt; ++r) {
            const uint32_t q = b->read_qual_off[r + 1] - b->read_qual_off[r], cl = b->read_col_off[r + 1] - b->read_col_off[r];
            ms = std::max(ms, b->read_seg_off[r + 1] - b->read_seg_off[r]);
            mq = std::max(mq, q);
            mc = std::max(mc, cl);
            mx = std::max(mx, q > cl ? q - cl : 0u);
        }
        if (hc_wave_kernel_fits(ms, mq, mc, mean_segs, mean_cols)) {
            ScopedTimer t(c, VGAN_HC_K_PACK);
            if ((rc = pack_into(c, b, d, nt, c->scratch_pack, false, ms, mq, mc, mx))) return rc;
            pk = &c->scratch_pack;
        }
    }
    ScopedTimer t(c, VGAN_HC_K_SEGMENT);
    if (pk && wave_kernel_enabled()) {
        if ((rc = ensure_work_queue(c))) return rc;
        if (nodeW && !segD && hc_col8_kernel_fits(c->g, pk->d)) launch_hc_segments_col8(c->g, pk->d, c->prm, nodeW, totals, c->stream);
        else launch_hc_segments_wave(c->g, pk->d, c->prm, segD, nodeW, totals, c->work_ctr.p, &c->work_base, c->stream);
        if (hipPeekAtLastError() != hipSuccess) c->work_dirty = true;
        launch_hc_segments_general(c->g, d, c->prm, nt, nullptr, nullptr, segD, nodeW, totals, c->stream);
    } else {
        // (a device batch without a companion, reads beyond every variant of the wave kernel)
        launch_hc_segments(c->g, d, c->prm, nt, mean_cols, nullptr, nullptr, segD, nodeW, totals, c->stream);
    }
    return VGAN_OK;
}

// A packed batch on the device: in place when it is there already, else copied as it is into the context's scratch
// (the one copy of the batch HBM holds; no layout pass).
// want_qual false: the kernel that takes the batch reads the quality bytes from the column records (hc_col8_kernels.hip), and
// the second copy of the quality strings is neither asked for nor sent.
int stage_packed(vgan_hc_ctx *c, const vgan_hc_packed_view *v, HcPackedDev &d, bool want_qual = true) {
    if (!v) return fail(VGAN_EINVAL, "null packed batch");
    d = HcPackedDev{};
    d.n_reads = v->n_reads;
    d.n_segments = v->n_segments;
    d.n_cols = v->n_cols;
    d.n_qual = v->n_qual;
    d.max_read_segs = v->max_read_segs;
    d.max_read_qual = v->max_read_qual;
    d.max_read_cols = v->max_read_cols;
    d.max_read_node_span = v->max_read_node_span;
    if (v->n_reads == 0) return VGAN_OK;
    if (!v->rhdr || !v->srec || !v->crec || (want_qual && !v->qualp)) return fail(VGAN_EINVAL, "packed batch: null array");
    if (v->max_read_segs > HC_TILE_MAX_READ_SEGS || v->max_read_qual > HC_TILE_MAX_READ_QUAL || v->max_read_cols > HC_TILE_MAX_READ_COLS ||
        v->max_read_segs == 0)
        return fail(VGAN_EINVAL, "packed batch: per-read maxima missing or beyond the tile contract (512 segments / 1280 quality bytes / 1280 columns)");
    if (v->n_cols > 0xFFFFFFF0ull || v->n_qual > 0xFFFFFFF0ull) return fail(VGAN_ERANGE, "packed batch: more than 2^32 columns or quality bytes");
    if (v->on_device) {
        d.rhdr = reinterpret_cast<const uint4 *>(v->rhdr);
        d.srec = v->srec;
        d.crec = v->crec;
        d.qualp = v->qualp;
        return VGAN_OK;
    }
    { // host arrays: the end offsets must be the stated totals (the kernels bound their loads by them; the full check is vgan_hc_packed_validate)
        const uint32_t *e = v->rhdr + 4 * (size_t)v->n_reads;
        if (v->rhdr[0] != 0 || v->rhdr[2] != 0 || e[0] != v->n_segments || e[2] != v->n_cols || e[1] != v->n_qual)
            return fail(VGAN_EINVAL, "packed batch: the headers' first / last offsets do not match n_segments / n_qual / n_cols");
    }
    int rc;
    vgan_hc_packed &P = c->scratch_pack;
    P.device = c->device;
    if ((rc = P.rhdr.reserve((size_t)v->n_reads + 1)) || (rc = P.srec.reserve(std::max<size_t>(1, v->n_segments))) ||
        (rc = P.crec.reserve(std::max<size_t>(1, v->n_cols))) || (want_qual && (rc = P.qualp.reserve((size_t)v->n_qual + 32))))
        return rc;
    HIPCHK(hipMemcpyAsync(P.rhdr.p, v->rhdr, ((size_t)v->n_reads + 1) * 16, hipMemcpyHostToDevice, c->stream));
    if (v->n_segments) HIPCHK(hipMemcpyAsync(P.srec.p, v->srec, (size_t)v->n_segments * 4, hipMemcpyHostToDevice, c->stream));
    if (v->n_cols) HIPCHK(hipMemcpyAsync(P.crec.p, v->crec, (size_t)v->n_cols * 4, hipMemcpyHostToDevice, c->stream));
    if (want_qual) HIPCHK(hipMemcpyAsync(P.qualp.p, v->qualp, (size_t)v->n_qual + 32, hipMemcpyHostToDevice, c->stream));
    d.rhdr = P.rhdr.p;
    d.srec = P.srec.p;
    d.crec = P.crec.p;
    d.qualp = want_qual ? P.qualp.p : nullptr;
    return VGAN_OK;
}

int ensure_work_queue(vgan_hc_ctx *c) {
    int rc;
    if (!c->work_ctr.p || c->work_dirty) {
        if (!c->work_ctr.p && (rc = c->work_ctr.reserve(16))) return rc;
        HIPCHK(hipMemsetAsync(c->work_ctr.p, 0, 64, c->stream));
        c->work_base = 0;
        c->work_dirty = false;
    }
    return VGAN_OK;
}

// the wave kernel over a packed batch on the device: D_m per segment and / or W[node] += D_m, and the totals
int run_packed(vgan_hc_ctx *c, const HcPackedDev &d, double *segD, double *nodeW, double *totals) {
    int rc;
    if (d.n_reads == 0) return VGAN_OK;
    if (nodeW || totals) c->touched = true;
    if (!wave_kernel_enabled())
        return fail(VGAN_EINVAL, "VGAN_HC_KERNEL=tile asks for the LDS-tiled kernel, which reads SoA batches: a packed batch has no such form");
    if ((rc = ensure_work_queue(c))) return rc;
    ScopedTimer t(c, VGAN_HC_K_SEGMENT);
    // node-weights accumulation alone: eight columns to a lane and the table of column terms, where the batch and the graph fit
    if (nodeW && !segD && hc_col8_kernel_fits(c->g, d)) launch_hc_segments_col8(c->g, d, c->prm, nodeW, totals, c->stream);
    else if (!d.qualp) return fail(VGAN_EINVAL, "packed batch: this batch takes a kernel that reads the quality strings' own copy (qualp), which the view does not carry");
    else launch_hc_segments_wave(c->g, d, c->prm, segD, nodeW, totals, c->work_ctr.p, &c->work_base, c->stream);
    if (hipGetLastError() != hipSuccess) {
        c->work_dirty = true; // (the host's mirror of the ticket counter no longer holds: the next launch starts it over)
        return fail(VGAN_ENODEV, "the segment kernel could not be launched");
    }
    return VGAN_OK;
}

} // namespace

extern "C" int vgan_hc_set_stream(vgan_hc_ctx *c, void *hip_stream) {
    if (!c) return fail(VGAN_EINVAL, "vgan_hc_set_stream: null context");
    // (what is queued on the stream left behind is not waited for: a caller that changes streams between two accumulates
    // orders them itself -- include/vgan_gpu.h; the work queue's counter starts over on the new stream)
    const hipStream_t ns = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    if (ns != c->stream) c->work_dirty = true;
    c->stream = ns;
    return VGAN_OK;
}

extern "C" int vgan_hc_set_mode(vgan_hc_ctx *c, int mode) {
    if (!c) return fail(VGAN_EINVAL, "vgan_hc_set_mode: null context");
    if (mode != VGAN_HC_MODE_NODE_WEIGHTS && mode != VGAN_HC_MODE_PER_READ && mode != VGAN_HC_MODE_PER_READ_DENSE)
        return fail(VGAN_EINVAL, "vgan_hc_set_mode: unknown mode %d", mode);
    c->mode = mode;
    return VGAN_OK;
}

extern "C" int vgan_hc_reset(vgan_hc_ctx *c) {
    if (!c) return fail(VGAN_EINVAL, "vgan_hc_reset: null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemsetAsync(c->accum.p, 0, c->accum_n * 8, c->stream));
    c->touched = false;
    return VGAN_OK;
}

extern "C" int vgan_hc_batch_validate(const vgan_hc_ctx *c, const vgan_hc_batch *b) {
    if (!c) return fail(VGAN_EINVAL, "vgan_hc_batch_validate: null context");
    int rc = check_batch(b);
    if (rc) return rc;
    if (b->on_device) return fail(VGAN_EINVAL, "vgan_hc_batch_validate: the batch must be in host memory");
    const uint64_t R = b->n_reads, S = b->n_segments;
    if (R == 0) return VGAN_OK;
    if (b->n_tileable > R) return fail(VGAN_EINVAL, "batch: n_tileable exceeds n_reads");
    if (b->read_seg_off[0] != 0 || b->read_col_off[0] != 0 || b->read_qual_off[0] != 0)
        return fail(VGAN_EINVAL, "batch: offsets do not start at 0");
    if (b->read_seg_off[R] != S || b->read_col_off[R] != b->n_cols || b->read_qual_off[R] != b->n_qual)
        return fail(VGAN_EINVAL, "batch: final offsets do not match n_segments / n_cols / n_qual");
    for (uint64_t r = 0; r < R; ++r) {
        const uint32_t s0 = b->read_seg_off[r], s1 = b->read_seg_off[r + 1], c0 = b->read_col_off[r], c1 = b->read_col_off[r + 1];
        if (s1 < s0 || c1 < c0 || b->read_qual_off[r + 1] < b->read_qual_off[r])
            return fail(VGAN_EINVAL, "batch: offsets of read %llu descend", (unsigned long long)r);
        const uint32_t cols = c1 - c0, ql = b->read_qual_off[r + 1] - b->read_qual_off[r];
        if (b->read_algn_len[r] > cols) return fail(VGAN_EINVAL, "batch: |algnseq| of read %llu exceeds its column region", (unsigned long long)r);
        if (b->read_mapq[r] > 99) return fail(VGAN_EINVAL, "batch: mapping quality of read %llu exceeds 99", (unsigned long long)r);
        if (cols > 65535 || ql > 65535) return fail(VGAN_EINVAL, "batch: read %llu has more than 65535 columns or quality bytes", (unsigned long long)r);
        const bool tile = r < b->n_tileable;
        if (tile && (cols > HC_TILE_MAX_READ_COLS || ql > HC_TILE_MAX_READ_QUAL || s1 - s0 > HC_TILE_MAX_READ_SEGS))
            return fail(VGAN_EINVAL, "batch: read %llu is below n_tileable but exceeds the tile limits", (unsigned long long)r);
        if (tile && s1 == s0) return fail(VGAN_EINVAL, "batch: read %llu is below n_tileable but has no segment", (unsigned long long)r);
        if (tile && b->read_algn_len[r] != cols)
            return fail(VGAN_EINVAL, "batch: read %llu is below n_tileable but |algnseq| differs from its column count", (unsigned long long)r);
        uint32_t prev_end = 0, prev_start = 0;
        for (uint32_t s = s0; s < s1; ++s) {
            const uint32_t st = b->seg_start[s], ln = b->seg_len[s];
            if (b->seg_node[s] >= c->rows) return fail(VGAN_EINVAL, "batch: segment %u names node %u, beyond the graph", s, b->seg_node[s]);
            if ((uint64_t)st + ln > cols) return fail(VGAN_EINVAL, "batch: segment %u leaves the columns of read %llu", s, (unsigned long long)r);
            if (st < prev_start) return fail(VGAN_EINVAL, "batch: segments of read %llu do not ascend in seg_start", (unsigned long long)r);
            if (tile && ln == 0) return fail(VGAN_EINVAL, "batch: read %llu is below n_tileable but has an empty segment", (unsigned long long)r);
            if (tile && st < prev_end) return fail(VGAN_EINVAL, "batch: read %llu is below n_tileable but its segments overlap", (unsigned long long)r);
            prev_start = st;
            if (ln) prev_end = st + ln;
        }
    }
    return VGAN_OK;
}

extern "C" int vgan_hc_pack(vgan_hc_ctx *c, const vgan_hc_batch *b, vgan_hc_packed **out) {
    if (!c || !out) return fail(VGAN_EINVAL, "vgan_hc_pack: null argument");
    *out = nullptr;
    int rc = check_batch(b);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    auto P = new vgan_hc_packed();
    const uint32_t nt = std::min(b->n_tileable, b->n_reads);
    HcBatchDev d{};
    if (nt) {
        if ((rc = stage_batch(c, b, d)) || (rc = pack_into(c, b, d, nt, *P, true, 0, 0, 0))) {
            P->release();
            delete P;
            return rc;
        }
    } else {
        P->device = c->device;
        P->d.n_segments = b->n_segments;
    }
    *out = P;
    return VGAN_OK;
}

extern "C" void vgan_hc_packed_free(vgan_hc_packed *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    p->release();
    delete p;
}

extern "C" int vgan_hc_accumulate(vgan_hc_ctx *c, const vgan_hc_batch *b) {
    if (!c) return fail(VGAN_EINVAL, "vgan_hc_accumulate: null context");
    int rc = check_batch(b);
    if (rc) return rc;
    if (b->n_reads == 0) return VGAN_OK;
    HIPCHK(hipSetDevice(c->device));
    if (c->mode == VGAN_HC_MODE_NODE_WEIGHTS) {
        // W[node] += D_m inside the segment kernels (LDS window over the node ids of a wave's reads): no per-segment
        // array leaves the chip
        if ((rc = run_segments(c, b, nullptr, c->nodeW.p, c->totals.p))) return rc;
    } else {
        if ((rc = c->segD.reserve(b->n_segments))) return rc;
        HcBatchDev d{}; // (the sweep reads the node ids from the batch's own arrays)
        if ((rc = run_segments(c, b, c->segD.p, nullptr, c->totals.p, &d))) return rc;
        ScopedTimer t(c, VGAN_HC_K_SWEEP_SEG);
        launch_hc_sweep(c->g, d.seg_node, c->segD.p, b->n_segments, c->mode == VGAN_HC_MODE_PER_READ, c->acc_seg.p, c->stream);
    }
    HIPCHK(hipGetLastError());
    return VGAN_OK;
}

extern "C" int vgan_hc_accumulate_packed(vgan_hc_ctx *c, const vgan_hc_packed_view *v) {
    if (!c) return fail(VGAN_EINVAL, "vgan_hc_accumulate_packed: null context");
    if (!v) return fail(VGAN_EINVAL, "vgan_hc_accumulate_packed: null batch");
    if (v->n_reads == 0) return VGAN_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    HcPackedDev d{};
    // (node-weights accumulation through hc_segment_col8_kernel: the quality bytes are read from the column records, their second
    // copy stays where it is -- 150 of a 150 bp read's 1006 bytes that do not cross the link)
    bool want_qual = true;
    if (c->mode == VGAN_HC_MODE_NODE_WEIGHTS) {
        HcPackedDev probe{};
        probe.n_reads = v->n_reads;
        probe.n_segments = v->n_segments;
        probe.n_cols = v->n_cols;
        probe.max_read_segs = v->max_read_segs;
        probe.max_read_qual = v->max_read_qual;
        probe.max_read_cols = v->max_read_cols;
        probe.max_read_node_span = v->max_read_node_span;
        want_qual = !hc_col8_kernel_fits(c->g, probe);
    }
    if ((rc = stage_packed(c, v, d, want_qual))) return rc;
    if (c->mode == VGAN_HC_MODE_NODE_WEIGHTS) return run_packed(c, d, nullptr, c->nodeW.p, c->totals.p);
    // the reference's loop order: D_m per segment, then every segment's mask row over the P accumulators
    if ((rc = c->segD.reserve(v->n_segments)) || (rc = c->s_u32.reserve(v->n_segments))) return rc;
    if ((rc = run_packed(c, d, c->segD.p, nullptr, c->totals.p))) return rc;
    launch_hc_srec_nodes(d.srec, v->n_segments, c->s_u32.p, c->stream);
    ScopedTimer t(c, VGAN_HC_K_SWEEP_SEG);
    launch_hc_sweep(c->g, c->s_u32.p, c->segD.p, v->n_segments, c->mode == VGAN_HC_MODE_PER_READ, c->acc_seg.p, c->stream);
    HIPCHK(hipGetLastError());
    return VGAN_OK;
}

extern "C" int vgan_hc_segment_weights_packed(vgan_hc_ctx *c, const vgan_hc_packed_view *v, double *D) {
    if (!c || !v || !D) return fail(VGAN_EINVAL, "vgan_hc_segment_weights_packed: null argument");
    if (v->n_reads == 0 || v->n_segments == 0) return VGAN_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    HcPackedDev d{};
    if ((rc = stage_packed(c, v, d)) || (rc = c->segD.reserve(v->n_segments))) return rc;
    if ((rc = run_packed(c, d, c->segD.p, nullptr, nullptr))) return rc;
    HIPCHK(hipMemcpyAsync(D, c->segD.p, (size_t)v->n_segments * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VGAN_OK;
}

extern "C" int vgan_hc_packed_validate(const vgan_hc_ctx *c, const vgan_hc_packed_view *v) {
    if (!c || !v) return fail(VGAN_EINVAL, "vgan_hc_packed_validate: null argument");
    if (v->on_device) return fail(VGAN_EINVAL, "vgan_hc_packed_validate: the batch must be in host memory");
    const uint64_t R = v->n_reads, S = v->n_segments;
    if (R == 0) return VGAN_OK;
    if (!v->rhdr || !v->srec || !v->crec || !v->qualp) return fail(VGAN_EINVAL, "packed batch: null array");
    const uint32_t *h = v->rhdr;
    if (h[0] != 0 || h[1] != 0 || h[2] != 0) return fail(VGAN_EINVAL, "packed batch: offsets do not start at 0");
    if (h[4 * R] != S || h[4 * R + 1] != v->n_qual || h[4 * R + 2] != v->n_cols)
        return fail(VGAN_EINVAL, "packed batch: final offsets do not match n_segments / n_qual / n_cols");
    uint32_t ms = 0, mq = 0, mc = 0;
    for (uint64_t r = 0; r < R; ++r) {
        const uint32_t *a = h + 4 * r, *b = a + 4;
        if (b[0] < a[0] || b[1] < a[1] || b[2] < a[2]) return fail(VGAN_EINVAL, "packed batch: offsets of read %llu descend", (unsigned long long)r);
        const uint32_t ns = b[0] - a[0], nq = b[1] - a[1], ncol = b[2] - a[2], A = a[3] & 0xFFFFu, mapq = a[3] >> 16;
        if (ns == 0 || ns > HC_TILE_MAX_READ_SEGS || nq > HC_TILE_MAX_READ_QUAL || ncol > HC_TILE_MAX_READ_COLS)
            return fail(VGAN_EINVAL, "packed batch: read %llu is outside the tile contract", (unsigned long long)r);
        if (nq > ncol) return fail(VGAN_EINVAL, "packed batch: the quality string of read %llu is longer than its columns", (unsigned long long)r);
        if (A != ncol) return fail(VGAN_EINVAL, "packed batch: |algnseq| of read %llu differs from its column count", (unsigned long long)r);
        if (mapq > 99) return fail(VGAN_EINVAL, "packed batch: mapping quality of read %llu exceeds 99", (unsigned long long)r);
        ms = std::max(ms, ns);
        mq = std::max(mq, nq);
        mc = std::max(mc, ncol);
        uint32_t prev_end = 0;
        for (uint32_t s = a[0]; s < b[0]; ++s) {
            const uint32_t w = v->srec[s], node = w & VGAN_HC_SREC_MAX_NODE, st = (w >> 18) & 0x7FFu;
            if (node >= c->rows) return fail(VGAN_EINVAL, "packed batch: segment %u names node %u, beyond the graph", s, node);
            if ((w >> 29) != ((uint32_t)r & 7u)) return fail(VGAN_EINVAL, "packed batch: segment %u does not carry its read's index", s);
            if (st < prev_end || st >= ncol) return fail(VGAN_EINVAL, "packed batch: segments of read %llu overlap, descend or leave its columns", (unsigned long long)r);
            if (!(v->crec[(size_t)a[2] + st] & VGAN_HC_CREC_HEAD)) return fail(VGAN_EINVAL, "packed batch: segment %u has no head bit at its first column", s);
            prev_end = st + 1;
        }
        uint32_t heads = 0;
        for (uint32_t col = 0; col < ncol; ++col) heads += v->crec[(size_t)a[2] + col] >= VGAN_HC_CREC_HEAD ? 1u : 0u;
        if (heads != ns) return fail(VGAN_EINVAL, "packed batch: read %llu has %u head bits for %u segments", (unsigned long long)r, heads, ns);
    }
    if (ms > v->max_read_segs || mq > v->max_read_qual || mc > v->max_read_cols)
        return fail(VGAN_EINVAL, "packed batch: a read exceeds the stated per-read maxima");
    for (int i = 0; i < 32; ++i)
        if (v->qualp[v->n_qual + i] != 0) return fail(VGAN_EINVAL, "packed batch: the quality array is not followed by 32 zero bytes");
    return VGAN_OK;
}

extern "C" int vgan_hc_packed_view_download(const vgan_hc_packed_view *v, uint32_t *rhdr, uint32_t *srec, uint32_t *crec, uint8_t *qualp) {
    if (!v) return fail(VGAN_EINVAL, "vgan_hc_packed_view_download: null argument");
    if (v->n_reads == 0) return VGAN_OK;
    const hipMemcpyKind k = v->on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost;
    if (rhdr) HIPCHK(hipMemcpy(rhdr, v->rhdr, ((size_t)v->n_reads + 1) * 16, k));
    if (srec && v->n_segments) HIPCHK(hipMemcpy(srec, v->srec, (size_t)v->n_segments * 4, k));
    if (crec && v->n_cols) HIPCHK(hipMemcpy(crec, v->crec, (size_t)v->n_cols * 4, k));
    if (qualp) HIPCHK(hipMemcpy(qualp, v->qualp, (size_t)v->n_qual + 32, k));
    return VGAN_OK;
}

extern "C" int vgan_hc_packed_download(const vgan_hc_packed *p, uint64_t n[4], uint32_t *rhdr, uint32_t *srec, uint32_t *crec, uint8_t *qualp) {
    if (!p) return fail(VGAN_EINVAL, "vgan_hc_packed_download: null argument");
    HIPCHK(hipSetDevice(p->device));
    if (n) {
        n[0] = p->d.n_reads;
        n[1] = p->d.n_segments;
        n[2] = p->d.n_cols;
        n[3] = p->d.n_qual;
    }
    if (p->d.n_reads == 0) return VGAN_OK;
    if (rhdr) HIPCHK(hipMemcpy(rhdr, p->d.rhdr, ((size_t)p->d.n_reads + 1) * 16, hipMemcpyDeviceToHost));
    if (srec && p->d.n_segments) HIPCHK(hipMemcpy(srec, p->d.srec, (size_t)p->d.n_segments * 4, hipMemcpyDeviceToHost));
    if (crec && p->d.n_cols) HIPCHK(hipMemcpy(crec, p->d.crec, (size_t)p->d.n_cols * 4, hipMemcpyDeviceToHost));
    if (qualp) HIPCHK(hipMemcpy(qualp, p->d.qualp, (size_t)p->d.n_qual + 32, hipMemcpyDeviceToHost));
    return VGAN_OK;
}

extern "C" int vgan_hc_segment_scalars(vgan_hc_ctx *c, const vgan_hc_batch *b, double *S, double *U) {
    if (!c || !S || !U) return fail(VGAN_EINVAL, "vgan_hc_segment_scalars: null argument");
    int rc = check_batch(b);
    if (rc) return rc;
    if (b->n_reads == 0 || b->n_segments == 0) return VGAN_OK;
    HIPCHK(hipSetDevice(c->device));
    HcBatchDev d{};
    if ((rc = stage_batch(c, b, d))) return rc;
    if ((rc = c->segS.reserve(b->n_segments)) || (rc = c->segU.reserve(b->n_segments))) return rc;
    launch_hc_segments(c->g, d, c->prm, b->n_tileable, (uint32_t)(b->n_cols / std::max<uint32_t>(1, b->n_reads)), c->segS.p, c->segU.p, nullptr, nullptr, nullptr, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(S, c->segS.p, (size_t)b->n_segments * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(U, c->segU.p, (size_t)b->n_segments * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VGAN_OK;
}

extern "C" int vgan_hc_segment_weights(vgan_hc_ctx *c, const vgan_hc_batch *b, double *D) {
    if (!c || !D) return fail(VGAN_EINVAL, "vgan_hc_segment_weights: null argument");
    int rc = check_batch(b);
    if (rc) return rc;
    if (b->n_reads == 0 || b->n_segments == 0) return VGAN_OK;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = c->segD.reserve(b->n_segments))) return rc;
    if ((rc = run_segments(c, b, c->segD.p, nullptr, nullptr))) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(D, c->segD.p, (size_t)b->n_segments * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VGAN_OK;
}

extern "C" int vgan_hc_read_loglik(vgan_hc_ctx *c, const vgan_hc_batch *b, double *out) {
    if (!c || !out) return fail(VGAN_EINVAL, "vgan_hc_read_loglik: null argument");
    int rc = check_batch(b);
    if (rc) return rc;
    if (b->n_reads == 0) return VGAN_OK;
    if (b->n_reads > 65535) return fail(VGAN_ERANGE, "vgan_hc_read_loglik: at most 65535 reads per call");
    HIPCHK(hipSetDevice(c->device));
    HcBatchDev d{};
    if ((rc = stage_batch(c, b, d))) return rc;
    const size_t n = (size_t)b->n_reads * c->P;
    if ((rc = c->segS.reserve(b->n_segments + 1)) || (rc = c->segU.reserve(b->n_segments + 1)) || (rc = c->dump.reserve(n)))
        return rc;
    launch_hc_segments(c->g, d, c->prm, b->n_tileable, (uint32_t)(b->n_cols / std::max<uint32_t>(1, b->n_reads)), c->segS.p, c->segU.p, nullptr, nullptr, nullptr, c->stream);
    launch_hc_read_loglik(c->g, d, c->segS.p, c->segU.p, c->dump.p, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, c->dump.p, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VGAN_OK;
}

extern "C" int vgan_hc_finalize(vgan_hc_ctx *c, double *d_out, double *out) {
    if (!c) return fail(VGAN_EINVAL, "vgan_hc_finalize: null context");
    HIPCHK(hipSetDevice(c->device));
    // NODE_WEIGHTS accumulations: one pass of the unsupported-path bitmask over the node weights (into acc_node, which every
    // finalize leaves zero again: calling it twice gives the same vector)
    {
        ScopedTimer t(c, VGAN_HC_K_SWEEP_NODE);
        launch_hc_sweep(c->g, nullptr, c->nodeW.p, c->rows, 1, c->acc_node.p, c->stream);
    }
    {
        ScopedTimer t(c, VGAN_HC_K_FINISH);
        launch_hc_finish(c->totals.p, c->acc_seg.p, c->acc_node.p, c->P, c->W * 64u, c->final_vec.p, d_out, c->stream);
    }
    HIPCHK(hipGetLastError());
    if (out) {
        HIPCHK(hipMemcpyAsync(out, c->final_vec.p, (size_t)c->P * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return VGAN_OK;
}

extern "C" int vgan_hc_synchronize(vgan_hc_ctx *c) {
    if (!c) return fail(VGAN_EINVAL, "vgan_hc_synchronize: null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VGAN_OK;
}

static void resolve_timers(vgan_hc_ctx *c) {
    for (auto &t : c->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            c->prof_ms[t.slot] += ms;
            c->prof_n[t.slot] += 1;
        }
        c->event_pool.push_back(t.a);
        c->event_pool.push_back(t.b);
    }
    c->timed.clear();
}

extern "C" int vgan_hc_profile_enable(vgan_hc_ctx *c, int enable) {
    if (!c) return fail(VGAN_EINVAL, "vgan_hc_profile_enable: null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    resolve_timers(c);
    for (int i = 0; i < VGAN_HC_K_COUNT; ++i) {
        c->prof_ms[i] = 0;
        c->prof_n[i] = 0;
    }
    c->profiling = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    return VGAN_OK;
}

extern "C" int vgan_hc_profile_read(vgan_hc_ctx *c, double ms[5], uint64_t launches[5]) {
    if (!c || !ms || !launches) return fail(VGAN_EINVAL, "vgan_hc_profile_read: null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    resolve_timers(c);
    for (int i = 0; i < VGAN_HC_K_COUNT; ++i) {
        ms[i] = c->prof_ms[i];
        launches[i] = c->prof_n[i];
    }
    return VGAN_OK;
}

extern "C" int vgan_hc_argmax(const double *v, uint32_t n) {
    if (!v || n == 0) return fail(VGAN_EINVAL, "vgan_hc_argmax: empty vector");
    uint32_t best = 0;
    for (uint32_t i = 1; i < n; ++i)
        if (v[i] > v[best]) best = i; // std::max_element: first maximum
    // (include/vgan_gpu.h) the first of the paths whose sums are that maximum up to the order of the additions
    const double floor_ = v[best] - 1e-12 * fabs(v[best]);
    for (uint32_t i = 0; i < best; ++i)
        if (v[i] >= floor_) return (int)i;
    return (int)best;
}

extern "C" int vgan_hc_posterior(vgan_hc_ctx *c, const double *final_vec, const char *predicted, char *clade_buf,
                                 int64_t clade_cap, double *conf, int32_t conf_cap) {
    if (!c || !final_vec || !predicted || !clade_buf || !conf) return fail(VGAN_EINVAL, "vgan_hc_posterior: null argument");
    if (c->path_names.size() != c->P) return fail(VGAN_ESTATE, "vgan_hc_posterior: graph has %zu path names for %u paths", c->path_names.size(), c->P);
    auto pi = c->path_index.find(predicted);
    if (pi == c->path_index.end()) return fail(VGAN_EINVAL, "vgan_hc_posterior: '%s' is not a path name", predicted);
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if (c->post_predicted != predicted || c->post_ns == 0) {
        // src/get_posterior.cpp:94-123: the predicted haplotype, then each ancestor with its strict descendants
        std::vector<std::string> parent_vec;
        auto pv = c->parents.find(predicted);
        if (pv != c->parents.end()) parent_vec = pv->second;
        std::vector<std::string> clades{predicted};
        // One index list per record, in the order the reference builds all_top (:51-76): per recursion level the members of
        // that level's child set in path order; a fresh set per level, so a path reachable at two depths is listed twice.
        if (c->by_name.empty()) // graph_paths may repeat a name: every index counts (:60-66)
            for (uint32_t p = 0; p < c->P; ++p) c->by_name[c->path_names[p]].push_back(p);
        const auto &by_name = c->by_name;
        std::vector<uint32_t> off{0}, idx{(uint32_t)pi->second};
        off.push_back(1);
        constexpr size_t kMaxList = (size_t)1 << 27; // a cyclic children.txt recurses without end in the reference
        for (size_t j = 0; j < parent_vec.size(); ++j) {
            const bool emit = j == 0 || parent_vec[j] != parent_vec[j - 1]; // :110,117 (Q9: j = 0 always emitted)
            if (!emit) continue; // its all_top is computed and dropped by the reference
            clades.push_back(parent_vec[j]);
            std::set<std::string> preds{parent_vec[j]};
            for (int depth = 0; !preds.empty() && depth <= 100000; ++depth) { // get_children(), :36-49
                std::set<std::string> child_set;
                for (const std::string &p : preds) {
                    auto ch = c->children.find(p);
                    if (ch == c->children.end()) continue; // the reference dereferences end() here; defined as "no children"
                    child_set.insert(ch->second.begin(), ch->second.end());
                }
                const size_t level0 = idx.size();
                for (const std::string &k : child_set) {
                    auto ki = by_name.find(k);
                    if (ki != by_name.end()) idx.insert(idx.end(), ki->second.begin(), ki->second.end());
                }
                std::sort(idx.begin() + level0, idx.end()); // path order within the level
                if (idx.size() > kMaxList) return fail(VGAN_ERANGE, "vgan_hc_posterior: children of '%s' do not end (cycle?)", parent_vec[j].c_str());
                preds.swap(child_set);
            }
            off.push_back((uint32_t)idx.size());
        }
        c->post_ns = 0; // (nothing is cached while the device copy is being replaced)
        if ((rc = c->lists.reserve(off.size() + idx.size()))) return rc;
        HIPCHK(hipMemcpyAsync(c->lists.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->lists.p + off.size(), idx.data(), idx.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream)); // (off / idx are locals)
        c->post_clades.clear();
        for (auto &s : clades) c->post_clades += s + "\n";
        c->post_predicted = predicted;
        c->post_n_off = (uint32_t)off.size();
        c->post_ns = (uint32_t)off.size() - 1;
    }
    const uint32_t ns = c->post_ns;
    if ((int32_t)ns > conf_cap) return fail(VGAN_ERANGE, "vgan_hc_posterior: conf_cap too small (%u records)", ns);
    if ((int64_t)c->post_clades.size() + 1 > clade_cap) return fail(VGAN_ERANGE, "vgan_hc_posterior: clade buffer too small");
    if ((rc = c->conf.reserve(ns))) return rc;
    // (the vector the caller holds may be a reduce over several contexts: it is sent, not assumed to be this context's own)
    HIPCHK(hipMemcpyAsync(c->final_vec.p, final_vec, (size_t)c->P * 8, hipMemcpyHostToDevice, c->stream));
    launch_hc_posterior(c->final_vec.p, c->P, c->lists.p, c->lists.p + c->post_n_off, ns, c->conf.p, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(conf, c->conf.p, (size_t)ns * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(clade_buf, c->post_clades.c_str(), c->post_clades.size() + 1);
    return (int)ns;
}
