// soibean's front half on the device (reference: src/getLCAfromGAM.h:92-186,537-544 -- reconstruct_graph_sequence
// (vgan_utils.h:6-79), then the edit-level slicing with the baseIX walk -- through csrc/host/sb_host.cpp: sb_flatten_range): the arrays a
// vgan_gamdev parse left in HBM -> the rows of a vgan_sb_batch in HBM, APPENDED to the batch the object holds (analyse_GAM runs once over
// all the reads of a context: vgan_sb_precompute), for the reads whose edits are all matches or substitutions on known nodes (the one-walk
// form of csrc/host/flatten.cpp: reconstruct_matches_only) and whose slices stay inside the strings whatever the walk does -- see
// sb_df_classify_kernel; every other read is left to the host (host_mask), which decides and reports as it always did and whose batch is
// appended behind (vgan_sb_devflat_append_host).  Byte / index work: what this writes is, array for array, vgan_sb_flatten's batch of
// the same reads in the same order (tests/test_sb_pipe_gpu.py).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "gam_device.h"
#include "gam_object.h"
#include "wave_scan.h"
#include "host/common.h"
#include "sb_device.h"
#include "vgan_gpu.h"

using namespace vgan;

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? VGAN_ENOMEM : VGAN_ENODEV, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace vgan {
namespace sdf {

enum : uint8_t { SDF_DEVICE = 0, SDF_HOST = 1 };
struct SdfGraph {
    const int64_t *node_seq_off;
    const uint8_t *node_seq;
    int64_t min_id, max_id;
};

__device__ __forceinline__ uint8_t sdf_comp(uint8_t c) { // csrc/host/flatten.cpp: comp()
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'a': return 't';
    case 'c': return 'g';
    case 'g': return 'c';
    case 't': return 'a';
    default: return 'N';
    }
}

// A wave per read, lanes over its mappings.  The device takes a read when
//   - reconstruct_matches_only() would take it (every edit a match or a substitution, known nodes, offsets inside their nodes), the node
//     ids positive (sb_flatten_range: node <= 0 is a read the reference throws on);
//   - G = |graph_seq| lies within 15..1000 (subDeamDiNuc[Lseq], getLCAfromGAM.h:109), A = |read_seq| fits 16 bits and A >= G;
//   - 1 <= |sequence| <= 65535, and on the reverse strand |sequence| <= G.
// With these the walk of sb_flatten_range cannot go out of range: forward, baseIX_i = sum of the sizes before edit i <= G - size_i and
// the slice [baseIX_i, + size_i) lies inside both strings; reverse, baseIX_0 = |sequence| - 1 < G and every later baseIX and startIndex
// is smaller still.  Anything else is the host's to judge.  info = {G, A, edits, quality bytes}.
__global__ __launch_bounds__(256) void sb_df_classify_kernel(GamdevSlice s, uint32_t n_reads, SdfGraph g, uint8_t *__restrict__ flag, uint4 *__restrict__ info,
                                                             uint32_t *__restrict__ take, uint32_t *__restrict__ cols, uint32_t *__restrict__ quals,
                                                             uint32_t *__restrict__ segs) {
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t lo_id = g.min_id > 1 ? g.min_id : 1;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (scalar: the read's offsets are scalar loads)
    for (uint32_t r = blockIdx.x * 4u + wv; r <= n_reads; r += gridDim.x * 4u) {
        if (r == n_reads) { // (the scans' last input: their output there is the total)
            if (lane == 0) take[r] = cols[r] = quals[r] = segs[r] = 0;
            break;
        }
        const int64_t m0 = s.map_off[r], m1 = s.map_off[r + 1];
        const int64_t q_len = (int64_t)s.qual_off[r + 1] - (int64_t)s.qual_off[r];
        const uint32_t nm = (uint32_t)(m1 - m0), lseq = s.seq_len[r];
        bool ok = m1 > m0 && lseq >= 1u && lseq <= 65535u;
        uint32_t gn = 0, an = 0, ne = 0;
        if (ok) {
            bool bad = false;
            for (uint32_t mi = lane; mi < nm; mi += 64u) {
                const int64_t m = m0 + mi;
                const int64_t id = s.m_node[m];
                if (id < lo_id || id > g.max_id) {
                    bad = true;
                    continue;
                }
                const int64_t len = g.node_seq_off[id + 1] - g.node_seq_off[id];
                int64_t off = s.m_offset[m];
                if (off == (int64_t)INT32_MIN) { // (the offset did not fit 32 bits: the general walk's read)
                    bad = true;
                    continue;
                }
                for (int64_t e = s.edit_off[m]; e < s.edit_off[m + 1]; ++e) {
                    const int64_t from = s.e_len[e];
                    if (from < 0 || off > len || off < 0) { // (an edit that is not a match or a substitution: -1)
                        bad = true;
                        break;
                    }
                    const int64_t sl = (int64_t)s.e_seq_off[e + 1] - (int64_t)s.e_seq_off[e];
                    const int64_t n = min(from, len - off);
                    gn += (uint32_t)min<int64_t>(n, 0x10000);
                    an += (uint32_t)min<int64_t>(sl > 0 ? sl : n, 0x10000);
                    ne += 1u;
                    if (gn > 0x20000u || an > 0x20000u) { // (the sums must not wrap)
                        bad = true;
                        break;
                    }
                    off += from;
                }
            }
            ok = __builtin_amdgcn_ballot_w64(bad) == 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                gn += __shfl_xor(gn, o, 64);
                an += __shfl_xor(an, o, 64);
                ne += __shfl_xor(ne, o, 64);
            }
            ok = ok && gn >= 15u && gn <= 1000u && an <= 65535u && an >= gn && (s.m_rev[m0] == 0 || lseq <= gn);
        }
        if (lane == 0) {
            flag[r] = ok ? SDF_DEVICE : SDF_HOST;
            info[r] = uint4{gn, an, ne, (uint32_t)q_len};
            take[r] = ok ? 1u : 0u;
            cols[r] = ok ? max(gn, an) : 0u;
            quals[r] = ok ? (uint32_t)q_len : 0u;
            segs[r] = ok ? ne : 0u;
        }
    }
}

struct SdfOut {
    uint32_t *read_seg_off, *read_col_off, *read_qual_off, *read_src, *seg_node;
    uint16_t *read_gseq_len, *read_rseq_len, *seg_col, *seg_len, *seg_base_ix;
    uint8_t *read_rev, *graph_seq, *read_seq, *qual;
};
struct SdfBase { // what the batch holds already: the piece's rows go behind
    uint32_t reads, segs, cols, quals;
};

// A wave per taken read, in input order: lanes over mappings (wave scans of the per-mapping totals say where each one's bases and segments
// go), then over the bytes that are copied as they are.  One segment per EDIT, its node the node of the MAPPING of the same index (or 0
// beyond the mappings: getLCAfromGAM.h:156-160), forward: col = baseIX = the bases before it; reverse: baseIX_i = max(|sequence| - 1 -
// sum_{j<i}(size_j + 1), 0), col = max(|sequence| - 1 - sum_{j<=i}(size_j + 1), 0), len clipped at the end of graph_seq (:179-186,537-544).
__global__ __launch_bounds__(256) void sb_df_write_kernel(GamdevSlice s, SdfGraph g, const uint8_t *__restrict__ flag, const uint4 *__restrict__ info,
                                                          const uint32_t *__restrict__ tpos, const uint32_t *__restrict__ coff, const uint32_t *__restrict__ qoff,
                                                          const uint32_t *__restrict__ soff, uint32_t n_reads, uint32_t src_base, SdfBase base, SdfOut out) {
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (uint32_t r = blockIdx.x * 4u + wave; r < n_reads; r += gridDim.x * 4u) {
        if (flag[r] != SDF_DEVICE) continue;
        const uint32_t o = base.reads + tpos[r], c0 = base.cols + coff[r], q0 = base.quals + qoff[r], sg0 = base.segs + soff[r];
        const uint4 inf = info[r];
        const uint32_t G = inf.x, A = inf.y, nE = inf.z, nq = inf.w, region = max(G, A);
        const int64_t m0 = s.map_off[r], m1 = s.map_off[r + 1];
        const uint32_t nm = (uint32_t)(m1 - m0);
        const bool rrev = s.m_rev[m0] != 0;
        const int64_t S = (int64_t)s.seq_len[r] - 1;
        uint32_t g_base = 0, a_base = 0, e_base = 0;
        for (uint32_t mb = 0; mb < nm; mb += 64u) {
            const uint32_t mi = mb + lane;
            const bool on = mi < nm;
            uint32_t gn = 0, an = 0, ne = 0;
            int64_t id = 0, len = 0, off0 = 0;
            bool rev = false;
            if (on) {
                const int64_t m = m0 + mi;
                id = s.m_node[m];
                len = g.node_seq_off[id + 1] - g.node_seq_off[id];
                off0 = s.m_offset[m];
                rev = s.m_rev[m] != 0;
                int64_t off = off0;
                for (int64_t e = s.edit_off[m]; e < s.edit_off[m + 1]; ++e) {
                    const int64_t from = s.e_len[e], sl = (int64_t)s.e_seq_off[e + 1] - (int64_t)s.e_seq_off[e];
                    const uint32_t n = (uint32_t)min(from, len - off);
                    gn += n;
                    an += sl > 0 ? (uint32_t)sl : n;
                    ne += 1u;
                    off += from;
                }
            }
            const uint32_t gp = wave_incl_scan_u32(gn), ap = wave_incl_scan_u32(an), ep = wave_incl_scan_u32(ne); // inclusive prefix sums over the lanes (DPP)
            const uint32_t g_tot = wave_last_u32(gp), a_tot = wave_last_u32(ap), e_tot = wave_last_u32(ep);
            uint32_t gq = g_base + gp - gn, aq = a_base + ap - an, ei = e_base + ep - ne; // this mapping's first places
            if (on) {
                const int64_t m = m0 + mi;
                const uint8_t *ns = g.node_seq + g.node_seq_off[id];
                int64_t off = off0;
                for (int64_t e = s.edit_off[m]; e < s.edit_off[m + 1]; ++e) {
                    const int64_t from = s.e_len[e], sl = (int64_t)s.e_seq_off[e + 1] - (int64_t)s.e_seq_off[e];
                    const uint32_t n = (uint32_t)min(from, len - off);
                    for (uint32_t k = 0; k < n; ++k) {
                        const uint8_t b = rev ? sdf_comp(ns[len - 1 - (off + k)]) : ns[off + k];
                        out.graph_seq[c0 + gq + k] = b;
                        if (sl <= 0) out.read_seq[c0 + aq + k] = b;
                    }
                    if (sl > 0) {
                        const uint8_t *es = s.e_seq + s.e_seq_off[e];
                        for (int64_t k = 0; k < sl; ++k) out.read_seq[c0 + aq + k] = es[k];
                    }
                    { // the segment of edit `ei`
                        int64_t bix, start, sl_len;
                        if (!rrev) {
                            bix = start = gq;
                            sl_len = n;
                        } else {
                            const int64_t b = S - ((int64_t)gq + (int64_t)ei);
                            bix = max<int64_t>(b, 0);
                            start = max<int64_t>(b - (int64_t)n - 1, 0);
                            sl_len = min<int64_t>(n, (int64_t)G - start);
                        }
                        out.seg_node[sg0 + ei] = ei < nm ? s.m_node[m0 + ei] : 0u;
                        out.seg_col[sg0 + ei] = (uint16_t)start;
                        out.seg_len[sg0 + ei] = (uint16_t)sl_len;
                        out.seg_base_ix[sg0 + ei] = (uint16_t)bix;
                    }
                    gq += n;
                    aq += sl > 0 ? (uint32_t)sl : n;
                    ei += 1u;
                    off += from;
                }
            }
            g_base += g_tot;
            a_base += a_tot;
            e_base += e_tot;
        }
        // the shorter of the two strings is padded with zero bytes to the longer (sb_flatten_range)
        for (uint32_t c = g_base + lane; c < region; c += 64u) out.graph_seq[c0 + c] = 0;
        for (uint32_t c = a_base + lane; c < region; c += 64u) out.read_seq[c0 + c] = 0;
        const uint8_t *q = s.qual + s.qual_off[r];
        for (uint32_t i = lane; i < nq; i += 64u) out.qual[q0 + i] = q[i];
        if (lane == 0) {
            out.read_seg_off[o + 1] = sg0 + nE;
            out.read_col_off[o + 1] = c0 + region;
            out.read_qual_off[o + 1] = q0 + nq;
            out.read_gseq_len[o] = (uint16_t)G;
            out.read_rseq_len[o] = (uint16_t)A;
            out.read_rev[o] = s.m_rev[m0];
            out.read_src[o] = src_base + r;
        }
    }
}

} // namespace sdf
} // namespace vgan

using namespace vgan::sdf;

namespace {
template <class T> struct SBuf { // scratch: contents need not survive a growth
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap && p) return VGAN_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = n + std::min<size_t>(n / 4, ((size_t)16 << 20) / sizeof(T)) + 256;
        HIPCHK(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        return VGAN_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
template <class T> struct Grow { // an array of the batch: the first `keep` elements survive a growth
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t keep, size_t want, size_t hint, hipStream_t st) {
        if (want <= cap && p) return VGAN_OK;
        const size_t ncap = std::max(want + want / 2 + 4096, hint);
        T *q = nullptr;
        HIPCHK(hipMalloc((void **)&q, ncap * sizeof(T)));
        static const bool poison = getenv("VGAN_POISON_ALLOCS") != nullptr; // (test aid, as csrc/gam_object.h: GBuf)
        if (poison) HIPCHK(hipMemsetAsync(q, 0xA5, ncap * sizeof(T), st));
        if (p && keep) HIPCHK(hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        if (p) (void)hipFree(p);
        p = q;
        cap = ncap;
        return VGAN_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
} // namespace

struct vgan_sb_devflat {
    int device = 0;
    const vgan_sb_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    SdfGraph g{};
    SBuf<int64_t> node_seq_off;
    SBuf<uint8_t> node_seq;
    SBuf<uint8_t> flag;
    SBuf<uint32_t> take, cols, quals, segs, tpos, coff, qoff, soff;
    SBuf<uint4> info;
    SBuf<uint64_t> tot64;
    SBuf<uint8_t> cub_tmp;
    // the batch so far
    Grow<uint32_t> read_seg_off, read_col_off, read_qual_off, read_src, seg_node;
    Grow<uint16_t> read_gseq_len, read_rseq_len, seg_col, seg_len, seg_base_ix;
    Grow<uint8_t> read_rev, graph_seq, read_seq, qual;
    uint64_t n_reads = 0, n_segs = 0, n_cols = 0, n_qual = 0;
    double hint_scale = 0; // the whole input over the first piece (vgan_sb_devflat_expect): the arrays are sized once
    size_t device_bytes() const {
        size_t b = node_seq_off.cap * 8 + node_seq.cap + flag.cap + info.cap * 16 + tot64.cap * 8 + cub_tmp.cap + read_rev.cap + graph_seq.cap + read_seq.cap + qual.cap;
        for (auto *x : {&take, &cols, &quals, &segs, &tpos, &coff, &qoff, &soff}) b += x->cap * 4;
        for (auto *x : {&read_seg_off, &read_col_off, &read_qual_off, &read_src, &seg_node}) b += x->cap * 4;
        for (auto *x : {&read_gseq_len, &read_rseq_len, &seg_col, &seg_len, &seg_base_ix}) b += x->cap * 2;
        return b;
    }
    void release() {
        node_seq_off.release(), node_seq.release(), flag.release(), info.release(), tot64.release(), cub_tmp.release();
        for (auto *x : {&take, &cols, &quals, &segs, &tpos, &coff, &qoff, &soff}) x->release();
        for (auto *x : {&read_seg_off, &read_col_off, &read_qual_off, &read_src, &seg_node}) x->release();
        for (auto *x : {&read_gseq_len, &read_rseq_len, &seg_col, &seg_len, &seg_base_ix}) x->release();
        read_rev.release(), graph_seq.release(), read_seq.release(), qual.release();
    }
    // room for nr / ns / nc / nq more reads, segments, columns, quality bytes (the rows so far survive)
    int grow(uint64_t nr, uint64_t ns, uint64_t nc, uint64_t nq) {
        const uint64_t R = n_reads + nr, S = n_segs + ns, C = n_cols + nc, Q = n_qual + nq;
        if (R > 0xFFFFFFF0ull || S > 0xFFFFFFF0ull || C > 0xFFFFFFF0ull || Q > 0xFFFFFFF0ull)
            return fail(VGAN_ERANGE, "vgan_sb_devflat: the batch exceeds 32-bit offsets (%llu reads, %llu segments, %llu columns); give the reads to more contexts",
                        (unsigned long long)R, (unsigned long long)S, (unsigned long long)C);
        auto hint = [&](uint64_t first) { return n_reads == 0 && hint_scale > 1 ? (size_t)std::min<double>((double)first * hint_scale, 4.2e9) : (size_t)0; };
        const size_t hr = hint(nr), hs = hint(ns), hc = hint(nc), hq = hint(nq);
        const bool fresh = read_seg_off.p == nullptr;
        int rc;
        hipStream_t st = stream;
        if ((rc = read_seg_off.ensure(n_reads + 1, R + 1, hr, st)) || (rc = read_col_off.ensure(n_reads + 1, R + 1, hr, st)) || (rc = read_qual_off.ensure(n_reads + 1, R + 1, hr, st)) ||
            (rc = read_src.ensure(n_reads, R, hr, st)) || (rc = read_gseq_len.ensure(n_reads, R, hr, st)) || (rc = read_rseq_len.ensure(n_reads, R, hr, st)) ||
            (rc = read_rev.ensure(n_reads, R, hr, st)) || (rc = seg_node.ensure(n_segs, S, hs, st)) || (rc = seg_col.ensure(n_segs, S, hs, st)) ||
            (rc = seg_len.ensure(n_segs, S, hs, st)) || (rc = seg_base_ix.ensure(n_segs, S, hs, st)) || (rc = graph_seq.ensure(n_cols, C + 64, hc, st)) ||
            (rc = read_seq.ensure(n_cols, C + 64, hc, st)) || (rc = qual.ensure(n_qual, Q + 64, hq, st)))
            return rc;
        if (fresh) {
            HIPCHK(hipMemsetAsync(read_seg_off.p, 0, 4, st));
            HIPCHK(hipMemsetAsync(read_col_off.p, 0, 4, st));
            HIPCHK(hipMemsetAsync(read_qual_off.p, 0, 4, st));
        }
        return VGAN_OK;
    }
};

size_t vgan::sb_devflat_device_bytes(const vgan_sb_devflat *f) { return f ? f->device_bytes() : 0; }

extern "C" int vgan_sb_devflat_create(vgan_sb_ctx *c, const vgan_graph *graph, vgan_sb_devflat **out) {
    if (!c || !graph || !out) return fail(VGAN_EINVAL, "vgan_sb_devflat_create: null argument");
    const SbCtxInfo ci = sb_ctx_info(c);
    HIPCHK(hipSetDevice(ci.device));
    auto f = new vgan_sb_devflat();
    f->device = ci.device;
    f->ctx = c;
    f->stream = ci.stream;
    int rc;
    auto bail = [&](int code) {
        f->release();
        delete f;
        return code;
    };
    const size_t n_off = graph->node_seq_off.size(), n_seq = graph->node_seq.size();
    if ((rc = f->node_seq_off.reserve(n_off)) || (rc = f->node_seq.reserve(n_seq + 1))) return bail(rc);
    if (hipMemcpy(f->node_seq_off.p, graph->node_seq_off.data(), n_off * 8, hipMemcpyHostToDevice) != hipSuccess ||
        (n_seq && hipMemcpy(f->node_seq.p, graph->node_seq.data(), n_seq, hipMemcpyHostToDevice) != hipSuccess))
        return bail(fail(VGAN_ENODEV, "vgan_sb_devflat_create: upload failed"));
    f->g.node_seq_off = f->node_seq_off.p;
    f->g.node_seq = f->node_seq.p;
    f->g.min_id = graph->min_id;
    f->g.max_id = graph->max_id;
    *out = f;
    return VGAN_OK;
}

extern "C" void vgan_sb_devflat_free(vgan_sb_devflat *f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    f->release();
    delete f;
}

// the first append sizes the arrays for `scale` times what it brings (the input's size over the first piece's): no growth later
extern "C" int vgan_sb_devflat_expect(vgan_sb_devflat *f, double scale) {
    if (!f) return fail(VGAN_EINVAL, "vgan_sb_devflat_expect: null argument");
    f->hint_scale = scale;
    return VGAN_OK;
}

extern "C" int vgan_sb_devflat_append_gamdev(vgan_sb_devflat *f, const vgan_gamdev *gd, uint32_t base, uint8_t *host_mask, vgan_sb_flatten_stats *stats) {
    if (!f || !gd || !host_mask) return fail(VGAN_EINVAL, "vgan_sb_devflat_append_gamdev: null argument");
    if (stats) memset(stats, 0, sizeof *stats);
    GamdevSlice gs{};
    if (!gamdev_slice(gd, &gs)) return fail(VGAN_ESTATE, "vgan_sb_devflat_append_gamdev: the front end holds no parse");
    if (gs.n_reads == 0) return VGAN_OK;
    if (gs.device != f->device) return fail(VGAN_EINVAL, "vgan_sb_devflat_append_gamdev: the parse lives on another device");
    if (gs.n_reads > 0x7FFFFFF0ull || (uint64_t)base + gs.n_reads > 0xFFFFFFF0ull) return fail(VGAN_ERANGE, "vgan_sb_devflat_append_gamdev: too many reads in one parse");
    HIPCHK(hipSetDevice(f->device));
    {
        const hipStream_t now = sb_ctx_info(f->ctx).stream; // (analyse_GAM reads this object's batch on the context's stream)
        if (now != f->stream) {
            if (f->stream) HIPCHK(hipStreamSynchronize(f->stream));
            f->stream = now;
        }
    }
    hipStream_t st = f->stream;
    const uint32_t R = (uint32_t)gs.n_reads;
    int rc;
    if ((rc = f->flag.reserve(R)) || (rc = f->info.reserve(R)) || (rc = f->take.reserve(R + 1)) || (rc = f->cols.reserve(R + 1)) || (rc = f->quals.reserve(R + 1)) ||
        (rc = f->segs.reserve(R + 1)) || (rc = f->tpos.reserve(R + 1)) || (rc = f->coff.reserve(R + 1)) || (rc = f->qoff.reserve(R + 1)) || (rc = f->soff.reserve(R + 1)) ||
        (rc = f->tot64.reserve(1)))
        return rc;
    hipLaunchKernelGGL(sb_df_classify_kernel, dim3(std::min<uint32_t>((R + 1 + 3) / 4, 8192u)), dim3(256), 0, st, gs, R, f->g, f->flag.p, f->info.p, f->take.p, f->cols.p, f->quals.p,
                       f->segs.p);
    HIPCHK(hipGetLastError());
    {
        size_t tmp = 0, tmp2 = 0;
        struct Widen {
            __host__ __device__ uint64_t operator()(uint32_t v) const { return v; }
        };
        hipcub::TransformInputIterator<uint64_t, Widen, const uint32_t *> it(f->cols.p, Widen());
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, f->cols.p, f->coff.p, (int)(R + 1), st) != hipSuccess ||
            hipcub::DeviceReduce::Sum(nullptr, tmp2, it, f->tot64.p, (int)R, st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_sb_devflat_append_gamdev: scan sizing failed");
        tmp = std::max(tmp, tmp2);
        if ((rc = f->cub_tmp.reserve(tmp))) return rc;
        // (segments and quality bytes are subsets of the piece's bytes, fewer than 2^32: their 32-bit sums cannot wrap.  Columns are sums
        // of edit LENGTHS, so their total is taken in 64 bits as well)
        if (hipcub::DeviceScan::ExclusiveSum(f->cub_tmp.p, tmp, f->take.p, f->tpos.p, (int)(R + 1), st) != hipSuccess ||
            hipcub::DeviceScan::ExclusiveSum(f->cub_tmp.p, tmp, f->cols.p, f->coff.p, (int)(R + 1), st) != hipSuccess ||
            hipcub::DeviceScan::ExclusiveSum(f->cub_tmp.p, tmp, f->quals.p, f->qoff.p, (int)(R + 1), st) != hipSuccess ||
            hipcub::DeviceScan::ExclusiveSum(f->cub_tmp.p, tmp, f->segs.p, f->soff.p, (int)(R + 1), st) != hipSuccess ||
            hipcub::DeviceReduce::Sum(f->cub_tmp.p, tmp, it, f->tot64.p, (int)R, st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_sb_devflat_append_gamdev: scan failed");
    }
    uint32_t tot[4] = {0, 0, 0, 0};
    uint64_t cols64 = 0;
    HIPCHK(hipMemcpyAsync(&tot[0], f->tpos.p + R, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&tot[1], f->soff.p + R, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&tot[2], f->coff.p + R, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&tot[3], f->qoff.p + R, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&cols64, f->tot64.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(host_mask, f->flag.p, R, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint32_t n_dev = tot[0];
    if (stats) {
        stats->n_in = R;
        stats->n_out = n_dev;
    }
    if (n_dev == 0) return VGAN_OK;
    static const char *lim = getenv("VGAN_SB_DEVFLAT_MAX_COLS"); // (test aid: the refusal without four billion columns)
    if (cols64 > (lim ? strtoull(lim, nullptr, 10) : 0xFFFFFFF0ull))
        return fail(VGAN_ERANGE, "vgan_sb_devflat_append_gamdev: %llu alignment columns in one piece are beyond 32-bit offsets; parse fewer bytes at a time", (unsigned long long)cols64);
    if ((rc = f->grow(n_dev, tot[1], tot[2], tot[3]))) return rc;
    SdfOut o{f->read_seg_off.p, f->read_col_off.p, f->read_qual_off.p, f->read_src.p, f->seg_node.p,  f->read_gseq_len.p, f->read_rseq_len.p,
             f->seg_col.p,      f->seg_len.p,      f->seg_base_ix.p,   f->read_rev.p, f->graph_seq.p, f->read_seq.p,      f->qual.p};
    const SdfBase bs{(uint32_t)f->n_reads, (uint32_t)f->n_segs, (uint32_t)f->n_cols, (uint32_t)f->n_qual};
    hipLaunchKernelGGL(sb_df_write_kernel, dim3(std::min<uint32_t>((R + 3) / 4, 16384u)), dim3(256), 0, st, gs, f->g, f->flag.p, f->info.p, f->tpos.p, f->coff.p, f->qoff.p, f->soff.p, R,
                       base, bs, o);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st)); // (the parse's arrays are the next piece's after this call)
    f->n_reads += n_dev;
    f->n_segs += tot[1];
    f->n_cols += tot[2];
    f->n_qual += tot[3];
    return VGAN_OK;
}

// a host batch (vgan_sb_flatten's, of the reads the device left) behind the rows so far; src_map (or null): read_src[i] becomes
// src_map[read_src[i]] -- the reads' places in the input, where the host batch counts within what it was given
extern "C" int vgan_sb_devflat_append_host(vgan_sb_devflat *f, const vgan_sb_batch *b, const uint32_t *src_map) {
    if (!f || !b || b->on_device) return fail(VGAN_EINVAL, "vgan_sb_devflat_append_host: a host batch is needed");
    const size_t R = b->n_reads, S = b->n_segments, C = (size_t)b->n_cols, Q = (size_t)b->n_qual;
    if (R == 0) return VGAN_OK;
    HIPCHK(hipSetDevice(f->device));
    int rc;
    if ((rc = f->grow(R, S, C, Q))) return rc;
    hipStream_t st = f->stream;
    std::vector<uint32_t> so(R), co(R), qo(R), src(R);
    for (size_t i = 0; i < R; ++i) {
        so[i] = b->read_seg_off[i + 1] + (uint32_t)f->n_segs;
        co[i] = b->read_col_off[i + 1] + (uint32_t)f->n_cols;
        qo[i] = b->read_qual_off[i + 1] + (uint32_t)f->n_qual;
        src[i] = src_map ? src_map[b->read_src[i]] : b->read_src[i];
    }
#define UP(dst, at, srcp, n)                                                                                                              \
    do {                                                                                                                                \
        if ((n) > 0) HIPCHK(hipMemcpyAsync((dst) + (at), (srcp), (n) * sizeof(*(srcp)), hipMemcpyHostToDevice, st));                        \
    } while (0)
    UP(f->read_seg_off.p, f->n_reads + 1, so.data(), R);
    UP(f->read_col_off.p, f->n_reads + 1, co.data(), R);
    UP(f->read_qual_off.p, f->n_reads + 1, qo.data(), R);
    UP(f->read_src.p, f->n_reads, src.data(), R);
    UP(f->read_gseq_len.p, f->n_reads, b->read_gseq_len, R);
    UP(f->read_rseq_len.p, f->n_reads, b->read_rseq_len, R);
    UP(f->read_rev.p, f->n_reads, b->read_rev, R);
    UP(f->seg_node.p, f->n_segs, b->seg_node, S);
    UP(f->seg_col.p, f->n_segs, b->seg_col, S);
    UP(f->seg_len.p, f->n_segs, b->seg_len, S);
    UP(f->seg_base_ix.p, f->n_segs, b->seg_base_ix, S);
    UP(f->graph_seq.p, f->n_cols, b->graph_seq, C);
    UP(f->read_seq.p, f->n_cols, b->read_seq, C);
    UP(f->qual.p, f->n_qual, b->qual, Q);
#undef UP
    HIPCHK(hipStreamSynchronize(st));
    f->n_reads += R;
    f->n_segs += S;
    f->n_cols += C;
    f->n_qual += Q;
    return VGAN_OK;
}

// the batch so far (device pointers: on_device = 1), valid until the next append or the object's end
extern "C" int vgan_sb_devflat_batch(const vgan_sb_devflat *f, vgan_sb_batch *out) {
    if (!f || !out) return fail(VGAN_EINVAL, "vgan_sb_devflat_batch: null argument");
    memset(out, 0, sizeof *out);
    out->n_reads = (uint32_t)f->n_reads;
    out->n_segments = (uint32_t)f->n_segs;
    out->n_cols = f->n_cols;
    out->n_qual = f->n_qual;
    out->read_seg_off = f->read_seg_off.p;
    out->read_col_off = f->read_col_off.p;
    out->read_qual_off = f->read_qual_off.p;
    out->read_gseq_len = f->read_gseq_len.p;
    out->read_rseq_len = f->read_rseq_len.p;
    out->read_rev = f->read_rev.p;
    out->read_src = f->read_src.p;
    out->seg_node = f->seg_node.p;
    out->seg_col = f->seg_col.p;
    out->seg_len = f->seg_len.p;
    out->seg_base_ix = f->seg_base_ix.p;
    out->graph_seq = f->graph_seq.p;
    out->read_seq = f->read_seq.p;
    out->qual = f->qual.p;
    out->on_device = 1;
    return VGAN_OK;
}

// (test aid, and the read indices of a run) a device batch's arrays copied into caller arrays sized as the batch says (host: the
// pointers of *host, any may be NULL)
extern "C" int vgan_sb_batch_download(const vgan_sb_batch *dev, const vgan_sb_batch *host) {
    if (!dev || !host || !dev->on_device) return fail(VGAN_EINVAL, "vgan_sb_batch_download: a device batch and host arrays are needed");
    const size_t R = dev->n_reads, S = dev->n_segments;
    if (R == 0) return VGAN_OK;
#define DL(name, count)                                                                                                                         \
    if (host->name && dev->name && (count) && hipMemcpy((void *)host->name, dev->name, (count) * sizeof(*dev->name), hipMemcpyDeviceToHost) != hipSuccess) \
        return fail(VGAN_ENODEV, "vgan_sb_batch_download: copy failed");
    DL(read_seg_off, R + 1)
    DL(read_col_off, R + 1)
    DL(read_qual_off, R + 1)
    DL(read_gseq_len, R)
    DL(read_rseq_len, R)
    DL(read_rev, R)
    DL(read_src, R)
    DL(seg_node, S)
    DL(seg_col, S)
    DL(seg_len, S)
    DL(seg_base_ix, S)
    DL(graph_seq, (size_t)dev->n_cols)
    DL(read_seq, (size_t)dev->n_cols)
    DL(qual, (size_t)dev->n_qual)
#undef DL
    return VGAN_OK;
}
#include "module_anchor.h"
const void *vgan::anchor_sb_flatten() { return (const void *)&vgan::sdf::sb_df_classify_kernel; }
