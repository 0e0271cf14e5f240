// euka's front half on the device (reference: src/readGAM_Euka.h:67-216 -- reconstruct_graph_sequence
// (vgan_utils.h:6-79) and what the lambda reads off the Alignment -- through csrc/host/euka_host.cpp: euka_flatten_range): the arrays a
// vgan_gamdev parse left in HBM -> a vgan_euka_batch in HBM, for the reads whose edits are all matches or substitutions on known nodes
// (the one-walk form of csrc/host/flatten.cpp: reconstruct_matches_only); every other read -- indels, soft clips, reads the reference
// would index out of bounds on -- is left to the host (host_mask), which decides and reports as it always did.  Byte work: what this
// writes is, array for array, vgan_euka_flatten's batch of the same reads (tests/test_euka_pipe_gpu.py); the kernels are
// hc_flatten_kernels.hip's walk with euka's outputs (two byte strings padded to the longer, the mappings' node ids, the read's scalars).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "euka_device.h"
#include "gam_device.h"
#include "gam_object.h"
#include "wave_scan.h"
#include "host/common.h"
#include "vgan_gpu.h"

using namespace vgan;

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? VGAN_ENOMEM : VGAN_ENODEV, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace vgan {
namespace edf {

enum : uint8_t { EDF_DEVICE = 0, EDF_HOST = 1 };
struct EdfGraph {
    const int64_t *node_seq_off;
    const uint8_t *node_seq;
    int64_t min_id, max_id;
};
struct EdfCounters {
    unsigned int n_dev;
};

__device__ __forceinline__ uint8_t edf_comp(uint8_t c) { // csrc/host/flatten.cpp: comp()
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

// A wave per read, lanes over its mappings: reconstruct_matches_only()'s conditions and euka_flatten_range()'s
// (Lseq within 15..1000: subDeamDiNuc[Lseq], damage.h:42-43; 16-bit lengths and counts).  info = {|graph_seq|, |read_seq|, mappings,
// quality bytes}; key = the first mapping's node id (vgan_euka_flatten orders its batch by it).
__global__ __launch_bounds__(256) void euka_df_classify_kernel(GamdevSlice s, uint32_t n_reads, EdfGraph g, uint8_t *__restrict__ flag, uint32_t *__restrict__ key,
                                                               uint4 *__restrict__ info, EdfCounters *__restrict__ ctr) {
    const uint32_t lane = threadIdx.x & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (scalar: the read's offsets are scalar loads)
    uint32_t c_dev = 0;
    for (uint32_t r = blockIdx.x * 4u + wv; r < n_reads; r += gridDim.x * 4u) {
        const int64_t m0 = s.map_off[r], m1 = s.map_off[r + 1];
        const int64_t q_len = (int64_t)s.qual_off[r + 1] - (int64_t)s.qual_off[r];
        const uint32_t nm = (uint32_t)(m1 - m0), lseq = s.seq_len[r];
        bool ok = m1 > m0 && m1 - m0 <= 0xFFFF && q_len <= 0xFFFF && lseq >= 15u && lseq <= 1000u;
        uint32_t gn = 0, an = 0;
        if (ok) {
            bool bad = false;
            for (uint32_t mi = lane; mi < nm; mi += 64u) {
                const int64_t m = m0 + mi;
                const int64_t id = s.m_node[m];
                if (id < g.min_id || id > g.max_id) {
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
                    if (gn > 0x20000u || an > 0x20000u) { // (a read beyond 16-bit lengths is not this kernel's; the sums must not wrap either)
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
            }
            ok = ok && gn <= 65535u && an <= 65535u;
        }
        if (lane == 0) {
            flag[r] = ok ? EDF_DEVICE : EDF_HOST;
            key[r] = ok ? s.m_node[m0] : 0xFFFFFFFFu;
            info[r] = uint4{gn, an, nm, (uint32_t)q_len};
            c_dev += ok ? 1u : 0u;
        }
    }
    if (lane == 0 && c_dev) atomicAdd(&ctr->n_dev, c_dev);
}

// sizes of the taken reads in sorted order (taken reads come first)
__global__ __launch_bounds__(256) void euka_df_gather_kernel(const uint32_t *__restrict__ order, const uint4 *__restrict__ info, uint32_t n_dev, uint32_t *__restrict__ cols,
                                                             uint32_t *__restrict__ quals, uint32_t *__restrict__ maps) {
    const uint32_t o = blockIdx.x * 256u + threadIdx.x;
    if (o > n_dev) return;
    if (o == n_dev) { // (the scans' last input: their output there is the total)
        cols[o] = quals[o] = maps[o] = 0;
        return;
    }
    const uint4 v = info[order[o]];
    cols[o] = max(v.x, v.y);
    quals[o] = v.w;
    maps[o] = v.z;
}

struct EdfOut {
    uint32_t *read_col_off, *read_qual_off, *read_map_off, *read_src, *map_node;
    uint16_t *read_gseq_len, *read_rseq_len, *read_seq_len;
    int32_t *read_mapq;
    uint8_t *read_rev, *graph_seq, *read_seq, *qual;
};

// A wave per taken read, in sorted order: lanes over mappings (wave scans of the per-mapping totals say where each one's bases go), then
// over the bytes that are copied as they are.
__global__ __launch_bounds__(256) void euka_df_write_kernel(GamdevSlice s, EdfGraph g, const uint32_t *__restrict__ order, const uint32_t *__restrict__ coff,
                                                            const uint32_t *__restrict__ qoff, const uint32_t *__restrict__ moff, uint32_t n_dev, uint32_t src_base,
                                                            EdfOut out) {
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (uint32_t o = blockIdx.x * 4u + wave; o <= n_dev; o += gridDim.x * 4u) {
        if (lane == 0) {
            out.read_col_off[o] = coff[o];
            out.read_qual_off[o] = qoff[o];
            out.read_map_off[o] = moff[o];
        }
        if (o == n_dev) break;
        const uint32_t r = order[o];
        const int64_t m0 = s.map_off[r], m1 = s.map_off[r + 1];
        const uint32_t nm = (uint32_t)(m1 - m0);
        const uint32_t c0 = coff[o], q0 = qoff[o], mp0 = moff[o], region = coff[o + 1] - c0, nq = qoff[o + 1] - q0;
        uint32_t g_base = 0, a_base = 0;
        for (uint32_t mb = 0; mb < nm; mb += 64u) {
            const uint32_t mi = mb + lane;
            const bool on = mi < nm;
            uint32_t gn = 0, an = 0;
            int64_t id = 0, len = 0, off0 = 0;
            bool rev = false;
            if (on) {
                const int64_t m = m0 + mi;
                id = s.m_node[m];
                out.map_node[mp0 + mi] = (uint32_t)id;
                len = g.node_seq_off[id + 1] - g.node_seq_off[id];
                off0 = s.m_offset[m];
                rev = s.m_rev[m] != 0;
                int64_t off = off0;
                for (int64_t e = s.edit_off[m]; e < s.edit_off[m + 1]; ++e) {
                    const int64_t from = s.e_len[e], sl = (int64_t)s.e_seq_off[e + 1] - (int64_t)s.e_seq_off[e];
                    const uint32_t n = (uint32_t)min(from, len - off);
                    gn += n;
                    an += sl > 0 ? (uint32_t)sl : n;
                    off += from;
                }
            }
            const uint32_t gp = wave_incl_scan_u32(gn), ap = wave_incl_scan_u32(an); // inclusive prefix sums over the lanes (DPP)
            const uint32_t g_tot = wave_last_u32(gp), a_tot = wave_last_u32(ap);
            uint32_t gq = g_base + gp - gn, aq = a_base + ap - an; // this mapping's first places
            if (on) {
                const int64_t m = m0 + mi;
                const uint8_t *ns = g.node_seq + g.node_seq_off[id];
                int64_t off = off0;
                for (int64_t e = s.edit_off[m]; e < s.edit_off[m + 1]; ++e) {
                    const int64_t from = s.e_len[e], sl = (int64_t)s.e_seq_off[e + 1] - (int64_t)s.e_seq_off[e];
                    const uint32_t n = (uint32_t)min(from, len - off);
                    for (uint32_t k = 0; k < n; ++k) {
                        const uint8_t b = rev ? edf_comp(ns[len - 1 - (off + k)]) : ns[off + k];
                        out.graph_seq[c0 + gq + k] = b;
                        if (sl <= 0) out.read_seq[c0 + aq + k] = b;
                    }
                    if (sl > 0) {
                        const uint8_t *es = s.e_seq + s.e_seq_off[e];
                        for (int64_t k = 0; k < sl; ++k) out.read_seq[c0 + aq + k] = es[k];
                    }
                    gq += n;
                    aq += sl > 0 ? (uint32_t)sl : n;
                    off += from;
                }
            }
            g_base += g_tot;
            a_base += a_tot;
        }
        // the shorter of the two strings is padded with zero bytes to the longer (euka_flatten_range)
        for (uint32_t c = g_base + lane; c < region; c += 64u) out.graph_seq[c0 + c] = 0;
        for (uint32_t c = a_base + lane; c < region; c += 64u) out.read_seq[c0 + c] = 0;
        const uint8_t *q = s.qual + s.qual_off[r];
        for (uint32_t i = lane; i < nq; i += 64u) out.qual[q0 + i] = q[i];
        if (lane == 0) {
            out.read_gseq_len[o] = (uint16_t)g_base;
            out.read_rseq_len[o] = (uint16_t)a_base;
            out.read_seq_len[o] = (uint16_t)s.seq_len[r];
            out.read_mapq[o] = s.mapq[r];
            out.read_rev[o] = s.m_rev[m0];
            out.read_src[o] = src_base + r;
        }
    }
}

} // namespace edf
} // namespace vgan

using namespace vgan::edf;

namespace {
template <class T> struct EBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap && p) return VGAN_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = n + std::min<size_t>(n / 4, ((size_t)16 << 20) / sizeof(T)) + 256; // (slack for the next piece to fit)
        HIPCHK(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        static const bool poison = getenv("VGAN_POISON_ALLOCS") != nullptr; // (test aid, as csrc/gam_object.h: GBuf)
        if (poison) {
            HIPCHK(hipMemset(p, 0xA5, want * sizeof(T)));
            HIPCHK(hipDeviceSynchronize());
        }
        return VGAN_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
} // namespace

struct vgan_euka_devflat {
    int device = 0;
    const vgan_euka_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    EdfGraph g{};
    EBuf<int64_t> node_seq_off;
    EBuf<uint8_t> node_seq;
    EBuf<uint8_t> flag;
    EBuf<uint32_t> key, key_out, val, val_out, cols, quals, maps, coff, qoff, moff;
    EBuf<uint4> info;
    EBuf<EdfCounters> ctr;
    EBuf<uint64_t> tot64;
    EBuf<uint8_t> cub_tmp;
    // the batch
    EBuf<uint32_t> read_col_off, read_qual_off, read_map_off, read_src, map_node;
    EBuf<uint16_t> read_gseq_len, read_rseq_len, read_seq_len;
    EBuf<int32_t> read_mapq;
    EBuf<uint8_t> read_rev, graph_seq, read_seq, qual;
    size_t iota_on_dev = 0;
    std::vector<uint32_t> h_src;
    std::vector<uint16_t> h_len;
    size_t device_bytes() const {
        size_t b = node_seq_off.cap * 8 + node_seq.cap + flag.cap + info.cap * 16 + ctr.cap * sizeof(EdfCounters) + cub_tmp.cap + read_rev.cap + graph_seq.cap + read_seq.cap + qual.cap +
                   read_mapq.cap * 4 + (read_gseq_len.cap + read_rseq_len.cap + read_seq_len.cap) * 2;
        for (auto *x : {&key, &key_out, &val, &val_out, &cols, &quals, &maps, &coff, &qoff, &moff, &read_col_off, &read_qual_off, &read_map_off, &read_src, &map_node}) b += x->cap * 4;
        return b;
    }
    void release() {
        node_seq_off.release(), node_seq.release(), flag.release(), info.release(), ctr.release(), cub_tmp.release(), tot64.release();
        for (auto *x : {&key, &key_out, &val, &val_out, &cols, &quals, &maps, &coff, &qoff, &moff, &read_col_off, &read_qual_off, &read_map_off, &read_src, &map_node}) x->release();
        read_gseq_len.release(), read_rseq_len.release(), read_seq_len.release(), read_mapq.release();
        read_rev.release(), graph_seq.release(), read_seq.release(), qual.release();
    }
};

size_t vgan::euka_devflat_device_bytes(const vgan_euka_devflat *f) { return f ? f->device_bytes() : 0; }

extern "C" int vgan_euka_devflat_create(vgan_euka_ctx *c, const vgan_graph *graph, vgan_euka_devflat **out) {
    if (!c || !graph || !out) return fail(VGAN_EINVAL, "vgan_euka_devflat_create: null argument");
    const EukaCtxInfo ci = euka_ctx_info(c);
    HIPCHK(hipSetDevice(ci.device));
    auto f = new vgan_euka_devflat();
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
    if ((rc = f->node_seq_off.reserve(n_off)) || (rc = f->node_seq.reserve(n_seq + 1)) || (rc = f->ctr.reserve(1))) return bail(rc);
    if (hipMemcpy(f->node_seq_off.p, graph->node_seq_off.data(), n_off * 8, hipMemcpyHostToDevice) != hipSuccess ||
        (n_seq && hipMemcpy(f->node_seq.p, graph->node_seq.data(), n_seq, hipMemcpyHostToDevice) != hipSuccess))
        return bail(fail(VGAN_ENODEV, "vgan_euka_devflat_create: upload failed"));
    f->g.node_seq_off = f->node_seq_off.p;
    f->g.node_seq = f->node_seq.p;
    f->g.min_id = graph->min_id;
    f->g.max_id = graph->max_id;
    *out = f;
    return VGAN_OK;
}

extern "C" void vgan_euka_devflat_free(vgan_euka_devflat *f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    f->release();
    delete f;
}

extern "C" int vgan_euka_devflat_run_gamdev(vgan_euka_devflat *f, const vgan_gamdev *gd, uint32_t base, vgan_euka_batch *out, uint8_t *host_mask,
                                            vgan_euka_flatten_stats *stats) {
    if (!f || !gd || !out || !host_mask) return fail(VGAN_EINVAL, "vgan_euka_devflat_run_gamdev: null argument");
    memset(out, 0, sizeof *out);
    if (stats) memset(stats, 0, sizeof *stats);
    GamdevSlice gs{};
    if (!gamdev_slice(gd, &gs)) return fail(VGAN_ESTATE, "vgan_euka_devflat_run_gamdev: the front end holds no parse");
    if (gs.n_reads == 0) return VGAN_OK;
    if (gs.device != f->device) return fail(VGAN_EINVAL, "vgan_euka_devflat_run_gamdev: the parse lives on another device");
    if (gs.n_reads > 0x7FFFFFF0ull || (uint64_t)base + gs.n_reads > 0xFFFFFFF0ull) return fail(VGAN_ERANGE, "vgan_euka_devflat_run_gamdev: too many reads in one parse");
    HIPCHK(hipSetDevice(f->device));
    {
        const hipStream_t now = euka_ctx_info(f->ctx).stream; // (the read kernel reads this object's output on the context's stream)
        if (now != f->stream) {
            if (f->stream) HIPCHK(hipStreamSynchronize(f->stream));
            f->stream = now;
        }
    }
    hipStream_t st = f->stream;
    const uint32_t R = (uint32_t)gs.n_reads;
    int rc;
    if ((rc = f->flag.reserve(R)) || (rc = f->key.reserve(R)) || (rc = f->key_out.reserve(R)) || (rc = f->val_out.reserve(R)) || (rc = f->info.reserve(R)) ||
        (rc = f->cols.reserve(R + 1)) || (rc = f->quals.reserve(R + 1)) || (rc = f->maps.reserve(R + 1)) || (rc = f->coff.reserve(R + 1)) || (rc = f->qoff.reserve(R + 1)) ||
        (rc = f->moff.reserve(R + 1)))
        return rc;
    HIPCHK(hipMemsetAsync(f->ctr.p, 0, sizeof(EdfCounters), st));
    hipLaunchKernelGGL(euka_df_classify_kernel, dim3(std::min<uint32_t>((R + 3) / 4, 2048u)), dim3(256), 0, st, gs, R, f->g, f->flag.p, f->key.p, f->info.p, f->ctr.p);
    HIPCHK(hipGetLastError());
    { // the taken reads in ascending order of their first node id, input order kept among equals (the others' key is 2^32 - 1)
        if (f->iota_on_dev < R) {
            const size_t want = (size_t)R + R / 4 + 1024;
            std::vector<uint32_t> iota(want);
            for (size_t i = 0; i < want; ++i) iota[i] = (uint32_t)i;
            if ((rc = f->val.reserve(want))) return rc;
            HIPCHK(hipMemcpy(f->val.p, iota.data(), want * 4, hipMemcpyHostToDevice));
            f->iota_on_dev = want;
        }
        size_t tmp = 0;
        if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, f->key.p, f->key_out.p, f->val.p, f->val_out.p, (int)R, 0, 32, st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_euka_devflat_run_gamdev: sort sizing failed");
        if ((rc = f->cub_tmp.reserve(tmp))) return rc;
        if (hipcub::DeviceRadixSort::SortPairs(f->cub_tmp.p, tmp, f->key.p, f->key_out.p, f->val.p, f->val_out.p, (int)R, 0, 32, st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_euka_devflat_run_gamdev: sort failed");
    }
    EdfCounters hc{};
    HIPCHK(hipMemcpyAsync(&hc, f->ctr.p, sizeof hc, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(host_mask, f->flag.p, R, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint32_t n_dev = hc.n_dev;
    if (stats) {
        stats->n_in = R;
        stats->n_out = n_dev;
    }
    if (n_dev == 0) return VGAN_OK;
    hipLaunchKernelGGL(euka_df_gather_kernel, dim3((n_dev + 1 + 255) / 256), dim3(256), 0, st, f->val_out.p, f->info.p, n_dev, f->cols.p, f->quals.p, f->maps.p);
    {
        size_t tmp = 0;
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, f->cols.p, f->coff.p, (int)(n_dev + 1), st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_euka_devflat_run_gamdev: scan sizing failed");
        if ((rc = f->cub_tmp.reserve(tmp))) return rc;
        if (hipcub::DeviceScan::ExclusiveSum(f->cub_tmp.p, tmp, f->cols.p, f->coff.p, (int)(n_dev + 1), st) != hipSuccess ||
            hipcub::DeviceScan::ExclusiveSum(f->cub_tmp.p, tmp, f->quals.p, f->qoff.p, (int)(n_dev + 1), st) != hipSuccess ||
            hipcub::DeviceScan::ExclusiveSum(f->cub_tmp.p, tmp, f->maps.p, f->moff.p, (int)(n_dev + 1), st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_euka_devflat_run_gamdev: scan failed");
    }
    // (mappings and quality bytes are subsets of the piece's bytes, fewer than 2^32: their 32-bit sums cannot wrap.  Columns are sums of
    // edit LENGTHS -- a five-byte edit can claim 65535 of them -- so their total is taken in 64 bits as well, and a piece whose columns
    // do not fit 32-bit offsets is refused)
    if ((rc = f->tot64.reserve(1))) return rc;
    {
        struct Widen {
            __host__ __device__ uint64_t operator()(uint32_t v) const { return v; }
        };
        hipcub::TransformInputIterator<uint64_t, Widen, const uint32_t *> it(f->cols.p, Widen());
        size_t tmp = 0;
        if (hipcub::DeviceReduce::Sum(nullptr, tmp, it, f->tot64.p, (int)n_dev, st) != hipSuccess) return fail(VGAN_ENODEV, "vgan_euka_devflat_run_gamdev: sum sizing failed");
        if ((rc = f->cub_tmp.reserve(tmp))) return rc;
        if (hipcub::DeviceReduce::Sum(f->cub_tmp.p, tmp, it, f->tot64.p, (int)n_dev, st) != hipSuccess) return fail(VGAN_ENODEV, "vgan_euka_devflat_run_gamdev: sum failed");
    }
    uint32_t tot[3] = {0, 0, 0};
    uint64_t cols64 = 0;
    HIPCHK(hipMemcpyAsync(&tot[0], f->coff.p + n_dev, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&tot[1], f->qoff.p + n_dev, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&tot[2], f->moff.p + n_dev, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&cols64, f->tot64.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    static const char *lim = getenv("VGAN_EUKA_DEVFLAT_MAX_COLS"); // (test aid: the refusal without four billion columns)
    if (cols64 > (lim ? strtoull(lim, nullptr, 10) : 0xFFFFFFF0ull))
        return fail(VGAN_ERANGE, "vgan_euka_devflat_run_gamdev: %llu alignment columns in one piece are beyond 32-bit offsets; parse fewer bytes at a time", (unsigned long long)cols64);
    const size_t nd = n_dev;
    if ((rc = f->read_col_off.reserve(nd + 1)) || (rc = f->read_qual_off.reserve(nd + 1)) || (rc = f->read_map_off.reserve(nd + 1)) || (rc = f->read_src.reserve(nd)) ||
        (rc = f->map_node.reserve((size_t)tot[2] + 1)) || (rc = f->read_gseq_len.reserve(nd)) || (rc = f->read_rseq_len.reserve(nd)) || (rc = f->read_seq_len.reserve(nd)) ||
        (rc = f->read_mapq.reserve(nd)) || (rc = f->read_rev.reserve(nd)) || (rc = f->graph_seq.reserve((size_t)tot[0] + 64)) || (rc = f->read_seq.reserve((size_t)tot[0] + 64)) ||
        (rc = f->qual.reserve((size_t)tot[1] + 64)))
        return rc;
    EdfOut o{f->read_col_off.p, f->read_qual_off.p, f->read_map_off.p, f->read_src.p, f->map_node.p, f->read_gseq_len.p, f->read_rseq_len.p, f->read_seq_len.p,
             f->read_mapq.p,    f->read_rev.p,       f->graph_seq.p,    f->read_seq.p, f->qual.p};
    hipLaunchKernelGGL(euka_df_write_kernel, dim3(std::min<uint32_t>((n_dev + 1 + 3) / 4, 16384u)), dim3(256), 0, st, gs, f->g, f->val_out.p, f->coff.p, f->qoff.p, f->moff.p, n_dev, base, o);
    HIPCHK(hipGetLastError());
    f->h_src.resize(n_dev);
    f->h_len.resize(n_dev);
    HIPCHK(hipMemcpyAsync(f->h_src.data(), f->read_src.p, nd * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(f->h_len.data(), f->read_seq_len.p, nd * 2, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    out->n_reads = n_dev;
    out->n_cols = tot[0];
    out->n_qual = tot[1];
    out->n_maps = tot[2];
    out->read_col_off = f->read_col_off.p;
    out->read_qual_off = f->read_qual_off.p;
    out->read_map_off = f->read_map_off.p;
    out->read_gseq_len = f->read_gseq_len.p;
    out->read_rseq_len = f->read_rseq_len.p;
    out->read_seq_len = f->read_seq_len.p;
    out->read_mapq = f->read_mapq.p;
    out->read_rev = f->read_rev.p;
    out->read_src = f->read_src.p;
    out->map_node = f->map_node.p;
    out->graph_seq = f->graph_seq.p;
    out->read_seq = f->read_seq.p;
    out->qual = f->qual.p;
    out->on_device = 1;
    return VGAN_OK;
}

// the batch's read_src and read_seq_len on the host (valid until the next run)
extern "C" int vgan_euka_devflat_host_arrays(const vgan_euka_devflat *f, const uint32_t **read_src, const uint16_t **read_seq_len) {
    if (!f) return fail(VGAN_EINVAL, "vgan_euka_devflat_host_arrays: null argument");
    if (read_src) *read_src = f->h_src.data();
    if (read_seq_len) *read_seq_len = f->h_len.data();
    return VGAN_OK;
}

// (test aid) a device batch's arrays copied into caller arrays sized as the batch says (host: the pointers of *host, any may be NULL)
extern "C" int vgan_euka_batch_download(const vgan_euka_batch *dev, const vgan_euka_batch *host) {
    if (!dev || !host || !dev->on_device) return fail(VGAN_EINVAL, "vgan_euka_batch_download: a device batch and host arrays are needed");
    const size_t R = dev->n_reads;
#define DL(name, count)                                                                                                                         \
    if (host->name && dev->name && (count) && hipMemcpy((void *)host->name, dev->name, (count) * sizeof(*dev->name), hipMemcpyDeviceToHost) != hipSuccess) \
        return fail(VGAN_ENODEV, "vgan_euka_batch_download: copy failed");
    DL(read_col_off, R + 1)
    DL(read_qual_off, R + 1)
    DL(read_map_off, R + 1)
    DL(read_gseq_len, R)
    DL(read_rseq_len, R)
    DL(read_seq_len, R)
    DL(read_mapq, R)
    DL(read_rev, R)
    DL(read_src, R)
    DL(map_node, (size_t)dev->n_maps)
    DL(graph_seq, (size_t)dev->n_cols)
    DL(read_seq, (size_t)dev->n_cols)
    DL(qual, (size_t)dev->n_qual)
#undef DL
    return VGAN_OK;
}
#include "module_anchor.h"
const void *vgan::anchor_euka_flatten() { return (const void *)&vgan::edf::euka_df_gather_kernel; }
