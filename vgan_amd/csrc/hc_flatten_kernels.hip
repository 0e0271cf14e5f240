// a1 on the device: reconstruct_graph_sequence + the slicing of update_likelihood + the packed layout, in one pass over the
// parser's arrays (reference: src/vgan_utils.h:6-79, src/update_likelihood.cpp:28-45, src/HaploCart.cpp:410).  The host keeps
// the general walk (csrc/host/flatten.cpp): this path takes the reads whose edits are all matches or substitutions
// (from_length == to_length) on known nodes and which satisfy the tile contract -- the common read -- and leaves every
// other read, flagged, to the host.  What it writes is the host flatten's packed batch of the same reads, word for word
// (tests/test_devflat_gpu.py): integer and byte work only.
//
//   hc_df_classify_kernel   a thread per read: the walk over mappings and edits without moving a byte -- may the device take
//                           it, how many columns / segments / quality bytes, its lowest node id
//   (hipcub)                stable radix sort of the taken reads by lowest node id, exclusive sums of the three sizes
//   hc_df_write_kernel      a wave per read: node bases (reverse complement for reverse mappings) and read bases into LDS,
//                           lanes over mappings; segment records, lanes over segments; column records, lanes over columns
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <vector>

#include "gam_device.h"
#include "wave_scan.h"
#include "gam_object.h"
#include "hc_device.h"
#include "host/common.h"
#include "vgan_gpu.h"

using namespace vgan;

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(VGAN_ENODEV, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace vgan {
namespace df {

constexpr uint32_t DF_COLS = HC_TILE_MAX_READ_COLS, DF_QUAL = HC_TILE_MAX_READ_QUAL, DF_SEGS = HC_TILE_MAX_READ_SEGS;
enum : uint8_t { DF_DEVICE = 0, DF_HOST = 1, DF_SKIP = 2, DF_UNMAPPED = 3 };

// One parser slice's arrays on the device, narrowed on their way into the staging block (the parser keeps 64-bit offsets and ids:
// 2.7 KB per read; what the kernels need of them is 1.4 KB): offsets within the slice and node ids as 32 bits (an id that does
// not fit is no node of the graph: 0xFFFFFFFF), the mapping's offset as int32 (INT32_MIN: does not fit -- the host's read), an
// edit as its length when it is a match or a substitution (from_length == to_length >= 0) and -1 otherwise, the identity as
// the one bit HaploCart.cpp:410 asks of it.
struct DfSlice {
    const uint32_t *map_off, *qual_off, *edit_off, *e_seq_off, *m_node;
    const int32_t *m_offset, *mapq, *e_len;
    const uint8_t *unmapped, *m_rev, *e_seq, *qual, *skip; // skip: NULL or per read of the slice
    uint32_t n_reads, read0;                               // read0: the slice's first read within the chunk
};
struct DfGraph {
    const int64_t *node_seq_off;
    const uint8_t *node_seq;
    const int32_t *pangenome_base;
    int64_t min_id, max_id;
    uint64_t n_mapp;
};
struct DfCounters { // zeroed per chunk
    unsigned int n_dev, n_in, n_unmapped, n_clamped, max_segs, max_qual, max_cols, max_span;
};

__device__ __forceinline__ uint8_t df_comp(uint8_t c) { // csrc/host/flatten.cpp: comp()
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

// A wave per read of the slice, lanes over its mappings.  Mirrors reconstruct_matches_only() + the route of flatten_range()
// (csrc/host/flatten.cpp): anything the one-walk form does not cover is the host's.  (A thread per read walked its ~60 mappings
// one dependent load after the other: 5 ms per 65 536 reads, half of the whole stage.)
__global__ __launch_bounds__(256) void hc_df_classify_kernel(DfSlice s, DfGraph g, uint8_t *__restrict__ flag, uint32_t *__restrict__ key,
                                                             uint4 *__restrict__ info, DfCounters *__restrict__ ctr) {
    const uint32_t lane = threadIdx.x & 63u;
    // (a wave takes many reads and sends its counts once: an atomic per read on the same few words serialises the whole launch --
    // 10 M reads x 6 atomics at ~12 ns were 0.69 of the 0.8 s a 10 M-read file's flatten took)
    uint32_t c_in = 0, c_unm = 0, c_dev = 0, c_clamped = 0, mx_segs = 0, mx_qual = 0, mx_cols = 0, mx_span = 0;
    // (the wave's index through a scalar register: the read's offsets and the slice's pointers are then scalar loads, not 64 copies of one)
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (uint32_t r = blockIdx.x * 4u + wv; r < s.n_reads; r += gridDim.x * 4u) {
    const uint32_t gr = s.read0 + r;
    uint8_t f = DF_HOST;
    uint32_t A = 0, G = 0, kmin = 0xFFFFFFFFu, kmax = 0u, nm = 0, nq = 0;
    if (s.skip && s.skip[r]) {
        f = DF_SKIP;
    } else if (s.unmapped[r]) { // identity < 1e-10, HaploCart.cpp:410
        f = DF_UNMAPPED;
    } else {
        const int64_t m0 = s.map_off[r], m1 = s.map_off[r + 1];
        const int64_t e0 = m1 > m0 ? s.edit_off[m0] : 0, e1 = m1 > m0 ? s.edit_off[m1] : 0;
        const int64_t q_len = (int64_t)s.qual_off[r + 1] - (int64_t)s.qual_off[r];
        bool ok = m1 > m0 && m1 - m0 <= (int64_t)DF_SEGS && q_len <= (int64_t)DF_QUAL && e1 - e0 >= m1 - m0;
        nm = (uint32_t)(m1 - m0);
        nq = (uint32_t)q_len;
        uint32_t a_len = 0, g_len = 0;
        bool empty_seg = false; // one of the read's first nm edits takes no column
        if (ok) { // ---- every mapping on known ground, the read's lengths
            bool bad = false;
            uint32_t gn = 0, an = 0;
            for (uint32_t mi = lane; mi < nm; mi += 64u) {
                // everything a mapping's checks read is asked for at once, at indices made safe first (an id outside the graph reads the
                // lowest node's words, a mapping without an edit the read's first edit's): the loads of a mapping are then two trips to
                // memory deep -- the mapping's own words, then what they point at -- where the checks between them made them five
                const int64_t m = m0 + mi;
                const int64_t id = s.m_node[m];
                const int64_t off_raw = s.m_offset[m];
                const int64_t ea = s.edit_off[m], eb = s.edit_off[m + 1];
                const bool id_ok = id >= g.min_id && id <= g.max_id;
                const int64_t idc = id_ok ? id : g.min_id;
                const int32_t pb = g.pangenome_base[idc];
                const int64_t len = g.node_seq_off[idc + 1] - g.node_seq_off[idc];
                const int64_t ec = eb > ea ? ea : e0; // (e0 is an edit of this read: e1 - e0 >= nm > 0)
                const int64_t from_first = s.e_len[ec];
                const int64_t sl_first = (int64_t)s.e_seq_off[ec + 1] - (int64_t)s.e_seq_off[ec];
                if (!id_ok || pb < 0 || (uint64_t)pb >= g.n_mapp) {
                    bad = true;
                    continue;
                }
                kmin = min(kmin, (uint32_t)id);
                kmax = max(kmax, (uint32_t)id);
                int64_t off = off_raw;
                if (off == (int64_t)INT32_MIN) { // (the offset did not fit 32 bits: the two walks of the general form disagree on such a read)
                    bad = true;
                    continue;
                }
                for (int64_t e = ea; e < eb; ++e) {
                    const int64_t from = e == ea ? from_first : (int64_t)s.e_len[e];
                    if (from < 0 || off > len || off < 0) { // (an edit that is not a match or a substitution: -1)
                        bad = true;
                        break;
                    }
                    const int64_t sl = e == ea ? sl_first : (int64_t)s.e_seq_off[e + 1] - (int64_t)s.e_seq_off[e];
                    const int64_t n = min(from, len - off);
                    gn += (uint32_t)n;
                    an += (uint32_t)(sl > 0 ? sl : n);
                    empty_seg = empty_seg || (n == 0 && e - e0 < (int64_t)nm);
                    off += from;
                }
            }
            ok = __builtin_amdgcn_ballot_w64(bad) == 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                gn += __shfl_xor(gn, o, 64);
                an += __shfl_xor(an, o, 64);
                kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64));
                kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
            }
            g_len = gn;
            a_len = an;
            // (|quality| <= |algnseq| and node ids within VGAN_HC_SREC's 18 bits: flatten.cpp's tile contract of a packed batch)
            ok = ok && a_len == g_len && a_len <= DF_COLS && a_len > 0 && nq <= a_len && kmax <= VGAN_HC_SREC_MAX_NODE;
        }
        // (segment i = mapping i with the size of the read's i-th EDIT, start = min(A, sum of the sizes before), length min(size, A - start);
        // a segment without a column sends the read to the general kernel, the host's business.  For a read that got here the sizes add
        // up to A, so a segment is empty exactly when its edit's size is zero: seen in the pass above, edit by edit -- a second pass over
        // the mappings with a scan of the sizes stood here and was half of this kernel's time)
        ok = ok && __builtin_amdgcn_ballot_w64(empty_seg) == 0;
        if (ok) {
            f = DF_DEVICE;
            A = a_len;
            G = g_len;
        }
    }
    if (lane == 0) {
        flag[gr] = f;
        // (csrc/host/flatten.cpp: sort_key() -- the reads of mapping quality VGAN_HC_MAPQ_MAJOR first, the others behind them)
        key[gr] = f == DF_DEVICE ? (min(kmin, 0x3FFFFFFFu) | (s.mapq[r] == VGAN_HC_MAPQ_MAJOR ? 0u : 0x40000000u)) : 0xFFFFFFFFu;
        info[gr] = uint4{G, nm, nq, A};
        c_in += f != DF_SKIP ? 1u : 0u;
        c_unm += f == DF_UNMAPPED ? 1u : 0u;
        if (f == DF_DEVICE) {
            c_dev += 1u;
            mx_segs = max(mx_segs, nm);
            mx_qual = max(mx_qual, nq);
            mx_cols = max(mx_cols, A);
            mx_span = max(mx_span, kmax - kmin);
            const int32_t mq = s.mapq[r];
            c_clamped += mq < 0 || mq > 99 ? 1u : 0u;
        }
    }
    } // (the wave's next read)
    // the workgroup's counts through LDS, then one lane's atomics: a launch of 32 768 waves sending seven atomics each to the same few
    // words took 2.4 ms for 500 k reads whatever else it did -- ~10 ns an atomic, one after the other (the launch is now 2 048 workgroups)
    __shared__ uint32_t red_s[4][8];
    if (lane == 0) {
        uint32_t *w = red_s[threadIdx.x >> 6];
        w[0] = c_in, w[1] = c_unm, w[2] = c_dev, w[3] = c_clamped, w[4] = mx_segs, w[5] = mx_qual, w[6] = mx_cols, w[7] = mx_span;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int k = 1; k < 4; ++k) {
        c_in += red_s[k][0], c_unm += red_s[k][1], c_dev += red_s[k][2], c_clamped += red_s[k][3];
        mx_segs = max(mx_segs, red_s[k][4]), mx_qual = max(mx_qual, red_s[k][5]), mx_cols = max(mx_cols, red_s[k][6]), mx_span = max(mx_span, red_s[k][7]);
    }
    if (c_in) atomicAdd(&ctr->n_in, c_in);
    if (c_unm) atomicAdd(&ctr->n_unmapped, c_unm);
    if (c_dev) {
        atomicAdd(&ctr->n_dev, c_dev);
        atomicMax(&ctr->max_segs, mx_segs);
        atomicMax(&ctr->max_qual, mx_qual);
        atomicMax(&ctr->max_cols, mx_cols);
        atomicMax(&ctr->max_span, mx_span);
    }
    if (c_clamped) atomicAdd(&ctr->n_clamped, c_clamped);
}

// sizes of the taken reads in sorted order (the sort's values are chunk read indices; taken reads come first)
__global__ __launch_bounds__(256) void hc_df_gather_kernel(const uint32_t *__restrict__ order, const uint4 *__restrict__ info, uint32_t n_dev,
                                                           uint32_t *__restrict__ segs, uint32_t *__restrict__ quals, uint32_t *__restrict__ cols) {
    const uint32_t o = blockIdx.x * 256u + threadIdx.x;
    if (o > n_dev) return;
    if (o == n_dev) { // (the scans' last input: their output there is the total)
        segs[o] = quals[o] = cols[o] = 0;
        return;
    }
    const uint4 v = info[order[o]];
    segs[o] = v.y;
    quals[o] = v.z;
    cols[o] = v.w;
}

struct DfOut {
    uint4 *rhdr;
    uint32_t *srec;
    uint32_t *crec;
    uint8_t *qualp;
    uint32_t *read_src;
};

// A wave per taken read, in sorted order.
__global__ __launch_bounds__(256) void hc_df_write_kernel(const DfSlice *__restrict__ slices, uint32_t n_slices, DfGraph g,
                                                          const uint32_t *__restrict__ order, const uint32_t *__restrict__ soff,
                                                          const uint32_t *__restrict__ qoff, const uint32_t *__restrict__ coff, uint32_t n_dev,
                                                          uint32_t src_base, DfOut out) {
    __shared__ uint8_t gs_s[4][DF_COLS], ps_s[4][DF_COLS];
    __shared__ uint16_t own_s[4][DF_COLS], sz_s[4][DF_SEGS];
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (scalar: see the classify kernel)
    uint8_t *gs = gs_s[wave], *ps = ps_s[wave];
    uint16_t *own = own_s[wave], *sz = sz_s[wave];
    for (uint32_t o = blockIdx.x * 4u + wave; o <= n_dev; o += gridDim.x * 4u) {
        if (o == n_dev) { // the end offsets, and the zero bytes behind the quality strings
            if (lane == 0) out.rhdr[o] = uint4{soff[o], qoff[o], coff[o], 0u};
            if (lane < 32) out.qualp[qoff[o] + lane] = 0;
            break;
        }
        const uint32_t gr = order[o];
        uint32_t si = 0;
        while (si + 1 < n_slices && slices[si + 1].read0 <= gr) ++si;
        const DfSlice s = slices[si];
        const uint32_t r = gr - s.read0;
        const int64_t m0 = s.map_off[r], m1 = s.map_off[r + 1];
        const uint32_t nm = (uint32_t)(m1 - m0);
        const uint32_t s0 = soff[o], q0 = qoff[o], c0 = coff[o], A = coff[o + 1] - c0, nq = qoff[o + 1] - q0;
        int32_t mq = s.mapq[r];
        mq = mq < 0 ? 0 : (mq > 99 ? 99 : mq);
        if (lane == 0) {
            out.rhdr[o] = uint4{s0, q0, c0, A | ((uint32_t)mq << 16)};
            out.read_src[o] = src_base + gr;
        }
        for (uint32_t c = lane; c < A; c += 64u) own[c] = 0;
        // ---- lanes over mappings: where each one's bases go (wave scans of the per-mapping totals), then the bytes.  Node bases
        // land at the running |graph_seq|, read bases at the running |path_string| (an edit's sequence where it has one, its
        // node bases otherwise): the two agree in total for a read that got here, not necessarily edit by edit
        uint32_t g_base = 0, a_base = 0, e_base = 0; // columns (graph / read side) and edits of the mappings before this pass
        for (uint32_t mb = 0; mb < nm; mb += 64u) {
            const uint32_t mi = mb + lane;
            const bool on = mi < nm;
            uint32_t gn = 0, an = 0, ne = 0;
            int64_t id = 0, len = 0, off0 = 0, ea = 0, eb = 0, nso = 0, from_first = 0, sl_first = 0;
            bool rev = false;
            if (on) {
                // (a mapping's words asked for at once, then what they point at -- its node's place, its first edit: the mappings of
                // these files have one edit --, as in the classify kernel; the second loop below takes them from here)
                const int64_t m = m0 + mi;
                id = s.m_node[m];
                off0 = s.m_offset[m];
                rev = s.m_rev[m] != 0;
                ea = s.edit_off[m], eb = s.edit_off[m + 1];
                nso = g.node_seq_off[id];
                len = g.node_seq_off[id + 1] - nso;
                if (eb > ea) {
                    from_first = s.e_len[ea];
                    sl_first = (int64_t)s.e_seq_off[ea + 1] - (int64_t)s.e_seq_off[ea];
                }
                int64_t off = off0;
                for (int64_t e = ea; e < eb; ++e) {
                    const int64_t from = e == ea ? from_first : (int64_t)s.e_len[e];
                    const int64_t sl = e == ea ? sl_first : (int64_t)s.e_seq_off[e + 1] - (int64_t)s.e_seq_off[e];
                    const uint32_t n = (uint32_t)min(from, len - off);
                    gn += n;
                    an += sl > 0 ? (uint32_t)sl : n;
                    off += from;
                    ++ne;
                }
            }
            const uint32_t gp = wave_incl_scan_u32(gn), ap = wave_incl_scan_u32(an), ep = wave_incl_scan_u32(ne); // inclusive prefix sums over the lanes
            const uint32_t g_tot = wave_last_u32(gp), a_tot = wave_last_u32(ap), e_tot = wave_last_u32(ep);
            uint32_t gq = g_base + gp - gn, aq = a_base + ap - an, eq = e_base + ep - ne; // this mapping's first places
            if (on) {
                const uint8_t *ns = g.node_seq + nso;
                int64_t off = off0;
                for (int64_t e = ea; e < eb; ++e, ++eq) {
                    const int64_t from = e == ea ? from_first : (int64_t)s.e_len[e];
                    const int64_t sl = e == ea ? sl_first : (int64_t)s.e_seq_off[e + 1] - (int64_t)s.e_seq_off[e];
                    const uint32_t n = (uint32_t)min(from, len - off);
                    for (uint32_t k = 0; k < n && gq + k < A; ++k) {
                        const uint8_t b = rev ? df_comp(ns[len - 1 - (off + k)]) : ns[off + k];
                        gs[gq + k] = b;
                        if (sl <= 0 && aq + k < A) ps[aq + k] = b;
                    }
                    if (sl > 0) {
                        const uint8_t *es = s.e_seq + s.e_seq_off[e];
                        for (int64_t k = 0; k < sl && aq + k < A; ++k) ps[aq + k] = es[k];
                    }
                    if (eq < nm) sz[eq] = (uint16_t)n;
                    gq += n;
                    aq += sl > 0 ? (uint32_t)sl : n;
                    off += from;
                }
            }
            g_base += g_tot;
            a_base += a_tot;
            e_base += e_tot;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- lanes over segments: segment i = mapping i with the size of the read's i-th edit (update_likelihood.cpp:33-45, Q6)
        {
            uint32_t p_base = 0;
            for (uint32_t sb = 0; sb < nm; sb += 64u) {
                const uint32_t i = sb + lane;
                const bool on = i < nm;
                const uint32_t n = on ? sz[i] : 0u;
                const uint32_t pp = wave_incl_scan_u32(n);
                const uint32_t p_tot = wave_last_u32(pp);
                if (on) {
                    const uint32_t start = min(A, p_base + pp - n), sl = min(n, A - start);
                    out.srec[s0 + i] = VGAN_HC_SREC(s.m_node[m0 + i], start, o);
                    for (uint32_t j = 0; j < sl; ++j) own[start + j] = (uint16_t)(start + 1u);
                }
                p_base += p_tot;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- lanes over columns and over quality bytes
        const uint8_t *q = s.qual + s.qual_off[r];
        for (uint32_t c = lane; c < A; c += 64u) {
            const uint32_t ow = own[c];
            uint32_t rec = (c < nq ? (uint32_t)q[c] : 0u) << 16; // (the quality byte on every column, scored or not)
            if (ow) {
                const uint32_t j = c - (ow - 1u);
                rec |= (uint32_t)gs[c] | ((uint32_t)ps[j] << 8) | (j == 0 ? VGAN_HC_CREC_HEAD : 0u);
            }
            out.crec[c0 + c] = rec;
        }
        for (uint32_t i = lane; i < nq; i += 64u) out.qualp[q0 + i] = q[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the next read's marks come after these reads of the wave's LDS)
    }
}

} // namespace df
} // namespace vgan

using namespace vgan::df;

// ------------------------------------------------------------------------------------------------------------ the C-ABI
namespace {
template <class T> struct DBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap && p) return VGAN_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = n + std::min<size_t>(n / 4, ((size_t)16 << 20) / sizeof(T)) + 256; // (slack for the next chunk to fit)
        HIPCHK(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        static const bool poison = getenv("VGAN_POISON_ALLOCS") != nullptr; // (test aid, as csrc/gam_kernels.hip: GBuf)
        if (poison) {
            HIPCHK(hipMemset(p, 0xA5, want * sizeof(T)));
            HIPCHK(hipDeviceSynchronize()); // (the fill runs on the null stream, the kernels that write the block on others)
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

struct vgan_hc_devflat {
    int device = 0;
    const vgan_hc_ctx *ctx = nullptr; // (its stream is asked for at every run: vgan_hc_set_stream may have changed it)
    hipStream_t stream = nullptr;
    uint32_t rows = 0;
    DfGraph g{};
    DBuf<int64_t> node_seq_off;
    DBuf<uint8_t> node_seq;
    DBuf<int32_t> pangenome_base;
    // one chunk's parser arrays (all slices, one after the other) and what the kernels make of them
    DBuf<uint8_t> stage;
    DBuf<DfSlice> slices;
    DBuf<uint8_t> flag;
    DBuf<uint32_t> key, key_out, val, val_out, segs, quals, cols, soff, qoff, coff, read_src;
    DBuf<uint4> info, rhdr;
    DBuf<uint32_t> srec;
    DBuf<uint32_t> crec;
    DBuf<uint8_t> qualp;
    DBuf<DfCounters> ctr;
    DBuf<uint64_t> tot64;
    DBuf<uint8_t> cub_tmp;
    std::vector<uint32_t> h_src;
    // the parser's arrays reach the device through pinned staging: the pieces are copied into it on several host threads (a
    // pageable hipMemcpy is ONE thread's memcpy into the runtime's own staging: 12.6 ms per 65 536 reads, the whole stage's
    // time), then one DMA takes the block
    uint8_t *pin = nullptr;
    size_t pin_cap = 0;
    uint32_t *pin_iota = nullptr; // 0, 1, 2, ...: the sort's values, uploaded once per size class
    size_t iota_cap = 0, iota_on_dev = 0;
    void release() {
        node_seq_off.release();
        node_seq.release();
        pangenome_base.release();
        stage.release();
        slices.release();
        flag.release();
        for (auto *b : {&key, &key_out, &val, &val_out, &segs, &quals, &cols, &soff, &qoff, &coff, &read_src, &crec}) b->release();
        if (pin) (void)hipHostFree(pin);
        pin = nullptr;
        pin_cap = 0;
        info.release();
        rhdr.release();
        srec.release();
        qualp.release();
        ctr.release();
        tot64.release();
        cub_tmp.release();
    }
};

extern "C" int vgan_hc_devflat_create(vgan_hc_ctx *c, const vgan_graph *graph, vgan_hc_devflat **out) {
    if (!c || !graph || !out) return fail(VGAN_EINVAL, "vgan_hc_devflat_create: null argument");
    const HcCtxInfo ci = hc_ctx_info(c);
    HIPCHK(hipSetDevice(ci.device));
    auto f = new vgan_hc_devflat();
    f->device = ci.device;
    f->ctx = c;
    f->stream = ci.stream;
    f->rows = ci.rows;
    int rc;
    auto bail = [&](int code) {
        f->release();
        delete f;
        return code;
    };
    const size_t n_off = graph->node_seq_off.size(), n_seq = graph->node_seq.size(), n_pb = graph->pangenome_base.size();
    if ((rc = f->node_seq_off.reserve(n_off)) || (rc = f->node_seq.reserve(n_seq + 1)) || (rc = f->pangenome_base.reserve(n_pb + 1)) ||
        (rc = f->ctr.reserve(1)))
        return bail(rc);
    if (hipMemcpy(f->node_seq_off.p, graph->node_seq_off.data(), n_off * 8, hipMemcpyHostToDevice) != hipSuccess ||
        (n_seq && hipMemcpy(f->node_seq.p, graph->node_seq.data(), n_seq, hipMemcpyHostToDevice) != hipSuccess) ||
        (n_pb && hipMemcpy(f->pangenome_base.p, graph->pangenome_base.data(), n_pb * 4, hipMemcpyHostToDevice) != hipSuccess))
        return bail(fail(VGAN_ENODEV, "vgan_hc_devflat_create: upload failed"));
    f->g.node_seq_off = f->node_seq_off.p;
    f->g.node_seq = f->node_seq.p;
    f->g.pangenome_base = f->pangenome_base.p;
    f->g.min_id = graph->min_id;
    f->g.max_id = std::min<int64_t>(graph->max_id, (int64_t)n_pb - 1);
    f->g.n_mapp = graph->mappability.size();
    *out = f;
    return VGAN_OK;
}

size_t vgan::hc_devflat_device_bytes(const vgan_hc_devflat *f) {
    if (!f) return 0;
    size_t b = f->node_seq_off.cap * 8 + f->node_seq.cap + f->pangenome_base.cap * 4 + f->stage.cap + f->slices.cap * sizeof(DfSlice) + f->flag.cap + f->qualp.cap +
               f->ctr.cap * sizeof(DfCounters) + f->cub_tmp.cap + (f->info.cap + f->rhdr.cap) * 16;
    for (auto *x : {&f->key, &f->key_out, &f->val, &f->val_out, &f->segs, &f->quals, &f->cols, &f->soff, &f->qoff, &f->coff, &f->read_src, &f->srec, &f->crec}) b += x->cap * 4;
    return b;
}

extern "C" void vgan_hc_devflat_free(vgan_hc_devflat *f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    f->release();
    delete f;
}

static int df_run_slices(vgan_hc_devflat *f, const std::vector<DfSlice> &hs, uint32_t R_all, uint32_t base, vgan_hc_packed_view *out, uint8_t *host_mask,
                         vgan_hc_flatten_stats *stats, PhaseTimer &pt, void (*mask_ready)(void *) = nullptr, void *user = nullptr);
static int df_prepare(vgan_hc_devflat *f, uint32_t R_all, size_t n_slices);

extern "C" int vgan_hc_devflat_run(vgan_hc_devflat *f, const vgan_alnparts *chunk, const uint8_t *skip, vgan_hc_packed_view *out,
                                   uint8_t *host_mask, vgan_hc_flatten_stats *stats) {
    if (!f || !chunk || !out || !host_mask) return fail(VGAN_EINVAL, "vgan_hc_devflat_run: null argument");
    memset(out, 0, sizeof *out);
    if (stats) memset(stats, 0, sizeof *stats);
    const size_t np = chunk->parts.size();
    const int64_t n_reads = chunk->first.back();
    if (n_reads == 0) return VGAN_OK;
    if (n_reads > 0x7FFFFFF0ll || chunk->base + n_reads > 0xFFFFFFF0ll) return fail(VGAN_ERANGE, "vgan_hc_devflat_run: too many reads in one chunk");
    HIPCHK(hipSetDevice(f->device));
    // the context's stream as it is NOW: the segment kernel reads this object's output on it, and this run overwrites (or
    // re-allocates) that output -- the two must sit on one stream.  A change of streams: what the old one still holds is waited for.
    {
        const hipStream_t now = hc_ctx_info(f->ctx).stream;
        if (now != f->stream) {
            if (f->stream) HIPCHK(hipStreamSynchronize(f->stream));
            f->stream = now;
        }
    }
    hipStream_t st = f->stream;
    int rc;
    // ---- the slices' arrays, narrowed (DfSlice), one after the other in one staging block (8-byte aligned pieces)
    auto up8 = [](size_t n) { return (n + 7) & ~(size_t)7; };
    size_t total = 0;
    for (const vgan_alnset &a : chunk->parts) {
        const size_t R = (size_t)a.n_reads(), M = a.m_node.size(), E = a.e_from.size();
        total += up8((R + 1) * 4) * 2 + up8((M + 1) * 4) + up8((E + 1) * 4) + up8(M * 4) * 2 + up8(R * 4) + up8(E * 4) + up8(R) + up8(M) +
                 up8(a.e_seq.size()) + up8(a.qual.size()) + up8(R);
        if (a.seq_off.size() && (a.qual.size() > 0xFFFFFFF0ull || a.e_seq.size() > 0xFFFFFFF0ull || M > 0xFFFFFFF0ull || E > 0xFFFFFFF0ull))
            return fail(VGAN_ERANGE, "vgan_hc_devflat_run: a parser slice beyond 32-bit offsets");
    }
    const uint32_t R_all = (uint32_t)n_reads;
    if ((rc = f->stage.reserve(total)) || (rc = df_prepare(f, R_all, np))) return rc;
    PhaseTimer pt("hc_devflat");
    if (total > f->pin_cap) {
        if (f->pin) (void)hipHostFree(f->pin);
        f->pin = nullptr;
        f->pin_cap = 0;
        const size_t want = total + total / 4 + (1u << 20);
        HIPCHK(hipHostMalloc((void **)&f->pin, want, hipHostMallocDefault));
        f->pin_cap = want;
    }
    std::vector<DfSlice> hs(np);
    // what a piece is made of: bytes as they are, or a narrowing of the parser's wider array
    enum Kind { K_BYTES, K_OFF64, K_NODE64, K_MOFF64, K_EDIT, K_UNMAPPED };
    struct Piece {
        Kind kind;
        const void *src, *src2;
        size_t off, n; // destination offset in the block; elements (bytes for K_BYTES)
    };
    std::vector<Piece> pieces;
    size_t cur = 0;
    auto put = [&](Kind k, const void *src, const void *src2, size_t n, size_t elem) -> const void * {
        const size_t off = cur;
        cur += up8(n * elem);
        if (n) pieces.push_back({k, src, src2, off, n});
        return f->stage.p + off;
    };
    for (size_t i = 0; i < np; ++i) {
        const vgan_alnset &a = chunk->parts[i];
        const size_t R = (size_t)a.n_reads(), M = a.m_node.size(), E = a.e_from.size();
        DfSlice &s = hs[i];
        s.n_reads = (uint32_t)R;
        s.read0 = (uint32_t)chunk->first[i];
        s.map_off = (const uint32_t *)put(K_OFF64, a.map_off.data(), nullptr, R + 1, 4);
        s.qual_off = (const uint32_t *)put(K_OFF64, a.qual_off.data(), nullptr, R + 1, 4);
        s.edit_off = (const uint32_t *)put(K_OFF64, a.edit_off.data(), nullptr, M + 1, 4);
        s.e_seq_off = (const uint32_t *)put(K_OFF64, a.e_seq_off.data(), nullptr, E + 1, 4);
        s.m_node = (const uint32_t *)put(K_NODE64, a.m_node.data(), nullptr, M, 4);
        s.m_offset = (const int32_t *)put(K_MOFF64, a.m_offset.data(), nullptr, M, 4);
        s.mapq = (const int32_t *)put(K_BYTES, a.mapq.data(), nullptr, R * 4, 1);
        s.e_len = (const int32_t *)put(K_EDIT, a.e_from.data(), a.e_to.data(), E, 4);
        s.unmapped = (const uint8_t *)put(K_UNMAPPED, a.identity.data(), nullptr, R, 1);
        s.m_rev = (const uint8_t *)put(K_BYTES, a.m_rev.data(), nullptr, M, 1);
        s.e_seq = (const uint8_t *)put(K_BYTES, a.e_seq.data(), nullptr, a.e_seq.size(), 1);
        s.qual = (const uint8_t *)put(K_BYTES, a.qual.data(), nullptr, a.qual.size(), 1);
        s.skip = skip ? (const uint8_t *)put(K_BYTES, skip + chunk->first[i], nullptr, R, 1) : nullptr;
    }
    { // host copies / narrowings into the pinned block, in jobs of ~256k elements over the host threads, then one DMA
        struct Job {
            const Piece *pc;
            size_t e0, e1;
        };
        std::vector<Job> jobs;
        constexpr size_t STEP = 1u << 18;
        for (const Piece &pc : pieces)
            for (size_t o = 0; o < pc.n; o += STEP) jobs.push_back({&pc, o, std::min(pc.n, o + STEP)});
        const int nth = (int)std::max<size_t>(1, std::min<size_t>({(size_t)usable_cpus(), (size_t)8, jobs.size() / 4 + 1}));
        std::atomic<size_t> next{0};
        parallel_run(nth, [&](int) {
            for (;;) {
                const size_t j = next.fetch_add(1);
                if (j >= jobs.size()) break;
                const Piece &pc = *jobs[j].pc;
                const size_t e0 = jobs[j].e0, e1 = jobs[j].e1;
                uint8_t *dst = f->pin + pc.off;
                switch (pc.kind) {
                case K_BYTES: memcpy(dst + e0, (const uint8_t *)pc.src + e0, e1 - e0); break;
                case K_OFF64: {
                    const int64_t *src = (const int64_t *)pc.src;
                    uint32_t *d = (uint32_t *)dst;
                    for (size_t e = e0; e < e1; ++e) d[e] = (uint32_t)src[e];
                    break;
                }
                case K_NODE64: {
                    const int64_t *src = (const int64_t *)pc.src;
                    uint32_t *d = (uint32_t *)dst;
                    for (size_t e = e0; e < e1; ++e) d[e] = src[e] < 0 || src[e] > 0xFFFFFFFEll ? 0xFFFFFFFFu : (uint32_t)src[e];
                    break;
                }
                case K_MOFF64: {
                    const int64_t *src = (const int64_t *)pc.src;
                    int32_t *d = (int32_t *)dst;
                    for (size_t e = e0; e < e1; ++e) d[e] = src[e] != (int64_t)(int32_t)src[e] || (int32_t)src[e] == INT32_MIN ? INT32_MIN : (int32_t)src[e];
                    break;
                }
                case K_EDIT: {
                    const int32_t *from = (const int32_t *)pc.src, *to = (const int32_t *)pc.src2;
                    int32_t *d = (int32_t *)dst;
                    for (size_t e = e0; e < e1; ++e) d[e] = from[e] == to[e] && from[e] >= 0 ? from[e] : -1;
                    break;
                }
                case K_UNMAPPED: {
                    const double *src = (const double *)pc.src;
                    for (size_t e = e0; e < e1; ++e) dst[e] = src[e] < 1e-10 ? 1 : 0;
                    break;
                }
                }
            }
        });
        pt.lap("staging (narrowing copy)");
        if (cur) HIPCHK(hipMemcpyAsync(f->stage.p, f->pin, cur, hipMemcpyHostToDevice, st));
    }
    return df_run_slices(f, hs, R_all, (uint32_t)chunk->base, out, host_mask, stats, pt);
}

// classify -> sort -> offsets -> write over slices whose arrays are on the device (uploaded above, or left there by the GAM front
// end on the device: gam_kernels.hip)
static int df_run_slices(vgan_hc_devflat *f, const std::vector<DfSlice> &hs, uint32_t R_all, uint32_t base, vgan_hc_packed_view *out, uint8_t *host_mask,
                         vgan_hc_flatten_stats *stats, PhaseTimer &pt, void (*mask_ready)(void *), void *user) {
    hipStream_t st = f->stream;
    const size_t np = hs.size();
    int rc;
    for (size_t i = 0; i < np; ++i)
        if (hs[i].n_reads)
            hipLaunchKernelGGL(hc_df_classify_kernel, dim3(std::min<uint32_t>((hs[i].n_reads + 3) / 4, 2048u)), dim3(256), 0, st, hs[i], f->g, f->flag.p, f->key.p,
                               f->info.p, f->ctr.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(f->slices.p, hs.data(), np * sizeof(DfSlice), hipMemcpyHostToDevice, st));
    // ---- the taken reads in ascending order of their lowest node id, input order kept among equals (the others' key is 2^32 - 1)
    {
        if (f->iota_on_dev < R_all) { // 0, 1, 2, ...: on the device once, a prefix of it serves every smaller chunk
            const size_t want = (size_t)R_all + R_all / 4 + 1024;
            std::vector<uint32_t> iota(want);
            for (size_t i = 0; i < want; ++i) iota[i] = (uint32_t)i;
            if ((rc = f->val.reserve(want))) return rc;
            HIPCHK(hipMemcpy(f->val.p, iota.data(), want * 4, hipMemcpyHostToDevice));
            f->iota_on_dev = want;
        }
        size_t tmp = 0;
        if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, f->key.p, f->key_out.p, f->val.p, f->val_out.p, (int)R_all, 0, 32, st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_hc_devflat_run: sort sizing failed");
        if ((rc = f->cub_tmp.reserve(tmp))) return rc;
        if (hipcub::DeviceRadixSort::SortPairs(f->cub_tmp.p, tmp, f->key.p, f->key_out.p, f->val.p, f->val_out.p, (int)R_all, 0, 32, st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_hc_devflat_run: sort failed");
    }
    DfCounters hc{};
    HIPCHK(hipMemcpyAsync(&hc, f->ctr.p, sizeof hc, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(host_mask, f->flag.p, R_all, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    pt.lap("upload + classify + sort");
    for (uint32_t i = 0; i < R_all; ++i) host_mask[i] = host_mask[i] == DF_HOST ? 1 : 0;
    if (mask_ready) mask_ready(user); // (the caller's work on the reads left to the host can start beside the offsets and the write pass)
    const uint32_t n_dev = hc.n_dev;
    if (stats) {
        stats->n_in = hc.n_in;
        stats->n_unmapped = hc.n_unmapped;
        stats->n_clamped = hc.n_clamped;
        stats->n_out = n_dev;
    }
    if (n_dev == 0) return VGAN_OK;
    // ---- offsets of the sorted reads: exclusive sums of their three sizes (entry n_dev: the totals)
    hipLaunchKernelGGL(hc_df_gather_kernel, dim3((n_dev + 1 + 255) / 256), dim3(256), 0, st, f->val_out.p, f->info.p, n_dev, f->segs.p, f->quals.p, f->cols.p);
    {
        size_t tmp = 0;
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, f->segs.p, f->soff.p, (int)(n_dev + 1), st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_hc_devflat_run: scan sizing failed");
        if ((rc = f->cub_tmp.reserve(tmp))) return rc;
        if (hipcub::DeviceScan::ExclusiveSum(f->cub_tmp.p, tmp, f->segs.p, f->soff.p, (int)(n_dev + 1), st) != hipSuccess ||
            hipcub::DeviceScan::ExclusiveSum(f->cub_tmp.p, tmp, f->quals.p, f->qoff.p, (int)(n_dev + 1), st) != hipSuccess ||
            hipcub::DeviceScan::ExclusiveSum(f->cub_tmp.p, tmp, f->cols.p, f->coff.p, (int)(n_dev + 1), st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_hc_devflat_run: scan failed");
    }
    // (the offsets are 32 bits wide: segments and quality bytes are subsets of the parsed bytes, whose number the parse bounds; columns are
    // sums of edit LENGTHS -- up to DF_COLS per read whatever its bytes -- so their total is taken in 64 bits too and a chunk whose
    // columns would wrap the offsets is refused: the caller flattens fewer reads at a time, or on the host)
    if ((rc = f->tot64.reserve(1))) return rc;
    {
        struct Widen {
            __host__ __device__ uint64_t operator()(uint32_t v) const { return v; }
        };
        hipcub::TransformInputIterator<uint64_t, Widen, const uint32_t *> it(f->cols.p, Widen());
        size_t tmp = 0;
        if (hipcub::DeviceReduce::Sum(nullptr, tmp, it, f->tot64.p, (int)n_dev, st) != hipSuccess) return fail(VGAN_ENODEV, "vgan_hc_devflat_run: sum sizing failed");
        if ((rc = f->cub_tmp.reserve(tmp))) return rc;
        if (hipcub::DeviceReduce::Sum(f->cub_tmp.p, tmp, it, f->tot64.p, (int)n_dev, st) != hipSuccess) return fail(VGAN_ENODEV, "vgan_hc_devflat_run: sum failed");
    }
    uint32_t tot[3] = {0, 0, 0};
    uint64_t cols64 = 0;
    HIPCHK(hipMemcpyAsync(&tot[0], f->soff.p + n_dev, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&tot[1], f->qoff.p + n_dev, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&tot[2], f->coff.p + n_dev, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&cols64, f->tot64.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    {
        static const char *lim = getenv("VGAN_HC_DEVFLAT_MAX_COLS"); // (test aid: the refusal without four billion columns)
        const uint64_t max_cols = lim ? strtoull(lim, nullptr, 10) : 0xFFFFFFF0ull;
        if (cols64 > max_cols)
            return fail(VGAN_ERANGE, "vgan_hc_devflat_run: %llu alignment columns in one chunk are beyond the packed batch's 32-bit offsets; flatten fewer reads at a time",
                        (unsigned long long)cols64);
    }
    pt.lap("offsets");
    if ((rc = f->rhdr.reserve((size_t)n_dev + 1)) || (rc = f->srec.reserve(tot[0] + 1)) || (rc = f->crec.reserve(tot[2] + 1)) ||
        (rc = f->qualp.reserve((size_t)tot[1] + 32)) || (rc = f->read_src.reserve(n_dev)))
        return rc;
    pt.lap("output blocks");
    DfOut o{f->rhdr.p, f->srec.p, f->crec.p, f->qualp.p, f->read_src.p};
    hipLaunchKernelGGL(hc_df_write_kernel, dim3(std::min<uint32_t>((n_dev + 1 + 3) / 4, 16384u)), dim3(256), 0, st, f->slices.p, (uint32_t)np, f->g,
                       f->val_out.p, f->soff.p, f->qoff.p, f->coff.p, n_dev, base, o);
    HIPCHK(hipGetLastError());
    f->h_src.resize(n_dev);
    HIPCHK(hipMemcpyAsync(f->h_src.data(), f->read_src.p, (size_t)n_dev * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st)); // (the staging block is reused by the next call; the caller's arrays may go)
    pt.lap("write");
    out->n_reads = n_dev;
    out->n_segments = tot[0];
    out->n_qual = tot[1];
    out->n_cols = tot[2];
    out->rhdr = reinterpret_cast<const uint32_t *>(f->rhdr.p);
    out->srec = f->srec.p;
    out->crec = f->crec.p;
    out->qualp = f->qualp.p;
    out->max_read_segs = hc.max_segs;
    out->max_read_qual = hc.max_qual;
    out->max_read_cols = hc.max_cols;
    out->max_read_node_span = hc.max_span;
    out->on_device = 1;
    out->read_src = f->h_src.data();
    if (stats) {
        stats->n_segments = tot[0];
        stats->n_cols = tot[2];
    }
    return VGAN_OK;
}

static int df_prepare(vgan_hc_devflat *f, uint32_t R_all, size_t n_slices) {
    int rc;
    if ((rc = f->slices.reserve(n_slices)) || (rc = f->flag.reserve(R_all)) || (rc = f->key.reserve(R_all)) || (rc = f->key_out.reserve(R_all)) ||
        (rc = f->val_out.reserve(R_all)) || (rc = f->info.reserve(R_all)) || (rc = f->segs.reserve(R_all + 1)) || (rc = f->quals.reserve(R_all + 1)) ||
        (rc = f->cols.reserve(R_all + 1)) || (rc = f->soff.reserve(R_all + 1)) || (rc = f->qoff.reserve(R_all + 1)) || (rc = f->coff.reserve(R_all + 1)))
        return rc;
    HIPCHK(hipMemsetAsync(f->ctr.p, 0, sizeof(DfCounters), f->stream));
    return VGAN_OK;
}

// (ABI 5) The same over the arrays the GAM front end on the device left in HBM (vgan_gamdev_parse): nothing crosses the link but the
// duplicate marks (skip: host, per read of the parse, or NULL) on their way up and the host-read mask on its way down.
extern "C" int vgan_hc_devflat_run_gamdev(vgan_hc_devflat *f, const vgan_gamdev *gd, const uint8_t *skip, int skip_on_device, uint32_t base,
                                          vgan_hc_packed_view *out, uint8_t *host_mask, vgan_hc_flatten_stats *stats) {
    return vgan_hc_devflat_run_gamdev_cb(f, gd, skip, skip_on_device, base, out, host_mask, stats, nullptr, nullptr);
}

extern "C" int vgan_hc_devflat_run_gamdev_cb(vgan_hc_devflat *f, const vgan_gamdev *gd, const uint8_t *skip, int skip_on_device, uint32_t base,
                                             vgan_hc_packed_view *out, uint8_t *host_mask, vgan_hc_flatten_stats *stats, void (*mask_ready)(void *),
                                             void *user) {
    if (!f || !gd || !out || !host_mask) return fail(VGAN_EINVAL, "vgan_hc_devflat_run_gamdev: null argument");
    memset(out, 0, sizeof *out);
    if (stats) memset(stats, 0, sizeof *stats);
    GamdevSlice gs{};
    if (!gamdev_slice(gd, &gs)) return fail(VGAN_ESTATE, "vgan_hc_devflat_run_gamdev: the front end holds no parse");
    if (gs.n_reads == 0) return VGAN_OK;
    if (gs.device != f->device) return fail(VGAN_EINVAL, "vgan_hc_devflat_run_gamdev: the parse lives on another device");
    if (gs.n_reads > 0x7FFFFFF0ull || (uint64_t)base + gs.n_reads > 0xFFFFFFF0ull) return fail(VGAN_ERANGE, "vgan_hc_devflat_run_gamdev: too many reads in one parse");
    HIPCHK(hipSetDevice(f->device));
    {
        const hipStream_t now = hc_ctx_info(f->ctx).stream;
        if (now != f->stream) {
            if (f->stream) HIPCHK(hipStreamSynchronize(f->stream));
            f->stream = now;
        }
    }
    const uint32_t R_all = (uint32_t)gs.n_reads;
    int rc;
    if ((rc = df_prepare(f, R_all, 1))) return rc;
    PhaseTimer pt("hc_devflat (device parse)");
    const uint8_t *d_skip = nullptr;
    if (skip && skip_on_device) {
        d_skip = skip;
    } else if (skip) {
        if ((rc = f->stage.reserve(R_all))) return rc;
        HIPCHK(hipMemcpyAsync(f->stage.p, skip, R_all, hipMemcpyHostToDevice, f->stream));
        d_skip = f->stage.p;
    }
    std::vector<DfSlice> hs(1);
    DfSlice &s = hs[0];
    s.map_off = gs.map_off, s.qual_off = gs.qual_off, s.edit_off = gs.edit_off, s.e_seq_off = gs.e_seq_off, s.m_node = gs.m_node;
    s.m_offset = gs.m_offset, s.mapq = gs.mapq, s.e_len = gs.e_len;
    s.unmapped = gs.unmapped, s.m_rev = gs.m_rev, s.e_seq = gs.e_seq, s.qual = gs.qual, s.skip = d_skip;
    s.n_reads = R_all, s.read0 = 0;
    return df_run_slices(f, hs, R_all, base, out, host_mask, stats, pt, mask_ready, user);
}
#include "module_anchor.h"
const void *vgan::anchor_hc_flatten() { return (const void *)&vgan::df::hc_df_gather_kernel; }
