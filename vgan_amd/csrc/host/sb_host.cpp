// soibean host side: the front half of analyse_GAM (reference src/getLCAfromGAM.h:92-186,537-544) as an SoA batch --
// reconstruct_graph_sequence, then the edit-level slicing with the reference's baseIX walk (forward: running start;
// reverse: from the end backwards, startIndex = max(baseIX - size - 1, 0), Q12).
#include "common.h"

#include <algorithm>
#include <cstring>
#include <thread>

using namespace vgan;

struct vgan_sb_host_batch {
    std::vector<uint32_t> read_seg_off{0}, read_col_off{0}, read_qual_off{0}, read_src, seg_node;
    std::vector<uint16_t> read_gseq_len, read_rseq_len, seg_col, seg_len, seg_base_ix;
    std::vector<uint8_t> read_rev, graph_seq, read_seq, qual;
};

namespace {

struct SChunk {
    vgan_sb_host_batch b;
    vgan_sb_flatten_stats st{};
};

void sb_flatten_range(const vgan_graph &g, const vgan_alnset &a, int64_t r0, int64_t r1, SChunk &c) {
    Recon rc;
    auto &b = c.b;
    for (int64_t r = r0; r < r1; ++r) {
        c.st.n_in++;
        if (a.identity[r] == 0) { // getLCAfromGAM.h:101
            c.st.n_unmapped++;
            continue;
        }
        const int64_t m0 = a.map_off[r], nM = a.map_off[r + 1] - m0;
        int bad = nM == 0 ? 1 : reconstruct(g, a, r, rc);
        const size_t G = rc.gseq.size(), A = rc.ps.size();
        if (!bad && (G < 15 || G > 1000 || A > 65535)) bad = 1; // subDeamDiNuc[Lseq] with Lseq = |graph_seq| (:109)
        const size_t seg_mark = b.seg_node.size();
        if (!bad) {
            const bool rev = a.m_rev[m0] != 0;
            const int64_t seq_size = a.seq_off[r + 1] - a.seq_off[r];
            int64_t baseIX = rev ? seq_size - 1 : 0; // :107
            const size_t nE = rc.sizes.size();
            for (size_t i = 0; i < nE; ++i) {
                const int64_t size = rc.sizes[i];
                int64_t node = 0; // "No_support" for edit-level segments beyond the mappings (:156-160)
                if (i < (size_t)nM) {
                    node = a.m_node[m0 + (int64_t)i];
                    if (!g.has_node(node) || node <= 0) { // nodepaths.at(nID - minid) throws
                        bad = 1;
                        break;
                    }
                }
                const int64_t start = rev ? std::max<int64_t>(baseIX - size - 1, 0) : baseIX; // :179-186
                if (start < 0 || (size_t)start > G || (size_t)start > A) {                     // substr throws
                    bad = 1;
                    break;
                }
                const size_t len = std::min<size_t>((size_t)std::max<int64_t>(size, 0), G - (size_t)start);
                if (len > 0 && (baseIX < 0 || (size_t)baseIX >= G)) { // subDeamDiNuc[Lseq][baseIX] out of range
                    bad = 1;
                    break;
                }
                b.seg_node.push_back((uint32_t)node);
                b.seg_col.push_back((uint16_t)start);
                b.seg_len.push_back((uint16_t)len);
                b.seg_base_ix.push_back((uint16_t)std::max<int64_t>(baseIX, 0));
                if (rev) baseIX = start; // :537-544
                else baseIX += size;
            }
        }
        if (bad) {
            b.seg_node.resize(seg_mark);
            b.seg_col.resize(seg_mark);
            b.seg_len.resize(seg_mark);
            b.seg_base_ix.resize(seg_mark);
            c.st.n_bad++;
            continue;
        }
        const size_t region = std::max(G, A);
        b.graph_seq.insert(b.graph_seq.end(), rc.gseq.begin(), rc.gseq.end());
        b.graph_seq.insert(b.graph_seq.end(), region - G, 0);
        b.read_seq.insert(b.read_seq.end(), rc.ps.begin(), rc.ps.end());
        b.read_seq.insert(b.read_seq.end(), region - A, 0);
        const char *q = a.qual.data() + a.qual_off[r];
        b.qual.insert(b.qual.end(), q, q + (a.qual_off[r + 1] - a.qual_off[r]));
        b.read_gseq_len.push_back((uint16_t)G);
        b.read_rseq_len.push_back((uint16_t)A);
        b.read_rev.push_back(a.m_rev[m0]);
        b.read_src.push_back((uint32_t)r);
        b.read_seg_off.push_back((uint32_t)b.seg_node.size());
        b.read_col_off.push_back((uint32_t)b.graph_seq.size());
        b.read_qual_off.push_back((uint32_t)b.qual.size());
        c.st.n_out++;
    }
}

template <class T> void cat_shift(std::vector<T> &dst, const std::vector<T> &src, T shift) {
    for (size_t i = 1; i < src.size(); ++i) dst.push_back((T)(src[i] + shift));
}
template <class T> void cat(std::vector<T> &dst, const std::vector<T> &src) { dst.insert(dst.end(), src.begin(), src.end()); }

} // namespace

extern "C" int vgan_sb_flatten(const vgan_graph *g, const vgan_alnset *a, int64_t r0, int64_t r1, int n_threads,
                               vgan_sb_host_batch **out, vgan_sb_flatten_stats *stats) {
    if (!g || !a || !out) return fail(VGAN_EINVAL, "vgan_sb_flatten: null argument");
    if (r0 < 0 || r1 > a->n_reads() || r0 > r1) return fail(VGAN_EINVAL, "vgan_sb_flatten: bad read range");
    if (n_threads <= 0) n_threads = (int)usable_cpus();
    const int64_t n = r1 - r0;
    n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, (n + 4095) / 4096));
    std::vector<SChunk> chunks((size_t)n_threads);
    parallel_run(n_threads, [&](int t) {
        const int64_t b0 = r0 + n * t / n_threads, b1 = r0 + n * (t + 1) / n_threads;
        sb_flatten_range(*g, *a, b0, b1, chunks[(size_t)t]);
    });
    auto res = new vgan_sb_host_batch();
    vgan_sb_flatten_stats st{};
    uint64_t tc = 0, tq = 0, ts = 0;
    for (auto &c : chunks) {
        tc += c.b.graph_seq.size();
        tq += c.b.qual.size();
        ts += c.b.seg_node.size();
    }
    if (tc > 0xFFFFFFF0ull || tq > 0xFFFFFFF0ull || ts > 0xFFFFFFF0ull) {
        delete res;
        return fail(VGAN_ERANGE, "vgan_sb_flatten: batch exceeds 32-bit offsets; flatten fewer reads per batch");
    }
    // every array sized once, the chunks copied into their places side by side (appending them one after the other on one
    // thread, with the arrays growing under it, cost as much as flattening them)
    struct Base {
        size_t r, s, c, q;
    };
    std::vector<Base> base(chunks.size() + 1, Base{0, 0, 0, 0});
    for (size_t i = 0; i < chunks.size(); ++i) {
        const auto &cb = chunks[i].b;
        base[i + 1] = Base{base[i].r + cb.read_rev.size(), base[i].s + cb.seg_node.size(), base[i].c + cb.graph_seq.size(), base[i].q + cb.qual.size()};
        st.n_in += chunks[i].st.n_in;
        st.n_out += chunks[i].st.n_out;
        st.n_unmapped += chunks[i].st.n_unmapped;
        st.n_bad += chunks[i].st.n_bad;
    }
    const Base tot = base.back();
    res->read_seg_off.resize(tot.r + 1);
    res->read_col_off.resize(tot.r + 1);
    res->read_qual_off.resize(tot.r + 1);
    res->read_gseq_len.resize(tot.r);
    res->read_rseq_len.resize(tot.r);
    res->read_rev.resize(tot.r);
    res->read_src.resize(tot.r);
    res->seg_node.resize(tot.s);
    res->seg_col.resize(tot.s);
    res->seg_len.resize(tot.s);
    res->seg_base_ix.resize(tot.s);
    res->graph_seq.resize(tot.c);
    res->read_seq.resize(tot.c);
    res->qual.resize(tot.q);
    res->read_seg_off[0] = res->read_col_off[0] = res->read_qual_off[0] = 0;
    parallel_run((int)chunks.size(), [&](int ti) {
        auto &cb = chunks[(size_t)ti].b;
        const Base &bs = base[(size_t)ti];
        for (size_t j = 1; j < cb.read_seg_off.size(); ++j) {
            res->read_seg_off[bs.r + j] = cb.read_seg_off[j] + (uint32_t)bs.s;
            res->read_col_off[bs.r + j] = cb.read_col_off[j] + (uint32_t)bs.c;
            res->read_qual_off[bs.r + j] = cb.read_qual_off[j] + (uint32_t)bs.q;
        }
        auto cp = [](auto &dst, size_t at, const auto &src) {
            if (!src.empty()) memcpy(&dst[at], src.data(), src.size() * sizeof(src[0]));
        };
        cp(res->read_gseq_len, bs.r, cb.read_gseq_len);
        cp(res->read_rseq_len, bs.r, cb.read_rseq_len);
        cp(res->read_rev, bs.r, cb.read_rev);
        cp(res->read_src, bs.r, cb.read_src);
        cp(res->seg_node, bs.s, cb.seg_node);
        cp(res->seg_col, bs.s, cb.seg_col);
        cp(res->seg_len, bs.s, cb.seg_len);
        cp(res->seg_base_ix, bs.s, cb.seg_base_ix);
        cp(res->graph_seq, bs.c, cb.graph_seq);
        cp(res->read_seq, bs.c, cb.read_seq);
        cp(res->qual, bs.q, cb.qual);
        cb = vgan_sb_host_batch();
    });
    if (stats) *stats = st;
    *out = res;
    return VGAN_OK;
}

extern "C" int vgan_sb_host_batch_get(const vgan_sb_host_batch *b, vgan_sb_batch *o) {
    if (!b || !o) return fail(VGAN_EINVAL, "vgan_sb_host_batch_get: null argument");
    memset(o, 0, sizeof *o);
    o->n_reads = (uint32_t)b->read_rev.size();
    o->n_segments = (uint32_t)b->seg_node.size();
    o->n_cols = b->graph_seq.size();
    o->n_qual = b->qual.size();
    o->read_seg_off = b->read_seg_off.data();
    o->read_col_off = b->read_col_off.data();
    o->read_qual_off = b->read_qual_off.data();
    o->read_gseq_len = b->read_gseq_len.data();
    o->read_rseq_len = b->read_rseq_len.data();
    o->read_rev = b->read_rev.data();
    o->read_src = b->read_src.data();
    o->seg_node = b->seg_node.data();
    o->seg_col = b->seg_col.data();
    o->seg_len = b->seg_len.data();
    o->seg_base_ix = b->seg_base_ix.data();
    o->graph_seq = b->graph_seq.data();
    o->read_seq = b->read_seq.data();
    o->qual = b->qual.data();
    return VGAN_OK;
}

extern "C" void vgan_sb_host_batch_free(vgan_sb_host_batch *b) { delete b; }

// soibean.cpp:655-712: signature paths from the per-path counts of reads with a unique best path
extern "C" int vgan_sb_signature_paths(const int64_t *sig_count, uint32_t n_paths, int64_t n_reads, int32_t cutk, int32_t *paths, int32_t *n) {
    if (!sig_count || !paths || !n) return fail(VGAN_EINVAL, "vgan_sb_signature_paths: null argument");
    std::vector<std::pair<int64_t, uint32_t>> seen; // (count, path) of the paths some read singles out
    for (uint32_t p = 0; p < n_paths; ++p)
        if (sig_count[p] > 0) seen.emplace_back(sig_count[p], p);
    std::sort(seen.begin(), seen.end(), [](const auto &a, const auto &b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    const double thres = (double)n_reads * 0.01;
    int32_t k = 0;
    for (const auto &e : seen)
        if ((double)e.first >= thres) paths[k++] = (int32_t)e.second;
    if (cutk > 0 && k > cutk) k = cutk; // the reference resize()s: a cutk beyond the list would pad it with node 0, which no run relies on
    if (k == 0)
        for (const auto &e : seen) paths[k++] = (int32_t)e.second; // "Rerunning with no minimum threshold"
    *n = k;
    return VGAN_OK;
}
