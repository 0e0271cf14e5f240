// Front half of the HaploCart hot path on the host: alignment -> (graph_seq, algnseq, per-edit sizes)
// -> per-mapping segments -> SoA batch for the device.
//
// Mirrors what the reference does before its per-path loops, quirks included:
//   reconstruct_graph_sequence  src/vgan_utils.h:6-79      (Q7 softclip = insertion at running offset 0,
//                                                           Q8 deletion gap inserted at sum(from_length))
//   slicing                     src/update_likelihood.cpp:33-45 (Q6 sizes are per EDIT, indexed per MAPPING)
//   read filter                 src/HaploCart.cpp:410       (identity < 1e-10 skipped)
// Reads on which the reference would std::terminate (unknown node, substr/insert out of range, a mapping
// without edits) are skipped and counted instead.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>

using namespace vgan;

void vgan_hc_host_batch::fill(vgan_hc_batch *b) const {
    memset(b, 0, sizeof *b);
    b->n_reads = (uint32_t)read_algn_len.size();
    b->n_segments = (uint32_t)seg_node.size();
    b->n_cols = graph_seq.size();
    b->n_qual = qual.size();
    b->read_seg_off = read_seg_off.data();
    b->read_col_off = read_col_off.data();
    b->read_qual_off = read_qual_off.data();
    b->read_algn_len = read_algn_len.data();
    b->read_mapq = read_mapq.data();
    b->seg_node = seg_node.data();
    b->seg_start = seg_start.data();
    b->seg_len = seg_len.data();
    b->graph_seq = graph_seq.data();
    b->algnseq = algnseq.data();
    b->qual = qual.data();
    b->on_device = 0;
    b->n_tileable = n_tileable;
    b->read_src = read_src.size() == read_algn_len.size() ? read_src.data() : nullptr;
}

namespace {

enum { BAD_NODE = 1, BAD_SUBSTR = 2, BAD_SIZES = 3, BAD_TABLE = 4, BAD_RANGE = 5 };

inline char comp(char c) {
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

// appends oriented_node[off, off+n) (clamped like std::string::substr); false if off > node length
inline bool append_node(const vgan_graph &g, int64_t id, bool rev, int64_t off, int64_t n, std::string &out,
                        int64_t *appended) {
    const int64_t len = g.seq_len(id);
    if (off > len || off < 0) return false;
    if (n < 0) n = len - off; // substr(pos, npos-like) for negative counts converted to size_t
    n = std::min(n, len - off);
    const char *s = g.seq_ptr(id);
    if (!rev) {
        out.append(s + off, (size_t)n);
    } else {
        for (int64_t k = 0; k < n; ++k) out += comp(s[len - 1 - (off + k)]);
    }
    if (appended) *appended = n;
    return true;
}

} // namespace

// a1, shared by the HaploCart / euka / soibean front halves.  Returns 0 or a BAD_* code.
namespace {
// The common read: every edit a match or a substitution (from_length == to_length).  Then path_string and graph_seq grow
// side by side and nothing is ever inserted into path_string, so one walk over the mappings writes both through plain
// pointers into strings sized once.  Anything else -- an indel, an edit of unequal lengths, an offset past its node, an unknown
// node -- returns false and the general walk below decides (and reports) as before.
bool reconstruct_matches_only(const vgan_graph &g, const vgan_alnset &a, int64_t m0, int64_t m1, Recon &o) {
    const int64_t e0 = a.edit_off[m0], e1 = a.edit_off[m1];
    int64_t tot_from = 0, tot_ps = 0;
    for (int64_t e = e0; e < e1; ++e) {
        const int32_t from = a.e_from[e];
        if (from != a.e_to[e] || from < 0) return false;
        const int64_t sl = a.e_seq_off[e + 1] - a.e_seq_off[e];
        tot_from += from;
        tot_ps += sl > 0 ? sl : from;
    }
    o.gseq.resize((size_t)tot_from);
    o.ps.resize((size_t)tot_ps);
    o.sizes.resize((size_t)(e1 - e0));
    char *gq = &o.gseq[0], *pq = &o.ps[0];
    int32_t *sz = o.sizes.data();
    for (int64_t m = m0; m < m1; ++m) {
        const int64_t id = a.m_node[m];
        if (!g.has_node(id)) return false;
        const int64_t len = g.seq_len(id);
        const char *ns = g.seq_ptr(id);
        const bool rev = a.m_rev[m];
        int64_t off = a.m_offset[m];
        if (off != (int64_t)(int32_t)off) return false; // (the two walks of the general form read the offset through different types)
        for (int64_t e = a.edit_off[m]; e < a.edit_off[m + 1]; ++e) {
            const int64_t from = a.e_from[e];
            const int64_t sl = a.e_seq_off[e + 1] - a.e_seq_off[e];
            if (off > len || off < 0) return false;
            const int64_t n = std::min(from, len - off);
            // (a node of this graph holds one or two bases: byte loops, not calls)
            if (!rev) {
                const char *src = ns + off;
                for (int64_t k = 0; k < n; ++k) gq[k] = src[k];
            } else {
                for (int64_t k = 0; k < n; ++k) gq[k] = comp(ns[len - 1 - (off + k)]);
            }
            if (sl > 0) {
                const char *src = a.e_seq.data() + a.e_seq_off[e];
                for (int64_t k = 0; k < sl; ++k) pq[k] = src[k];
                pq += sl;
            } else {
                for (int64_t k = 0; k < n; ++k) pq[k] = gq[k];
                pq += n;
            }
            gq += n;
            *sz++ = (int32_t)n;
            off += from;
        }
    }
    o.gseq.resize((size_t)(gq - o.gseq.data()));
    o.ps.resize((size_t)(pq - o.ps.data()));
    return true;
}
} // namespace

int vgan::reconstruct(const vgan_graph &g, const vgan_alnset &a, int64_t r, Recon &o) {
    const int64_t m0 = a.map_off[r], m1 = a.map_off[r + 1];
    if (m1 > m0 && reconstruct_matches_only(g, a, m0, m1, o)) return 0;
    o.gseq.clear();
    o.ps.clear();
    o.sizes.clear();
    // path_string: node bases for matches, edit.sequence for substitutions/insertions
    for (int64_t m = m0; m < m1; ++m) {
        const int64_t id = a.m_node[m];
        if (!g.has_node(id)) return BAD_NODE;
        int64_t f = a.m_offset[m];
        for (int64_t e = a.edit_off[m]; e < a.edit_off[m + 1]; ++e) {
            const int32_t from = a.e_from[e], to = a.e_to[e];
            const int64_t sl = a.e_seq_off[e + 1] - a.e_seq_off[e];
            if (from == to && sl == 0) {
                if (!append_node(g, id, a.m_rev[m], f, from, o.ps, nullptr)) return BAD_SUBSTR;
            } else if ((from == to && sl > 0) || (from == 0 && to > 0 && sl > 0)) {
                o.ps.append(a.e_seq.data() + a.e_seq_off[e], (size_t)sl);
            }
            f += from;
        }
    }
    int64_t f = 0;
    for (int64_t m = m0; m < m1; ++m) {
        const int64_t id = a.m_node[m];
        const bool rev = a.m_rev[m];
        int64_t offset = (int32_t)a.m_offset[m];
        int32_t aligned = 0;
        for (int64_t e = a.edit_off[m]; e < a.edit_off[m + 1]; ++e) {
            const int32_t from = a.e_from[e], to = a.e_to[e];
            const int64_t sl = a.e_seq_off[e + 1] - a.e_seq_off[e];
            const bool is_ins = from == 0 && to > 0 && sl > 0;
            if (from == to) { // match or substitution
                int64_t n;
                if (!append_node(g, id, rev, offset, from, o.gseq, &n)) return BAD_SUBSTR;
                aligned = (int32_t)n;
            } else if (is_ins) {
                o.gseq.append((size_t)to, offset == 0 ? 'S' : '-'); // Q7
                aligned = to;
            } else if (from > 0 && to == 0) { // deletion
                int64_t n;
                if (!append_node(g, id, rev, offset, from, o.gseq, &n)) return BAD_SUBSTR;
                aligned = (int32_t)n;
                if (f < 0 || (size_t)f > o.ps.size()) return BAD_SUBSTR;
                o.ps.insert((size_t)f, (size_t)from, '-'); // Q8
            }
            offset += from;
            f += from;
            o.sizes.push_back(aligned);
        }
    }
    return 0;
}

namespace {

// per-read limits of the device's LDS-tiled kernel (hc_device.h keeps the same numbers)
constexpr size_t TILE_MAX_COLS = 1280, TILE_MAX_QUAL = 1280, TILE_MAX_SEGS = 512;

// the tileable reads of a chunk in the layout the segment kernel streams (include/vgan_gpu.h: vgan_hc_packed_view), offsets
// local to the chunk, the read index of the segment records still open (the merge knows a read's place in the batch)
struct PkChunk {
    BigVec<uint32_t> seg_off{0}, col_off{0}, qual_off{0};
    BigVec<uint32_t> am;   // |algnseq| | mapq << 16
    BigVec<uint32_t> src;
    BigVec<uint32_t> srec; // {node, start} pairs
    BigVec<uint32_t> crec;
    BigVec<uint8_t> qual;
};

struct Chunk {
    vgan_hc_host_batch b;   // reads that satisfy the tile contract
    PkChunk pk;             // ... or, for a packed batch, the same reads in the kernel's own layout
    vgan_hc_host_batch gen; // the others: indels / soft clips (|graph_seq| != |algnseq|, segments may overlap), long reads
    std::vector<uint32_t> key; // per read of b: its lowest node id (the merged batch is ordered by it)
    uint32_t max_span = 0;     // largest (highest - lowest node id) of a tileable read
    vgan_hc_flatten_stats st{};
};

// Order of the tileable reads in a batch: the reads of mapping quality VGAN_HC_MAPQ_MAJOR by their lowest node id, then the
// others by theirs (include/vgan_gpu.h: a tile of the segment kernel whose reads share that mapping quality takes its column
// terms from a table).  A read without mappings (no lowest node) sorts first.
constexpr uint32_t KEY_MINOR = 0x40000000u;
inline uint32_t sort_key(uint32_t min_node, int32_t mapq) {
    const uint32_t n = min_node == 0xFFFFFFFFu ? 0u : std::min(min_node, KEY_MINOR - 1u);
    return n | (mapq == VGAN_HC_MAPQ_MAJOR ? 0u : KEY_MINOR);
}

struct SegTmp {
    uint32_t node;
    uint16_t start, len;
};

void flatten_range(const vgan_graph &g, const vgan_alnset &a, int64_t r0, int64_t r1, const uint8_t *skip, int64_t src_base,
                   bool packed, Chunk &c) {
    Recon rc;
    std::vector<SegTmp> seg_tmp;
    if (packed) { // size the chunk's arrays
        auto &k = c.pk;
        const size_t nr = (size_t)(r1 - r0), nm = (size_t)(a.map_off[r1] - a.map_off[r0]);
        const size_t nb = (size_t)(a.seq_off[r1] - a.seq_off[r0]), nq = (size_t)(a.qual_off[r1] - a.qual_off[r0]);
        k.seg_off.reserve(nr + 1);
        k.col_off.reserve(nr + 1);
        k.qual_off.reserve(nr + 1);
        k.am.reserve(nr);
        k.src.reserve(nr);
        k.srec.reserve(2 * nm);
        k.crec.reserve(nb + nb / 16 + 64);
        k.qual.reserve(nq);
    } else {
        auto &b = c.b; // from the input volume, so that they grow at most once or twice
        const size_t nr = (size_t)(r1 - r0), nm = (size_t)(a.map_off[r1] - a.map_off[r0]);
        const size_t nb = (size_t)(a.seq_off[r1] - a.seq_off[r0]), nq = (size_t)(a.qual_off[r1] - a.qual_off[r0]);
        b.read_seg_off.reserve(nr + 1);
        b.read_col_off.reserve(nr + 1);
        b.read_qual_off.reserve(nr + 1);
        b.read_algn_len.reserve(nr);
        b.read_mapq.reserve(nr);
        b.read_src.reserve(nr);
        b.seg_node.reserve(nm);
        b.seg_start.reserve(nm);
        b.seg_len.reserve(nm);
        b.graph_seq.reserve(nb + nb / 16 + 64);
        b.algnseq.reserve(nb + nb / 16 + 64);
        b.qual.reserve(nq);
    }
    for (int64_t r = r0; r < r1; ++r) {
        if (skip && skip[r]) continue;
        c.st.n_in++;
        if (a.identity[r] < 1e-10) {
            c.st.n_unmapped++;
            continue;
        }
        int bad = reconstruct(g, a, r, rc);
        const int64_t nm = a.map_off[r + 1] - a.map_off[r];
        const size_t A = rc.ps.size(), G = rc.gseq.size();
        const size_t n_qual_r = (size_t)(a.qual_off[r + 1] - a.qual_off[r]);
        // 16-bit per-read positions; the quality string is parsed independently of |sequence| (gam.cpp) and the general
        // kernel keeps one prefix sum per 64 quality bytes for at most 65536 of them
        if (!bad && (A > 65535 || G > 65535 || nm > 65535 || n_qual_r > 65535)) bad = BAD_RANGE;
        // segments first (into scratch), then the route: the tile contract also wants every segment to score a column
        seg_tmp.clear();
        bool empty_seg = false;
        uint32_t min_node = 0xFFFFFFFFu, max_node = 0u;
        if (!bad) {
            size_t pos = 0;
            for (int64_t i = 0; i < nm; ++i) {
                if ((size_t)i >= rc.sizes.size()) {
                    bad = BAD_SIZES;
                    break;
                }
                if (pos > G || pos > A) {
                    bad = BAD_SUBSTR;
                    break;
                }
                const int64_t id = a.m_node[a.map_off[r] + i];
                const int32_t pb = g.pangenome_base[id]; // id validated by reconstruct()
                if (pb < 0) {
                    bad = BAD_NODE;
                    break;
                }
                if ((uint64_t)pb >= g.mappability.size()) {
                    bad = BAD_TABLE;
                    break;
                }
                const size_t n = (size_t)std::max(0, rc.sizes[i]);
                const size_t sl = std::min(n, G - pos);
                seg_tmp.push_back({(uint32_t)id, (uint16_t)pos, (uint16_t)sl});
                empty_seg |= sl == 0;
                min_node = std::min(min_node, (uint32_t)id);
                max_node = std::max(max_node, (uint32_t)id);
                pos += std::min(n, A - pos);
            }
        }
        if (bad) {
            c.st.n_bad++;
            continue;
        }
        // (|quality| <= |algnseq|: the packed layout carries the quality string in the column records, include/vgan_gpu.h)
        // (a packed batch: one word a mapping, include/vgan_gpu.h VGAN_HC_SREC -- a graph with node ids beyond its 18 bits hands every
        // read over in the SoA form)
        const bool tile = A == G && A <= TILE_MAX_COLS && n_qual_r <= TILE_MAX_QUAL && n_qual_r <= A && (size_t)nm <= TILE_MAX_SEGS && !empty_seg &&
                          !seg_tmp.empty() && (!packed || max_node <= VGAN_HC_SREC_MAX_NODE);
        int32_t mq = a.mapq[r];
        if (mq < 0 || mq > 99) {
            mq = mq < 0 ? 0 : 99;
            c.st.n_clamped++;
        }
        if (tile) c.max_span = std::max(c.max_span, max_node - min_node);
        if (tile && packed) {
            // The kernel's own layout, written here once (bytes are moved, nothing is compared, clamped or looked up): a record
            // per alignment column {graph byte, the read byte update_likelihood.cpp:46 pairs it with -- algnseq from the READ
            // start --, quality byte by column (0 past the string), VGAN_HC_CREC_HEAD on a mapping's first column}; a column no
            // mapping scores keeps its quality byte alone
            auto &k = c.pk;
            const size_t c0 = k.crec.size();
            k.crec.resize(c0 + A);
            uint32_t *cr = k.crec.data() + c0;
            const uint8_t *gs = reinterpret_cast<const uint8_t *>(rc.gseq.data()), *rs = reinterpret_cast<const uint8_t *>(rc.ps.data());
            const uint8_t *q = reinterpret_cast<const uint8_t *>(a.qual.data() + a.qual_off[r]);
            for (size_t col = 0; col < A; ++col) cr[col] = (col < n_qual_r ? (uint32_t)q[col] : 0u) << 16;
            for (const SegTmp &sg : seg_tmp) {
                k.srec.push_back(sg.node);
                k.srec.push_back(sg.start);
                const size_t st = sg.start, ln = sg.len;
                for (size_t j = 0; j < ln; ++j) cr[st + j] |= (uint32_t)gs[st + j] | ((uint32_t)rs[j] << 8);
                cr[st] |= VGAN_HC_CREC_HEAD;
            }
            k.qual.insert(k.qual.end(), q, q + n_qual_r);
            k.am.push_back((uint32_t)A | ((uint32_t)mq << 16));
            k.src.push_back((uint32_t)(r + src_base));
            k.seg_off.push_back((uint32_t)(k.srec.size() / 2));
            k.col_off.push_back((uint32_t)k.crec.size());
            k.qual_off.push_back((uint32_t)k.qual.size());
            c.key.push_back(sort_key(min_node, mq));
            c.st.n_out++;
            continue;
        }
        auto &b = tile ? c.b : c.gen;
        for (const SegTmp &sg : seg_tmp) {
            b.seg_node.push_back(sg.node);
            b.seg_start.push_back(sg.start);
            b.seg_len.push_back(sg.len);
        }
        if (tile) c.key.push_back(sort_key(min_node, mq));
        const size_t region = std::max(A, G);
        b.graph_seq.insert(b.graph_seq.end(), rc.gseq.begin(), rc.gseq.end());
        b.graph_seq.insert(b.graph_seq.end(), region - G, 0);
        b.algnseq.insert(b.algnseq.end(), rc.ps.begin(), rc.ps.end());
        b.algnseq.insert(b.algnseq.end(), region - A, 0);
        const char *q = a.qual.data() + a.qual_off[r];
        b.qual.insert(b.qual.end(), q, q + (a.qual_off[r + 1] - a.qual_off[r]));
        b.read_algn_len.push_back((uint16_t)A);
        b.read_mapq.push_back((uint8_t)mq);
        b.read_src.push_back((uint32_t)(r + src_base));
        b.read_seg_off.push_back((uint32_t)b.seg_node.size());
        b.read_col_off.push_back((uint32_t)b.graph_seq.size());
        b.read_qual_off.push_back((uint32_t)b.qual.size());
        c.st.n_out++;
    }
}

template <class T> void append_shifted(std::vector<T> &dst, const std::vector<T> &src, T shift) {
    // src[0] == 0 is the leading offset; skip it
    for (size_t i = 1; i < src.size(); ++i) dst.push_back((T)(src[i] + shift));
}

} // namespace

namespace {
// chunks -> one batch (chunks are consumed): the tileable reads of all chunks ordered by their lowest node id -- the
// segment kernels keep W[node] of a wave's (workgroup's) reads in an LDS window, which wants neighbouring reads on
// neighbouring nodes (a stable counting sort, so reads on the same node keep their input order) -- then every chunk's other
// reads.  `packed`: the tileable reads arrive, and leave, in the segment kernel's own layout (res->pk_*), and the SoA arrays
// hold the other reads alone (a batch of its own, offsets from 0).
int merge_chunks(std::vector<Chunk> &chunks, bool packed, PhaseTimer &pt, vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats) {
    auto res = new vgan_hc_host_batch();
    vgan_hc_flatten_stats st{};
    const size_t nc = chunks.size();
    // ---- the tileable reads: (chunk, local index) in sorted order
    std::vector<size_t> rbase(nc + 1, 0);
    for (size_t i = 0; i < nc; ++i) rbase[i + 1] = rbase[i] + chunks[i].key.size();
    const size_t nt_reads = rbase[nc];
    uint32_t kmax = 0; // (the keys' node part; the counting sort's bins: the major reads' nodes, then the others')
    for (auto &c : chunks)
        for (uint32_t k : c.key) kmax = std::max(kmax, k & (KEY_MINOR - 1u));
    auto bin = [kmax](uint32_t k) { return (size_t)(k & (KEY_MINOR - 1u)) + ((k & KEY_MINOR) ? (size_t)kmax + 1 : 0); };
    std::vector<uint32_t> order(nt_reads); // output position -> global tileable index (chunk-major)
    {
        std::vector<uint32_t> cnt(2 * ((size_t)kmax + 1) + 1, 0);
        for (auto &c : chunks)
            for (uint32_t k : c.key) cnt[bin(k) + 1]++;
        for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];
        for (size_t i = 0; i < nc; ++i)
            for (size_t j = 0; j < chunks[i].key.size(); ++j) order[cnt[bin(chunks[i].key[j])]++] = (uint32_t)(rbase[i] + j);
    }
    // ---- totals
    uint64_t t_cols = 0, t_segs = 0, t_qual = 0;
    for (auto &c : chunks) {
        t_cols += packed ? c.pk.crec.size() : c.b.graph_seq.size();
        t_segs += packed ? c.pk.srec.size() / 2 : c.b.seg_node.size();
        t_qual += packed ? c.pk.qual.size() : c.b.qual.size();
    }
    struct Base {
        size_t r, s, c, q;
    };
    std::vector<Base> gbase(nc);
    // (a packed batch keeps its SoA arrays for the other reads alone)
    uint64_t tot_reads = packed ? 0 : nt_reads, tot_cols = packed ? 0 : t_cols, tot_segs = packed ? 0 : t_segs, tot_qual = packed ? 0 : t_qual;
    for (size_t i = 0; i < nc; ++i) {
        gbase[i] = {(size_t)tot_reads, (size_t)tot_segs, (size_t)tot_cols, (size_t)tot_qual};
        tot_cols += chunks[i].gen.graph_seq.size();
        tot_segs += chunks[i].gen.seg_node.size();
        tot_qual += chunks[i].gen.qual.size();
        tot_reads += chunks[i].gen.read_mapq.size();
    }
    if (tot_cols > 0xFFFFFFF0ull || tot_segs > 0xFFFFFFF0ull || tot_qual > 0xFFFFFFF0ull || tot_reads > 0xFFFFFFF0ull ||
        t_cols > 0xFFFFFFF0ull || t_segs > 0xFFFFFFF0ull || t_qual > 0xFFFFFFF0ull) {
        delete res;
        return fail(VGAN_ERANGE, "vgan_hc_flatten: batch exceeds 32-bit offsets; flatten fewer reads per batch");
    }
    res->n_tileable = packed ? 0 : (uint32_t)nt_reads;
    res->read_seg_off.resize(tot_reads + 1);
    res->read_col_off.resize(tot_reads + 1);
    res->read_qual_off.resize(tot_reads + 1);
    res->read_algn_len.resize(tot_reads);
    res->read_mapq.resize(tot_reads);
    res->read_src.resize(tot_reads);
    res->seg_node.resize(tot_segs);
    res->seg_start.resize(tot_segs);
    res->seg_len.resize(tot_segs);
    res->graph_seq.resize(tot_cols);
    res->algnseq.resize(tot_cols);
    res->qual.resize(tot_qual);
    res->read_seg_off[0] = res->read_col_off[0] = res->read_qual_off[0] = 0;
    auto chunk_of = [&](uint32_t gidx) { return (size_t)(std::upper_bound(rbase.begin(), rbase.end(), (size_t)gidx) - rbase.begin()) - 1; };
    // ---- offsets of the sorted reads (serial prefix sums over three lengths per read)
    std::vector<uint32_t> pseg, pcol, pqual; // (packed: the sorted reads' offsets)
    if (packed) {
        res->is_packed = true;
        res->pk_reads = (uint32_t)nt_reads;
        res->pk_segments = (uint32_t)t_segs;
        res->pk_cols = t_cols;
        res->pk_qual = t_qual;
        res->pk_rhdr.resize(4 * (nt_reads + 1));
        res->pk_srec.resize(t_segs);
        res->pk_crec.resize(t_cols);
        res->pk_qualp.resize(t_qual + 32);
        res->pk_src.resize(nt_reads);
        pseg.resize(nt_reads + 1);
        pcol.resize(nt_reads + 1);
        pqual.resize(nt_reads + 1);
        pseg[0] = pcol[0] = pqual[0] = 0;
        uint32_t ms = 0, mq = 0, mc = 0;
        for (size_t o = 0; o < nt_reads; ++o) {
            const size_t ci = chunk_of(order[o]), j = order[o] - rbase[ci];
            const auto &k = chunks[ci].pk;
            const uint32_t ns = k.seg_off[j + 1] - k.seg_off[j], ncl = k.col_off[j + 1] - k.col_off[j], nq = k.qual_off[j + 1] - k.qual_off[j];
            pseg[o + 1] = pseg[o] + ns;
            pcol[o + 1] = pcol[o] + ncl;
            pqual[o + 1] = pqual[o] + nq;
            ms = std::max(ms, ns);
            mq = std::max(mq, nq);
            mc = std::max(mc, ncl);
        }
        res->pk_max_segs = ms;
        res->pk_max_qual = mq;
        res->pk_max_cols = mc;
        for (auto &c : chunks) res->pk_max_span = std::max(res->pk_max_span, c.max_span);
        uint32_t *h = res->pk_rhdr.data() + 4 * nt_reads; // the end offsets
        h[0] = (uint32_t)t_segs;
        h[1] = (uint32_t)t_qual;
        h[2] = (uint32_t)t_cols;
        h[3] = 0;
        memset(res->pk_qualp.data() + t_qual, 0, 32);
    } else {
        for (size_t o = 0; o < nt_reads; ++o) {
            const size_t ci = chunk_of(order[o]), j = order[o] - rbase[ci];
            const auto &cb = chunks[ci].b;
            res->read_seg_off[o + 1] = res->read_seg_off[o] + (cb.read_seg_off[j + 1] - cb.read_seg_off[j]);
            res->read_col_off[o + 1] = res->read_col_off[o] + (cb.read_col_off[j + 1] - cb.read_col_off[j]);
            res->read_qual_off[o + 1] = res->read_qual_off[o] + (cb.read_qual_off[j + 1] - cb.read_qual_off[j]);
        }
    }
    const size_t hw = nt_reads <= 300000 ? burst_cpus() : usable_cpus();
    // ---- the sorted reads' data, by output range
    auto copy_sorted = [&](size_t o0, size_t o1) {
        for (size_t o = o0; o < o1; ++o) {
            const size_t ci = chunk_of(order[o]), j = order[o] - rbase[ci];
            const auto &cb = chunks[ci].b;
            res->read_algn_len[o] = cb.read_algn_len[j];
            res->read_mapq[o] = cb.read_mapq[j];
            res->read_src[o] = cb.read_src[j];
            const size_t s0 = cb.read_seg_off[j], ns = cb.read_seg_off[j + 1] - s0, so = res->read_seg_off[o];
            if (ns) {
                memcpy(&res->seg_node[so], &cb.seg_node[s0], ns * sizeof(uint32_t));
                memcpy(&res->seg_start[so], &cb.seg_start[s0], ns * sizeof(uint16_t));
                memcpy(&res->seg_len[so], &cb.seg_len[s0], ns * sizeof(uint16_t));
            }
            const size_t c0 = cb.read_col_off[j], ncol = cb.read_col_off[j + 1] - c0, co = res->read_col_off[o];
            if (ncol) {
                memcpy(&res->graph_seq[co], &cb.graph_seq[c0], ncol);
                memcpy(&res->algnseq[co], &cb.algnseq[c0], ncol);
            }
            const size_t q0 = cb.read_qual_off[j], nq = cb.read_qual_off[j + 1] - q0;
            if (nq) memcpy(&res->qual[res->read_qual_off[o]], &cb.qual[q0], nq);
        }
    };
    auto copy_sorted_packed = [&](size_t o0, size_t o1) {
        for (size_t o = o0; o < o1; ++o) {
            const size_t ci = chunk_of(order[o]), j = order[o] - rbase[ci];
            const auto &k = chunks[ci].pk;
            uint32_t *h = res->pk_rhdr.data() + 4 * o;
            h[0] = pseg[o];
            h[1] = pqual[o];
            h[2] = pcol[o];
            h[3] = k.am[j];
            res->pk_src[o] = k.src[j];
            const size_t s0 = k.seg_off[j], ns = k.seg_off[j + 1] - s0;
            const uint32_t *sp = k.srec.data() + 2 * s0;
            uint32_t *sd = res->pk_srec.data() + (size_t)pseg[o];
            for (size_t t = 0; t < ns; ++t) sd[t] = VGAN_HC_SREC(sp[2 * t], sp[2 * t + 1], o); // (the read's index & 7: as the kernel finds it within a tile)
            const size_t c0 = k.col_off[j], ncol = k.col_off[j + 1] - c0;
            if (ncol) memcpy(res->pk_crec.data() + pcol[o], k.crec.data() + c0, ncol * sizeof(uint32_t));
            const size_t q0 = k.qual_off[j], nq = k.qual_off[j + 1] - q0;
            if (nq) memcpy(res->pk_qualp.data() + pqual[o], k.qual.data() + q0, nq);
        }
    };
    // ---- the other reads: whole parts, offsets shifted
    auto copy_gen = [&](size_t i) {
        auto &cb = chunks[i].gen;
        const Base &bs = gbase[i];
        for (size_t k = 1; k < cb.read_seg_off.size(); ++k) {
            res->read_seg_off[bs.r + k] = cb.read_seg_off[k] + (uint32_t)bs.s;
            res->read_col_off[bs.r + k] = cb.read_col_off[k] + (uint32_t)bs.c;
            res->read_qual_off[bs.r + k] = cb.read_qual_off[k] + (uint32_t)bs.q;
        }
        auto cp = [](auto &dst, size_t at, const auto &src) {
            if (!src.empty()) memcpy(&dst[at], src.data(), src.size() * sizeof(src[0]));
        };
        cp(res->read_algn_len, bs.r, cb.read_algn_len);
        cp(res->read_mapq, bs.r, cb.read_mapq);
        cp(res->read_src, bs.r, cb.read_src);
        cp(res->seg_node, bs.s, cb.seg_node);
        cp(res->seg_start, bs.s, cb.seg_start);
        cp(res->seg_len, bs.s, cb.seg_len);
        cp(res->graph_seq, bs.c, cb.graph_seq);
        cp(res->algnseq, bs.c, cb.algnseq);
        cp(res->qual, bs.q, cb.qual);
    };
    {
        const size_t nth = std::max<size_t>(1, std::min<size_t>(hw, (nt_reads + 8191) / 8192));
        auto sorted = [&](size_t o0, size_t o1) {
            if (packed) copy_sorted_packed(o0, o1);
            else copy_sorted(o0, o1);
        };
        if (nth <= 1) {
            sorted(0, nt_reads);
            for (size_t i = 0; i < nc; ++i) copy_gen(i);
        } else {
            parallel_run((int)nth, [&](int ti) {
                const size_t t = (size_t)ti;
                sorted(nt_reads * t / nth, nt_reads * (t + 1) / nth);
                for (size_t i = t; i < nc; i += nth) copy_gen(i);
            });
        }
    }
    for (auto &c : chunks) {
        st.n_in += c.st.n_in;
        st.n_out += c.st.n_out;
        st.n_unmapped += c.st.n_unmapped;
        st.n_bad += c.st.n_bad;
        st.n_clamped += c.st.n_clamped;
        c = Chunk();
    }
    pt.lap("merge");
    st.n_segments = (int64_t)res->seg_node.size() + (int64_t)res->pk_segments;
    st.n_cols = (int64_t)res->graph_seq.size() + (int64_t)res->pk_cols;
    if (stats) *stats = st;
    *out = res;
    return VGAN_OK;
}

int flatten_set(const vgan_graph *g, const vgan_alnset *a, int64_t r0, int64_t r1, const uint8_t *skip, int n_threads, bool packed,
                vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats) {
    if (!g || !a || !out) return fail(VGAN_EINVAL, "vgan_hc_flatten: null argument");
    if (r0 < 0 || r1 > a->n_reads() || r0 > r1) return fail(VGAN_EINVAL, "vgan_hc_flatten: bad read range");
    if (n_threads <= 0) n_threads = (int)(r1 - r0 <= 300000 ? burst_cpus() : usable_cpus());
    const int64_t n = r1 - r0;
    // every chunk allocates a dozen arrays of a few hundred KB (one mmap each in glibc) and faults them in: beyond ~32
    // threads the address-space lock, not the cores, sets the pace (1M reads on 256 cores: 8 threads 0.34 s, 32: 0.08 s,
    // 64: 0.13 s, 256: 0.21 s for the chunk phase)
    n_threads = std::min(n_threads, 40);
    n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, (n + 4095) / 4096));
    PhaseTimer pt("hc_flatten");
    std::vector<Chunk> chunks((size_t)n_threads);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) {
        const int64_t b0 = r0 + n * t / n_threads, b1 = r0 + n * (t + 1) / n_threads;
        if (n_threads == 1) flatten_range(*g, *a, b0, b1, skip, 0, packed, chunks[t]);
        else th.emplace_back(flatten_range, std::cref(*g), std::cref(*a), b0, b1, skip, (int64_t)0, packed, std::ref(chunks[t]));
    }
    for (auto &t : th) t.join();
    pt.lap("chunks");
    return merge_chunks(chunks, packed, pt, out, stats);
}

int flatten_parts(const vgan_graph *g, const vgan_alnparts *ps, int64_t part0, int64_t part1, const uint8_t *skip, int n_threads,
                  bool packed, vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats) {
    if (!g || !ps || !out) return fail(VGAN_EINVAL, "vgan_hc_flatten_parts: null argument");
    if (part0 < 0 || part1 > (int64_t)ps->parts.size() || part0 > part1) return fail(VGAN_EINVAL, "vgan_hc_flatten_parts: bad slice range");
    if (ps->base + ps->first.back() > 0xFFFFFFF0ll) return fail(VGAN_ERANGE, "vgan_hc_flatten_parts: more than 2^32 reads");
    if (n_threads <= 0) n_threads = (int)usable_cpus();
    // Work items: sub-ranges of the slices (a slice is 8192 reads: with one item per slice a chunk of 32 slices keeps 32 threads
    // busy for one slice's time each, whatever the machine)
    constexpr int64_t SUB = 2048;
    struct Item {
        size_t part;
        int64_t r0, r1;
    };
    std::vector<Item> items;
    for (int64_t pi = part0; pi < part1; ++pi) {
        const int64_t n = ps->parts[(size_t)pi].n_reads();
        for (int64_t r0 = 0; r0 < n; r0 += SUB) items.push_back({(size_t)pi, r0, std::min(n, r0 + SUB)});
    }
    const size_t np = items.size();
    n_threads = std::min(n_threads, 40); // as in flatten_set
    n_threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, np));
    PhaseTimer pt("hc_flatten_parts");
    std::vector<Chunk> chunks(np); // one per item, in input order
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= np) break;
            const Item &it = items[i];
            const vgan_alnset &a = ps->parts[it.part];
            flatten_range(*g, a, it.r0, it.r1, skip ? skip + ps->first[it.part] : nullptr, ps->base + ps->first[it.part], packed, chunks[i]);
        }
    };
    parallel_run(n_threads, [&](int) {
        const double c0 = pt.on ? thread_cpu_ms() : 0;
        work();
        if (pt.on) cpu_account().flatten += (int64_t)((thread_cpu_ms() - c0) * 1e3);
    });
    pt.lap("chunks");
    const double m0 = pt.on ? thread_cpu_ms() : 0;
    const int rc = merge_chunks(chunks, packed, pt, out, stats);
    if (pt.on) cpu_account().merge += (int64_t)((thread_cpu_ms() - m0) * 1e3); // (the calling thread's share: the serial part)
    return rc;
}
} // namespace

extern "C" int vgan_hc_flatten(const vgan_graph *g, const vgan_alnset *a, int64_t r0, int64_t r1, int n_threads,
                               vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats) {
    return flatten_set(g, a, r0, r1, nullptr, n_threads, false, out, stats);
}

extern "C" int vgan_hc_flatten_masked(const vgan_graph *g, const vgan_alnset *a, int64_t r0, int64_t r1, const uint8_t *skip,
                                      int n_threads, vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats) {
    return flatten_set(g, a, r0, r1, skip, n_threads, false, out, stats);
}

extern "C" int vgan_hc_flatten_parts(const vgan_graph *g, const vgan_alnparts *ps, int64_t part0, int64_t part1, const uint8_t *skip,
                                     int n_threads, vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats) {
    return flatten_parts(g, ps, part0, part1, skip, n_threads, false, out, stats);
}

extern "C" int vgan_hc_flatten_packed(const vgan_graph *g, const vgan_alnset *a, int64_t r0, int64_t r1, const uint8_t *skip,
                                      int n_threads, vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats) {
    return flatten_set(g, a, r0, r1, skip, n_threads, true, out, stats);
}

extern "C" int vgan_hc_flatten_parts_packed(const vgan_graph *g, const vgan_alnparts *ps, int64_t part0, int64_t part1,
                                            const uint8_t *skip, int n_threads, vgan_hc_host_batch **out, vgan_hc_flatten_stats *stats) {
    return flatten_parts(g, ps, part0, part1, skip, n_threads, true, out, stats);
}

extern "C" int vgan_hc_host_batch_get_packed(const vgan_hc_host_batch *b, vgan_hc_packed_view *out) {
    if (!b || !out) return fail(VGAN_EINVAL, "vgan_hc_host_batch_get_packed: null argument");
    memset(out, 0, sizeof *out);
    if (!b->is_packed) return fail(VGAN_ESTATE, "vgan_hc_host_batch_get_packed: the batch was not flattened into the packed layout");
    out->n_reads = b->pk_reads;
    out->n_segments = b->pk_segments;
    out->n_cols = b->pk_cols;
    out->n_qual = b->pk_qual;
    out->rhdr = b->pk_rhdr.data();
    out->srec = b->pk_srec.data();
    out->crec = b->pk_crec.data();
    out->qualp = b->pk_qualp.data();
    out->max_read_segs = b->pk_max_segs;
    out->max_read_qual = b->pk_max_qual;
    out->max_read_cols = b->pk_max_cols;
    out->max_read_node_span = b->pk_max_span;
    out->on_device = 0;
    out->read_src = b->pk_src.data();
    return VGAN_OK;
}

extern "C" int vgan_hc_host_batch_get(const vgan_hc_host_batch *b, vgan_hc_batch *out) {
    if (!b || !out) return fail(VGAN_EINVAL, "vgan_hc_host_batch_get: null argument");
    b->fill(out);
    return VGAN_OK;
}

extern "C" void vgan_hc_host_batch_free(vgan_hc_host_batch *b) { delete b; }

extern "C" int vgan_reconstruct(const vgan_graph *g, const vgan_alnset *a, int64_t r, char *graph_seq, char *read_seq,
                                int32_t *mppg_sizes, int64_t cap, int64_t *lens) {
    if (!g || !a || !graph_seq || !read_seq || !mppg_sizes || !lens) return fail(VGAN_EINVAL, "vgan_reconstruct: null argument");
    if (r < 0 || r >= a->n_reads()) return fail(VGAN_EINVAL, "vgan_reconstruct: read index out of range");
    Recon rc;
    const int bad = reconstruct(*g, *a, r, rc);
    if (bad) return fail(VGAN_ERANGE, "vgan_reconstruct: the reference would terminate on this read (code %d)", bad);
    if ((int64_t)rc.gseq.size() + 1 > cap || (int64_t)rc.ps.size() + 1 > cap || (int64_t)rc.sizes.size() > cap)
        return fail(VGAN_ERANGE, "vgan_reconstruct: buffers too small");
    memcpy(graph_seq, rc.gseq.c_str(), rc.gseq.size() + 1);
    memcpy(read_seq, rc.ps.c_str(), rc.ps.size() + 1);
    std::copy(rc.sizes.begin(), rc.sizes.end(), mppg_sizes);
    lens[0] = (int64_t)rc.gseq.size();
    lens[1] = (int64_t)rc.ps.size();
    lens[2] = (int64_t)rc.sizes.size();
    return VGAN_OK;
}
