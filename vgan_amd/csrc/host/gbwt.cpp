// GBWT reader: path extraction from the compressed threads of a `vg gbwt` index.
//
// The reference loads <dbprefix>.gbwt beside the ODGI graph and walks every path with gbwt->extract(path_id)
// (src/readOG_Euka.h:36-74).  gbwt / sdsl are not in the reference tree, so the layout below is the one observed in its
// fixture test/reconstructInputSeq/target_graph.gbwt (written by `vg gbwt -o`, GBWT file version 4, sdsl serialisation)
// and checked against the P lines of target_graph.gfa; anything deviating from it is refused:
//
//   VPKG framing  groups of { varint count, count items of { varint length, bytes } }, first item the tag "GBWT";
//                 the payload is the concatenation of the other items (a bare GBWT without framing is accepted too)
//   header        u32 tag 0x6B376B37, u32 version, u64 sequences, size, offset, alphabet_size, flags
//                 (flags: 1 bidirectional, 2 metadata, 4 simple-sds -- the last is not read)
//   record array  u64 records (= alphabet_size - offset); sdsl::sd_vector over the record start offsets
//                 (u64 size, u8 low width, int_vector<0> low, bit_vector high, two select_support_mcl); then `size` data
//                 bytes.  Record r belongs to node (r + offset), record 0 to the endmarker.
//   record        ByteCode sigma; sigma x { ByteCode node delta, ByteCode offset } outgoing edges; the body as runs:
//                 sigma < 255: one byte = rank + sigma * (length - 1), a byte holding the largest length (256 / sigma) is
//                 followed by a ByteCode with the rest; else { ByteCode rank, ByteCode length - 1 }
//   the rest      (document-array samples, metadata) is not needed to extract paths and is not parsed.
//
// extract(s): position = LF(endmarker, s); while its node is not the endmarker: emit the node, position = LF(position)
// with LF((v, i)) = (edge[rank].node, edge[rank].offset + occurrences of rank in body[0, i)), rank = body[i].
// Nodes are GBWT encodings: 2 * id + is_reverse; in a bidirectional index sequence 2k is path k forward, 2k+1 reverse.
#include "common.h"

#include <cstring>
#include <memory>

using namespace vgan;

struct vgan_gbwt {
    uint64_t sequences = 0, size = 0, offset = 0, alphabet_size = 0, flags = 0;
    uint32_t version = 0;
    std::vector<uint64_t> rec_start; // [records + 1]
    std::string data;
    struct Rec {
        bool parsed = false;
        std::vector<std::pair<uint64_t, uint64_t>> out;  // (node, offset)
        std::vector<std::pair<uint32_t, uint64_t>> runs; // (rank, length)
        uint64_t body = 0;
    };
    mutable std::vector<Rec> recs;
};

namespace {

struct Cursor {
    const unsigned char *p, *end;
    bool ok = true;
    template <class T> T get() {
        T v{};
        if ((size_t)(end - p) < sizeof(T)) {
            ok = false;
            return v;
        }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    bool skip(uint64_t n) {
        if ((uint64_t)(end - p) < n) return ok = false;
        p += n;
        return true;
    }
};

bool bytecode(const unsigned char *&p, const unsigned char *end, uint64_t &v) {
    v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
        if (p >= end) return false;
        const unsigned char b = *p++;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}

// sdsl::int_vector<0>::serialize: u64 size in bits, u8 width, the bits in 64-bit words
bool read_int_vector(Cursor &c, std::vector<uint64_t> &words, uint64_t &bits, unsigned &width) {
    bits = c.get<uint64_t>();
    width = c.get<uint8_t>();
    if (!c.ok || bits > ((uint64_t)1 << 40) || width > 64) return c.ok = false;
    const uint64_t nw = (bits + 63) / 64;
    if ((uint64_t)(c.end - c.p) < nw * 8) return c.ok = false;
    words.resize(nw);
    if (nw) memcpy(words.data(), c.p, nw * 8);
    c.p += nw * 8;
    return true;
}
// sdsl::bit_vector::serialize: u64 size in bits, the bits in 64-bit words
bool read_bit_vector(Cursor &c, std::vector<uint64_t> &words, uint64_t &bits) {
    bits = c.get<uint64_t>();
    if (!c.ok || bits > ((uint64_t)1 << 40)) return c.ok = false;
    const uint64_t nw = (bits + 63) / 64;
    if ((uint64_t)(c.end - c.p) < nw * 8) return c.ok = false;
    words.resize(nw);
    if (nw) memcpy(words.data(), c.p, nw * 8);
    c.p += nw * 8;
    return true;
}
// sdsl::select_support_mcl::serialize: u64 count; if count: int_vector superblock, bit_vector mini_or_long, then one
// int_vector per superblock of 4096 arguments (a mini block or a long one: the same serialisation)
bool skip_select_mcl(Cursor &c) {
    const uint64_t cnt = c.get<uint64_t>();
    if (!c.ok) return false;
    if (cnt == 0) return true;
    std::vector<uint64_t> w;
    uint64_t bits;
    unsigned width;
    if (!read_int_vector(c, w, bits, width)) return false;
    if (!read_bit_vector(c, w, bits)) return false;
    const uint64_t sb = (cnt + 4095) >> 12;
    for (uint64_t i = 0; i < sb; ++i)
        if (!read_int_vector(c, w, bits, width)) return false;
    return true;
}

bool parse_record(const vgan_gbwt &g, uint64_t r, vgan_gbwt::Rec &rec) {
    const unsigned char *p = (const unsigned char *)g.data.data() + g.rec_start[r];
    const unsigned char *end = (const unsigned char *)g.data.data() + g.rec_start[r + 1];
    rec.out.clear();
    rec.runs.clear();
    rec.body = 0;
    rec.parsed = true;
    if (p == end) return true; // a node no path visits
    uint64_t sigma;
    if (!bytecode(p, end, sigma) || sigma > (uint64_t)(end - p)) return false;
    uint64_t node = 0;
    for (uint64_t k = 0; k < sigma; ++k) {
        uint64_t d, off;
        if (!bytecode(p, end, d) || !bytecode(p, end, off)) return false;
        node += d;
        rec.out.push_back({node, off});
    }
    if (sigma == 0) return p == end;
    const uint64_t run_continues = sigma < 255 ? 256 / sigma : 0;
    while (p < end) {
        uint64_t rank, len;
        if (sigma < 255) {
            const unsigned b = *p++;
            rank = b % sigma;
            len = b / sigma + 1;
            if (len == run_continues) {
                uint64_t more;
                if (!bytecode(p, end, more)) return false;
                len += more;
            }
        } else {
            if (!bytecode(p, end, rank) || !bytecode(p, end, len)) return false;
            len += 1;
        }
        if (rank >= sigma) return false;
        rec.runs.push_back({(uint32_t)rank, len});
        rec.body += len;
    }
    return true;
}

const vgan_gbwt::Rec *record_of(const vgan_gbwt &g, uint64_t node) {
    uint64_t r;
    if (node == 0) r = 0;
    else if (node <= g.offset || node - g.offset >= g.recs.size()) return nullptr;
    else r = node - g.offset;
    vgan_gbwt::Rec &rec = g.recs[r];
    if (!rec.parsed && !parse_record(g, r, rec)) return nullptr;
    return &rec;
}

// (node, offset) -> the next position of the same sequence; false at a malformed record
bool lf(const vgan_gbwt &g, uint64_t node, uint64_t i, uint64_t &next_node, uint64_t &next_off) {
    const vgan_gbwt::Rec *rec = record_of(g, node);
    if (!rec || i >= rec->body) return false;
    std::vector<uint64_t> seen(rec->out.size(), 0);
    uint64_t at = 0;
    for (const auto &run : rec->runs) {
        if (i < at + run.second) {
            next_node = rec->out[run.first].first;
            next_off = rec->out[run.first].second + seen[run.first] + (i - at);
            return true;
        }
        seen[run.first] += run.second;
        at += run.second;
    }
    return false;
}

int load(const std::string &raw, vgan_gbwt &g) {
    std::string payload;
    const unsigned char *b = (const unsigned char *)raw.data();
    const size_t n = raw.size();
    if (n >= 4 && b[0] == 0x37 && b[1] == 0x6B && b[2] == 0x37 && b[3] == 0x6B) {
        payload = raw; // a bare GBWT
    } else { // vg's type-tagged framing
        const unsigned char *p = b, *end = b + n;
        bool tagged = false;
        while (p < end) {
            uint64_t count;
            if (!bytecode(p, end, count) || count == 0 || count > (1u << 30)) return fail(VGAN_EIO, "gbwt: broken framing");
            for (uint64_t k = 0; k < count; ++k) {
                uint64_t len;
                if (!bytecode(p, end, len) || len > (uint64_t)(end - p)) return fail(VGAN_EIO, "gbwt: truncated item");
                if (k == 0) {
                    if (len != 4 || memcmp(p, "GBWT", 4) != 0) return fail(VGAN_EIO, "gbwt: the file is not tagged GBWT");
                    tagged = true;
                } else {
                    payload.append((const char *)p, (size_t)len);
                }
                p += len;
            }
        }
        if (!tagged) return fail(VGAN_EIO, "gbwt: empty file");
    }
    Cursor c{(const unsigned char *)payload.data(), (const unsigned char *)payload.data() + payload.size()};
    const uint32_t tag = c.get<uint32_t>();
    g.version = c.get<uint32_t>();
    g.sequences = c.get<uint64_t>();
    g.size = c.get<uint64_t>();
    g.offset = c.get<uint64_t>();
    g.alphabet_size = c.get<uint64_t>();
    g.flags = c.get<uint64_t>();
    if (!c.ok || tag != 0x6B376B37u) return fail(VGAN_EIO, "gbwt: not a GBWT header");
    if (g.version != 4) return fail(VGAN_EIO, "gbwt: file version %u (only version 4, as written by the vg the reference pins, is read)", g.version);
    if (g.flags & ~(uint64_t)3) return fail(VGAN_EIO, "gbwt: unknown flags %llu (simple-sds files are not read)", (unsigned long long)g.flags);
    if (g.alphabet_size <= g.offset || g.alphabet_size - g.offset > ((uint64_t)1 << 32)) return fail(VGAN_EIO, "gbwt: implausible alphabet");
    const uint64_t records = c.get<uint64_t>();
    if (!c.ok || records != g.alphabet_size - g.offset) return fail(VGAN_EIO, "gbwt: record count does not match the alphabet");
    // the record index spends at least one bit per record: a count the file cannot hold is a corrupt header, not an allocation
    if (records > (uint64_t)(c.end - c.p) * 8) return fail(VGAN_EIO, "gbwt: %llu records in a file of %zu bytes", (unsigned long long)records, payload.size());
    // sd_vector over the record starts
    const uint64_t sd_size = c.get<uint64_t>();
    const unsigned wl = c.get<uint8_t>();
    std::vector<uint64_t> low, high;
    uint64_t low_bits, high_bits;
    unsigned low_w;
    if (!c.ok || wl > 63 || !read_int_vector(c, low, low_bits, low_w) || !read_bit_vector(c, high, high_bits) || !skip_select_mcl(c) ||
        !skip_select_mcl(c))
        return fail(VGAN_EIO, "gbwt: broken record index");
    if (low_bits != records * wl || (wl && low_w != wl)) return fail(VGAN_EIO, "gbwt: record index of another shape");
    if ((uint64_t)(c.end - c.p) < sd_size) return fail(VGAN_EIO, "gbwt: truncated record data");
    g.data.assign((const char *)c.p, (size_t)sd_size);
    g.rec_start.clear();
    {
        uint64_t j = 0;
        for (uint64_t pos = 0; pos < high_bits && j < records; ++pos) {
            if (!((high[pos >> 6] >> (pos & 63)) & 1)) continue;
            uint64_t lo = 0;
            if (wl) {
                const uint64_t bit = j * wl, w = bit >> 6, sh = bit & 63;
                lo = low[w] >> sh;
                if (sh + wl > 64) lo |= low[w + 1] << (64 - sh);
                lo &= ((uint64_t)1 << wl) - 1;
            }
            g.rec_start.push_back(((pos - j) << wl) | lo);
            ++j;
        }
        if (j != records) return fail(VGAN_EIO, "gbwt: record index holds %llu starts for %llu records", (unsigned long long)j, (unsigned long long)records);
    }
    g.rec_start.push_back(sd_size);
    for (uint64_t r = 0; r < records; ++r)
        if (g.rec_start[r] > g.rec_start[r + 1] || g.rec_start[r + 1] > sd_size) return fail(VGAN_EIO, "gbwt: record starts do not ascend");
    g.recs.assign(records, vgan_gbwt::Rec());
    // every record up front: the bodies together are the whole BWT (header.size positions), which also bounds every walk
    uint64_t positions = 0;
    for (uint64_t r = 0; r < records; ++r) {
        if (!parse_record(g, r, g.recs[r])) return fail(VGAN_EIO, "gbwt: record %llu does not decode", (unsigned long long)r);
        if (g.recs[r].body > g.size) return fail(VGAN_EIO, "gbwt: record %llu is longer than the index", (unsigned long long)r);
        positions += g.recs[r].body;
    }
    if (positions != g.size) return fail(VGAN_EIO, "gbwt: the records hold %llu positions, the header says %llu", (unsigned long long)positions, (unsigned long long)g.size);
    if (g.recs[0].body != g.sequences) return fail(VGAN_EIO, "gbwt: the endmarker record does not list every sequence");
    return VGAN_OK;
}

} // namespace

extern "C" int vgan_gbwt_load(const char *path, vgan_gbwt **out) {
    if (!path || !out) return fail(VGAN_EINVAL, "vgan_gbwt_load: null argument");
    std::string raw;
    if (!read_file(path, raw)) return fail(VGAN_EIO, "cannot read %s", path);
    try {
        auto g = std::make_unique<vgan_gbwt>();
        const int rc = load(raw, *g);
        if (rc) return rc;
        *out = g.release();
        return VGAN_OK;
    } catch (const std::bad_alloc &) {
        return fail(VGAN_ENOMEM, "gbwt: out of memory reading %s", path);
    }
}

extern "C" void vgan_gbwt_free(vgan_gbwt *g) { delete g; }

extern "C" int64_t vgan_gbwt_sequences(const vgan_gbwt *g) { return g ? (int64_t)g->sequences : 0; }
extern "C" int vgan_gbwt_bidirectional(const vgan_gbwt *g) { return g && (g->flags & 1) ? 1 : 0; }

extern "C" int64_t vgan_gbwt_extract(const vgan_gbwt *g, int64_t sequence, uint64_t *nodes, int64_t cap) {
    if (!g || (cap > 0 && !nodes)) return fail(VGAN_EINVAL, "vgan_gbwt_extract: null argument");
    if (sequence < 0 || (uint64_t)sequence >= g->sequences) return 0; // gbwt::GBWT::extract of an invalid id: empty
    uint64_t node, off;
    if (!lf(*g, 0, (uint64_t)sequence, node, off)) return fail(VGAN_EIO, "gbwt: broken endmarker record");
    int64_t n = 0;
    while (node != 0) {
        if ((uint64_t)n > g->size) return fail(VGAN_EIO, "gbwt: sequence %lld does not end", (long long)sequence);
        if (n < cap) nodes[n] = node;
        ++n;
        uint64_t nn, no;
        if (!lf(*g, node, off, nn, no)) return fail(VGAN_EIO, "gbwt: broken record of node %llu", (unsigned long long)node);
        node = nn;
        off = no;
    }
    return n;
}

// src/readOG_Euka.h:55-73, literally: for path_id < n_paths the nodes of gbwt->extract(path_id) -- GBWT SEQUENCE path_id,
// i.e. in a bidirectional index the forward and reverse strands of the first n_paths / 2 paths -- are handed to
// graph.get_handle() as if they were node ids (they are 2 * id + strand), and row (that number - 1) is marked.
extern "C" int vgan_gbwt_node_path_matrix(const vgan_gbwt *g, int64_t n_nodes, int64_t n_paths, uint8_t *matrix) {
    if (!g || !matrix || n_nodes < 0 || n_paths < 0) return fail(VGAN_EINVAL, "vgan_gbwt_node_path_matrix: bad argument");
    memset(matrix, 0, (size_t)n_nodes * (size_t)n_paths);
    std::vector<uint64_t> nodes;
    for (int64_t p = 0; p < n_paths; ++p) {
        int64_t n = vgan_gbwt_extract(g, p, nullptr, 0);
        if (n < 0) return (int)n;
        nodes.resize((size_t)n);
        n = vgan_gbwt_extract(g, p, nodes.data(), n);
        if (n < 0) return (int)n;
        for (uint64_t v : nodes) {
            const int64_t index = (int64_t)v - 1;
            if (index >= 0 && index < n_nodes) matrix[(size_t)index * (size_t)n_paths + (size_t)p] = 1;
        }
    }
    return VGAN_OK;
}
