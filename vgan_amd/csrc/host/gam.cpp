// GAM reader/writer: the host-side replacement of readGAM() (reference src/readGAM.h:20-68), which goes
// through libvgio + protobuf.  Here: zlib inflate of the gzip/BGZF members, then a zero-copy walk of the
// type-tagged length-prefixed groups and of the protobuf wire format of vg.Alignment
// (field numbers verified on the reference's fixture test_reads.gam).
#include "common.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <sys/mman.h>

using namespace vgan;

void vgan_alnset::fill_view(vgan_alnset_view *v) const {
    v->n_reads = n_reads();
    v->seq_off = seq_off.data();
    v->seq = seq.data();
    v->qual_off = qual_off.data();
    v->qual = qual.data();
    v->mapq = mapq.data();
    v->identity = identity.data();
    v->name_off = name_off.data();
    v->name = name.data();
    v->map_off = map_off.data();
    v->m_node = m_node.data();
    v->m_offset = m_offset.data();
    v->m_rev = m_rev.data();
    v->edit_off = edit_off.data();
    v->e_from = e_from.data();
    v->e_to = e_to.data();
    v->e_seq_off = e_seq_off.data();
    v->e_seq = e_seq.data();
}

namespace {

struct Cur {
    const uint8_t *p, *e;
    bool ok = true;
    bool done() const { return p >= e; }
    uint64_t varint() {
        if (p < e && !(*p & 0x80)) return *p++; // one byte: field keys, lengths, node offsets, edit lengths
        uint64_t v = 0;
        int shift = 0;
        while (p < e) {
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
            shift += 7;
            if (shift > 63) break;
        }
        ok = false;
        return 0;
    }
    Cur sub() {
        const uint64_t n = varint();
        if (!ok || n > (uint64_t)(e - p)) {
            ok = false;
            return Cur{p, p, false};
        }
        Cur c{p, p + n, true};
        p += n;
        return c;
    }
    void skip(int wt) {
        switch (wt) {
        case 0: varint(); break;
        case 1: if (e - p >= 8) p += 8; else ok = false; break;
        case 2: sub(); break;
        case 5: if (e - p >= 4) p += 4; else ok = false; break;
        default: ok = false;
        }
    }
};

bool parse_edit(Cur c, vgan_alnset &a) {
    int32_t from = 0, to = 0;
    const uint8_t *sb = nullptr, *se = nullptr;
    while (!c.done() && c.ok) {
        const uint64_t key = c.varint();
        const int f = (int)(key >> 3), wt = (int)(key & 7);
        if (f == 1 && wt == 0) from = (int32_t)c.varint();
        else if (f == 2 && wt == 0) to = (int32_t)c.varint();
        else if (f == 3 && wt == 2) {
            Cur s = c.sub();
            sb = s.p;
            se = s.e;
        } else c.skip(wt);
    }
    if (!c.ok) return false;
    a.e_from.push_back(from);
    a.e_to.push_back(to);
    if (sb) a.e_seq.append((const char *)sb, se - sb);
    a.e_seq_off.push_back((int64_t)a.e_seq.size());
    return true;
}

bool parse_mapping(Cur c, vgan_alnset &a) {
    int64_t node = 0, off = 0;
    uint8_t rev = 0;
    while (!c.done() && c.ok) {
        const uint64_t key = c.varint();
        const int f = (int)(key >> 3), wt = (int)(key & 7);
        if (f == 1 && wt == 2) {
            Cur p = c.sub();
            while (!p.done() && p.ok) {
                const uint64_t k2 = p.varint();
                const int f2 = (int)(k2 >> 3), w2 = (int)(k2 & 7);
                if (f2 == 1 && w2 == 0) node = (int64_t)p.varint();
                else if (f2 == 2 && w2 == 0) off = (int64_t)p.varint();
                else if (f2 == 4 && w2 == 0) rev = p.varint() != 0;
                else p.skip(w2);
            }
            if (!p.ok) return false;
        } else if (f == 2 && wt == 2) {
            if (!parse_edit(c.sub(), a)) return false;
        } else c.skip(wt);
    }
    if (!c.ok) return false;
    a.m_node.push_back(node);
    a.m_offset.push_back(off);
    a.m_rev.push_back(rev);
    a.edit_off.push_back((int64_t)a.e_from.size());
    return true;
}

struct Mark {
    size_t seq, qual, name, e_seq, m, e;
};

bool parse_alignment(Cur c, vgan_alnset &a, int keep_unmapped) {
    const Mark mk{a.seq.size(), a.qual.size(), a.name.size(), a.e_seq.size(), a.m_node.size(), a.e_from.size()};
    int32_t mapq = 0;
    double identity = 0.0;
    while (!c.done() && c.ok) {
        const uint64_t key = c.varint();
        const int f = (int)(key >> 3), wt = (int)(key & 7);
        if (f == 1 && wt == 2) {
            Cur s = c.sub();
            a.seq.append((const char *)s.p, s.e - s.p);
        } else if (f == 2 && wt == 2) {
            Cur path = c.sub();
            while (!path.done() && path.ok) {
                const uint64_t k2 = path.varint();
                const int f2 = (int)(k2 >> 3), w2 = (int)(k2 & 7);
                if (f2 == 2 && w2 == 2) {
                    if (!parse_mapping(path.sub(), a)) return false;
                } else path.skip(w2);
            }
            if (!path.ok) return false;
        } else if (f == 3 && wt == 2) {
            Cur s = c.sub();
            a.name.append((const char *)s.p, s.e - s.p);
        } else if (f == 4 && wt == 2) {
            Cur s = c.sub();
            a.qual.append((const char *)s.p, s.e - s.p);
        } else if (f == 5 && wt == 0) {
            mapq = (int32_t)c.varint();
        } else if (f == 16 && wt == 1) {
            if (c.e - c.p < 8) return false;
            memcpy(&identity, c.p, 8);
            c.p += 8;
        } else c.skip(wt);
    }
    if (!c.ok) return false;
    if (!keep_unmapped && identity == 0.0) { // readGAM.h:47: "Discard unmapped reads"
        a.seq.resize(mk.seq);
        a.qual.resize(mk.qual);
        a.name.resize(mk.name);
        a.e_seq.resize(mk.e_seq);
        a.m_node.resize(mk.m);
        a.m_offset.resize(mk.m);
        a.m_rev.resize(mk.m);
        a.edit_off.resize(mk.m + 1);
        a.e_from.resize(mk.e);
        a.e_to.resize(mk.e);
        a.e_seq_off.resize(mk.e + 1);
        return true;
    }
    a.seq_off.push_back((int64_t)a.seq.size());
    a.qual_off.push_back((int64_t)a.qual.size());
    a.name_off.push_back((int64_t)a.name.size());
    a.map_off.push_back((int64_t)a.m_node.size());
    a.mapq.push_back(mapq);
    a.identity.push_back(identity);
    return true;
}

void put_varint(std::string &o, uint64_t v) {
    while (v >= 0x80) {
        o += (char)((v & 0x7f) | 0x80);
        v >>= 7;
    }
    o += (char)v;
}
void put_ld(std::string &o, int f, const std::string &payload) {
    put_varint(o, ((uint64_t)f << 3) | 2);
    put_varint(o, payload.size());
    o += payload;
}
void put_ld(std::string &o, int f, const char *p, size_t n) {
    put_varint(o, ((uint64_t)f << 3) | 2);
    put_varint(o, n);
    o.append(p, n);
}
void put_vi(std::string &o, int f, uint64_t v) {
    put_varint(o, (uint64_t)f << 3);
    put_varint(o, v);
}

} // namespace

// Concatenates per-thread alignment sets: sizes first, one allocation per array, then block copies with the offsets
// of each part shifted (no per-element push_back, no reallocation).
void vgan::merge_alnsets(std::vector<vgan_alnset> &parts, vgan_alnset &o) {
    size_t R = 0, M = 0, E = 0, nseq = 0, nqual = 0, nname = 0, neseq = 0;
    for (auto &p : parts) {
        R += p.mapq.size();
        M += p.m_node.size();
        E += p.e_from.size();
        nseq += p.seq.size();
        nqual += p.qual.size();
        nname += p.name.size();
        neseq += p.e_seq.size();
    }
    o.seq_off.resize(R + 1);
    o.qual_off.resize(R + 1);
    o.name_off.resize(R + 1);
    o.map_off.resize(R + 1);
    o.mapq.resize(R);
    o.identity.resize(R);
    o.m_node.resize(M);
    o.m_offset.resize(M);
    o.m_rev.resize(M);
    o.edit_off.resize(M + 1);
    o.e_from.resize(E);
    o.e_to.resize(E);
    o.e_seq_off.resize(E + 1);
    o.seq.resize(nseq);
    o.qual.resize(nqual);
    o.name.resize(nname);
    o.e_seq.resize(neseq);
    o.seq_off[0] = o.qual_off[0] = o.name_off[0] = o.map_off[0] = o.edit_off[0] = o.e_seq_off[0] = 0;
    struct Base {
        size_t r, m, e, seq, qual, name, eseq;
    };
    std::vector<Base> base(parts.size());
    Base acc{0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < parts.size(); ++i) {
        base[i] = acc;
        acc.r += parts[i].mapq.size();
        acc.m += parts[i].m_node.size();
        acc.e += parts[i].e_from.size();
        acc.seq += parts[i].seq.size();
        acc.qual += parts[i].qual.size();
        acc.name += parts[i].name.size();
        acc.eseq += parts[i].e_seq.size();
    }
    auto copy_part = [&](size_t i) {
        vgan_alnset &p = parts[i];
        const Base &bs = base[i];
        auto shift = [](const BigVec<int64_t> &src, int64_t *dst, int64_t by) {
            for (size_t k = 1; k < src.size(); ++k) dst[k - 1] = src[k] + by;
        };
        shift(p.seq_off, &o.seq_off[bs.r + 1], (int64_t)bs.seq);
        shift(p.qual_off, &o.qual_off[bs.r + 1], (int64_t)bs.qual);
        shift(p.name_off, &o.name_off[bs.r + 1], (int64_t)bs.name);
        shift(p.map_off, &o.map_off[bs.r + 1], (int64_t)bs.m);
        shift(p.edit_off, &o.edit_off[bs.m + 1], (int64_t)bs.e);
        shift(p.e_seq_off, &o.e_seq_off[bs.e + 1], (int64_t)bs.eseq);
        auto cp = [](auto &dst, size_t at, const auto &src) {
            if (!src.empty()) memcpy(&dst[at], src.data(), src.size() * sizeof(src[0]));
        };
        cp(o.mapq, bs.r, p.mapq);
        cp(o.identity, bs.r, p.identity);
        cp(o.m_node, bs.m, p.m_node);
        cp(o.m_offset, bs.m, p.m_offset);
        cp(o.m_rev, bs.m, p.m_rev);
        cp(o.e_from, bs.e, p.e_from);
        cp(o.e_to, bs.e, p.e_to);
        cp(o.seq, bs.seq, p.seq);
        cp(o.qual, bs.qual, p.qual);
        cp(o.name, bs.name, p.name);
        cp(o.e_seq, bs.eseq, p.e_seq);
        p = vgan_alnset();
    };
    const size_t nt = std::min<size_t>(parts.size(), burst_cpus());
    if (nt <= 1) {
        for (size_t i = 0; i < parts.size(); ++i) copy_part(i);
    } else {
        parallel_run((int)nt, [&](int t) {
            for (size_t i = (size_t)t; i < parts.size(); i += nt) copy_part(i);
        });
    }
}

// ---- BGZF input (what vg writes): the segment pipeline ---------------------------------------------------------------
// The compressed file is cut into segments of a few dozen BGZF blocks.  A pool of workers takes them in file order; each
// inflates its segment into a recycled buffer, FRAMES it speculatively (a walk from the first group header it can
// recognise: the item "GAM" that opens every group vg writes) and parses the messages it framed.  One serial thread, the
// stitcher, carries the true state of the framing across the segments: it walks from where the previous segment ended
// until it stands on a group header the segment's own walk started from or passed -- from there on the two walks are the
// same deterministic function of the same bytes, so the segment's result is taken whole -- and hands the few messages it
// framed itself (half a group per segment, and the message that straddles the boundary, which it joins in the headroom in
// front of the next buffer) to the pool.  A stream without tags, or a walk that never meets the true one, is framed by the
// stitcher alone and still parsed by the pool: the result is the serial walk's in every case.
// Memory is bounded: the buffers form a ring (a segment's buffer returns when its messages are parsed and the stitcher
// has passed it), and no segment is started more than `ahead` segments beyond the last one the caller has taken.
namespace {

std::atomic<int64_t> g_decode_counts[5];

struct WalkState {
    bool in_group = false; // false: a group header (the item count) comes next
    bool first = false;    // the next item is the first of its group: it may be the type tag
    uint64_t rem = 0;      // items left in the group
};

// 1: read, q is past it; 0: the bytes end inside it; -1: longer than ten bytes
inline int get_varint(const uint8_t *p, const uint8_t *e, uint64_t &v, const uint8_t *&q) {
    v = 0;
    for (int shift = 0; shift <= 63; shift += 7) {
        if (p >= e) return 0;
        const uint8_t b = *p++;
        v |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) {
            q = p;
            return 1;
        }
    }
    return -1;
}

enum { WALK_MORE = 0, WALK_BAD = -1, WALK_STOP = 1 };

// The framing of readGAM's stream (libvgio's groups: {count, count x (length, bytes)}, the first item of a group possibly
// the tag "GAM") from state st at p, over [p, e).  WALK_MORE: the item at p does not end inside the bytes (need = its size
// when the length could be read, else 0); WALK_BAD: a malformed varint at p; WALK_STOP: at_header(p) said so.
template <class Emit, class AtHeader>
int walk(WalkState &st, const uint8_t *&p, const uint8_t *e, uint64_t &need, Emit &&emit, AtHeader &&at_header) {
    need = 0;
    for (;;) {
        uint64_t v;
        const uint8_t *q;
        if (!st.in_group) {
            if (at_header(p)) return WALK_STOP;
            const int r = get_varint(p, e, v, q);
            if (r <= 0) return r;
            st.rem = v;
            st.first = true;
            st.in_group = v != 0;
            p = q;
            continue;
        }
        const int r = get_varint(p, e, v, q);
        if (r <= 0) return r;
        if (v > (uint64_t)(e - q)) {
            need = v > UINT64_MAX - 16 ? UINT64_MAX : v + (uint64_t)(q - p);
            return WALK_MORE;
        }
        if (!(st.first && v == 3 && memcmp(q, "GAM", 3) == 0)) emit(q, q + v);
        st.first = false;
        p = q + v;
        if (--st.rem == 0) st.in_group = false;
    }
}

void reserve_for(vgan_alnset &a, size_t nbytes, size_t nr) { // a mapping with one edit is 13-16 bytes on the wire
    a.seq_off.reserve(nr + 1);
    a.qual_off.reserve(nr + 1);
    a.name_off.reserve(nr + 1);
    a.map_off.reserve(nr + 1);
    a.mapq.reserve(nr);
    a.identity.reserve(nr);
    a.seq.reserve(nbytes / 6);
    a.qual.reserve(nbytes / 6);
    a.name.reserve(nr * 16);
    a.m_node.reserve(nbytes / 12 + 16);
    a.m_offset.reserve(nbytes / 12 + 16);
    a.m_rev.reserve(nbytes / 12 + 16);
    a.edit_off.reserve(nbytes / 12 + 16);
    a.e_from.reserve(nbytes / 12 + 16);
    a.e_to.reserve(nbytes / 12 + 16);
    a.e_seq_off.reserve(nbytes / 12 + 16);
}

struct SegPipe {
    using Msg = std::pair<const uint8_t *, const uint8_t *>;
    struct Hdr {
        uint32_t off, msg; // a group header the segment's walk stood on: its offset, and the messages framed before it
    };
    struct Seg {
        size_t b0 = 0, b1 = 0, out_off = 0, size = 0;
        uint8_t *buf = nullptr; // data at buf + head; the bytes before it take the tail of the previous segment
        size_t cap = 0, head = 0;
        bool special = false; // a buffer of its own (a message larger than the headroom)
        std::atomic<int> refs{0};
        bool ready = false; // under mu
        bool inflate_ok = true;
        bool anchored = false, closed = false; // closed: the walk saw the end of the group it started in
        size_t anchor_tag = 0;
        uint32_t n_first = 0;
        std::vector<Hdr> hdrs;
        std::vector<std::pair<uint32_t, uint32_t>> msgs; // (offset, length) of the framed messages
        WalkState end;
        size_t end_off = 0;
        vgan_alnset parsed;
        bool parse_ok = true;
        uint8_t *data() const { return buf + head; }
    };
    struct Entry { // what take() hands out, in input order
        vgan_alnset a;
        std::vector<Msg> msgs; // to be parsed by the pool (empty: `a` came parsed with its segment)
        Seg *seg = nullptr;
        size_t k = 0;
        bool parsed = false;
    };

    const uint8_t *in = nullptr;
    bool in_is_file_mapping = false; // the compressed bytes are a private read-only mapping of the file: pages behind the
    size_t in_released = 0;          // stitcher are dropped from the resident set (they stay in the page cache)
    std::vector<BgzfBlock> blocks;
    int keep_unmapped = 0;
    size_t total_out = 0, seg_blocks = 0, buf_bytes = 0, ring = 0, ahead = 0;
    std::vector<Seg> segs;
    std::deque<Entry> entries;
    std::deque<size_t> head_tasks;
    std::vector<uint8_t *> free_bufs, all_bufs;
    std::mutex mu;
    std::condition_variable cv_work, cv_seg, cv_done;
    size_t next_seg = 0, stitched = 0, delivered = 0;
    int64_t delivered_reads = 0;
    bool stop = false, stitch_done = false, bad_inflate = false, bad_frame = false;
    std::thread stitcher;
    std::vector<std::thread> workers;
    PhaseTimer pt{"parse_gam"};

    ~SegPipe() {
        join();
        for (Seg &s : segs)
            if (s.buf && s.special) big_free_bytes(s.buf, s.cap);
        for (uint8_t *b : all_bufs) big_free_bytes(b, buf_bytes);
    }

    void join() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        cv_seg.notify_all();
        if (stitcher.joinable()) stitcher.join();
        for (auto &t : workers)
            if (t.joinable()) t.join();
        workers.clear();
    }

    bool start(const void *bytes, size_t n, int keep) {
        in = (const uint8_t *)bytes;
        if (!bgzf_index(in, n, blocks) || blocks.empty()) return false;
        keep_unmapped = keep;
        total_out = blocks.back().out_off + blocks.back().out_size;
        // a short input is a burst; a long one runs at the quota (twice as many workers as processors: they wait on each other)
        unsigned nt = std::min(blocks.size() <= 8192 ? burst_cpus() : 2 * usable_cpus(), 64u);
        // segments of 2-8 MB of inflated bytes: several per worker on a small input, and long against a group (vg: up to
        // 1000 messages) on a large one -- the part of a segment before its first group header is framed serially
        seg_blocks = std::min<size_t>(128, std::max<size_t>(32, blocks.size() / (4 * (size_t)nt)));
        if (const char *e = getenv("VGAN_GAM_SEG_BLOCKS")) { // tests: segments of a block or two, so that every way a group
            const long v = atol(e);                          // or a message can lie across a boundary occurs in a small file
            if (v >= 1) seg_blocks = (size_t)std::min<long>(v, 1024);
        }
        if (const char *e = getenv("VGAN_GAM_THREADS")) {
            const long v = atol(e);
            if (v >= 1) nt = (unsigned)std::min<long>(v, 256);
        }
        const size_t nseg = (blocks.size() + seg_blocks - 1) / seg_blocks;
        nt = (unsigned)std::min<size_t>(nt, nseg);
        buf_bytes = seg_blocks * 65536 * 3 / 2; // a size class of the block pool; the half in front is the headroom
        ring = nt + 8;
        size_t ahead_blocks = 4096; // ~0.25 GB of inflated input (160k short reads) parsed ahead of the caller
        if (const char *e = getenv("VGAN_GAM_AHEAD_BLOCKS")) ahead_blocks = (size_t)std::max<long>(1, atol(e));
        ahead = std::max<size_t>(nt + 8, ahead_blocks / seg_blocks);
        segs = std::vector<Seg>(nseg);
        for (size_t k = 0; k < nseg; ++k) {
            Seg &s = segs[k];
            s.b0 = k * seg_blocks;
            s.b1 = std::min(blocks.size(), s.b0 + seg_blocks);
            s.out_off = blocks[s.b0].out_off;
            s.size = blocks[s.b1 - 1].out_off + blocks[s.b1 - 1].out_size - s.out_off;
        }
        pt.lap("inflate started");
        for (unsigned t = 0; t < nt; ++t) workers.emplace_back([this] { guarded([this] { work(); }); });
        stitcher = std::thread([this] { guarded([this] { stitch(); }); });
        return true;
    }

    // a thread of the pipeline that runs out of memory (or throws anything else) ends the stream with an error instead of
    // the process: take() reports it as a malformed stream would be
    template <class F> void guarded(F &&f) {
        try {
            f();
        } catch (...) {
            {
                std::lock_guard<std::mutex> lk(mu);
                bad_frame = stop = true;
            }
            cv_work.notify_all();
            cv_seg.notify_all();
            cv_done.notify_all();
        }
    }

    void release(Seg &s) {
        if (s.refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
        uint8_t *b = s.buf;
        s.buf = nullptr;
        if (s.special) {
            big_free_bytes(b, s.cap);
            s.special = false;
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            free_bufs.push_back(b);
        }
        cv_work.notify_one();
    }

    size_t taken() const { return delivered < entries.size() ? entries[delivered].k : stitched; } // under mu

    bool parse_msgs(const Msg *m, size_t n, vgan_alnset &a) const {
        if (!n) return true;
        reserve_for(a, (size_t)(m[n - 1].second - m[0].first), n);
        for (size_t i = 0; i < n; ++i)
            if (!parse_alignment(Cur{m[i].first, m[i].second, true}, a, keep_unmapped)) return false;
        return true;
    }

    // the segment's own walk: from the first "GAM" item whose group it can follow
    void frame_segment(Seg &s) const {
        const uint8_t *d = s.data(), *e = d + s.size;
        static const uint8_t TAG[4] = {3, 'G', 'A', 'M'};
        const uint8_t *t = s.size >= 5 ? (const uint8_t *)memmem(d + 1, s.size - 1, TAG, 4) : nullptr; // its count before it
        if (!t) return;
        s.anchored = true;
        s.anchor_tag = (size_t)(t - d);
        // the group of the anchor: its count lies before the tag and cannot be told from the previous message's last
        // bytes, so its end is recognised instead (a count followed by the tag; no message starts with field number 0)
        const uint8_t *p = t + 4;
        for (;;) {
            uint64_t v;
            const uint8_t *q;
            if (get_varint(p, e, v, q) <= 0 || e - q < 4) break;
            if (memcmp(q, TAG, 4) == 0) {
                s.closed = true;
                break;
            }
            if (v > (uint64_t)(e - q)) break;
            s.msgs.emplace_back((uint32_t)(q - d), (uint32_t)v);
            p = q + v;
        }
        s.n_first = (uint32_t)s.msgs.size();
        s.end_off = (size_t)(p - d);
        if (!s.closed) return; // the stitcher knows the count and resumes at end_off
        uint64_t need;
        walk(s.end, p, e, need, [&](const uint8_t *a, const uint8_t *b) { s.msgs.emplace_back((uint32_t)(a - d), (uint32_t)(b - a)); },
             [&](const uint8_t *h) {
                 s.hdrs.push_back({(uint32_t)(h - d), (uint32_t)s.msgs.size()});
                 return false;
             });
        s.end_off = (size_t)(p - d); // WALK_MORE and WALK_BAD alike: the stitcher walks the item at end_off itself
    }

    void do_segment(Seg &s) {
        const bool acct = pt.on;
        const double c0 = acct ? thread_cpu_ms() : 0;
        struct Acct {
            bool on;
            double c0, c1 = 0;
            ~Acct() {
                if (!on) return;
                const double c2 = thread_cpu_ms();
                cpu_account().inflate += (int64_t)(((c1 ? c1 : c2) - c0) * 1e3);
                if (c1) cpu_account().frame_parse += (int64_t)((c2 - c1) * 1e3);
            }
        } ac{acct, c0};
        for (size_t b = s.b0; b < s.b1; ++b) {
            const BgzfBlock &bl = blocks[b];
            if (!inflate_member(in + bl.in_off, bl.in_size, s.data() + (bl.out_off - s.out_off), bl.out_size)) {
                s.inflate_ok = false;
                return;
            }
        }
        if (acct) ac.c1 = thread_cpu_ms();
        frame_segment(s);
        if (s.msgs.empty()) return;
        const uint8_t *d = s.data();
        reserve_for(s.parsed, (size_t)s.msgs.back().first + s.msgs.back().second - s.msgs.front().first, s.msgs.size());
        for (const auto &m : s.msgs)
            if (!parse_alignment(Cur{d + m.first, d + m.first + m.second, true}, s.parsed, keep_unmapped)) {
                s.parse_ok = false;
                return;
            }
    }

    void work() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] {
                return stop || !head_tasks.empty() || (next_seg >= segs.size() && stitch_done) ||
                       (next_seg < segs.size() && next_seg < taken() + ahead && (!free_bufs.empty() || all_bufs.size() < ring));
            });
            if (stop) return;
            if (!head_tasks.empty()) {
                Entry *en = &entries[head_tasks.front()]; // element addresses are stable while entries are appended
                head_tasks.pop_front();
                lk.unlock();
                const double hc0 = pt.on ? thread_cpu_ms() : 0;
                const bool good = parse_msgs(en->msgs.data(), en->msgs.size(), en->a);
                if (pt.on) cpu_account().frame_parse += (int64_t)((thread_cpu_ms() - hc0) * 1e3);
                std::vector<Msg>().swap(en->msgs);
                release(*en->seg);
                lk.lock();
                en->parsed = true;
                if (!good) bad_frame = stop = true;
                cv_done.notify_all();
                if (!good) cv_work.notify_all(), cv_seg.notify_all();
                continue;
            }
            if (next_seg < segs.size() && next_seg < taken() + ahead && (!free_bufs.empty() || all_bufs.size() < ring)) {
                Seg &s = segs[next_seg++];
                if (!free_bufs.empty()) {
                    s.buf = free_bufs.back();
                    free_bufs.pop_back();
                } else {
                    s.buf = (uint8_t *)big_alloc_bytes(buf_bytes);
                    all_bufs.push_back(s.buf);
                }
                s.cap = buf_bytes;
                s.head = buf_bytes - seg_blocks * 65536;
                s.refs.store(1, std::memory_order_relaxed); // the stitcher's
                lk.unlock();
                do_segment(s);
                lk.lock();
                s.ready = true;
                cv_seg.notify_all();
                continue;
            }
            if (next_seg >= segs.size() && stitch_done) return;
        }
    }

    // (stitcher) messages framed serially, in buffer k: an entry the pool parses
    void hand_over(std::vector<Msg> &cur, size_t k) {
        if (cur.empty()) return;
        segs[k].refs.fetch_add(1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu);
            entries.emplace_back();
            Entry &en = entries.back();
            en.msgs.swap(cur);
            en.seg = &segs[k];
            en.k = k;
            head_tasks.push_back(entries.size() - 1);
        }
        cv_work.notify_one();
        cur.clear();
        cur.reserve(1024);
    }

    bool wait_ready(size_t k) {
        std::unique_lock<std::mutex> lk(mu);
        cv_seg.wait(lk, [&] { return stop || segs[k].ready; });
        return !stop;
    }

    void stitch() {
        constexpr size_t HEAD_SLICE = 2048;
        if (in_is_file_mapping) {
            // indexing the blocks touched a page or two of each, and the kernel mapped their neighbours with them: the whole
            // file counts as resident 0.1 s after the start.  What lies beyond the workers' first reach is let go again (the
            // pages stay in the page cache and come back as the workers get there)
            const size_t page = 4096, far_block = std::min(blocks.size() - 1, (ahead + 2) * seg_blocks);
            const size_t from = (blocks[far_block].in_off + page - 1) / page * page;
            const size_t end = blocks.back().in_off + blocks.back().in_size;
            if ((uintptr_t)in % page == 0 && end > from) (void)madvise((void *)((uintptr_t)in + from), end - from, MADV_DONTNEED);
        }
        WalkState st;
        std::vector<Msg> cur;
        bool frame_ok = true, inflate_ok = true;
        const uint8_t *pos = nullptr;
        for (size_t k = 0; k < segs.size();) {
            Seg &s = segs[k];
            if (k == 0) {
                if (!wait_ready(0)) return;
                pos = s.data();
            }
            if (!s.inflate_ok) {
                inflate_ok = false;
                break;
            }
            const uint8_t *const data = s.data(), *const end = data + s.size;
            size_t hi = 0;
            uint32_t from_msg = 0;
            uint64_t anchor_count = 0;
            bool at_anchor = false, met = !s.anchored, own = false;
            uint64_t need = 0;
            int r;
            for (;;) {
                r = walk(st, pos, end, need,
                         [&](const uint8_t *a, const uint8_t *b) {
                             cur.emplace_back(a, b);
                             if (cur.size() == HEAD_SLICE) hand_over(cur, k);
                         },
                         [&](const uint8_t *h) {
                             if (met || h < data) return false;
                             const size_t x = (size_t)(h - data);
                             while (hi < s.hdrs.size() && s.hdrs[hi].off < x) ++hi;
                             if (hi < s.hdrs.size() && s.hdrs[hi].off == x) {
                                 from_msg = s.hdrs[hi].msg;
                                 return true;
                             }
                             if (x < s.anchor_tag && s.anchor_tag - x <= 10) {
                                 uint64_t c;
                                 const uint8_t *q;
                                 if (get_varint(h, data + s.anchor_tag, c, q) == 1 && q == data + s.anchor_tag && c >= 1 &&
                                     (s.closed ? c - 1 == s.n_first : c - 1 >= s.n_first)) {
                                     at_anchor = true;
                                     anchor_count = c;
                                     return true;
                                 }
                             }
                             return false;
                         });
                if (r != WALK_STOP) break;
                // the true walk stands where the segment's own walk stood: the rest of the segment is that walk's
                met = own = true;
                g_decode_counts[at_anchor ? 1 : 2].fetch_add(1, std::memory_order_relaxed);
                hand_over(cur, k);
                if (at_anchor || from_msg == 0) {
                    if (!s.parse_ok) {
                        frame_ok = false;
                        break;
                    }
                    if (s.parsed.n_reads() > 0) {
                        std::lock_guard<std::mutex> lk(mu);
                        entries.emplace_back();
                        Entry &en = entries.back();
                        en.a = std::move(s.parsed);
                        en.k = k;
                        en.parsed = true;
                    }
                    cv_done.notify_all();
                } else { // met at a later group: the messages before it are not the stream's
                    for (size_t i = from_msg; i < s.msgs.size(); ++i) {
                        cur.emplace_back(data + s.msgs[i].first, data + s.msgs[i].first + s.msgs[i].second);
                        if (cur.size() == HEAD_SLICE) hand_over(cur, k);
                    }
                    hand_over(cur, k);
                }
                if (at_anchor && !s.closed) {
                    st.rem = anchor_count - 1 - s.n_first;
                    st.in_group = st.rem != 0;
                    st.first = false;
                } else {
                    st = s.end;
                }
                pos = data + s.end_off;
            }
            g_decode_counts[0].fetch_add(1, std::memory_order_relaxed);
            if (!own) g_decode_counts[3].fetch_add(1, std::memory_order_relaxed);
            s.parsed = vgan_alnset();
            std::vector<Hdr>().swap(s.hdrs);
            std::vector<std::pair<uint32_t, uint32_t>>().swap(s.msgs);
            if (!frame_ok || r == WALK_BAD) {
                frame_ok = false;
                break;
            }
            hand_over(cur, k);
            const size_t left = (size_t)(end - pos);
            if (k + 1 == segs.size()) {
                frame_ok = left == 0 && !st.in_group;
                break;
            }
            if (need && need - left > total_out - (s.out_off + s.size)) { // longer than the rest of the stream
                frame_ok = false;
                break;
            }
            if (!wait_ready(k + 1)) return;
            Seg &nx = segs[k + 1];
            if (!nx.inflate_ok) {
                inflate_ok = false;
                break;
            }
            if (left > nx.head) { // a message longer than the headroom: a buffer of its own for the two pieces
                g_decode_counts[4].fetch_add(1, std::memory_order_relaxed);
                const size_t cap = left + nx.size + 64;
                uint8_t *nb = (uint8_t *)big_alloc_bytes(cap);
                memcpy(nb + left, nx.data(), nx.size);
                uint8_t *old = nx.buf;
                const bool was_special = nx.special;
                const size_t old_cap = nx.cap;
                nx.buf = nb;
                nx.cap = cap;
                nx.head = left;
                nx.special = true;
                if (was_special) {
                    big_free_bytes(old, old_cap);
                } else {
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        free_bufs.push_back(old);
                    }
                    cv_work.notify_one();
                }
            }
            if (left) memcpy(nx.data() - left, pos, left);
            pos = nx.data() - left;
            if (in_is_file_mapping && (k & 63u) == 63u) { // every 64 segments: the compressed input the workers are done with
                const size_t page = 4096, upto = blocks[s.b1 - 1].in_off / page * page; // (segments up to k are inflated: k + 1 is ready)
                const size_t from = (in_released + page - 1) / page * page;
                const uintptr_t base = (uintptr_t)in;
                if (upto > from && base % page == 0) (void)madvise((void *)(base + from), upto - from, MADV_DONTNEED);
                in_released = upto;
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                stitched = k + 1;
            }
            release(s);
            ++k;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            stitched = segs.size();
            stitch_done = true;
            if (!inflate_ok) bad_inflate = true;
            if (!frame_ok) bad_frame = true;
            if (!inflate_ok || !frame_ok) stop = true;
        }
        pt.lap("framing");
        cv_work.notify_all();
        cv_done.notify_all();
    }

    int take(int64_t min_reads, std::vector<vgan_alnset> &out, int64_t *first_read) {
        out.clear();
        std::unique_lock<std::mutex> lk(mu);
        if (first_read) *first_read = delivered_reads;
        int64_t got = 0;
        bool moved = false;
        auto ready = [&] { return bad_inflate || bad_frame || (delivered < entries.size() && entries[delivered].parsed) || (stitch_done && delivered >= entries.size()); };
        for (;;) {
            if (!ready()) {
                if (moved) cv_work.notify_all(); // the workers may run further ahead
                moved = false;
                cv_done.wait(lk, ready);
            }
            if (bad_inflate || bad_frame) break;
            if (delivered >= entries.size()) break; // end of the stream
            Entry &en = entries[delivered];
            const int64_t nr = en.a.n_reads();
            if (nr > 0) {
                out.emplace_back(std::move(en.a));
                got += nr;
            }
            en.a = vgan_alnset();
            ++delivered;
            moved = true;
            if (got >= min_reads) break;
        }
        delivered_reads += got;
        const bool bi = bad_inflate, bf = bad_frame;
        lk.unlock();
        if (moved) cv_work.notify_all();
        if (bi) return fail(VGAN_EIO, "GAM: gzip stream is corrupt");
        if (bf) return fail(VGAN_EIO, "GAM: malformed group or Alignment message");
        return VGAN_OK;
    }
};

} // namespace

// GAM bytes -> slices of reads in input order, as a pipeline that runs behind the caller; take() returns the next parsed
// slices as soon as they exist.  BGZF input goes through the segment pipeline above; anything else (plain bytes, a gzip
// stream that is not BGZF: inflated whole first) through a serial framing pass that hands slices of ~SLICE messages to a
// parser pool.
struct vgan_gam_stream {
    using Msg = std::pair<const uint8_t *, const uint8_t *>;
    static constexpr size_t SLICE = 8192;
    MappedFile file;
    int keep_unmapped = 0;
    ByteBuf inflated;
    std::unique_ptr<SegPipe> seg;
    const uint8_t *p = nullptr;
    size_t n = 0;
    std::deque<std::vector<Msg>> slices; // grown by the framing thread only, under mu
    std::deque<vgan_alnset> slice_parts;
    std::deque<uint8_t> parsed;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    size_t next_slice = 0, delivered = 0;
    int64_t delivered_reads = 0;
    bool framing_done = false, framed = true;
    std::atomic<bool> ok{true};
    std::thread framer;
    std::vector<std::thread> workers;
    PhaseTimer pt{"parse_gam"};

    ~vgan_gam_stream() { join(); }

    void join() {
        if (seg) seg->join();
        if (framer.joinable()) framer.join();
        for (auto &t : workers)
            if (t.joinable()) t.join();
        workers.clear();
    }

    int start(const void *bytes, size_t nbytes, int keep) {
        if (!bytes) return fail(VGAN_EINVAL, "GAM: null buffer");
        keep_unmapped = keep;
        p = (const uint8_t *)bytes;
        n = nbytes;
        if (n >= 2 && p[0] == 0x1f && p[1] == 0x8b) {
            seg.reset(new SegPipe());
            seg->in_is_file_mapping = file.mapped && bytes == file.p;
            if (seg->start(bytes, n, keep)) return VGAN_OK;
            seg.reset();
            if (!gunzip_members(bytes, n, inflated)) return fail(VGAN_EIO, "GAM: gzip stream is corrupt");
            p = (const uint8_t *)inflated.data();
            n = inflated.size();
            pt.lap("inflate");
        }
        const unsigned hw = usable_cpus();
        const unsigned nt = (unsigned)std::min<size_t>(std::min(hw, 64u), std::max<size_t>(1, n / (4u << 20)));
        for (unsigned t = 0; t < nt; ++t) workers.emplace_back([this] { work(); });
        framer = std::thread([this] { frame(); });
        return VGAN_OK;
    }

    void parse_slice(const std::vector<Msg> &ms, vgan_alnset &a) {
        if (ms.empty()) return;
        reserve_for(a, (size_t)(ms.back().second - ms.front().first), ms.size());
        for (const Msg &m : ms) {
            if (!parse_alignment(Cur{m.first, m.second, true}, a, keep_unmapped)) {
                ok = false;
                return;
            }
        }
    }

    void work() { // parser pool
        for (;;) {
            size_t i;
            const std::vector<Msg> *ms;
            vgan_alnset *dst;
            {
                std::unique_lock<std::mutex> lk(mu);
                // a slice is complete once a later one exists, or framing has ended
                cv_work.wait(lk, [&] { return next_slice + 1 < slices.size() || framing_done; });
                if (next_slice + 1 < slices.size() || (framing_done && next_slice < slices.size())) i = next_slice++;
                else return;
                ms = &slices[i]; // element addresses are stable while the framing thread appends; the deque's index
                dst = &slice_parts[i]; // is not, so it is only walked under the lock
            }
            parse_slice(*ms, *dst);
            {
                std::lock_guard<std::mutex> lk(mu);
                parsed[i] = 1;
            }
            cv_done.notify_all();
        }
    }

    void frame() { // serial: the same walk as the segment pipeline's, over the whole buffer
        auto open_slice = [&]() {
            std::lock_guard<std::mutex> lk(mu);
            slices.emplace_back();
            slices.back().reserve(SLICE);
            slice_parts.emplace_back();
            parsed.push_back(0);
        };
        open_slice();
        std::vector<Msg> *cur = &slices.back();
        WalkState st;
        const uint8_t *at = p;
        uint64_t need = 0;
        const int r = walk(st, at, p + n, need,
                           [&](const uint8_t *a, const uint8_t *b) {
                               cur->emplace_back(a, b);
                               if (cur->size() == SLICE) {
                                   open_slice();
                                   cur = &slices.back();
                                   cv_work.notify_one();
                               }
                           },
                           [](const uint8_t *) { return false; });
        {
            std::lock_guard<std::mutex> lk(mu);
            framed = r == WALK_MORE && at == p + n && !st.in_group;
            framing_done = true;
        }
        pt.lap("framing");
        cv_work.notify_all();
        cv_done.notify_all();
    }

    // The next parsed slices in input order holding at least min_reads reads (fewer at the end); empty at the end.
    int take(int64_t min_reads, std::vector<vgan_alnset> &out, int64_t *first_read) {
        if (seg) return seg->take(min_reads, out, first_read);
        out.clear();
        std::unique_lock<std::mutex> lk(mu);
        if (first_read) *first_read = delivered_reads;
        int64_t got = 0;
        for (;;) {
            cv_done.wait(lk, [&] { return !ok || (delivered < parsed.size() && parsed[delivered]) || (framing_done && delivered >= slices.size()); });
            if (!ok) break;
            if (delivered >= slices.size()) break; // end of the stream
            if (!parsed[delivered]) continue;
            vgan_alnset &a = slice_parts[delivered];
            const int64_t nr = a.n_reads();
            if (nr > 0) {
                out.emplace_back(std::move(a));
                got += nr;
            }
            a = vgan_alnset();
            std::vector<Msg>().swap(slices[delivered]);
            ++delivered;
            if (got >= min_reads) break;
        }
        delivered_reads += got;
        const bool bad_frame = framing_done && !framed;
        lk.unlock();
        if (bad_frame || !ok) return fail(VGAN_EIO, "GAM: malformed group or Alignment message");
        return VGAN_OK;
    }
};

// the whole input as slices
static int parse_gam_parts(const void *bytes, size_t n, int keep_unmapped, std::vector<vgan_alnset> &parts) {
    vgan_gam_stream st;
    int rc = st.start(bytes, n, keep_unmapped);
    if (rc) return rc;
    rc = st.take(INT64_MAX, parts, nullptr);
    st.join();
    if (rc) return rc;
    st.pt.lap("parse");
    return VGAN_OK;
}

extern "C" int vgan_aln_parse_gam(const void *bytes, size_t n, int keep_unmapped, vgan_alnset **out) {
    if (!bytes || !out) return fail(VGAN_EINVAL, "vgan_aln_parse_gam: null argument");
    std::vector<vgan_alnset> parts;
    const int rc = parse_gam_parts(bytes, n, keep_unmapped, parts);
    if (rc) return rc;
    PhaseTimer pt("parse_gam");
    auto a = new vgan_alnset();
    merge_alnsets(parts, *a);
    pt.lap("merge");
    *out = a;
    return VGAN_OK;
}

// ---- the same input kept as its slices: consumers that walk reads in order (duplicate marking, flattening) do not
// need the merged copy (GBs for millions of reads)
extern "C" int vgan_alnparts_read_gam(const char *path, int keep_unmapped, vgan_alnparts **out) {
    if (!path || !out) return fail(VGAN_EINVAL, "vgan_alnparts_read_gam: null argument");
    MappedFile f;
    if (!f.open_path(path)) return fail(VGAN_EIO, "cannot read %s", path);
    auto ps = new vgan_alnparts();
    const int rc = parse_gam_parts(f.p, f.n, keep_unmapped, ps->parts);
    if (rc) {
        delete ps;
        return rc;
    }
    ps->index();
    *out = ps;
    return VGAN_OK;
}

// ---- the same pipeline handed out chunk by chunk: the caller flattens / accumulates chunk i while the rest of the file
// is still being inflated and parsed
extern "C" int vgan_gam_stream_open(const char *path, int keep_unmapped, vgan_gam_stream **out) {
    if (!path || !out) return fail(VGAN_EINVAL, "vgan_gam_stream_open: null argument");
    auto st = new vgan_gam_stream();
    if (!st->file.open_path(path)) {
        delete st;
        return fail(VGAN_EIO, "cannot read %s", path);
    }
    const int rc = st->start(st->file.p, st->file.n, keep_unmapped);
    if (rc) {
        delete st;
        return rc;
    }
    *out = st;
    return VGAN_OK;
}

extern "C" int vgan_gam_stream_next(vgan_gam_stream *st, int64_t min_reads, vgan_alnparts **out) {
    if (!st || !out) return fail(VGAN_EINVAL, "vgan_gam_stream_next: null argument");
    *out = nullptr;
    auto ps = new vgan_alnparts();
    const int rc = st->take(std::max<int64_t>(1, min_reads), ps->parts, &ps->base);
    if (rc || ps->parts.empty()) {
        delete ps;
        return rc; // VGAN_OK with *out == NULL: end of the stream
    }
    ps->index();
    *out = ps;
    return VGAN_OK;
}

extern "C" void vgan_gam_stream_close(vgan_gam_stream *st) { delete st; }

extern "C" void vgan_gam_decode_counts(int64_t out[5]) {
    if (!out) return;
    for (int i = 0; i < 5; ++i) out[i] = g_decode_counts[i].load(std::memory_order_relaxed);
}

extern "C" int64_t vgan_alnparts_base(const vgan_alnparts *p) { return p ? p->base : 0; }

extern "C" int64_t vgan_alnparts_n_reads(const vgan_alnparts *p) { return p ? p->first.back() : 0; }
extern "C" int64_t vgan_alnparts_count(const vgan_alnparts *p) { return p ? (int64_t)p->parts.size() : 0; }
extern "C" int64_t vgan_alnparts_first_read(const vgan_alnparts *p, int64_t i) {
    return (p && i >= 0 && i <= (int64_t)p->parts.size()) ? p->first[(size_t)i] : -1;
}
extern "C" void vgan_alnparts_free(vgan_alnparts *p) { delete p; }

extern "C" int vgan_alnparts_merge(vgan_alnparts *p, vgan_alnset **out) {
    if (!p || !out) return fail(VGAN_EINVAL, "vgan_alnparts_merge: null argument");
    auto a = new vgan_alnset();
    merge_alnsets(p->parts, *a); // consumes the slices
    p->parts.clear();
    p->first.assign(1, 0);
    *out = a;
    return VGAN_OK;
}

extern "C" int vgan_aln_read_gam(const char *path, int keep_unmapped, vgan_alnset **out) {
    if (!path || !out) return fail(VGAN_EINVAL, "vgan_aln_read_gam: null argument");
    MappedFile f;
    if (!f.open_path(path)) return fail(VGAN_EIO, "cannot read %s", path);
    return vgan_aln_parse_gam(f.p, f.n, keep_unmapped, out);
}

extern "C" int vgan_aln_write_gam(const vgan_alnset *a, const char *path, int group_size) {
    if (!a || !path) return fail(VGAN_EINVAL, "vgan_aln_write_gam: null argument");
    if (group_size <= 0) group_size = 512;
    std::string body, msg, pth, mp, pos, ed, file;
    const int64_t R = a->n_reads();
    for (int64_t g0 = 0; g0 < R; g0 += group_size) {
        const int64_t g1 = std::min<int64_t>(R, g0 + group_size);
        put_varint(body, (uint64_t)(g1 - g0 + 1));
        put_varint(body, 3);
        body += "GAM";
        for (int64_t r = g0; r < g1; ++r) {
            msg.clear();
            if (a->seq_off[r + 1] > a->seq_off[r]) put_ld(msg, 1, a->seq.data() + a->seq_off[r], a->seq_off[r + 1] - a->seq_off[r]);
            pth.clear();
            for (int64_t m = a->map_off[r]; m < a->map_off[r + 1]; ++m) {
                mp.clear();
                pos.clear();
                if (a->m_node[m]) put_vi(pos, 1, (uint64_t)a->m_node[m]);
                if (a->m_offset[m]) put_vi(pos, 2, (uint64_t)a->m_offset[m]);
                if (a->m_rev[m]) put_vi(pos, 4, 1);
                put_ld(mp, 1, pos);
                for (int64_t e = a->edit_off[m]; e < a->edit_off[m + 1]; ++e) {
                    ed.clear();
                    if (a->e_from[e]) put_vi(ed, 1, (uint64_t)(int64_t)a->e_from[e]);
                    if (a->e_to[e]) put_vi(ed, 2, (uint64_t)(int64_t)a->e_to[e]);
                    if (a->e_seq_off[e + 1] > a->e_seq_off[e])
                        put_ld(ed, 3, a->e_seq.data() + a->e_seq_off[e], a->e_seq_off[e + 1] - a->e_seq_off[e]);
                    put_ld(mp, 2, ed);
                }
                put_vi(mp, 5, (uint64_t)(m - a->map_off[r] + 1));
                put_ld(pth, 2, mp);
            }
            put_ld(msg, 2, pth);
            if (a->name_off[r + 1] > a->name_off[r]) put_ld(msg, 3, a->name.data() + a->name_off[r], a->name_off[r + 1] - a->name_off[r]);
            if (a->qual_off[r + 1] > a->qual_off[r]) put_ld(msg, 4, a->qual.data() + a->qual_off[r], a->qual_off[r + 1] - a->qual_off[r]);
            if (a->mapq[r]) put_vi(msg, 5, (uint64_t)a->mapq[r]);
            if (a->identity[r] != 0.0) {
                put_varint(msg, (16u << 3) | 1);
                msg.append((const char *)&a->identity[r], 8);
            }
            put_varint(body, msg.size());
            body += msg;
        }
    }
    if (!gzip_bytes(body, file)) return fail(VGAN_EIO, "BGZF compression failed"); // BGZF blocks, as vg writes GAM
    if (!write_file(path, file)) return fail(VGAN_EIO, "cannot write %s", path);
    return VGAN_OK;
}

extern "C" int vgan_aln_from_arrays(const vgan_alnset_view *v, vgan_alnset **out) {
    if (!v || !out || v->n_reads < 0) return fail(VGAN_EINVAL, "vgan_aln_from_arrays: bad view");
    auto a = new vgan_alnset();
    const int64_t R = v->n_reads;
    a->seq_off.assign(v->seq_off, v->seq_off + R + 1);
    a->seq.assign(v->seq, (size_t)v->seq_off[R]);
    a->qual_off.assign(v->qual_off, v->qual_off + R + 1);
    a->qual.assign(v->qual, (size_t)v->qual_off[R]);
    if (v->name_off) {
        a->name_off.assign(v->name_off, v->name_off + R + 1);
        a->name.assign(v->name, (size_t)v->name_off[R]);
    } else {
        a->name_off.assign((size_t)R + 1, 0);
    }
    a->mapq.assign(v->mapq, v->mapq + R);
    a->identity.assign(v->identity, v->identity + R);
    a->map_off.assign(v->map_off, v->map_off + R + 1);
    const int64_t M = v->map_off[R];
    a->m_node.assign(v->m_node, v->m_node + M);
    a->m_offset.assign(v->m_offset, v->m_offset + M);
    a->m_rev.assign(v->m_rev, v->m_rev + M);
    a->edit_off.assign(v->edit_off, v->edit_off + M + 1);
    const int64_t E = v->edit_off[M];
    a->e_from.assign(v->e_from, v->e_from + E);
    a->e_to.assign(v->e_to, v->e_to + E);
    a->e_seq_off.assign(v->e_seq_off, v->e_seq_off + E + 1);
    a->e_seq.assign(v->e_seq, (size_t)v->e_seq_off[E]);
    *out = a;
    return VGAN_OK;
}

extern "C" int vgan_aln_view_get(const vgan_alnset *a, vgan_alnset_view *out) {
    if (!a || !out) return fail(VGAN_EINVAL, "vgan_aln_view_get: null argument");
    a->fill_view(out);
    return VGAN_OK;
}

extern "C" void vgan_aln_free(vgan_alnset *a) { delete a; }

// Alignment messages lying one after the other in memory (message k = bytes[offsets[k], offsets[k + 1]): what the GAM front end on
// the device hands back for the reads its flatten leaves to the host, csrc/gam_kernels.hip) -> a sliced set of one slice, parsed by
// the parser every other input goes through.
extern "C" int vgan_alnparts_from_messages(const uint8_t *bytes, const uint64_t *offsets, int64_t n, int keep_unmapped, int n_threads, vgan_alnparts **out) {
    if (!out || n < 0 || (n > 0 && (!bytes || !offsets))) return fail(VGAN_EINVAL, "vgan_alnparts_from_messages: null argument");
    auto ps = new vgan_alnparts();
    // (slices of the messages on threads of their own: 135 k messages on one thread were 167 ms of a 10 M-read file's 1.45 s)
    const int64_t T = std::max<int64_t>(1, std::min<int64_t>({n_threads > 0 ? (int64_t)n_threads : (int64_t)vgan_host_cpus(), 32, n / 4096}));
    ps->parts.resize((size_t)T);
    std::vector<int64_t> bad((size_t)T, -1);
    auto slice = [&](int64_t t) {
        const int64_t k0 = n * t / T, k1 = n * (t + 1) / T;
        vgan_alnset &a = ps->parts[(size_t)t];
        try {
            if (k1 > k0 && offsets[k1] >= offsets[k0]) reserve_for(a, (size_t)(offsets[k1] - offsets[k0]), (size_t)(k1 - k0));
            for (int64_t k = k0; k < k1; ++k)
                if (offsets[k + 1] < offsets[k] || !parse_alignment(Cur{bytes + offsets[k], bytes + offsets[k + 1], true}, a, keep_unmapped)) {
                    bad[(size_t)t] = k;
                    return;
                }
        } catch (const std::bad_alloc &) { // (offsets that do not describe the bytes: a request for memory nobody has)
            bad[(size_t)t] = k0;
        }
    };
    parallel_run((int)T, [&](int t) { slice(t); });
    for (int64_t t = 0; t < T; ++t)
        if (bad[(size_t)t] >= 0) {
            const int64_t k = bad[(size_t)t];
            delete ps;
            return fail(VGAN_EIO, "vgan_alnparts_from_messages: message %lld is malformed", (long long)k);
        }
    ps->index();
    *out = ps;
    return VGAN_OK;
}

// Duplicate marking with the reference's single-end semantics (src/rmdup.cpp:20-41,68-110): a read is a
// duplicate when an EARLIER read of the set has the same (node id, offset) in its first mapping -- strand is not
// compared.  The reference does this in O(n^2); a hash of first-seen keys gives the same marks in O(n).
// (The reference's paired branch reads mapping()[n_mappings], one past the end: undefined, not reproduced.)
#include <unordered_set>

extern "C" int vgan_aln_mark_duplicates(const vgan_alnset *a, uint8_t *is_dup, int64_t *n_dup) {
    if (!a || !is_dup) return fail(VGAN_EINVAL, "vgan_aln_mark_duplicates: null argument");
    struct KeyHash {
        size_t operator()(const std::pair<int64_t, int64_t> &k) const {
            return (size_t)(k.first * 0x9E3779B97F4A7C15ull) ^ (size_t)(k.second + 0x7F4A7C15ull + ((uint64_t)k.first << 6));
        }
    };
    std::unordered_set<std::pair<int64_t, int64_t>, KeyHash> seen;
    seen.reserve((size_t)a->n_reads());
    int64_t nd = 0;
    for (int64_t r = 0; r < a->n_reads(); ++r) {
        is_dup[r] = 0;
        if (a->map_off[r + 1] == a->map_off[r]) continue; // the reference would index mapping()[0] of an empty path (UB)
        const int64_t m = a->map_off[r];
        if (!seen.insert({a->m_node[m], a->m_offset[m]}).second) {
            is_dup[r] = 1;
            ++nd;
        }
    }
    if (n_dup) *n_dup = nd;
    return VGAN_OK;
}

namespace {
struct DupKeyHash {
    size_t operator()(const std::pair<int64_t, int64_t> &k) const {
        return (size_t)(k.first * 0x9E3779B97F4A7C15ull) ^ (size_t)(k.second + 0x7F4A7C15ull + ((uint64_t)k.first << 6));
    }
};
} // namespace

// first-seen keys of everything marked so far: chunks of a stream are marked one after the other
struct vgan_dedup {
    std::unordered_set<std::pair<int64_t, int64_t>, DupKeyHash> seen;
};

extern "C" int vgan_dedup_create(vgan_dedup **out) {
    if (!out) return fail(VGAN_EINVAL, "vgan_dedup_create: null argument");
    *out = new vgan_dedup();
    return VGAN_OK;
}

extern "C" void vgan_dedup_free(vgan_dedup *d) { delete d; }

extern "C" int vgan_dedup_mark(vgan_dedup *d, const vgan_alnparts *ps, uint8_t *is_dup, int64_t *n_dup) {
    if (!d || !ps || !is_dup) return fail(VGAN_EINVAL, "vgan_dedup_mark: null argument");
    int64_t nd = 0;
    for (size_t i = 0; i < ps->parts.size(); ++i) {
        const vgan_alnset &a = ps->parts[i];
        uint8_t *m = is_dup + ps->first[i];
        for (int64_t r = 0; r < a.n_reads(); ++r) {
            m[r] = 0;
            if (a.map_off[r + 1] == a.map_off[r]) continue;
            const int64_t k = a.map_off[r];
            if (!d->seen.insert({a.m_node[k], a.m_offset[k]}).second) {
                m[r] = 1;
                ++nd;
            }
        }
    }
    if (n_dup) *n_dup = nd;
    return VGAN_OK;
}

extern "C" int vgan_alnparts_mark_duplicates(const vgan_alnparts *ps, uint8_t *is_dup, int64_t *n_dup) {
    if (!ps || !is_dup) return fail(VGAN_EINVAL, "vgan_alnparts_mark_duplicates: null argument");
    vgan_dedup d;
    d.seen.reserve((size_t)std::min<int64_t>(ps->first.back(), 1 << 22));
    return vgan_dedup_mark(&d, ps, is_dup, n_dup);
}

extern "C" int vgan_aln_filter(const vgan_alnset *a, const uint8_t *drop, vgan_alnset **out) {
    if (!a || !drop || !out) return fail(VGAN_EINVAL, "vgan_aln_filter: null argument");
    auto o = new vgan_alnset();
    for (int64_t r = 0; r < a->n_reads(); ++r) {
        if (drop[r]) continue;
        o->seq.append(a->seq, (size_t)a->seq_off[r], (size_t)(a->seq_off[r + 1] - a->seq_off[r]));
        o->seq_off.push_back((int64_t)o->seq.size());
        o->qual.append(a->qual, (size_t)a->qual_off[r], (size_t)(a->qual_off[r + 1] - a->qual_off[r]));
        o->qual_off.push_back((int64_t)o->qual.size());
        o->name.append(a->name, (size_t)a->name_off[r], (size_t)(a->name_off[r + 1] - a->name_off[r]));
        o->name_off.push_back((int64_t)o->name.size());
        o->mapq.push_back(a->mapq[r]);
        o->identity.push_back(a->identity[r]);
        for (int64_t m = a->map_off[r]; m < a->map_off[r + 1]; ++m) {
            o->m_node.push_back(a->m_node[m]);
            o->m_offset.push_back(a->m_offset[m]);
            o->m_rev.push_back(a->m_rev[m]);
            for (int64_t e = a->edit_off[m]; e < a->edit_off[m + 1]; ++e) {
                o->e_from.push_back(a->e_from[e]);
                o->e_to.push_back(a->e_to[e]);
                o->e_seq.append(a->e_seq, (size_t)a->e_seq_off[e], (size_t)(a->e_seq_off[e + 1] - a->e_seq_off[e]));
                o->e_seq_off.push_back((int64_t)o->e_seq.size());
            }
            o->edit_off.push_back((int64_t)o->e_from.size());
        }
        o->map_off.push_back((int64_t)o->m_node.size());
    }
    *out = o;
    return VGAN_OK;
}
