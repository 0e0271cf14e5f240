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
#include <cstring>
#include <thread>

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
    const size_t nt = std::min<size_t>(parts.size(), std::max(1u, std::thread::hardware_concurrency()));
    if (nt <= 1) {
        for (size_t i = 0; i < parts.size(); ++i) copy_part(i);
    } else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                for (size_t i = t; i < parts.size(); i += nt) copy_part(i);
            });
        for (auto &t : th) t.join();
    }
}

// GAM bytes -> slices of ~SLICE reads in input order, as a pipeline that runs behind the caller: block-parallel
// inflate (BGZF) hands its finished prefix to the serial framing pass, which hands slices of messages to the wire
// parser pool; take() returns the next parsed slices as soon as they exist.
struct vgan_gam_stream {
    using Msg = std::pair<const uint8_t *, const uint8_t *>;
    static constexpr size_t SLICE = 8192;
    MappedFile file;
    int keep_unmapped = 0;
    ByteBuf inflated;
    AsyncInflate bg; // BGZF (what vg writes): framing runs on the prefix inflated so far
    bool streaming = false;
    const uint8_t *p = nullptr;
    size_t n = 0;
    std::deque<std::vector<Msg>> slices; // grown by the framing thread only, under mu
    std::deque<vgan_alnset> slice_parts;
    std::deque<uint8_t> parsed;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    size_t next_slice = 0, delivered = 0;
    int64_t delivered_reads = 0;
    bool framing_done = false, framed = true, inflate_ok = true;
    std::atomic<bool> ok{true};
    std::thread framer;
    std::vector<std::thread> workers;
    PhaseTimer pt{"parse_gam"};

    ~vgan_gam_stream() { join(); }

    void join() {
        if (framer.joinable()) framer.join();
        for (auto &t : workers)
            if (t.joinable()) t.join();
        workers.clear();
        (void)bg.finish();
    }

    int start(const void *bytes, size_t nbytes, int keep) {
        if (!bytes) return fail(VGAN_EINVAL, "GAM: null buffer");
        keep_unmapped = keep;
        p = (const uint8_t *)bytes;
        n = nbytes;
        if (n >= 2 && p[0] == 0x1f && p[1] == 0x8b) {
            if (bg.start(bytes, n)) {
                streaming = true;
                p = (const uint8_t *)bg.out.data();
                n = bg.out.size();
            } else {
                if (!gunzip_members(bytes, n, inflated)) return fail(VGAN_EIO, "GAM: gzip stream is corrupt");
                p = (const uint8_t *)inflated.data();
                n = inflated.size();
            }
        }
        pt.lap(streaming ? "inflate started" : "inflate");
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const unsigned nt = (unsigned)std::min<size_t>(std::min(hw, 64u), std::max<size_t>(1, n / (4u << 20)));
        for (unsigned t = 0; t < nt; ++t) workers.emplace_back([this] { work(); });
        framer = std::thread([this] { frame(); });
        return VGAN_OK;
    }

    void parse_slice(const std::vector<Msg> &ms, vgan_alnset &a) {
        if (ms.empty()) return;
        // reserve from the byte volume of the slice: ~1 mapping per 20 bytes, ~1 edit per 16
        const size_t nbytes = (size_t)(ms.back().second - ms.front().first), nr = ms.size();
        a.seq_off.reserve(nr + 1);
        a.qual_off.reserve(nr + 1);
        a.name_off.reserve(nr + 1);
        a.map_off.reserve(nr + 1);
        a.mapq.reserve(nr);
        a.identity.reserve(nr);
        a.seq.reserve(nbytes / 6);
        a.qual.reserve(nbytes / 6);
        a.name.reserve(nr * 16);
        a.m_node.reserve(nbytes / 18);
        a.m_offset.reserve(nbytes / 18);
        a.m_rev.reserve(nbytes / 18);
        a.edit_off.reserve(nbytes / 18);
        a.e_from.reserve(nbytes / 14);
        a.e_to.reserve(nbytes / 14);
        a.e_seq_off.reserve(nbytes / 14);
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

    void frame() { // serial: groups of {count, count x (len, bytes)}, the first item of a group possibly the tag "GAM"
        const uint8_t *const base = p;
        auto need = [&](const uint8_t *upto) { // bytes [0, upto) must be final before they are read
            if (streaming && inflate_ok) inflate_ok = bg.wait_for((size_t)(upto - base));
        };
        auto open_slice = [&]() {
            std::lock_guard<std::mutex> lk(mu);
            slices.emplace_back();
            slices.back().reserve(SLICE);
            slice_parts.emplace_back();
            parsed.push_back(0);
        };
        open_slice();
        std::vector<Msg> *cur = &slices.back();
        Cur c{p, p + n, true};
        while (!c.done() && inflate_ok) {
            need(c.p + 10); // a varint
            const uint64_t count = c.varint();
            if (!c.ok) break;
            bool first = true;
            for (uint64_t i = 0; i < count && c.ok && inflate_ok; ++i) {
                need(c.p + 10);
                Cur item = c.sub(); // the length only: the payload is not touched here
                if (!c.ok) break;
                if (first) {
                    first = false;
                    if (item.e - item.p == 3) {
                        need(item.e);
                        if (memcmp(item.p, "GAM", 3) == 0) continue;
                    }
                }
                cur->emplace_back(item.p, item.e);
                if (cur->size() == SLICE) {
                    need(item.e); // the parsers read the payloads
                    open_slice();
                    cur = &slices.back();
                    cv_work.notify_one();
                }
            }
            if (!c.ok) break;
        }
        need(p + n);
        {
            std::lock_guard<std::mutex> lk(mu);
            framed = c.ok && inflate_ok;
            framing_done = true;
        }
        pt.lap("framing");
        cv_work.notify_all();
        cv_done.notify_all();
    }

    // The next parsed slices in input order holding at least min_reads reads (fewer at the end); empty at the end.
    int take(int64_t min_reads, std::vector<vgan_alnset> &out, int64_t *first_read) {
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
        const bool bad_frame = framing_done && !framed, bad_inflate = framing_done && !inflate_ok;
        lk.unlock();
        if (bad_inflate) return fail(VGAN_EIO, "GAM: gzip stream is corrupt");
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
    if (!st.bg.finish()) return fail(VGAN_EIO, "GAM: gzip stream is corrupt");
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
