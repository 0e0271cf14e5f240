// euka host side: clade / bin tables (reference src/load.cpp:71-157), damage profiles (src/miscfunc.h:84-136,
// src/damage.cpp:41-136), the front half of readGAM3's lambda (src/readGAM_Euka.h:67-216) as an SoA batch, and a
// synthetic clade graph with aDNA-like reads.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>
#include <thread>

using namespace vgan;

struct vgan_euka_db {
    std::vector<int32_t> clade_id, clade_npaths, clade_snode, clade_enode;
    std::vector<double> clade_dist;
    std::string clade_names;
    std::vector<uint32_t> bin_off{0};
    std::vector<int32_t> bin_lo, bin_hi;
    std::vector<double> bin_entropy;
};

struct vgan_damage {
    std::vector<double> sub5p, sub3p; // [n][16]
};

struct vgan_euka_host_batch {
    std::vector<uint32_t> read_col_off{0}, read_qual_off{0}, read_map_off{0}, read_src, map_node;
    std::vector<uint16_t> read_gseq_len, read_rseq_len, read_seq_len;
    std::vector<int32_t> read_mapq;
    std::vector<uint8_t> read_rev, graph_seq, read_seq, qual;
};

namespace {

std::vector<std::string> ws_tokens(const std::string &line) {
    std::vector<std::string> t;
    std::istringstream ls(line);
    std::string tok;
    while (ls >> tok) t.push_back(tok);
    return t;
}

// std::stoi semantics on "1836.0": leading integer part
bool stoi_like(const std::string &s, int32_t &v) {
    char *end = nullptr;
    const long x = strtol(s.c_str(), &end, 10);
    if (end == s.c_str()) return false;
    v = (int32_t)x;
    return true;
}

int parse_db(const std::string &clade_txt, const std::string &bins_txt, vgan_euka_db &d) {
    std::istringstream in(clade_txt);
    std::string line;
    while (std::getline(in, line)) {
        const auto t = ws_tokens(line);
        if (t.empty()) continue;
        if (t.size() != 6) return fail(VGAN_EIO, "clade file: line has %zu fields rather than 6", t.size()); // load.cpp:121 assert
        int32_t id, np, sn, en;
        if (!stoi_like(t[0], id) || !stoi_like(t[3], np) || !stoi_like(t[4], sn) || !stoi_like(t[5], en))
            return fail(VGAN_EIO, "clade file: non-numeric field");
        d.clade_id.push_back(id);
        d.clade_names += t[1] + "\n";
        d.clade_dist.push_back(strtod(t[2].c_str(), nullptr));
        d.clade_npaths.push_back(np);
        d.clade_snode.push_back(sn);
        d.clade_enode.push_back(en);
    }
    std::istringstream inb(bins_txt);
    while (std::getline(inb, line)) {
        const auto t = ws_tokens(line);
        for (size_t j = 1; j + 2 < t.size(); j += 3) { // load.cpp:81-88 (an incomplete trailing triple is ignored)
            int32_t lo, hi;
            if (!stoi_like(t[j], lo) || !stoi_like(t[j + 1], hi)) return fail(VGAN_EIO, "bins file: non-numeric bound");
            d.bin_lo.push_back(lo);
            d.bin_hi.push_back(hi);
            d.bin_entropy.push_back(strtod(t[j + 2].c_str(), nullptr));
        }
        d.bin_off.push_back((uint32_t)d.bin_lo.size());
    }
    if (d.bin_off.size() - 1 != d.clade_id.size())
        return fail(VGAN_EIO, "bins file has %zu lines for %zu clades", d.bin_off.size() - 1, d.clade_id.size());
    return VGAN_OK;
}

// 12-column profile -> row-stochastic 4x4 per position (damage.cpp:66-88); index of X>Y = n1*3 + (n2 > n1 ? n2-1 : n2)
int parse_profile(const char *text, std::vector<double> &out) {
    out.clear();
    if (!text || !*text) { // no file: zero rates (damage.cpp:47-55) -> identity
        out.assign(16, 0.0);
        for (int i = 0; i < 4; ++i) out[(size_t)i * 5] = 1.0;
        return VGAN_OK;
    }
    std::istringstream in(text);
    std::string line;
    auto fields_of = [](const std::string &l) {
        std::vector<std::string> f;
        size_t p = 0;
        while (true) {
            const size_t q = l.find('\t', p);
            if (q == std::string::npos) {
                f.push_back(l.substr(p));
                break;
            }
            f.push_back(l.substr(p, q - p));
            p = q + 1;
        }
        if (f.size() == 13) f.pop_back();
        return f;
    };
    if (!std::getline(in, line)) return fail(VGAN_EIO, "damage profile is empty");
    if (fields_of(line).size() != 12) return fail(VGAN_EIO, "line from error profile has %zu fields rather than 12", fields_of(line).size());
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const auto f = fields_of(line);
        if (f.size() != 12) return fail(VGAN_EIO, "line from error profile has %zu fields rather than 12", f.size());
        long double s[12];
        for (int k = 0; k < 12; ++k) s[k] = strtold(f[(size_t)k].c_str(), nullptr);
        double m[16];
        for (int n1 = 0; n1 < 4; ++n1) {
            double ident = 1.0;
            for (int n2 = 0; n2 < 4; ++n2) {
                if (n1 == n2) continue;
                const int ind = n1 * 3 + (n2 > n1 ? n2 - 1 : n2);
                ident = (double)(ident - s[ind]);
                m[n1 * 4 + n2] = (double)s[ind];
            }
            if (ident < 0) return fail(VGAN_EIO, "Error with deamination profile, identity probability is less than 0");
            m[n1 * 5] = ident;
        }
        out.insert(out.end(), m, m + 16);
    }
    if (out.empty()) return fail(VGAN_EIO, "damage profile has no rows");
    if (out.size() / 16 > 1000) out.resize(16 * 1000); // positions beyond MAXLENGTHFRAGMENT are never read
    return VGAN_OK;
}

struct EChunk {
    vgan_euka_host_batch b;
    vgan_euka_flatten_stats st{};
};

void euka_flatten_range(const vgan_graph &g, const vgan_alnset &a, int64_t r0, int64_t r1, EChunk &c) {
    Recon rc;
    auto &b = c.b;
    for (int64_t r = r0; r < r1; ++r) {
        c.st.n_in++;
        if (a.identity[r] == 0) { // readGAM_Euka.h:72
            c.st.n_unmapped++;
            continue;
        }
        const int64_t nm = a.map_off[r + 1] - a.map_off[r];
        const int64_t Lseq = a.seq_off[r + 1] - a.seq_off[r];
        int bad = nm == 0 ? 1 : reconstruct(g, a, r, rc);
        if (!bad && (Lseq < 15 || Lseq > 1000)) bad = 1; // subDeamDiNuc[Lseq] empty / out of range (damage.h:42-43)
        if (!bad && (rc.gseq.size() > 65535 || rc.ps.size() > 65535)) bad = 1;
        if (bad) {
            c.st.n_bad++;
            continue;
        }
        const size_t G = rc.gseq.size(), A = rc.ps.size(), region = std::max(G, A);
        b.graph_seq.insert(b.graph_seq.end(), rc.gseq.begin(), rc.gseq.end());
        b.graph_seq.insert(b.graph_seq.end(), region - G, 0);
        b.read_seq.insert(b.read_seq.end(), rc.ps.begin(), rc.ps.end());
        b.read_seq.insert(b.read_seq.end(), region - A, 0);
        const char *q = a.qual.data() + a.qual_off[r];
        b.qual.insert(b.qual.end(), q, q + (a.qual_off[r + 1] - a.qual_off[r]));
        for (int64_t m = a.map_off[r]; m < a.map_off[r + 1]; ++m) b.map_node.push_back((uint32_t)a.m_node[m]);
        b.read_gseq_len.push_back((uint16_t)G);
        b.read_rseq_len.push_back((uint16_t)A);
        b.read_seq_len.push_back((uint16_t)Lseq);
        b.read_mapq.push_back(a.mapq[r]);
        b.read_rev.push_back(a.m_rev[a.map_off[r]]);
        b.read_src.push_back((uint32_t)r);
        b.read_col_off.push_back((uint32_t)b.graph_seq.size());
        b.read_qual_off.push_back((uint32_t)b.qual.size());
        b.read_map_off.push_back((uint32_t)b.map_node.size());
        c.st.n_out++;
    }
}

template <class T> void cat_shift(std::vector<T> &dst, const std::vector<T> &src, T shift) {
    for (size_t i = 1; i < src.size(); ++i) dst.push_back((T)(src[i] + shift));
}
template <class T> void cat(std::vector<T> &dst, const std::vector<T> &src) { dst.insert(dst.end(), src.begin(), src.end()); }

} // namespace

extern "C" int vgan_euka_db_load(const char *clade_path, const char *bins_path, vgan_euka_db **out) {
    if (!clade_path || !bins_path || !out) return fail(VGAN_EINVAL, "vgan_euka_db_load: null argument");
    std::string ct, bt;
    if (!read_text_maybe_gz(clade_path, ct)) return fail(VGAN_EIO, "cannot read %s", clade_path);
    if (!read_text_maybe_gz(bins_path, bt)) return fail(VGAN_EIO, "cannot read %s", bins_path);
    auto d = new vgan_euka_db();
    const int rc = parse_db(ct, bt, *d);
    if (rc) {
        delete d;
        return rc;
    }
    *out = d;
    return VGAN_OK;
}

extern "C" int vgan_euka_db_from_arrays(const vgan_euka_db_view *v, vgan_euka_db **out) {
    if (!v || !out) return fail(VGAN_EINVAL, "vgan_euka_db_from_arrays: null argument");
    auto d = new vgan_euka_db();
    const uint32_t C = v->n_clades;
    d->clade_dist.assign(v->clade_dist, v->clade_dist + C);
    auto opt = [&](const int32_t *p, std::vector<int32_t> &dst) {
        if (p) dst.assign(p, p + C);
        else dst.assign(C, 0);
    };
    opt(v->clade_id, d->clade_id);
    opt(v->clade_npaths, d->clade_npaths);
    opt(v->clade_snode, d->clade_snode);
    opt(v->clade_enode, d->clade_enode);
    d->clade_names = v->clade_names ? v->clade_names : "";
    d->bin_off.assign(v->bin_off, v->bin_off + C + 1);
    const uint32_t nb = v->bin_off[C];
    d->bin_lo.assign(v->bin_lo, v->bin_lo + nb);
    d->bin_hi.assign(v->bin_hi, v->bin_hi + nb);
    if (v->bin_entropy) d->bin_entropy.assign(v->bin_entropy, v->bin_entropy + nb);
    else d->bin_entropy.assign(nb, 0.0);
    *out = d;
    return VGAN_OK;
}

extern "C" int vgan_euka_db_view_get(const vgan_euka_db *d, vgan_euka_db_view *v) {
    if (!d || !v) return fail(VGAN_EINVAL, "vgan_euka_db_view_get: null argument");
    v->n_clades = (uint32_t)d->clade_dist.size();
    v->clade_id = d->clade_id.data();
    v->clade_dist = d->clade_dist.data();
    v->clade_npaths = d->clade_npaths.data();
    v->clade_snode = d->clade_snode.data();
    v->clade_enode = d->clade_enode.data();
    v->clade_names = d->clade_names.c_str();
    v->bin_off = d->bin_off.data();
    v->bin_lo = d->bin_lo.data();
    v->bin_hi = d->bin_hi.data();
    v->bin_entropy = d->bin_entropy.data();
    return VGAN_OK;
}

extern "C" void vgan_euka_db_free(vgan_euka_db *d) { delete d; }

extern "C" int vgan_damage_from_text(const char *p5, const char *p3, vgan_damage **out) {
    if (!out) return fail(VGAN_EINVAL, "vgan_damage_from_text: null argument");
    auto d = new vgan_damage();
    int rc;
    if ((rc = parse_profile(p5, d->sub5p)) || (rc = parse_profile(p3, d->sub3p))) {
        delete d;
        return rc;
    }
    *out = d;
    return VGAN_OK;
}

extern "C" int vgan_damage_load(const char *path5, const char *path3, vgan_damage **out) {
    std::string t5, t3;
    if (path5 && *path5 && !read_text_maybe_gz(path5, t5)) return fail(VGAN_EIO, "Unable to open file %s", path5);
    if (path3 && *path3 && !read_text_maybe_gz(path3, t3)) return fail(VGAN_EIO, "Unable to open file %s", path3);
    return vgan_damage_from_text(t5.c_str(), t3.c_str(), out);
}

extern "C" int vgan_damage_view_get(const vgan_damage *d, vgan_damage_view *v) {
    if (!d || !v) return fail(VGAN_EINVAL, "vgan_damage_view_get: null argument");
    v->n5 = (uint32_t)(d->sub5p.size() / 16);
    v->n3 = (uint32_t)(d->sub3p.size() / 16);
    v->sub5p = d->sub5p.data();
    v->sub3p = d->sub3p.data();
    return VGAN_OK;
}

extern "C" void vgan_damage_free(vgan_damage *d) { delete d; }

extern "C" int vgan_euka_flatten(const vgan_graph *g, const vgan_alnset *a, int64_t r0, int64_t r1, int n_threads,
                                 vgan_euka_host_batch **out, vgan_euka_flatten_stats *stats) {
    if (!g || !a || !out) return fail(VGAN_EINVAL, "vgan_euka_flatten: null argument");
    if (r0 < 0 || r1 > a->n_reads() || r0 > r1) return fail(VGAN_EINVAL, "vgan_euka_flatten: bad read range");
    if (n_threads <= 0) n_threads = (int)usable_cpus();
    const int64_t n = r1 - r0;
    n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, (n + 4095) / 4096));
    std::vector<EChunk> chunks((size_t)n_threads);
    PhaseTimer pt("euka_flatten");
    parallel_run(n_threads, [&](int t) {
        const int64_t b0 = r0 + n * t / n_threads, b1 = r0 + n * (t + 1) / n_threads;
        euka_flatten_range(*g, *a, b0, b1, chunks[(size_t)t]);
    });
    pt.lap("chunks");
    auto res = new vgan_euka_host_batch();
    vgan_euka_flatten_stats st{};
    uint64_t tc = 0, tq = 0, tm = 0;
    size_t R = 0;
    for (auto &c : chunks) {
        tc += c.b.graph_seq.size();
        tq += c.b.qual.size();
        tm += c.b.map_node.size();
        R += c.b.read_mapq.size();
        st.n_in += c.st.n_in;
        st.n_out += c.st.n_out;
        st.n_unmapped += c.st.n_unmapped;
        st.n_bad += c.st.n_bad;
    }
    if (tc > 0xFFFFFFF0ull || tq > 0xFFFFFFF0ull || tm > 0xFFFFFFF0ull) {
        delete res;
        return fail(VGAN_ERANGE, "vgan_euka_flatten: batch exceeds 32-bit offsets; flatten fewer reads per batch");
    }
    // The reads go to the device in ascending order of their first mapping's node id (stable; read_src says which read of
    // the input each one is): clades own contiguous node ranges, so a wave of the read kernel sees one clade for long
    // stretches and keeps that clade's counters in LDS instead of adding to the global tables read by read.
    // Straight from the threads' chunks into the ordered batch: keys gathered once, two stable 16-bit counting passes (a
    // comparison sort whose comparator chases two arrays per key took 0.2-0.3 s per million reads on one thread), the
    // destination offsets by one prefix pass, the bytes moved in parallel -- no concatenated copy in between.
    std::vector<uint32_t> keys(R), order(R), tmp(R), chunk_of(R), local_of(R);
    {
        size_t i = 0;
        for (size_t c = 0; c < chunks.size(); ++c) {
            const auto &cb = chunks[c].b;
            for (size_t j = 0; j < cb.read_mapq.size(); ++j, ++i) {
                keys[i] = cb.read_map_off[j] < cb.read_map_off[j + 1] ? cb.map_node[cb.read_map_off[j]] : 0u;
                chunk_of[i] = (uint32_t)c;
                local_of[i] = (uint32_t)j;
            }
        }
        std::vector<uint32_t> cnt(65537);
        for (int pass = 0; pass < 2; ++pass) {
            const int sh = 16 * pass;
            std::fill(cnt.begin(), cnt.end(), 0u);
            for (size_t k = 0; k < R; ++k) cnt[((keys[k] >> sh) & 0xFFFFu) + 1u]++;
            for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];
            if (pass == 0) {
                for (size_t k = 0; k < R; ++k) tmp[cnt[keys[k] & 0xFFFFu]++] = (uint32_t)k;
            } else {
                for (size_t k = 0; k < R; ++k) order[cnt[(keys[tmp[k]] >> 16) & 0xFFFFu]++] = tmp[k];
            }
        }
    }
    pt.lap("order");
    res->read_col_off.assign(R + 1, 0);
    res->read_qual_off.assign(R + 1, 0);
    res->read_map_off.assign(R + 1, 0);
    for (size_t i = 0; i < R; ++i) {
        const auto &cb = chunks[chunk_of[order[i]]].b;
        const uint32_t j = local_of[order[i]];
        res->read_col_off[i + 1] = res->read_col_off[i] + (cb.read_col_off[j + 1] - cb.read_col_off[j]);
        res->read_qual_off[i + 1] = res->read_qual_off[i] + (cb.read_qual_off[j + 1] - cb.read_qual_off[j]);
        res->read_map_off[i + 1] = res->read_map_off[i] + (cb.read_map_off[j + 1] - cb.read_map_off[j]);
    }
    res->read_gseq_len.resize(R);
    res->read_rseq_len.resize(R);
    res->read_seq_len.resize(R);
    res->read_mapq.resize(R);
    res->read_rev.resize(R);
    res->read_src.resize(R);
    res->map_node.resize((size_t)tm);
    res->graph_seq.resize((size_t)tc);
    res->read_seq.resize((size_t)tc);
    res->qual.resize((size_t)tq);
    auto move_range = [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; ++i) {
            const auto &cb = chunks[chunk_of[order[i]]].b;
            const uint32_t j = local_of[order[i]];
            res->read_gseq_len[i] = cb.read_gseq_len[j];
            res->read_rseq_len[i] = cb.read_rseq_len[j];
            res->read_seq_len[i] = cb.read_seq_len[j];
            res->read_mapq[i] = cb.read_mapq[j];
            res->read_rev[i] = cb.read_rev[j];
            res->read_src[i] = cb.read_src[j];
            const size_t nc = cb.read_col_off[j + 1] - cb.read_col_off[j], nq = cb.read_qual_off[j + 1] - cb.read_qual_off[j],
                         nm = cb.read_map_off[j + 1] - cb.read_map_off[j];
            if (nc) {
                memcpy(&res->graph_seq[res->read_col_off[i]], &cb.graph_seq[cb.read_col_off[j]], nc);
                memcpy(&res->read_seq[res->read_col_off[i]], &cb.read_seq[cb.read_col_off[j]], nc);
            }
            if (nq) memcpy(&res->qual[res->read_qual_off[i]], &cb.qual[cb.read_qual_off[j]], nq);
            if (nm) memcpy(&res->map_node[res->read_map_off[i]], &cb.map_node[cb.read_map_off[j]], nm * sizeof(uint32_t));
        }
    };
    const int nth = (int)std::max<size_t>(1, std::min<size_t>((size_t)burst_cpus(), (R + 16383) / 16384));
    parallel_run(nth, [&](int t) { move_range(R * (size_t)t / (size_t)nth, R * ((size_t)t + 1) / (size_t)nth); });
    pt.lap("reorder");
    if (stats) *stats = st;
    *out = res;
    return VGAN_OK;
}

extern "C" int vgan_euka_host_batch_get(const vgan_euka_host_batch *b, vgan_euka_batch *o) {
    if (!b || !o) return fail(VGAN_EINVAL, "vgan_euka_host_batch_get: null argument");
    memset(o, 0, sizeof *o);
    o->n_reads = (uint32_t)b->read_mapq.size();
    o->n_cols = b->graph_seq.size();
    o->n_qual = b->qual.size();
    o->n_maps = b->map_node.size();
    o->read_col_off = b->read_col_off.data();
    o->read_qual_off = b->read_qual_off.data();
    o->read_map_off = b->read_map_off.data();
    o->read_gseq_len = b->read_gseq_len.data();
    o->read_rseq_len = b->read_rseq_len.data();
    o->read_seq_len = b->read_seq_len.data();
    o->read_mapq = b->read_mapq.data();
    o->read_rev = b->read_rev.data();
    o->read_src = b->read_src.data();
    o->map_node = b->map_node.data();
    o->graph_seq = b->graph_seq.data();
    o->read_seq = b->read_seq.data();
    o->qual = b->qual.data();
    return VGAN_OK;
}

extern "C" void vgan_euka_host_batch_free(vgan_euka_host_batch *b) { delete b; }

// ---------------------------------------------------------------------------------------------- synthetic euka input
extern "C" int vgan_synth_euka(const vgan_synth_euka_cfg *cfg, const vgan_damage *dmg, vgan_graph **gout, vgan_euka_db **dbout,
                               vgan_alnset **rout) {
    if (!cfg || !gout || !dbout || !rout) return fail(VGAN_EINVAL, "vgan_synth_euka: null argument");
    const uint32_t C = cfg->n_clades, NPC = cfg->nodes_per_clade;
    if (C < 1 || NPC < 40) return fail(VGAN_EINVAL, "vgan_synth_euka: need >= 1 clade and >= 40 nodes per clade");
    static const char B[4] = {'A', 'C', 'G', 'T'};
    SplitMix64 rg(cfg->seed ^ 0x65756b61ull);
    // graph: per clade a chain of backbone nodes (<= 5 bp) with ~15 % SNP bubbles; ids contiguous per clade
    auto g = new vgan_graph();
    auto db = new vgan_euka_db();
    g->min_id = 1;
    g->node_seq_off.assign(2, 0);
    struct Site {
        uint32_t ref, alt; // node ids (alt = 0: no bubble)
    };
    std::vector<std::vector<Site>> sites(C);
    uint32_t id = 1;
    for (uint32_t c = 0; c < C; ++c) {
        const uint32_t first = id;
        while (id - first + 2 <= NPC) {
            const uint32_t len = 1 + (uint32_t)rg.below(5);
            std::string s(len, 'A');
            for (auto &ch : s) ch = B[rg.below(4)];
            Site st{id, 0};
            g->node_seq += s;
            g->node_seq_off.push_back((int64_t)g->node_seq.size());
            ++id;
            if (rg.uniform() < .15 && id - first + 1 <= NPC) {
                std::string t = s;
                const size_t k = rg.below(len);
                char x;
                do x = B[rg.below(4)];
                while (x == t[k]);
                t[k] = x;
                st.alt = id;
                g->node_seq += t;
                g->node_seq_off.push_back((int64_t)g->node_seq.size());
                ++id;
            }
            sites[c].push_back(st);
        }
        const uint32_t last = id - 1;
        db->clade_id.push_back((int32_t)c);
        char nm[32];
        snprintf(nm, sizeof nm, "clade%03u", c);
        db->clade_names += std::string(nm) + "\n";
        db->clade_dist.push_back(.05 + .21 * rg.uniform()); // observed range of euka_db.clade
        db->clade_npaths.push_back(2 + (int32_t)rg.below(100));
        db->clade_snode.push_back((int32_t)first);
        db->clade_enode.push_back((int32_t)last);
        // 10-14 overlapping bins over the clade's node range, as in euka_db.bins
        const uint32_t nb = 10 + (uint32_t)rg.below(5);
        const double span = (double)(last - first + 1) / nb;
        for (uint32_t j = 0; j < nb; ++j) {
            const int32_t lo = (int32_t)(first + std::max(0.0, j * span - .1 * span));
            const int32_t hi = j + 1 == nb ? (int32_t)last : (int32_t)(first + (j + 1) * span + .1 * span);
            db->bin_lo.push_back(lo);
            db->bin_hi.push_back(std::min<int32_t>(hi, (int32_t)last));
            db->bin_entropy.push_back(1.1 + .25 * rg.uniform());
        }
        db->bin_off.push_back((uint32_t)db->bin_lo.size());
    }
    g->max_id = id - 1;
    g->n_paths = 0;
    g->mask_words = 0;
    g->pangenome_base.assign((size_t)g->max_id + 1, -1);
    // reads
    vgan_damage_view dv{};
    if (dmg) vgan_damage_view_get(dmg, &dv);
    auto a = new vgan_alnset();
    const uint64_t R = cfg->n_reads;
    const uint64_t rseed = cfg->read_seed ? cfg->read_seed : cfg->seed;
    for (uint64_t r = 0; r < R; ++r) {
        SplitMix64 q(rseed * 0x9E3779B97F4A7C15ull + r * 0xD1B54A32D192ED03ull + 0x61646e61ull);
        const uint32_t c = (uint32_t)q.below(C);
        const auto &st = sites[c];
        // aDNA-like fragment length: mean read_len_mean, clipped 30..150
        double u = q.uniform() + q.uniform() + q.uniform() - 1.5; // ~N(0, .5)
        int L = (int)std::lround(cfg->read_len_mean * (1.0 + .45 * u));
        L = std::max(30, std::min(150, L));
        struct Piece {
            uint32_t node, off, len;
        };
        std::vector<Piece> pieces;
        std::string fwd;
        size_t s = q.below(st.size());
        uint32_t off = 0;
        {
            const uint32_t nd = st[s].ref;
            off = (uint32_t)q.below((uint64_t)g->seq_len(nd));
        }
        const bool donor_alt = q.uniform() < .5; // the fragment's source carries alt alleles at half of the bubbles
        while ((int)fwd.size() < L && s < st.size()) {
            const uint32_t nd = (st[s].alt && donor_alt && (q.next() & 1)) ? st[s].alt : st[s].ref;
            const uint32_t nl = (uint32_t)g->seq_len(nd);
            if (off < nl) {
                const uint32_t take = std::min<uint32_t>(nl - off, (uint32_t)(L - (int)fwd.size()));
                pieces.push_back({nd, off, take});
                fwd.append(g->seq_ptr(nd) + off, take);
            }
            off = 0;
            ++s;
        }
        if ((int)fwd.size() < 15) { // too close to the end of the clade: restart at its beginning
            pieces.clear();
            fwd.clear();
            for (s = 0; (int)fwd.size() < L && s < st.size(); ++s) {
                const uint32_t nd = st[s].ref, nl = (uint32_t)g->seq_len(nd);
                const uint32_t take = std::min<uint32_t>(nl, (uint32_t)(L - (int)fwd.size()));
                pieces.push_back({nd, 0, take});
                fwd.append(g->seq_ptr(nd), take);
            }
        }
        const size_t n = fwd.size();
        const bool rev = q.uniform() < .5;
        std::string readseq = fwd;
        if (rev) {
            std::reverse(readseq.begin(), readseq.end());
            for (auto &ch : readseq) ch = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : 'A';
            std::reverse(pieces.begin(), pieces.end());
            for (auto &pc : pieces) pc.off = (uint32_t)g->seq_len(pc.node) - (pc.off + pc.len);
        }
        // divergence from the graph (substitutions), then deamination by position from the fragment ends, then errors
        std::string qual(n, (char)35);
        std::vector<uint8_t> is_sub(n, 0);
        for (size_t i = 0; i < n; ++i) {
            const double uq = q.uniform();
            qual[i] = (char)(uq < .05 ? 2 + (int)q.below(20) : 30 + (int)q.below(11));
            char base = readseq[i];
            const char orig = base;
            if (q.uniform() < .02) {
                char x;
                do x = B[q.below(4)];
                while (x == base);
                base = x;
            }
            if (dv.n5 && dv.n3) {
                const int b1 = base == 'A' ? 0 : base == 'C' ? 1 : base == 'G' ? 2 : 3;
                const double *m5 = dv.sub5p + 16 * std::min<size_t>(i, dv.n5 - 1) + 4 * b1;
                const double *m3 = dv.sub3p + 16 * std::min<size_t>(n - 1 - i, dv.n3 - 1) + 4 * b1;
                const double *row = m5[b1] <= m3[b1] ? m5 : m3;
                double x = q.uniform();
                int b2 = 0;
                while (b2 < 3 && x >= row[b2]) x -= row[b2++];
                base = B[b2];
            }
            if (q.uniform() < std::pow(10.0, -0.1 * qual[i])) {
                char x;
                do x = B[q.below(4)];
                while (x == base);
                base = x;
            }
            readseq[i] = base;
            is_sub[i] = base != orig;
        }
        const bool softclip = q.uniform() < .02;
        const uint32_t clip_len = softclip ? 3 + (uint32_t)q.below(8) : 0;
        const bool indel = q.uniform() < .02;
        const bool is_ins = indel && q.uniform() < .5;
        const uint32_t indel_piece = indel ? (uint32_t)q.below(pieces.size()) : 0;
        std::string seq_out, qual_out;
        size_t rp = 0;
        uint32_t n_match = 0;
        for (size_t k = 0; k < pieces.size(); ++k) {
            const Piece &pc = pieces[k];
            a->m_node.push_back(pc.node);
            a->m_offset.push_back(pc.off);
            a->m_rev.push_back(rev);
            auto push_edit = [&](int32_t from, int32_t to, const char *sq, size_t sl) {
                a->e_from.push_back(from);
                a->e_to.push_back(to);
                if (sl) a->e_seq.append(sq, sl);
                a->e_seq_off.push_back((int64_t)a->e_seq.size());
            };
            if (k == 0 && softclip) {
                std::string clip(clip_len, 'A');
                for (auto &ch : clip) ch = B[q.below(4)];
                push_edit(0, (int32_t)clip_len, clip.data(), clip.size());
                seq_out += clip;
                qual_out.append(clip_len, (char)20);
            }
            uint32_t done = 0;
            bool did_indel = !(indel && k == indel_piece);
            while (done < pc.len) {
                if (!did_indel && done == pc.len / 2) {
                    did_indel = true;
                    if (is_ins) {
                        const char x = B[q.below(4)];
                        push_edit(0, 1, &x, 1);
                        seq_out += x;
                        qual_out += (char)30;
                        continue;
                    } else if (pc.len - done > 1) {
                        push_edit(1, 0, nullptr, 0);
                        done += 1;
                        rp += 1;
                        continue;
                    }
                }
                uint32_t stop = pc.len;
                if (!did_indel && pc.len / 2 > done) stop = pc.len / 2;
                uint32_t run = 0;
                while (done + run < stop && !is_sub[rp + run]) ++run;
                if (run) {
                    push_edit((int32_t)run, (int32_t)run, nullptr, 0);
                    seq_out.append(readseq, rp, run);
                    qual_out.append(qual, rp, run);
                    done += run;
                    rp += run;
                    n_match += run;
                } else {
                    push_edit(1, 1, &readseq[rp], 1);
                    seq_out += readseq[rp];
                    qual_out += qual[rp];
                    done += 1;
                    rp += 1;
                }
            }
            a->edit_off.push_back((int64_t)a->e_from.size());
        }
        a->map_off.push_back((int64_t)a->m_node.size());
        a->seq += seq_out;
        a->seq_off.push_back((int64_t)a->seq.size());
        a->qual += qual_out;
        a->qual_off.push_back((int64_t)a->qual.size());
        char nm[32];
        const int ln = snprintf(nm, sizeof nm, "e%llu", (unsigned long long)r);
        a->name.append(nm, (size_t)ln);
        a->name_off.push_back((int64_t)a->name.size());
        a->mapq.push_back(q.uniform() < .15 ? (int32_t)q.below(60) : 60);
        a->identity.push_back(seq_out.empty() ? 0.0 : std::max(1e-3, (double)n_match / (double)seq_out.size()));
    }
    *gout = g;
    *dbout = db;
    *rout = a;
    return VGAN_OK;
}
