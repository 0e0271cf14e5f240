// Synthetic inputs of the published HaploCart shape: the real hcfiles graph is not in the
// reference tree (share/vgan/hcfiles/.keep) and there is no vg giraffe here, so benchmarks and parity tests
// run on a seeded mtDNA-like variation graph (16 569 bp, 11 821 nodes <= 8 bp, 5 179 haplogroup paths on a
// random tree) and on alignments sampled from its paths with sequencing errors, indels and softclips.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

using namespace vgan;

namespace {

const char BASES[4] = {'A', 'C', 'G', 'T'};

struct Variant {
    uint32_t branch; // tree node whose subtree carries the alt allele
    uint32_t node;   // alt node id
    uint32_t depth;
};

inline char comp(char c) {
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return 'N';
    }
}

} // namespace

extern "C" int vgan_synth_hc_graph(const vgan_synth_graph_cfg *cfg, vgan_graph **out) {
    if (!cfg || !out) return fail(VGAN_EINVAL, "vgan_synth_hc_graph: null argument");
    const uint32_t L = cfg->genome_len, N = cfg->n_nodes, P = cfg->n_paths;
    if (L < 16 || P < 1 || N < 2) return fail(VGAN_EINVAL, "vgan_synth_hc_graph: degenerate configuration");
    SplitMix64 rg(cfg->seed ^ 0x6763726170680001ull);
    std::string genome(L, 'A');
    for (auto &c : genome) c = BASES[rg.below(4)];
    // backbone segmentation: lengths {1:.35, 2:.30, 3:.15, 4:.08, 5..8:.12}; stretched (long nodes) when N is too small
    std::vector<uint32_t> site_start, site_len;
    {
        const double min_mean = (double)L / std::max<uint32_t>(1, N * 6 / 10);
        for (uint32_t pos = 0; pos < L;) {
            const double u = rg.uniform();
            uint32_t len = u < .35 ? 1 : u < .65 ? 2 : u < .80 ? 3 : u < .88 ? 4 : 5 + (uint32_t)rg.below(4);
            if (min_mean > 2.6) len = std::min<uint32_t>(std::max<uint32_t>(8, (uint32_t)(min_mean * 4)), (uint32_t)std::ceil(len * min_mean / 2.5));
            len = std::min(len, L - pos);
            site_start.push_back(pos);
            site_len.push_back(len);
            pos += len;
        }
    }
    const uint32_t n_sites = (uint32_t)site_start.size();
    if (n_sites >= N) return fail(VGAN_EINVAL, "vgan_synth_hc_graph: n_nodes=%u too small for genome_len=%u (%u backbone nodes)", N, L, n_sites);
    const uint32_t n_alt = N - n_sites;
    // random rooted tree over the P haplogroups (every tree node is a path; node 0 is the root)
    std::vector<uint32_t> parent(P, 0), depth(P, 0);
    std::vector<std::vector<uint32_t>> kids(P);
    for (uint32_t i = 1; i < P; ++i) {
        parent[i] = (uint32_t)rg.below(i);
        depth[i] = depth[parent[i]] + 1;
        kids[parent[i]].push_back(i);
    }
    std::vector<uint32_t> tin(P), tout(P);
    {
        uint32_t t = 0;
        std::vector<std::pair<uint32_t, size_t>> st{{0, 0}};
        tin[0] = t++;
        while (!st.empty()) {
            auto &top = st.back();
            if (top.second < kids[top.first].size()) {
                const uint32_t k = kids[top.first][top.second++];
                tin[k] = t++;
                st.push_back({k, 0});
            } else {
                tout[top.first] = t;
                st.pop_back();
            }
        }
    }
    // variants: site, branch, alt sequence
    std::vector<std::vector<Variant>> site_vars(n_sites);
    std::vector<std::string> alt_seq(n_alt);
    std::vector<uint32_t> alt_site(n_alt);
    for (uint32_t v = 0; v < n_alt; ++v) {
        const uint32_t s = (uint32_t)rg.below(n_sites);
        alt_site[v] = s;
        std::string ref = genome.substr(site_start[s], site_len[s]);
        if (rg.uniform() < .85) {
            const uint32_t k = (uint32_t)rg.below(ref.size());
            char c;
            do c = BASES[rg.below(4)];
            while (c == ref[k]);
            ref[k] = c;
            alt_seq[v] = ref;
        } else {
            std::string s2(1 + rg.below(8), 'A');
            for (auto &c : s2) c = BASES[rg.below(4)];
            if (s2 == ref) s2[0] = s2[0] == 'A' ? 'C' : 'A';
            alt_seq[v] = s2;
        }
    }
    // node ids follow the genome: backbone node of the site, then its alternative alleles
    std::vector<std::vector<uint32_t>> alts_of_site(n_sites);
    for (uint32_t v = 0; v < n_alt; ++v) alts_of_site[alt_site[v]].push_back(v);
    auto g = new vgan_graph();
    g->min_id = 1;
    g->max_id = N;
    g->node_seq_off.assign((size_t)N + 2, 0);
    g->pangenome_base.assign((size_t)N + 1, -1);
    std::vector<uint32_t> backbone_id(n_sites);
    {
        uint32_t id = 1;
        g->node_seq_off[0] = 0;
        g->node_seq_off[1] = 0;
        for (uint32_t s = 0; s < n_sites; ++s) {
            backbone_id[s] = id;
            g->node_seq.append(genome, site_start[s], site_len[s]);
            g->pangenome_base[id] = (int32_t)site_start[s] + 1;
            g->node_seq_off[++id] = (int64_t)g->node_seq.size();
            for (uint32_t v : alts_of_site[s]) {
                const uint32_t branch = P > 1 ? 1 + (uint32_t)rg.below(P - 1) : 0;
                site_vars[s].push_back({branch, id, depth[branch]});
                g->node_seq += alt_seq[v];
                g->pangenome_base[id] = (int32_t)site_start[s] + 1;
                g->node_seq_off[++id] = (int64_t)g->node_seq.size();
            }
        }
    }
    // path membership: path p takes the alt whose branch is the deepest ancestor-or-self of p
    g->n_paths = P;
    g->mask_words = (P + 63) / 64;
    g->mask.assign((size_t)(N + 1) * g->mask_words, 0);
    const bool keep_steps = (uint64_t)P * n_sites <= 4000000ull;
    if (keep_steps) g->path_steps.assign(P, {});
    for (uint32_t p = 0; p < P; ++p) {
        for (uint32_t s = 0; s < n_sites; ++s) {
            uint32_t node = backbone_id[s];
            int best = -1;
            for (const Variant &v : site_vars[s]) {
                if (tin[v.branch] <= tin[p] && tin[p] < tout[v.branch] && (int)v.depth > best) {
                    best = (int)v.depth;
                    node = v.node;
                }
            }
            g->mask[(size_t)node * g->mask_words + (p >> 6)] |= 1ull << (p & 63);
            if (keep_steps) g->path_steps[p].push_back({node, false});
        }
    }
    // mappability: 1.0 except 2 % of 50-bp windows uniform[0.3, 1]
    g->mappability.assign((size_t)L + 2, 1.0);
    for (uint32_t w = 0; w < L; w += 50) {
        if (rg.uniform() < .02) {
            const double v = .3 + .7 * rg.uniform();
            for (uint32_t i = w; i < std::min(L + 2, w + 50); ++i) g->mappability[i] = v;
        }
    }
    // names and tree sidecars: parents = ancestors nearest first; children = direct children
    char buf[32];
    std::vector<std::string> names(P);
    for (uint32_t p = 0; p < P; ++p) {
        snprintf(buf, sizeof buf, "hg%05u", p);
        names[p] = buf;
        g->path_names += names[p] + "\n";
    }
    for (uint32_t p = 0; p < P; ++p) {
        g->parents_txt += names[p];
        for (uint32_t a = p; a != 0;) {
            a = parent[a];
            g->parents_txt += " " + names[a];
        }
        g->parents_txt += "\n";
        g->children_txt += names[p];
        for (uint32_t k : kids[p]) g->children_txt += " " + names[k];
        g->children_txt += "\n";
    }
    *out = g;
    return VGAN_OK;
}

namespace {

struct Walk {
    uint32_t n_sites = 0;
    std::vector<uint32_t> site_node; // [P * n_sites] node id of path p at site s
};

// recover the per-path walks from the mask: nodes sharing a pangenome_base form a site
void build_walks(const vgan_graph &g, Walk &w, std::vector<uint32_t> &site_first_node) {
    site_first_node.clear();
    int32_t last = -2;
    for (int64_t id = g.min_id; id <= g.max_id; ++id) {
        if (g.pangenome_base[id] != last) {
            site_first_node.push_back((uint32_t)id);
            last = g.pangenome_base[id];
        }
    }
    w.n_sites = (uint32_t)site_first_node.size();
    site_first_node.push_back((uint32_t)g.max_id + 1);
    w.site_node.assign((size_t)g.n_paths * w.n_sites, 0);
    for (uint32_t s = 0; s < w.n_sites; ++s) {
        for (uint32_t id = site_first_node[s]; id < site_first_node[s + 1]; ++id) {
            const uint64_t *row = &g.mask[(size_t)id * g.mask_words];
            for (uint32_t wd = 0; wd < g.mask_words; ++wd) {
                uint64_t bits = row[wd];
                while (bits) {
                    const uint32_t p = wd * 64 + (uint32_t)__builtin_ctzll(bits);
                    bits &= bits - 1;
                    if (p < g.n_paths) w.site_node[(size_t)p * w.n_sites + s] = id;
                }
            }
        }
    }
}

struct Piece {
    uint32_t node, off, len;
};

void gen_reads(const vgan_graph &g, const Walk &w, const vgan_synth_reads_cfg &cfg, uint64_t r0, uint64_t r1,
               vgan_alnset &a) {
    const uint32_t L = cfg.read_len;
    std::vector<Piece> pieces;
    std::string fwd, qual, readseq;
    std::vector<uint8_t> is_err;
    char nm[32];
    for (uint64_t r = r0; r < r1; ++r) {
        SplitMix64 rg(cfg.seed * 0x9E3779B97F4A7C15ull + r * 0xD1B54A32D192ED03ull + 0x7265616473ull);
        const uint32_t p = (uint32_t)rg.below(g.n_paths);
        const uint32_t *walk = &w.site_node[(size_t)p * w.n_sites];
        // sample a start: site and offset; collect L bases (or fewer at the end of the linearised genome)
        pieces.clear();
        fwd.clear();
        uint32_t s = (uint32_t)rg.below(w.n_sites);
        uint32_t off = (uint32_t)rg.below((uint64_t)std::max<int64_t>(1, g.seq_len(walk[s])));
        while (fwd.size() < L && s < w.n_sites) {
            const uint32_t node = walk[s];
            const uint32_t nl = (uint32_t)g.seq_len(node);
            if (off < nl) {
                const uint32_t take = std::min<uint32_t>(nl - off, L - (uint32_t)fwd.size());
                pieces.push_back({node, off, take});
                fwd.append(g.seq_ptr(node) + off, take);
            }
            off = 0;
            ++s;
        }
        const uint32_t n = (uint32_t)fwd.size();
        if (n == 0) { // cannot happen for genome_len >= 16, keep the set dense anyway
            pieces.push_back({walk[0], 0, 1});
            fwd.assign(g.seq_ptr(walk[0]), 1);
        }
        const bool rev = rg.uniform() < .5;
        // qualities in read orientation
        qual.resize(fwd.size());
        for (auto &q : qual) {
            const double u = rg.uniform();
            q = (char)(u < .01 ? 2 : u < .81 ? 37 : 3 + (int)rg.below(34));
        }
        // oriented pieces + read sequence
        readseq = fwd;
        if (rev) {
            std::reverse(readseq.begin(), readseq.end());
            for (auto &c : readseq) c = comp(c);
            std::reverse(pieces.begin(), pieces.end());
            for (auto &pc : pieces) pc.off = (uint32_t)g.seq_len(pc.node) - (pc.off + pc.len);
        }
        // sequencing errors (substitutions) in read orientation
        is_err.assign(readseq.size(), 0);
        uint32_t n_match = (uint32_t)readseq.size();
        if (cfg.errors) {
            for (size_t i = 0; i < readseq.size(); ++i) {
                const double pe = std::pow(10.0, -0.1 * (double)qual[i]);
                if (rg.uniform() < pe) {
                    char c;
                    do c = BASES[rg.below(4)];
                    while (c == readseq[i]);
                    readseq[i] = c;
                    is_err[i] = 1;
                    --n_match;
                }
            }
        }
        // optional events
        const bool softclip = rg.uniform() < cfg.softclip_rate;
        const bool clip_tail = softclip && rg.uniform() < .3;
        const uint32_t clip_len = softclip ? 5 + (uint32_t)rg.below(16) : 0;
        const bool indel = rg.uniform() < cfg.indel_rate;
        const bool is_ins = indel && rg.uniform() < .5;
        const uint32_t indel_len = indel ? 1 + (uint32_t)rg.below(3) : 0;
        const uint32_t indel_piece = indel ? (uint32_t)rg.below(pieces.size()) : 0;
        // emit
        std::string seq_out, qual_out;
        size_t rp = 0; // position in readseq
        for (size_t k = 0; k < pieces.size(); ++k) {
            const Piece &pc = pieces[k];
            a.m_node.push_back(pc.node);
            a.m_offset.push_back(pc.off);
            a.m_rev.push_back(rev);
            auto push_edit = [&](int32_t from, int32_t to, const char *s, size_t sl) {
                a.e_from.push_back(from);
                a.e_to.push_back(to);
                if (sl) a.e_seq.append(s, sl);
                a.e_seq_off.push_back((int64_t)a.e_seq.size());
            };
            if (k == 0 && softclip && !clip_tail) { // leading softclip: insertion edit in front of the first mapping
                std::string clip(clip_len, 'A');
                for (auto &c : clip) c = BASES[rg.below(4)];
                push_edit(0, (int32_t)clip_len, clip.data(), clip.size());
                seq_out += clip;
                qual_out.append(clip_len, (char)20);
            }
            uint32_t done = 0;
            uint32_t del_at = 0xFFFFFFFFu, ins_at = 0xFFFFFFFFu;
            if (indel && k == indel_piece) {
                if (is_ins) ins_at = (uint32_t)rg.below(pc.len + 1);
                else if (pc.len > indel_len) del_at = (uint32_t)rg.below(pc.len - indel_len);
            }
            while (done < pc.len || ins_at == done) {
                if (ins_at == done) {
                    std::string ins(indel_len, 'A');
                    for (auto &c : ins) c = BASES[rg.below(4)];
                    push_edit(0, (int32_t)indel_len, ins.data(), ins.size());
                    seq_out += ins;
                    qual_out.append(indel_len, (char)30);
                    ins_at = 0xFFFFFFFFu;
                    continue;
                }
                if (del_at == done) { // read lacks indel_len graph bases
                    push_edit((int32_t)indel_len, 0, nullptr, 0);
                    done += indel_len;
                    rp += indel_len;
                    del_at = 0xFFFFFFFFu;
                    continue;
                }
                uint32_t stop = pc.len;
                if (del_at != 0xFFFFFFFFu && del_at > done) stop = std::min(stop, del_at);
                if (ins_at != 0xFFFFFFFFu && ins_at > done) stop = std::min(stop, ins_at);
                // run of matches up to the next substitution
                uint32_t run = 0;
                while (done + run < stop && !is_err[rp + run]) ++run;
                if (run) {
                    push_edit((int32_t)run, (int32_t)run, nullptr, 0);
                    seq_out.append(readseq, rp, run);
                    qual_out.append(qual, rp, run);
                    done += run;
                    rp += run;
                } else {
                    push_edit(1, 1, &readseq[rp], 1);
                    seq_out += readseq[rp];
                    qual_out += qual[rp];
                    done += 1;
                    rp += 1;
                }
            }
            if (k + 1 == pieces.size() && softclip && clip_tail) {
                std::string clip(clip_len, 'A');
                for (auto &c : clip) c = BASES[rg.below(4)];
                push_edit(0, (int32_t)clip_len, clip.data(), clip.size());
                seq_out += clip;
                qual_out.append(clip_len, (char)20);
            }
            a.edit_off.push_back((int64_t)a.e_from.size());
        }
        a.map_off.push_back((int64_t)a.m_node.size());
        a.seq += seq_out;
        a.seq_off.push_back((int64_t)a.seq.size());
        a.qual += qual_out;
        a.qual_off.push_back((int64_t)a.qual.size());
        const int len = snprintf(nm, sizeof nm, "r%llu", (unsigned long long)r);
        a.name.append(nm, (size_t)len);
        a.name_off.push_back((int64_t)a.name.size());
        a.mapq.push_back(rg.uniform() < cfg.low_mapq_rate ? (int32_t)rg.below(60) : 60);
        a.identity.push_back(seq_out.empty() ? 0.0 : (double)n_match / (double)seq_out.size());
    }
}

template <class T> void cat(std::vector<T> &d, const std::vector<T> &s, size_t skip, T shift) {
    for (size_t i = skip; i < s.size(); ++i) d.push_back(s[i] + shift);
}

} // namespace

extern "C" int vgan_synth_hc_reads(const vgan_graph *g, const vgan_synth_reads_cfg *cfg, vgan_alnset **out) {
    if (!g || !cfg || !out) return fail(VGAN_EINVAL, "vgan_synth_hc_reads: null argument");
    if (cfg->read_len < 1 || cfg->read_len > 20000) return fail(VGAN_EINVAL, "vgan_synth_hc_reads: read_len must be 1..20000");
    Walk w;
    std::vector<uint32_t> first;
    build_walks(*g, w, first);
    if (w.n_sites == 0) return fail(VGAN_EINVAL, "vgan_synth_hc_reads: graph has no sites");
    for (uint32_t p = 0; p < g->n_paths; ++p)
        for (uint32_t s = 0; s < w.n_sites; ++s)
            if (w.site_node[(size_t)p * w.n_sites + s] == 0)
                return fail(VGAN_EINVAL, "vgan_synth_hc_reads: path %u has no node at site %u (not a synthetic hc graph?)", p, s);
    const uint64_t R = cfg->n_reads;
    int nt = (int)usable_cpus();
    nt = (int)std::max<uint64_t>(1, std::min<uint64_t>(nt, (R + 8191) / 8192));
    std::vector<vgan_alnset> parts((size_t)nt);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
        const uint64_t b0 = cfg->first_read + R * t / nt, b1 = cfg->first_read + R * (t + 1) / nt;
        if (nt == 1) gen_reads(*g, w, *cfg, b0, b1, parts[t]);
        else th.emplace_back(gen_reads, std::cref(*g), std::cref(w), std::cref(*cfg), b0, b1, std::ref(parts[t]));
    }
    for (auto &t : th) t.join();
    auto a = new vgan_alnset();
    merge_alnsets(parts, *a);
    *out = a;
    return VGAN_OK;
}
