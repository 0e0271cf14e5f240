// Graph + hcfiles sidecars: the host-side replacement of readPathHandleGraph() and the load_*() readers
// (reference src/readPathHandleGraph.cpp:14-37, src/load.cpp:6-58,283-345).  The ODGI .og binary cannot
// be parsed without libbdsg, so the node sequences and paths come from GFA S/P lines.
#include "common.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <thread>

using namespace vgan;

void vgan_graph::fill_view(vgan_graph_view *v) const {
    v->min_id = min_id;
    v->max_id = max_id;
    v->node_seq_off = node_seq_off.data();
    v->node_seq = node_seq.data();
    v->n_paths = n_paths;
    v->mask_words = mask_words;
    v->mask = mask.data();
    v->pangenome_base = pangenome_base.data();
    v->mappability = mappability.data();
    v->n_mappability = mappability.size();
    v->path_names = path_names.c_str();
    v->parents_txt = parents_txt.c_str();
    v->children_txt = children_txt.c_str();
}

namespace {

struct LineIter {
    const std::string &s;
    size_t pos = 0;
    explicit LineIter(const std::string &str) : s(str) {}
    bool next(const char *&b, const char *&e) {
        if (pos >= s.size()) return false;
        size_t nl = s.find('\n', pos);
        if (nl == std::string::npos) nl = s.size();
        b = s.data() + pos;
        e = s.data() + nl;
        if (e > b && e[-1] == '\r') --e;
        pos = nl + 1;
        return true;
    }
};

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

void tokens_ws(const char *b, const char *e, std::vector<std::pair<const char *, const char *>> &out) {
    out.clear();
    while (b < e) {
        while (b < e && is_ws(*b)) ++b;
        if (b >= e) break;
        const char *t = b;
        while (b < e && !is_ws(*b)) ++b;
        out.emplace_back(t, b);
    }
}

bool parse_i64(const char *b, const char *e, int64_t &v) {
    if (b >= e) return false;
    bool neg = false;
    if (*b == '-' || *b == '+') {
        neg = *b == '-';
        ++b;
    }
    if (b >= e || *b < '0' || *b > '9') return false;
    int64_t x = 0;
    while (b < e && *b >= '0' && *b <= '9') x = x * 10 + (*b++ - '0'); // stoi semantics: stop at first non-digit
    v = neg ? -x : x;
    return true;
}

int load_gfa(const std::string &txt, vgan_graph &g) {
    std::map<int64_t, std::string> seqs;
    std::vector<std::string> names;
    LineIter it(txt);
    const char *b, *e;
    while (it.next(b, e)) {
        if (b >= e) continue;
        if (*b == 'S') {
            const char *t1 = (const char *)memchr(b, '\t', e - b);
            if (!t1) return fail(VGAN_EIO, "GFA: malformed S line");
            const char *t2 = (const char *)memchr(t1 + 1, '\t', e - t1 - 1);
            if (!t2) return fail(VGAN_EIO, "GFA: malformed S line");
            const char *t3 = (const char *)memchr(t2 + 1, '\t', e - t2 - 1);
            if (!t3) t3 = e;
            int64_t id;
            if (!parse_i64(t1 + 1, t2, id) || id < 0) return fail(VGAN_EIO, "GFA: non-numeric segment name");
            if (id >= ((int64_t)1 << 28)) return fail(VGAN_ERANGE, "GFA: segment id %lld is beyond the id-indexed graph layout (2^28)", (long long)id);
            seqs[id].assign(t2 + 1, t3);
        } else if (*b == 'P') {
            const char *t1 = (const char *)memchr(b, '\t', e - b);
            if (!t1) return fail(VGAN_EIO, "GFA: malformed P line");
            const char *t2 = (const char *)memchr(t1 + 1, '\t', e - t1 - 1);
            if (!t2) return fail(VGAN_EIO, "GFA: malformed P line");
            const char *t3 = (const char *)memchr(t2 + 1, '\t', e - t2 - 1);
            if (!t3) t3 = e;
            names.emplace_back(t1 + 1, t2);
            std::vector<std::pair<int64_t, bool>> steps;
            const char *p = t2 + 1;
            while (p < t3) {
                const char *c = (const char *)memchr(p, ',', t3 - p);
                if (!c) c = t3;
                if (c - p >= 2) {
                    int64_t id;
                    if (!parse_i64(p, c - 1, id)) return fail(VGAN_EIO, "GFA: bad path step");
                    steps.emplace_back(id, c[-1] == '-');
                }
                p = c + 1;
            }
            g.path_steps.push_back(std::move(steps));
        }
    }
    if (seqs.empty()) return fail(VGAN_EIO, "GFA: no S lines");
    g.min_id = seqs.begin()->first;
    g.max_id = seqs.rbegin()->first;
    g.node_seq_off.assign((size_t)g.max_id + 2, 0);
    g.node_seq.clear();
    for (int64_t id = 0; id <= g.max_id; ++id) {
        g.node_seq_off[id] = (int64_t)g.node_seq.size();
        auto f = seqs.find(id);
        if (f != seqs.end()) g.node_seq += f->second;
    }
    g.node_seq_off[g.max_id + 1] = (int64_t)g.node_seq.size();
    g.path_names.clear();
    for (auto &n : names) g.path_names += n + "\n";
    g.n_paths = (uint32_t)names.size();
    return VGAN_OK;
}

// ODGI binary graph (.og), as `odgi build` / vg write it and bdsg::ODGI::deserialize reads it (readPathHandleGraph.cpp:14-37
// opens <dbprefix>.og this way).  odgi's sources are not part of the reference tree, so the layout below is the one observed
// on the reference's own fixture test/reconstructInputSeq/target_graph.og (every node sequence, path name and path
// membership checked against the GFA of the same graph, tests/test_host_frontend.py); anything that deviates from it is
// rejected, not guessed at:
//   u32 magic ab ad 79 34 | u64 max_rank, min_rank, node_count, edge_count, path_count, path_handle_next, deleted_count,
//   id_increment (node id = rank + id_increment), one more u64 |
//   per node in rank order: u32 seq_len, u32 edge_bytes, u32 edge_count, u64 blob_len (= seq_len + edge_bytes), the blob
//     (sequence, then 2 bytes per edge end), then the packed path steps of the node: u64 n_words, the words, u64 max value,
//     u64 n_values, u8 bit width, u8 values per word -- five values per step, the first one 2 * (path handle - 1) + strand |
//   deleted-node bitmap and per-path metadata (skipped) | at the very end the name table: u64 path_count, then per path
//   u64 name_len, the name, u64 path handle (1-based).
// Kept: node ids and sequences, path names in handle order, which paths visit each node (not the order of the steps).
constexpr uint32_t ODGI_MAGIC = 0x3479adabu;

bool is_odgi(const std::string &bytes) {
    uint32_t m = 0;
    if (bytes.size() >= 4) memcpy(&m, bytes.data(), 4);
    return m == ODGI_MAGIC;
}

int load_odgi(const std::string &bytes, vgan_graph &g) {
    const unsigned char *d = (const unsigned char *)bytes.data();
    const size_t n = bytes.size();
    size_t p = 4;
    auto u64_at = [&](size_t at, uint64_t &v) {
        if (at + 8 > n) return false;
        memcpy(&v, d + at, 8);
        return true;
    };
    uint64_t h[9];
    for (int i = 0; i < 9; ++i, p += 8)
        if (!u64_at(p, h[i])) return fail(VGAN_EIO, "ODGI: truncated header");
    const uint64_t max_rank = h[0], min_rank = h[1], node_count = h[2], path_count = h[4], deleted = h[6], incr = h[7];
    if (node_count == 0 || node_count > (1ull << 31) || path_count > (1ull << 24) || incr > (1ull << 40) || deleted != 0 || min_rank != 0 ||
        max_rank + 1 != node_count)
        return fail(VGAN_EIO, "ODGI: unsupported header (nodes %llu, paths %llu, deleted %llu)", (unsigned long long)node_count,
                    (unsigned long long)path_count, (unsigned long long)deleted);
    // nothing is sized by a header field before the records behind it have been seen: a record takes at least 46 bytes
    if (node_count > (n - p) / 46 || path_count > (n - p) / 17 || incr + node_count > (1ull << 28))
        return fail(VGAN_EIO, "ODGI: %llu nodes from id %llu do not fit this file / the id-indexed graph layout", (unsigned long long)node_count,
                    (unsigned long long)incr);
    std::string all_seq;
    std::vector<int64_t> rank_off((size_t)node_count + 1, 0);
    std::vector<std::vector<std::pair<int64_t, bool>>> visits((size_t)path_count);
    for (uint64_t r = 0; r < node_count; ++r) {
        const int64_t id = (int64_t)(incr + r);
        if (p + 20 > n) return fail(VGAN_EIO, "ODGI: truncated node record %llu", (unsigned long long)r);
        uint32_t seq_len, edge_bytes, edge_count;
        uint64_t blob_len;
        memcpy(&seq_len, d + p, 4);
        memcpy(&edge_bytes, d + p + 4, 4);
        memcpy(&edge_count, d + p + 8, 4);
        memcpy(&blob_len, d + p + 12, 8);
        p += 20;
        if ((uint64_t)seq_len + edge_bytes != blob_len || edge_bytes != 2 * (uint64_t)edge_count || blob_len > n - p)
            return fail(VGAN_EIO, "ODGI: node record %llu does not have the known layout", (unsigned long long)r);
        for (uint32_t i = 0; i < seq_len; ++i) {
            const unsigned char c = d[p + i];
            if (c < 'A' || c > 'z' || (c > 'Z' && c < 'a')) return fail(VGAN_EIO, "ODGI: node %lld holds a non-letter", (long long)id);
        }
        rank_off[(size_t)r] = (int64_t)all_seq.size();
        all_seq.append((const char *)d + p, seq_len);
        p += blob_len;
        uint64_t n_words, max_value, n_values;
        if (!u64_at(p, n_words) || n_words > (n - p) / 8) return fail(VGAN_EIO, "ODGI: truncated path steps of node %lld", (long long)id);
        const size_t words_at = p + 8;
        p = words_at + 8 * (size_t)n_words;
        if (!u64_at(p, max_value) || !u64_at(p + 8, n_values) || p + 18 > n) return fail(VGAN_EIO, "ODGI: truncated path steps of node %lld", (long long)id);
        const unsigned width = d[p + 16], per_word = d[p + 17];
        p += 18;
        if (width == 0 || width > 64 || per_word != 64 / width || (width < 64 && max_value != (1ull << width) - 1) || n_values % 5 != 0 ||
            n_values > n_words * per_word)
            return fail(VGAN_EIO, "ODGI: path steps of node %lld are not a packed vector of 5-tuples", (long long)id);
        for (uint64_t s5 = 0; s5 < n_values; s5 += 5) {
            uint64_t w;
            memcpy(&w, d + words_at + 8 * (size_t)(s5 / per_word), 8);
            const uint64_t v = (w >> ((s5 % per_word) * width)) & max_value;
            const uint64_t handle = v / 2; // 0-based
            if (handle >= path_count) return fail(VGAN_EIO, "ODGI: node %lld is on path %llu of %llu", (long long)id, (unsigned long long)handle + 1, (unsigned long long)path_count);
            auto &vs = visits[(size_t)handle];
            if (vs.empty() || vs.back().first != id) vs.emplace_back(id, (v & 1) != 0);
        }
    }
    rank_off[(size_t)node_count] = (int64_t)all_seq.size();
    // the name table closes the file: find the offset from which path_count well-formed entries run exactly to the end
    std::vector<std::string> names((size_t)path_count);
    bool found = path_count == 0;
    for (size_t q = n >= 8 ? n - 8 : 0; !found && q >= p && q + 8 <= n; --q) {
        uint64_t cnt;
        u64_at(q, cnt);
        if (cnt == path_count) {
            size_t x = q + 8;
            std::vector<std::string> got((size_t)path_count);
            std::vector<char> seen((size_t)path_count, 0);
            bool ok = true;
            for (uint64_t i = 0; i < path_count && ok; ++i) {
                uint64_t len, handle;
                ok = u64_at(x, len) && len > 0 && len < 4096 && x + 8 + len + 8 <= n;
                if (!ok) break;
                memcpy(&handle, d + x + 8 + len, 8);
                ok = handle >= 1 && handle <= path_count && !seen[(size_t)handle - 1];
                if (!ok) break;
                for (uint64_t k = 0; k < len && ok; ++k) ok = d[x + 8 + k] >= 0x20 && d[x + 8 + k] < 0x7f;
                seen[(size_t)handle - 1] = 1;
                got[(size_t)handle - 1].assign((const char *)d + x + 8, (size_t)len);
                x += 16 + (size_t)len;
            }
            if (ok && x == n) {
                names.swap(got);
                found = true;
            }
        }
        if (q == 0) break;
    }
    if (!found) return fail(VGAN_EIO, "ODGI: no path name table at the end of the file");
    g.min_id = (int64_t)incr;
    g.max_id = (int64_t)(incr + node_count - 1);
    g.node_seq_off.assign((size_t)g.max_id + 2, 0);
    for (uint64_t r = 0; r <= node_count; ++r) g.node_seq_off[(size_t)(incr + r)] = rank_off[(size_t)r];
    g.node_seq.swap(all_seq);
    g.path_names.clear();
    for (auto &nm : names) g.path_names += nm + "\n";
    g.n_paths = (uint32_t)path_count;
    g.path_steps = std::move(visits); // membership in node-id order: enough for the mask, not the walk order of the path
    return VGAN_OK;
}

void mask_from_steps(vgan_graph &g) {
    g.mask_words = (g.n_paths + 63) / 64;
    g.mask.assign((size_t)(g.max_id + 1) * g.mask_words, 0);
    for (size_t p = 0; p < g.path_steps.size(); ++p)
        for (auto &st : g.path_steps[p])
            if (st.first >= 0 && st.first <= g.max_id) g.mask[(size_t)st.first * g.mask_words + p / 64] |= 1ull << (p % 64);
}

} // namespace

static int graph_load_impl(const char *gfa_path, const char *hcfiles_dir, vgan_graph **out) {
    if (!gfa_path || !out) return fail(VGAN_EINVAL, "vgan_graph_load: null argument");
    PhaseTimer pt("graph_load");
    std::string txt;
    if (!read_file(gfa_path, txt)) return fail(VGAN_EIO, "cannot read %s", gfa_path);
    pt.lap("read gfa");
    std::unique_ptr<vgan_graph> holder(new vgan_graph()); // freed on every early return and on exceptions
    vgan_graph *g = holder.get();
    int rc = is_odgi(txt) ? load_odgi(txt, *g) : load_gfa(txt, *g);
    pt.lap("parse gfa");
    if (rc) {
        return rc;
    }
    std::string dir = hcfiles_dir ? hcfiles_dir : "";
    if (!dir.empty() && dir.back() != '/') dir += '/';
    std::string side;
    const std::string names_in_graph = g->path_names; // P-line / ODGI handle order: the order of path_steps
    // graph_paths (load.cpp:43-58): first whitespace token of each line
    if (!dir.empty() && read_text_maybe_gz(dir + "graph_paths", side)) {
        g->path_names.clear();
        uint32_t n = 0;
        LineIter it(side);
        const char *b, *e;
        std::vector<std::pair<const char *, const char *>> tk;
        while (it.next(b, e)) {
            tokens_ws(b, e, tk);
            if (tk.empty()) continue;
            g->path_names.append(tk[0].first, tk[0].second);
            g->path_names += '\n';
            ++n;
        }
        if (g->n_paths && n != g->n_paths && !g->path_steps.empty()) {
            // sidecar wins (it is what indexes path_supports); GFA P lines are then only used for writing
            g->path_steps.clear();
        }
        g->n_paths = n;
    }
    pt.lap("graph_paths");
    // path_supports (load.cpp:283-300): row = line index = node id, first P characters
    ByteBuf ps_bytes;
    if (!dir.empty() && read_bytes_maybe_gz(dir + "path_supports", ps_bytes)) {
        pt.lap("read path_supports");
        g->mask_words = (g->n_paths + 63) / 64;
        g->mask.assign((size_t)(g->max_id + 1) * g->mask_words, 0);
        // 60 MB of '0'/'1' text for the hcfiles shape: the rows are parsed by several threads, each starting at a line
        // boundary and knowing its first row from a count of the newlines before it
        const char *base = ps_bytes.data();
        const size_t nbytes = ps_bytes.size();
        const unsigned nth = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)usable_cpus(), nbytes / (1u << 20) + 1}));
        std::vector<size_t> cut(nth + 1, nbytes);
        cut[0] = 0;
        for (unsigned t = 1; t < nth; ++t) {
            size_t pos = nbytes * t / nth;
            const void *nl = pos < nbytes ? memchr(base + pos, '\n', nbytes - pos) : nullptr;
            cut[t] = nl ? (size_t)((const char *)nl - base) + 1 : nbytes;
        }
        std::vector<int64_t> rows_in(nth, 0);
        auto count = [&](unsigned t) {
            int64_t c = 0;
            for (const char *q = base + cut[t], *e = base + cut[t + 1]; q < e;) {
                const void *nl = memchr(q, '\n', (size_t)(e - q));
                if (!nl) break;
                ++c;
                q = (const char *)nl + 1;
            }
            rows_in[t] = c;
        };
        auto parse = [&](unsigned t, int64_t row) {
            const char *q = base + cut[t], *e = base + cut[t + 1];
            while (q < e) {
                const char *nl = (const char *)memchr(q, '\n', (size_t)(e - q));
                const char *le = nl ? nl : e;
                if (row <= g->max_id) {
                    int64_t len = le - q;
                    if (len > 0 && q[len - 1] == '\r') --len;
                    const int64_t n = std::min<int64_t>(len, g->n_paths);
                    uint64_t *w = &g->mask[(size_t)row * g->mask_words];
                    for (int64_t j = 0; j < n; ++j)
                        if (q[j] == '1') w[j >> 6] |= 1ull << (j & 63);
                }
                ++row;
                q = le + 1;
            }
        };
        auto run = [&](auto &&fn) {
            if (nth == 1) {
                fn(0u);
                return;
            }
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nth; ++t) th.emplace_back(fn, t);
            for (auto &x : th) x.join();
        };
        run(count);
        std::vector<int64_t> first_row(nth, 0);
        for (unsigned t = 1; t < nth; ++t) first_row[t] = first_row[t - 1] + rows_in[t - 1];
        run([&](unsigned t) { parse(t, first_row[t]); });
    } else {
        // mask from the graph's own paths.  When graph_paths renamed / reordered the paths (an ODGI file lists its paths in
        // handle order, not in the order of the sidecar), the steps follow the sidecar's order by name
        if (g->path_names != names_in_graph && !g->path_steps.empty()) {
            std::map<std::string, size_t> col;
            {
                LineIter it(names_in_graph);
                const char *b, *e;
                size_t i = 0;
                while (it.next(b, e)) col.emplace(std::string(b, e), i++);
            }
            std::vector<std::vector<std::pair<int64_t, bool>>> ordered;
            LineIter it(g->path_names);
            const char *b, *e;
            bool all = true;
            while (it.next(b, e)) {
                const auto f = col.find(std::string(b, e));
                if (f == col.end() || f->second >= g->path_steps.size()) {
                    all = false;
                    break;
                }
                ordered.push_back(g->path_steps[f->second]);
            }
            if (all && ordered.size() == g->n_paths) g->path_steps.swap(ordered); // otherwise: positional, as before
        }
        mask_from_steps(*g);
    }
    pt.lap("path_supports rows");
    // parsed_pangenome_mapping (load.cpp:27-41): value = stoi(tok[1]) + 1
    g->pangenome_base.assign((size_t)g->max_id + 1, -1);
    if (!dir.empty() && read_text_maybe_gz(dir + "parsed_pangenome_mapping", side)) {
        LineIter it(side);
        const char *b, *e;
        std::vector<std::pair<const char *, const char *>> tk;
        while (it.next(b, e)) {
            tokens_ws(b, e, tk);
            if (tk.size() < 2) continue;
            int64_t id, v;
            if (!parse_i64(tk[0].first, tk[0].second, id) || !parse_i64(tk[1].first, tk[1].second, v)) continue;
            // the reference keys the map by the token string; "007" != "7": keep only canonical decimal keys
            if (id >= 0 && id <= g->max_id && std::to_string(id).size() == (size_t)(tk[0].second - tk[0].first))
                g->pangenome_base[id] = (int32_t)(v + 1);
        }
    } else {
        int64_t pos = 0;
        for (int64_t id = g->min_id; id <= g->max_id; ++id) {
            if (g->seq_len(id) > 0) g->pangenome_base[id] = (int32_t)(pos + 1);
            pos += g->seq_len(id);
        }
    }
    // mappability.tsv (load.cpp:6-24): append value for i in [start,end)
    g->mappability.clear();
    if (!dir.empty() && read_text_maybe_gz(dir + "mappability.tsv", side)) {
        LineIter it(side);
        const char *b, *e;
        std::vector<std::pair<const char *, const char *>> tk;
        while (it.next(b, e)) {
            tokens_ws(b, e, tk);
            if (tk.size() < 4) continue;
            int64_t s0, s1;
            if (!parse_i64(tk[1].first, tk[1].second, s0) || !parse_i64(tk[2].first, tk[2].second, s1)) continue;
            const double v = strtod(std::string(tk[3].first, tk[3].second).c_str(), nullptr);
            if (s1 - s0 > (int64_t)1 << 28 || g->mappability.size() > ((size_t)1 << 30)) { // a corrupt range, not a genome
                        return fail(VGAN_ERANGE, "mappability.tsv: interval [%lld, %lld) is out of any plausible range", (long long)s0, (long long)s1);
            }
            for (int64_t i = s0; i < s1; ++i) g->mappability.push_back(v);
        }
    } else {
        int32_t mx = 0;
        for (int32_t v : g->pangenome_base) mx = std::max(mx, v);
        g->mappability.assign((size_t)mx + 16, 1.0);
    }
    if (!dir.empty()) {
        read_text_maybe_gz(dir + "parents.txt", g->parents_txt);
        read_text_maybe_gz(dir + "children.txt", g->children_txt);
    }
    pt.lap("other sidecars");
    *out = holder.release();
    return VGAN_OK;
}

extern "C" int vgan_graph_load(const char *gfa_path, const char *hcfiles_dir, vgan_graph **out) {
    try {
        return graph_load_impl(gfa_path, hcfiles_dir, out);
    } catch (const std::bad_alloc &) {
        return fail(VGAN_ENOMEM, "vgan_graph_load: out of memory reading %s", gfa_path ? gfa_path : "(null)");
    } catch (const std::exception &e) {
        return fail(VGAN_EIO, "vgan_graph_load: %s", e.what());
    }
}

extern "C" int vgan_graph_from_arrays(const vgan_graph_view *v, vgan_graph **out) {
    if (!v || !out || v->max_id < 0 || !v->node_seq_off) return fail(VGAN_EINVAL, "vgan_graph_from_arrays: bad view");
    auto g = new vgan_graph();
    g->min_id = v->min_id;
    g->max_id = v->max_id;
    g->node_seq_off.assign(v->node_seq_off, v->node_seq_off + v->max_id + 2);
    g->node_seq.assign(v->node_seq, (size_t)v->node_seq_off[v->max_id + 1]);
    g->n_paths = v->n_paths;
    g->mask_words = (v->n_paths + 63) / 64;
    g->mask.assign(v->mask, v->mask + (size_t)(v->max_id + 1) * g->mask_words);
    g->pangenome_base.assign(v->pangenome_base, v->pangenome_base + v->max_id + 1);
    g->mappability.assign(v->mappability, v->mappability + v->n_mappability);
    g->path_names = v->path_names ? v->path_names : "";
    g->parents_txt = v->parents_txt ? v->parents_txt : "";
    g->children_txt = v->children_txt ? v->children_txt : "";
    *out = g;
    return VGAN_OK;
}

extern "C" int vgan_graph_view_get(const vgan_graph *g, vgan_graph_view *out) {
    if (!g || !out) return fail(VGAN_EINVAL, "vgan_graph_view_get: null argument");
    g->fill_view(out);
    return VGAN_OK;
}

extern "C" int vgan_graph_write(const vgan_graph *g, const char *dir_c) {
    if (!g || !dir_c) return fail(VGAN_EINVAL, "vgan_graph_write: null argument");
    std::string dir = dir_c;
    if (!dir.empty() && dir.back() != '/') dir += '/';
    std::string gfa = "H\tVN:Z:1.0\n";
    for (int64_t id = g->min_id; id <= g->max_id; ++id) {
        if (g->seq_len(id) == 0) continue;
        gfa += "S\t" + std::to_string(id) + "\t";
        gfa.append(g->seq_ptr(id), (size_t)g->seq_len(id));
        gfa += "\n";
    }
    {
        std::istringstream names(g->path_names);
        std::string nm;
        std::set<std::pair<std::pair<int64_t, bool>, std::pair<int64_t, bool>>> links;
        for (size_t p = 0; p < g->path_steps.size(); ++p) {
            std::getline(names, nm);
            gfa += "P\t" + nm + "\t";
            const auto &st = g->path_steps[p];
            for (size_t i = 0; i < st.size(); ++i) {
                if (i) gfa += ",";
                gfa += std::to_string(st[i].first) + (st[i].second ? "-" : "+");
                if (i) links.insert({st[i - 1], st[i]});
            }
            gfa += "\t*\n";
        }
        for (auto &l : links) {
            gfa += "L\t" + std::to_string(l.first.first) + "\t" + (l.first.second ? "-" : "+") + "\t" +
                   std::to_string(l.second.first) + "\t" + (l.second.second ? "-" : "+") + "\t*\n";
        }
    }
    if (!write_file(dir + "graph.gfa", gfa)) return fail(VGAN_EIO, "cannot write %sgraph.gfa", dir.c_str());
    std::string ps;
    ps.reserve((size_t)(g->max_id + 1) * (g->n_paths + 1));
    for (int64_t id = 0; id <= g->max_id; ++id) {
        const uint64_t *w = &g->mask[(size_t)id * g->mask_words];
        for (uint32_t p = 0; p < g->n_paths; ++p) ps += ((w[p >> 6] >> (p & 63)) & 1) ? '1' : '0';
        ps += '\n';
    }
    std::string gz;
    if (!gzip_bytes(ps, gz) || !write_file(dir + "path_supports.gz", gz)) return fail(VGAN_EIO, "cannot write path_supports");
    std::string pm;
    for (int64_t id = 0; id <= g->max_id; ++id)
        if (g->pangenome_base[id] >= 0) pm += std::to_string(id) + "\t" + std::to_string(g->pangenome_base[id] - 1) + "\n";
    if (!write_file(dir + "parsed_pangenome_mapping", pm)) return fail(VGAN_EIO, "cannot write parsed_pangenome_mapping");
    std::string mp;
    char buf[128];
    for (size_t i = 0; i < g->mappability.size();) {
        size_t j = i + 1;
        while (j < g->mappability.size() && g->mappability[j] == g->mappability[i]) ++j;
        snprintf(buf, sizeof buf, "chrM\t%zu\t%zu\t%.17g\n", i, j, g->mappability[i]);
        mp += buf;
        i = j;
    }
    if (!write_file(dir + "mappability.tsv", mp)) return fail(VGAN_EIO, "cannot write mappability.tsv");
    if (!write_file(dir + "graph_paths", g->path_names)) return fail(VGAN_EIO, "cannot write graph_paths");
    if (!write_file(dir + "parents.txt", g->parents_txt)) return fail(VGAN_EIO, "cannot write parents.txt");
    if (!write_file(dir + "children.txt", g->children_txt)) return fail(VGAN_EIO, "cannot write children.txt");
    return VGAN_OK;
}

extern "C" void vgan_graph_free(vgan_graph *g) { delete g; }
