/*
 * oracle/hc_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * HaploCart per-read likelihood + posterior, restated for the CPU in long double from the
 * reference's arithmetic.  Each function cites the reference lines it follows
 * (paths relative to /root/reference/src).  Quirk numbers Qn refer to SURVEY.md section 8a.
 * "parity unpinned" for the numeric values (no reference test pins them); a1 is pinned by the
 * reference's reconstruction KATs (tests/test_oracle_golden.py).
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

using std::string;
using std::vector;

namespace {

struct Edit {
    int32_t from_length, to_length;
    string sequence;
};
struct Mapping {
    int64_t node_id, offset;
    bool is_reverse;
    vector<Edit> edit;
};
struct Path {
    vector<Mapping> mapping;
};

/* libvgio edit predicates (not in the tree; published semantics) */
inline bool edit_is_match(const Edit &e) { return e.from_length == e.to_length && e.sequence.empty(); }
inline bool edit_is_sub(const Edit &e) { return e.from_length == e.to_length && !e.sequence.empty(); }
inline bool edit_is_insertion(const Edit &e) { return e.from_length == 0 && e.to_length > 0 && !e.sequence.empty(); }
inline bool edit_is_deletion(const Edit &e) { return e.from_length > 0 && e.to_length == 0; }

struct NodeMissing : std::runtime_error {
    NodeMissing() : std::runtime_error("node") {}
};

Path get_path(const orc_alnset_t *a, int64_t r) {
    Path p;
    for (int64_t m = a->map_off[r]; m < a->map_off[r + 1]; ++m) {
        Mapping mp;
        mp.node_id = a->m_node[m];
        mp.offset = a->m_offset[m];
        mp.is_reverse = a->m_rev[m] != 0;
        for (int64_t e = a->edit_off[m]; e < a->edit_off[m + 1]; ++e) {
            Edit ed;
            ed.from_length = a->e_from[e];
            ed.to_length = a->e_to[e];
            ed.sequence.assign(a->e_seq + a->e_seq_off[e], a->e_seq + a->e_seq_off[e + 1]);
            mp.edit.push_back(ed);
        }
        p.mapping.push_back(mp);
    }
    return p;
}

char complement(char c) {
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

/* graph.get_sequence(graph.get_handle(id, is_reverse)) */
string get_sequence(const orc_graph_t *g, int64_t id, bool rev) {
    if (id < g->min_id || id > g->max_id) throw NodeMissing();
    const int64_t i = id - g->min_id;
    string s(g->node_seq + g->node_seq_off[i], g->node_seq + g->node_seq_off[i + 1]);
    if (rev) {
        std::reverse(s.begin(), s.end());
        for (char &c : s) c = complement(c);
    }
    return s;
}

/* vg::algorithms::path_string (vg, not in tree): the sequence the path spells out --
 * node bases for matches, edit.sequence for substitutions and insertions, nothing for deletions. */
string path_string(const orc_graph_t *g, const Path &path) {
    string seq;
    for (const Mapping &m : path.mapping) {
        const string node_seq = get_sequence(g, m.node_id, m.is_reverse);
        size_t f = m.offset;
        for (const Edit &e : m.edit) {
            if (edit_is_match(e)) {
                seq.append(node_seq.substr(f, e.from_length));
            } else if (edit_is_sub(e)) {
                seq.append(e.sequence);
            } else if (edit_is_insertion(e)) {
                seq.append(e.sequence);
            }
            f += e.from_length;
        }
    }
    return seq;
}

/* vgan_utils.h:6-79.  Q7: mppg_counter and edit_counter never advance, so the softclip test is
 * "insertion while the running node offset is 0"; Q8: deletion gap inserted at f = sum of from_length. */
std::tuple<string, string, vector<int>> reconstruct_graph_sequence(const orc_graph_t *g, const Path &path,
                                                                   const string & /*algnseq*/) {
    string graph_seq = "";
    vector<int> mppg_sizes;
    const int mppg_counter = 0;
    int edit_counter = 0;
    string ps = path_string(g, path); /* :18 */
    const auto &mppgs = path.mapping;
    int f = 0;
    for (const Mapping &mppg : mppgs) { /* :22 */
        const string node_seq = get_sequence(g, mppg.node_id, mppg.is_reverse); /* :24 */
        int aligned_length = 0;
        const vector<Edit> &ed = mppg.edit;
        edit_counter = 0;
        int offset = (int)mppg.offset;
        for (const Edit &edit : ed) { /* :31 */
            const int32_t to_length = edit.to_length;
            const int32_t from_length = edit.from_length;
            const bool softclip =
                (mppg_counter == 0 && offset == 0 && edit_counter == 0 && from_length == 0 && to_length > 0 &&
                 edit_is_insertion(edit)) ||
                ((size_t)mppg_counter == mppgs.size() - 1 && offset == 0 && (size_t)edit_counter == ed.size() &&
                 from_length == 0 && to_length > 0 && edit_is_insertion(edit)); /* :38-39 */
            if (edit_is_match(edit) || edit_is_sub(edit)) { /* :41 */
                graph_seq += node_seq.substr(offset, from_length);
                aligned_length = (int)node_seq.substr(offset, from_length).size();
            } else if (edit_is_insertion(edit)) { /* :48 */
                if (softclip) {
                    graph_seq += string(to_length, 'S');
                    aligned_length = to_length;
                } else {
                    graph_seq += string(to_length, '-');
                    aligned_length = to_length;
                }
            } else if (edit_is_deletion(edit)) { /* :64 */
                graph_seq += node_seq.substr(offset, from_length);
                aligned_length = (int)node_seq.substr(offset, from_length).size();
                ps.insert(f, string(from_length, '-')); /* :67 */
            }
            offset += from_length; /* :69 */
            f += from_length;
            mppg_sizes.emplace_back(aligned_length); /* :72 */
        }
    }
    return std::make_tuple(graph_seq, ps, mppg_sizes);
}

/* miscfunc.h:180-188 (Q10) */
inline double get_p_seq_error(const int &Q) {
    if (Q > 2) {
        return pow(10, ((-1 * Q) * 0.1));
    } else {
        return 0.25;
    }
}

/* miscfunc.h:199-212: Q>=2 takes get_p_seq_error (which itself yields 0.25 for Q==2) */
vector<double> get_qscore_vec() {
    vector<double> qscore_vec;
    for (int Q = 0; Q < 100; ++Q) {
        if (Q >= 2) {
            qscore_vec.emplace_back(get_p_seq_error(Q));
        } else {
            qscore_vec.emplace_back(0.25);
        }
    }
    return qscore_vec;
}

/* haplocart_functions.cpp:101-107 */
vector<double> precompute_incorrect_mapping_probs() {
    vector<double> v;
    for (int Q = 0; Q != 100; ++Q) v.emplace_back(pow(10, ((-1 * Q) * 0.1)));
    return v;
}

/* haplocart_functions.cpp:81-98 */
double get_background_freq(const char base) {
    switch (base) {
    case 'A': return 0.27532;
    case 'C': return 0.30044;
    case 'G': return 0.16644;
    case 'T': return 0.25780;
    }
    return 0.25;
}

/* libgab isValidDNA (not in tree): upper-case A,C,G,T only */
inline bool isValidDNA(const char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

inline bool inRange(const unsigned low, const unsigned high, const unsigned x) { return (low <= x && x <= high); }

/* get_p_obs_base.cpp:38-69.  Q1: (22/23) and (1/46) are int divisions = 0; Q2: protein-coding mu = 0. */
long double get_p_obs_base(const int pangenome_base, const double epsilon, const int generations) {
    double mu;
    if (inRange(57, 372, pangenome_base)) {
        mu = 1.64273e-7;
    } else if (inRange(1, 56, pangenome_base) || inRange(373, 576, pangenome_base)) {
        mu = 2.29640e-8;
    } else if (inRange(16384, 16569, pangenome_base)) {
        mu = 1.54555e-8;
    } else if (inRange(3307, 4262, pangenome_base) || inRange(4470, 5511, pangenome_base) ||
               inRange(5904, 7445, pangenome_base) || inRange(7586, 8269, pangenome_base) ||
               inRange(8366, 9990, pangenome_base) || inRange(10059, 10403, pangenome_base) ||
               inRange(10470, 12137, pangenome_base) || inRange(12337, 14673, pangenome_base) ||
               inRange(14747, 15886, pangenome_base)) {
        mu = 8.87640e-9 * (2 / 3) * 1.92596e-8 * (1 / 3);
    } else if (inRange(577, 647, pangenome_base) || inRange(1602, 1670, pangenome_base) ||
               inRange(3230, 3304, pangenome_base) || inRange(4263, 4400, pangenome_base) ||
               inRange(4402, 4469, pangenome_base) || inRange(5512, 5579, pangenome_base) ||
               inRange(5587, 5654, pangenome_base) || inRange(5657, 5728, pangenome_base) ||
               inRange(5761, 5891, pangenome_base) || inRange(7446, 7514, pangenome_base) ||
               inRange(7518, 7585, pangenome_base) || inRange(8295, 8364, pangenome_base) ||
               inRange(15888, 15953, pangenome_base) || inRange(15956, 16023, pangenome_base)) {
        mu = 6.91285e-9;
    } else if (inRange(648, 1601, pangenome_base) || inRange(1671, 3229, pangenome_base)) {
        mu = 6.91285e-9;
    } else {
        mu = 2.48537e-8;
    }
    mu *= 30;
    const double match = pow((1 - mu), generations);
    const double tv = (1 - pow((1 - mu), generations)) * (22 / 23);
    const double ts = (1 - pow((1 - mu), generations)) * (1 / 46);
    long double ret = match * (1 - epsilon) + (epsilon * (2 * tv + ts));
    return ret;
}

/* libgab oplusnatl / oplusInitnatl (not in tree, Q11) */
inline long double oplusnatl(long double x, long double y) {
    if (x > y) return x + log1pl(expl(y - x));
    return y + log1pl(expl(x - y));
}
inline long double oplusInitnatl(long double x, long double y) {
    if (x == 0) return y;
    return oplusnatl(x, y);
}

struct Terminated : std::runtime_error {
    int code;
    explicit Terminated(int c) : std::runtime_error("terminated"), code(c) {}
};

struct HcCtx {
    const orc_graph_t *g;
    vector<double> qscore_vec;
    vector<double> incorrect_mapping_vec;
    std::map<const string, int> pangenome_map; /* kept as a string map: part of the reference's cost */
    int32_t flags = 0;
    /* optional path view for orc_hc_read_segments: 2 virtual paths (always / never supported) */
    bool virtual_paths = false;
};

inline int clampq(int Q, int32_t &flags) {
    if (Q < 0) {
        flags |= 1;
        return 0;
    }
    if (Q > 99) {
        flags |= 1;
        return 99;
    }
    return Q;
}

/* process_mapping.cpp:4-24 (Q3: counter % 4 == 4 is never true) */
inline double get_log_lik_if_unsupported(const vector<int> &quality_scores) {
    bool mismatch = true;
    int counter = 0;
    double ret = 0;
    for (const int &Q : quality_scores) {
        if (mismatch) {
            ret += log(get_p_seq_error(Q));
        } else {
            ret += log(1 - get_p_seq_error(Q));
        }
        counter += 1;
        if (counter % 4 == 4) {
            mismatch = false;
        } else {
            mismatch = true;
        }
    }
    return ret;
}

/* get_p_obs_base.cpp:3-27 */
vector<long double> get_p_no_seq_error_mapping(HcCtx &c, const string &mapping_seq, const vector<int> &quality_scores,
                                               const string &graph_seq, const bool use_background_error_prob,
                                               const double &background_error_prob) {
    vector<long double> ret;
    for (size_t i = 0; i < graph_seq.size(); ++i) {
        const char graph_base = graph_seq[i];
        /* std::string::operator[] at size() is '\0'; beyond is UB -> oracle defines '\0' */
        const char read_base = i < mapping_seq.size() ? mapping_seq[i] : '\0';
        if (use_background_error_prob) {
            if (graph_base == read_base) {
                ret.emplace_back(background_error_prob);
            } else {
                ret.emplace_back(1 - background_error_prob);
            }
        } else {
            /* quality_scores[i] past its end is UB -> oracle defines Q=0 */
            const int Q = clampq(i < quality_scores.size() ? quality_scores[i] : 0, c.flags);
            if (graph_base == read_base) {
                ret.emplace_back(c.qscore_vec[Q]);
            } else {
                ret.emplace_back(1 - c.qscore_vec[Q]);
            }
        }
    }
    return ret;
}

inline bool pathsgo(const HcCtx &c, int64_t node_id, int i) {
    if (c.virtual_paths) return i == 0;
    return c.g->pathsgo[(size_t)node_id * c.g->n_paths + i] != 0;
}

/* process_mapping.cpp:26-91 (vector passed and returned by value as in the reference) */
vector<long double> process_mapping(HcCtx &c, const int32_t mapping_quality, vector<long double> log_likelihood_vec,
                                    const Mapping &mppg, string &mapping_seq, const vector<int> &quality_scores,
                                    string &graph_seq, const int &nbpaths, bool &use_background_error_prob,
                                    const double &background_error_prob, const bool &is_consensus_fasta) {
    auto it = c.pangenome_map.find(std::to_string(mppg.node_id)); /* :33 .at() */
    if (it == c.pangenome_map.end()) throw Terminated(ORC_ERR_NODE);
    const int pangenome_base = it->second;
    if (pangenome_base < 0 || pangenome_base >= c.g->n_mappability) throw Terminated(ORC_ERR_TABLE);
    const double mappability = c.g->mappability[pangenome_base]; /* :35 */
    vector<double> background_freqs;
    long double p_correctly_mapped = 0;
    if (is_consensus_fasta == false) {
        p_correctly_mapped = (1 - c.incorrect_mapping_vec[clampq(mapping_quality, c.flags)]) * mappability; /* :41 */
    }
    for (size_t i = 0; i != mapping_seq.size(); ++i) { /* :46-48 */
        background_freqs.emplace_back(get_background_freq(mapping_seq[i]));
    }
    const vector<long double> p_no_seq_error = get_p_no_seq_error_mapping(
        c, mapping_seq, quality_scores, graph_seq, use_background_error_prob, background_error_prob); /* :50 */

    if (mppg.node_id < c.g->min_id || mppg.node_id > c.g->max_id) throw Terminated(ORC_ERR_NODE); /* :55 .at() */

    for (int i = 0; i != nbpaths; ++i) { /* :54 */
        const bool is_supported = pathsgo(c, mppg.node_id, i);
        if (is_supported) {
            long double log_lik_if_mapped = 0;
            for (size_t j = 0; j != graph_seq.size(); ++j) {
                const char mj = j < mapping_seq.size() ? mapping_seq[j] : '\0';
                if (graph_seq[j] == 'N' || mj == 'N') continue;            /* :62 */
                if (!isValidDNA(graph_seq[j]) || !isValidDNA(mj)) continue; /* :63 */
                const long double p_obs_base = get_p_obs_base(pangenome_base, (double)p_no_seq_error[j], 8); /* :66 */
                long double log_prod;
                if (is_consensus_fasta == false) {
                    log_prod = logl(((1 - p_correctly_mapped) * background_freqs[j]) + (p_correctly_mapped * p_obs_base)); /* :70 */
                } else {
                    log_prod = logl((1 - background_error_prob) * p_obs_base); /* :73 */
                }
                log_lik_if_mapped += log_prod;
            }
            log_likelihood_vec[i] += log_lik_if_mapped; /* :79 */
        } else {
            const double log_lik_if_unsupported = get_log_lik_if_unsupported(quality_scores); /* :84 */
            log_likelihood_vec[i] += log_lik_if_unsupported;
        }
    }
    return log_likelihood_vec;
}

/* hoisted variant: same sums, S_m/U_m computed once per mapping.  NOT the reference's cost profile. */
void process_mapping_hoisted(HcCtx &c, const int32_t mapping_quality, vector<long double> &log_likelihood_vec,
                             const Mapping &mppg, string &mapping_seq, const vector<int> &quality_scores,
                             string &graph_seq, const int &nbpaths, bool &use_background_error_prob,
                             const double &background_error_prob, const bool &is_consensus_fasta,
                             long double *S_out = nullptr, double *U_out = nullptr) {
    auto it = c.pangenome_map.find(std::to_string(mppg.node_id));
    if (it == c.pangenome_map.end()) throw Terminated(ORC_ERR_NODE);
    const int pangenome_base = it->second;
    if (pangenome_base < 0 || pangenome_base >= c.g->n_mappability) throw Terminated(ORC_ERR_TABLE);
    const double mappability = c.g->mappability[pangenome_base];
    long double p_correctly_mapped = 0;
    if (is_consensus_fasta == false) {
        p_correctly_mapped = (1 - c.incorrect_mapping_vec[clampq(mapping_quality, c.flags)]) * mappability;
    }
    const vector<long double> p_no_seq_error = get_p_no_seq_error_mapping(
        c, mapping_seq, quality_scores, graph_seq, use_background_error_prob, background_error_prob);
    if (mppg.node_id < c.g->min_id || mppg.node_id > c.g->max_id) throw Terminated(ORC_ERR_NODE);
    long double S = 0;
    for (size_t j = 0; j != graph_seq.size(); ++j) {
        const char mj = j < mapping_seq.size() ? mapping_seq[j] : '\0';
        if (graph_seq[j] == 'N' || mj == 'N') continue;
        if (!isValidDNA(graph_seq[j]) || !isValidDNA(mj)) continue;
        const long double p_obs_base = get_p_obs_base(pangenome_base, (double)p_no_seq_error[j], 8);
        long double log_prod;
        if (is_consensus_fasta == false) {
            log_prod = logl(((1 - p_correctly_mapped) * get_background_freq(mj)) + (p_correctly_mapped * p_obs_base));
        } else {
            log_prod = logl((1 - background_error_prob) * p_obs_base);
        }
        S += log_prod;
    }
    const double U = get_log_lik_if_unsupported(quality_scores);
    if (S_out) *S_out = S;
    if (U_out) *U_out = U;
    if (nbpaths > 0) {
        for (int i = 0; i != nbpaths; ++i) {
            if (pathsgo(c, mppg.node_id, i))
                log_likelihood_vec[i] += S;
            else
                log_likelihood_vec[i] += U;
        }
    }
}

struct SegSink {
    vector<double> S, U;
    vector<int64_t> node;
};

/* update_likelihood.cpp:19-53.  Q4: process_mapping receives the whole algnseq as mapping_seq;
 * Q5: the quality window runs algnseq.size() entries from position_in_read (out-of-range -> 0);
 * Q6: mppg_sizes is indexed by MAPPING but holds per-EDIT sizes. */
vector<long double> update_likelihood(HcCtx &c, const orc_alnset_t *a, int64_t r, vector<long double> log_likelihood_vec,
                                      const int nbpaths, bool use_background_error_prob,
                                      const double &background_error_prob, const bool is_consensus_fasta,
                                      bool faithful, SegSink *sink = nullptr) {
    const Path path = get_path(a, r);
    const string seq(a->seq + a->seq_off[r], a->seq + a->seq_off[r + 1]);
    const string quality(a->qual + a->qual_off[r], a->qual + a->qual_off[r + 1]);
    std::tuple<string, string, vector<int>> seq_tuple;
    try {
        seq_tuple = reconstruct_graph_sequence(c.g, path, seq); /* :29 */
    } catch (const NodeMissing &) {
        throw Terminated(ORC_ERR_NODE);
    } catch (const std::out_of_range &) {
        throw Terminated(ORC_ERR_SUBSTR);
    }
    string algnseq = std::get<1>(seq_tuple);
    int position_in_read = 0;
    for (size_t i = 0; i < path.mapping.size(); ++i) { /* :33 */
        vector<int> quality_scores;
        const vector<int> mppg_sizes = std::get<2>(seq_tuple); /* :35 (copy per mapping, as in the reference) */
        if (i >= mppg_sizes.size()) throw Terminated(ORC_ERR_SIZES);
        if ((size_t)position_in_read > std::get<0>(seq_tuple).size() || (size_t)position_in_read > algnseq.size())
            throw Terminated(ORC_ERR_SUBSTR);
        string graph_seq = std::get<0>(seq_tuple).substr(position_in_read, mppg_sizes[i]); /* :36 */
        string read_seq = algnseq.substr(position_in_read, mppg_sizes[i]);                 /* :37 */
        for (size_t j = position_in_read; j < position_in_read + algnseq.size(); ++j) {    /* :40 */
            const int qscore = j < quality.size() ? int(quality[j]) : 0; /* Q5 */
            if (qscore >= 90) use_background_error_prob = true; /* :42 sticky */
            quality_scores.emplace_back(qscore);
        }
        position_in_read += (int)read_seq.size(); /* :45 */
        if (sink) {
            long double S;
            double U;
            vector<long double> none;
            process_mapping_hoisted(c, a->mapq[r], none, path.mapping[i], algnseq, quality_scores, graph_seq, 0,
                                    use_background_error_prob, background_error_prob, is_consensus_fasta, &S, &U);
            sink->S.push_back((double)S);
            sink->U.push_back(U);
            sink->node.push_back(path.mapping[i].node_id);
        } else if (faithful) {
            log_likelihood_vec = process_mapping(c, a->mapq[r], log_likelihood_vec, path.mapping[i], algnseq,
                                                 quality_scores, graph_seq, nbpaths, use_background_error_prob,
                                                 background_error_prob, is_consensus_fasta); /* :46 */
        } else {
            process_mapping_hoisted(c, a->mapq[r], log_likelihood_vec, path.mapping[i], algnseq, quality_scores,
                                    graph_seq, nbpaths, use_background_error_prob, background_error_prob,
                                    is_consensus_fasta);
        }
    }
    return log_likelihood_vec;
}

void init_ctx(HcCtx &c, const orc_graph_t *g) {
    c.g = g;
    c.qscore_vec = get_qscore_vec();
    c.incorrect_mapping_vec = precompute_incorrect_mapping_probs();
    for (int64_t id = 0; id <= g->max_id; ++id) {
        if (g->pangenome_base[id] >= 0) c.pangenome_map.insert(std::make_pair(std::to_string(id), g->pangenome_base[id]));
    }
}

vector<string> split_ws(const string &line) {
    vector<string> t;
    std::istringstream is(line);
    string tok;
    while (is >> tok) t.push_back(tok);
    return t;
}

/* load.cpp:303-345 */
std::map<string, vector<string>> load_relatives(const char *txt) {
    std::map<string, vector<string>> rel;
    std::istringstream in(txt ? txt : "");
    string line;
    while (std::getline(in, line)) {
        const vector<string> tokens = split_ws(line);
        if (tokens.size() == 0) continue;
        vector<string> v;
        for (size_t j = 1; j < tokens.size(); ++j) {
            if (tokens[j].find('[') == string::npos) v.emplace_back(tokens[j]);
        }
        rel.insert(std::make_pair(tokens[0], v));
    }
    return rel;
}

/* get_posterior.cpp:36-49 */
std::set<string> get_children(const std::set<string> &preds, const std::map<string, vector<string>> &children) {
    std::set<string> all_child_set;
    for (const string &p : preds) {
        auto it = children.find(p);
        if (it == children.end()) continue; /* reference dereferences end(): UB; oracle: no children */
        for (const string &child : it->second) all_child_set.insert(child);
    }
    return all_child_set;
}

/* get_posterior.cpp:51-76 */
vector<long double> get_posterior_of_clade(vector<long double> &all_top, const vector<long double> &final_vec,
                                           const std::set<string> &preds,
                                           const std::map<string, vector<string>> &children,
                                           const vector<string> &path_names, int depth = 0) {
    std::set<string> child_set = get_children(preds, children);
    int idx = 0;
    for (const string &path : path_names) {
        if (child_set.find(path) != child_set.end()) all_top.emplace_back(final_vec[idx]);
        idx += 1;
    }
    if (child_set.size() > 0 && depth < 100000) {
        all_top = get_posterior_of_clade(all_top, final_vec, child_set, children, path_names, depth + 1);
    }
    return all_top;
}

/* get_posterior.cpp:78-85 */
long double sum_log_likelihoods(const vector<long double> &v) {
    if (v.empty()) return 0; /* reference reads v[0] of an empty vector: UB; oracle: 0 */
    long double ret = v[0];
    for (size_t i = 1; i < v.size(); ++i) ret = oplusInitnatl(ret, v[i]);
    return ret;
}

} // namespace

extern "C" {

int orc_reconstruct(const orc_graph_t *g, const orc_alnset_t *a, int64_t r, char *graph_seq, char *read_seq,
                    int32_t *sizes, int64_t cap, int64_t *lens) {
    try {
        const Path path = get_path(a, r);
        const string seq(a->seq + a->seq_off[r], a->seq + a->seq_off[r + 1]);
        auto t = reconstruct_graph_sequence(g, path, seq);
        const string &gs = std::get<0>(t);
        const string &rs = std::get<1>(t);
        const vector<int> &sz = std::get<2>(t);
        if ((int64_t)gs.size() > cap || (int64_t)rs.size() > cap || (int64_t)sz.size() > cap) return ORC_ERR_CAP;
        memcpy(graph_seq, gs.data(), gs.size());
        memcpy(read_seq, rs.data(), rs.size());
        for (size_t i = 0; i < sz.size(); ++i) sizes[i] = sz[i];
        lens[0] = (int64_t)gs.size();
        lens[1] = (int64_t)rs.size();
        lens[2] = (int64_t)sz.size();
        return ORC_OK;
    } catch (const NodeMissing &) {
        return ORC_ERR_NODE;
    } catch (const std::out_of_range &) {
        return ORC_ERR_SUBSTR;
    }
}

int orc_hc_read(const orc_graph_t *g, const orc_alnset_t *a, int64_t r, const orc_hc_params_t *p, long double *out,
                int32_t *flags) {
    HcCtx c;
    init_ctx(c, g);
    vector<long double> empty_vec(g->n_paths, 0.0L);
    try {
        vector<long double> v = update_likelihood(c, a, r, empty_vec, g->n_paths, p->use_background_error_prob != 0,
                                                  p->background_error_prob, p->is_consensus_fasta != 0, true);
        for (int i = 0; i < g->n_paths; ++i) out[i] = v[i];
    } catch (const Terminated &t) {
        if (flags) *flags = c.flags;
        return t.code;
    }
    if (flags) *flags = c.flags;
    return ORC_OK;
}

int orc_hc_read_segments(const orc_graph_t *g, const orc_alnset_t *a, int64_t r, const orc_hc_params_t *p, double *S,
                         double *U, int64_t *node, int64_t cap, int64_t *n_seg) {
    HcCtx c;
    init_ctx(c, g);
    SegSink sink;
    vector<long double> none;
    int rc = ORC_OK;
    try {
        update_likelihood(c, a, r, none, 0, p->use_background_error_prob != 0, p->background_error_prob,
                          p->is_consensus_fasta != 0, false, &sink);
    } catch (const Terminated &t) {
        rc = t.code;
    }
    if ((int64_t)sink.S.size() > cap) return ORC_ERR_CAP;
    for (size_t i = 0; i < sink.S.size(); ++i) {
        S[i] = sink.S[i];
        U[i] = sink.U[i];
        node[i] = sink.node[i];
    }
    *n_seg = (int64_t)sink.S.size();
    return rc;
}

int orc_hc_run(const orc_graph_t *g, const orc_alnset_t *a, int64_t r0, int64_t r1, const orc_hc_params_t *p,
               int n_threads, int faithful, long double *final_ld, double *final_d, int64_t *n_bad) {
    HcCtx c0;
    init_ctx(c0, g);
    const int nbpaths = g->n_paths;
    vector<long double> final_vec(nbpaths, 0.0L);
    const vector<long double> empty_vec(nbpaths, 0.0L);
    int64_t bad = 0;
    if (n_threads < 1) n_threads = 1;
    /* HaploCart.cpp:408-421: omp parallel for over reads, private per-read vector, critical accumulate */
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
    for (int64_t i = r0; i < r1; ++i) {
        if (a->identity[i] < 1e-10) continue; /* :410 */
        HcCtx c = c0;                         /* tables are read-only shared state in the reference; flags are ours */
        vector<long double> log_likelihood_vec;
        bool ok = true;
        try {
            log_likelihood_vec = update_likelihood(c, a, i, empty_vec, nbpaths, p->use_background_error_prob != 0,
                                                   p->background_error_prob, p->is_consensus_fasta != 0, faithful != 0);
        } catch (const Terminated &) {
            ok = false;
        }
#pragma omp critical
        {
            if (ok) {
                for (int j = 0; j < nbpaths; ++j) final_vec[j] += log_likelihood_vec[j]; /* :420 */
            } else {
                bad += 1;
            }
        }
    }
    for (int j = 0; j < nbpaths; ++j) {
        if (final_ld) final_ld[j] = final_vec[j];
        if (final_d) final_d[j] = (double)final_vec[j];
    }
    if (n_bad) *n_bad = bad;
    return ORC_OK;
}

int orc_hc_posterior(const long double *final_in, int32_t n_paths, const char *path_names_txt, const char *parents_txt,
                     const char *children_txt, const char *predicted, char *out, int64_t cap, double *conf,
                     int32_t conf_cap) {
    vector<string> path_names;
    {
        std::istringstream in(path_names_txt ? path_names_txt : "");
        string line;
        while (std::getline(in, line)) {
            const vector<string> tokens = split_ws(line);
            if (tokens.empty()) continue;
            path_names.emplace_back(tokens[0]); /* load.cpp:53, observable behaviour = whole first token */
        }
    }
    if ((int32_t)path_names.size() != n_paths) return -10;
    const vector<long double> final_vec(final_in, final_in + n_paths);
    const auto parents = load_relatives(parents_txt);
    const auto children = load_relatives(children_txt);
    const string predicted_haplotype(predicted);

    /* get_posterior.cpp:87-127 */
    const long double total_ll = sum_log_likelihoods(final_vec);
    vector<string> parent_vec;
    if (parents.count(predicted_haplotype) > 0) parent_vec = parents.find(predicted_haplotype)->second;
    vector<string> clade_vec{predicted_haplotype};
    vector<double> confidence_vec;
    std::set<string> pred{predicted_haplotype};
    const int predicted_haplotype_idx =
        (int)(std::find(path_names.begin(), path_names.end(), predicted_haplotype) - path_names.begin());
    if (predicted_haplotype_idx >= n_paths) return -11;
    vector<long double> all_top{final_vec[predicted_haplotype_idx]};
    long double considered_ll = sum_log_likelihoods(all_top);
    long double top_ratio = expl(considered_ll - total_ll);
    confidence_vec.emplace_back((double)top_ratio);
    all_top.clear();
    pred.clear();
    for (size_t j = 0; j < parent_vec.size(); ++j) {
        const bool differs = (j == 0) || (parent_vec[j] != parent_vec[j - 1]); /* Q9: j-1 at j=0 is OOB -> "different" */
        if (differs) clade_vec.emplace_back(parent_vec[j]);
        pred.insert(parent_vec[j]);
        all_top = get_posterior_of_clade(all_top, final_vec, pred, children, path_names);
        considered_ll = sum_log_likelihoods(all_top);
        top_ratio = expl(considered_ll - total_ll);
        if (differs) confidence_vec.emplace_back((double)top_ratio);
        all_top.clear();
        pred.clear();
    }
    string s;
    char buf[64];
    for (size_t i = 0; i < clade_vec.size(); ++i) {
        snprintf(buf, sizeof buf, "%.17g", confidence_vec[i]);
        s += clade_vec[i] + "\t" + buf + "\t" + std::to_string(i) + "\n";
        if ((int32_t)i < conf_cap && conf) conf[i] = confidence_vec[i];
    }
    if ((int64_t)s.size() + 1 > cap) return ORC_ERR_CAP;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)clade_vec.size();
}

double orc_p_seq_error(int Q) { return get_p_seq_error(Q); }
double orc_qscore(int Q) {
    static const vector<double> v = get_qscore_vec();
    return v[Q < 0 ? 0 : (Q > 99 ? 99 : Q)];
}
double orc_p_incorrect_mapping(int Q) { return pow(10, ((-1 * Q) * 0.1)); }
double orc_background_freq(char c) { return get_background_freq(c); }
long double orc_oplusnatl(long double x, long double y) { return oplusnatl(x, y); }
long double orc_oplusInitnatl(long double x, long double y) { return oplusInitnatl(x, y); }
long double orc_p_obs_base(int pangenome_base, double epsilon, int generations) {
    return get_p_obs_base(pangenome_base, epsilon, generations);
}

/* load.cpp:6-24 */
int64_t orc_load_mappabilities(const char *txt, double *out, int64_t cap) {
    std::istringstream in(txt);
    string line;
    int64_t n = 0;
    while (std::getline(in, line)) {
        const vector<string> tokens = split_ws(line);
        if (tokens.size() < 4) continue; /* reference indexes tokens[3] unconditionally */
        const double v = std::stod(tokens[3]);
        for (int i = std::stoi(tokens[1]); i < std::stoi(tokens[2]); ++i) {
            if (n < cap) out[n] = v;
            ++n;
        }
    }
    return n;
}

/* load.cpp:27-41: key = node id string, value = stoi(tokens[1]) + 1 */
int64_t orc_load_pangenome_map(const char *txt, int32_t *base_by_id, int64_t cap) {
    std::istringstream in(txt);
    string line;
    int64_t n = 0;
    while (std::getline(in, line)) {
        const vector<string> tokens = split_ws(line);
        if (tokens.size() < 2) continue;
        const long id = std::stol(tokens[0]);
        const int val = std::stoi(tokens[1]) + 1;
        if (id >= 0 && id < cap) base_by_id[id] = val;
        ++n;
    }
    return n;
}

/* load.cpp:283-300: one row per line from index 0, first n_paths characters, '1' = supported */
int64_t orc_load_path_supports(const char *txt, int32_t n_paths, uint8_t *out, int64_t cap_rows) {
    std::istringstream in(txt);
    string line;
    int64_t index = 0;
    while (std::getline(in, line)) {
        if (index < cap_rows) {
            for (int j = 0; j < n_paths; ++j) {
                const char supported = j < (int)line.size() ? line[j] : '0';
                out[(size_t)index * n_paths + j] = supported == '1';
            }
        }
        index += 1;
    }
    return index;
}

} /* extern "C" */
