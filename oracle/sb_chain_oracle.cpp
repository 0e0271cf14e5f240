/*
 * oracle/sb_chain_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * What `vgan soibean` does after analyse_GAM, restated for the CPU: a pointer-linked tree like the reference's, the same
 * sequence of random draws, comparisons and floating-point operations (incl. its double / long double mixes and unsigned
 * arithmetic), written independently of the product's index-based implementation (paths relative to /root/reference/src):
 *   MCMC.cpp:169-470       MCMC::updatePosition
 *   MCMC.cpp:487-520       MCMC::sample_normal
 *   MCMC.cpp:522-1093      MCMC::run_tree_proportion (likelihood through orc_sb_loglike, oracle/sb_oracle.cpp)
 *   MCMC.cpp:23-150        MCMC::processMCMCiterations;  MCMC.h:424-505 state initialisation, :507-526 getQuantile2,
 *                          :528-625 findLCA / calculateDistanceToAncestor / getPatristicDistances / calculateEuclideanDistance
 *   miscfunc.h:12-66       mean, variance, autocorrelation, effectiveSampleSize
 *   soibean.cpp:157-202    generateRandomNumbers, calculateRhat;  :738-944 the source / chain loop and the diagnostics file
 * "parity unpinned": the reference's soibean tests (src/test.cpp) map FASTQ with giraffe against a database and check the
 * estimated source; neither is available here.  spidir (the tree library) is not in the reference tree: the Newick reader
 * below numbers nodes in pre-order of the text.
 * Randomness and undefined behaviour are resolved as include/vgan_gpu.h states for vgan_sb_estimate (seed stream standing in
 * for std::random_device, per chain one mt19937 for rand() and one for sample_normal's static engine; distances beyond #leaves ignored;
 * branches in name order with the reference's {1, 1, 1, 1} defaults for chains that did not end on them).
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <numeric>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include <zlib.h>

using namespace std;

namespace {

struct Node {
    int name = 0;
    string longname;
    double dist = 0.0;
    Node *parent = nullptr;
    Node **children = nullptr;
    int nchildren = 0;
    vector<Node *> kids;
    bool isLeaf() const { return nchildren == 0; }
};
struct Tree {
    vector<Node *> nodes;
    ~Tree() {
        for (Node *n : nodes) delete n;
    }
};

Node *read_node(const string &s, size_t &p, Tree &t, Node *parent) {
    Node *n = new Node();
    n->name = (int)t.nodes.size();
    n->parent = parent;
    t.nodes.push_back(n);
    auto ws = [&]() { while (p < s.size() && isspace((unsigned char)s[p])) ++p; };
    ws();
    if (p < s.size() && s[p] == '(') {
        ++p;
        while (true) {
            n->kids.push_back(read_node(s, p, t, n));
            ws();
            if (p < s.size() && s[p] == ',') { ++p; continue; }
            if (p < s.size() && s[p] == ')') { ++p; break; }
            throw runtime_error("bad newick");
        }
    }
    ws();
    size_t b = p;
    while (p < s.size() && !strchr(",():;", s[p]) && !isspace((unsigned char)s[p])) ++p;
    n->longname = s.substr(b, p - b);
    ws();
    if (p < s.size() && s[p] == ':') {
        ++p;
        ws();
        char *e = nullptr;
        n->dist = strtod(s.c_str() + p, &e);
        if (e == s.c_str() + p) throw runtime_error("bad newick length");
        p = (size_t)(e - s.c_str());
    }
    return n;
}

void read_newick(const string &s, Tree &t) {
    size_t p = 0;
    read_node(s, p, t, nullptr);
    for (Node *n : t.nodes) {
        n->nchildren = (int)n->kids.size();
        n->children = n->kids.empty() ? nullptr : n->kids.data();
    }
}

struct Entropy {
    uint64_t s;
    bool hw;
    explicit Entropy(uint64_t seed) : s(seed), hw(seed == 0) {}
    uint32_t operator()() {
        if (hw) return random_device{}();
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (uint32_t)(z >> 32);
    }
};

// A source's place on the tree (PosTree, MCMC.h:38-44): the branch above `at`, a fraction of the way down it, and its share
struct Placement {
    Node *at;
    double frac;  // pos_branch
    double share; // theta
};

// One state of the chain (MCMCiteration, MCMC.h:46-52)
struct State {
    vector<Placement> src;
    vector<double> shares; // proportions
    double loglik;
};

struct Run {
    const void *h;               /* orc_sb_analyse handle */
    map<string, int> path_index; /* node label -> graph path */
    Entropy rd;
    mt19937 rand_engine;  /* stands in for rand() */
    mt19937 theta_engine; /* sample_normal's function-static generator */
    Run(const void *handle, uint64_t seed) : h(handle), rd(seed) {} /* both engines are re-seeded from rd when a chain starts */
    unsigned pick(unsigned n) { return rand_engine() % n; } /* rand() % n */
};

/* ---- series statistics (miscfunc.h:12-66): the reference keeps the running sums of mean and autocorrelation in doubles while
 * the samples are long doubles; the same here ---- */
long double series_mean(const vector<long double> &x) {
    double running = 0.0;
    for (size_t i = 0; i < x.size(); ++i) running = running + x[i];
    return running / x.size();
}
long double series_variance(const vector<long double> &x, long double about) {
    long double ss = 0.0;
    for (size_t i = 0; i < x.size(); ++i) ss += (x[i] - about) * (x[i] - about);
    return ss / (x.size() - 1);
}
long double series_autocorr(const vector<long double> &x, int lag) {
    const long double mu = series_mean(x);
    const long double var = series_variance(x, mu);
    double cross = 0.0;
    const size_t pairs = x.size() - lag;
    for (size_t i = 0; i < pairs; ++i) cross += ((x[i] - mu) * (x[i + lag] - mu));
    return cross / (pairs * var);
}
double series_ess(const vector<long double> &x) {
    const int half = x.size() / 2;
    double a = 1.0, b = series_autocorr(x, 1);
    double total = a + b;
    for (int lag = 1; (lag < half - 2) && (a + b > 0); lag += 2) {
        a = series_autocorr(x, lag + 1);
        b = series_autocorr(x, lag + 2);
        total += 2.0 * (a + b);
    }
    if (a + b < 0) total -= (a + b);
    return x.size() / (1 + total);
}
long double sorted_quantile(const vector<long double> &sorted, double q) { /* MCMC.h getQuantile2 */
    const double where = (sorted.size() - 1) * q;
    const size_t below = static_cast<size_t>(floor(where)), above = static_cast<size_t>(ceil(where));
    if (below == above) return sorted[below];
    const double w = where - below;
    return (1.0 - w) * sorted[below] + w * sorted[above];
}

/* ---- distance of a placement to the leaves (MCMC.h:528-625).  The reference indexes a #leaves-long vector by node number and
 * compares it with a vector of ones: only node numbers below #leaves take part ---- */
double climb(Node *from, Node *until) { /* sum of branch lengths from `from` up to, not including, `until`; -1 if never met */
    double d = 0.0;
    Node *n = from;
    for (; n != nullptr && n != until; n = n->parent) d += (n->dist);
    return n == until ? d : -1.0;
}
Node *meet(Node *a, Node *b) {
    unordered_map<Node *, bool> above_a;
    for (Node *n = a; n; n = n->parent) above_a[n] = true;
    for (Node *n = b; n; n = n->parent)
        if (above_a.count(n)) return n;
    throw runtime_error("No common ancestor found.");
}
long double leaf_profile_distance(const Tree &tr, Node *at, int n_leaves, double down_from_top) {
    vector<double> profile(n_leaves, numeric_limits<double>::max());
    for (size_t i = 0; i < tr.nodes.size(); ++i) {
        Node *leaf = tr.nodes[i];
        if (!leaf->isLeaf()) continue;
        Node *m = meet(at, leaf);
        const double mine = climb(at, m) - down_from_top, theirs = climb(leaf, m);
        if (mine >= 0.0 && theirs >= 0.0 && i < profile.size()) profile[i] = mine + theirs;
    }
    long double ss = 0.0;
    for (int i = 0; i < n_leaves; ++i) { /* against initialPatristicDistances = 1.0 everywhere */
        if (profile[i] == numeric_limits<double>::max()) continue;
        const double diff = profile[i] - 1.0;
        ss += diff * diff;
    }
    return sqrt(ss);
}

/* ---- MCMC::updatePosition (MCMC.cpp:169-470): walk `distance` branch units from a placement, towards the leaves or the root.
 * Every branch counts as one unit; a leaf turns the walk round, the root sends it down a random child, and on the way up an
 * inner node may slip sideways into a sibling ---- */
Node *random_child(Run &R, Node *of) {
    Node *c = of->children[R.pick(of->nchildren)];
    if (c->dist < 0.0) throw runtime_error("Error: next branch length cannot be negative.");
    return c;
}
void walk(Run &R, Placement &pl, double distance, bool down) {
    if (pl.frac < 0.0 || pl.frac > 1.0) throw runtime_error("Error: Initial pos_branch is out of valid range.");
    if (!pl.at) throw runtime_error("Error: current position pointer is null.");
    if (distance < 0.0) throw runtime_error("Error: move distance cannot be negative.");
    double todo = abs(distance);
    while (todo > 0.0) {
        if (down) {
            if (pl.frac + todo < 1.0) { /* stays on this branch */
                pl.frac += todo;
                todo = 0.0;
            } else if (pl.at->children == nullptr) {
                down = false; /* bounce off the leaf */
            } else {
                double over = todo - (1.0 - pl.frac);
                if (over < 0.0) over = 0.0;
                pl.at = random_child(R, pl.at);
                if (over > 1.0) {
                    pl.frac = 1.0;
                    todo = over - 1.0;
                } else {
                    pl.frac = over;
                    todo = 0.0;
                }
            }
            continue;
        }
        /* upwards */
        if (pl.frac - todo > 0.0) {
            pl.frac = pl.frac - todo;
            todo = 0.0;
            continue;
        }
        Node *up = pl.at->parent;
        if (up == nullptr) { /* at the root: down again through a random child, nothing consumed */
            down = true;
            pl.at = random_child(R, pl.at);
            continue;
        }
        vector<Node *> choice(1, up);
        if (pl.at->children != nullptr) /* only an inner node considers its siblings */
            for (int i = 0; i < up->nchildren; ++i)
                if (up->children[i] != pl.at) choice.push_back(up->children[i]);
        Node *next = choice[R.pick(choice.size())];
        if (next == up) {
            double over = todo - pl.frac;
            if (over < 0.0) over = 0.0;
            pl.at = up;
            if (pl.at->dist < 0.0) throw runtime_error("Error: parent branch length cannot be negative.");
            if (over > 1.0) {
                pl.frac = 0.0;
                todo = over - 1.0;
            } else {
                const double f = 1.0 - over;
                if (f <= 0.0 || f >= 1.0) throw runtime_error("Error: new position branch is not in the valid range.");
                pl.frac = f;
                todo = 0.0;
            }
            continue;
        }
        /* sideways: enter the sibling at its top and carry on downwards */
        down = true;
        pl.at = next;
        const double over = todo - pl.frac;
        pl.frac = 0.0;
        if (pl.frac + over < 1.0) {
            pl.frac = over;
            if (pl.frac < 0.0 || pl.frac > 1.0) throw runtime_error("Error: pos_branch is out of valid range after increment.");
            todo = 0.0;
        } else if (pl.at->children == nullptr) {
            down = false; /* the sibling is a leaf: turn round there, the distance is kept */
        } else {
            double deeper = todo - (1.0 - pl.frac);
            if (deeper < 0.0) deeper = 0.0;
            pl.at = random_child(R, pl.at);
            if (deeper > 1.0) {
                pl.frac = 1.0;
                todo = deeper - 1.0;
            } /* otherwise position and distance stay as they are and the loop goes on downwards */
        }
    }
    if (pl.frac < 0.0 || pl.frac > 1.0) throw runtime_error("Error: pos_branch is out of valid range after movement.");
}

/* MCMC::sample_normal (MCMC.cpp:487-520): each share redrawn from N(share, 0.1) until it lies in [0, 1], then normalised */
vector<double> redraw_shares(Run &R, const vector<double> &cur) {
    if (cur.empty()) throw invalid_argument("vector can't be empty");
    vector<double> out;
    long double total = 0.0L;
    for (size_t i = 0; i < cur.size(); ++i) {
        normal_distribution<double> around(cur[i], 0.1);
        double v = around(R.theta_engine);
        while (v < 0.0L || v > 1.0L) v = around(R.theta_engine);
        out.push_back(v);
        total += v;
    }
    for (size_t i = 0; i < out.size(); ++i) out[i] /= total;
    return out;
}

/* initializeState (MCMC.h:424-505): uniform shares normalised to one from an engine of their own, every source half way
 * down the branch above its start node */
State start_state(Run &R, Tree &tr, const vector<int> &start_nodes, double start_loglik) {
    mt19937 gen(R.rd());
    uniform_real_distribution<> u(0.0, 1.0);
    vector<double> raw(start_nodes.size());
    double sum = 0.0;
    for (size_t i = 0; i < raw.size(); ++i) {
        raw[i] = u(gen);
        sum += raw[i];
    }
    State st;
    for (size_t i = 0; i < raw.size(); ++i) {
        raw[i] /= sum;
        st.src.push_back(Placement{tr.nodes.at(start_nodes[i]), 0.5, raw[i]});
        st.shares.push_back(max(0.001, raw[i]));
    }
    st.loglik = start_loglik;
    return st;
}

struct GzOut {
    gzFile f;
    explicit GzOut(const string &p) : f(gzopen(p.c_str(), "wb")) {
        if (!f) throw runtime_error("cannot write " + p);
    }
    ~GzOut() { gzclose(f); }
    void write(const string &s) {
        if (!s.empty()) gzwrite(f, s.data(), (unsigned)s.size());
    }
};

struct ChainCfg {
    unsigned max_iter, burn;
    double con;
    const double *freqs7;
    int n_paths;
};

double state_loglik(Run &R, const State &st, const ChainCfg &cfg) { /* MCMC.cpp:738-993 through orc_sb_loglike */
    const int k = (int)st.src.size();
    vector<int32_t> child(k), parent(k);
    vector<double> dist(k), pos(k), theta(k);
    for (int y = 0; y < k; ++y) {
        Node *n = st.src[y].at;
        child[y] = R.path_index.at(n->longname);
        parent[y] = R.path_index.at(n->parent ? n->parent->longname : n->longname);
        dist[y] = n->dist;
        pos[y] = st.src[y].frac;
        theta[y] = st.shares[y];
    }
    double ll = 0.0;
    if (orc_sb_loglike(R.h, k, child.data(), parent.data(), dist.data(), pos.data(), theta.data(), cfg.con, cfg.freqs7, 1, &ll) != 0)
        throw runtime_error("Problem in the likelihood compuation! Intermediate log likelihood is -nan, -inf or positive.");
    return ll;
}

string state_line(const State &st, double loglik, const char *verdict) {
    ostringstream o;
    o << setprecision(14);
    for (const Placement &p : st.src) {
        o << p.at->longname << '\t' << loglik << '\t' << p.share << '\t' << p.frac << '\t';
        if (verdict) o << verdict << '\t';
    }
    o << endl;
    return o.str();
}

/* MCMC::run_tree_proportion (MCMC.cpp:522-1093): returns the states recorded after the burn-in */
vector<State> run_chain(Run &R, Tree &tr, const vector<int> &start_nodes, double start_loglik, const ChainCfg &cfg, const string &prefix, int chain) {
    const unsigned k = start_nodes.size();
    R.rand_engine.seed(R.rd()); /* each chain has its own stand-ins for rand() and for sample_normal's static engine ... */
    R.theta_engine.seed(R.rd());
    mt19937 gen(R.rd()); /* ... next to the reference's own per-chain engine */
    uniform_real_distribution<> unit(0.0, 1.0);
    State cur = start_state(R, tr, start_nodes, start_loglik);
    const double widest = cfg.n_paths <= 30.0 ? 3.0 : cfg.n_paths * (3.0 / 30.0);
    const string tag = to_string(k) + to_string(chain);
    GzOut result(prefix + "Result" + tag + ".mcmc"), trace(prefix + "Trace" + tag + ".detail.mcmc");
    {
        ostringstream a, b;
        for (unsigned s = 1; s <= k; ++s) {
            a << "Source_" << s << '\t' << "Log-likelihood" << '\t' << "proportion" << '\t' << "branch_position_derived" << '\t';
            b << "Source_" << s << '\t' << "Log-likelihood" << '\t' << "proportion_" << s << '\t' << "branch_position_derived_" << s << '\t' << "Move" << '\t';
        }
        a << endl;
        b << endl;
        result.write(a.str());
        trace.write(b.str());
    }
    vector<State> kept;
    for (unsigned it = 0; it <= cfg.max_iter; it++) {
        if (cfg.burn >= cfg.max_iter) throw runtime_error("Number of brun in iteration exceedes the number of total iterations. Exiting. ");
        /* proposal width: linear from `widest` to 0.1 over the burn-in, then from 0.1 towards 1e-5; both denominators are unsigned */
        const double per_burn = (widest - 0.1) / std::max(static_cast<unsigned int>(1), cfg.burn - 1);
        const double per_rest = (0.1 - 1e-5) / std::max(static_cast<unsigned int>(1), (cfg.max_iter - cfg.burn) - 1);
        double width;
        if (it < cfg.burn) width = std::max(1e-5, widest - it * per_burn);
        else if (it % 100000 == 0) width = 1;
        else width = std::max(1e-5, 0.1 - (it - cfg.burn) * per_rest);
        State prop = cur;
        if (it != 0)
            for (size_t i = 0; i < prop.src.size(); i++) {
                normal_distribution<double> jump(0, width);
                const double d = jump(gen);
                if (d < 0.0) walk(R, prop.src[i], -d, false);
                else walk(R, prop.src[i], d, true);
            }
        vector<double> shares;
        for (const Placement &p : prop.src) shares.push_back(p.share);
        shares = redraw_shares(R, shares);
        for (size_t i = 0; i < prop.src.size(); ++i) prop.src[i].share = shares[i];
        prop.shares = shares;
        prop.loglik = state_loglik(R, prop, cfg);
        const double gain = prop.loglik - cur.loglik;
        const double accept = (gain > 0) ? 1.0 : exp(gain);
        const double u = unit(gen);
        const bool take = u <= accept || it == 0;
        trace.write(state_line(prop, prop.loglik, take ? "accepted" : "rejected"));
        if (it > cfg.burn) { /* what is recorded is the state the chain is in before the move */
            result.write(state_line(cur, cur.loglik, nullptr));
            kept.push_back(cur);
        }
        if (take) cur = prop;
    }
    return kept;
}

/* MCMC::processMCMCiterations (MCMC.cpp:23-150): per source the summaries of share and position over the recorded states, one line
 * each in the two estimate files; returns {mean share, its variance, mean position, its variance} under the branch the source
 * ended on, an empty entry for every other branch it visited, and the best log-likelihood */
struct ChainSummary {
    map<string, vector<vector<double>>> by_branch;
    double best;
};
ChainSummary summarise(const vector<State> &kept, int k, const string &prefix, int chain, const Tree &tr, int n_leaves) {
    ofstream shares_out(prefix + "ProportionEstimates" + to_string(k) + ".txt", ios::app | ios::out);
    ofstream pos_out(prefix + "BranchEstimate" + to_string(k) + ".txt", ios::app | ios::out);
    shares_out << "Source\tChain\tMean Proportion Estimate\t5% CI\tMedian Proportion Estimate\t95% CI\tEffective Sample Size\tAutocorrelation\tVariance\n";
    pos_out << "Source\tChain\tMean Branch Position\t5% CI\tMedian Branch Position\t95% CI\tEffective Sample Size\tAutocorrelation\tVariance\tEffective Sample Size for the source estimation\n";
    ChainSummary out;
    out.best = kept.at(0).loglik;
    for (int s = 0; s < k; ++s) {
        vector<long double> share, pos, spread;
        string last_branch;
        for (const State &st : kept) {
            if (st.loglik > out.best) out.best = st.loglik;
            const Placement &p = st.src[s];
            last_branch = p.at->longname;
            out.by_branch[last_branch]; /* created empty on first sight */
            share.emplace_back(st.shares[s]);
            pos.emplace_back(p.frac);
            const double from_bottom = p.at->dist * p.frac;
            spread.emplace_back(leaf_profile_distance(tr, p.at, n_leaves, p.at->dist - from_bottom));
        }
        const long double share_mean = series_mean(share), pos_mean = series_mean(pos);
        const long double share_ac = series_autocorr(share, 1);
        const long double share_ess = series_ess(share);
        const long double share_var = series_variance(share, share_mean);
        const long double pos_ac = series_autocorr(pos, 1);
        const long double pos_ess = series_ess(pos);
        const long double spread_ess = series_ess(spread);
        const long double pos_var = series_variance(pos, pos_mean);
        sort(pos.begin(), pos.end());
        sort(share.begin(), share.end());
        shares_out << last_branch << '\t' << chain << '\t' << share_mean << '\t' << sorted_quantile(share, 0.05) << '\t' << sorted_quantile(share, 0.5) << '\t'
                   << sorted_quantile(share, 0.95) << '\t' << share_ess << '\t' << share_ac << '\t' << share_var << '\n';
        pos_out << last_branch << '\t' << chain << '\t' << pos_mean << '\t' << sorted_quantile(pos, 0.05) << '\t' << sorted_quantile(pos, 0.5) << '\t'
                << sorted_quantile(pos, 0.95) << '\t' << pos_ess << '\t' << pos_ac << '\t' << pos_var << '\t' << spread_ess << '\n';
        out.by_branch[last_branch].push_back({(double)share_mean, (double)share_var, (double)pos_mean, (double)pos_var});
    }
    return out;
}

vector<int> random_start_nodes(Run &R, int n_nodes, int k) { /* soibean::generateRandomNumbers (soibean.cpp:157-172) */
    mt19937 gen(R.rd());
    uniform_int_distribution<> any(0, n_nodes - 1);
    vector<int> v;
    while ((int)v.size() < k) v.push_back(any(gen));
    return v;
}

double gelman_rubin(const vector<double> &means, const vector<double> &vars, int chain_len) { /* soibean::calculateRhat (:174-202) */
    const int m = means.size();
    if (m < 2) return -1;
    double within = 0.0, centre = 0.0;
    for (double v : vars) within += v;
    within /= m;
    for (double x : means) centre += x;
    centre /= m;
    double between = 0.0;
    for (int i = 0; i < m; ++i) between += pow(means[i] - centre, 2);
    between *= chain_len / (m - 1); /* integer quotient */
    const double pooled = ((chain_len - 1.0) * within + between) / chain_len;
    return sqrt(pooled / within);
}

} // namespace

extern "C" int orc_sb_estimate(const void *h, const char *newick, const char *path_names, const int32_t *sig_nodes, int32_t n_sig,
                               const orc_sb_estimate_cfg *cfg, const char *prefix) {
    try {
        Tree tr;
        read_newick(newick, tr);
        Run R(h, cfg->seed);
        {
            istringstream names(path_names);
            string name;
            for (int idx = 0; getline(names, name); ++idx) R.path_index[name] = idx;
        }
        int n_leaves = 0;
        for (Node *n : tr.nodes) n_leaves += n->isLeaf() ? 1 : 0;
        ChainCfg cc{cfg->max_iter, cfg->burn, cfg->con, cfg->freqs7, (int)R.path_index.size()};
        vector<int32_t> sig_paths;
        for (int32_t i = 0; i < n_sig; ++i) sig_paths.push_back(R.path_index.at(tr.nodes.at(sig_nodes[i])->longname));
        const string out = prefix;
        for (int32_t k = 1; k <= n_sig; ++k) { /* soibean.cpp:738-944: the first k starting nodes as sources */
            /* initial log-likelihood: a lone source is the plain sum, otherwise an equal-weight mixture with weight 1 / #starting nodes */
            const double start_ll = n_sig == 1 ? orc_sb_mixture_loglike(h, 1, &sig_paths[k - 1], 0.0)
                                               : orc_sb_mixture_loglike(h, k, sig_paths.data(), log(1.0 / n_sig));
            if (!cfg->run_mcmc) continue;
            ofstream diag(out + "Diagnostics" + to_string(k) + "0.txt");
            diag << "Source\tHighest log-likelihood\tfor chain\tRhat for the proportion estimate\tRhat for the branch position estimate" << endl;
            map<string, vector<vector<vector<double>>>> seen; /* branch -> chain -> entries (definition: one slot per chain) */
            vector<double> best_of_chain;
            vector<int> start(sig_nodes, sig_nodes + k);
            for (unsigned chain = 0; chain < cfg->chains; ++chain) {
                if (chain != 0) start = random_start_nodes(R, cc.n_paths, k);
                const vector<State> kept = run_chain(R, tr, start, start_ll, cc, out, chain);
                const ChainSummary sum = summarise(kept, k, out, chain, tr, n_leaves);
                best_of_chain.push_back(sum.best);
                for (const auto &b : sum.by_branch) {
                    seen[b.first].resize(cfg->chains);
                    seen[b.first][chain] = b.second;
                }
            }
            const int chain_len = cfg->max_iter - cfg->burn;
            int winner = 0;
            for (size_t c = 0; c < best_of_chain.size(); ++c)
                if (best_of_chain[c] > best_of_chain[winner]) winner = c;
            for (const auto &b : seen) {
                vector<double> sm, sv, pm, pv;
                for (unsigned c = 0; c < cfg->chains; ++c) {
                    vector<double> row = (b.second[c].empty() || b.second[c][0].size() < 4) ? vector<double>{1.0, 1.0, 1.0, 1.0} : b.second[c][0];
                    sm.push_back(row[0]);
                    sv.push_back(row[1]);
                    pm.push_back(row[2]);
                    pv.push_back(row[3]);
                }
                diag << b.first << '\t' << best_of_chain[winner] << '\t' << winner << '\t' << gelman_rubin(sm, sv, chain_len) << '\t' << gelman_rubin(pm, pv, chain_len)
                     << std::endl;
            }
        }
        return 0;
    } catch (const std::exception &e) {
        cerr << "orc_sb_estimate: " << e.what() << endl;
        return -1;
    }
}
