// soibean downstream of analyse_GAM (host control flow): the taxon tree, the tree-placement / proportion
// chain around the likelihood refresh, its summaries and diagnostics, and the files `vgan soibean` writes.
// Reference: src/MCMC.cpp:23-150 (processMCMCiterations), :169-470 (updatePosition), :487-520 (sample_normal), :522-1093
// (run_tree_proportion), src/MCMC.h:424-505,507-625 (state initialisation, quantiles, patristic distances),
// src/miscfunc.h:12-66 (mean / variance / autocorrelation / effective sample size), src/soibean.cpp:157-202,738-944.
//
// The likelihood itself is not computed here: the chain calls a vgan_sb_engine (vgan_sb_engine_gpu in sb_capi.hip binds the
// device context); this file has no device dependency.
//
// Definitions where the reference leaves the behaviour open (all documented in include/vgan_gpu.h):
//  * randomness: every std::random_device call is replaced by the next output of the caller's seed stream (seed 0 = the
//    hardware source as the reference); libc rand() (unseeded there, so one fixed sequence per process) and the function-local
//    static generator of sample_normal by one mt19937 each per chain, seeded from that stream when the chain starts -- the
//    chains are then independent of each other and are advanced together, one likelihood call per iteration for all of them;
//  * tree node numbering (spidir is not part of the reference tree): pre-order of the Newick text;
//  * getPatristicDistances writes distances[node index] into a vector of length #leaves: only indices below #leaves
//    are ever compared (calculateEuclideanDistance stops at the shorter vector), which is what is computed here;
//  * the per-branch diagnostics iterate an unordered_map: here branches are reported in name order, and a branch a chain
//    never ended on takes the reference's own "empty" default {1, 1, 1, 1} for that chain.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <limits>
#include <map>
#include <memory>
#include <numeric>
#include <random>
#include <sstream>
#include <thread>

#include <zlib.h>

using namespace vgan;

struct vgan_tree {
    std::vector<int32_t> parent, child_off{0}, children;
    std::vector<double> dist;
    std::vector<std::string> name;
    std::string names_joined;
    uint32_t n_leaves = 0;
    int32_t root = -1;
    uint32_t n() const { return (uint32_t)parent.size(); }
    bool leaf(int32_t v) const { return child_off[(size_t)v + 1] == child_off[(size_t)v]; }
    int32_t n_children(int32_t v) const { return child_off[(size_t)v + 1] - child_off[(size_t)v]; }
    int32_t child(int32_t v, int32_t i) const { return children[(size_t)child_off[(size_t)v] + (size_t)i]; }
};

namespace {

// ------------------------------------------------------------------------------------------------ Newick
struct NewickParser {
    const std::string &s;
    size_t p = 0;
    std::vector<int32_t> parent;
    std::vector<std::vector<int32_t>> kids;
    std::vector<double> dist;
    std::vector<std::string> name;
    explicit NewickParser(const std::string &text) : s(text) {}
    void skip_ws() {
        while (p < s.size() && isspace((unsigned char)s[p])) ++p;
    }
    int depth = 0;
    int32_t node(int32_t up) {
        struct Level { // the parser recurses per nesting level: a text of a million '(' must not run the stack out
            int &d;
            explicit Level(int &x) : d(x) {
                if (++d > 10000) throw std::runtime_error("Newick: nesting deeper than 10000 levels");
            }
            ~Level() { --d; }
        } level(depth);
        const int32_t id = (int32_t)parent.size();
        parent.push_back(up);
        kids.emplace_back();
        dist.push_back(0.0);
        name.emplace_back();
        skip_ws();
        if (p < s.size() && s[p] == '(') {
            ++p;
            for (;;) {
                const int32_t c = node(id);
                kids[(size_t)id].push_back(c);
                skip_ws();
                if (p < s.size() && s[p] == ',') {
                    ++p;
                    continue;
                }
                if (p < s.size() && s[p] == ')') {
                    ++p;
                    break;
                }
                throw std::runtime_error("Newick: expected ',' or ')' at offset " + std::to_string(p));
            }
        }
        skip_ws();
        const size_t b = p;
        while (p < s.size() && !strchr(",():;", s[p]) && !isspace((unsigned char)s[p])) ++p;
        name[(size_t)id] = s.substr(b, p - b);
        skip_ws();
        if (p < s.size() && s[p] == ':') {
            ++p;
            skip_ws();
            char *end = nullptr;
            dist[(size_t)id] = strtod(s.c_str() + p, &end);
            if (end == s.c_str() + p) throw std::runtime_error("Newick: branch length expected at offset " + std::to_string(p));
            p = (size_t)(end - s.c_str());
        }
        return id;
    }
};

// ------------------------------------------------------------------------------------------------ randomness
class SeedStream {
  public:
    explicit SeedStream(uint64_t seed) : state_(seed), hardware_(seed == 0) {}
    uint32_t next() {
        if (hardware_) return std::random_device{}();
        state_ += 0x9E3779B97F4A7C15ull;
        uint64_t z = state_;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return (uint32_t)((z ^ (z >> 31)) >> 32);
    }

  private:
    uint64_t state_;
    bool hardware_;
};

struct Position { // PosTree
    int32_t node;
    double pos_branch, theta;
};

struct ChainState { // MCMCiteration
    std::vector<Position> at;
    std::vector<double> proportions;
    double log_like = 0.0;
};

class GzLines {
  public:
    // level 1: the trace holds a line per chain and iteration, and deflate at the default level costs more than the GPU call
    explicit GzLines(const std::string &path) : f_(gzopen(path.c_str(), "wb1")), path_(path) {
        if (!f_) throw std::runtime_error("cannot write " + path);
    }
    ~GzLines() {
        if (f_) gzclose(f_);
    }
    void put(const std::string &s) {
        if (!s.empty() && gzwrite(f_, s.data(), (unsigned)s.size()) <= 0) throw std::runtime_error("write error on " + path_);
    }

  private:
    gzFile f_;
    std::string path_;
};

// ------------------------------------------------------------------------------------------------ statistics (miscfunc.h)
long double mean_of(const std::vector<long double> &v) {
    double acc = 0.0; // std::accumulate with a double seed: the running sum is a double
    for (long double x : v) acc = (double)(acc + x);
    return acc / v.size();
}

long double variance_of(const std::vector<long double> &v, long double m) {
    long double s = 0.0;
    for (long double x : v) s += (x - m) * (x - m);
    return s / (v.size() - 1);
}

// autocorrelation(v, lag) of miscfunc.h recomputes mean and variance of v on every call; they do not depend on the lag, so
// the effective-sample-size loop (hundreds of lags on a slowly mixing chain) passes them in -- same values, same arithmetic
long double autocorr(const std::vector<long double> &v, int lag, long double m, long double denom) {
    double numer = 0.0;
    for (size_t i = 0; i + (size_t)lag < v.size(); ++i) numer = (double)(numer + (v[i] - m) * (v[i + (size_t)lag] - m));
    return numer / ((v.size() - (size_t)lag) * denom);
}

long double autocorr(const std::vector<long double> &v, int lag) {
    const long double m = mean_of(v);
    return autocorr(v, lag, m, variance_of(v, m));
}

// miscfunc.h effectiveSampleSize: 1 + 2 * (sum of autocorrelations, taken in pairs until a pair is no longer positive).  A
// slowly mixing chain needs thousands of lags of O(n) each -- minutes at the default 425 000 recorded states -- and the lags
// are independent of each other: they are evaluated a block at a time on several threads and then consumed in order with the
// reference's stopping rule, so the result is the same number.
double effective_sample_size(const std::vector<long double> &v) {
    const long double m = mean_of(v), denom = variance_of(v, m);
    const int max_lag = (int)(v.size() / 2);
    double even = 1.0, odd = (double)autocorr(v, 1, m, denom), total = even + odd;
    const unsigned hw = std::max(1u, std::min(16u, usable_cpus()));
    int t = 1;
    int block = 8; // lags per round; the first rounds stay small and serial: a well mixed chain stops within a few lags
    std::vector<double> ac;
    while (t < max_lag - 2 && even + odd > 0) {
        const int n_lags = std::min(block, 2 * ((max_lag - 2 - t + 1) / 2)); // whole pairs only: lags t+1 .. t+n_lags
        if (n_lags <= 0) break;
        ac.assign((size_t)n_lags, 0.0);
        const unsigned threads = (size_t)n_lags * v.size() < (1u << 18) ? 1u : std::min<unsigned>(hw, (unsigned)n_lags);
        auto work = [&](unsigned tid) {
            for (int j = (int)tid; j < n_lags; j += (int)threads) ac[(size_t)j] = (double)autocorr(v, t + 1 + j, m, denom);
        };
        if (threads == 1) work(0);
        else {
            std::vector<std::thread> pool;
            for (unsigned i = 1; i < threads; ++i) pool.emplace_back(work, i);
            work(0);
            for (auto &th : pool) th.join();
        }
        for (int j = 0; j + 1 < n_lags && t < max_lag - 2 && even + odd > 0; j += 2) {
            even = ac[(size_t)j];
            odd = ac[(size_t)j + 1];
            total += 2.0 * (even + odd);
            t += 2;
        }
        block = std::min(block * 4, 1024);
    }
    if (even + odd < 0) total -= even + odd;
    return v.size() / (1 + total);
}

long double quantile2(const std::vector<long double> &sorted, double q) { // MCMC.h getQuantile2
    const double index = (double)(sorted.size() - 1) * q;
    const size_t lo = (size_t)std::floor(index), hi = (size_t)std::ceil(index);
    if (lo == hi) return sorted[lo];
    const double frac = index - (double)lo;
    return (1.0 - frac) * sorted[lo] + frac * sorted[hi];
}

double rhat(const std::vector<double> &means, const std::vector<double> &variances, int chain_length) { // soibean.cpp:174-202
    const int m = (int)means.size();
    if (m < 2) return -1;
    const double W = std::accumulate(variances.begin(), variances.end(), 0.0) / m;
    const double grand = std::accumulate(means.begin(), means.end(), 0.0) / m;
    double B = 0.0;
    for (double x : means) B += std::pow(x - grand, 2);
    B *= chain_length / (m - 1); // integer division, as there
    const double var_est = ((chain_length - 1.0) * W + B) / chain_length;
    return std::sqrt(var_est / W);
}

// ------------------------------------------------------------------------------------------------ the run
struct ChainSummary { // processMCMCiterations' return value
    std::map<std::string, std::vector<std::vector<double>>> per_branch;
    double best_log_like = 0.0;
};

class Estimator {
  public:
    Estimator(const vgan_sb_engine &e, const vgan_tree &t, const int32_t *node_path, const vgan_sb_estimate_cfg &cfg, std::string prefix)
        : eng_(e), tree_(t), node_path_(node_path), cfg_(cfg), out_(std::move(prefix)), seeds_(cfg.seed) {}

    // soibean.cpp:738-944 for one list of starting nodes
    void run(const int32_t *sig_nodes, uint32_t n_sig) {
        for (uint32_t i = 0; i < n_sig; ++i) {
            std::vector<int32_t> sources(sig_nodes, sig_nodes + i + 1);
            // initial log-likelihood (:742-760): source count 1 in total -> plain sum, else the equal-weight mixture of the first i + 1
            std::vector<int32_t> paths;
            for (uint32_t j = 0; j <= i; ++j) paths.push_back(path_of(sig_nodes[j]));
            double ll = 0.0;
            int rc;
            if (n_sig == 1) rc = eng_.mixture(eng_.user, 1, &paths[i], 0.0, &ll);
            else rc = eng_.mixture(eng_.user, i + 1, paths.data(), std::log(1.0 / n_sig), &ll);
            if (rc != VGAN_OK) throw std::runtime_error(std::string("initial log-likelihood: ") + vgan_last_error());
            if (!cfg_.quiet) fprintf(stderr, "Initial log-likelihood: %g\n", ll);
            if (!cfg_.run_mcmc) continue;
            const uint32_t k = i + 1;
            std::ofstream diag((out_ + "Diagnostics" + std::to_string(k) + "0.txt").c_str());
            diag << "Source\tHighest log-likelihood\tfor chain\tRhat for the proportion estimate\tRhat for the branch position estimate" << std::endl;
            std::map<std::string, std::vector<std::vector<std::vector<double>>>> by_branch; // branch -> chain -> entries
            std::vector<double> chain_best;
            // every chain owns its generators, so the chains are independent of each other and advance together: one
            // likelihood call per iteration covers all of them (one launch on the GPU)
            std::vector<std::unique_ptr<Chain>> chains;
            for (uint32_t chain = 0; chain < cfg_.chains; ++chain) {
                if (chain != 0) sources = random_nodes(k);
                chains.push_back(start_chain(sources, ll, chain));
            }
            if (!cfg_.quiet) fprintf(stderr, "Running %u chains of %u iterations\n", cfg_.chains, cfg_.max_iter);
            PhaseTimer pt("sb_estimate");
            advance_together(chains, k);
            pt.lap("chains");
            for (uint32_t chain = 0; chain < cfg_.chains; ++chain) {
                const std::vector<ChainState> &kept = chains[chain]->kept;
                const ChainSummary sum = summarise(kept, (int)k, (int)chain);
                chains[chain].reset(); // closes its files, frees the recorded states
                chain_best.push_back(sum.best_log_like);
                for (const auto &b : sum.per_branch) {
                    auto &slot = by_branch[b.first];
                    slot.resize(cfg_.chains);
                    slot[chain] = b.second;
                }
            }
            pt.lap("summaries");
            const int chain_length = (int)cfg_.max_iter - (int)cfg_.burn;
            size_t best = 0;
            for (size_t h = 0; h < chain_best.size(); ++h)
                if (chain_best[h] > chain_best[best]) best = h;
            for (const auto &b : by_branch) {
                std::vector<double> pm(cfg_.chains, 1.0), pv(cfg_.chains, 1.0), qm(cfg_.chains, 1.0), qv(cfg_.chains, 1.0);
                for (uint32_t c = 0; c < cfg_.chains; ++c) {
                    if (b.second[c].empty() || b.second[c][0].size() < 4) continue; // the reference's defaults of 1.0
                    pm[c] = b.second[c][0][0];
                    pv[c] = b.second[c][0][1];
                    qm[c] = b.second[c][0][2];
                    qv[c] = b.second[c][0][3];
                }
                diag << b.first << '\t' << chain_best[best] << '\t' << best << '\t' << rhat(pm, pv, chain_length) << '\t' << rhat(qm, qv, chain_length)
                     << std::endl;
            }
        }
    }

  private:
    int32_t path_of(int32_t node) const {
        if (node < 0 || (uint32_t)node >= tree_.n() || node_path_[node] < 0) throw std::runtime_error("tree node without a graph path");
        return node_path_[node];
    }

    std::vector<int32_t> random_nodes(uint32_t k) { // soibean::generateRandomNumbers
        std::mt19937 gen(seeds_.next());
        std::uniform_int_distribution<> pick(0, (int)cfg_.n_paths - 1);
        std::vector<int32_t> v;
        for (uint32_t i = 0; i < k; ++i) v.push_back(pick(gen));
        return v;
    }

    static uint32_t pick(std::mt19937 &walk, uint32_t n) { return walk() % n; } // rand() % n

    // MCMC::updatePosition: walk `distance` along the tree from p, forwards (towards the leaves) or backwards.  Positions are
    // fractions of a branch; every branch counts as length 1 for the walk.
    void move(std::mt19937 &walk, Position &p, double distance, bool forward) const {
        if (p.pos_branch < 0.0 || p.pos_branch > 1.0) throw std::runtime_error("Error: Initial pos_branch is out of valid range.");
        if (distance < 0.0) throw std::runtime_error("Error: move distance cannot be negative.");
        double left = std::abs(distance);
        while (left > 0.0) {
            if (forward) {
                if (p.pos_branch + left < 1.0) {
                    p.pos_branch += left;
                    left = 0.0;
                } else if (tree_.leaf(p.node)) {
                    forward = false; // bounce off the leaf
                } else {
                    const double rest = std::max(0.0, left - (1.0 - p.pos_branch));
                    p.node = tree_.child(p.node, (int32_t)pick(walk, (uint32_t)tree_.n_children(p.node)));
                    check_branch(p.node, "Error: next branch length cannot be negative.");
                    if (rest > 1.0) {
                        p.pos_branch = 1.0;
                        left = rest - 1.0;
                    } else {
                        p.pos_branch = rest;
                        left = 0.0;
                    }
                }
                continue;
            }
            if (p.pos_branch - left > 0.0) {
                p.pos_branch -= left;
                left = 0.0;
                continue;
            }
            const int32_t up = tree_.parent[(size_t)p.node];
            if (up < 0) { // the root: turn round into one of its children, position and remaining distance unchanged
                forward = true;
                p.node = tree_.child(p.node, (int32_t)pick(walk, (uint32_t)tree_.n_children(p.node)));
                check_branch(p.node, "Error: next branch length cannot be negative.");
                continue;
            }
            // the parent, or -- only from an inner node -- one of the siblings
            std::vector<int32_t> options{up};
            if (!tree_.leaf(p.node))
                for (int32_t i = 0; i < tree_.n_children(up); ++i)
                    if (tree_.child(up, i) != p.node) options.push_back(tree_.child(up, i));
            const int32_t chosen = options[pick(walk, (uint32_t)options.size())];
            if (chosen == up) {
                const double rest = std::max(0.0, left - p.pos_branch);
                p.node = up;
                check_branch(p.node, "Error: parent branch length cannot be negative.");
                if (rest > 1.0) {
                    p.pos_branch = 0.0;
                    left = rest - 1.0;
                } else {
                    const double np = 1.0 - rest;
                    if (np <= 0.0 || np >= 1.0) throw std::runtime_error("Error: new position branch is not in the valid range.");
                    p.pos_branch = np;
                    left = 0.0;
                }
                continue;
            }
            // into the sibling, downwards from its top
            forward = true;
            p.node = chosen;
            const double rest = left - p.pos_branch;
            p.pos_branch = 0.0;
            if (rest < 1.0) {
                p.pos_branch = rest;
                if (p.pos_branch < 0.0 || p.pos_branch > 1.0) throw std::runtime_error("Error: pos_branch is out of valid range after increment.");
                left = 0.0;
            } else if (tree_.leaf(p.node)) {
                forward = false; // `left` is kept: the walk turns round at the sibling's top
            } else {
                const double rest2 = std::max(0.0, left - (1.0 - p.pos_branch));
                p.node = tree_.child(p.node, (int32_t)pick(walk, (uint32_t)tree_.n_children(p.node)));
                check_branch(p.node, "Error: next branch length cannot be negative.");
                if (rest2 > 1.0) {
                    p.pos_branch = 1.0;
                    left = rest2 - 1.0;
                } // else: position 0 on the child and the distance stays as it is
            }
        }
        if (p.pos_branch < 0.0 || p.pos_branch > 1.0) throw std::runtime_error("Error: pos_branch is out of valid range after movement.");
    }

    void check_branch(int32_t node, const char *msg) const {
        if (tree_.dist[(size_t)node] < 0.0) throw std::runtime_error(msg);
    }

    static std::vector<double> sample_thetas(std::mt19937 &theta_engine_, const std::vector<double> &x) { // MCMC::sample_normal
        std::vector<double> r;
        long double sum = 0.0L;
        for (double xi : x) {
            std::normal_distribution<double> d(xi, 0.1);
            double s;
            do s = d(theta_engine_);
            while (s < 0.0L || s > 1.0L);
            r.push_back(s);
            sum += s;
        }
        for (double &v : r) v /= sum;
        return r;
    }

    void sources_of(const ChainState &st, vgan_sb_source *src) const {
        for (size_t y = 0; y < st.at.size(); ++y) {
            const int32_t node = st.at[y].node, up = tree_.parent[(size_t)node];
            src[y].child = path_of(node);
            src[y].parent = path_of(up < 0 ? node : up);
            src[y].dist = tree_.dist[(size_t)node];
            src[y].pos = st.at[y].pos_branch;
            src[y].theta = st.proportions[y];
        }
    }

    // the log-likelihoods of one proposed state per chain
    void refresh(std::vector<ChainState> &props, uint32_t k) {
        const uint32_t n = (uint32_t)props.size();
        std::vector<vgan_sb_source> src((size_t)n * k);
        for (uint32_t c = 0; c < n; ++c) sources_of(props[c], src.data() + (size_t)c * k);
        std::vector<double> ll(n, 0.0);
        std::vector<uint64_t> guard(n, 0);
        int rc = VGAN_OK;
        if (eng_.refresh_many && n > 1) rc = eng_.refresh_many(eng_.user, n, k, src.data(), cfg_.con, cfg_.freqs7, ll.data(), guard.data());
        else
            for (uint32_t c = 0; c < n && rc == VGAN_OK; ++c)
                rc = eng_.refresh(eng_.user, k, src.data() + (size_t)c * k, cfg_.con, cfg_.freqs7, &ll[c], &guard[c]);
        if (rc != VGAN_OK) throw std::runtime_error(std::string("likelihood refresh: ") + vgan_last_error());
        for (uint32_t c = 0; c < n; ++c) {
            if (guard[c]) throw std::runtime_error("Problem in the likelihood compuation! Intermediate log likelihood is -nan, -inf or positive.");
            props[c].log_like = ll[c];
        }
    }

    // "<branch>\t<logLike>\t<theta>\t<pos_branch>\t[<verdict>\t]" per source, numbers as an ostream prints them under
    // setprecision(14) (= %.14g); formatted by hand because two such lines per chain and iteration are a visible share of
    // an iteration once the likelihood takes 25 us
    static std::string line_of(const vgan_tree &t, const ChainState &s, double ll, const char *verdict) {
        std::string o;
        char num[96];
        for (const Position &p : s.at) {
            o += t.name[(size_t)p.node];
            const int n = snprintf(num, sizeof num, "\t%.14g\t%.14g\t%.14g\t", ll, p.theta, p.pos_branch);
            o.append(num, (size_t)n);
            if (verdict) {
                o += verdict;
                o += '\t';
            }
        }
        o += '\n';
        return o;
    }

    // One chain of MCMC::run_tree_proportion: its generators (the proposal / acceptance engine, the one standing in for rand()
    // in the tree walk, the one standing in for sample_normal's static engine), its state, its two output files
    struct Chain {
        std::mt19937 walk, theta, gen;
        ChainState cur;
        std::unique_ptr<GzLines> result, trace;
        std::vector<ChainState> kept;
    };

    std::unique_ptr<Chain> start_chain(const std::vector<int32_t> &sources, double start_ll, uint32_t chain) {
        const uint32_t k = (uint32_t)sources.size();
        if (cfg_.burn >= cfg_.max_iter) throw std::runtime_error("Number of brun in iteration exceedes the number of total iterations. Exiting. ");
        auto c = std::make_unique<Chain>();
        c->walk.seed(seeds_.next());
        c->theta.seed(seeds_.next());
        c->gen.seed(seeds_.next());
        { // initializeState: normalised uniform thetas from an engine of their own, every source in the middle of its branch
            std::mt19937 g0(seeds_.next());
            std::uniform_real_distribution<> u01(0.0, 1.0);
            std::vector<double> th(k);
            double sum = 0.0;
            for (double &x : th) sum += (x = u01(g0));
            for (uint32_t i = 0; i < k; ++i) {
                if (sources[i] < 0 || (uint32_t)sources[i] >= tree_.n()) throw std::runtime_error("source node outside the tree");
                c->cur.at.push_back({sources[i], 0.5, th[i] / sum});
                c->cur.proportions.push_back(std::max(0.001, th[i] / sum));
            }
            c->cur.log_like = start_ll;
        }
        const std::string tag = std::to_string(k) + std::to_string(chain);
        c->result = std::make_unique<GzLines>(out_ + "Result" + tag + ".mcmc");
        c->trace = std::make_unique<GzLines>(out_ + "Trace" + tag + ".detail.mcmc");
        std::ostringstream h1, h2;
        for (uint32_t s = 1; s <= k; ++s) {
            h1 << "Source_" << s << "\tLog-likelihood\tproportion\tbranch_position_derived\t";
            h2 << "Source_" << s << "\tLog-likelihood\tproportion_" << s << "\tbranch_position_derived_" << s << "\tMove\t";
        }
        c->result->put(h1.str() + "\n");
        c->trace->put(h2.str() + "\n");
        return c;
    }

    void advance_together(std::vector<std::unique_ptr<Chain>> &chains, uint32_t k) {
        const uint32_t burn = cfg_.burn, max_iter = cfg_.max_iter;
        const double init_sd = cfg_.n_paths <= 30 ? 3.0 : cfg_.n_paths * (3.0 / 30.0);
        const double step = (init_sd - 0.1) / std::max(1u, burn - 1u), step2 = (0.1 - 1e-5) / std::max(1u, (max_iter - burn) - 1u);
        std::uniform_real_distribution<> unit(0.0, 1.0);
        std::vector<ChainState> props(chains.size());
        for (uint32_t it = 0; it <= max_iter; ++it) {
            double sd;
            if (it < burn) sd = std::max(1e-5, init_sd - it * step);
            else if (it % 100000u == 0) sd = 1;
            else sd = std::max(1e-5, 0.1 - (it - burn) * step2);
            for (size_t c = 0; c < chains.size(); ++c) {
                Chain &ch = *chains[c];
                ChainState &prop = props[c];
                prop = ch.cur;
                if (it != 0)
                    for (Position &p : prop.at) {
                        std::normal_distribution<double> jump(0, sd);
                        const double d = jump(ch.gen);
                        if (d < 0.0) move(ch.walk, p, -d, false);
                        else move(ch.walk, p, d, true);
                    }
                std::vector<double> th;
                for (const Position &p : prop.at) th.push_back(p.theta);
                th = sample_thetas(ch.theta, th);
                for (uint32_t i = 0; i < k; ++i) prop.at[i].theta = th[i];
                prop.proportions = th;
            }
            refresh(props, k);
            for (size_t c = 0; c < chains.size(); ++c) {
                Chain &ch = *chains[c];
                const ChainState &prop = props[c];
                const double delta = prop.log_like - ch.cur.log_like;
                const double accept = delta > 0 ? 1.0 : std::exp(delta);
                const double u = unit(ch.gen);
                const bool take = u <= accept || it == 0;
                ch.trace->put(line_of(tree_, prop, prop.log_like, take ? "accepted" : "rejected"));
                if (it > burn) { // the state the chain is leaving / staying in is what gets recorded
                    ch.result->put(line_of(tree_, ch.cur, ch.cur.log_like, nullptr));
                    ch.kept.push_back(ch.cur);
                }
                if (take) ch.cur = prop;
            }
        }
    }

    // distance from `node`, `below_top` down its branch... see getPatristicDistances: to every leaf whose node index is below #leaves
    long double leaf_distance_norm(int32_t node, double posonbranch) const {
        long double sum = 0.0L;
        const uint32_t n_cmp = std::min<uint32_t>(tree_.n_leaves, tree_.n());
        std::vector<char> on_path(tree_.n(), 0);
        for (int32_t v = node; v >= 0; v = tree_.parent[(size_t)v]) on_path[(size_t)v] = 1;
        for (uint32_t leaf = 0; leaf < n_cmp; ++leaf) {
            if (!tree_.leaf((int32_t)leaf)) continue;
            int32_t lca = (int32_t)leaf;
            double from_leaf = 0.0;
            while (!on_path[(size_t)lca]) {
                from_leaf += tree_.dist[(size_t)lca];
                lca = tree_.parent[(size_t)lca];
            }
            double from_node = 0.0;
            for (int32_t v = node; v != lca; v = tree_.parent[(size_t)v]) from_node += tree_.dist[(size_t)v];
            from_node -= posonbranch;
            if (!(from_node >= 0.0 && from_leaf >= 0.0)) continue; // stays numeric_limits::max() there: skipped
            const double diff = (from_node + from_leaf) - 1.0;     // against initialPatristicDistances = 1.0
            sum += diff * diff;
        }
        return std::sqrt(sum);
    }

    // MCMC::processMCMCiterations
    ChainSummary summarise(const std::vector<ChainState> &kept, int k, int chain) {
        ChainSummary out;
        std::ofstream est((out_ + "ProportionEstimates" + std::to_string(k) + ".txt").c_str(), std::ios::app | std::ios::out);
        std::ofstream bra((out_ + "BranchEstimate" + std::to_string(k) + ".txt").c_str(), std::ios::app | std::ios::out);
        est << "Source\tChain\tMean Proportion Estimate\t5% CI\tMedian Proportion Estimate\t95% CI\tEffective Sample Size\tAutocorrelation\tVariance\n";
        bra << "Source\tChain\tMean Branch Position\t5% CI\tMedian Branch Position\t95% CI\tEffective Sample Size\tAutocorrelation\tVariance\tEffective "
               "Sample Size for the source estimation\n";
        out.best_log_like = kept.at(0).log_like;
        for (int s = 0; s < k; ++s) {
            std::vector<long double> prop, pos, euc;
            std::string branch;
            for (const ChainState &st : kept) {
                out.best_log_like = std::max(out.best_log_like, st.log_like);
                const Position &p = st.at[(size_t)s];
                branch = tree_.name[(size_t)p.node];
                out.per_branch[branch]; // every branch the source visited gets an (empty) entry
                prop.push_back(st.proportions[(size_t)s]);
                pos.push_back(p.pos_branch);
                const double d = tree_.dist[(size_t)p.node];
                euc.push_back(leaf_distance_norm(p.node, d - d * p.pos_branch));
            }
            const long double m_prop = mean_of(prop), m_pos = mean_of(pos);
            const long double ac_prop = autocorr(prop, 1), ess_prop = effective_sample_size(prop), var_prop = variance_of(prop, m_prop);
            const long double ac_pos = autocorr(pos, 1), ess_pos = effective_sample_size(pos);
            const long double ess_euc = effective_sample_size(euc), var_pos = variance_of(pos, m_pos);
            if (!cfg_.quiet) {
                if (ess_prop < 100) fprintf(stderr, "Warning: The effective sample size for the proportion estimation of chain %d is below 100. The estimation of the proportion for the branch %s can not be ensured. A rerun using a higher number of iterations is recommended.\n", chain, branch.c_str());
                if (ess_pos < 100) fprintf(stderr, "Warning: The effective sample size for the estimation of the branch position for chain %d is below 100. The estimation of the position for the branch %s can not be ensured. A rerun using a higher number of iterations is recommended.\n", chain, branch.c_str());
                if (ess_euc < 100) fprintf(stderr, "Warning: The effective sample size for the estimation of the branch for chain %d is below 100. The estimation of the branch %s as a source can not be ensured.\n", chain, branch.c_str());
            }
            std::sort(pos.begin(), pos.end());
            std::sort(prop.begin(), prop.end());
            est << branch << '\t' << chain << '\t' << m_prop << '\t' << quantile2(prop, 0.05) << '\t' << quantile2(prop, 0.5) << '\t' << quantile2(prop, 0.95)
                << '\t' << ess_prop << '\t' << ac_prop << '\t' << var_prop << '\n';
            bra << branch << '\t' << chain << '\t' << m_pos << '\t' << quantile2(pos, 0.05) << '\t' << quantile2(pos, 0.5) << '\t' << quantile2(pos, 0.95)
                << '\t' << ess_pos << '\t' << ac_pos << '\t' << var_pos << '\t' << ess_euc << '\n';
            out.per_branch[branch].push_back({(double)m_prop, (double)var_prop, (double)m_pos, (double)var_pos});
        }
        return out;
    }

    const vgan_sb_engine &eng_;
    const vgan_tree &tree_;
    const int32_t *node_path_;
    const vgan_sb_estimate_cfg &cfg_;
    std::string out_;
    SeedStream seeds_;
};

} // namespace

extern "C" int vgan_tree_parse(const char *newick, vgan_tree **out) {
    if (!newick || !out) return fail(VGAN_EINVAL, "vgan_tree_parse: null argument");
    try {
        const std::string text(newick);
        NewickParser ps(text);
        ps.skip_ws();
        if (ps.p >= text.size()) return fail(VGAN_EINVAL, "vgan_tree_parse: The tree is empty");
        const int32_t root = ps.node(-1);
        ps.skip_ws();
        if (ps.p >= text.size() || text[ps.p] != ';') return fail(VGAN_EINVAL, "vgan_tree_parse: ';' expected at offset %zu", ps.p);
        auto t = std::make_unique<vgan_tree>();
        t->root = root;
        t->parent = ps.parent;
        t->dist = ps.dist;
        t->name = ps.name;
        for (size_t v = 0; v < ps.kids.size(); ++v) {
            t->children.insert(t->children.end(), ps.kids[v].begin(), ps.kids[v].end());
            t->child_off.push_back((int32_t)t->children.size());
            if (ps.kids[v].empty()) ++t->n_leaves;
            t->names_joined += ps.name[v];
            t->names_joined += '\n';
        }
        *out = t.release();
        return VGAN_OK;
    } catch (const std::exception &e) {
        return fail(VGAN_EINVAL, "vgan_tree_parse: %s", e.what());
    }
}

extern "C" int vgan_tree_load(const char *path, vgan_tree **out) {
    if (!path || !out) return fail(VGAN_EINVAL, "vgan_tree_load: null argument");
    std::string text;
    if (!read_text_maybe_gz(path, text)) return fail(VGAN_EIO, "vgan_tree_load: cannot read %s", path);
    return vgan_tree_parse(text.c_str(), out);
}

extern "C" int vgan_tree_view_get(const vgan_tree *t, vgan_tree_view *out) {
    if (!t || !out) return fail(VGAN_EINVAL, "vgan_tree_view_get: null argument");
    out->n_nodes = t->n();
    out->n_leaves = t->n_leaves;
    out->root = t->root;
    out->parent = t->parent.data();
    out->dist = t->dist.data();
    out->child_off = t->child_off.data();
    out->children = t->children.data();
    out->names = t->names_joined.c_str();
    return VGAN_OK;
}

extern "C" void vgan_tree_free(vgan_tree *t) { delete t; }

extern "C" int vgan_sb_estimate(const vgan_sb_engine *engine, const vgan_tree *tree, const int32_t *node_path, const int32_t *sig_nodes,
                                uint32_t n_sig, const vgan_sb_estimate_cfg *cfg, const char *out_prefix) {
    if (!engine || !engine->refresh || !engine->mixture || !tree || !node_path || !sig_nodes || !cfg || !out_prefix)
        return fail(VGAN_EINVAL, "vgan_sb_estimate: null argument");
    if (n_sig == 0) return fail(VGAN_EINVAL, "vgan_sb_estimate: no starting node");
    if (cfg->n_paths == 0 || cfg->n_paths > tree->n()) return fail(VGAN_EINVAL, "vgan_sb_estimate: n_paths must be 1..%u (the tree's nodes)", tree->n());
    if (cfg->run_mcmc && (cfg->chains == 0 || cfg->burn >= cfg->max_iter))
        return fail(VGAN_EINVAL, "vgan_sb_estimate: The number of iterations must be higher than the burn-in period. Unable to proceed.");
    for (uint32_t i = 0; i < n_sig; ++i)
        if (sig_nodes[i] < 0 || (uint32_t)sig_nodes[i] >= tree->n()) return fail(VGAN_EINVAL, "vgan_sb_estimate: starting node outside the tree");
    try {
        Estimator(*engine, *tree, node_path, *cfg, out_prefix).run(sig_nodes, n_sig);
        return VGAN_OK;
    } catch (const std::exception &e) {
        return fail(VGAN_EINVAL, "vgan_sb_estimate: %s", e.what());
    }
}
