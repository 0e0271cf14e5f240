/*
 * oracle/sb_chain_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * What `vgan soibean` does after analyse_GAM, restated for the CPU with the reference's structures (node pointers, PosTree,
 * MCMCiteration) and loops (paths relative to /root/reference/src):
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
 * for std::random_device, one mt19937 for rand(), one for sample_normal's static engine; distances beyond #leaves ignored;
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

typedef struct PosTree {
    Node *pos;
    double pos_branch;
    double theta;
    double branch_place_anc;
    double branch_place_der;
} PosTree;

typedef struct MCMCiteration {
    int n_components;
    vector<double> proportions;
    vector<double> max_branch_lens;
    vector<PosTree> positions_tree;
    double logLike;
} MCMCiteration;

struct Params {
    Tree *tr;
    vector<int> sources;
    unsigned int maxIter, burn, chains;
    double logLike;
    const double *freqs7;
};

struct Run {
    const void *h;               /* orc_sb_analyse handle */
    map<string, int> path_index; /* longname -> path */
    Entropy rd;
    mt19937 rand_engine;   /* rand() */
    mt19937 theta_engine;  /* sample_normal's static generator */
    Run(const void *handle, uint64_t seed) : h(handle), rd(seed), rand_engine(rd()), theta_engine(rd()) {}
};
/* stands in for `rand() % n` (the definition of include/vgan_gpu.h: the run's mt19937, full 32-bit output, modulo n) */
inline unsigned pick(Run &r, unsigned n) { return r.rand_engine() % n; }

inline long double mean(const vector<long double> &v) { return accumulate(v.begin(), v.end(), 0.0) / v.size(); }
inline long double variance(const vector<long double> &v, long double mean) {
    long double sum = 0.0;
    for (const auto &i : v) {
        long double diff = i - mean;
        sum += diff * diff;
    }
    return sum / (v.size() - 1);
}
inline long double autocorrelation(const vector<long double> &v, int k) {
    long double m = mean(v);
    long double denom = variance(v, m);
    double numer = 0.0;
    for (size_t i = 0; i < v.size() - k; ++i) numer += ((v[i] - m) * (v[i + k] - m));
    return numer / ((v.size() - k) * denom);
}
inline double effectiveSampleSize(const vector<long double> &v) {
    int max_lag = v.size() / 2;
    double rho_hat_even = 1.0;
    double rho_hat_odd = autocorrelation(v, 1);
    double rho_hat_tot = rho_hat_even + rho_hat_odd;
    int t = 1;
    while ((t < max_lag - 2) && (rho_hat_even + rho_hat_odd > 0)) {
        rho_hat_even = autocorrelation(v, t + 1);
        rho_hat_odd = autocorrelation(v, t + 2);
        rho_hat_tot += 2.0 * (rho_hat_even + rho_hat_odd);
        t += 2;
    }
    if (rho_hat_even + rho_hat_odd < 0) rho_hat_tot -= (rho_hat_even + rho_hat_odd);
    return v.size() / (1 + rho_hat_tot);
}
long double getQuantile2(const vector<long double> &sortedData, double q) {
    const auto n = sortedData.size();
    const auto index = (n - 1) * q;
    const auto lowerIndex = static_cast<size_t>(floor(index));
    const auto upperIndex = static_cast<size_t>(ceil(index));
    if (lowerIndex == upperIndex) return sortedData[lowerIndex];
    const auto frac = index - lowerIndex;
    return (1.0 - frac) * sortedData[lowerIndex] + frac * sortedData[upperIndex];
}

Node *findLCA(Node *node1, Node *node2) {
    unordered_map<Node *, bool> ancestors;
    for (Node *c = node1; c != nullptr; c = c->parent) ancestors[c] = true;
    for (Node *c = node2; c != nullptr; c = c->parent)
        if (ancestors.find(c) != ancestors.end()) return c;
    throw runtime_error("No common ancestor found.");
}
double calculateDistanceToAncestor(Node *startNode, Node *ancestor) {
    double distance = 0.0;
    Node *current = startNode;
    while (current != nullptr && current != ancestor) {
        distance += (current->dist);
        current = current->parent;
    }
    return current == ancestor ? distance : -1.0;
}
/* the reference sizes the result by the number of leaves and indexes it by node number; entries at or beyond that size
 * are never read back (calculateEuclideanDistance stops at the shorter vector), so they are not stored here */
vector<double> getPatristicDistances(const Tree *tr, Node *node, int numofLeafs, double posonbranch) {
    vector<double> distances(numofLeafs, numeric_limits<double>::max());
    for (size_t i = 0; i < tr->nodes.size(); ++i) {
        Node *leafNode = tr->nodes[i];
        if (!leafNode->isLeaf()) continue;
        Node *lca = findLCA(node, leafNode);
        double distanceToLCAFromNode = calculateDistanceToAncestor(node, lca) - posonbranch;
        double distanceToLCAFromLeaf = calculateDistanceToAncestor(leafNode, lca);
        if (distanceToLCAFromNode >= 0.0 && distanceToLCAFromLeaf >= 0.0 && i < distances.size()) distances[i] = distanceToLCAFromNode + distanceToLCAFromLeaf;
    }
    return distances;
}
long double calculateEuclideanDistance(const vector<double> &vec1, const vector<double> &vec2) {
    size_t minSize = min(vec1.size(), vec2.size());
    long double sum = 0.0;
    for (size_t i = 0; i < minSize; ++i) {
        if (vec1[i] == numeric_limits<double>::max() || vec2[i] == numeric_limits<double>::max()) continue;
        double diff = vec1[i] - vec2[i];
        sum += diff * diff;
    }
    return sqrt(sum);
}

void updatePosition(Run &R, PosTree &current_position, double move_distance, bool move_forward) {
    if (current_position.pos_branch < 0.0 || current_position.pos_branch > 1.0) throw runtime_error("Error: Initial pos_branch is out of valid range.");
    if (current_position.pos == nullptr) throw runtime_error("Error: current position pointer is null.");
    if (move_distance < 0.0) throw runtime_error("Error: move distance cannot be negative.");
    double move_distance_abs = abs(move_distance);
    while (move_distance_abs > 0.0) {
        if (move_forward) {
            if (current_position.pos_branch + move_distance_abs < 1.0) {
                current_position.pos_branch += move_distance_abs;
                move_distance_abs = 0.0;
            } else {
                if (current_position.pos->children == nullptr) {
                    move_forward = false;
                    continue;
                }
                double remaining_distance = move_distance_abs - (1.0 - current_position.pos_branch);
                if (remaining_distance < 0.0) remaining_distance = 0.0;
                int num_children = current_position.pos->nchildren;
                int random_index = pick(R, num_children);
                current_position.pos = current_position.pos->children[random_index];
                if (current_position.pos->dist < 0.0) throw runtime_error("Error: next branch length cannot be negative.");
                if (remaining_distance > 1.0) {
                    current_position.pos_branch = 1.0;
                    move_distance_abs = remaining_distance - 1.0;
                } else {
                    current_position.pos_branch = remaining_distance;
                    move_distance_abs = 0.0;
                }
            }
        } else {
            if (current_position.pos_branch - move_distance_abs > 0.0) {
                current_position.pos_branch = current_position.pos_branch - move_distance_abs;
                move_distance_abs = 0.0;
            } else {
                vector<Node *> possible_nodes;
                if (current_position.pos->parent == nullptr) {
                    move_forward = true;
                    int num_children = current_position.pos->nchildren;
                    int random_index = pick(R, num_children);
                    current_position.pos = current_position.pos->children[random_index];
                    if (current_position.pos->dist < 0.0) throw runtime_error("Error: next branch length cannot be negative.");
                    continue;
                } else {
                    possible_nodes.push_back(current_position.pos->parent);
                }
                if (current_position.pos->children != nullptr) {
                    for (int i = 0; i < current_position.pos->parent->nchildren; ++i)
                        if (current_position.pos->parent->children[i] != current_position.pos) possible_nodes.push_back(current_position.pos->parent->children[i]);
                }
                Node *chosen_node = possible_nodes[pick(R, possible_nodes.size())];
                if (chosen_node == current_position.pos->parent) {
                    double remaining_distance = move_distance_abs - current_position.pos_branch;
                    if (remaining_distance < 0.0) remaining_distance = 0.0;
                    current_position.pos = current_position.pos->parent;
                    if (current_position.pos->dist < 0.0) throw runtime_error("Error: parent branch length cannot be negative.");
                    if (remaining_distance > 1.0) {
                        current_position.pos_branch = 0.0;
                        move_distance_abs = remaining_distance - 1.0;
                        continue;
                    } else {
                        double new_pos_branch = 1.0 - remaining_distance;
                        if (new_pos_branch <= 0.0 || new_pos_branch >= 1.0) throw runtime_error("Error: new position branch is not in the valid range.");
                        current_position.pos_branch = new_pos_branch;
                        move_distance_abs = 0.0;
                    }
                } else {
                    move_forward = true;
                    current_position.pos = chosen_node;
                    double remaining_distance = move_distance_abs - current_position.pos_branch;
                    current_position.pos_branch = 0.0;
                    if (current_position.pos_branch + remaining_distance < 1.0) {
                        current_position.pos_branch = remaining_distance;
                        if (current_position.pos_branch < 0.0 || current_position.pos_branch > 1.0) throw runtime_error("Error: pos_branch is out of valid range after increment.");
                        move_distance_abs = 0.0;
                    } else {
                        if (current_position.pos->children == nullptr) {
                            move_forward = false;
                            continue;
                        }
                        double remaining_distance = move_distance_abs - (1.0 - current_position.pos_branch);
                        if (remaining_distance < 0.0) remaining_distance = 0.0;
                        int num_children = current_position.pos->nchildren;
                        int random_index = pick(R, num_children);
                        current_position.pos = current_position.pos->children[random_index];
                        if (current_position.pos->dist < 0.0) throw runtime_error("Error: next branch length cannot be negative.");
                        if (remaining_distance > 1.0) {
                            current_position.pos_branch = 1.0;
                            move_distance_abs = remaining_distance - 1.0;
                        }
                    }
                }
            }
        }
    }
    if (current_position.pos_branch < 0.0 || current_position.pos_branch > 1.0) throw runtime_error("Error: pos_branch is out of valid range after movement.");
}

vector<double> sample_normal(Run &R, vector<double> &x) {
    vector<double> result;
    if (x.empty()) throw invalid_argument("vector can't be empty");
    long double sum = 0.0L;
    for (size_t i = 0; i < x.size(); ++i) {
        normal_distribution<double> dist(x[i], 0.1);
        double sample;
        do {
            sample = dist(R.theta_engine);
        } while (sample < 0.0L || sample > 1.0L);
        result.emplace_back(sample);
        sum += sample;
    }
    for (size_t i = 0; i < result.size(); ++i) result[i] /= sum;
    return result;
}

vector<double> generateRandomNumbers(Run &R, int size) { /* MCMC.h:453-465 */
    mt19937 gen(R.rd());
    uniform_real_distribution<> dis(0.0, 1.0);
    vector<double> random_numbers(size);
    double sum = 0.0;
    for (double &num : random_numbers) {
        num = dis(gen);
        sum += num;
    }
    for (double &num : random_numbers) num /= sum;
    return random_numbers;
}

MCMCiteration initializeState(Run &R, Params &params) {
    MCMCiteration state;
    state.n_components = params.sources.size();
    vector<double> random_numbers = generateRandomNumbers(R, state.n_components);
    vector<PosTree> current_positions(random_numbers.size());
    int index = 0;
    for (auto &p : current_positions) {
        p.pos = params.tr->nodes.at(params.sources[index]);
        p.pos_branch = 0.5;
        p.theta = random_numbers[index];
        p.branch_place_anc = 0.5;
        p.branch_place_der = 0.5;
        index++;
    }
    state.positions_tree = current_positions;
    for (auto &p : state.positions_tree) {
        state.proportions.emplace_back(max(0.001, p.theta));
        state.max_branch_lens.emplace_back(p.pos->dist);
    }
    state.logLike = params.logLike;
    return state;
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

vector<MCMCiteration> run_tree_proportion(Run &R, Params params, vector<MCMCiteration> state_t_vec, const string &num, int numPaths, int chainindex, double con) {
    const unsigned int n_sources = params.sources.size();
    MCMCiteration state_t_1;
    double likelihood_t_1;
    mt19937 gen(R.rd());
    uniform_real_distribution<> dis(0.0, 1.0);
    vector<PosTree> current_positions(n_sources);
    MCMCiteration state_t = initializeState(R, params);
    double proposal_sd;
    double initSD;
    if (numPaths <= 30.0) initSD = 3.0;
    else initSD = numPaths * (3.0 / 30.0);
    GzOut mcmcout(num + "Result" + to_string(n_sources) + to_string(chainindex) + ".mcmc");
    {
        ostringstream o;
        for (unsigned sou = 1; sou < n_sources + 1; ++sou) o << "Source_" << sou << '\t' << "Log-likelihood" << '\t' << "proportion" << '\t' << "branch_position_derived" << '\t';
        o << endl;
        mcmcout.write(o.str());
    }
    GzOut mcmcdetail(num + "Trace" + to_string(n_sources) + to_string(chainindex) + ".detail.mcmc");
    {
        ostringstream o;
        for (unsigned sou = 1; sou < n_sources + 1; ++sou)
            o << "Source_" << sou << '\t' << "Log-likelihood" << '\t' << "proportion_" << sou << '\t' << "branch_position_derived_" << sou << '\t' << "Move" << '\t';
        o << endl;
        mcmcdetail.write(o.str());
    }
    for (unsigned int iteration = 0; iteration <= params.maxIter; iteration++) {
        if (params.burn >= params.maxIter) throw runtime_error("Number of brun in iteration exceedes the number of total iterations. Exiting. ");
        double step = (initSD - 0.1) / std::max(static_cast<unsigned int>(1), params.burn - 1);
        double step2 = (0.1 - 1e-5) / std::max(static_cast<unsigned int>(1), (params.maxIter - params.burn) - 1);
        state_t_1 = state_t;
        if (iteration < params.burn) {
            proposal_sd = std::max(1e-5, initSD - iteration * step);
        } else {
            if (iteration % 100000 == 0) proposal_sd = 1;
            else proposal_sd = std::max(1e-5, 0.1 - (iteration - params.burn) * step2);
        }
        if (iteration != 0) {
            for (int i = 0; i < state_t_1.n_components; i++) {
                normal_distribution<double> distribution_bl(0, proposal_sd);
                double proposed_position = distribution_bl(gen);
                if (proposed_position < 0.0) updatePosition(R, state_t_1.positions_tree[i], -proposed_position, false);
                else updatePosition(R, state_t_1.positions_tree[i], proposed_position, true);
            }
        }
        vector<double> tmp_theta;
        for (auto &p : state_t_1.positions_tree) tmp_theta.emplace_back(p.theta);
        tmp_theta = sample_normal(R, tmp_theta);
        for (size_t idx = 0; idx < current_positions.size(); ++idx) state_t_1.positions_tree[idx].theta = tmp_theta[idx];
        state_t_1.proportions = tmp_theta;
        vector<string> pathNames, parentpathNames;
        for (auto &p : state_t_1.positions_tree) {
            pathNames.emplace_back(p.pos->longname);
            if (p.pos->parent != nullptr) parentpathNames.emplace_back(p.pos->parent->longname);
            else parentpathNames.emplace_back(p.pos->longname);
        }
        double logLike = 0.0;
        {
            const int k = (int)pathNames.size();
            vector<int32_t> child(k), parent(k);
            vector<double> dist(k), pos(k), theta(k);
            for (int y = 0; y < k; ++y) {
                child[y] = R.path_index.at(pathNames[y]);
                parent[y] = R.path_index.at(parentpathNames[y]);
                dist[y] = state_t_1.positions_tree[y].pos->dist;
                pos[y] = state_t_1.positions_tree[y].pos_branch;
                theta[y] = state_t_1.proportions[y];
            }
            if (orc_sb_loglike(R.h, k, child.data(), parent.data(), dist.data(), pos.data(), theta.data(), con, params.freqs7, 1, &logLike) != 0)
                throw runtime_error("Problem in the likelihood compuation! Intermediate log likelihood is -nan, -inf or positive.");
        }
        likelihood_t_1 = logLike;
        state_t_1.logLike = likelihood_t_1;
        double acceptance_prob = (state_t_1.logLike - state_t.logLike > 0) ? 1.0 : exp(state_t_1.logLike - state_t.logLike);
        double u = dis(gen);
        if (u <= acceptance_prob || iteration == 0) {
            ostringstream d;
            for (auto p : state_t_1.positions_tree) d << std::setprecision(14) << p.pos->longname << "\t" << state_t_1.logLike << '\t' << p.theta << '\t' << p.pos_branch << '\t' << "accepted" << '\t';
            d << endl;
            mcmcdetail.write(d.str());
            if (iteration > params.burn) {
                ostringstream o;
                for (auto &p : state_t.positions_tree) o << setprecision(14) << p.pos->longname << '\t' << state_t.logLike << '\t' << p.theta << '\t' << p.pos_branch << '\t';
                state_t_vec.emplace_back(state_t);
                o << endl;
                mcmcout.write(o.str());
            }
            state_t = state_t_1;
        } else {
            ostringstream d;
            for (auto p : state_t_1.positions_tree) d << std::setprecision(14) << p.pos->longname << "\t" << state_t_1.logLike << '\t' << p.theta << '\t' << p.pos_branch << '\t' << "rejected" << '\t';
            d << endl;
            mcmcdetail.write(d.str());
            if (iteration > params.burn) {
                ostringstream o;
                for (auto &p : state_t.positions_tree) o << setprecision(14) << p.pos->longname << '\t' << state_t.logLike << '\t' << p.theta << '\t' << p.pos_branch << '\t';
                o << endl;
                mcmcout.write(o.str());
                state_t_vec.emplace_back(state_t);
            }
        }
    }
    return state_t_vec;
}

pair<map<string, vector<vector<double>>>, double> processMCMCiterations(const vector<MCMCiteration> &MCMCiterations, int k, const string &num, int chain, const Tree *tr, int numofleafs) {
    map<string, vector<vector<double>>> branchStatisticsMap;
    ofstream estimatesFile, branchestimateFile;
    estimatesFile.open(num + "ProportionEstimates" + to_string(k) + ".txt", ios::app | ios::out);
    branchestimateFile.open(num + "BranchEstimate" + to_string(k) + ".txt", ios::app | ios::out);
    estimatesFile << "Source\tChain\tMean Proportion Estimate\t5% CI\tMedian Proportion Estimate\t95% CI\tEffective Sample Size\tAutocorrelation\tVariance\n";
    branchestimateFile << "Source\tChain\tMean Branch Position\t5% CI\tMedian Branch Position\t95% CI\tEffective Sample Size\tAutocorrelation\tVariance\tEffective Sample Size for the source estimation\n";
    double chainloglike = MCMCiterations.at(0).logLike;
    for (int source = 0; source < k; ++source) {
        vector<double> sourceStatistic = {};
        vector<long double> proportionVec = {};
        vector<long double> positionVec = {};
        string branchName;
        vector<double> initialPatristicDistances;
        vector<long double> euc_distances;
        initialPatristicDistances = vector<double>(numofleafs, 1.0);
        size_t totalIterations = MCMCiterations.size();
        for (size_t idx = 0; idx < totalIterations; ++idx) {
            const auto &iteration = MCMCiterations[idx];
            if (iteration.logLike > chainloglike) chainloglike = iteration.logLike;
            branchName = iteration.positions_tree[source].pos->longname;
            if (branchStatisticsMap.find(branchName) == branchStatisticsMap.end()) branchStatisticsMap[branchName] = vector<vector<double>>();
            proportionVec.emplace_back(iteration.proportions[source]);
            positionVec.emplace_back(iteration.positions_tree[source].pos_branch);
            double t1 = iteration.positions_tree[source].pos->dist * iteration.positions_tree[source].pos_branch;
            double posonbranch = iteration.positions_tree[source].pos->dist - t1;
            const vector<double> patristic_distances = getPatristicDistances(tr, iteration.positions_tree[source].pos, numofleafs, posonbranch);
            long double euc_dist = calculateEuclideanDistance(patristic_distances, initialPatristicDistances);
            euc_distances.emplace_back(euc_dist);
        }
        long double meanTheta = mean(proportionVec);
        long double meanPos = mean(positionVec);
        long double Theta_autoc = autocorrelation(proportionVec, 1);
        long double Theta_ess = effectiveSampleSize(proportionVec);
        long double Theat_var = variance(proportionVec, meanTheta);
        long double Pos_autoc = autocorrelation(positionVec, 1);
        long double Pos_ess = effectiveSampleSize(positionVec);
        long double dist_ess = effectiveSampleSize(euc_distances);
        long double Pos_var = variance(positionVec, meanPos);
        sort(positionVec.begin(), positionVec.end());
        sort(proportionVec.begin(), proportionVec.end());
        long double Theta_fq = getQuantile2(proportionVec, 0.05);
        long double Theta_tq = getQuantile2(proportionVec, 0.95);
        long double Theta_median = getQuantile2(proportionVec, 0.5);
        long double Pos_fq = getQuantile2(positionVec, 0.05);
        long double Pos_median = getQuantile2(positionVec, 0.5);
        long double Pos_tq = getQuantile2(positionVec, 0.95);
        estimatesFile << branchName << '\t' << chain << '\t' << meanTheta << '\t' << Theta_fq << '\t' << Theta_median << '\t' << Theta_tq << '\t' << Theta_ess << '\t' << Theta_autoc << '\t' << Theat_var << '\n';
        branchestimateFile << branchName << '\t' << chain << '\t' << meanPos << '\t' << Pos_fq << '\t' << Pos_median << '\t' << Pos_tq << '\t' << Pos_ess << '\t' << Pos_autoc << '\t' << Pos_var << '\t' << dist_ess << '\n';
        sourceStatistic.emplace_back(meanTheta);
        sourceStatistic.emplace_back(Theat_var);
        sourceStatistic.emplace_back(meanPos);
        sourceStatistic.emplace_back(Pos_var);
        branchStatisticsMap[branchName].emplace_back(sourceStatistic);
    }
    return make_pair(branchStatisticsMap, chainloglike);
}

vector<int> soibean_generateRandomNumbers(Run &R, const int maxNum, const int k) {
    vector<int> sigNodes;
    mt19937 gen(R.rd());
    uniform_int_distribution<> distrib(0, static_cast<int>(maxNum) - 1);
    for (int i = 0; i < k; ++i) sigNodes.emplace_back(static_cast<int>(distrib(gen)));
    return sigNodes;
}

double calculateRhat(const vector<double> &means, const vector<double> &variances, int chainLength) {
    int numChains = means.size();
    if (numChains < 2) return -1;
    double W = accumulate(variances.begin(), variances.end(), 0.0) / numChains;
    double grandMean = accumulate(means.begin(), means.end(), 0.0) / numChains;
    double B = 0.0;
    for (int i = 0; i < numChains; ++i) B += pow(means[i] - grandMean, 2);
    B *= chainLength / (numChains - 1);
    double varEstimate = ((chainLength - 1.0) * W + B) / chainLength;
    return sqrt(varEstimate / W);
}

} // namespace

extern "C" int orc_sb_estimate(const void *h, const char *newick, const char *path_names, const int32_t *sig_nodes, int32_t n_sig,
                               const orc_sb_estimate_cfg *cfg, const char *prefix) {
    try {
        Tree taxatree;
        read_newick(newick, taxatree);
        Run R(h, cfg->seed);
        {
            istringstream ns(path_names);
            string line;
            int idx = 0;
            while (getline(ns, line)) R.path_index[line] = idx++;
        }
        int leafcounter = 0;
        for (Node *n : taxatree.nodes)
            if (n->isLeaf()) leafcounter++;
        const int numPaths = (int)R.path_index.size();
        vector<int> sigNodes(sig_nodes, sig_nodes + n_sig);
        vector<string> sigPaths;
        for (int v : sigNodes) sigPaths.emplace_back(taxatree.nodes.at(v)->longname);
        const string num = prefix;
        for (size_t i = 0; i < sigNodes.size(); ++i) {
            vector<int> subVector(sigNodes.begin(), sigNodes.begin() + i + 1);
            double logLike = 0.0L;
            double freq = log(1.0 / sigNodes.size());
            if (sigPaths.size() == 1) {
                int32_t p = R.path_index.at(sigPaths[i]);
                logLike = orc_sb_mixture_loglike(h, 1, &p, 0.0);
            } else {
                vector<int32_t> ps;
                for (size_t j = 0; j < subVector.size(); ++j) ps.push_back(R.path_index.at(sigPaths[j]));
                logLike = orc_sb_mixture_loglike(h, (int32_t)ps.size(), ps.data(), freq);
            }
            Params params;
            params.tr = &taxatree;
            params.sources = subVector;
            params.burn = cfg->burn;
            params.maxIter = cfg->max_iter;
            params.chains = cfg->chains;
            params.logLike = logLike;
            params.freqs7 = cfg->freqs7;
            vector<vector<MCMCiteration>> MCMCiterationsVec(cfg->chains);
            if (cfg->run_mcmc) {
                ofstream diagnostics;
                map<string, vector<vector<vector<double>>>> branchStatsMap;
                unsigned int chainIndex = 0;
                vector<double> chainLogLikes;
                diagnostics.open(num + "Diagnostics" + to_string(subVector.size()) + to_string(chainIndex) + ".txt");
                diagnostics << "Source\tHighest log-likelihood\tfor chain\tRhat for the proportion estimate\tRhat for the branch position estimate" << endl;
                for (auto &chainVec : MCMCiterationsVec) {
                    if (chainIndex != 0) params.sources = soibean_generateRandomNumbers(R, numPaths, subVector.size());
                    vector<MCMCiteration> chainiter = run_tree_proportion(R, params, chainVec, num, numPaths, chainIndex, cfg->con);
                    auto intermStatsMapPair = processMCMCiterations(chainiter, subVector.size(), num, chainIndex, &taxatree, leafcounter);
                    chainLogLikes.emplace_back(intermStatsMapPair.second);
                    for (const auto &branchStat : intermStatsMapPair.first) {
                        auto &slot = branchStatsMap[branchStat.first];
                        slot.resize(cfg->chains); /* definition: one slot per chain, empty when the chain did not visit the branch */
                        slot[chainIndex] = branchStat.second;
                    }
                    chainIndex++;
                }
                int numChains = cfg->chains;
                int chainLength = cfg->max_iter - cfg->burn;
                for (const auto &branchStat : branchStatsMap) {
                    const auto &branchName = branchStat.first;
                    auto allChainStats = branchStat.second;
                    vector<double> Propmeans(numChains, 1.0), Propvariances(numChains, 1.0), Posmeans(numChains, 1.0), Posvariances(numChains, 1.0);
                    for (int chain = 0; chain < numChains; ++chain) {
                        if (allChainStats[chain].empty()) allChainStats[chain] = {{1.0, 1.0, 1.0, 1.0}};
                        if (allChainStats[chain][0].size() < 4) allChainStats[chain][0].resize(4, 1.0);
                        Propmeans[chain] = allChainStats[chain][0][0];
                        Propvariances[chain] = allChainStats[chain][0][1];
                        Posmeans[chain] = allChainStats[chain][0][2];
                        Posvariances[chain] = allChainStats[chain][0][3];
                    }
                    double maxLogLike = chainLogLikes[0];
                    int maxIndex = 0;
                    for (size_t hh = 0; hh < chainLogLikes.size(); ++hh)
                        if (chainLogLikes[hh] > maxLogLike) {
                            maxLogLike = chainLogLikes[hh];
                            maxIndex = hh;
                        }
                    double PropRhat = calculateRhat(Propmeans, Propvariances, chainLength);
                    double PosRhat = calculateRhat(Posmeans, Posvariances, chainLength);
                    diagnostics << branchName << '\t' << maxLogLike << '\t' << maxIndex << '\t' << PropRhat << '\t' << PosRhat << std::endl;
                }
            }
        }
        return 0;
    } catch (const std::exception &e) {
        cerr << "orc_sb_estimate: " << e.what() << endl;
        return -1;
    }
}
