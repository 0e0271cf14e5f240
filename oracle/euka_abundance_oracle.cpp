/*
 * oracle/euka_abundance_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * What `vgan euka` does with readGAM3's result, restated for the CPU from the reference (paths relative to
 * /root/reference/src), loops and types as there:
 *   readGAM_Euka.h:582-630   detected-clade list (bins above the entropy threshold, zero-bin / bin-count / read-count filters)
 *   compute_init_vec.cpp:9-84 initial abundance vector
 *   MCMC.cpp:1095-1172       MCMC::generate_proposal (log -> normal draw per element -> softmax), MCMC.h:631-652 softmax
 *   MCMC.cpp:1175-1215       MCMC::get_proposal_likelihood (the literal per-read loop, `(1/334)` is the integer 0: Q14)
 *   MCMC.cpp:1217-1366       MCMC::run (proposals recorded after burn-in, median + 85 % / 95 % quantiles per clade)
 *   miscfunc.h:239-251       quant
 *   Euka.cpp:540-1160        *_abundance.tsv, *_detected.tsv, *_coverage.tsv, *_inSize.tsv, *_FragNames.tsv, per-clade .prof,
 *                            *_5p.prof / *_3p.prof;  Euka.cpp:54-69 get_avg;  baseshift.cpp:143-230 Baseshift::display_prof
 * "parity unpinned": the reference's tests for this stage (src/test.cpp:1002-1172) map FASTQ files with vg giraffe against the
 * euka database (neither is available here) and assert abundance windows; they cannot run.
 *
 * Randomness.  The reference draws a fresh seed from std::random_device for every proposal and once for the acceptance
 * draws (MCMC.cpp:1132-1133,1227-1228).  Here that entropy source is a parameter: seed 0 = std::random_device, otherwise
 * the 32-bit outputs of a splitmix64 stream started at `seed` stand in for successive rd() calls, in call order.
 * Undefined behaviour and the oracle's definition:
 *   - an empty *.bins line (`chunks[i].size()-1` wraps): no bins;  iter <= burnin + 1 (empty sample, `sorted_clade[0]`): error;
 *   - `for (int j; ...)` at Euka.cpp:563 starts at 0;  a clade id outside the table, or --outGroup with --outFrag naming
 *     no clade (`clade_vec->at(-5)`): error.
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <numeric>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include <sys/stat.h>

using std::string;
using std::vector;

namespace {

struct Clade {
    int id = 0;
    string name;
    int count = 0;
    vector<double> clade_like{0.0}, clade_not_like{0.0}; // load.cpp:128: index 0 is a dummy
    vector<int> inSize{0};                               // load.cpp:129
    vector<string> nameStorage{""};                      // load.cpp:130
    const uint32_t *baseshift = nullptr;                 // [2*lengthToProf][16]
};
typedef vector<vector<std::tuple<int, int, double, double>>> Chunks;

struct Entropy {
    uint64_t s;
    bool hw;
    explicit Entropy(uint64_t seed) : s(seed), hw(seed == 0) {}
    uint32_t operator()() {
        if (hw) return std::random_device{}();
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (uint32_t)(z >> 32);
    }
};

// readGAM_Euka.h:596-630 (and again at Euka.cpp:616-646,913-930)
bool clade_rejected(const Chunks &chunks, const vector<Clade> &cl, size_t i, const orc_euka_report_cfg &c) {
    vector<int> check_for_zero; // an int vector: the coverage is truncated on the way in
    const size_t nb = chunks[i].empty() ? 0 : chunks[i].size() - 1;
    for (size_t k = 0; k < nb; k++)
        if (std::get<2>(chunks[i][k]) > c.ENTROPY_SCORE_THRESHOLD) check_for_zero.emplace_back(std::get<3>(chunks[i][k]));
    const int num_zero_bins = std::count(check_for_zero.begin(), check_for_zero.end(), 0.0);
    return num_zero_bins > c.MAXIMUMOFBINS || check_for_zero.size() < c.MINNUMOFBINS || (unsigned)cl[i].count < c.MINNUMOFREADS;
}

Clade &by_id(vector<Clade> &cl, int id) {
    if (id < 0 || (size_t)id >= cl.size()) throw std::runtime_error("clade id outside the clade table");
    return cl[(size_t)id];
}

vector<long double> compute_init_vec(vector<Clade> &cl, const vector<int> &ids) {
    long double total = 0;
    for (int id : ids) total += by_id(cl, id).count;
    vector<long double> v;
    for (int id : ids) {
        const long double frag = by_id(cl, id).count;
        v.emplace_back(frag / total);
    }
    return v;
}

vector<long double> softmax(const vector<long double> &lv) {
    long double K = 0.0;
    for (long double x : lv) K += std::exp(x);
    vector<long double> out;
    for (long double x : lv) out.emplace_back(std::exp(x) / K);
    return out;
}

vector<long double> generate_proposal(const vector<long double> &cur, double alpha, Entropy &rd) {
    vector<long double> lg;
    for (long double x : cur) lg.emplace_back(std::log(x));
    std::mt19937 g(rd());
    vector<long double> prop;
    for (long double e : lg) {
        std::normal_distribution<long double> d{e, alpha};
        prop.emplace_back(d(g));
    }
    return softmax(prop);
}

long double proposal_likelihood(const vector<long double> &prop, vector<Clade> &cl, const vector<int> &ids) {
    long double ll = 0.0;
    vector<double> flat; // (id, fraction) pairs, both narrowed to double
    for (size_t i = 0; i < ids.size(); i++) {
        flat.emplace_back(ids[i]);
        flat.emplace_back(prop.at(i));
    }
    for (size_t j = 0; j < flat.size(); j += 2) {
        const Clade &c = by_id(cl, (int)flat[j]);
        const double frac = flat[j + 1];
        double total_frac = 0.0;
        for (size_t k = 1; k < c.clade_like.size(); k++) total_frac += std::log((frac * c.clade_like[k]) + (c.clade_not_like[k] * (1 / 334)));
        ll += total_frac;
    }
    return ll;
}

long double quant(const vector<long double> &x, double q) {
    const auto n = x.size();
    const auto id = (n - 1) * q;
    const auto lo = std::floor(id);
    const auto hi = std::ceil(id);
    const auto qs = x[(size_t)lo];
    const auto h = (id - lo);
    return (1.0 - h) * qs + h * x[(size_t)hi];
}

vector<long double> mcmc_run(int iter, int burnin, const vector<long double> &init, vector<Clade> &cl, const vector<int> &ids, Entropy &rd) {
    if (iter - burnin - 1 <= 0) throw std::runtime_error("iter must exceed burnin + 1");
    vector<long double> current_best = init, proposal;
    long double current_ll = -9999999;
    std::mt19937 gen(rd());
    struct Move {
        long double ll;
        vector<long double> v;
    };
    vector<Move> move((size_t)iter);
    for (int it = 0; it < iter; it++) {
        proposal = generate_proposal(current_best, 0.1, rd);
        const long double pll = proposal_likelihood(proposal, cl, ids);
        if (it > burnin) {
            move[it].ll = pll;
            move[it].v = proposal;
        } else
            continue;
        const double acceptance = std::min((long double)(1.0), expl(pll - current_ll));
        std::uniform_real_distribution<> dis(0, 1);
        const double u = dis(gen);
        if (u <= acceptance || it == 0) {
            current_ll = pll;
            current_best = proposal;
        }
    }
    vector<long double> est, sorted;
    for (size_t j = 0; j < proposal.size(); j++) {
        for (int i = burnin + 1; i < iter; i++) sorted.emplace_back(move[i].v[j]);
        std::sort(sorted.begin(), sorted.end());
        est.emplace_back(sorted[sorted.size() / 2]);
        est.emplace_back(quant(sorted, 0.15));
        est.emplace_back(quant(sorted, 0.85));
        est.emplace_back(quant(sorted, 0.05));
        est.emplace_back(quant(sorted, 0.95));
        sorted.clear();
    }
    return est;
}

// baseshift.cpp:143-230
vector<vector<double>> display_prof(const uint32_t *arr, int ltp, const string &path) {
    std::ofstream o(path.c_str());
    const char *hdr = "A>C\tA>G\tA>T\tC>A\tC>G\tC>T\tG>A\tG>C\tG>T\tT>A\tT>C\tT>G\tPosition";
    o << hdr << std::endl;
    vector<double> CtoT, GtoA;
    for (int p = 0; p < ltp * 2; p++) {
        const uint32_t *row = arr + (size_t)p * 16;
        int col = 0, col_div = 0;
        for (int i = 0; i < 4; i++) {
            for (int j = 0; j < 4; j++) {
                if (i != j) {
                    const double v = (double)row[col] / ((double)row[0 + col_div] + (double)row[1 + col_div] + (double)row[2 + col_div] + (double)row[3 + col_div]);
                    o << std::setprecision(4) << v << "\t";
                    if (i == 1 && j == 3) CtoT.emplace_back(v);
                    if (i == 2 && j == 0) GtoA.emplace_back(v);
                }
                col++;
            }
            col_div += 4;
        }
        if (p < ltp) o << p << std::endl;
        if (p >= ltp) o << -(ltp * 2) + p << std::endl;
        if (p == ltp - 1) o << hdr << std::endl;
    }
    return {CtoT, GtoA};
}

vector<double> get_avg(const vector<double> &dam, int l, int no_clades) {
    vector<double> sum(l, 0.0), av;
    for (size_t i = 0; i < dam.size(); i++) sum[i % l] += dam[i];
    for (double s : sum) av.emplace_back(s / no_clades);
    return av;
}

void write_end_profile(const string &path, const vector<double> &v, int hot_col) {
    std::ofstream o(path.c_str(), std::ios::trunc);
    o << "A>C\tA>G\tA>T\tC>A\tC>G\tC>T\tG>A\tG>C\tG>T\tT>A\tT>C\tT>G" << std::endl;
    for (size_t pas = 0; pas < v.size(); pas++)
        for (int col = 0; col < 12; col++) {
            if (col == hot_col) o << v[pas] << '\t';
            else if (col == 11) o << 0 << std::endl;
            else o << 0 << '\t';
        }
}

void write_bins(std::ofstream &outbin, const Chunks &chunks, size_t i, size_t no_tab_at) {
    const size_t nb = chunks[i].empty() ? 0 : chunks[i].size() - 1;
    for (size_t j = 0; j < nb; j++) {
        outbin << std::fixed << std::setprecision(5);
        outbin << std::get<3>(chunks[i][j]) << '\t' << std::get<2>(chunks[i][j]);
        if (j != no_tab_at) outbin << '\t';
    }
}

void write_insize(std::ofstream &o, const Clade &c) {
    o << c.name << '\t';
    for (size_t s = 1; s < c.inSize.size(); s++) {
        o << c.inSize[s];
        if (s != c.inSize.size() - 1) o << '\t';
    }
    o << std::endl;
}

} // namespace

extern "C" int orc_euka_report(const orc_euka_db *db, const int32_t *clade_id, const char *clade_names, const orc_euka_out *res,
                               int64_t n_reads, const int32_t *read_seq_len, const char *names, const int64_t *name_off,
                               const orc_euka_report_cfg *cfg, const char *prefix, int32_t *detected, int32_t *n_detected,
                               double *estimates) {
    try {
        const orc_euka_report_cfg &c = *cfg;
        const int ltp = c.lengthToProf;
        vector<Clade> cl((size_t)db->n_clades);
        {
            std::istringstream ns(clade_names);
            for (int i = 0; i < db->n_clades; i++) {
                std::getline(ns, cl[i].name);
                cl[i].id = clade_id[i];
                cl[i].count = res->clade_count[i];
                cl[i].baseshift = res->baseshift + (size_t)i * 2 * ltp * 16;
            }
        }
        // what the lambda pushed, in read order (readGAM_Euka.h:485-512)
        for (int64_t r = 0; r < n_reads; r++) {
            const int cn = res->read_clade[r];
            if (cn < 0) continue;
            cl[cn].clade_like.emplace_back(res->read_like[r]);
            cl[cn].clade_not_like.emplace_back(res->read_not_like[r]);
            if (res->read_pass[r]) {
                cl[cn].inSize.emplace_back(read_seq_len[r]);
                cl[cn].nameStorage.emplace_back(names ? string(names + name_off[r], names + name_off[r + 1]) : string());
            }
        }
        Chunks chunks((size_t)db->n_clades);
        for (int i = 0; i < db->n_clades; i++)
            for (int j = db->bin_off[i]; j < db->bin_off[i + 1]; j++)
                chunks[i].emplace_back(db->bin_lo[j], db->bin_hi[j], db->bin_entropy[j], res->bin_cov[j]);

        vector<int> clade_id_list; // readGAM3's second return value
        for (size_t i = 0; i < chunks.size(); i++)
            if (!clade_rejected(chunks, cl, i, c)) clade_id_list.emplace_back(cl[i].id);

        const string out = prefix, outGroup = c.outGroup ? c.outGroup : "";
        if (c.outFrag && !clade_id_list.empty()) { // Euka.cpp:542-582
            if (outGroup != "") {
                int extra_id = -1;
                for (size_t j = 0; j < chunks.size(); j++)
                    if (cl[j].name == outGroup) extra_id = cl[j].id;
                clade_id_list.emplace_back(extra_id);
            }
            std::ofstream f((out + "_FragNames.tsv").c_str(), std::ios::trunc);
            for (size_t i = 0; i < clade_id_list.size(); ++i) {
                const Clade &k = by_id(cl, clade_id_list[i]);
                f << k.name << '\t';
                for (size_t s = 1; s < k.nameStorage.size(); ++s) {
                    f << k.nameStorage[s];
                    if (s != k.nameStorage.size() - 1) f << '\t';
                }
                f << std::endl;
            }
        }

        const bool no_mcmc = clade_id_list.size() < 2 || !c.run_mcmc;
        const vector<long double> init_vec = compute_init_vec(cl, clade_id_list);
        vector<long double> clade_res;
        if (!no_mcmc) {
            Entropy rd(c.seed);
            clade_res = mcmc_run(c.iter, c.burnin, init_vec, cl, clade_id_list, rd);
        }
        if (n_detected) *n_detected = (int32_t)clade_id_list.size();
        for (size_t i = 0; i < clade_id_list.size(); i++) {
            if (detected) detected[i] = clade_id_list[i];
            if (estimates)
                for (int q = 0; q < 5; q++) estimates[i * 5 + q] = no_mcmc ? (q == 0 ? (double)init_vec[i] : 0.0) : (double)clade_res[i * 5 + q];
        }

        std::ofstream outbin((out + "_coverage.tsv").c_str(), std::ios::trunc);
        std::ofstream oab((out + "_abundance.tsv").c_str(), std::ios::trunc);
        std::ofstream outsurv((out + "_detected.tsv").c_str(), std::ios::trunc);
        std::ofstream outinSize((out + "_inSize.tsv").c_str(), std::ios::trunc);
        const char *h4 = "#Taxa\tdetected\tNumber_of_reads\tproportion_estimate";
        const char *hci = "\t85%_confidence_interval_lower_bound\t85%_confidence_interval_higher_bound\t95%_confidence_interval_lower_bound\t95%_confidence_interval_higher_bound";
        oab << h4 << (no_mcmc ? "" : hci) << '\n';
        outsurv << h4 << (no_mcmc ? "" : hci) << '\n';
        outbin << "#Taxa" << '\t';
        for (int bins = 0; bins < 21; bins++) {
            outbin << "bin" << bins << '\t' << "entropy";
            if (bins != 20) outbin << '\t';
        }
        outbin << std::endl;

        vector<int> clade_list_id;
        int extra_id = -1;
        for (size_t i = 0; i < chunks.size(); i++) {
            if (outGroup == cl[i].name) extra_id = cl[i].id;
            const size_t last = chunks[i].empty() ? 0 : chunks[i].size() - 1; // chunks[i].size()-1
            if (clade_rejected(chunks, cl, i, c)) {
                oab << cl[i].name << '\t' << "no" << '\t' << cl[i].count << '\t' << 0;
                if (!no_mcmc) oab << '\t' << 0 << '\t' << 0 << '\t' << 0 << '\t' << 0;
                oab << std::endl;
                if (outGroup == cl[i].name) {
                    outbin << cl[i].name << '\t';
                    write_bins(outbin, chunks, i, last - 1);
                    outbin << std::endl;
                    write_insize(outinSize, cl[i]);
                }
            } else {
                clade_list_id.emplace_back(cl[i].id);
                outbin << cl[i].name << '\t';
                if (no_mcmc) { // Euka.cpp:693-702: the separator test never fails, '\n' ends the line
                    write_bins(outbin, chunks, i, last);
                    outbin << '\n';
                } else { // Euka.cpp:969-980
                    write_bins(outbin, chunks, i, last - 1);
                    outbin << std::endl;
                }
                oab << cl[i].name << '\t' << "yes" << '\t' << cl[i].count << '\t';
                outsurv << cl[i].name << '\t' << "yes" << '\t' << cl[i].count << '\t';
                write_insize(outinSize, cl[i]);
                if (no_mcmc) { // Euka.cpp:717-724: the first |clade_list_id| entries of init_vec
                    for (size_t index = 0; index < clade_list_id.size(); index++) {
                        oab << init_vec.at(index);
                        outsurv << init_vec.at(index);
                        if (index != clade_list_id.size() - 1) {
                            oab << '\t';
                            outsurv << '\t';
                        }
                    }
                } else { // Euka.cpp:993-1000
                    for (size_t index = (clade_list_id.size() - 1) * 5; index < clade_list_id.size() * 5; index++) {
                        oab << clade_res.at(index);
                        outsurv << clade_res.at(index);
                        if (index != clade_res.size() - 1) {
                            oab << '\t';
                            outsurv << '\t';
                        }
                    }
                }
                oab << std::endl;
                outsurv << std::endl;
            }
        }

        if (c.out_dir && *c.out_dir) { // Euka.cpp:738-747
            struct stat sb;
            if (stat(c.out_dir, &sb) != 0 || !S_ISDIR(sb.st_mode)) mkdir(c.out_dir, 0777);
        }
        vector<double> end5ct, end3ct, end5ga, end3ga;
        for (size_t i = 0; i < clade_list_id.size(); i++) {
            const Clade &k = by_id(cl, clade_list_id[i]);
            const vector<vector<double>> dam = display_prof(k.baseshift, ltp, out + "_" + k.name + ".prof");
            const size_t half = dam.at(0).size() / 2;
            for (size_t a = 0; a < dam.at(0).size(); a++) (a < half ? end5ct : end3ct).emplace_back(dam[0][a]);
            const size_t half1 = dam.at(1).size() / 2;
            for (size_t a = 0; a < dam.at(1).size(); a++) (a < half1 ? end5ga : end3ga).emplace_back(dam[1][a]);
        }
        if (extra_id != -1) {
            const Clade &k = by_id(cl, extra_id);
            display_prof(k.baseshift, ltp, out + "_" + k.name + ".prof");
        }
        const int nc = (int)clade_id_list.size();
        const vector<double> end5ct_av = get_avg(end5ct, ltp, nc);
        vector<double> end3ga_av = get_avg(end3ga, ltp, nc);
        write_end_profile(out + "_5p.prof", end5ct_av, 5);
        std::reverse(end3ga_av.begin(), end3ga_av.end());
        write_end_profile(out + "_3p.prof", end3ga_av, 6);
        return 0;
    } catch (const std::exception &e) {
        return -1;
    }
}
