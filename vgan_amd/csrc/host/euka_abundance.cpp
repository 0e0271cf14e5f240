// euka downstream of the per-read pass: detected clades, the abundance MCMC and Euka::run's output files.
// Reference: src/readGAM_Euka.h:582-630, src/compute_init_vec.cpp:9-84, src/MCMC.cpp:1095-1366, src/MCMC.h:631-652,
// src/miscfunc.h:239-251, src/Euka.cpp:54-69,540-1160, src/baseshift.cpp:143-230.
//
// The reference walks every read's clade_like entry per MCMC iteration (O(reads x iterations)).  Its mixture term
// `clade_not_like * (1/334)` is an integer-division zero, so a proposal's log-likelihood is
//     sum_clades ( n_c * log(frac_c) + sum_k log(clade_like_c[k]) )
// and the per-clade sums come off the GPU once (vgan_euka_like_sums); an iteration here costs O(clades).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <limits>
#include <random>
#include <sstream>

#include <sys/stat.h>

using namespace vgan;

namespace {

// stands in for the reference's std::random_device objects (one rd() call each)
class SeedSource {
  public:
    explicit SeedSource(uint64_t seed) : state_(seed), hardware_(seed == 0) {}
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

struct BinRange {
    uint32_t begin, end; // the clade's bins without its last one (Euka.cpp:627 "size()-1")
};

BinRange scored_bins(const vgan_euka_db_view &db, uint32_t c) {
    const uint32_t b0 = db.bin_off[c], b1 = db.bin_off[c + 1];
    return {b0, b1 > b0 ? b1 - 1 : b0};
}

bool clade_detected(const vgan_euka_db_view &db, uint32_t c, const int32_t *count, const double *cov, const vgan_euka_detect_params &p) {
    const BinRange br = scored_bins(db, c);
    uint32_t informative = 0;
    int32_t empty = 0;
    for (uint32_t j = br.begin; j < br.end; ++j) {
        if (!(db.bin_entropy[j] > p.entropy_threshold)) continue;
        ++informative;
        // the reference stores the coverage in a vector<int> before counting zeros.  Coverages are sums of 1/#mappings,
        // never negative; anything that does not fit an int cannot be a zero either
        const double v = cov[j];
        if (v > -1.0 && v < 1.0) ++empty;
    }
    return !(empty > p.max_zero_bins || informative < p.min_bins || (uint32_t)count[c] < p.min_reads);
}

std::vector<std::string> split_lines(const char *txt, size_t n) {
    std::vector<std::string> out;
    const char *p = txt ? txt : "";
    while (out.size() < n) {
        const char *q = strchr(p, '\n');
        if (!q) {
            out.emplace_back(p);
            p += strlen(p);
        } else {
            out.emplace_back(p, q);
            p = q + 1;
        }
    }
    return out;
}

// linear-interpolation quantile of a sorted sample (miscfunc.h:239-251), evaluated in long double like the sample
long double quantile(const std::vector<long double> &sorted, double q) {
    const double pos = (double)(sorted.size() - 1) * q;
    const double below = std::floor(pos), above = std::ceil(pos);
    const double w = pos - below;
    return (1.0 - w) * sorted[(size_t)below] + w * sorted[(size_t)above];
}

struct Chain {
    int32_t n;
    const int64_t *n_like;
    const double *sum_log_like;

    // MCMC::get_proposal_likelihood: per clade a double, added up in long double
    long double log_likelihood(const std::vector<long double> &frac) const {
        long double total = 0.0L;
        for (int32_t c = 0; c < n; ++c) {
            if (n_like[c] == 0) continue; // the reference's inner loop is empty
            const double f = (double)frac[(size_t)c];
            total += (double)n_like[c] * std::log(f) + sum_log_like[c];
        }
        return total;
    }

    // MCMC::generate_proposal with branch_pos == false: one N(log x, alpha) draw per element from a freshly seeded
    // engine, a fresh distribution object per element (its cached second variate is never used), then softmax
    static std::vector<long double> propose(const std::vector<long double> &cur, double alpha, SeedSource &seeds) {
        std::mt19937 engine(seeds.next());
        std::vector<long double> e(cur.size());
        long double norm = 0.0L;
        for (size_t i = 0; i < cur.size(); ++i) {
            std::normal_distribution<long double> draw{std::log(cur[i]), alpha};
            e[i] = std::exp(draw(engine));
            norm += e[i];
        }
        for (auto &x : e) x /= norm;
        return e;
    }
};

// est: 5 numbers per clade {median, 15 %, 85 %, 5 %, 95 %}
int run_chain(const std::vector<long double> &init, const int64_t *n_like, const double *sum_log_like, int32_t iter, int32_t burnin,
              uint64_t seed, std::vector<long double> &est) {
    const int32_t n = (int32_t)init.size();
    if (n <= 0 || !n_like || !sum_log_like) return fail(VGAN_EINVAL, "vgan_euka_abundance_mcmc: null argument");
    if (burnin < 0 || (int64_t)iter - burnin - 1 <= 0)
        return fail(VGAN_EINVAL, "vgan_euka_abundance_mcmc: iter (%d) must exceed burnin (%d) + 1", iter, burnin);
    const Chain chain{n, n_like, sum_log_like};
    SeedSource seeds(seed);
    std::mt19937 accept_engine(seeds.next());
    std::vector<long double> state = init;
    long double state_ll = -9999999; // MCMC.cpp:1224
    const size_t kept = (size_t)(iter - burnin - 1);
    std::vector<std::vector<long double>> sample((size_t)n, std::vector<long double>());
    for (auto &s : sample) s.reserve(kept);
    for (int32_t it = 0; it < iter; ++it) {
        std::vector<long double> prop = Chain::propose(state, 0.1, seeds);
        if (it <= burnin) continue; // nothing is recorded, accepted or drawn before the burn-in ends (MCMC.cpp:1251-1260)
        const long double ll = chain.log_likelihood(prop);
        for (int32_t c = 0; c < n; ++c) sample[(size_t)c].push_back(prop[(size_t)c]); // the proposals, accepted or not
        const double accept = (double)std::min<long double>(1.0L, expl(ll - state_ll));
        std::uniform_real_distribution<> unit(0, 1);
        const double u = unit(accept_engine);
        if (u <= accept) {
            state_ll = ll;
            state.swap(prop);
        }
    }
    for (int32_t c = 0; c < n; ++c) {
        std::vector<long double> &s = sample[(size_t)c];
        std::sort(s.begin(), s.end());
        est.push_back(s[s.size() / 2]);
        for (double q : {0.15, 0.85, 0.05, 0.95}) est.push_back(quantile(s, q));
    }
    return VGAN_OK;
}

const char *const PROF_COLUMNS = "A>C\tA>G\tA>T\tC>A\tC>G\tC>T\tG>A\tG>C\tG>T\tT>A\tT>C\tT>G";

// Baseshift::display_prof (ends == "both"): the 12 off-diagonal substitution frequencies per profiled position;
// returns the C>T and G>A columns
struct EndRates {
    std::vector<double> c_to_t, g_to_a; // 2 * ltp entries: 5' positions then 3'
};

EndRates write_clade_profile(const uint32_t *counts, int ltp, const std::string &path) {
    std::ofstream f(path.c_str());
    f << PROF_COLUMNS << "\tPosition" << std::endl;
    EndRates r;
    for (int p = 0; p < 2 * ltp; ++p) {
        const uint32_t *row = counts + (size_t)p * 16;
        for (int from = 0; from < 4; ++from) {
            const double total = (double)row[4 * from] + (double)row[4 * from + 1] + (double)row[4 * from + 2] + (double)row[4 * from + 3];
            for (int to = 0; to < 4; ++to) {
                if (from == to) continue;
                const double rate = (double)row[4 * from + to] / total; // 0/0 prints as the reference's "-nan"
                f << std::setprecision(4) << rate << "\t";
                if (from == 1 && to == 3) r.c_to_t.push_back(rate);
                if (from == 2 && to == 0) r.g_to_a.push_back(rate);
            }
        }
        f << (p < ltp ? p : p - 2 * ltp) << std::endl;
        if (p == ltp - 1) f << PROF_COLUMNS << "\tPosition" << std::endl;
    }
    return r;
}

// Euka::get_avg over the concatenated per-clade halves
std::vector<double> position_means(const std::vector<double> &v, int ltp, size_t divisor) {
    std::vector<double> m((size_t)std::max(ltp, 0), 0.0);
    for (size_t i = 0; i < v.size(); ++i) m[i % (size_t)ltp] += v[i];
    for (auto &x : m) x /= (double)(int)divisor;
    return m;
}

void write_combined_profile(const std::string &path, const std::vector<double> &rates, int column) {
    std::ofstream f(path.c_str(), std::ios::trunc);
    f << PROF_COLUMNS << std::endl;
    for (double r : rates) {
        for (int col = 0; col < 11; ++col) {
            if (col == column) f << r;
            else f << 0;
            f << '\t';
        }
        f << 0 << std::endl;
    }
}

struct CladeLists { // Clade::inSize / nameStorage past their dummy first entry
    std::vector<std::vector<uint32_t>> in_size;
    std::vector<std::vector<std::pair<int64_t, int64_t>>> name_span;
};

void write_joined_sizes(std::ofstream &f, const std::string &name, const std::vector<uint32_t> &v) {
    f << name << '\t';
    for (size_t i = 0; i < v.size(); ++i) {
        if (i) f << '\t';
        f << v[i];
    }
    f << std::endl;
}

// bins without the last one, "<coverage>\t<entropy>" each, tab separated; trailing_tab reproduces Euka.cpp:693-702
void write_coverage_row(std::ofstream &f, const vgan_euka_results &r, uint32_t c, bool trailing_tab) {
    const BinRange br = scored_bins(*r.db, c);
    f << std::fixed << std::setprecision(5);
    for (uint32_t j = br.begin; j < br.end; ++j) {
        f << r.bin_cov[j] << '\t' << r.db->bin_entropy[j];
        if (trailing_tab || j + 1 != br.end) f << '\t';
    }
}

} // namespace

extern "C" int vgan_euka_detect(const vgan_euka_db_view *db, const int32_t *clade_count, const double *bin_cov,
                                const vgan_euka_detect_params *p, int32_t *ids, int32_t *n_ids) {
    if (!db || !clade_count || !bin_cov || !p || !ids || !n_ids) return fail(VGAN_EINVAL, "vgan_euka_detect: null argument");
    int32_t n = 0;
    for (uint32_t c = 0; c < db->n_clades; ++c)
        if (clade_detected(*db, c, clade_count, bin_cov, *p)) ids[n++] = db->clade_id[c];
    *n_ids = n;
    return VGAN_OK;
}

extern "C" int vgan_euka_abundance_mcmc(int32_t n, const double *init, const int64_t *n_like, const double *sum_log_like, int32_t iter,
                                        int32_t burnin, uint64_t seed, double *est) {
    if (n <= 0 || !init || !est) return fail(VGAN_EINVAL, "vgan_euka_abundance_mcmc: null argument");
    try {
        std::vector<long double> start(init, init + n), out;
        const int rc = run_chain(start, n_like, sum_log_like, iter, burnin, seed, out);
        if (rc == VGAN_OK)
            for (size_t i = 0; i < out.size(); ++i) est[i] = (double)out[i];
        return rc;
    } catch (const std::exception &e) {
        return fail(VGAN_ENOMEM, "vgan_euka_abundance_mcmc: %s", e.what());
    }
}

extern "C" int vgan_euka_report(const vgan_euka_results *r, const vgan_euka_report_cfg *cfg, const char *prefix, int32_t *detected,
                                int32_t *n_detected, double *estimates) {
    if (!r || !cfg || !prefix || !r->db || !r->clade_count || !r->baseshift || !r->bin_cov || !r->n_like || !r->sum_log_like)
        return fail(VGAN_EINVAL, "vgan_euka_report: null argument");
    if (r->n_reads > 0 && (!r->read_clade || !r->read_pass || !r->read_seq_len)) return fail(VGAN_EINVAL, "vgan_euka_report: per-read arrays missing");
    if (cfg->length_to_prof < 0 || cfg->length_to_prof > 32) return fail(VGAN_EINVAL, "vgan_euka_report: length_to_prof outside 0..32");
    try {
        const vgan_euka_db_view &db = *r->db;
        const uint32_t C = db.n_clades;
        const int ltp = cfg->length_to_prof;
        const std::vector<std::string> names = split_lines(db.clade_names, C);
        const std::string out = prefix, out_group = cfg->out_group ? cfg->out_group : "";
        // the reference looks clades up by Clade::id where it means the *.clade line
        auto line_of = [&](int32_t id) -> uint32_t {
            if (id < 0 || (uint32_t)id >= C) throw std::runtime_error("clade id " + std::to_string(id) + " is not a line of the clade table");
            return (uint32_t)id;
        };

        CladeLists lists;
        lists.in_size.resize(C);
        lists.name_span.resize(C);
        for (int64_t i = 0; i < r->n_reads; ++i) {
            const int32_t c = r->read_clade[i];
            if (c < 0 || !r->read_pass[i]) continue;
            if ((uint32_t)c >= C) return fail(VGAN_EINVAL, "vgan_euka_report: read %lld has clade %d of %u", (long long)i, c, C);
            lists.in_size[(size_t)c].push_back(r->read_seq_len[i]);
            if (r->name_off) lists.name_span[(size_t)c].emplace_back(r->name_off[i], r->name_off[i + 1]);
        }

        std::vector<char> is_detected(C);
        std::vector<int32_t> found; // readGAM3's clade_id_list
        for (uint32_t c = 0; c < C; ++c)
            if ((is_detected[c] = clade_detected(db, c, r->clade_count, r->bin_cov, cfg->detect))) found.push_back(db.clade_id[c]);
        int32_t out_group_id = -1;
        for (uint32_t c = 0; c < C; ++c)
            if (!out_group.empty() && names[c] == out_group) out_group_id = db.clade_id[c];

        if (cfg->out_frag && !found.empty()) {
            if (!out_group.empty()) {
                if (out_group_id < 0) return fail(VGAN_EINVAL, "vgan_euka_report: out group %s is not a clade", out_group.c_str());
                found.push_back(out_group_id); // Euka.cpp:561-569: from here on it counts as a detected clade
            }
            std::ofstream f((out + "_FragNames.tsv").c_str(), std::ios::trunc);
            for (int32_t id : found) {
                const uint32_t c = line_of(id);
                f << names[c] << '\t';
                const auto &spans = lists.name_span[c];
                for (size_t k = 0; k < spans.size(); ++k) {
                    if (k) f << '\t';
                    if (r->names) f.write(r->names + spans[k].first, spans[k].second - spans[k].first);
                }
                f << std::endl;
            }
        }

        // Euka::compute_init_vec
        std::vector<long double> init(found.size());
        {
            long double total = 0;
            for (int32_t id : found) total += r->clade_count[line_of(id)];
            for (size_t k = 0; k < found.size(); ++k) init[k] = (long double)r->clade_count[line_of(found[k])] / total;
        }
        const bool sampled = cfg->run_mcmc && found.size() >= 2;
        std::vector<long double> est;
        if (sampled) {
            std::vector<int64_t> nl(found.size());
            std::vector<double> sl(found.size());
            for (size_t k = 0; k < found.size(); ++k) {
                nl[k] = r->n_like[line_of(found[k])];
                sl[k] = r->sum_log_like[line_of(found[k])];
            }
            const int rc = run_chain(init, nl.data(), sl.data(), cfg->iter, cfg->burnin, cfg->seed, est);
            if (rc != VGAN_OK) return rc;
        } else {
            est.assign(found.size() * 5, 0.0L);
            for (size_t k = 0; k < found.size(); ++k) est[k * 5] = init[k];
        }
        if (n_detected) *n_detected = (int32_t)found.size();
        if (detected) std::copy(found.begin(), found.end(), detected);
        if (estimates)
            for (size_t k = 0; k < est.size(); ++k) estimates[k] = (double)est[k];

        std::ofstream cov((out + "_coverage.tsv").c_str(), std::ios::trunc), abund((out + "_abundance.tsv").c_str(), std::ios::trunc),
            surv((out + "_detected.tsv").c_str(), std::ios::trunc), sizes((out + "_inSize.tsv").c_str(), std::ios::trunc);
        if (!cov || !abund || !surv || !sizes) return fail(VGAN_EIO, "vgan_euka_report: cannot write %s_*.tsv", prefix);
        std::string header = "#Taxa\tdetected\tNumber_of_reads\tproportion_estimate";
        if (sampled)
            for (const char *ci : {"85", "95"})
                for (const char *side : {"lower", "higher"}) header += std::string("\t") + ci + "%_confidence_interval_" + side + "_bound";
        abund << header << '\n';
        surv << header << '\n';
        cov << "#Taxa";
        for (int b = 0; b < 21; ++b) cov << '\t' << "bin" << b << '\t' << "entropy";
        cov << std::endl;

        std::vector<uint32_t> shown; // the detected clades met so far ("clade_list_id")
        for (uint32_t c = 0; c < C; ++c) {
            if (!is_detected[c]) {
                abund << names[c] << "\tno\t" << r->clade_count[c];
                for (int z = 0; z < (sampled ? 5 : 1); ++z) abund << '\t' << 0;
                abund << std::endl;
                if (!out_group.empty() && names[c] == out_group) {
                    cov << names[c] << '\t';
                    write_coverage_row(cov, *r, c, false);
                    cov << std::endl;
                    write_joined_sizes(sizes, names[c], lists.in_size[c]);
                }
                continue;
            }
            shown.push_back(c);
            cov << names[c] << '\t';
            write_coverage_row(cov, *r, c, !sampled);
            if (sampled) cov << std::endl;
            else cov << '\n';
            write_joined_sizes(sizes, names[c], lists.in_size[c]);
            std::ostringstream row; // default formatting: 6 significant digits of a long double
            row << names[c] << "\tyes\t" << r->clade_count[c] << '\t';
            if (sampled) {
                // five numbers of the shown.size()-th detected clade; only the very last entry of the result has no tab
                const size_t k0 = (shown.size() - 1) * 5;
                for (size_t k = k0; k < k0 + 5; ++k) {
                    row << est.at(k);
                    if (k != est.size() - 1) row << '\t';
                }
            } else {
                // Euka.cpp:717-724 prints init_vec[0 .. number of detected clades so far)
                for (size_t k = 0; k < shown.size(); ++k) {
                    if (k) row << '\t';
                    row << init.at(k);
                }
            }
            abund << row.str() << std::endl;
            surv << row.str() << std::endl;
        }

        if (cfg->out_dir && *cfg->out_dir) {
            struct stat sb;
            if (stat(cfg->out_dir, &sb) != 0 || !S_ISDIR(sb.st_mode)) (void)mkdir(cfg->out_dir, 0777);
        }
        std::vector<double> ct5, ga3;
        for (uint32_t shown_c : shown) {
            const uint32_t c = line_of(db.clade_id[shown_c]); // looked up again by Clade::id (Euka.cpp:757-766)
            const EndRates e = write_clade_profile(r->baseshift + (size_t)c * 2 * ltp * 16, ltp, out + "_" + names[c] + ".prof");
            ct5.insert(ct5.end(), e.c_to_t.begin(), e.c_to_t.begin() + ltp);
            ga3.insert(ga3.end(), e.g_to_a.begin() + ltp, e.g_to_a.end());
        }
        if (out_group_id >= 0) {
            const uint32_t c = line_of(out_group_id);
            (void)write_clade_profile(r->baseshift + (size_t)c * 2 * ltp * 16, ltp, out + "_" + names[c] + ".prof");
        }
        write_combined_profile(out + "_5p.prof", position_means(ct5, ltp, found.size()), 5);
        std::vector<double> ga = position_means(ga3, ltp, found.size());
        std::reverse(ga.begin(), ga.end());
        write_combined_profile(out + "_3p.prof", ga, 6);
        return VGAN_OK;
    } catch (const std::exception &e) {
        return fail(VGAN_EINVAL, "vgan_euka_report: %s", e.what());
    }
}
