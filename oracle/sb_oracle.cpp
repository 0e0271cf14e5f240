/*
 * oracle/sb_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * soibean per-read x per-path likelihood (analyse_GAM) and the per-MCMC-iteration refresh, restated for the CPU
 * from the reference (paths relative to /root/reference/src):
 *   getLCAfromGAM.h:92-560   per read, per edit-level segment, per path: supported / unsupported base terms,
 *                            pathMap (sum) and detailMap (per-base {readBase, referenceBase, pathSupport, logLikelihood})
 *   MCMC.h:111-296           computeBaseLogLike (HKY substitution term; kappa = 1/22 = 0, Q13)
 *   MCMC.h:299-315           calculateLogWeightedAverage
 *   MCMC.cpp:738-993         the per-iteration loop over reads (k = 1 and k > 1 branches)
 * "parity unpinned": the reference's soibean tests are statistical end-to-end asserts (src/test.cpp:226-333) and
 * need the databases + vg giraffe; this restatement is anchored on closed forms in tests/test_sb_cpu.py.
 * Quirks reproduced: quality indexed by the within-segment position; the damage matrix is taken at
 * subDeamDiNuc[|graph_seq|][baseIX] with baseIX constant within a segment; the supported marginal adds
 * log(post[b]) for every b whatever the read base (so it is log(sum post) clamped to log(0.9999999));
 * reverse-strand reads are sliced from the end backwards; edit-level segments beyond the number of mappings are
 * unsupported by every path; path names longer than 101 characters never match.
 * Oracle definitions where the reference is undefined: quality()[s] past the end = 0; a read whose |graph_seq| is
 * outside 15..1000, whose baseIX falls outside the damage table, or on which substr() would throw is skipped and
 * counted; the reference's runtime_error guards (nan / inf / positive values) make the call return an error status.
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

using std::string;
using std::vector;

extern "C" int orc_damage_matrix(const void *dmg, uint32_t L, uint32_t l, double *out16);

namespace {

struct BaseInfo {
    char readBase;
    char referenceBase;
    bool pathSupport;
    double logLikelihood;
};

struct SbRead {
    bool ok = false;
    vector<double> pathMap;                             /* [P] */
    vector<vector<vector<BaseInfo>>> detail;            /* [P][segment][base] */
};

struct SbHandle {
    int32_t n_paths = 0;
    vector<SbRead> reads;
    int64_t n_bad = 0;
};

inline long double oplusnatl(long double x, long double y) {
    if (x > y) return x + log1pl(expl(y - x));
    return y + log1pl(expl(x - y));
}
inline long double oplusInitnatl(long double x, long double y) {
    if (x == 0) return y;
    return oplusnatl(x, y);
}

struct Guard : std::runtime_error {
    explicit Guard(const char *m) : std::runtime_error(m) {}
};

inline void check_ll(double v, const char *what) {
    if (std::isnan(v) || std::isinf(v) || v > 1e-8) throw Guard(what);
}

vector<double> qscore_vec() { /* soibean.cpp:131-155 == Euka::get_qscore_vec */
    vector<double> q;
    for (int Q = 0; Q < 100; ++Q) q.emplace_back(Q >= 2 ? pow(10, ((-1 * Q) * 0.1)) : 0.25);
    return q;
}
inline int qidx(int q) { return q < 0 ? 0 : (q > 99 ? 99 : q); }

} // namespace

extern "C" {

void *orc_sb_analyse(const orc_graph_t *g, const orc_alnset_t *a, const uint8_t *path_findable, const void *dmg, int PENALTY,
                     int64_t *n_bad_out) {
    static const vector<double> qs = qscore_vec();
    const int P = g->n_paths;
    auto H = new SbHandle();
    H->n_paths = P;
    H->reads.resize((size_t)a->n_reads);
    std::vector<char> gsb(1 << 16), rsb(1 << 16);
    std::vector<int32_t> szb(1 << 16);
    for (int64_t r = 0; r < a->n_reads; ++r) {
        SbRead &out = H->reads[(size_t)r];
        if (a->identity[r] == 0) continue; /* getLCAfromGAM.h:101 */
        const int64_t m0 = a->map_off[r], nM = a->map_off[r + 1] - m0;
        int64_t lens[3];
        if (nM == 0 || orc_reconstruct(g, a, r, gsb.data(), rsb.data(), szb.data(), (int64_t)gsb.size(), lens) != 0) {
            H->n_bad++;
            continue;
        }
        const string graph_seq(gsb.data(), (size_t)lens[0]), read_seq(rsb.data(), (size_t)lens[1]);
        const vector<int> mppg_sizes(szb.begin(), szb.begin() + lens[2]);
        const string quality(a->qual + a->qual_off[r], a->qual + a->qual_off[r + 1]);
        const int64_t seq_size = a->seq_off[r + 1] - a->seq_off[r];
        const bool rev = a->m_rev[m0] != 0;
        int baseIX = rev ? (int)seq_size - 1 : 0; /* :107 */
        int baseOnRead = baseIX;
        const int Lseq = (int)graph_seq.size();
        if (Lseq < 15 || Lseq > 1000) { /* subDeamDiNuc[Lseq] is empty / out of range */
            H->n_bad++;
            continue;
        }
        vector<double> pathMap((size_t)P, 0.0);
        vector<vector<vector<BaseInfo>>> detail((size_t)P);
        bool bad = false;
        try {
            for (size_t i = 0; i < mppg_sizes.size() && !bad; ++i) {
                /* :150-173: which paths go through the node of this segment */
                int64_t nID = -1;
                if (mppg_sizes.size() != (size_t)nM) {
                    if (i >= mppg_sizes.size() - (mppg_sizes.size() - (size_t)nM)) nID = -1; /* "No_support" */
                    else nID = a->m_node[m0 + (int64_t)i];
                } else {
                    nID = a->m_node[m0 + (int64_t)i];
                }
                if (nID >= 0 && (nID < g->min_id || nID > g->max_id)) throw std::out_of_range("nodepaths.at"); /* :163 */
                string nodeSeq, partReadSeq;
                int startIndex = 0;
                if (rev) { /* :179-182 */
                    startIndex = (baseIX - mppg_sizes.at(i) - 1 >= 0) ? (baseIX - mppg_sizes.at(i) - 1) : 0;
                    nodeSeq = graph_seq.substr((size_t)startIndex, (size_t)mppg_sizes.at(i));
                    partReadSeq = read_seq.substr((size_t)startIndex, (size_t)mppg_sizes.at(i));
                } else {
                    nodeSeq = graph_seq.substr((size_t)baseIX, (size_t)mppg_sizes.at(i));
                    partReadSeq = read_seq.substr((size_t)baseIX, (size_t)mppg_sizes.at(i));
                }
                if (!nodeSeq.empty() && (baseIX < 0 || baseIX >= Lseq)) { /* subDeamDiNuc[Lseq][baseIX] out of range */
                    bad = true;
                    break;
                }
                double M16[16];
                bool haveM = false;
                for (int m = 0; m < P; ++m) {
                    vector<BaseInfo> readInfo;
                    const bool supported = nID >= 0 && path_findable[m] && g->pathsgo[(size_t)nID * P + m] != 0; /* :194 */
                    auto qual_at = [&](size_t s) { return qidx(s < quality.size() ? (int)quality[s] : 0); }; /* Q12 */
                    auto rd = [&](size_t s) { return s < partReadSeq.size() ? partReadSeq[s] : '\0'; };
                    if (supported) {
                        for (size_t s = 0; s < nodeSeq.size(); ++s) {
                            const int base_quality = qual_at(s);
                            const char G = nodeSeq[s], R = rd(s);
                            BaseInfo info;
                            info.readBase = R;
                            info.referenceBase = G;
                            info.pathSupport = false;
                            if (G == 'N' || R == 'N') { /* :227-245 */
                                info.logLikelihood = log(0.25);
                                check_ll(info.logLikelihood, "N");
                            } else if (G == 'S' || R == 'S') { /* :246-263 */
                                info.logLikelihood = log((qs[base_quality] / 3));
                                check_ll(info.logLikelihood, "Softclip");
                            } else if (G == '-' || R == '-') { /* :264-283 */
                                info.logLikelihood = log(0.02);
                                check_ll(info.logLikelihood, "GAP");
                            } else { /* :284-400 */
                                double pre[4], post[4] = {0, 0, 0, 0};
                                for (int bpo = 0; bpo < 4; bpo++) pre[bpo] = ("ACGT"[bpo] == G) ? 1 - qs[base_quality] : (qs[base_quality] / 3);
                                if (!haveM) {
                                    if (orc_damage_matrix(dmg, (uint32_t)Lseq, (uint32_t)baseIX, M16) != 0) throw std::out_of_range("subDeamDiNuc");
                                    haveM = true;
                                }
                                for (int bpd = 0; bpd < 4; bpd++)
                                    for (int bpo = 0; bpo < 4; bpo++) post[bpd] += pre[bpo] * M16[bpo * 4 + bpd]; /* :327 */
                                double log_lik_marg = -std::numeric_limits<double>::infinity();
                                for (int bpd = 0; bpd < 4; bpd++) log_lik_marg = (double)oplusInitnatl(log_lik_marg, log(post[bpd])); /* :338-348 */
                                if (log_lik_marg > log(0.9999999)) log_lik_marg = log(0.9999999); /* :349-351 */
                                check_ll(log_lik_marg, "calculated log like");
                                info.pathSupport = true;
                                info.logLikelihood = log_lik_marg;
                                if (rev && R != '-') baseOnRead--; /* :394-399 */
                                else if (!rev && R != '-') baseOnRead++;
                            }
                            pathMap[(size_t)m] += info.logLikelihood;
                            readInfo.push_back(info);
                        }
                    } else {
                        for (size_t s = 0; s < nodeSeq.size(); ++s) {
                            const int base_quality = qual_at(s);
                            const char G = nodeSeq[s], R = rd(s);
                            BaseInfo info;
                            info.readBase = '-';
                            info.referenceBase = G;
                            info.pathSupport = false;
                            if (G == 'N' || R == 'N') info.logLikelihood = log(0.25);
                            else if (G == 'S' || R == 'S') info.logLikelihood = log((qs[base_quality] / 3));
                            else if (G == '-' || R == '-') info.logLikelihood = log(0.02);
                            else if (std::abs(baseOnRead) % PENALTY == 0) info.logLikelihood = log(1 - (qs[base_quality])); /* :473-490 */
                            else info.logLikelihood = log((qs[base_quality] / 3));
                            check_ll(info.logLikelihood, "no sup");
                            pathMap[(size_t)m] += info.logLikelihood;
                            readInfo.push_back(info);
                            if (rev && R != '-') baseOnRead--; /* :515-519 */
                            else if (!rev && R != '-') baseOnRead++;
                        }
                    }
                    detail[(size_t)m].push_back(readInfo); /* :527 */
                    baseOnRead = baseIX;                    /* :528-532 */
                }
                if (rev) baseIX = startIndex; /* :537-544 */
                else baseIX += mppg_sizes.at(i);
                baseOnRead = baseIX;
            }
        } catch (const std::out_of_range &) {
            bad = true;
        } catch (const Guard &) {
            bad = true;
        }
        if (bad) {
            H->n_bad++;
            continue;
        }
        out.ok = true;
        out.pathMap = pathMap;
        out.detail = detail;
    }
    if (n_bad_out) *n_bad_out = H->n_bad;
    return H;
}

void orc_sb_free(void *h) { delete (SbHandle *)h; }

int orc_sb_read_ok(const void *h, int64_t r) { return ((const SbHandle *)h)->reads[(size_t)r].ok ? 1 : 0; }

int orc_sb_pathmap(const void *h, int64_t r, double *out) {
    const SbHandle *H = (const SbHandle *)h;
    const SbRead &R = H->reads[(size_t)r];
    if (!R.ok) return -1;
    for (int p = 0; p < H->n_paths; ++p) out[p] = R.pathMap[(size_t)p];
    return 0;
}

/* mostProbPath of every read (getLCAfromGAM.h:563-579): best[r] = the path with the highest pathMap value when it is the only
 * one, -1 on ties / skipped reads; sig_count[p] = frequencies[path] of soibean.cpp:655-668; returns gam->size() */
int64_t orc_sb_best_paths(const void *h, int32_t *best, int64_t *sig_count) {
    const SbHandle *H = (const SbHandle *)h;
    for (int p = 0; p < H->n_paths; ++p) sig_count[p] = 0;
    int64_t n = 0;
    for (size_t r = 0; r < H->reads.size(); ++r) {
        const SbRead &R = H->reads[r];
        best[r] = -1;
        if (!R.ok) continue;
        ++n;
        long double highestValue = std::numeric_limits<long double>::lowest();
        for (double v : R.pathMap) highestValue = std::max(highestValue, (long double)v);
        vector<int> keysWithHighestValue;
        for (int p = 0; p < H->n_paths; ++p)
            if (R.pathMap[(size_t)p] == highestValue) keysWithHighestValue.push_back(p);
        if (keysWithHighestValue.size() == 1) {
            best[r] = keysWithHighestValue[0];
            sig_count[keysWithHighestValue[0]]++;
        }
    }
    return n;
}

/* soibean.cpp:737-756: the initial log-likelihood of a set of source paths (n == 1: the caller passes log_freq = 0 for the
 * plain sum of :744-747) */
double orc_sb_mixture_loglike(const void *h, int32_t n, const int32_t *paths, double log_freq) {
    const SbHandle *H = (const SbHandle *)h;
    double logLike = 0.0L;
    for (const SbRead &R : H->reads) {
        if (!R.ok) continue;
        double inter = log_freq + R.pathMap[(size_t)paths[0]];
        for (int j = 1; j < n; ++j) inter = oplusInitnatl(inter, (log_freq + R.pathMap[(size_t)paths[j]]));
        logLike += inter;
    }
    return logLike;
}

/* counts of (referenceBase, readBase) over the path-supported bases of (read r, path p): 5x5, index ACGT or 4 = other;
 * plus the number of bases in total (test aid for the factorised device layout) */
int orc_sb_counts(const void *h, int64_t r, int32_t p, uint32_t *out25, uint32_t *n_bases) {
    const SbHandle *H = (const SbHandle *)h;
    const SbRead &R = H->reads[(size_t)r];
    if (!R.ok) return -1;
    memset(out25, 0, 25 * sizeof(uint32_t));
    uint32_t n = 0;
    auto idx = [](char c) { const char *s = "ACGT"; const char *q = c ? strchr(s, c) : nullptr; return q ? (int)(q - s) : 4; };
    for (const auto &seg : R.detail[(size_t)p])
        for (const BaseInfo &b : seg) {
            ++n;
            if (b.pathSupport) out25[idx(b.referenceBase) * 5 + idx(b.readBase)]++;
        }
    if (n_bases) *n_bases = n;
    return 0;
}

/* MCMC.h:111-296 with params.freqs = {A, C, G, T, R, Y, M} */
static double computeBaseLogLike(const BaseInfo &detail, const double *freqs7, const double t, const double branch_len) {
    const double kappa = 1 / 22; /* MCMC.h:66: integer division, 0 */
    auto F = [&](char c) -> double { return c == 'A' ? freqs7[0] : c == 'C' ? freqs7[1] : c == 'G' ? freqs7[2] : c == 'T' ? freqs7[3] : 0.0; };
    const double purinfreq = freqs7[4], pyrinfreq = freqs7[5], mu = freqs7[6];
    const char refb = detail.referenceBase, readb = detail.readBase;
    double probBaseHKY[4] = {0, 0, 0, 0};
    for (int bpo = 0; bpo < 4; bpo++) {
        const char rb = "ACGT"[bpo];
        if (rb == refb) {
            if (rb == 'A' || rb == 'G') {
                const double A = 1 + purinfreq * (kappa - 1);
                const double jut1 = F(rb) + F(rb) * ((1 / purinfreq) - 1) * exp(-(mu * t));
                const double jut11 = ((purinfreq - F(rb)) / purinfreq) * exp(-(mu * t * A));
                probBaseHKY[bpo] = jut1 + jut11;
            } else {
                const double A = 1 + pyrinfreq * (kappa - 1);
                const double jut1 = F(rb) + F(rb) * ((1 / pyrinfreq) - 1) * exp(-(mu * t));
                const double jut11 = ((pyrinfreq - F(rb)) / pyrinfreq) * exp(-(mu * t * A));
                probBaseHKY[bpo] = jut1 + jut11;
            }
        } else {
            if ((rb == 'A' && refb == 'G') || (rb == 'G' && refb == 'A')) {
                const double A = 1 + purinfreq * (kappa - 1);
                const double jut1 = F(rb) + F(rb) * ((1 / purinfreq) - 1) * exp(-(mu * t));
                const double jut11 = (F(rb) / purinfreq) * exp(-(mu * t * A));
                probBaseHKY[bpo] = jut1 > jut11 ? jut1 - jut11 : jut11 - jut1;
            } else if ((rb == 'C' && refb == 'T') || (rb == 'T' && refb == 'C')) {
                const double A = 1 + pyrinfreq * (kappa - 1);
                const double jut1 = F(rb) + F(rb) * ((1 / pyrinfreq) - 1) * exp(-(mu * t));
                const double jut11 = (F(rb) / pyrinfreq) * exp(-(mu * t * A));
                probBaseHKY[bpo] = jut1 > jut11 ? jut1 - jut11 : jut11 - jut1;
            } else {
                probBaseHKY[bpo] = F(rb) * (1 - exp(-(mu * t)));
            }
        }
        if (probBaseHKY[bpo] < 1e-8) probBaseHKY[bpo] = 1e-8;
        if (std::isnan(probBaseHKY[bpo]) || std::isinf(probBaseHKY[bpo]) || probBaseHKY[bpo] < 1e-8) throw Guard("HKY is invalid");
    }
    double log_lik_marg = -std::numeric_limits<double>::infinity();
    for (int bpd = 0; bpd < 4; bpd++) {
        if ("ACGT"[bpd] == readb) log_lik_marg = (double)oplusInitnatl(log_lik_marg, (log(probBaseHKY[bpd]) + log((1 - branch_len))));
        else log_lik_marg = (double)oplusInitnatl(log_lik_marg, (log(probBaseHKY[bpd]) + log((branch_len / 3))));
    }
    if (log_lik_marg > 1e-8) log_lik_marg = log(0.999999999);
    check_ll(log_lik_marg, "HKY loglikemarg");
    log_lik_marg = log_lik_marg + detail.logLikelihood;
    check_ll(log_lik_marg, "HKY loglikemarg");
    return log_lik_marg;
}

/* MCMC.h:299-315 */
static double calculateLogWeightedAverage(double logValueChild, double weightChild, double logValueParent, double weightParent) {
    const double maxLogValue = std::max(logValueChild + std::log(weightChild), logValueParent + std::log(weightParent));
    const double logSumExp = maxLogValue + std::log(std::exp(logValueChild + std::log(weightChild) - maxLogValue) +
                                                    std::exp(logValueParent + std::log(weightParent) - maxLogValue));
    const double logWeightSum = std::log(weightChild + weightParent);
    if (std::isinf(logWeightSum)) return -std::numeric_limits<double>::infinity();
    return logSumExp - logWeightSum;
}

/* MCMC.cpp:738-993 for one state: k sources with child path, parent path, branch length of the child node, position on
 * the branch, proportion.  Returns 0 and *logLike, or -20 when one of the reference's guards would throw. */
int orc_sb_loglike(const void *h, int32_t k, const int32_t *child, const int32_t *parent, const double *dist, const double *pos,
                   const double *theta, double con, const double *freqs7, int n_threads, double *logLike_out) {
    const SbHandle *H = (const SbHandle *)h;
    double logLike = 0.0;
    int fail = 0;
    const int64_t R = (int64_t)H->reads.size();
#pragma omp parallel for num_threads(n_threads > 0 ? n_threads : 1) reduction(+ : logLike)
    for (int64_t r = 0; r < R; ++r) {
        const SbRead &read = H->reads[(size_t)r];
        if (!read.ok) continue;
        try {
            auto branch_ll = [&](int path, double t) {
                double v = 0.0;
                for (const auto &seg : read.detail[(size_t)path])
                    for (const BaseInfo &b : seg) {
                        if (b.pathSupport) v += computeBaseLogLike(b, freqs7, t, con);
                        else v += b.logLikelihood;
                        if (v > 0 || std::isnan(v) || std::isinf(v)) throw Guard("Log-likelihood is nan");
                    }
                return v;
            };
            if (k == 1) {
                double t = dist[0];
                if (t == 0.0) t = 0.00001;
                const double t1 = pos[0] * t, t2 = t - t1;
                const double readLogLike = branch_ll(child[0], t2), readLogLikeP = branch_ll(parent[0], t1);
                logLike += calculateLogWeightedAverage(readLogLike, pos[0], readLogLikeP, (1 - pos[0])); /* :868 */
            } else {
                double inter = -std::numeric_limits<double>::infinity();
                for (int y = 0; y < k; ++y) {
                    double t = dist[y];
                    if (t == 0.0) t = 0.00001;
                    const double t1 = pos[y] * t, t2 = t - t1;
                    const double readLogLike = branch_ll(child[y], t2), readLogLikeP = branch_ll(parent[y], t1);
                    const double interc2 = log(pos[y]) + readLogLike;
                    const double interp2 = log((1 - pos[y])) + readLogLikeP;
                    const double inter2 = (double)oplusnatl(interc2, interp2);
                    inter = (double)oplusInitnatl(inter, (inter2 + log(theta[y]))); /* :967-974 */
                }
                logLike += inter;
            }
        } catch (const Guard &) {
#pragma omp atomic write
            fail = 1;
        }
    }
    *logLike_out = logLike;
    return fail ? -20 : 0;
}

} /* extern "C" */
