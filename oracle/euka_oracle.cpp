/*
 * oracle/euka_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * euka per-read two-model likelihood with ancient-DNA damage, restated for the CPU from the reference
 * (paths relative to /root/reference/src):
 *   readGAM_Euka.h:67-577   the per-alignment lambda (clade lookup, per-column models, clade_like, pass filter,
 *                           bin coverage)
 *   damage.cpp:18-36,41-323 Damage::combineDeamRates / initDeamProbabilities
 *   miscfunc.h:68-136       substitutionRates / diNucleotideProb, readNucSubstitionRatesFreq
 *   baseshift.cpp:57-88     Baseshift::baseshift_calc
 *   Euka.cpp:38-51,446-486  qscore_vec, base_freq, t_T_ratio, rare_bases
 *   load.cpp:71-157         load_clade_chunks, load_clade_info
 * "parity unpinned": the reference holds no numeric test for this path (src/test.cpp:1002-1172 asserts detected
 * taxa and abundance windows end to end) and cannot be built here; anchored on closed forms in
 * tests/test_euka_oracle.py.  libgab (oplusInitnatl, dimer2indexInt, allTokens) restated from published semantics.
 * Undefined behaviour of the reference and the oracle's definition:
 *   - a.quality()[m] with m >= size: quality 0;  Lseq outside 15..1000 or damage position outside [0,Lseq): the read is
 *     skipped and counted;  unassigned base_freq / t_T_ratio / rare_bases entries are zero / false;
 *   - baseshift positions outside the strings and bases outside ACGT are skipped.
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

using std::string;
using std::vector;

namespace {

struct substitutionRates {
    long double s[12];
};
struct probSubstition {
    long double s[16];
};
struct diNucleotideProb {
    long double p[4][4];
};

/* libgab dimer2indexInt (not in tree): index into the 12-column profile order A>C A>G A>T C>A C>G C>T G>A G>C G>T T>A T>C T>G */
inline int dimer2indexInt(int n1, int n2) { return n1 * 3 + (n2 > n1 ? n2 - 1 : n2); }

vector<string> all_tokens(const string &line, char sep) {
    vector<string> t;
    size_t p = 0;
    while (true) {
        size_t q = line.find(sep, p);
        if (q == string::npos) {
            t.push_back(line.substr(p));
            break;
        }
        t.push_back(line.substr(p, q - p));
        p = q + 1;
    }
    return t;
}

/* miscfunc.h:84-136 */
void readNucSubstitionRatesFreq(const string &text, vector<substitutionRates> &subVec) {
    std::istringstream in(text);
    string line;
    if (!std::getline(in, line)) throw std::runtime_error("empty profile");
    vector<string> fields = all_tokens(line, '\t');
    if (fields.size() == 13) fields.pop_back();
    if (fields.size() != 12) throw std::runtime_error("header has " + std::to_string(fields.size()) + " fields rather than 12");
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        fields = all_tokens(line, '\t');
        if (fields.size() == 13) fields.pop_back();
        if (fields.size() != 12) throw std::runtime_error("line has " + std::to_string(fields.size()) + " fields rather than 12");
        substitutionRates t;
        for (unsigned k = 0; k < 12; ++k) t.s[k] = strtold(fields[k].c_str(), nullptr);
        subVec.emplace_back(t);
    }
}

struct Damage {
    static constexpr unsigned MINLENGTHFRAGMENT = 15, MAXLENGTHFRAGMENT = 1000; /* damage.h:42-43 */
    vector<probSubstition> sub5p, sub3p;
    vector<diNucleotideProb> sub5pDiNuc, sub3pDiNuc;

    /* damage.cpp:18-36 */
    static void combineDeamRates(const long double f1[4], const long double f2[4], long double f[4], int b) {
        const long double minFreq = std::min(f1[b], f2[b]);
        if (f1[b] == minFreq) {
            for (int i = 0; i < 4; i++) f[i] = f1[i];
        } else {
            for (int i = 0; i < 4; i++) f[i] = f2[i];
        }
    }

    static void build_end(const vector<substitutionRates> &subT, vector<probSubstition> &sub, vector<diNucleotideProb> &di) {
        for (unsigned i = 0; i < subT.size(); i++) { /* damage.cpp:66-88 */
            probSubstition toadd;
            for (int nuc1 = 0; nuc1 < 4; nuc1++) {
                double probIdentical = 1.0;
                for (int nuc2 = 0; nuc2 < 4; nuc2++) {
                    const int nuc = nuc1 * 4 + nuc2;
                    if (nuc1 == nuc2) continue;
                    const int ind2 = dimer2indexInt(nuc1, nuc2);
                    probIdentical = probIdentical - subT[i].s[ind2];
                    toadd.s[nuc] = subT[i].s[ind2];
                }
                if (probIdentical < 0) throw std::runtime_error("identity probability is less than 0");
                toadd.s[nuc1 * 4 + nuc1] = probIdentical;
            }
            sub.emplace_back(toadd);
        }
        if (sub.empty()) throw std::runtime_error("profile has no rows");
        for (unsigned i = (unsigned)(sub.size() - 1); i < MAXLENGTHFRAGMENT; i++) sub.emplace_back(sub[sub.size() - 1]); /* :91-93 */
        for (unsigned i = 0; i < MAXLENGTHFRAGMENT; i++) { /* :96-105 */
            diNucleotideProb d;
            for (int n1 = 0; n1 < 4; n1++)
                for (int n2 = 0; n2 < 4; n2++) d.p[n1][n2] = sub[i].s[n1 * 4 + n2];
            di.emplace_back(d);
        }
    }

    /* damage.cpp:41-258; an empty profile text = "no file given" (:47-55): zero rates for every position */
    void init(const string &p5, const string &p3) {
        vector<substitutionRates> sub5pT, sub3pT;
        substitutionRates zero;
        for (auto &x : zero.s) x = 0.0L;
        if (!p5.empty()) readNucSubstitionRatesFreq(p5, sub5pT);
        else sub5pT.resize(MAXLENGTHFRAGMENT, zero);
        if (!p3.empty()) readNucSubstitionRatesFreq(p3, sub3pT);
        else sub3pT.resize(MAXLENGTHFRAGMENT, zero);
        build_end(sub5pT, sub5p, sub5pDiNuc);
        build_end(sub3pT, sub3p, sub3pDiNuc);
    }

    /* subDeamDiNuc[L][l] (damage.cpp:238-258), computed on demand instead of materialising ~500k matrices */
    diNucleotideProb at(unsigned L, unsigned l) const {
        diNucleotideProb d;
        for (int b1 = 0; b1 < 4; b1++) combineDeamRates(sub5pDiNuc[l].p[b1], sub3pDiNuc[L - l - 1].p[b1], d.p[b1], b1);
        return d;
    }
};

inline long double oplusnatl(long double x, long double y) {
    if (x > y) return x + log1pl(expl(y - x));
    return y + log1pl(expl(x - y));
}
inline long double oplusInitnatl(long double x, long double y) {
    if (x == 0) return y;
    return oplusnatl(x, y);
}

struct Tables {
    vector<double> qscore_vec;
    double base_freq[256];
    double t_T_ratio[256][256];
    bool rare_bases[256];
    Tables() {
        for (int Q = 0; Q < 100; ++Q) qscore_vec.emplace_back(Q >= 2 ? pow(10, ((-1 * Q) * 0.1)) : 0.25); /* Euka.cpp:38-51 */
        memset(base_freq, 0, sizeof base_freq);
        memset(t_T_ratio, 0, sizeof t_T_ratio);
        memset(rare_bases, 0, sizeof rare_bases);
        base_freq['A'] = log(0.362815); /* Euka.cpp:446-450 */
        base_freq['C'] = log(0.207743);
        base_freq['G'] = log(0.116809);
        base_freq['N'] = log(0.25);
        base_freq['T'] = log(0.312435);
        const char *b = "ACGT";
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) t_T_ratio[(int)b[i]][(int)b[j]] = i == j ? 1.0 : 0.02381; /* :453-468 */
        t_T_ratio['A']['G'] = t_T_ratio['G']['A'] = t_T_ratio['C']['T'] = t_T_ratio['T']['C'] = 0.95238;
        for (const char *p = "WMKRYBDHV"; *p; ++p) rare_bases[(int)*p] = true; /* :472-486 */
    }
};

inline int qidx(int q) { return q < 0 ? 0 : (q > 99 ? 99 : q); }

} // namespace

extern "C" {

void *orc_damage_create(const char *prof5, const char *prof3) {
    try {
        auto d = new Damage();
        d->init(prof5 ? prof5 : "", prof3 ? prof3 : "");
        return d;
    } catch (const std::exception &) {
        return nullptr;
    }
}
void orc_damage_free(void *d) { delete (Damage *)d; }

/* subDeamDiNuc[L][l].p as 16 long doubles / doubles, and the padded 5'/3' rows (for the product's compact tables) */
int orc_damage_matrix(const void *dmg, uint32_t L, uint32_t l, double *out16) {
    const Damage *d = (const Damage *)dmg;
    if (L < Damage::MINLENGTHFRAGMENT || L > Damage::MAXLENGTHFRAGMENT || l >= L) return -1;
    const diNucleotideProb m = d->at(L, l);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out16[i * 4 + j] = (double)m.p[i][j];
    return 0;
}

int orc_euka_run(const orc_graph_t *g, const orc_alnset_t *a, const orc_euka_db *db, const void *dmg,
                 const orc_euka_params *prm, orc_euka_out *o) {
    static const Tables T;
    const Damage &D = *(const Damage *)dmg;
    const int lengthToProf = prm->lengthToProf;
    std::vector<char> gs(1 << 16), rs(1 << 16);
    std::vector<int32_t> sz(1 << 16);
    o->n_bad = 0;
    for (int64_t r = 0; r < a->n_reads; ++r) {
        o->read_clade[r] = -1;
        o->read_in[r] = o->read_out[r] = o->read_like[r] = o->read_not_like[r] = 0.0;
        o->read_pass[r] = 0;
        if (a->identity[r] == 0) continue; /* readGAM_Euka.h:72 */
        const int64_t m0 = a->map_off[r], m1 = a->map_off[r + 1];
        if (m1 == m0) { /* mapping()[0] of an empty path: UB */
            o->n_bad++;
            continue;
        }
        const int n_index = (int)a->m_node[m0]; /* :99 */
        int c_n = 0;
        for (int i = 0; i < db->n_clades; i++) { /* :120-140: the last (clade, bin) containing the node wins */
            for (int j = db->bin_off[i]; j < db->bin_off[i + 1]; ++j) {
                if (n_index >= db->bin_lo[j] && n_index <= db->bin_hi[j]) c_n = i;
            }
        }
        const double pair_dist = db->clade_dist[c_n]; /* :152 */
        int64_t lens[3];
        const int rc = orc_reconstruct(g, a, r, gs.data(), rs.data(), sz.data(), (int64_t)gs.size(), lens);
        const unsigned Lseq = (unsigned)(a->seq_off[r + 1] - a->seq_off[r]);
        if (rc != 0 || Lseq < Damage::MINLENGTHFRAGMENT || Lseq > Damage::MAXLENGTHFRAGMENT) {
            o->n_bad++;
            continue;
        }
        const string graph_seq(gs.data(), (size_t)lens[0]), read_seq(rs.data(), (size_t)lens[1]);
        const string quality(a->qual + a->qual_off[r], a->qual + a->qual_off[r + 1]);
        auto qual_at = [&](unsigned m) -> int { return m < quality.size() ? (int)quality[m] : 0; }; /* Q15 */
        auto rd = [&](unsigned m) -> char { return m < read_seq.size() ? read_seq[m] : '\0'; };

        /* per-column models, :228-464 (checked before the counters are touched so a bad read leaves no trace) */
        double in_clade_lik = 0.0, not_in_clade_lik = 0.0;
        double log_lik = 0.0, log_lik_2 = 0.0;
        int softclip_count = 0;
        unsigned n = 0;
        const bool isrev = a->m_rev[m0] != 0;
        if (isrev) n = Lseq - 1;
        bool bad = false;
        for (unsigned m = 0; m < graph_seq.size(); m++) {
            const unsigned char G = (unsigned char)graph_seq[m], R = (unsigned char)rd(m);
            if (G == 'N' || R == 'N') { /* :236-241 */
                log_lik = T.base_freq[R];
                log_lik_2 = T.base_freq[R];
            } else if (G == '-' || R == '-') { /* :244-249 */
                log_lik = log(0.002);
                log_lik_2 = log(0.2);
            } else if (T.rare_bases[G] || T.rare_bases[R]) { /* :252-257 */
                log_lik = log((1 - pair_dist) * 0.001);
                log_lik_2 = log(0.001);
            } else if (G == 'S' || R == 'S') { /* :263-280 */
                const int base_quality = qidx(qual_at(m));
                ++softclip_count;
                if (softclip_count % 3 == 0) log_lik = log(1 - T.qscore_vec[base_quality]);
                else log_lik = log(T.qscore_vec[base_quality] / 3);
                log_lik_2 = log(0.25);
            } else {
                const int base_quality = qidx(qual_at(m));
                if (n >= Lseq) {
                    bad = true;
                    break;
                }
                double probBasePreDamage[4];
                for (int bpo = 0; bpo < 4; bpo++) { /* :312-318 */
                    if ("ACGT"[bpo] == (char)G) probBasePreDamage[bpo] = (1 - pair_dist);
                    else probBasePreDamage[bpo] = pair_dist * T.t_T_ratio[G][(int)"ACGT"[bpo]];
                }
                double probBasePostDamage[4] = {0, 0, 0, 0};
                const diNucleotideProb M = D.at(Lseq, n);
                for (int bpd = 0; bpd < 4; bpd++)
                    for (int bpo = 0; bpo < 4; bpo++) probBasePostDamage[bpd] += probBasePreDamage[bpo] * M.p[bpo][bpd]; /* :337-340 */
                double log_lik_marg = 0.0;
                for (int bpd = 0; bpd < 4; bpd++) { /* :385-394 */
                    if ("ACGT"[bpd] == (char)R)
                        log_lik_marg = (double)oplusInitnatl(log_lik_marg, log(probBasePostDamage[bpd] * (1 - T.qscore_vec[base_quality])));
                    else
                        log_lik_marg = (double)oplusInitnatl(log_lik_marg, log(probBasePostDamage[bpd] * (T.qscore_vec[base_quality] / 3)));
                }
                log_lik = log_lik_marg;
                if (G == R) log_lik_2 = log(1 - 0.25536); /* :420-443 */
                else log_lik_2 = log(0.25536);
            }
            in_clade_lik += log_lik;
            not_in_clade_lik += log_lik_2;
            if (R != '-' && !isrev) n++; /* :457-461 */
            else if (R != '-' && isrev) n--;
        }
        if (bad) {
            o->n_bad++;
            continue;
        }
        /* Baseshift::baseshift_calc, baseshift.cpp:57-88 (applies to every mapped read, :173) */
        for (int p = 0; p < lengthToProf * 2; p++) {
            int64_t gi, ri;
            if (p < lengthToProf) {
                gi = ri = p;
            } else {
                const int pos = -(lengthToProf * 2) + p;
                gi = (int64_t)graph_seq.length() + pos;
                ri = (int64_t)read_seq.length() + pos;
            }
            if (gi < 0 || ri < 0 || gi >= (int64_t)graph_seq.size() || ri >= (int64_t)read_seq.size()) continue;
            const char gb = (char)toupper(graph_seq[(size_t)gi]), rb = (char)toupper(read_seq[(size_t)ri]);
            if (gb == 'S' || rb == 'S' || gb == 'I' || rb == 'I' || gb == '-' || rb == '-' || gb == 'N' || rb == 'N') continue;
            const char *acgt = "ACGT";
            const char *pg = strchr(acgt, gb), *pr = strchr(acgt, rb);
            if (!pg || !pr || !gb || !rb) continue; /* dna2int of other letters is uninitialised in the reference */
            o->baseshift[((size_t)c_n * 2 * lengthToProf + p) * 16 + (size_t)(pg - acgt) * 4 + (size_t)(pr - acgt)]++;
        }
        /* :485-492 */
        const double map_q = (1 - pow(10, ((-1 * a->mapq[r]) * 0.1)));
        const double like = map_q * exp((in_clade_lik) - (double)oplusInitnatl(in_clade_lik, not_in_clade_lik));
        o->read_clade[r] = c_n;
        o->read_in[r] = in_clade_lik;
        o->read_out[r] = not_in_clade_lik;
        o->read_like[r] = like;
        o->read_not_like[r] = 1 - like;
        /* :504-549 */
        if (((in_clade_lik) - (not_in_clade_lik) > 1) && ((unsigned)a->mapq[r] > prm->MINIMUMMQ)) {
            o->read_pass[r] = 1;
            o->clade_count[c_n]++;
            const int64_t nm = m1 - m0;
            for (int64_t i = m0; i < m1; ++i) {
                const int n_id = (int)a->m_node[i];
                for (int j = db->bin_off[c_n]; j < db->bin_off[c_n + 1]; ++j) {
                    if (n_id >= db->bin_lo[j] && n_id <= db->bin_hi[j]) o->bin_cov[j] += 1.0 / (double)nm;
                }
            }
        }
    }
    return 0;
}

/* load.cpp:81-88: "name (lo hi entropy)x" per line; lo/hi parsed with stoi ("1836.0" -> 1836) */
int64_t orc_load_clade_chunks(const char *txt, int32_t *bin_off, int32_t *lo, int32_t *hi, double *entropy, int64_t cap_clades,
                              int64_t cap_bins) {
    std::istringstream in(txt);
    string line;
    int64_t c = 0, nb = 0;
    bin_off[0] = 0;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        vector<string> tok;
        string t;
        while (ls >> t) tok.push_back(t);
        for (size_t j = 1; j + 2 < tok.size() + 1; j += 3) { /* the reference reads tokens[j+2] unconditionally */
            if (nb >= cap_bins) return -1;
            lo[nb] = std::stoi(tok[j]);
            hi[nb] = std::stoi(tok[j + 1]);
            entropy[nb] = std::stod(tok[j + 2]);
            ++nb;
        }
        if (c >= cap_clades) return -1;
        bin_off[++c] = (int32_t)nb;
    }
    return c;
}

/* load.cpp:118-152: "id name dist nPaths snode enode" */
int64_t orc_load_clade_info(const char *txt, int32_t *id, double *dist, int32_t *npaths, int32_t *snode, int32_t *enode,
                            char *names, int64_t names_cap, int64_t cap) {
    std::istringstream in(txt);
    string line, all;
    int64_t c = 0;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        vector<string> tok;
        string t;
        while (ls >> t) tok.push_back(t);
        if (tok.size() != 6) continue; /* the reference asserts 6 tokens */
        if (c >= cap) return -1;
        id[c] = std::stoi(tok[0]);
        dist[c] = std::stod(tok[2]);
        npaths[c] = std::stoi(tok[3]);
        snode[c] = std::stoi(tok[4]);
        enode[c] = std::stoi(tok[5]);
        all += tok[1] + "\n";
        ++c;
    }
    if ((int64_t)all.size() + 1 > names_cap) return -1;
    memcpy(names, all.c_str(), all.size() + 1);
    return c;
}

} /* extern "C" */
