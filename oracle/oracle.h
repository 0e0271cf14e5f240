/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of vgan's per-read likelihood hot path (HaploCart / euka / soibean), written
 * from the reference's arithmetic (including its behaviour-defining quirks Q1..Q15).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / reported CPU baseline -- never as part of the product path.
 *
 * Parity pinning:
 *   - a1 (reconstruct_graph_sequence) is pinned by the reference's own 10 reconstruction KATs
 *     (src/test.cpp:855-994) on test/reconstructInputSeq/{target_graph.gfa,test_reads.gam}:
 *     tests/test_oracle_golden.py checks all 20 strings.
 *   - The numeric log-likelihoods / posteriors are NOT pinned by any reference test
 *     (src/test.cpp holds classification-level asserts only) and the reference cannot be built
 *     here (vg/libbdsg/libgab/protobuf absent, fetched from the network by src/Makefile:102-128):
 *     for those values this oracle says "parity unpinned"; it is anchored on closed-form KATs
 *      checked with mpmath in tests/test_oracle_kat.py.
 *   - libgab (unpinned git HEAD, src/Makefile:121) is absent: oplusnatl / oplusInitnatl /
 *     isValidDNA are restated from their published semantics.
 *
 * Layouts are plain flat arrays so numpy can fill them through ctypes.
 */
#ifndef VGAN_ORACLE_H
#define VGAN_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Graph as the reference sees it after readPathHandleGraph (src/readPathHandleGraph.cpp:14-37). */
typedef struct orc_graph {
    int64_t min_id, max_id;        /* graph.min_node_id()/max_node_id() */
    const int64_t *node_seq_off;   /* [max_id-min_id+2] offsets into node_seq, index = id-min_id */
    const char *node_seq;          /* forward-strand node sequences, concatenated */
    int32_t n_paths;               /* nbpaths */
    const uint8_t *pathsgo;        /* [(max_id+1)*n_paths] row = NODE ID (path_supports[i][j], :28-31) */
    const int32_t *pangenome_base; /* [max_id+1] pangenome_map.at(to_string(id)), <0 = key absent */
    const double *mappability;     /* load_mappabilities() vector (src/load.cpp:6-24) */
    int64_t n_mappability;
} orc_graph_t;

/* Decoded GAM: what readGAM keeps per vg::Alignment (src/readGAM.h:37-49). */
typedef struct orc_alnset {
    int64_t n_reads;
    const int64_t *seq_off;   /* [n_reads+1] */
    const char *seq;          /* a.sequence() */
    const int64_t *qual_off;  /* [n_reads+1] */
    const char *qual;         /* a.quality(): raw phred bytes */
    const int32_t *mapq;      /* a.mapping_quality() */
    const double *identity;   /* a.identity() */
    const int64_t *map_off;   /* [n_reads+1] -> mappings */
    const int64_t *m_node;    /* position.node_id */
    const int64_t *m_offset;  /* position.offset */
    const uint8_t *m_rev;     /* position.is_reverse */
    const int64_t *edit_off;  /* [n_mappings+1] -> edits */
    const int32_t *e_from;    /* edit.from_length */
    const int32_t *e_to;      /* edit.to_length */
    const int64_t *e_seq_off; /* [n_edits+1] */
    const char *e_seq;        /* edit.sequence */
} orc_alnset_t;

typedef struct orc_hc_params {
    double background_error_prob;      /* -e, default 0.0001 (src/HaploCart.cpp:70) */
    int32_t use_background_error_prob; /* true for -f FASTA input (src/HaploCart.cpp:399) */
    int32_t is_consensus_fasta;
} orc_hc_params_t;

/* status codes (the reference would std::terminate / read out of bounds in these cases) */
enum {
    ORC_OK = 0,
    ORC_ERR_NODE = -1,     /* node id not in graph / pangenome_map.at() throws */
    ORC_ERR_SUBSTR = -2,   /* std::string::substr / insert out_of_range */
    ORC_ERR_SIZES = -3,    /* mppg_sizes[i] past its end (mapping without edits) */
    ORC_ERR_TABLE = -4,    /* mappabilities[] index out of range */
    ORC_ERR_CAP = -5       /* caller buffer too small */
};

/* a1: reconstruct_graph_sequence (src/vgan_utils.h:6-79). lens[0..2] = |graph_seq|, |read_seq|, #sizes */
int orc_reconstruct(const orc_graph_t *g, const orc_alnset_t *a, int64_t r,
                    char *graph_seq, char *read_seq, int32_t *sizes, int64_t cap, int64_t *lens);

/* a2+a3 for one read starting from a zero vector (src/update_likelihood.cpp:19-53,
 * src/process_mapping.cpp:26-91); out[n_paths] as long double. flags bit0: a Q/mapq >= 100 was clamped. */
int orc_hc_read(const orc_graph_t *g, const orc_alnset_t *a, int64_t r, const orc_hc_params_t *p,
                long double *out, int32_t *flags);

/* Per-segment scalars of one read (test aid): S_m (supported sum), U_m (unsupported sum), node id.
 * Derived by calling the literal a3 on a graph view with one always-supported and one never-supported path. */
int orc_hc_read_segments(const orc_graph_t *g, const orc_alnset_t *a, int64_t r, const orc_hc_params_t *p,
                         double *S, double *U, int64_t *node, int64_t cap, int64_t *n_seg);

/* a8: the OpenMP loop of src/HaploCart.cpp:408-421 over reads [r0,r1). final_ld/final_d: [n_paths].
 * faithful=1 restates the reference literally (unsupported penalty recomputed per path, vectors by value);
 * faithful=0 is the "hoisted" variant (S_m/U_m once per mapping) -- NOT the reference, same results.
 * n_bad receives the number of reads whose processing would have terminated the reference. */
int orc_hc_run(const orc_graph_t *g, const orc_alnset_t *a, int64_t r0, int64_t r1, const orc_hc_params_t *p,
               int n_threads, int faithful, long double *final_ld, double *final_d, int64_t *n_bad);

/* a9: get_posterior (src/get_posterior.cpp:36-127). Text inputs use the sidecar formats
 * (graph_paths: one name per line; parents.txt/children.txt: "name tok tok ..." per line).
 * Output: '\t'-joined "clade\tconfidence(%.17g)\tdepth" records separated by '\n' into out (cap bytes);
 * conf[] receives the confidences as double; returns number of records or <0.
 * Followed literally: one child set per recursion level (:51-76), so a path reachable at two depths of
 * children.txt is summed twice.  Where the reference is undefined the definition is:
 *   - an ancestor whose levels contain no path name (sum_log_likelihoods reads v[0] of an empty vector, :78-85):
 *     the sum of nothing is 0, i.e. confidence = exp(0 - total);
 *   - a name absent from children.txt (get_children dereferences end(), :41): no children;
 *   - a cyclic children.txt (unbounded recursion): the walk stops after 100000 levels. */
int orc_hc_posterior(const long double *final_vec, int32_t n_paths, const char *path_names_txt,
                     const char *parents_txt, const char *children_txt, const char *predicted,
                     char *out, int64_t cap, double *conf, int32_t conf_cap);

/* helpers restated from src/miscfunc.h:180-212, src/haplocart_functions.cpp:81-107, libgab */
double orc_p_seq_error(int Q);
double orc_qscore(int Q);
double orc_p_incorrect_mapping(int Q);
double orc_background_freq(char c);
long double orc_oplusnatl(long double x, long double y);
long double orc_oplusInitnatl(long double x, long double y);
long double orc_p_obs_base(int pangenome_base, double epsilon, int generations);

/* sidecar loaders restated from src/load.cpp:6-58,283-345 (text already inflated by the caller). */
int64_t orc_load_mappabilities(const char *txt, double *out, int64_t cap);
int64_t orc_load_pangenome_map(const char *txt, int32_t *base_by_id, int64_t cap); /* fills base_by_id[node]=coord+1 */
int64_t orc_load_path_supports(const char *txt, int32_t n_paths, uint8_t *out, int64_t cap_rows);

/* ------------------------------------------------------------------ euka (oracle/euka_oracle.cpp) */
typedef struct orc_euka_db {
    int32_t n_clades;
    const double *clade_dist;  /* [n_clades] Clade::dist of line c (consumers index c*6+1, src/load.cpp:118-152) */
    const int32_t *bin_off;    /* [n_clades+1] */
    const int32_t *bin_lo;     /* stoi of "1836.0"-style tokens (src/load.cpp:81-88) */
    const int32_t *bin_hi;
    const double *bin_entropy;
} orc_euka_db;

typedef struct orc_euka_params {
    uint32_t MINIMUMMQ;   /* --minMQ, default 29 (src/Euka.cpp:152-190) */
    int32_t lengthToProf; /* -l, default 5 */
} orc_euka_params;

typedef struct orc_euka_out {
    int32_t *read_clade;        /* [n_reads] c_n, -1 = read skipped (identity 0, or a read the oracle defines as bad) */
    double *read_in, *read_out; /* in_clade_lik, not_in_clade_lik */
    double *read_like, *read_not_like; /* the values pushed to Clade::clade_like / clade_not_like */
    uint8_t *read_pass;
    int32_t *clade_count;       /* [n_clades] Clade::count */
    uint32_t *baseshift;        /* [n_clades][2*lengthToProf][16] */
    double *bin_cov;            /* [n_bins] get<3>(chunks[c][j]) */
    int64_t n_bad;
} orc_euka_out;

void *orc_damage_create(const char *prof5_text, const char *prof3_text); /* NULL on a malformed profile; "" = no file */
void orc_damage_free(void *d);
int orc_damage_matrix(const void *dmg, uint32_t L, uint32_t l, double *out16); /* subDeamDiNuc[L][l].p row-major */
/* output arrays must be zeroed by the caller (counts / coverages accumulate) */
int orc_euka_run(const orc_graph_t *g, const orc_alnset_t *a, const orc_euka_db *db, const void *dmg,
                 const orc_euka_params *prm, orc_euka_out *o);
int64_t orc_load_clade_chunks(const char *txt, int32_t *bin_off, int32_t *lo, int32_t *hi, double *entropy,
                              int64_t cap_clades, int64_t cap_bins);
int64_t orc_load_clade_info(const char *txt, int32_t *id, double *dist, int32_t *npaths, int32_t *snode, int32_t *enode,
                            char *names, int64_t names_cap, int64_t cap);

/* what `vgan euka` does with that result (oracle/euka_abundance_oracle.cpp): detected clades (readGAM_Euka.h:582-630),
 * compute_init_vec, the abundance MCMC (MCMC.cpp:1095-1366) and every output file (Euka.cpp:540-1160) written to
 * <prefix>_*.  seed = 0 draws from std::random_device as the reference does; otherwise successive rd() calls are replaced
 * by the high halves of a splitmix64 stream started at seed.  detected[] / estimates[5 per detected clade] may be NULL. */
typedef struct orc_euka_report_cfg {
    uint32_t MINNUMOFBINS, MINNUMOFREADS; /* --minBins 6, --minFrag 10 (Euka.cpp:183-184) */
    int32_t MAXIMUMOFBINS;                /* --maxBins 0 */
    double ENTROPY_SCORE_THRESHOLD;       /* --entropy 1.17 */
    int32_t lengthToProf;
    int32_t run_mcmc, iter, burnin;       /* --no-mcmc, --iter 10000, --burnin 100 */
    uint64_t seed;
    int32_t outFrag;
    const char *outGroup, *out_dir;       /* NULL or "" = not given */
} orc_euka_report_cfg;
int orc_euka_report(const orc_euka_db *db, const int32_t *clade_id, const char *clade_names, const orc_euka_out *res,
                    int64_t n_reads, const int32_t *read_seq_len, const char *names, const int64_t *name_off,
                    const orc_euka_report_cfg *cfg, const char *prefix, int32_t *detected, int32_t *n_detected,
                    double *estimates);

/* ------------------------------------------------------------------ soibean (oracle/sb_oracle.cpp) */
/* analyse_GAM (src/getLCAfromGAM.h:31-732): g->pathsgo[node][p] = path p goes through the node (nodepaths);
 * path_findable[p] = 0 for path names longer than 101 characters (never matched, :80-88).  Returns a handle holding
 * pathMap / detailMap of every read. */
void *orc_sb_analyse(const orc_graph_t *g, const orc_alnset_t *a, const uint8_t *path_findable, const void *dmg, int PENALTY,
                     int64_t *n_bad);
void orc_sb_free(void *h);
int orc_sb_read_ok(const void *h, int64_t r);
int orc_sb_pathmap(const void *h, int64_t r, double *out /* [n_paths] */);
int64_t orc_sb_best_paths(const void *h, int32_t *best /* [n_reads] */, int64_t *sig_count /* [n_paths] */);
double orc_sb_mixture_loglike(const void *h, int32_t n, const int32_t *paths, double log_freq);
int orc_sb_counts(const void *h, int64_t r, int32_t p, uint32_t *out25, uint32_t *n_bases);
/* one likelihood refresh of the tree-placement MCMC (src/MCMC.cpp:738-993, src/MCMC.h:111-315);
 * freqs7 = {A, C, G, T, R, Y, M}; dist[y] = branch length of the child node */
int orc_sb_loglike(const void *h, int32_t k, const int32_t *child, const int32_t *parent, const double *dist, const double *pos,
                   const double *theta, double con, const double *freqs7, int n_threads, double *logLike);

/* `vgan soibean` after analyse_GAM (oracle/sb_chain_oracle.cpp): for k = 1..n_sig sources the initial log-likelihood and, with
 * run_mcmc, `chains` chains of MCMC::run_tree_proportion, MCMC::processMCMCiterations per chain and the R-hat diagnostics
 * (soibean.cpp:738-944); writes <prefix>Result<k><c>.mcmc, <prefix>Trace<k><c>.detail.mcmc (gzip), <prefix>ProportionEstimates<k>.txt,
 * <prefix>BranchEstimate<k>.txt, <prefix>Diagnostics<k>0.txt.  path_names: one per line, index = path of the handle;
 * sig_nodes: tree node numbers (pre-order of the Newick text). */
typedef struct orc_sb_estimate_cfg {
    uint32_t max_iter, burn, chains;
    uint64_t seed;
    double con;
    double freqs7[7];
    int32_t run_mcmc;
} orc_sb_estimate_cfg;
int orc_sb_estimate(const void *h, const char *newick, const char *path_names, const int32_t *sig_nodes, int32_t n_sig,
                    const orc_sb_estimate_cfg *cfg, const char *prefix);

#ifdef __cplusplus
}
#endif
#endif
