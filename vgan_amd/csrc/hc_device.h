// Device-side views shared by the HaploCart kernels (hc_kernels.hip) and the C-ABI layer (hc_capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vgan {

struct HcNodeDev {
    double mappability; // mappabilities[pangenome_base[node]]         (src/process_mapping.cpp:33-35)
    double match;       // pow(1 - 30*mu(pangenome_base[node]), 8)     (src/get_p_obs_base.cpp:44-64)
};

// HBM layout of the graph side:
//   umask     uint64 [rows][row_words]  UNSUPPORTED-path bitmask: bit p of row i = !path_supports[i][p], zero beyond
//             P; row_words = ceil(P/64) rounded up to the sweep tile so a wave's tile is one aligned scalar burst.
//   node_tab  {mappability, match} per node id: the two per-node scalars the likelihood needs.
//   lq        log(p_seq_error((int8)byte)) for every raw quality byte (src/miscfunc.h:180-188, process_mapping.cpp:12)
//   qscore    qscore_vec[100] (src/miscfunc.h:199-212);  incmap: incorrect_mapping_vec[100]
struct HcGraphDev {
    const uint64_t *umask;
    const HcNodeDev *node_tab;
    const double *lq;
    const double *qscore;
    const double *incmap;
    uint32_t rows;
    uint32_t row_words;
    uint32_t n_paths;
};

struct HcBatchDev {
    uint32_t n_reads, n_segments;
    const uint32_t *read_seg_off, *read_col_off, *read_qual_off;
    const uint16_t *read_algn_len;
    const uint8_t *read_mapq;
    const uint32_t *seg_node;
    const uint16_t *seg_start, *seg_len;
    const uint8_t *graph_seq, *algnseq, *qual;
};

struct HcParamsDev {
    double bep;
    int use_bep;
    int consensus;
};

void launch_hc_segments(const HcGraphDev &g, const HcBatchDev &b, const HcParamsDev &prm, double *segS, double *segU,
                        double *segD, double *nodeW, double *totals, hipStream_t st);
void launch_hc_sweep(const HcGraphDev &g, const uint32_t *item_node, const double *D, uint32_t n_items, int skip_zero,
                     double *acc, hipStream_t st);
void launch_hc_finish(const double *totals, const double *acc_seg, const double *acc_node, uint32_t n_paths, double *out,
                      hipStream_t st);
void launch_hc_read_loglik(const HcGraphDev &g, const HcBatchDev &b, const double *segS, const double *segU, double *out,
                           hipStream_t st);
void launch_hc_posterior(const double *final_vec, uint32_t n_paths, const uint64_t *sets, uint32_t set_words,
                         uint32_t n_sets, double *conf, hipStream_t st);

constexpr uint32_t HC_SWEEP_TILE_WORDS = 16;

} // namespace vgan
