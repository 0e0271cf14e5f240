// Device-side views shared by the HaploCart kernels (hc_kernels.hip) and the C-ABI layer (hc_capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vgan_gpu.h" // the packed layout's constants (VGAN_HC_CREC_HEAD, VGAN_HC_MAPQ_MAJOR)

namespace vgan {

struct alignas(32) HcNodeDev {
    // the tiled kernel's factorised column term (hc_kernels.hip): log and reciprocal of the node's share of
    // wobs = pcm * match, taken once on the host in long double.  First, so that one 16-byte load fetches both.
    double ln_w;        // log(mappability * match); log(match) in a consensus-FASTA context
    double inv_mm;      // 1 / (mappability * match)
    double mappability; // mappabilities[pangenome_base[node]]         (src/process_mapping.cpp:33-35)
    double match;       // pow(1 - 30*mu(pangenome_base[node]), 8)     (src/get_p_obs_base.cpp:44-64)
};

// HBM layout of the graph side:
//   umask     uint64 [rows][mask_words]  UNSUPPORTED-path bitmask: bit p of row i = !path_supports[i][p], zero
//             beyond P (plain layout; debug kernel only).
//   umaskT    uint16 [rows][n_tiles][64]  the same bits, transposed per tile for the sweep: the ceil(P/64) mask words
//             are dealt as evenly as possible to n_tiles = 8*ceil(W/127) tiles (tile t owns words
//             tile_word0[t] .. tile_word0[t+1]-1, i.e. tile_base_words or one more); entry [row][t][l] holds, from bit
//             15 downwards, bit l of each of the tile's words.  One wave-wide 2-byte load (128 B, coalesced) gives
//             every lane the membership bits of "its" path in all words of the tile.  With tile = blockIdx % 8 each
//             XCD's L2 sees one eighth of the table (1.5 MB for the hcfiles shape).
//   node_tab  {ln_w, inv_mm, mappability, match} per node id: the per-node scalars the likelihood needs.
//   lq        log(p_seq_error((int8)byte)) for every raw quality byte (src/miscfunc.h:180-188, process_mapping.cpp:12)
//   qscore    qscore_vec[100] (src/miscfunc.h:199-212);  incmap: incorrect_mapping_vec[100]
//   rdtab     per mapping quality {1 - incmap, log(1 - incmap), 1 / (1 - incmap)}: the read's share of pcm = (1 - p_inc) *
//             mappability (process_mapping.cpp:41); in a consensus-FASTA context the log is log(1 - background_error_prob)
struct HcGraphDev {
    const uint64_t *umask;
    const uint16_t *umaskT;
    const uint16_t *tile_word0; // [n_tiles+1]
    const HcNodeDev *node_tab;
    const double *lq;
    const double *qscore;
    const double *incmap;
    const double *rdtab; // [100][3]
    // node classes (hc_col8_kernels.hip): the distinct {ln_w, inv_mm, mappability} triples of node_tab, most frequent first
    const uint16_t *node_hi;  // [rows] class * (bytes of a class in the kernel's table of column terms) for the classes the table
                              // covers, 0xE000 | class for the others
    const HcNodeDev *cls_tab; // [n_cls]
    uint32_t n_cls;           // 0: more than HC_EXT_NODE_CLASSES classes (the kernel is not taken)
    const double *col_memo;   // the context's table of column terms (hc_col8_memo_kernel), or NULL
    const double *col_memo2;  // ... and its wide form: every class of the graph, every quality below HC_EXT_QMAX (hc_col8_memo2_kernel), or NULL
    uint32_t rows;
    uint32_t mask_words;
    uint32_t row_entries; // n_tiles * 64
    uint32_t n_tiles;
    uint32_t tile_base_words;
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

// The tileable reads of a batch in the layout the wave kernel streams (hc_wave_kernels.hip), written once per batch by
// the layout pass launch_hc_pack -- byte moves only: no comparison, clamp or table lookup happens there.
//   rhdr   uint4 [n_reads + 1]   {seg_off, qual_off, col_off, |algnseq| | mapq << 16}; entry n_reads holds the end offsets
//   srec   uint32 [segments]     VGAN_HC_SREC(node id, seg_start, read index): bits 0-17, 18-28, 29-31 (the index & 7)
//   crec   uint32 [columns]      one record per alignment column of the read, at the column's own position col_off + c:
//                                byte 0 graph_seq[col_off + c], byte 1 algnseq[col_off + c - seg_start] (the read bases are
//                                taken from the read start: update_likelihood.cpp:46), byte 2 qual[qual_off + c] (0 beyond the
//                                quality string), bit 31 set on the first column of a segment; columns no segment scores are 0
//   qualp  uint8 [quality bytes + 32]  the quality strings, zero padded so that any aligned 8-byte word can be read whole
struct HcPackedDev {
    const uint4 *rhdr;
    const uint32_t *srec;
    const uint32_t *crec;
    const uint8_t *qualp;
    uint32_t n_reads;       // tileable reads packed
    uint32_t n_segments;    // upper bounds of the three streams (the batch's totals)
    uint64_t n_cols;
    uint64_t n_qual;
    uint32_t max_read_segs; // over the packed reads: select the kernel variant
    uint32_t max_read_qual;
    uint32_t max_read_cols;
    uint32_t max_read_node_span; // 0: not known
    uint32_t qual_excess;        // largest (quality length - columns) of a read, 0 when no quality string outruns its read's columns
                                 // (the contract of a packed VIEW; a batch packed from SoA arrays may break it: such a batch
                                 // keeps to the kernels that read qualp)
};

// what another translation unit needs of a context (hc_flatten_kernels.hip)
struct HcCtxInfo {
    int device;
    hipStream_t stream;
    uint32_t rows;
};
} // namespace vgan
struct vgan_hc_ctx;
namespace vgan {
HcCtxInfo hc_ctx_info(const vgan_hc_ctx *c);

struct HcParamsDev {
    double bep;
    int use_bep;
    int consensus;
};

// totals: sum of S_m (and of U_m) over everything accumulated, kept in HC_TOTAL_SLOTS partial sums 128 bytes apart -- every
// wave of a segment kernel ends with one add, and tens of thousands of adds to ONE address serialise in the L2 (~14 ns each,
// measured: 65 536 of them set the length of a 0.6 ms launch to 0.9 ms).  Slot s at totals[s * HC_TOTAL_STRIDE + {0, 1}].
constexpr uint32_t HC_TOTAL_SLOTS = 64;
constexpr uint32_t HC_TOTAL_STRIDE = 16; // doubles

constexpr uint32_t HC_MAX_NODE_CLASSES = 32;   // node classes hc_segment_col8_kernel keeps the scalars of in LDS (the others': the context's array)
constexpr uint32_t HC_EXT_NODE_CLASSES = 256;  // node classes the kernel takes at all (the wide table of column terms covers every one of them)
constexpr uint32_t HC_EXT_QMAX = 94;           // quality values the wide table covers: [0, HC_EXT_QMAX) -- printable FASTQ qualities end at 93
constexpr uint32_t HC_MEMO_CLASSES = 16;       // ... of which its table of column terms covers the most frequent (C8_NMEMO)
constexpr uint32_t HC_MEMO_CLASS_BYTES = 3072; // a class's part of that table (C8_CLS_BYTES)

// per-read limits of the LDS-tiled segment kernel (the tile contract of include/vgan_gpu.h; flatten.cpp applies them)
// (a read has to fit one LDS tile; every phase of the kernel is flat over the tile, so there is no smaller per-read bound)
constexpr uint32_t HC_TILE_MAX_READ_COLS = 1280;
constexpr uint32_t HC_TILE_MAX_READ_QUAL = 1280;
constexpr uint32_t HC_TILE_MAX_READ_SEGS = 512;

// reads [0, n_tileable) go through the LDS-tiled kernel, the rest -- or everything when S_m / U_m are asked for
// separately -- through the general one.  segD (or NULL): D_m = S_m - U_m per segment.  nodeW (or NULL): W[node] += D_m,
// through a workgroup's LDS window over the node ids of its reads in the tiled kernel.
void launch_hc_segments(const HcGraphDev &g, const HcBatchDev &b, const HcParamsDev &prm, uint32_t n_tileable,
                        uint32_t mean_cols_per_read, double *segS, double *segU, double *segD, double *nodeW, double *totals,
                        hipStream_t st);
// layout pass: packs reads [0, n_tileable) of b into the caller's buffers (sized from the batch totals: n_tileable + 1
// headers, n_segments records, n_cols column records, n_qual + 32 quality bytes).  maxima (device, 3 words,
// or NULL) receives max segments / quality bytes / columns per packed read.
void launch_hc_pack(const HcBatchDev &b, uint32_t n_tileable, uint64_t n_cols, uint64_t n_qual, uint4 *rhdr, uint32_t *srec,
                    uint32_t *crec, uint8_t *qualp, uint32_t *maxima, hipStream_t st);
// true when the wave kernel is the faster route for reads of that shape (otherwise the LDS-tiled kernel takes the batch)
bool hc_wave_kernel_fits(uint32_t max_read_segs, uint32_t max_read_qual, uint32_t max_read_cols, uint32_t mean_read_segs,
                         uint32_t mean_read_cols);
// work_ctr: one device word, zero when first used, that only this context's launches touch; *work_base: its value when the
// next launch starts (kept by the caller between launches on the one stream)
void launch_hc_segments_wave(const HcGraphDev &g, const HcPackedDev &pk, const HcParamsDev &prm, double *segD, double *nodeW,
                             double *totals, uint32_t *work_ctr, uint32_t *work_base, hipStream_t st);
// the same for node-weights accumulation alone (W[node] += D_m and the totals), eight columns to a lane with a table of column
// terms (hc_col8_kernels.hip); hc_col8_kernel_fits: the batches and graphs it takes
bool hc_col8_kernel_fits(const HcGraphDev &g, const HcPackedDev &pk);
size_t hc_col8_memo_doubles(); // the context's table of column terms: its size, and the launch that fills it (after the graph side is up)
void launch_hc_col8_memo(const HcGraphDev &g, const HcParamsDev &prm, double *out, hipStream_t st);
size_t hc_col8_memo2_doubles(uint32_t n_cls); // the wide table: [100 mapping qualities][n_cls][HC_EXT_QMAX][match][read base]
void launch_hc_col8_memo2(const HcGraphDev &g, const HcParamsDev &prm, double *out, hipStream_t st);
void launch_hc_segments_col8(const HcGraphDev &g, const HcPackedDev &pk, const HcParamsDev &prm, double *nodeW, double *totals, hipStream_t st);
// node ids of the packed segment records into a plain array (the per-segment mask sweep reads them eight at a time)
void launch_hc_srec_nodes(const uint32_t *srec, uint32_t n_segments, uint32_t *out, hipStream_t st);
// reads [r_begin, n_reads) through the general kernel (one wave per read, any length)
void launch_hc_segments_general(const HcGraphDev &g, const HcBatchDev &b, const HcParamsDev &prm, uint32_t r_begin, double *segS,
                                double *segU, double *segD, double *nodeW, double *totals, hipStream_t st);
void launch_hc_sweep(const HcGraphDev &g, const uint32_t *item_node, const double *D, uint32_t n_items, int skip_zero,
                     double *acc, hipStream_t st);
void launch_hc_finish(const double *totals, const double *acc_seg, double *acc_node, uint32_t n_paths, uint32_t n_slots,
                      double *out, double *out2, hipStream_t st);
void launch_hc_read_loglik(const HcGraphDev &g, const HcBatchDev &b, const double *segS, const double *segU, double *out,
                           hipStream_t st);
void launch_hc_posterior(const double *final_vec, uint32_t n_paths, const uint32_t *list_off, const uint32_t *list_idx,
                         uint32_t n_lists, double *conf, hipStream_t st);

} // namespace vgan
