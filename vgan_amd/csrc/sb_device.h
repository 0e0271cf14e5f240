// Device-side views of the soibean path (sb_kernels.hip / sb_capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vgan {

constexpr uint32_t SB_MAX_PATHS = 256;
constexpr uint8_t SB_OK_DEFERRED = 2; // (while the tables are built: a read the column kernel left to the segment kernel)
constexpr uint32_t SB_NCNT = 25; // (reference base, read base) in {A, C, G, T, other}^2

struct SbGraphDev {
    const uint64_t *mask;   // [rows][mask_words] paths through each node (nodepaths, soibean.cpp:476-491)
    const uint8_t *findable; // [n_paths] 0 for names longer than 101 characters (getLCAfromGAM.h:80-88)
    const double *sub5p, *sub3p;
    uint32_t n5, n3;
    const double *qscore; // [100]
    uint32_t rows, mask_words, n_paths;
    int32_t penalty;
};

struct SbBatchDev {
    uint32_t n_reads;
    const uint32_t *read_seg_off, *read_col_off, *read_qual_off;
    const uint16_t *read_gseq_len, *read_rseq_len;
    const uint8_t *read_rev;
    const uint32_t *seg_node;
    const uint16_t *seg_col, *seg_len, *seg_base_ix;
    const uint8_t *graph_seq, *read_seq, *qual;
};

// Factorised result of analyse_GAM, HBM resident: the 2k paths one MCMC iteration touches are whole rows.
//   pm  double [n_paths][R]          sum of per-base log-likelihoods (pathMap)
//   cnt uint16 [n_paths][R / 64][25][64]  counts of (reference, read) pairs over path-supported bases, tiled by 64 reads
//   ok  uint8  [R]
struct SbTablesDev {
    double *pm;    // [path][read]
    uint16_t *cnt; // [path][read / 64][pair (25)][read % 64]: the 25 counts of a wave's 64 reads on one path are 3200 contiguous
                   // bytes -- the refresh streams them (one row per pair, n_reads apart, was 25 pages per path and wave)
    uint8_t *ok;
    uint32_t n_reads;
};
// where (path, read)'s count of pair 0 lies (pair j: + j * 64); n_tiles = sb_cnt_tiles(n_reads)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t sb_cnt_tiles(uint32_t n_reads) { return (n_reads + 63u) / 64u; }
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t sb_cnt_index(uint32_t path, uint32_t read, uint32_t n_tiles) {
    return (((size_t)path * n_tiles + read / 64u) * 25u) * 64u + read % 64u;
}

// Sums over reads are taken in fixed point, in integers (units of 2^-44): the result does not depend on how the reads are
// dealt to lanes, waves, workgroups, contexts or GPUs -- an MCMC accept / reject must not depend on the number of devices.
// A term that does not fit (not finite, or 2^18 and beyond in magnitude) goes into a plain double beside the integers.
// value = (hi * 2^32 + lo) * 2^-44 + nf (the layout of vgan_sb_sum, include/vgan_gpu.h)
struct SbFix {
    long long hi;
    unsigned long long lo;
    double nf;
};
#if defined(__HIPCC__)
__host__ __device__
#endif
inline double sb_fix_value(const SbFix &f) {
    // (both products are exact -- powers of two --, so a fused multiply-add here gives the same bits as two operations)
    return ((double)f.hi * 0x1p-12 + (double)f.lo * 0x1p-44) + f.nf;
}

struct SbSourceDev {
    int32_t child, parent;
    double t1, t2;      // pos*t, t - t1
    double log_pos, log_1mpos, log_theta, pos;
};

// kernel arguments of the single-launch refresh (the chain driver's per-iteration call)
constexpr uint32_t SB_FUSED_MAX_K = 16;
struct SbFusedArgs {
    SbSourceDev src[SB_FUSED_MAX_K];
    double freqs7[7];
    double con;
};

// stage_pm: chunk_reads * n_paths doubles, stage_cnt: chunk_reads * 25 * n_paths uint16 of scratch (read-major rows of one
// chunk of reads, transposed into the path-major tables chunk by chunk)
void launch_sb_precompute(const SbGraphDev &g, const SbBatchDev &b, const SbTablesDev &t, double *stage_pm, uint16_t *stage_cnt,
                          uint32_t chunk_reads, unsigned long long *n_bad, hipStream_t st);
void launch_sb_loglike(const SbTablesDev &t, uint32_t n_paths, uint32_t n_states, uint32_t k, const SbSourceDev *src,
                       const double *hky /* [n_states*k][2][25] */, SbFix *partial, uint32_t n_blocks, double *out,
                       double *out2 /* optional second copy of out */, SbFix *out_fix /* optional: the sums themselves */,
                       unsigned long long *guard, hipStream_t st);
// also zeroes guard[0..n_states)
void launch_sb_hky(uint32_t n_entries, const SbSourceDev *src, double con, const double *freqs7, double *hky,
                   unsigned long long *guard, uint32_t n_states, hipStream_t st);

// n_states * k <= SB_FUSED_MAX_K sources, one launch, no copies: out_host / guard_host[n_states] are pinned host memory;
// fix_host[n_states] likewise (the sums themselves: a caller adding over several contexts takes these);
// guard = n_states zeroed device words (left zeroed), ticket = one zeroed device word (left zeroed); partial: n_states * n_blocks
// entries.  Bit-identical to launch_sb_hky + launch_sb_loglike.
void launch_sb_refresh_fused(const SbTablesDev &t, uint32_t n_states, uint32_t k, const SbFusedArgs &a, SbFix *partial, uint32_t n_blocks,
                             unsigned long long *guard, unsigned int *ticket, double *out_host, unsigned long long *guard_host, SbFix *fix_host,
                             unsigned long long *seq_host, unsigned long long seq, hipStream_t st);
// The same refresh served by a kernel that stays on the device (sb_kernels.hip: sb_refresh_resident_kernel): mailbox = sb_mailbox_bytes() of
// zeroed pinned host memory, resident = sb_resident_bytes() of zeroed device memory (seq = done), n_blocks <= sb_resident_grid().  The host
// posts a refresh with sb_mailbox_post and watches seq_host as after launch_sb_refresh_fused.
uint32_t sb_refresh_grid(int device); // workgroups of launch_sb_refresh_fused the device holds at once
size_t sb_mailbox_bytes();
size_t sb_resident_bytes();
void sb_mailbox_post(void *mailbox, uint32_t n_states, uint32_t k, const SbFusedArgs &a, unsigned long long seq);
void sb_mailbox_idle(void *mailbox, unsigned long long done); // nothing posted: the number of the last refresh served
void sb_mailbox_stop(void *mailbox, bool on);
unsigned long long sb_mailbox_exited(const void *mailbox);
void sb_mailbox_busy(void *mailbox, unsigned long long *ticks, unsigned long long *served, unsigned long long stamp[8]); // (as the kernel left them when it last left)
uint32_t sb_resident_grid(int device, uint32_t want);
void launch_sb_refresh_resident(const SbTablesDev &t, void *mailbox, void *resident, SbFix *partial, uint32_t n_blocks, unsigned long long *guard,
                                double *out_host, unsigned long long *guard_host, SbFix *fix_host, unsigned long long *seq_host, unsigned long long done,
                                unsigned long long idle_ticks, unsigned long long launch_id, hipStream_t st);
// per-read best path (-1: tie or excluded read), per-path signature counts and the number of usable reads; counters zeroed by the caller
void launch_sb_best_paths(const SbTablesDev &t, uint32_t n_paths, int32_t *best, unsigned long long *sig_count,
                          unsigned long long *n_ok, hipStream_t st);
// out[0] = sum over reads of (+)_j (log_freq + pm[paths[j]]); partial: n_blocks entries of scratch; out_fix (or NULL): the sum itself
void launch_sb_mixture(const SbTablesDev &t, uint32_t n, const int32_t *paths, double log_freq, SbFix *partial, uint32_t n_blocks,
                       double *out, SbFix *out_fix, hipStream_t st);

struct SbCtxInfo { // what another translation unit needs of a context (sb_flatten_kernels.hip)
    int device;
    hipStream_t stream;
};

} // namespace vgan
struct vgan_sb_ctx;
struct vgan_sb_devflat;
namespace vgan {
SbCtxInfo sb_ctx_info(const vgan_sb_ctx *c);
size_t sb_devflat_device_bytes(const vgan_sb_devflat *f);
}
