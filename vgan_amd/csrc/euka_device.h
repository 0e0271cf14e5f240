// Device-side views of the euka path (euka_kernels.hip / euka_capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vgan {

// HBM layout, graph/db side (built once per context):
//   bp / bp_clade   sorted node-id breakpoints of all bins and, per elementary interval, the clade the reference's
//                   scan ends on ("last (clade, bin) containing the node", readGAM_Euka.h:120-140); -1 = no bin.
//   sub5p / sub3p   [n][4][4] fp64 damage matrices per distance from the 5' / 3' end (damage.cpp:66-136); positions
//                   beyond the last row reuse it, so subDeamDiNuc[L][l] (128 MB in the reference) is never built:
//                   row b1 comes from the 5' matrix at l when its diagonal <= the 3' matrix's at L-l-1 (damage.cpp:18-36).
//   node_clade      the same lookup tabulated per node id (one load instead of a 13-step search) while the largest
//                   bin bound stays below 2^26; n_node_clade = 0 otherwise (the kernel searches bp then).
//   dmg_pair        [n5][n3][20] fp64: for every (5' row i5, 3' row i3) pair the selected 4x4 matrix, TRANSPOSED
//                   (MT[read base][original base]), followed by its four row sums.  Model 1 needs, per original base o,
//                   sum_b M[o][b] * w[b] with w = qs/3 except w[read base] = 1 - qs, i.e. w_miss * rowsum[o] +
//                   (w_hit - w_miss) * M[o][read base]: two 32-byte reads per column instead of sixteen 16-byte ones.
struct EukaDev {
    const uint32_t *bp;
    const int32_t *bp_clade;
    uint32_t n_bp;
    const int32_t *node_clade;
    uint32_t n_node_clade;
    const double *dmg_pair;
    const double *clade_dist;
    const uint32_t *bin_off;
    const int32_t *bin_lo, *bin_hi;
    const double *sub5p, *sub3p;
    uint32_t n5, n3;
    const double *qscore;  // [100] Euka::get_qscore_vec (Euka.cpp:38-51)
    const double *mapq_ok; // [256] 1 - pow(10, -mapq*0.1)  (miscfunc.h:215-216)
    uint32_t n_clades;
    uint32_t min_mapq;
    int32_t ltp; // lengthToProf
};

struct EukaBatchDev {
    uint32_t n_reads;
    const uint32_t *read_col_off, *read_qual_off, *read_map_off;
    const uint16_t *read_gseq_len, *read_rseq_len, *read_seq_len;
    const int32_t *read_mapq;
    const uint8_t *read_rev;
    const uint32_t *map_node;
    const uint8_t *graph_seq, *read_seq, *qual;
};

// Per-clade accumulators are kept in EUKA_REPLICAS copies (replica = blockIdx % EUKA_REPLICAS) and summed at finalize:
// a real sample concentrates on a handful of clades, and every read of a clade hits the same ~200 counters --
// same-address atomics serialise in L2 (1 clade: 18.8 ms per 1M reads without replicas).
constexpr uint32_t EUKA_REPLICAS = 32;

struct EukaOutDev {
    int32_t *clade;
    double *in_lik, *out_lik, *like, *not_like;
    uint8_t *pass;
    int32_t *clade_count; // [EUKA_REPLICAS][n_clades]
    uint32_t *baseshift;  // [EUKA_REPLICAS][n_clades][2*ltp][16]
    double *bin_cov;      // [EUKA_REPLICAS][n_bins]
    uint32_t *like_n;     // [EUKA_REPLICAS][n_clades] entries pushed to Clade::clade_like (readGAM_Euka.h:491)
    double *like_logsum;  // [EUKA_REPLICAS][n_clades] sum of log(clade_like[k]): what MCMC::get_proposal_likelihood needs
    uint32_t n_bins;
    unsigned long long *n_bad;
};

void launch_euka_reads(const EukaDev &d, const EukaBatchDev &b, const EukaOutDev &o, hipStream_t st);
// sums the replicas into replica 0 (count/baseshift/bin_cov element-wise)
void launch_euka_reduce(const EukaOutDev &o, uint32_t n_clades, int32_t ltp, hipStream_t st);
void launch_euka_clear(void *p, size_t bytes, hipStream_t st); // zero fill, bytes a multiple of 16

struct EukaCtxInfo { // what another translation unit needs of a context (euka_flatten_kernels.hip)
    int device;
    hipStream_t stream;
};

} // namespace vgan
struct vgan_euka_ctx;
struct vgan_euka_devflat;
namespace vgan {
EukaCtxInfo euka_ctx_info(const vgan_euka_ctx *c);
size_t euka_devflat_device_bytes(const vgan_euka_devflat *f);
}
