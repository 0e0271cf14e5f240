// soibean kernels for gfx950.
//
//   sb_precompute_kernel  analyse_GAM (reference src/getLCAfromGAM.h:92-560), one wave per read:
//       (A) one lane per edit-level segment: the supported sum, the unsupported sum (penalty pattern on the running
//           read position) and the 5x5 (reference, read) counts of its regular columns -- none depends on the path;
//       (B) one lane per path: pm[p] = sum_m (path through node_m ? sup_m : uns_m), cnt[p] = sum of the supported
//           segments' counts; bit p of the node's path mask decides.
//   sb_hky_kernel         the 25-entry HKY tables of an MCMC state (src/MCMC.h:111-296; kappa = 1/22 = 0, Q13).
//   sb_loglike_kernel     one likelihood refresh (src/MCMC.cpp:738-993): per read and source
//       LL = pm[child] + sum_j cnt[child][j] * hky_t2[j], LLP likewise for the parent with t1, mixed over the branch
//       position and the sources; reads reduced with wave shuffles, per-block partials summed in a fixed order
//       (deterministic: the MCMC's accept/reject must not depend on atomics' arrival order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "device_math.h"
#include "sb_device.h"

namespace vgan {

__device__ __forceinline__ int acgt5(uint32_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }

constexpr int SBP_WAVES = 4;
constexpr double SB_LOG_025 = -1.3862943611198906;      // log(0.25)
constexpr double SB_LOG_002 = -3.912023005428146;       // log(0.02)
constexpr double SB_LOG_CLAMP = -1.0000000494736474e-07; // log(0.9999999)

struct SbSegLds {
    double sup, uns;
    uint32_t node;
    uint16_t cnt[SB_NCNT]; // a segment has up to 65535 columns (seg_len is 16 bit), e.g. a 300-column homopolymer match edit
    uint16_t pad;
};

template <int PP>
__global__ __launch_bounds__(SBP_WAVES * 64) void sb_precompute_kernel(SbGraphDev g, SbBatchDev b, SbTablesDev t, uint32_t r_begin,
                                                                        uint32_t r_end, double *__restrict__ stage_pm,
                                                                        uint16_t *__restrict__ stage_cnt,
                                                                        unsigned long long *n_bad, bool only_deferred) {
    __shared__ double qs_s[100];
    __shared__ SbSegLds seg_s[SBP_WAVES][64];
    for (int i = threadIdx.x; i < 100; i += blockDim.x) qs_s[i] = g.qscore[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t P = g.n_paths;

    for (uint32_t r = r_begin + blockIdx.x * SBP_WAVES + wave; r < r_end; r += gridDim.x * SBP_WAVES) {
        if (only_deferred && t.ok[r] != SB_OK_DEFERRED) continue; // (the column kernel left this read to this one)
        const uint32_t s0 = b.read_seg_off[r], s1 = b.read_seg_off[r + 1];
        const uint32_t col0 = b.read_col_off[r];
        const uint32_t q0 = b.read_qual_off[r], QL = b.read_qual_off[r + 1] - q0;
        const uint32_t Lseq = b.read_gseq_len[r], A = b.read_rseq_len[r];
        const bool rev = b.read_rev[r] != 0;
        double pm[PP];
        uint32_t cnt[PP][SB_NCNT];
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            pm[u] = 0.0;
#pragma unroll
            for (int j = 0; j < (int)SB_NCNT; ++j) cnt[u][j] = 0;
        }
        bool bad = false;
        for (uint32_t sb = s0; sb < s1; sb += 64) {
            // ---- (A) one lane per segment
            const uint32_t s = sb + lane;
            if (s < s1) {
                const uint32_t node = b.seg_node[s], col = b.seg_col[s], len = b.seg_len[s];
                const int32_t bix = b.seg_base_ix[s];
                double sup = 0.0, uns = 0.0;
                uint16_t c25[SB_NCNT];
#pragma unroll
                for (int j = 0; j < (int)SB_NCNT; ++j) c25[j] = 0;
                if (len > 0 && (uint32_t)bix >= Lseq) bad = true;
                const uint32_t nn = min((uint32_t)bix, Lseq - 1u);
                const double *m5 = g.sub5p + 16u * min(nn, g.n5 - 1u);
                const double *m3 = g.sub3p + 16u * min(Lseq - 1u - nn, g.n3 - 1u);
                // row sums of the combined damage matrix are what the supported marginal needs (it adds log(post[b]) for
                // every b, getLCAfromGAM.h:338-348): sum_b post[b] = sum_o pre[o] * rowsum(M[o])
                double rowsum[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const double *row5 = m5 + 4 * o, *row3 = m3 + 4 * o;
                    const double *row = row5[o] <= row3[o] ? row5 : row3; // damage.cpp:18-36
                    rowsum[o] = ((row[0] + row[1]) + row[2]) + row[3];
                }
                int32_t bo = bix; // baseOnRead of the unsupported walk
                for (uint32_t k = 0; k < len; ++k) {
                    const uint32_t gc = b.graph_seq[col0 + col + k];
                    const uint32_t rc = (col + k) < A ? b.read_seq[col0 + col + k] : 0u;
                    int q = k < QL ? (int)(int8_t)b.qual[q0 + k] : 0; // Q12: within-segment index
                    q = q < 0 ? 0 : (q > 99 ? 99 : q);
                    const double qs = qs_s[q];
                    // supported term = cs + log(as) (clamped), unsupported term = cu + log(au): one log each
                    double as = 1.0, au = 1.0, cs = 0.0;
                    bool regular = false;
                    if (gc == 'N' || rc == 'N') {
                        cs = SB_LOG_025;
                    } else if (gc == 'S' || rc == 'S') {
                        as = au = qs / 3.0;
                    } else if (gc == '-' || rc == '-') {
                        cs = SB_LOG_002;
                    } else {
                        regular = true;
                        const int gi = acgt5(gc);
                        double p = 0.0;
#pragma unroll
                        for (int o = 0; o < 4; ++o) p += (o == gi ? 1.0 - qs : qs / 3.0) * rowsum[o];
                        as = p;
                        const uint32_t ab = (uint32_t)(bo < 0 ? -bo : bo);
                        au = (ab % (uint32_t)g.penalty == 0u) ? 1.0 - qs : qs / 3.0; // :473-512
                        const int j = gi * 5 + acgt5(rc);
#pragma unroll
                        for (int jj = 0; jj < (int)SB_NCNT; ++jj) c25[jj] += (jj == j) ? 1 : 0;
                    }
                    double ls = cs + log_pos(as);
                    const double lu = cs + log_pos(au);
                    if (regular && ls > SB_LOG_CLAMP) ls = SB_LOG_CLAMP; // :349-351
                    sup += ls;
                    uns += lu;
                    if (rc != '-') bo += rev ? -1 : 1; // :515-519
                }
                SbSegLds &o = seg_s[wave][lane];
                o.sup = sup;
                o.uns = uns;
                o.node = node;
#pragma unroll
                for (int j = 0; j < (int)SB_NCNT; ++j) o.cnt[j] = c25[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- (B) one lane per path (PP paths per lane)
            const uint32_t nseg = min(64u, s1 - sb);
            for (uint32_t m = 0; m < nseg; ++m) {
                const SbSegLds &sg = seg_s[wave][m];
                const uint32_t node = sg.node;
#pragma unroll
                for (int u = 0; u < PP; ++u) {
                    const uint32_t p = u * 64 + lane;
                    bool sup = false;
                    if (p < g.n_paths && node != 0u && node < g.rows)
                        sup = ((g.mask[(size_t)node * g.mask_words + (p >> 6)] >> (p & 63)) & 1ull) && g.findable[p];
                    pm[u] += sup ? sg.sup : sg.uns;
                    if (sup) {
#pragma unroll
                        for (int j = 0; j < (int)SB_NCNT; ++j) cnt[u][j] += sg.cnt[j];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
        bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) {
            t.ok[r] = bad ? 0 : 1;
            if (bad) atomicAdd(n_bad, 1ull);
        }
        // Read-major staging rows [read][path] and [read][pair][path]: a lane's stores sit beside its neighbours' (one wave
        // writes whole 128-byte lines).  Writing the path-major tables from here scattered 2-byte stores R * 50 bytes apart,
        // which HBM took as one 32-byte write each (13.8x the table's size); sb_transpose_kernel brings the rows into place.
        const size_t rl = r - r_begin;
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const uint32_t p = u * 64 + lane;
            if (p < P) {
                stage_pm[rl * P + p] = pm[u];
#pragma unroll
                for (int j = 0; j < (int)SB_NCNT; ++j) stage_cnt[(rl * SB_NCNT + j) * P + p] = (uint16_t)min(cnt[u][j], 65535u);
            }
        }
    }
}

// ---- the same tables, one lane per alignment column ----------------------------------------------------------------------
// sb_precompute_kernel gives a lane to every edit-level SEGMENT and lets it walk its columns: the walk is as long as the
// wave's longest segment while the average one has 1.4 columns, and the per-path pass adds all 25 pair counters of every
// segment on P of 64 lanes.  Here a wave still owns a read, but
//   A  lanes over the segments: their column ranges become SLOTS (slot = (segment, k), off_m = prefix sum of the lengths: the
//      ranges of a reverse read step backwards and may overlap, slots do not), the damage row sums, the nodes' path masks;
//   B  lanes over the slots: the column's supported / unsupported log terms (one log: the unsupported term is one of two
//      table values per quality) and its pair index; the running read position of the penalty pattern is the segment's base
//      index plus a rank difference (ballot + mbcnt over "read base is not a gap");
//   C  lanes over the segments again: sup_m / uns_m summed in column order (a few adds per segment);
//   D  lanes over the paths: pm[p] in segment order;
//   E  lanes over (slot, path) pairs: one LDS counter add per supported regular column and path -- the 5x5 counts as a
//      histogram instead of 25 adds per segment and path.
// Same arithmetic in the same order per value as the segment kernel (bit-identical tables).  A read beyond the capacities
// (SBC_CAPS segments, SBC_CAPT slots) is marked SB_OK_DEFERRED and taken by the segment kernel in a second launch.
constexpr int SBC_WAVES = 4;
constexpr int SBC_CAPS = 96, SBC_CAPT = 160;
constexpr int SBC_DMG_POS = 40; // positions of the two damage profiles kept in LDS (beyond: read from HBM)
struct SbcSeg {
    uint32_t node;
    uint16_t col, len;
    int32_t bix;
    uint16_t off, pad; // first slot
};
template <int MW> struct alignas(16) SbcSlice { // MW: 32-bit words of a node's path mask
    union {
        double rowsum[SBC_CAPS][4]; // phases A, B
        struct {
            double sup[SBC_CAPS], uns[SBC_CAPS]; // phases C, D
        } su;
    };
    SbcSeg seg[SBC_CAPS];
    uint32_t smask[SBC_CAPS][MW];
    uint16_t slot[SBC_CAPT]; // owner segment | pair index << 8 (pair index SB_NCNT: no pair)
    // behind it, sized by the launcher: {ls[CAPT], lu[CAPT], rank[CAPT]} (phases B, C), then the pair histogram in their place (E)
};
constexpr size_t SBC_TAIL_MIN = (size_t)SBC_CAPT * (8 + 8 + 2);
inline size_t sbc_slice_bytes(size_t fixed, uint32_t hist_n) { return (fixed + std::max(SBC_TAIL_MIN, (size_t)hist_n * 4) + 15) & ~(size_t)15; }

__device__ __forceinline__ uint32_t sbc_scan_incl(uint32_t v, int lane) { // wave64 inclusive prefix sum
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

template <int MWT>
__global__ __launch_bounds__(SBC_WAVES * 64) void sb_precompute_cols_kernel(SbGraphDev g, SbBatchDev b, SbTablesDev t, uint32_t r_begin,
                                                                            uint32_t r_end, double *__restrict__ stage_pm,
                                                                            uint16_t *__restrict__ stage_cnt, unsigned long long *n_bad,
                                                                            uint32_t ppad_log2, uint32_t pen_magic) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sbc_smem[];
    __shared__ double qs_s[100], q3_s[100], l1m_s[100], lq3_s[100]; // eps(Q), eps / 3, log(1 - eps), log(eps / 3)
    __shared__ uint32_t fw_s[8];                            // findable paths, as mask words
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t P = g.n_paths, MW = g.mask_words;
    const uint32_t ppad = 1u << ppad_log2, hist_n = SB_NCNT << ppad_log2;
    using Slice = SbcSlice<MWT>;
    const size_t tail_bytes = SBC_TAIL_MIN > (size_t)hist_n * 4 ? SBC_TAIL_MIN : (size_t)hist_n * 4;
    const size_t slice_bytes = (sizeof(Slice) + tail_bytes + 15) & ~(size_t)15;
    Slice &L = *reinterpret_cast<Slice *>(sbc_smem + (size_t)wave * slice_bytes);
    uint8_t *tail = sbc_smem + (size_t)wave * slice_bytes + sizeof(Slice);
    double *const ls_s = reinterpret_cast<double *>(tail), *const lu_s = ls_s + SBC_CAPT;
    uint16_t *const rank_s = reinterpret_cast<uint16_t *>(lu_s + SBC_CAPT);
    uint32_t *const hist = reinterpret_cast<uint32_t *>(tail); // [pair][ppad], once ls / lu / rank are dead
    for (int i = threadIdx.x; i < 100; i += blockDim.x) {
        const double e = g.qscore[i];
        qs_s[i] = e;
        q3_s[i] = e / 3.0; // (the quotient the segment kernel forms per column: the same bits, one division per table entry)
        l1m_s[i] = log_pos(1.0 - e);
        lq3_s[i] = log_pos(e / 3.0);
    }
    if (threadIdx.x < 8) {
        uint32_t w = 0;
        for (uint32_t p = threadIdx.x * 32u; p < min(P, threadIdx.x * 32u + 32u); ++p) w |= (uint32_t)(g.findable[p] != 0) << (p & 31u);
        fw_s[threadIdx.x] = w;
    }
    // per position of either profile and per row o: the row's diagonal element and its sum (what a segment needs of the two
    // matrices at its base: damage.cpp:18-36 picks the row by the diagonals, getLCAfromGAM.h:338-348 needs the row's sum)
    __shared__ double dg_s[SBC_DMG_POS][4], rs_s[SBC_DMG_POS][4];
    const uint32_t n5 = g.n5, n3 = g.n3;
    const bool dmg_lds = n5 + n3 <= (uint32_t)SBC_DMG_POS;
    if (dmg_lds) {
        for (uint32_t i = threadIdx.x; i < (n5 + n3) * 4u; i += blockDim.x) {
            const uint32_t pos = i >> 2, o = i & 3u;
            const double *row = (pos < n5 ? g.sub5p + 16u * pos : g.sub3p + 16u * (pos - n5)) + 4u * o;
            dg_s[pos][o] = row[o];
            rs_s[pos][o] = ((row[0] + row[1]) + row[2]) + row[3];
        }
    }
    __syncthreads();
    const uint32_t pen = (uint32_t)g.penalty;

    for (uint32_t r = r_begin + blockIdx.x * SBC_WAVES + wave; r < r_end; r += gridDim.x * SBC_WAVES) {
        const uint32_t s0 = b.read_seg_off[r], M = b.read_seg_off[r + 1] - s0;
        const uint32_t col0 = b.read_col_off[r];
        const uint32_t q0 = b.read_qual_off[r], QL = b.read_qual_off[r + 1] - q0;
        const uint32_t Lseq = b.read_gseq_len[r], A = b.read_rseq_len[r];
        const int dir = b.read_rev[r] != 0 ? -1 : 1;
        const size_t rl = r - r_begin;
        // ---- A: segments -> slots
        bool bad = false;
        uint32_t T = 0; // slots so far
        if (M <= (uint32_t)SBC_CAPS) {
            for (uint32_t m0 = 0; m0 < M; m0 += 64) {
                const uint32_t m = m0 + lane;
                const bool on = m < M;
                const uint32_t len = on ? b.seg_len[s0 + m] : 0u;
                const uint32_t incl = sbc_scan_incl(len, lane);
                const uint32_t off = T + incl - len;
                T += (uint32_t)__shfl((int)incl, 63, 64);
                if (on && off + len <= (uint32_t)SBC_CAPT) {
                    const uint32_t node = b.seg_node[s0 + m], col = b.seg_col[s0 + m];
                    const int32_t bix = b.seg_base_ix[s0 + m];
                    if (len > 0 && (uint32_t)bix >= Lseq) bad = true;
                    const uint32_t nn = min((uint32_t)bix, Lseq - 1u);
                    const uint32_t i5 = min(nn, n5 - 1u), i3 = min(Lseq - 1u - nn, n3 - 1u);
                    if (dmg_lds) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) L.rowsum[m][o] = dg_s[i5][o] <= dg_s[n5 + i3][o] ? rs_s[i5][o] : rs_s[n5 + i3][o];
                    } else {
                        const double *m5 = g.sub5p + 16u * i5, *m3 = g.sub3p + 16u * i3;
#pragma unroll
                        for (int o = 0; o < 4; ++o) { // sum_b post[b] = sum_o pre[o] * rowsum(M[o]) (getLCAfromGAM.h:338-348)
                            const double *row5 = m5 + 4 * o, *row3 = m3 + 4 * o;
                            const double *row = row5[o] <= row3[o] ? row5 : row3; // damage.cpp:18-36
                            L.rowsum[m][o] = ((row[0] + row[1]) + row[2]) + row[3];
                        }
                    }
                    L.seg[m] = SbcSeg{node, (uint16_t)col, (uint16_t)len, bix, (uint16_t)off, 0};
                    const bool known = node != 0u && node < g.rows;
                    const uint32_t *mrow = reinterpret_cast<const uint32_t *>(g.mask + (size_t)node * MW);
#pragma unroll
                    for (uint32_t w = 0; w < (uint32_t)MWT; ++w) L.smask[m][w] = (known && w < 2u * MW) ? (mrow[w] & fw_s[w]) : 0u;
                    for (uint32_t k = 0; k < len; ++k) L.slot[off + k] = (uint16_t)m;
                }
            }
        }
        if (M > (uint32_t)SBC_CAPS || T > (uint32_t)SBC_CAPT) { // wave uniform: the segment kernel takes this read
            if (lane == 0) t.ok[r] = SB_OK_DEFERRED;
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- B: slots
        uint32_t rank_base = 0;
        for (uint32_t t0 = 0; t0 < T; t0 += 64) {
            const uint32_t sl = t0 + lane;
            const bool on = sl < T;
            const uint32_t m = on ? L.slot[sl] : 0u;
            const SbcSeg sg = L.seg[m];
            const uint32_t k = sl - sg.off, c = (uint32_t)sg.col + k;
            const uint32_t gc = on ? b.graph_seq[col0 + c] : 0u;
            const uint32_t rc = (on && c < A) ? b.read_seq[col0 + c] : 0u;
            int q = (on && k < QL) ? (int)(int8_t)b.qual[q0 + k] : 0; // Q12: within-segment index
            const uint64_t ng = __builtin_amdgcn_ballot_w64(on && rc != '-');
            const uint32_t rank = rank_base + __builtin_amdgcn_mbcnt_hi((uint32_t)(ng >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ng, 0u));
            rank_base += (uint32_t)__builtin_popcountll(ng);
            if (on) rank_s[sl] = (uint16_t)rank;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (on) {
                const uint32_t before = rank - (uint32_t)rank_s[sg.off]; // read bases (not gaps) of the segment before this column
                const int32_t bo = sg.bix + dir * (int32_t)before;     // baseOnRead of the unsupported walk (:515-519)
                q = q < 0 ? 0 : (q > 99 ? 99 : q);
                const double qs = qs_s[q];
                double ls, lu;
                uint32_t j = SB_NCNT; // no pair
                if (gc == 'N' || rc == 'N') {
                    ls = lu = SB_LOG_025;
                } else if (gc == 'S' || rc == 'S') {
                    ls = lu = lq3_s[q];
                } else if (gc == '-' || rc == '-') {
                    ls = lu = SB_LOG_002;
                } else {
                    const int gi = acgt5(gc);
                    double p = 0.0;
                    const double e3 = q3_s[q];
#pragma unroll
                    for (int o = 0; o < 4; ++o) p += (o == gi ? 1.0 - qs : e3) * L.rowsum[m][o];
                    ls = log_pos(p);
                    if (ls > SB_LOG_CLAMP) ls = SB_LOG_CLAMP; // :349-351
                    const uint32_t ab = (uint32_t)(bo < 0 ? -bo : bo);
                    // ab % PENALTY (:473-512) by the launcher's reciprocal (exact for ab < 2^16: |bo| <= base index + columns)
                    const uint32_t quo = pen == 1u ? ab : (ab < 65536u ? __umulhi(ab, pen_magic) : ab / pen);
                    lu = (ab - quo * pen == 0u) ? l1m_s[q] : lq3_s[q];
                    j = (uint32_t)(gi * 5 + acgt5(rc));
                }
                ls_s[sl] = ls;
                lu_s[sl] = lu;
                L.slot[sl] = (uint16_t)(m | (j << 8));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- C: per segment, in column order (rowsum is dead: sup / uns take its place)
        for (uint32_t m0 = 0; m0 < M; m0 += 64) {
            const uint32_t m = m0 + lane;
            double sup = 0.0, uns = 0.0;
            if (m < M) {
                const SbcSeg sg = L.seg[m];
                for (uint32_t k = 0; k < sg.len; ++k) {
                    sup += ls_s[sg.off + k];
                    uns += lu_s[sg.off + k];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (every lane of the pass has read its rowsum-free inputs)
            if (m < M) {
                L.su.sup[m] = sup;
                L.su.uns[m] = uns;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = lane; i < hist_n; i += 64) hist[i] = 0u; // (ls / lu / rank are dead)
        // ---- D: per path, in segment order
        for (uint32_t p = lane; p < P; p += 64) {
            double pm = 0.0;
            const uint32_t w = MWT == 1 ? 0u : p >> 5, bit = p & 31u;
            for (uint32_t m = 0; m < M; ++m) pm += ((L.smask[m][w] >> bit) & 1u) ? L.su.sup[m] : L.su.uns[m];
            stage_pm[rl * P + p] = pm;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- E: (slot, path) pairs
        const uint32_t n_pairs = T << ppad_log2;
        for (uint32_t i = lane; i < n_pairs; i += 64) {
            const uint32_t sl = i >> ppad_log2, p = i & (ppad - 1u);
            const uint32_t rec = L.slot[sl], j = rec >> 8;
            if (j < SB_NCNT && ((L.smask[rec & 0xFFu][MWT == 1 ? 0u : p >> 5] >> (p & 31u)) & 1u)) atomicAdd(&hist[(j << ppad_log2) + p], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // read-major staging rows (see sb_precompute_kernel): [read][pair][path]
        for (uint32_t h = lane; h < hist_n; h += 64) {
            const uint32_t j = h >> ppad_log2, p = h & (ppad - 1u);
            if (p < P) stage_cnt[(rl * SB_NCNT + j) * P + p] = (uint16_t)min(hist[h], 65535u);
        }
        bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) {
            t.ok[r] = bad ? 0 : 1;
            if (bad) atomicAdd(n_bad, 1ull);
        }
    }
}

// in[r][f] for r < n_r, f < F: the staging rows of a chunk of reads become columns of the path-major tables (ncnt = 1: pm[f][read],
// F = P; ncnt = 25: cnt, F = 25 * P with f = pair * P + path, into the tiled layout of sb_device.h).  64 x 64 tiles through LDS,
// reads and writes both run along the fast axis of their array.
template <class T>
__global__ __launch_bounds__(256) void sb_transpose_kernel(const T *__restrict__ in, T *__restrict__ out, uint32_t n_r, uint32_t F,
                                                            uint32_t P, uint32_t ncnt, uint32_t r_begin, uint32_t R) {
    __shared__ T tile[64][64 + (sizeof(T) == 2 ? 2 : 1)];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r0 = blockIdx.x * 64, f0 = blockIdx.y * 64;
#pragma unroll 4
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t rl = wave * 16 + i, f = f0 + lane;
        if (r0 + rl < n_r && f < F) tile[rl][lane] = in[(size_t)(r0 + rl) * F + f];
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t fc = wave * 16 + i, f = f0 + fc;
        if (f < F && r0 + lane < n_r) {
            if (ncnt == 1u) {
                out[(size_t)(f % P) * R + r_begin + r0 + lane] = tile[lane][fc];
            } else { // the counts: tiled by 64 reads (sb_device.h)
                out[sb_cnt_index(f % P, r_begin + r0 + lane, sb_cnt_tiles(R)) + (size_t)(f / P) * 64u] = tile[lane][fc];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- HKY tables
// hky[e][which][ref*5+read]: which = 0 child (t2), 1 parent (t1); MCMC.h:111-296 minus the "+ detail.logLikelihood"
// log of the HKY transition probability ref -> bpo over time t, floored at 1e-8 (MCMC.h:144-233)
__device__ __forceinline__ double sb_hky_logp(double t, int ref, int bpo, const double *__restrict__ freqs7) {
    const double fR = freqs7[4], fY = freqs7[5], mu = freqs7[6];
    const double kappa = 0.0; // 1/22 in integer arithmetic (MCMC.h:66)
    const double f = freqs7[bpo];
    const bool pur = bpo == 0 || bpo == 2; // A, G
    const double grp = pur ? fR : fY;
    const double Aexp = 1 + grp * (kappa - 1);
    double v;
    if (bpo == ref) {
        const double jut1 = f + f * ((1 / grp) - 1) * exp(-(mu * t));
        const double jut11 = ((grp - f) / grp) * exp(-(mu * t * Aexp));
        v = jut1 + jut11;
    } else if (ref < 4 && (bpo ^ ref) == 2) { // transition partner: A<->G, C<->T
        const double jut1 = f + f * ((1 / grp) - 1) * exp(-(mu * t));
        const double jut11 = (f / grp) * exp(-(mu * t * Aexp));
        v = jut1 > jut11 ? jut1 - jut11 : jut11 - jut1;
    } else {
        v = f * (1 - exp(-(mu * t)));
    }
    return log(v < 1e-8 ? 1e-8 : v); // NaN stays NaN and trips the guard downstream
}

// log-sum-exp over the four post-mutation bases of log P_b + log(b == read ? 1 - con : con/3), folded as the reference does
__device__ __forceinline__ double sb_hky_fold(const double *__restrict__ logp4, int rd, double con) {
    double acc = -INFINITY;
    for (int bpd = 0; bpd < 4; ++bpd) {
        const double y = logp4[bpd] + (bpd == rd ? log(1 - con) : log(con / 3));
        if (acc == 0.0) acc = y; // oplusInitnatl
        else if (acc == -INFINITY) acc = y;
        else acc = fmax(acc, y) + log1p(exp(-fabs(acc - y)));
    }
    if (acc > 1e-8) acc = log(0.999999999);
    return acc;
}

__device__ __forceinline__ double sb_hky_entry(const SbSourceDev &s, uint32_t which, uint32_t j, double con, const double *__restrict__ freqs7) {
    const int ref = j / 5, rd = j % 5;
    const double t = which ? s.t1 : s.t2;
    double lp[4];
    for (int bpo = 0; bpo < 4; ++bpo) lp[bpo] = sb_hky_logp(t, ref, bpo, freqs7);
    return sb_hky_fold(lp, rd, con);
}

__global__ void sb_hky_kernel(uint32_t n_entries, const SbSourceDev *__restrict__ src, double con,
                              const double *__restrict__ freqs7, double *__restrict__ hky, unsigned long long *__restrict__ guard,
                              uint32_t n_states) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_states) guard[i] = 0; // the refresh that follows counts into it
    if (i >= n_entries * 2 * SB_NCNT) return;
    const uint32_t e = i / (2 * SB_NCNT), which = (i / SB_NCNT) & 1u, j = i % SB_NCNT;
    hky[i] = sb_hky_entry(src[e], which, j, con, freqs7);
}

// ---------------------------------------------------------------------------------------------- refresh
constexpr int SBL_THREADS = 256;

// one read's contribution to a state's log-likelihood (MCMC.cpp:738-993): k sources, src / hk_s of that state
__device__ __forceinline__ double sb_read_term(const SbTablesDev &t, uint32_t r, uint32_t k, const SbSourceDev *__restrict__ src,
                                               const double *__restrict__ hk_s, unsigned long long &bad) {
    const uint32_t R = t.n_reads;
    double inter = -INFINITY;
    for (uint32_t y = 0; y < k; ++y) {
        const SbSourceDev s = src[y];
        const double *hc = hk_s + (size_t)y * 2 * SB_NCNT, *hp = hc + SB_NCNT;
        double LL = t.pm[(size_t)s.child * R + r], LLP = t.pm[(size_t)s.parent * R + r];
        const uint32_t n_tiles = sb_cnt_tiles(R);
        const uint16_t *cc = t.cnt + sb_cnt_index((uint32_t)s.child, r, n_tiles);
        const uint16_t *cp = t.cnt + sb_cnt_index((uint32_t)s.parent, r, n_tiles);
        // all fifty counts are requested before the first is used (left to itself the compiler orders this loop load, wait, use)
        uint32_t cv[SB_NCNT], pv[SB_NCNT];
#pragma unroll
        for (int j = 0; j < (int)SB_NCNT; ++j) {
            cv[j] = cc[j * 64];
            pv[j] = cp[j * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < (int)SB_NCNT; ++j) {
            LL += (double)cv[j] * hc[j];
            LLP += (double)pv[j] * hp[j];
        }
        if (!(LL <= 0.0) || !(LLP <= 0.0) || isinf(LL) || isinf(LLP)) bad++; // MCMC.cpp:857-862,953-958
        if (k == 1) { // calculateLogWeightedAverage (MCMC.h:299-315)
            const double a = LL + s.log_pos, bb = LLP + s.log_1mpos;
            const double mx = fmax(a, bb);
            const double lse = mx + log(exp(a - mx) + exp(bb - mx));
            const double lws = log(s.pos + (1 - s.pos));
            inter = isinf(lws) ? -INFINITY : lse - lws;
        } else { // :967-974
            const double a = s.log_pos + LL, bb = s.log_1mpos + LLP;
#ifdef SB_HACK_NOLIBM // (developer aid, wrong results: what the exp / log1p pairs cost)
            const double inter2 = fmax(a, bb) + 0.5 * fabs(a - bb);
            const double yv = inter2 + s.log_theta;
            if (inter == 0.0 || inter == -INFINITY) inter = yv;
            else inter = fmax(inter, yv) + 0.5 * fabs(inter - yv);
#else
            const double inter2 = fmax(a, bb) + softplus_neg(fabs(a - bb));
            const double yv = inter2 + s.log_theta;
            if (inter == 0.0 || inter == -INFINITY) inter = yv;
            else inter = fmax(inter, yv) + softplus_neg(fabs(inter - yv));
#endif
        }
    }
    return inter;
}

// ---- fixed-point sums over reads (sb_device.h: SbFix)
__device__ __forceinline__ void sb_fix_add(SbFix &f, double x) {
    if (fabs(x) < 0x1p18) { // (a NaN fails the comparison)
        // x * 2^44 = h * 2^32 + l with h = floor(x * 2^12) (an int32) and 0 <= l < 2^32, l rounded to an integer: a function of x
        // alone, so the sum of the (h, l) pairs is the same in any order
        const double h = floor(x * 0x1p12);
        const double l = rint(fma(-h, 0x1p32, x * 0x1p44));
        const bool carry = l >= 0x1p32; // (rounded up to the next unit of hi)
        f.hi += (long long)(int)h + (carry ? 1 : 0);
        f.lo += carry ? 0ull : (unsigned long long)(unsigned int)l;
    } else {
        f.nf += x;
    }
}
__device__ __forceinline__ SbFix sb_fix_wave_sum(SbFix f) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        f.hi += __shfl_xor(f.hi, o, 64);
        f.lo += __shfl_xor(f.lo, o, 64);
        f.nf += __shfl_xor(f.nf, o, 64);
    }
    return f;
}
// the workgroup's sum into partial[slot] (thread 0 writes it); red_s: one entry per wave.  Ends with a barrier.
__device__ __forceinline__ void sb_fix_block_store(SbFix f, SbFix *red_s, SbFix *dst) {
    f = sb_fix_wave_sum(f);
    if ((threadIdx.x & 63) == 0) red_s[threadIdx.x >> 6] = f;
    __syncthreads();
    if (threadIdx.x == 0) {
        SbFix s2{0, 0, 0.0};
        for (int w = 0; w < SBL_THREADS / 64; ++w) {
            s2.hi += red_s[w].hi;
            s2.lo += red_s[w].lo;
            s2.nf += red_s[w].nf;
        }
        *dst = s2;
    }
    __syncthreads();
}
// one wave: the sum of n entries (lane l takes l, l + 64, ...)
__device__ __forceinline__ SbFix sb_fix_fold(const SbFix *p, uint32_t n) {
    SbFix s{0, 0, 0.0};
    for (uint32_t i = threadIdx.x; i < n; i += 64) {
        s.hi += p[i].hi;
        s.lo += p[i].lo;
        s.nf += p[i].nf;
    }
    return sb_fix_wave_sum(s);
}

__global__ __launch_bounds__(SBL_THREADS) void sb_loglike_kernel(SbTablesDev t, uint32_t n_states, uint32_t k,
                                                                  const SbSourceDev *__restrict__ src,
                                                                  const double *__restrict__ hky, SbFix *__restrict__ partial,
                                                                  unsigned long long *__restrict__ guard) {
    extern __shared__ double hk_s[]; // [n_states*k][2][25]
    __shared__ SbFix red_s[SBL_THREADS / 64];
    const uint32_t n_tab = n_states * k * 2 * SB_NCNT;
    for (uint32_t i = threadIdx.x; i < n_tab; i += SBL_THREADS) hk_s[i] = hky[i];
    __syncthreads();
    const uint32_t R = t.n_reads;
    for (uint32_t e = 0; e < n_states; ++e) {
        SbFix sum{0, 0, 0.0};
        unsigned long long bad = 0;
        for (uint32_t r = blockIdx.x * SBL_THREADS + threadIdx.x; r < R; r += gridDim.x * SBL_THREADS) {
            if (!t.ok[r]) continue;
            sb_fix_add(sum, sb_read_term(t, r, k, src + (size_t)e * k, hk_s + (size_t)e * k * 2 * SB_NCNT, bad));
        }
        if (bad) atomicAdd(&guard[e], bad);
        sb_fix_block_store(sum, red_s, &partial[(size_t)e * gridDim.x + blockIdx.x]);
    }
}

// One wave per state folds the workgroups' sums (integers: any order gives the same bits; the first version walked all
// partials on one thread: 79 us of dependent loads per refresh).
__global__ __launch_bounds__(64) void sb_finish_kernel(const SbFix *__restrict__ partial, uint32_t n_blocks, uint32_t n_states,
                                                       double *__restrict__ out, double *__restrict__ out2, SbFix *__restrict__ out_fix) {
    const uint32_t e = blockIdx.x;
    if (e >= n_states) return;
    const SbFix s = sb_fix_fold(partial + (size_t)e * n_blocks, n_blocks);
    if (threadIdx.x == 0) {
        const double v = sb_fix_value(s);
        out[e] = v;
        if (out2) out2[e] = v; // the caller's device buffer (handed to RCCL)
        if (out_fix) out_fix[e] = s;
    }
}

// Everything one workgroup writes for another (or for the host) goes and comes as RELAXED atomic accesses of the scope that reaches the
// reader (they pass the caches on their own), ordered by the issuing thread's wait for its own stores: an acquire / release FENCE of agent
// scope on this device writes back and invalidates the whole L2 of the XCD it runs on -- the first version, with fences, took 107 us for
// the reads' loop the launched kernel does in 76, and 26 us for the fold.
__device__ __forceinline__ unsigned long long sbr_ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sbr_st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long sbr_ld_sys(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sbr_st_sys(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sbr_stores_done() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); } // (the thread's stores are acknowledged)

// The chain driver's refresh: one launch, no copies.  The sources and frequencies arrive as kernel arguments and every block builds the
// HKY table it needs in LDS (150 entries for k = 3); the last workgroup to finish folds the workgroups' sums (integers: any order gives
// the same bits as sb_finish_kernel's) and stores the log-likelihood, the sums and the guard count straight into pinned host memory,
// then the refresh's number, which the host watches.  An MCMC iteration is launch bound at typical read counts (5 us of kernel time at
// 20k reads).  (The fold was a second launch -- sb_finish_host_kernel -- through round 4: behind agent-scope fences the last workgroup's
// fold measured slower than that launch; with the relaxed accesses above it is 4 us against the launch's 7 + the gap before it.)
__global__ __launch_bounds__(SBL_THREADS) void sb_refresh_fused_kernel(SbTablesDev t, uint32_t n_states, uint32_t k, SbFusedArgs a, SbFix *partial,
                                                                        unsigned long long *guard, unsigned int *ticket, double *out_host,
                                                                        unsigned long long *guard_host, SbFix *fix_host, unsigned long long *seq_host,
                                                                        unsigned long long seq) {
    __shared__ double hk_s[SB_FUSED_MAX_K * 2 * SB_NCNT];
    __shared__ SbSourceDev src_s[SB_FUSED_MAX_K];
    __shared__ SbFix red_s[SBL_THREADS / 64];
    __shared__ uint32_t last_s;
    const uint32_t ne = n_states * k; // <= SB_FUSED_MAX_K
#pragma unroll
    for (uint32_t y = 0; y < SB_FUSED_MAX_K; ++y)
        if (threadIdx.x == y && y < ne) src_s[y] = a.src[y]; // constant indices into the kernel arguments
    // the table in two steps: the 20 transition log-probabilities of a (source, branch) once (sb_hky_entry computes the four its entry
    // needs: every one five times over the table), then the 25 folds over them -- the same functions on the same arguments, so the same
    // bits as sb_hky_entry's, in 1.3 - 1.8 us less per refresh.  (Round 4's attempt at splitting the entries' chains made the compiler
    // allocate 85 instead of 116 VGPRs and serialise the loads of the main loop, 82 -> 200 us at 1M reads; this one: 144, as before.)
    __shared__ double lp_s[SB_FUSED_MAX_K * 2 * 20];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ne * 2 * 20; i += SBL_THREADS) {
        const uint32_t e = i / 40, which = (i / 20) & 1u, ref = (i % 20) / 4, bpo = i % 4;
        lp_s[i] = sb_hky_logp(which ? src_s[e].t1 : src_s[e].t2, (int)ref, (int)bpo, a.freqs7);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ne * 2 * SB_NCNT; i += SBL_THREADS) {
        const uint32_t ew = i / SB_NCNT, j = i % SB_NCNT;
        hk_s[i] = sb_hky_fold(&lp_s[ew * 20 + (j / 5) * 4], (int)(j % 5), a.con);
    }
    __syncthreads();
    const uint32_t R = t.n_reads;
    for (uint32_t e = 0; e < n_states; ++e) { // the chains of one source count advance together: one state each
        SbFix sum{0, 0, 0.0};
        unsigned long long bad = 0;
        for (uint32_t r = blockIdx.x * SBL_THREADS + threadIdx.x; r < R; r += gridDim.x * SBL_THREADS) {
            if (!t.ok[r]) continue;
            sb_fix_add(sum, sb_read_term(t, r, k, src_s + (size_t)e * k, hk_s + (size_t)e * k * 2 * SB_NCNT, bad));
        }
        if (bad) atomicAdd(&guard[e], bad);
        sum = sb_fix_wave_sum(sum);
        if ((threadIdx.x & 63) == 0) red_s[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            SbFix s2{0, 0, 0.0};
            for (int w = 0; w < SBL_THREADS / 64; ++w) {
                s2.hi += red_s[w].hi;
                s2.lo += red_s[w].lo;
                s2.nf += red_s[w].nf;
            }
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(&partial[(size_t)e * gridDim.x + blockIdx.x]);
            sbr_st(dst, (unsigned long long)s2.hi);
            sbr_st(dst + 1, s2.lo);
            sbr_st(dst + 2, (unsigned long long)__double_as_longlong(s2.nf));
        }
        __syncthreads();
    }
    // ---- the last workgroup to get here folds
    if (threadIdx.x == 0) {
        sbr_stores_done(); // (its partial sums are out, its guard counts too)
        last_s = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last_s) return;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t e = wave; e < n_states; e += SBL_THREADS / 64) { // the results first ...
        SbFix f{0, 0, 0.0};
        for (uint32_t i = lane; i < gridDim.x; i += 64) {
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&partial[(size_t)e * gridDim.x + i]);
            f.hi += (long long)sbr_ld(src);
            f.lo += sbr_ld(src + 1);
            f.nf += __longlong_as_double((long long)sbr_ld(src + 2));
        }
        f = sb_fix_wave_sum(f);
        if (lane == 0) {
            sbr_st_sys(reinterpret_cast<unsigned long long *>(&out_host[e]), (unsigned long long)__double_as_longlong(sb_fix_value(f)));
            if (fix_host) {
                unsigned long long *fx = reinterpret_cast<unsigned long long *>(&fix_host[e]);
                sbr_st_sys(fx, (unsigned long long)f.hi);
                sbr_st_sys(fx + 1, f.lo);
                sbr_st_sys(fx + 2, (unsigned long long)__double_as_longlong(f.nf));
            }
            sbr_st_sys(&guard_host[e], sbr_ld(&guard[e]));
            sbr_st(&guard[e], 0ull); // ready for the next refresh (stream ordered)
            sbr_stores_done();
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // ... then the number the host may be watching instead of waiting for the stream
    if (seq_host && threadIdx.x < n_states) sbr_st_sys(&seq_host[threadIdx.x], seq);
}

// The refresh as a resident kernel (MCMC.cpp:738-993 asks for one likelihood per iteration, and the iteration cannot go on before it has
// it: at 1 M reads the two launches above and the wait for them were 22 us around 76 us of kernel; at 20 k reads 25 us around 5).  The
// workgroups stay on the device between refreshes: workgroup 0 watches a mailbox in pinned host memory (the host writes the sources,
// then the refresh's number), copies the arguments into device memory and publishes the number there for the others; each workgroup
// computes its partial sums as sb_refresh_fused_kernel does, the last one to finish folds them and writes the results and the number into
// pinned host memory.  A refresh costs two trips over the host link and one through L2 instead of two launches.
// The kernel leaves when the host says so (any other call on the context: the tables may move), and BY ITSELF when no refresh came for
// idle_ticks of the 100 MHz clock -- a host that died or forgot leaves nothing running.  Its grid must be resident as a whole (the
// launcher sizes it; a workgroup that never started would be waited for by the last-to-finish count).
struct SbMailbox { // pinned host memory
    SbFusedArgs a;
    uint32_t n_states, k;
    unsigned long long seq;    // host: the number of the refresh whose arguments stand above (written last)
    unsigned long long stop;   // host: non-zero = leave
    unsigned long long exited; // device: the launch id, once the kernel of that launch has left
    unsigned long long busy;   // device, written when the kernel leaves: ticks between seeing a number and publishing its results, summed
    unsigned long long served; // device, likewise: refreshes served (both summed over the launches: vgan_sb_kernel_ms)
    unsigned long long stamp[8]; // device, likewise: the last refresh's way through the kernel (developer aid, 100 MHz ticks since it was seen)
};
struct SbResident { // device memory
    SbFusedArgs a;
    uint32_t n_states, k;
    unsigned long long seq, stop, t_seen;
    unsigned int ticket;
    unsigned long long busy, served, stamp[8];
};

static_assert(sizeof(SbFusedArgs) % 8 == 0 && sizeof(SbSourceDev) % 8 == 0 && sizeof(SbFix) == 24, "the resident kernel moves these as 64-bit words");

__global__ __launch_bounds__(SBL_THREADS) void sb_refresh_resident_kernel(SbTablesDev t, SbMailbox *mb, SbResident *rs, SbFix *partial,
                                                                           unsigned long long *guard, double *out_host, unsigned long long *guard_host,
                                                                           SbFix *fix_host, unsigned long long *seq_host, unsigned long long done,
                                                                           unsigned long long idle_ticks, unsigned long long launch_id) {
    __shared__ double hk_s[SB_FUSED_MAX_K * 2 * SB_NCNT];
    __shared__ SbSourceDev src_s[SB_FUSED_MAX_K];
    __shared__ double fr_s[8];
    __shared__ double lp_s[SB_FUSED_MAX_K * 2 * 20];
    __shared__ SbFix red_s[SBL_THREADS / 64];
    __shared__ unsigned long long seq_s;
    __shared__ uint32_t stop_s, last_s, nk_s[2];
    const uint32_t R = t.n_reads;
    constexpr uint32_t n_arg_words = (uint32_t)(sizeof(SbFusedArgs) / 8) + 1; // a; {n_states, k}
    constexpr uint32_t W_SRC = (uint32_t)(sizeof(SbSourceDev) * SB_FUSED_MAX_K / 8);
    static_assert(n_arg_words == W_SRC + 9 && n_arg_words <= SBL_THREADS, "sources, seven frequencies, con, the two counts: a word per thread");
    unsigned long long *rs_w = reinterpret_cast<unsigned long long *>(rs);
    for (;;) {
        // ---- the next refresh's number (or the order to leave)
        if (threadIdx.x == 0) {
            unsigned long long s = done;
            uint32_t stop = 0;
            if (blockIdx.x == 0) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    s = sbr_ld_sys(&mb->seq);
                    if (s != done) break;
                    if (sbr_ld_sys(&mb->stop) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > idle_ticks) {
                        stop = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (!stop) sbr_st(&rs->t_seen, __builtin_amdgcn_s_memrealtime());
            } else {
                for (;;) {
                    s = sbr_ld(&rs->seq);
                    if (s != done) break;
                    if (sbr_ld(&rs->stop) != 0) {
                        stop = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            seq_s = s;
            stop_s = stop;
        }
        __syncthreads();
        if (stop_s) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                sbr_st(&rs->stop, 1ull);
                // (a refresh the others are still finishing is not in these sums yet: rs keeps them for the next launch)
                sbr_st_sys(&mb->busy, sbr_ld(&rs->busy));
                sbr_st_sys(&mb->served, sbr_ld(&rs->served));
                for (int q = 0; q < 8; ++q) sbr_st_sys(&mb->stamp[q], sbr_ld(&rs->stamp[q]));
                sbr_stores_done();
                sbr_st_sys(&mb->exited, launch_id);
            }
            return;
        }
        const unsigned long long seq = seq_s;
        // ---- the arguments: workgroup 0 has them from the host and passes them on through device memory
        if (blockIdx.x == 0) {
            const unsigned long long *from = reinterpret_cast<const unsigned long long *>(mb);
            unsigned long long w = 0;
            if (threadIdx.x < n_arg_words) {
                w = sbr_ld_sys(from + threadIdx.x);
                sbr_st(rs_w + threadIdx.x, w);
                sbr_stores_done();
            }
            if (threadIdx.x < n_arg_words) { // (its own copy straight from the registers)
                if (threadIdx.x < W_SRC) reinterpret_cast<unsigned long long *>(src_s)[threadIdx.x] = w;
                else if (threadIdx.x < W_SRC + 7) fr_s[threadIdx.x - W_SRC] = __longlong_as_double((long long)w);
                else if (threadIdx.x == W_SRC + 7) fr_s[7] = __longlong_as_double((long long)w); // con
                else nk_s[0] = (uint32_t)w, nk_s[1] = (uint32_t)(w >> 32);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                sbr_st(&rs->stamp[0], __builtin_amdgcn_s_memrealtime() - sbr_ld(&rs->t_seen)); // the arguments are over
                sbr_st(&rs->seq, seq);
            }
        } else {
            if (threadIdx.x < n_arg_words) {
                const unsigned long long w = sbr_ld(rs_w + threadIdx.x);
                if (threadIdx.x < W_SRC) reinterpret_cast<unsigned long long *>(src_s)[threadIdx.x] = w;
                else if (threadIdx.x < W_SRC + 7) fr_s[threadIdx.x - W_SRC] = __longlong_as_double((long long)w);
                else if (threadIdx.x == W_SRC + 7) fr_s[7] = __longlong_as_double((long long)w);
                else nk_s[0] = (uint32_t)w, nk_s[1] = (uint32_t)(w >> 32);
            }
            __syncthreads();
        }
        // ---- as sb_refresh_fused_kernel
        const uint32_t n_states = nk_s[0], k = nk_s[1], ne = n_states * k;
        const double con = fr_s[7];
        for (uint32_t i = threadIdx.x; i < ne * 2 * 20; i += SBL_THREADS) { // (the table in two steps, as sb_refresh_fused_kernel)
            const uint32_t e = i / 40, which = (i / 20) & 1u, ref = (i % 20) / 4, bpo = i % 4;
            lp_s[i] = sb_hky_logp(which ? src_s[e].t1 : src_s[e].t2, (int)ref, (int)bpo, fr_s);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < ne * 2 * SB_NCNT; i += SBL_THREADS) {
            const uint32_t ew = i / SB_NCNT, j = i % SB_NCNT;
            hk_s[i] = sb_hky_fold(&lp_s[ew * 20 + (j / 5) * 4], (int)(j % 5), con);
        }
        __syncthreads();
        const unsigned long long t_tab = __builtin_amdgcn_s_memrealtime();
        for (uint32_t e = 0; e < n_states; ++e) {
            SbFix sum{0, 0, 0.0};
            unsigned long long bad = 0;
            for (uint32_t r = blockIdx.x * SBL_THREADS + threadIdx.x; r < R; r += gridDim.x * SBL_THREADS) {
                if (!t.ok[r]) continue;
                sb_fix_add(sum, sb_read_term(t, r, k, src_s + (size_t)e * k, hk_s + (size_t)e * k * 2 * SB_NCNT, bad));
            }
            if (bad) atomicAdd(&guard[e], bad);
            sum = sb_fix_wave_sum(sum);
            if ((threadIdx.x & 63) == 0) red_s[threadIdx.x >> 6] = sum;
            __syncthreads();
            if (threadIdx.x == 0) {
                SbFix s2{0, 0, 0.0};
                for (int w = 0; w < SBL_THREADS / 64; ++w) {
                    s2.hi += red_s[w].hi;
                    s2.lo += red_s[w].lo;
                    s2.nf += red_s[w].nf;
                }
                unsigned long long *dst = reinterpret_cast<unsigned long long *>(&partial[(size_t)e * gridDim.x + blockIdx.x]);
                sbr_st(dst, (unsigned long long)s2.hi);
                sbr_st(dst + 1, s2.lo);
                sbr_st(dst + 2, (unsigned long long)__double_as_longlong(s2.nf));
            }
            __syncthreads();
        }
        const unsigned long long t_loop = __builtin_amdgcn_s_memrealtime();
        // ---- the last workgroup to get here folds
        if (threadIdx.x == 0) {
            sbr_stores_done(); // (its partial sums are out, its guard counts too)
            last_s = __hip_atomic_fetch_add(&rs->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (last_s) {
            const unsigned long long t_last = __builtin_amdgcn_s_memrealtime();
            const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
            for (uint32_t e = wave; e < n_states; e += SBL_THREADS / 64) { // the results first ...
                SbFix f{0, 0, 0.0};
                for (uint32_t i = lane; i < gridDim.x; i += 64) {
                    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&partial[(size_t)e * gridDim.x + i]);
                    f.hi += (long long)sbr_ld(src);
                    f.lo += sbr_ld(src + 1);
                    f.nf += __longlong_as_double((long long)sbr_ld(src + 2));
                }
                f = sb_fix_wave_sum(f);
                if (lane == 0) {
                    sbr_st_sys(reinterpret_cast<unsigned long long *>(&out_host[e]), (unsigned long long)__double_as_longlong(sb_fix_value(f)));
                    unsigned long long *fx = reinterpret_cast<unsigned long long *>(&fix_host[e]);
                    sbr_st_sys(fx, (unsigned long long)f.hi);
                    sbr_st_sys(fx + 1, f.lo);
                    sbr_st_sys(fx + 2, (unsigned long long)__double_as_longlong(f.nf));
                    sbr_st_sys(&guard_host[e], sbr_ld(&guard[e]));
                    sbr_st(&guard[e], 0ull);
                    sbr_stores_done();
                }
            }
            const unsigned long long t_fold = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0) {
                __hip_atomic_store(&rs->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (the next refresh's counts start after the host has seen this one's number)
                sbr_stores_done();
            }
            __syncthreads();
            // ... then the number the host is watching
            if (threadIdx.x < n_states) sbr_st_sys(&seq_host[threadIdx.x], seq);
            if (threadIdx.x == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime(), t_seen = sbr_ld(&rs->t_seen);
                sbr_st(&rs->stamp[1], t_tab - t_seen), sbr_st(&rs->stamp[2], t_loop - t_seen), sbr_st(&rs->stamp[3], t_last - t_seen);
                sbr_st(&rs->stamp[4], t_fold - t_seen), sbr_st(&rs->stamp[5], now - t_seen);
                sbr_st(&rs->busy, sbr_ld(&rs->busy) + now - t_seen);
                sbr_st(&rs->served, sbr_ld(&rs->served) + 1);
            }
        }
        done = seq;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- read summaries
// analyse_GAM's mostProbPath (getLCAfromGAM.h:563-579): the paths holding the read's highest pathMap value.  One lane per
// read walks the pm rows (coalesced across lanes); best = that path when exactly one path holds the maximum, else -1.
// Paths with the same support pattern over a read have bit-identical sums (same terms, same order), so the ties of the
// reference are ties here.
__global__ __launch_bounds__(256) void sb_best_path_kernel(SbTablesDev t, uint32_t n_paths, int32_t *__restrict__ best,
                                                            unsigned long long *__restrict__ sig_count,
                                                            unsigned long long *__restrict__ n_ok) {
    __shared__ uint32_t hist_s[SB_MAX_PATHS];
    __shared__ uint32_t ok_s;
    for (uint32_t i = threadIdx.x; i < SB_MAX_PATHS; i += blockDim.x) hist_s[i] = 0;
    if (threadIdx.x == 0) ok_s = 0;
    __syncthreads();
    const uint32_t R = t.n_reads;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
        int32_t b = -1;
        if (t.ok[r]) {
            double hi = t.pm[r];
            uint32_t arg = 0, ties = 1;
            for (uint32_t p = 1; p < n_paths; ++p) {
                const double v = t.pm[(size_t)p * R + r];
                if (v > hi) {
                    hi = v;
                    arg = p;
                    ties = 1;
                } else if (v == hi)
                    ++ties;
            }
            if (ties == 1) {
                b = (int32_t)arg;
                atomicAdd(&hist_s[arg], 1u);
            }
            atomicAdd(&ok_s, 1u);
        }
        if (best) best[r] = b;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_paths; i += blockDim.x)
        if (hist_s[i]) atomicAdd(&sig_count[i], (unsigned long long)hist_s[i]);
    if (threadIdx.x == 0 && ok_s) atomicAdd(n_ok, (unsigned long long)ok_s);
}

// soibean.cpp:737-756: per read inter = log_freq + pathMap[paths[0]], then inter = oplusInitnatl(inter, log_freq +
// pathMap[paths[j]]) for the further sources; block partials in a fixed order as the refresh kernel's
__global__ __launch_bounds__(SBL_THREADS) void sb_mixture_kernel(SbTablesDev t, uint32_t n, const int32_t *__restrict__ paths,
                                                                  double log_freq, SbFix *__restrict__ partial) {
    __shared__ SbFix red_s[SBL_THREADS / 64];
    __shared__ int32_t path_s[SB_MAX_PATHS];
    for (uint32_t i = threadIdx.x; i < n; i += SBL_THREADS) path_s[i] = paths[i];
    __syncthreads();
    const uint32_t R = t.n_reads;
    SbFix sum{0, 0, 0.0};
    for (uint32_t r = blockIdx.x * SBL_THREADS + threadIdx.x; r < R; r += gridDim.x * SBL_THREADS) {
        if (!t.ok[r]) continue;
        double inter = log_freq + t.pm[(size_t)path_s[0] * R + r];
        for (uint32_t j = 1; j < n; ++j) {
            const double y = log_freq + t.pm[(size_t)path_s[j] * R + r];
            if (inter == 0.0) inter = y; // oplusInitnatl: a running value of 0 means "empty" 
            else inter = fmax(inter, y) + log1p(exp(-fabs(inter - y)));
        }
        sb_fix_add(sum, inter);
    }
    sb_fix_block_store(sum, red_s, &partial[blockIdx.x]);
}

// ---------------------------------------------------------------------------------------------- launchers
void launch_sb_refresh_fused(const SbTablesDev &t, uint32_t n_states, uint32_t k, const SbFusedArgs &a, SbFix *partial, uint32_t n_blocks,
                             unsigned long long *guard, unsigned int *ticket, double *out_host, unsigned long long *guard_host, SbFix *fix_host,
                             unsigned long long *seq_host, unsigned long long seq, hipStream_t st) {
    hipLaunchKernelGGL(sb_refresh_fused_kernel, dim3(n_blocks), dim3(SBL_THREADS), 0, st, t, n_states, k, a, partial, guard, ticket, out_host, guard_host,
                       fix_host, seq_host, seq);
}

// the workgroups of sb_refresh_fused_kernel the device holds at once: a grid of exactly that many streams the reads in one round (1024
// workgroups on 768 places were a full round and a third of one: 86 against 81 us per refresh at 1 M reads)
uint32_t sb_refresh_grid(int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sb_refresh_fused_kernel, SBL_THREADS, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || per_cu <= 0 || cus <= 0)
        return 1024u;
    return (uint32_t)per_cu * (uint32_t)cus;
}
size_t sb_mailbox_bytes() { return sizeof(SbMailbox); }
size_t sb_resident_bytes() { return sizeof(SbResident); }
void sb_mailbox_post(void *mailbox, uint32_t n_states, uint32_t k, const SbFusedArgs &a, unsigned long long seq) {
    SbMailbox *mb = static_cast<SbMailbox *>(mailbox);
    mb->a = a;
    mb->n_states = n_states;
    mb->k = k;
    __atomic_store_n(&mb->seq, seq, __ATOMIC_RELEASE);
}
void sb_mailbox_idle(void *mailbox, unsigned long long done) { __atomic_store_n(&static_cast<SbMailbox *>(mailbox)->seq, done, __ATOMIC_RELEASE); }
void sb_mailbox_stop(void *mailbox, bool on) { __atomic_store_n(&static_cast<SbMailbox *>(mailbox)->stop, on ? 1ull : 0ull, __ATOMIC_RELEASE); }
unsigned long long sb_mailbox_exited(const void *mailbox) { return __atomic_load_n(&static_cast<const SbMailbox *>(mailbox)->exited, __ATOMIC_ACQUIRE); }
void sb_mailbox_busy(void *mailbox, unsigned long long *ticks, unsigned long long *served, unsigned long long stamp[8]) {
    SbMailbox *mb = static_cast<SbMailbox *>(mailbox);
    *ticks = __atomic_load_n(&mb->busy, __ATOMIC_ACQUIRE);
    *served = __atomic_load_n(&mb->served, __ATOMIC_ACQUIRE);
    if (stamp)
        for (int q = 0; q < 8; ++q) stamp[q] = mb->stamp[q];
}
// the largest grid the device holds as a whole, capped (0: the query failed)
uint32_t sb_resident_grid(int device, uint32_t want) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sb_refresh_resident_kernel, SBL_THREADS, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || per_cu <= 0 || cus <= 0)
        return 0;
    // (half of what fits at most: a second context's kernel, or any other kernel of the process, still finds room beside it)
    return std::min<uint32_t>(want, (uint32_t)per_cu * (uint32_t)cus / 2u);
}
__global__ void sb_resident_init_kernel(SbResident *rs, unsigned long long done) {
    rs->seq = done;
    rs->stop = 0;
    rs->t_seen = 0;
    rs->ticket = 0;
}
// (busy / served / stamp live on from launch to launch: the block is zeroed when it is made)
void launch_sb_refresh_resident(const SbTablesDev &t, void *mailbox, void *resident, SbFix *partial, uint32_t n_blocks, unsigned long long *guard,
                                double *out_host, unsigned long long *guard_host, SbFix *fix_host, unsigned long long *seq_host, unsigned long long done,
                                unsigned long long idle_ticks, unsigned long long launch_id, hipStream_t st) {
    hipLaunchKernelGGL(sb_resident_init_kernel, dim3(1), dim3(1), 0, st, static_cast<SbResident *>(resident), done);
    hipLaunchKernelGGL(sb_refresh_resident_kernel, dim3(n_blocks), dim3(SBL_THREADS), 0, st, t, static_cast<SbMailbox *>(mailbox),
                       static_cast<SbResident *>(resident), partial, guard, out_host, guard_host, fix_host, seq_host, done, idle_ticks, launch_id);
}

void launch_sb_best_paths(const SbTablesDev &t, uint32_t n_paths, int32_t *best, unsigned long long *sig_count,
                          unsigned long long *n_ok, hipStream_t st) {
    if (t.n_reads == 0) return;
    const uint32_t blocks = min((t.n_reads + 255u) / 256u, 2048u);
    hipLaunchKernelGGL(sb_best_path_kernel, dim3(blocks), dim3(256), 0, st, t, n_paths, best, sig_count, n_ok);
}

void launch_sb_mixture(const SbTablesDev &t, uint32_t n, const int32_t *paths, double log_freq, SbFix *partial, uint32_t n_blocks,
                       double *out, SbFix *out_fix, hipStream_t st) {
    hipLaunchKernelGGL(sb_mixture_kernel, dim3(n_blocks), dim3(SBL_THREADS), 0, st, t, n, paths, log_freq, partial);
    hipLaunchKernelGGL(sb_finish_kernel, dim3(1), dim3(64), 0, st, partial, n_blocks, 1u, out, (double *)nullptr, out_fix);
}

void launch_sb_precompute(const SbGraphDev &g, const SbBatchDev &b, const SbTablesDev &t, double *stage_pm, uint16_t *stage_cnt,
                          uint32_t chunk_reads, unsigned long long *n_bad, hipStream_t st) {
    if (b.n_reads == 0) return;
    const uint32_t pp = (g.n_paths + 63) / 64, P = g.n_paths;
    uint32_t ppad_log2 = 0;
    while ((1u << ppad_log2) < P) ++ppad_log2;
    const int mw32 = P <= 32 ? 1 : 2; // 32-bit words of a node's path mask the column kernel keeps
    const size_t slice = sbc_slice_bytes(mw32 == 1 ? sizeof(SbcSlice<1>) : sizeof(SbcSlice<2>), SB_NCNT << ppad_log2);
    const uint32_t pen = (uint32_t)std::max(1, g.penalty);
    const uint32_t pen_magic = pen == 1 ? 0xFFFFFFFFu : (uint32_t)((0x100000000ull + pen - 1) / pen); // floor(x / pen) = mulhi(x, magic), x < 2^16
    const char *force = getenv("VGAN_SB_PRECOMPUTE"); // developer aid / tests: "segments" keeps every read on the segment kernel
    // up to 64 paths: the pair histogram is 25 x 64 counters and a (slot, path) sweep at most one step per column.  Beyond
    // (210 paths measured: 32.4 ms per 500k reads against the segment kernel's 29.8) the segment kernel keeps the reads.
    const bool cols = !(force && !strcmp(force, "segments")) && P <= 64u && slice * SBC_WAVES <= 60u * 1024u;
    for (uint32_t r0 = 0; r0 < b.n_reads; r0 += chunk_reads) {
        const uint32_t r1 = std::min(b.n_reads, r0 + chunk_reads), n = r1 - r0;
        if (cols) { // a lane per column; the reads beyond its capacities are left marked for the kernel below
            const uint32_t cblocks = std::min((n + SBC_WAVES - 1) / SBC_WAVES, 256u * 16u);
            if (mw32 == 1)
                hipLaunchKernelGGL(sb_precompute_cols_kernel<1>, dim3(cblocks), dim3(SBC_WAVES * 64), slice * SBC_WAVES, st, g, b, t, r0, r1, stage_pm,
                                   stage_cnt, n_bad, ppad_log2, pen_magic);
            else
                hipLaunchKernelGGL(sb_precompute_cols_kernel<2>, dim3(cblocks), dim3(SBC_WAVES * 64), slice * SBC_WAVES, st, g, b, t, r0, r1, stage_pm,
                                   stage_cnt, n_bad, ppad_log2, pen_magic);
        }
        const uint32_t blocks = std::min((n + SBP_WAVES - 1) / SBP_WAVES, 256u * 8u);
        if (pp <= 1) hipLaunchKernelGGL(sb_precompute_kernel<1>, dim3(blocks), dim3(SBP_WAVES * 64), 0, st, g, b, t, r0, r1, stage_pm, stage_cnt, n_bad, cols);
        else if (pp == 2) hipLaunchKernelGGL(sb_precompute_kernel<2>, dim3(blocks), dim3(SBP_WAVES * 64), 0, st, g, b, t, r0, r1, stage_pm, stage_cnt, n_bad, cols);
        else if (pp == 3) hipLaunchKernelGGL(sb_precompute_kernel<3>, dim3(blocks), dim3(SBP_WAVES * 64), 0, st, g, b, t, r0, r1, stage_pm, stage_cnt, n_bad, cols);
        else hipLaunchKernelGGL(sb_precompute_kernel<4>, dim3(blocks), dim3(SBP_WAVES * 64), 0, st, g, b, t, r0, r1, stage_pm, stage_cnt, n_bad, cols);
        hipLaunchKernelGGL(sb_transpose_kernel<double>, dim3((n + 63) / 64, (P + 63) / 64), dim3(256), 0, st, stage_pm, t.pm, n, P, P, 1u, r0, t.n_reads);
        hipLaunchKernelGGL(sb_transpose_kernel<uint16_t>, dim3((n + 63) / 64, (P * SB_NCNT + 63) / 64), dim3(256), 0, st, stage_cnt, t.cnt, n,
                           P * SB_NCNT, P, SB_NCNT, r0, t.n_reads);
    }
}

void launch_sb_hky(uint32_t n_entries, const SbSourceDev *src, double con, const double *freqs7, double *hky,
                   unsigned long long *guard, uint32_t n_states, hipStream_t st) {
    const uint32_t n = n_entries * 2 * SB_NCNT; // >= n_states
    hipLaunchKernelGGL(sb_hky_kernel, dim3((n + 127) / 128), dim3(128), 0, st, n_entries, src, con, freqs7, hky, guard, n_states);
}

void launch_sb_loglike(const SbTablesDev &t, uint32_t n_paths, uint32_t n_states, uint32_t k, const SbSourceDev *src,
                       const double *hky, SbFix *partial, uint32_t n_blocks, double *out, double *out2, SbFix *out_fix,
                       unsigned long long *guard, hipStream_t st) {
    (void)n_paths;
    const size_t lds = (size_t)n_states * k * 2 * SB_NCNT * sizeof(double);
    hipLaunchKernelGGL(sb_loglike_kernel, dim3(n_blocks), dim3(SBL_THREADS), lds, st, t, n_states, k, src, hky, partial, guard);
    hipLaunchKernelGGL(sb_finish_kernel, dim3(n_states), dim3(64), 0, st, partial, n_blocks, n_states, out, out2, out_fix);
}

} // namespace vgan
#include "module_anchor.h"
const void *vgan::anchor_sb_kernels() { return (const void *)&vgan::sb_finish_kernel; }
