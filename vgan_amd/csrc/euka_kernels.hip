// euka per-read kernel for gfx950: the body of readGAM3's per-alignment lambda (reference
// src/readGAM_Euka.h:67-577) with Baseshift::baseshift_calc (src/baseshift.cpp:57-88).
//
// Sixteen lanes (one DPP row) per read, four reads per wave; a lane takes every 16th alignment column of its read.
// Model 1 per column is pre[4] (divergence) x the 4x4 damage matrix selected from the 5'/3' tables, marginalised over
// the sequencing error: log(sum_b post[b] * w[b]) -- one log instead of the reference's four logs folded with
// oplusInitnatl (identical value; a fold whose running value is exactly 0 cannot occur since every weight is < 1).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "device_math.h"
#include "euka_device.h"

namespace vgan {

__device__ const LogTabEntry euka_log_table[64] = {VGAN_LOG_TABLE};

__device__ __forceinline__ int acgt_index(uint32_t c) { // "ACGT" order; -1 otherwise
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}

__device__ __forceinline__ bool is_rare(uint32_t c) { // Euka.cpp:472-486: W M K R Y B D H V
    const uint32_t d = c - 65u;
    // B=1 D=3 H=7 K=10 M=12 R=17 V=21 W=22 Y=24
    return d < 26u && ((0x162148Au >> d) & 1u);
}

__device__ __forceinline__ double base_freq_log(uint32_t c) { // Euka.cpp:446-450 (log values), 0 when unassigned
    return c == 'A' ? -1.0138622165021247 : c == 'C' ? -1.5714535401584102 : c == 'G' ? -2.147215156762058
         : c == 'N' ? -1.3862943611198906 : c == 'T' ? -1.1633588314406809 : 0.0;
}

__device__ __forceinline__ double tT_ratio(int g, int b) { // Euka.cpp:453-468
    return g == b ? 1.0 : ((g ^ b) == 2 ? 0.95238 : 0.02381); // A<->G (0,2), C<->T (1,3) are transitions
}

constexpr int EK_WAVES = 4;
constexpr int EK_GROUP = 16;                              // lanes per read: a DPP row
constexpr int EK_READS_PER_WAVE = 64 / EK_GROUP;          // 4 reads per wave,
constexpr int EK_READS_PER_BLOCK = EK_WAVES * EK_READS_PER_WAVE; // 16 per workgroup
#ifndef EK_BG
#define EK_BG 8 // lanes per read in the column stage (16: a DPP row per read, four reads per step; 8: eight reads per step)
#endif
constexpr uint32_t EK_DMG_LDS_PAIRS = 64;                 // (5' row, 3' row) pairs kept in LDS (512 bytes each)
#ifndef EK_ORDER_SPAN
// A block's reads are taken in the order of their lengths within spans of this many neighbours.  Over the whole block (64; round 5) a
// 128-byte line of the strings -- it holds the ends of two neighbouring reads -- is wanted twice, up to a whole block's column loops
// apart, and what the waves of an XCD hold in between (512 waves x 14 KB of strings) is more than its 4 MB of L2: the launch fetched
// 1.37 x its algorithmic bytes (524 MB against 384; `tools/dev/ek_order.sh`: spans 1 / 8 / 16: 388-390 MB, 32: 415-419 MB, 64: 524 MB, the
// same 0.317-0.321 ms at every span from 16 up, 0.324 without the ordering).
#define EK_ORDER_SPAN 32u
#endif
constexpr int EK_ACC_LTP = 8;                             // lengthToProf up to which a wave keeps the base shifts in LDS,
constexpr int EK_ACC_BINS = 32;                           // bins per clade up to which it keeps the coverage there

// A wave's accumulators for ONE clade (the batch is sorted by node id, clades own contiguous node ranges: a wave's reads
// are of one clade for long stretches).  Flushed to the global tables when the clade changes and when the wave ends.
struct EkAcc {
    uint32_t shift[2 * EK_ACC_LTP * 16];
    double cov[EK_ACC_BINS];
    double logsum;
    uint32_t n_like;
    int32_t count;
};

// clade_like and its log for one read (readGAM_Euka.h:485-492).  Out of line: inlined, libm's constants are hoisted out of
// the kernel's loops and held in registers -- or spilled -- through the column loop.
struct EkLike {
    double like, log_like;
};
__device__ __attribute__((noinline)) EkLike ek_clade_like(double in, double out, double map_q) {
    double lse;
    if (in == 0.0) lse = out; // oplusInitnatl: a running value of 0 means "empty"
    else lse = fmax(in, out) + log1p(exp(-fabs(in - out)));
    const double like = map_q * exp(in - lse);
    return EkLike{like, log(like)};
}
// 1 - 10^(-mapq/10) for a negative mapq (cannot be encoded by a well-formed GAM; evaluated for completeness)
__device__ __attribute__((noinline)) double ek_map_q_cold(int32_t mapq) { return 1.0 - exp(-0.1 * mapq * 2.302585092994046); }

// A wave's block of 64 reads between the stages of the kernel (a lane per read / a row of 16 lanes per read)
struct EkBlk {
    uint32_t col0[64], ga[64], lq[64], q0[64], m0[64]; // ga: |graph| | |read columns| << 16; lq: |sequence| | |quality| << 16
    uint32_t bins[64];  // the clade's first bin (24 bits) | its number of bins << 24
    uint32_t flags[64]; // 1 reverse strand, 2 a column beyond the damage tables, 4 passed; the read's mappings << 16
    int32_t mapq[64], cn[64];
    double pd[64], in[64], out[64];
    uint8_t passed[64]; // the block's reads that passed, in order (stage D takes them four at a time)
    uint8_t order[64];  // the block's reads by their number of columns (stage B takes them four at a time)
};

// cnt += the number of lanes k of the DPP row whose value v has v - lo <= w (unsigned): sixteen subtractions that take their operand
// from lane k of the row
template <int K> __device__ __forceinline__ void ek_count_row(int32_t v, uint32_t lo, uint32_t w, uint32_t &cnt) {
    if constexpr (K < 16) {
        const uint32_t vk = (uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, false);
        cnt += vk - lo <= w ? 1u : 0u;
        ek_count_row<K + 1>(v, lo, w, cnt);
    }
}

// minimum / bitwise or over the 16 lanes of a DPP row, result in every lane of the row (row_ror 8/4/2/1)
__device__ __forceinline__ uint32_t row_min16(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, true));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, true));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ uint64_t row_or16(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x128, 0xf, 0xf, true), hi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x128, 0xf, 0xf, true);
    lo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x124, 0xf, 0xf, true), hi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x124, 0xf, 0xf, true);
    lo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x122, 0xf, 0xf, true), hi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x122, 0xf, 0xf, true);
    lo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x121, 0xf, 0xf, true), hi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x121, 0xf, 0xf, true);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// sum over the 8 lanes of half a DPP row, result in every lane of the half (row_half_mirror, then the two exchanges within a quad)
__device__ __forceinline__ double row_sum8(double v) {
    v += dpp_mov0<0x141, 0xf, true>(v); // lane i <-> 7 - i
    v += dpp_mov0<0xB1, 0xf, true>(v);  // quad_perm [1,0,3,2]
    v += dpp_mov0<0x4E, 0xf, true>(v);  // quad_perm [2,3,0,1]
    return v;
}

// sum over the 16 lanes of a DPP row, result in every lane of the row (row_ror 8/4/2/1)
__device__ __forceinline__ double row_sum16(double v) {
    v += dpp_mov0<0x128, 0xf, true>(v);
    v += dpp_mov0<0x124, 0xf, true>(v);
    v += dpp_mov0<0x122, 0xf, true>(v);
    v += dpp_mov0<0x121, 0xf, true>(v);
    return v;
}

// Four reads per wave, 16 lanes (one DPP row) per read, 16 alignment columns of each read per step.  The two
// quantities the reference carries serially along a read -- the damage position n (advanced on every non-gap read
// column, readGAM_Euka.h:457-461) and the softclip counter (:269) -- are prefix popcounts of the row's 16 ballot bits
// plus a carry, so the columns are independent; per-read sums are row reductions; the per-read epilogue (log-sum-exp,
// outputs, base shifts, bin coverage) runs for the wave's four reads at once.
template <bool DMG_LDS>
#ifndef EK_MIN_WAVES
#define EK_MIN_WAVES 4 // waves per SIMD the register allocation aims at
#endif
__global__ __launch_bounds__(EK_WAVES * 64, EK_MIN_WAVES) void euka_read_kernel(EukaDev d, EukaBatchDev b, EukaOutDev o) {
    // {w_miss, w_hit - w_miss} = {eps(Q) / 3, (1 - eps(Q)) - eps(Q) / 3}: from a table -- an fp64 division is eleven instructions,
    // one of them the quarter-rate reciprocal, per column (w_hit itself is only needed on a softclip column: 1 - d.qscore[q] there)
    __shared__ double2 qs_s[100];
    __shared__ LogTabEntry logtab_s[64];
    // per (5' row, 3' row) pair, read base rb and graph base g, the column's sum over the original bases with everything that does
    // not depend on the read's clade folded in:  p = sum_o pre[o] * (w_miss * rowsum[o] + dw * M[o][rb]),  pre = 1 - dist at g,
    // dist * 0.95238 at g ^ 2, dist * 0.02381 at the other two, is  w_miss * (rs + dist * D_rs) + dw * (M + dist * D_M)  with
    // rs = rowsum[g], M = M[g][rb], D_x = 0.95238 * x[g ^ 2] + 0.02381 * (x[g ^ 1] + x[g ^ 3]) - x[g]: an entry {M, rs, D_M, D_rs}
    // of 32 bytes -- two 16-byte reads and four fused multiply-adds per column where the four original bases took four reads
    // and thirteen operations (512 bytes per pair; the global table of euka_device.h keeps its 160)
    // (sized by the launch: per pair of the context's tables, not the 64 pairs the variant allows -- the workgroups a CU holds
    // follow its LDS)
    extern __shared__ double2 dmg_s[];
    // byte -> class: low nibble = ACGT index 0..3, else 8; high nibble = the rank of the lambda's special cases in
    // the order it tests them (readGAM_Euka.h:236-280): 0 'N', 1 '-', 2 rare IUPAC code, 3 'S', 4 none
    __shared__ uint8_t cls_s[256];
    __shared__ double bfl_s[16]; // base_freq log of the read base by its low nibble; 0 unless A C G T / 'N' (slot 9)
    __shared__ uint8_t up4_s[256]; // byte -> ACGT index of its upper-case form, else 0x80 (Baseshift::baseshift_calc folds the case)
    __shared__ EkAcc acc_s[EK_WAVES];
    __shared__ EkBlk blk_s[EK_WAVES];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        const int ai = acgt_index((uint32_t)i);
        const int rank = i == 'N' ? 0 : i == '-' ? 1 : is_rare((uint32_t)i) ? 2 : i == 'S' ? 3 : 4;
        cls_s[i] = (uint8_t)((ai >= 0 ? ai : (i == 'N' ? 9 : 8)) | (rank << 4));
        const int au = acgt_index((uint32_t)((i >= 'a' && i <= 'z') ? i - 32 : i));
        up4_s[i] = (uint8_t)(au >= 0 ? au : 0x80);
    }
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        bfl_s[t] = base_freq_log(t == 0 ? 'A' : t == 1 ? 'C' : t == 2 ? 'G' : t == 3 ? 'T' : t == 9 ? 'N' : 0u);
    }
    for (int i = threadIdx.x; i < 100; i += blockDim.x) {
        const double qs = d.qscore[i], w_miss = qs / 3.0;
        qs_s[i] = double2{w_miss, (1.0 - qs) - w_miss};
    }
    for (int i = threadIdx.x; i < 64; i += blockDim.x) logtab_s[i] = euka_log_table[i];
    if (DMG_LDS)
        for (uint32_t i = threadIdx.x; i < d.n5 * d.n3 * 16u; i += blockDim.x) {
            const uint32_t pair = i >> 4, rb = (i >> 2) & 3u, g = i & 3u;
            const double *m = d.dmg_pair + pair * 20u + 4u * rb, *rs = d.dmg_pair + pair * 20u + 16u;
            dmg_s[2u * i] = double2{m[g], rs[g]};
            dmg_s[2u * i + 1u] = double2{(0.95238 * m[g ^ 2u] + 0.02381 * (m[g ^ 1u] + m[g ^ 3u])) - m[g],
                                         (0.95238 * rs[g ^ 2u] + 0.02381 * (rs[g ^ 1u] + rs[g ^ 3u])) - rs[g]};
        }
    {
        uint32_t *z = reinterpret_cast<uint32_t *>(acc_s);
        for (uint32_t i = threadIdx.x; i < sizeof(acc_s) / 4; i += blockDim.x) z[i] = 0u;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    EkAcc &acc = acc_s[wave];
    const bool shift_in_lds = d.ltp <= EK_ACC_LTP;
    const uint32_t sub = lane & 15u, gshift = lane & 48u, grp = (uint32_t)lane >> 4;
    const uint32_t below = (1u << sub) - 1u;
    auto row_bits = [&](bool p) { return (uint32_t)(__builtin_amdgcn_ballot_w64(p) >> gshift) & 0xFFFFu; };
    auto wave_max4 = [&](uint32_t v) { // max over the four rows of a value that is uniform within each row
        return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
                   max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
    };
    // the column stage's rows: BG lanes per read.  What a step costs beside its columns -- a read's set-up, the closing logarithm,
    // the row sums, the base shifts: some 210 vector instructions, four 16-column steps' worth -- is paid once per step whatever
    // the number of reads in it: with eight lanes per read a step takes eight reads (and twice the steps of half the columns)
    constexpr uint32_t BG = EK_BG, BROWS = 64u / BG;
    const uint32_t subB = lane & (BG - 1u), gshiftB = (uint32_t)lane & (64u - BG), grpB = (uint32_t)lane / BG;
    const uint32_t belowB = (1u << subB) - 1u;
    auto row_bitsB = [&](bool p) { return (uint32_t)(__builtin_amdgcn_ballot_w64(p) >> gshiftB) & ((1u << BG) - 1u); };
    auto wave_maxB = [&](uint32_t v) { // max over the rows of a value that is uniform within each row
        uint32_t m = 0;
#pragma unroll
        for (uint32_t k = 0; k < BROWS; ++k) m = max(m, (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(k * BG)));
        return m;
    };
    auto row_sumB = [&](double v) { return BG == 16u ? row_sum16(v) : row_sum8(v); };
    // this workgroup's replica of the per-clade accumulators
    const uint32_t rep = blockIdx.x % EUKA_REPLICAS;
    int32_t *const clade_count = o.clade_count + (size_t)rep * d.n_clades;
    uint32_t *const like_n = o.like_n + (size_t)rep * d.n_clades;
    double *const like_logsum = o.like_logsum + (size_t)rep * d.n_clades;
    uint32_t *const baseshift = o.baseshift + (size_t)rep * d.n_clades * 2 * (d.ltp > 0 ? d.ltp : 1) * 16;
    double *const bin_cov = o.bin_cov + (size_t)rep * o.n_bins;

    // the wave's accumulators go to the global tables (replica of this workgroup): a few hundred atomics per clade and wave
    // instead of ~15 per read
    int32_t cur = -1; // the clade the accumulators hold (wave uniform)
    auto flush = [&]() {
        if (cur < 0) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (shift_in_lds) {
            for (int i = lane; i < 2 * d.ltp * 16; i += 64) {
                const uint32_t v = acc.shift[i];
                if (v) {
                    atomicAdd(&baseshift[(size_t)cur * 2 * d.ltp * 16 + i], v);
                    acc.shift[i] = 0u;
                }
            }
        }
        const uint32_t b0 = d.bin_off[cur], nb = min(d.bin_off[cur + 1] - b0, (uint32_t)EK_ACC_BINS);
        if ((uint32_t)lane < nb) {
            const double v = acc.cov[lane];
            if (v != 0.0) {
                unsafeAtomicAdd(&bin_cov[b0 + lane], v);
                acc.cov[lane] = 0.0;
            }
        }
        if (lane == 0) {
            if (acc.count) atomicAdd(&clade_count[cur], acc.count);
            if (acc.n_like) {
                atomicAdd(&like_n[cur], acc.n_like);
                unsafeAtomicAdd(&like_logsum[cur], acc.logsum);
            }
            acc.count = 0;
            acc.n_like = 0u;
            acc.logsum = 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // Contiguous reads per wave, taken in blocks of 64 through four stages -- what is per READ (its offsets and lengths, its
    // clade, the closing log-sum-exp and the outputs) runs with a lane per read, 64 reads per instruction; only the column
    // loop and what needs a read's columns or mappings runs with a row of 16 lanes per read, 4 reads per instruction.  (In
    // the first form of this kernel every stage ran per group of four reads: half of its vector instructions were the
    // per-read work done sixteen lanes wide, the libm calls of the epilogue first.)
    EkBlk &blk = blk_s[wave];
    const uint32_t n_waves = gridDim.x * EK_WAVES;
    const uint32_t per_wave = ((b.n_reads + n_waves - 1) / n_waves + EK_READS_PER_WAVE - 1) / EK_READS_PER_WAVE * EK_READS_PER_WAVE;
    const uint32_t w_begin = (blockIdx.x * EK_WAVES + wave) * per_wave, w_end = min(b.n_reads, w_begin + per_wave);
    for (uint32_t bstart = w_begin; bstart < w_end; bstart += 64u) {
        const uint32_t nb = min(64u, w_end - bstart);
        // ---- A: a lane per read: what the row of the read will need, and its clade
        {
            const uint32_t r = bstart + min((uint32_t)lane, nb - 1u);
            const uint32_t q0 = b.read_qual_off[r], m0 = b.read_map_off[r];
            // clade of the first mapping's node: last (clade, bin) containing it, else clade 0 (readGAM_Euka.h:99-140)
            int32_t c_n = 0;
            {
                const uint32_t node = b.map_node[m0];
                if (d.n_node_clade) {
                    c_n = d.node_clade[min(node, d.n_node_clade - 1u)];
                } else {
                    uint32_t lo = 0, hi = d.n_bp; // first breakpoint > node
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (d.bp[mid] <= node) lo = mid + 1;
                        else hi = mid;
                    }
                    if (lo > 0) {
                        const int32_t c = d.bp_clade[lo - 1];
                        if (c >= 0) c_n = c;
                    }
                }
            }
            blk.col0[lane] = b.read_col_off[r];
            blk.ga[lane] = (uint32_t)b.read_gseq_len[r] | ((uint32_t)b.read_rseq_len[r] << 16);
            blk.lq[lane] = (uint32_t)b.read_seq_len[r] | (min(b.read_qual_off[r + 1] - q0, 0xFFFFu) << 16); // (columns are below 2^16)
            blk.q0[lane] = q0;
            blk.m0[lane] = m0;
            blk.flags[lane] = (b.read_rev[r] != 0 ? 1u : 0u) | (min(b.read_map_off[r + 1] - m0, 0xFFFFu) << 16);
            {
                const uint32_t b0 = d.bin_off[c_n];
                blk.bins[lane] = (b0 & 0xFFFFFFu) | (min(d.bin_off[c_n + 1] - b0, 255u) << 24);
            }
            blk.mapq[lane] = b.read_mapq[r];
            {
                // the four rows of a step walk their columns together, to the longest of the four: the block's reads are taken in
                // the order of their lengths, so that four of about one length share the steps (in file order the longest of
                // four fragments of 75 +- 17 columns costs a step more than the average one -- a sixth of the column loop).  What
                // a read adds to its outputs and to the sums does not depend on the rows beside it.
                const uint32_t key = (uint32_t)lane < nb ? (((uint32_t)lane / EK_ORDER_SPAN) << 26) | ((uint32_t)b.read_gseq_len[r] << 6) | (uint32_t)lane : 0xFFFFFFFFu;
                uint32_t rank = 0;
#pragma unroll
                for (int j = 0; j < 64; ++j) rank += (uint32_t)__builtin_amdgcn_readlane((int)key, j) < key ? 1u : 0u;
                blk.order[rank] = (uint8_t)lane;
            }
            blk.cn[lane] = c_n;
            blk.pd[lane] = d.clade_dist[c_n];
            // the block's first read names the wave's clade; a read of another clade (a boundary block) adds to the global tables
            const int32_t c_first = __builtin_amdgcn_readfirstlane(c_n);
            if (c_first != cur) {
                flush();
                cur = c_first;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- B: a row of BG lanes per read, 64 / BG reads per step
        // (a group's first three bytes per lane are requested while the group in front is closed: see the column loop)
        uint32_t gc_next, rc_next;
        int q_next;
        auto first_bytes = [&](uint32_t g4n) {
            const uint32_t i = blk.order[min(g4n + grpB, nb - 1u)];
            const uint32_t col0 = blk.col0[i], q0 = blk.q0[i];
            const uint32_t g_last = max(blk.ga[i] & 0xFFFFu, 1u) - 1u, q_last = max(blk.lq[i] >> 16, 1u) - 1u;
            gc_next = b.graph_seq[col0 + min(subB, g_last)];
            rc_next = b.read_seq[col0 + min(subB, g_last)];
            q_next = (int)(int8_t)b.qual[q0 + min(subB, q_last)];
        };
        first_bytes(0);
        for (uint32_t g4 = 0; g4 < nb; g4 += BROWS) {
            const bool have = g4 + grpB < nb; // this row has a read
            const uint32_t i = blk.order[min(g4 + grpB, nb - 1u)];
            const uint32_t col0 = blk.col0[i], ga = blk.ga[i], lq = blk.lq[i], q0 = blk.q0[i];
            const uint32_t G = have ? (ga & 0xFFFFu) : 0u, A = ga >> 16, Lseq = lq & 0xFFFFu, QL = lq >> 16;
            const bool rev = (blk.flags[i] & 1u) != 0;
            const int32_t c_n = blk.cn[i];
            const double pair_dist = blk.pd[i];
            const bool in_acc = c_n == cur;

            // model 1 of a column is c1 + log(a1): the a1 of a lane's columns are multiplied up and the log is taken once per eight
            // of them (a1 is a probability of a base under the damage and error model, 1e-12 at the very least: eight fit a
            // double with room to spare; a1 = 0 -- a graph byte outside ACGT in a regular column -- gives log 0 = -inf either way)
            double lik = 0.0, lik2 = 0.0, prod = 1.0;
            uint32_t carry_n = 0, carry_sc = 0, step = 0, n_reg = 0, n_same = 0;
            bool bad = false;
            // the base-shift columns of the read's ends (Baseshift::baseshift_calc: the first / last lengthToProf columns): a lane's
            // first one is loaded here, behind the column loop it would be a load waited for on the spot
            // (two of them: lengthToProf 5 is ten columns, a lane more than a row of eight has)
            uint32_t bs_g[2] = {0u, 0u}, bs_r[2] = {0u, 0u};
            bool bs_ok[2] = {false, false};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int p = (int)subB + k * (int)BG;
                if (p < 2 * d.ltp) {
                    const int gp = p < d.ltp ? p : (int)G - 2 * d.ltp + p; // (lengths are below 2^16)
                    const int rp = p < d.ltp ? p : (int)A - 2 * d.ltp + p;
                    bs_ok[k] = have && gp >= 0 && rp >= 0 && gp < (int)G && rp < (int)A;
                    if (bs_ok[k]) {
                        bs_g[k] = b.graph_seq[col0 + (uint32_t)gp];
                        bs_r[k] = b.read_seq[col0 + (uint32_t)rp];
                    }
                }
            }
            const uint32_t maxG = wave_maxB(G);
            // a step's three bytes are loaded a step ahead (unconditional loads at clamped addresses, selected afterwards):
            // their latency lies behind the step in front instead of in front of their own
            const uint32_t g_last = max(ga & 0xFFFFu, 1u) - 1u, q_last = max(QL, 1u) - 1u;
            const uint32_t n_sign = rev ? 0xFFFFFFFFu : 1u, n_base = rev ? Lseq - 1u : 0u; // the damage position from the non-gap count
            for (uint32_t base = 0; base < maxG; base += BG) {
                const uint32_t m = base + subB;
                // lane masks are kept as what the compares write -- a pair of scalar registers -- and combined there: as a bool a
                // condition that is the AND of two compares goes through a register and a second compare before a ballot
                uint64_t act_m, in_a_m, in_q_m;
                asm("v_cmp_lt_u32 %0, %1, %2" : "=s"(act_m) : "v"(m), "v"(G));
                asm("v_cmp_lt_u32 %0, %1, %2" : "=s"(in_a_m) : "v"(m), "v"(A));
                asm("v_cmp_lt_u32 %0, %1, %2" : "=s"(in_q_m) : "v"(m), "v"(QL));
                const bool active = __builtin_amdgcn_inverse_ballot_w64(act_m);
                const uint32_t gc = gc_next; // (an inactive lane holds the row's last column: everything it feeds is masked)
                const uint32_t rc = __builtin_amdgcn_inverse_ballot_w64(act_m & in_a_m) ? rc_next : 0u;
                const int q_raw = q_next;
                {
                    const uint32_t m2 = m + BG;
                    gc_next = b.graph_seq[col0 + min(m2, g_last)];
                    rc_next = b.read_seq[col0 + min(m2, g_last)];
                    q_next = (int)(int8_t)b.qual[q0 + min(m2, q_last)];
                }
                const uint32_t gcl = cls_s[gc], rcl = cls_s[rc];
                uint64_t nongap_m;
                asm("v_cmp_ne_u32 %0, 45, %1" : "=s"(nongap_m) : "v"(rc)); // '-'
                nongap_m &= act_m;
                const uint32_t nongap = (uint32_t)(nongap_m >> gshiftB) & ((1u << BG) - 1u);
                const uint32_t n_before = carry_n + (uint32_t)__builtin_popcount(nongap & belowB);
                // forward: n_before; reverse strand: Lseq - 1 - n_before (unsigned wrap as in the reference)
                uint32_t n;
                asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(n) : "v"(n_before), "v"(n_sign), "v"(n_base));
                int q = __builtin_amdgcn_inverse_ballot_w64(in_q_m) ? q_raw : 0; // Q15 (m < QL is false for an empty string)
                q = q < 0 ? 0 : (q > 99 ? 99 : q);
                const double2 qe = qs_s[q];
                // the regular column (:283-400), evaluated for every lane: p = sum_o pre[o] * sum_b M[o][b] * w[b] with
                // w = w_miss except w[read base] = w_hit, i.e. per original base o: w_miss * rowsum[o] + (w_hit - w_miss) * M[o][rb],
                // and pre = 1 - dist at the graph base g, dist * 0.95238 at its transition partner g^2, dist * 0.02381 at the
                // other two (:312-318, Euka.cpp:453-468).
                const uint32_t gi = gcl & 15u, ri = rcl & 15u; // 0..3, or 8 / 9 outside ACGT
                const uint32_t nn = min(n, Lseq - 1u);
                // (24-bit multiplies: positions and table sizes are below 2^24, and the full 32-bit multiply is a quarter-rate instruction)
                uint32_t pair_ix; // (spelled out: the compiler forms a quarter-rate 64-bit multiply-add for the same expression)
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(pair_ix) : "v"(min(nn, d.n5 - 1u)), "s"(d.n3), "v"(min(Lseq - 1u - nn, d.n3 - 1u)));
                const double w_miss = qe.x;
                // (a read byte outside ACGT matches no column: dw = 0 there -- a case for the branch of the odd columns below)
                double dw = qe.y;
                const uint32_t o0 = gi & 3u;
                double p;
                if constexpr (DMG_LDS) {
                    // byte offset of the entry of (pair, rb, g): pair * 512 + rb * 128 + g * 32
                    const uint32_t a0 = ((pair_ix << 9) | ((rcl << 7) & 0x180u)) | ((gcl << 5) & 0x60u);
                    const uint8_t *tab = reinterpret_cast<const uint8_t *>(dmg_s);
                    const double2 e0 = *reinterpret_cast<const double2 *>(tab + a0), e1 = *reinterpret_cast<const double2 *>(tab + a0 + 16u);
                    p = w_miss * (e0.y + pair_dist * e1.y) + dw * (e0.x + pair_dist * e1.x);
                } else {
                    dw = ri < 4u ? dw : 0.0;
                    uint32_t e_ix;
                    asm("v_mul_u32_u24 %0, 20, %1" : "=v"(e_ix) : "v"(pair_ix));
                    const double *e = d.dmg_pair + e_ix; // pair table layout: euka_device.h
                    const double *mcol = e + 4u * (ri & 3u); // M[.][read base]
                    const uint32_t o1 = o0 ^ 1u, o2 = o0 ^ 2u, o3 = o0 ^ 3u;
                    const double d0 = w_miss * e[16 + o0] + dw * mcol[o0], d2 = w_miss * e[16 + o2] + dw * mcol[o2];
                    const double d1 = w_miss * e[16 + o1] + dw * mcol[o1], d3 = w_miss * e[16 + o3] + dw * mcol[o3];
                    p = (1.0 - pair_dist) * d0 + pair_dist * (0.95238 * d2 + 0.02381 * (d1 + d3));
                }
                // model 1 = c1 + log(a1), model 2 = l2.  A regular column of two ACGT bytes -- all but a few per thousand -- has
                // a1 = p, c1 = 0 and l2 one of two constants, which are counted instead of added; everything else (an inactive
                // lane included) starts from a1 = 1, c1 = l2 = 0 and the cases below pick what differs
                // regular: both bytes' ranks are 4 (no special case) and both are ACGT -- one compare of the two class bytes side by
                // side (class byte: index in the low nibble, 8 / 9 outside ACGT; rank in the high one)
                const uint32_t cls2 = gcl | (rcl << 8);
                uint64_t regular_m, same_m, past_m;
                asm("v_cmp_eq_u32 %0, %1, %2" : "=s"(regular_m) : "v"(cls2 & 0xFCFCu), "v"(0x4040u));
                regular_m &= act_m;
                asm("v_cmp_eq_u32 %0, %1, %2" : "=s"(same_m) : "v"(gc), "v"(rc));
                asm("v_cmp_ge_u32 %0, %1, %2" : "=s"(past_m) : "v"(n), "v"(Lseq));
                const bool regular = __builtin_amdgcn_inverse_ballot_w64(regular_m);
                if (__builtin_expect((past_m & act_m) != 0, 0)) { // subDeamDiNuc[Lseq][n] out of range in the reference
                    if (active && n >= Lseq && min(gcl >> 4, rcl >> 4) == 4u) bad = true;
                }
                double a1 = regular ? p : 1.0, c1 = 0.0, l2 = 0.0;
                { // the counters take the masks as carries: an add each
                    uint64_t co;
                    asm("v_addc_co_u32 %0, %1, %0, 0, %2" : "+v"(n_same), "=s"(co) : "s"(regular_m & same_m));
                    asm("v_addc_co_u32 %0, %1, %0, 0, %2" : "+v"(n_reg), "=s"(co) : "s"(regular_m));
                }
                // N / gap / rare / softclip columns are a few per thousand: the whole wave skips their selects unless it has one
                if ((act_m & ~regular_m) != 0) {
                    const uint32_t kind = min(gcl >> 4, rcl >> 4); // 0 N, 1 gap, 2 rare, 3 softclip, 4 regular
                    const uint32_t scb = row_bitsB(active && kind == 3u);
                    const uint32_t sc_index = carry_sc + (uint32_t)__builtin_popcount(scb & belowB) + 1u; // ++softclip_count
                    carry_sc += (uint32_t)__builtin_popcount(scb);
                    const double bfl = bfl_s[rcl & 15u];
                    if (active && kind == 4u && !regular) {
                        // a graph byte outside ACGT has no t_T_ratio entry: p = 0; a read byte outside ACGT matches no column of the
                        // damage matrix: the column's p with dw = 0 (the same expression: d = w_miss * rowsum + 0 * M exactly)
                        a1 = 0.0;
                        if (gi < 4u) {
                            if constexpr (DMG_LDS) {
                                const uint32_t ix = ((pair_ix << 4) | o0) * 2u; // (any read base: only the row sums are used)
                                a1 = w_miss * (dmg_s[ix].y + pair_dist * dmg_s[ix + 1u].y);
                            } else {
                                a1 = p; // (dw was selected above)
                            }
                        }
                        l2 = gc == rc ? -0.2948543988682102 /* log(1-0.25536) */ : -1.3650809647206932 /* log(0.25536) */;
                    }
                    if (active && kind == 3u) { // :263-280
                        a1 = (sc_index % 3u == 0u) ? 1.0 - d.qscore[q] : w_miss; // w_hit
                        l2 = -1.3862943611198906; // log(0.25)
                    }
                    if (active && kind == 2u) { // :252-257
                        a1 = (1.0 - pair_dist) * 0.001;
                        l2 = -6.907755278982137; // log(0.001)
                    }
                    if (active && kind == 1u) { // :244-249
                        c1 = -6.214608098422191;  // log(0.002)
                        l2 = -1.6094379124341003; // log(0.2)
                    }
                    if (active && kind == 0u) c1 = l2 = bfl; // :236-241
                    lik += c1;
                    lik2 += l2;
                }
                prod *= a1;
                if ((++step & 7u) == 0u) {
                    lik += log_tab(prod, true, logtab_s); // log(1) == 0 exactly
                    prod = 1.0;
                }
                carry_n += (uint32_t)__builtin_popcount(nongap);
            }
            if (g4 + BROWS < nb) first_bytes(g4 + BROWS);
            if (step & 7u) lik += log_tab(prod, true, logtab_s);
            lik2 += (double)n_same * -0.2948543988682102 /* log(1-0.25536) */ + (double)(n_reg - n_same) * -1.3650809647206932 /* log(0.25536) */;
            const double in = row_sumB(lik), out = row_sumB(lik2);
            bad = row_bitsB(bad) != 0u;
            if (subB == 0 && have) {
                blk.in[i] = in;
                blk.out[i] = out;
                if (bad) blk.flags[i] |= 2u;
            }
            const bool live = have && !bad;
            // Baseshift::baseshift_calc: first / last lengthToProf columns (baseshift.cpp:57-88)
            for (int p = (int)subB, k = 0; p < 2 * d.ltp; p += (int)BG, ++k) {
                uint32_t gb = k == 0 ? bs_g[0] : bs_g[1], rb = k == 0 ? bs_r[0] : bs_r[1];
                bool ok = k == 0 ? bs_ok[0] : bs_ok[1];
                if (k >= 2) { // (a long lengthToProf: the further columns are loaded here)
                    const int gp = p < d.ltp ? p : (int)G - 2 * d.ltp + p;
                    const int rp = p < d.ltp ? p : (int)A - 2 * d.ltp + p;
                    ok = have && gp >= 0 && rp >= 0 && gp < (int)G && rp < (int)A;
                    if (ok) {
                        gb = b.graph_seq[col0 + (uint32_t)gp];
                        rb = b.read_seq[col0 + (uint32_t)rp];
                    }
                }
                if (live && ok) {
                    const uint32_t g4i = up4_s[gb], r4i = up4_s[rb];
                    if ((g4i | r4i) < 4u) {
                        if (in_acc && shift_in_lds) atomicAdd(&acc.shift[p * 16 + g4i * 4 + r4i], 1u);
                        else atomicAdd(&baseshift[(__umul24((uint32_t)c_n, 2u * (uint32_t)d.ltp) + (uint32_t)p) * 16u + g4i * 4u + r4i], 1u); // (32 bits: at most 2^20 clades, vgan_euka_create)
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- C: a lane per read: clade_like / clade_not_like (:485-492), the outputs, the clade's sums
        bool any_pass;
        uint32_t n_pass_blk = 0u;
        {
            // (lane k takes the k-th read in the order of the lengths: the reads that pass come out in that order, and stage D's
            // rows, too, hold four reads of about one length -- and of about as many mappings)
            const bool mine = (uint32_t)lane < nb;
            const uint32_t li = blk.order[min((uint32_t)lane, nb - 1u)];
            const uint32_t r = bstart + li;
            const uint32_t fl = blk.flags[li];
            const bool bad = mine && (fl & 2u) != 0;
            const bool live = mine && !bad;
            const double in = blk.in[li], out = blk.out[li];
            const int32_t mapq = blk.mapq[li], c_n = blk.cn[li];
            const bool in_acc = c_n == cur;
            if (bad) {
                o.clade[r] = -1;
                o.in_lik[r] = o.out_lik[r] = o.like[r] = o.not_like[r] = 0.0;
                o.pass[r] = 0;
            }
            const uint64_t bad_m = __builtin_amdgcn_ballot_w64(bad);
            if (bad_m && lane == 0) atomicAdd(o.n_bad, (unsigned long long)__builtin_popcountll(bad_m));
            // 1 - 10^(-mapq/10): table for 0..255; beyond it the subtrahend is below 1 ulp of 1 (negative mapq cannot be encoded
            // by a well-formed GAM, it is evaluated with exp for completeness)
            double map_q = mapq >= 256 ? 1.0 : d.mapq_ok[max(mapq, 0)];
            if (__builtin_expect(mapq < 0, 0)) map_q = ek_map_q_cold(mapq);
            const EkLike cl = ek_clade_like(in, out, map_q);
            const double like = cl.like;
            const bool pass = live && (in - out > 1.0) && ((uint32_t)mapq > d.min_mapq); // :504-510 (unsigned compare)
            if (live) {
                o.clade[r] = c_n;
                o.in_lik[r] = in;
                o.out_lik[r] = out;
                o.like[r] = like;
                o.not_like[r] = 1.0 - like;
                o.pass[r] = pass ? 1 : 0;
            }
            if (pass) blk.flags[li] = fl | 4u;
            const uint64_t pass_m = __builtin_amdgcn_ballot_w64(pass);
            if (pass) blk.passed[__builtin_popcountll(pass_m & ((1ull << lane) - 1ull))] = (uint8_t)li;
            n_pass_blk = (uint32_t)__builtin_popcountll(pass_m);
            // the abundance MCMC only ever uses sum_k log(frac * clade_like[k]) per clade (MCMC.cpp:1175-1215, (1/334) == 0):
            // keep the count and the sum of logs; a read with like == 0 (mapq 0, or exp underflow) makes the sum -inf as there
            const double ll = cl.log_like;
            const uint64_t acc_m = __builtin_amdgcn_ballot_w64(live && in_acc);
            if (acc_m) { // the reads of the wave's clade: one sum, one count
                const double s = wave_sum(live && in_acc ? ll : 0.0);
                const uint32_t n_pass = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(pass && in_acc));
                if (lane == 0) {
                    acc.count += (int32_t)n_pass;
                    acc.n_like += (uint32_t)__builtin_popcountll(acc_m);
                    acc.logsum += s;
                }
            }
            if (live && !in_acc) {
                if (pass) atomicAdd(&clade_count[c_n], 1);
                atomicAdd(&like_n[c_n], 1u);
                unsafeAtomicAdd(&like_logsum[c_n], ll);
            }
            any_pass = __builtin_amdgcn_ballot_w64(pass) != 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- D: bin coverage: every mapping's node adds 1/#mappings to each bin of the clade holding it (:520-546).  A row per
        // read again: lane j of a row owns the clade's bin jb + j and counts the read's mappings that fall into it (16 nodes per
        // step, handed round the row), then adds count / #mappings once.
        // A group's loads -- its lane's bin bounds, the first 32 of the row's mappings -- are requested while the group in front is
        // counted (the clade's bin range was looked up in stage A): behind each other they were three round trips per group.
        if (any_pass) {
            struct EkCovIn {
                int32_t lo, hi, n0, n1;
            };
            auto cov_request = [&](uint32_t g4, EkCovIn &q) {
                const bool pass = g4 + grp < n_pass_blk; // (the rows take the reads that passed, four at a time)
                const uint32_t i = blk.passed[min(g4 + grp, n_pass_blk - 1u)];
                const uint32_t fl = blk.flags[i], bins = blk.bins[i];
                const uint32_t b0 = bins & 0xFFFFFFu, nbin = pass ? bins >> 24 : 0u, m0 = blk.m0[i], nm = pass ? fl >> 16 : 0u;
                q.lo = sub < nbin ? d.bin_lo[b0 + sub] : 1;
                q.hi = sub < nbin ? d.bin_hi[b0 + sub] : 0;
                q.n0 = sub < nm ? (int32_t)b.map_node[m0 + sub] : -1; // no bin holds -1
                q.n1 = sub + EK_GROUP < nm ? (int32_t)b.map_node[m0 + sub + EK_GROUP] : -1;
            };
            EkCovIn cin, cnext;
            cov_request(0u, cin);
            for (uint32_t g4 = 0; g4 < n_pass_blk; g4 += EK_READS_PER_WAVE) {
                if (g4 + EK_READS_PER_WAVE < n_pass_blk) cov_request(g4 + EK_READS_PER_WAVE, cnext);
                const EkCovIn here = cin;
                cin = cnext;
                const bool pass = g4 + grp < n_pass_blk;
                const uint32_t i = blk.passed[min(g4 + grp, n_pass_blk - 1u)];
                const uint32_t fl = blk.flags[i], bins = blk.bins[i];
                const int32_t c_n = blk.cn[i];
                const bool in_acc = c_n == cur;
                const uint32_t m0 = blk.m0[i], nm_all = fl >> 16;
                const uint32_t b0 = bins & 0xFFFFFFu, nbin = pass ? bins >> 24 : 0u;
                const uint32_t nm = pass ? nm_all : 0u;
                const double inv = 1.0 / (double)nm_all;
                const uint32_t max_nb = wave_max4(nbin), max_nm = wave_max4(nm);
                // A read's nodes lie close together -- a walk of a few dozen mappings through nodes numbered along the graph --:
                // as bits of one 64-bit word, counted from the smallest of them, a bin's count is the population of the bits
                // between its bounds, a dozen instructions where the comparison of every node with every bin is three per node.
                // The word holds every node exactly when its population is the number of mappings: a node further than 64 ids from
                // the smallest, one met twice, or more than the 32 mappings the row holds in registers leave it short, and the
                // step (all four rows of it) counts by comparison instead.
                uint64_t node_bits;
                uint32_t node_min;
                bool by_bits;
                {
                    node_min = row_min16(min((uint32_t)here.n0, (uint32_t)here.n1)); // (no mapping: -1, the largest)
                    const uint32_t d0 = (uint32_t)here.n0 - node_min, d1 = (uint32_t)here.n1 - node_min;
                    node_bits = row_or16((here.n0 >= 0 && d0 < 64u ? 1ull << d0 : 0ull) | (here.n1 >= 0 && d1 < 64u ? 1ull << d1 : 0ull));
                    by_bits = __builtin_amdgcn_ballot_w64((uint32_t)__builtin_popcountll(node_bits) != nm) == 0;
                }
                for (uint32_t jb = 0; jb < max_nb; jb += EK_GROUP) {
                    const bool mine = jb + sub < nbin;
                    // node in [lo, hi] as ONE unsigned compare: node - lo <= hi - lo; a lane without a bin (and a bin whose bounds
                    // are the wrong way round) takes lo = INT_MIN, width 0: no node id -- they are below 2^31, the filler is -1 --
                    // passes
                    int32_t b_lo = here.lo, b_hi = here.hi;
                    if (jb) { // (a clade with more than 16 bins)
                        b_lo = mine ? d.bin_lo[b0 + jb + sub] : 1;
                        b_hi = mine ? d.bin_hi[b0 + jb + sub] : 0;
                    }
                    b_lo = max(b_lo, 0); // (node ids are >= 0)
                    const uint32_t my_lo = b_hi >= b_lo ? (uint32_t)b_lo : 0x80000000u, my_w = b_hi >= b_lo ? (uint32_t)(b_hi - b_lo) : 0u;
                    uint32_t cnt = 0;
                    if (by_bits) {
                        const int64_t a = (int64_t)my_lo - (int64_t)node_min, z = a + (int64_t)my_w; // the bin, in bit positions
                        if (a <= 63 && z >= 0) {
                            const uint32_t lo_c = (uint32_t)max(a, (int64_t)0), hi_c = (uint32_t)min(z, (int64_t)63);
                            cnt = (uint32_t)__builtin_popcountll(node_bits & (~0ull << lo_c) & (~0ull >> (63u - hi_c)));
                        }
                    } else
                    for (uint32_t mb = 0; mb < max_nm; mb += EK_GROUP) {
                        int32_t node = mb == 0u ? here.n0 : here.n1;
                        if (mb >= 2u * EK_GROUP) node = mb + sub < nm ? (int32_t)b.map_node[m0 + mb + sub] : -1; // (beyond 32 mappings)
                        // lane k of the row hands its node to the whole row (DPP row_newbcast:k -- an operand of the subtraction, no
                        // trip through the LDS crossbar as a lane exchange is); a lane without a mapping holds -1, which no bin holds
                        ek_count_row<0>(node, my_lo, my_w, cnt);
                    }
                    if (cnt) {
                        if (in_acc && jb + sub < (uint32_t)EK_ACC_BINS) unsafeAtomicAdd(&acc.cov[jb + sub], (double)cnt * inv);
                        else unsafeAtomicAdd(&bin_cov[b0 + jb + sub], (double)cnt * inv);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // The end of the workgroup: its four waves took consecutive ranges of the sorted reads, so they mostly end on one
    // clade -- their accumulators are added up in LDS and leave the chip once.
    __shared__ int32_t cur_s[EK_WAVES];
    if (lane == 0) cur_s[wave] = cur;
    __syncthreads();
    const int32_t cur0 = cur_s[0];
    if (wave > 0) {
        if (cur == cur0 && cur >= 0) {
            EkAcc &a0 = acc_s[0];
            if (shift_in_lds)
                for (int i = lane; i < 2 * d.ltp * 16; i += 64) {
                    const uint32_t v = acc.shift[i];
                    if (v) atomicAdd(&a0.shift[i], v);
                }
            if (lane < EK_ACC_BINS) {
                const double v = acc.cov[lane];
                if (v != 0.0) unsafeAtomicAdd(&a0.cov[lane], v);
            }
            if (lane == 0) {
                if (acc.count) atomicAdd(&a0.count, acc.count);
                if (acc.n_like) {
                    atomicAdd(&a0.n_like, acc.n_like);
                    unsafeAtomicAdd(&a0.logsum, acc.logsum);
                }
            }
        } else {
            flush();
        }
    }
    __syncthreads();
    if (wave == 0) flush();
}

// replica r > 0 added onto replica 0, replicas cleared: fixed order, so the sums do not depend on scheduling
__global__ void euka_reduce_kernel(EukaOutDev o, uint32_t n_count, uint32_t n_shift, uint32_t n_cov) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_count) {
        int32_t s = o.clade_count[i];
        for (uint32_t r = 1; r < EUKA_REPLICAS; ++r) {
            s += o.clade_count[(size_t)r * n_count + i];
            o.clade_count[(size_t)r * n_count + i] = 0;
        }
        o.clade_count[i] = s;
        uint32_t n = o.like_n[i];
        double ls = o.like_logsum[i];
        for (uint32_t r = 1; r < EUKA_REPLICAS; ++r) {
            n += o.like_n[(size_t)r * n_count + i];
            ls += o.like_logsum[(size_t)r * n_count + i];
            o.like_n[(size_t)r * n_count + i] = 0;
            o.like_logsum[(size_t)r * n_count + i] = 0.0;
        }
        o.like_n[i] = n;
        o.like_logsum[i] = ls;
    }
    if (i < n_shift) {
        uint32_t s = o.baseshift[i];
        for (uint32_t r = 1; r < EUKA_REPLICAS; ++r) {
            s += o.baseshift[(size_t)r * n_shift + i];
            o.baseshift[(size_t)r * n_shift + i] = 0;
        }
        o.baseshift[i] = s;
    }
    if (i < n_cov) {
        double s = o.bin_cov[i];
        for (uint32_t r = 1; r < EUKA_REPLICAS; ++r) {
            s += o.bin_cov[(size_t)r * n_cov + i];
            o.bin_cov[(size_t)r * n_cov + i] = 0.0;
        }
        o.bin_cov[i] = s;
    }
}

// the accumulators' clear: one launch over one allocation, 16 bytes per lane and step (the runtime's fill, one launch per array, was
// 27 us of a 0.33 ms step for 2.3 MB)
__global__ __launch_bounds__(256) void euka_clear_kernel(uint4 *p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = uint4{0u, 0u, 0u, 0u};
}

void launch_euka_clear(void *p, size_t bytes, hipStream_t st) { // (bytes: a multiple of 16, p aligned as hipMalloc returns it)
    const size_t n16 = bytes / 16;
    if (!n16) return;
    hipLaunchKernelGGL(euka_clear_kernel, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 1024)), dim3(256), 0, st, reinterpret_cast<uint4 *>(p), n16);
}

void launch_euka_reduce(const EukaOutDev &o, uint32_t n_clades, int32_t ltp, hipStream_t st) {
    const uint32_t n_shift = n_clades * 2 * (ltp > 0 ? ltp : 1) * 16;
    const uint32_t n = n_shift > o.n_bins ? n_shift : o.n_bins;
    hipLaunchKernelGGL(euka_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, o, n_clades, n_shift, o.n_bins);
}

void launch_euka_reads(const EukaDev &d, const EukaBatchDev &b, const EukaOutDev &o, hipStream_t st) {
    if (b.n_reads == 0) return;
    // each wave a contiguous range of the (sorted) reads; a workgroup's waves flush together at the end, so the number of
    // workgroups (not waves) sets the flush traffic: 4096 of them even out the CUs (1024: 0.96 ms, 4096: 0.89 ms per 1M reads)
    uint32_t blocks = (b.n_reads + EK_READS_PER_BLOCK - 1) / EK_READS_PER_BLOCK;
    uint32_t cap = 256u * 16u;
    if (const char *e = getenv("VGAN_EUKA_BLOCKS")) cap = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : cap; // developer aid
    blocks = blocks < cap ? blocks : cap;
    if (d.n5 * d.n3 <= EK_DMG_LDS_PAIRS)
        hipLaunchKernelGGL(euka_read_kernel<true>, dim3(blocks), dim3(EK_WAVES * 64), (size_t)d.n5 * d.n3 * 32u * sizeof(double2), st, d, b, o);
    else
        hipLaunchKernelGGL(euka_read_kernel<false>, dim3(blocks), dim3(EK_WAVES * 64), 0, st, d, b, o);
}

} // namespace vgan
#include "module_anchor.h"
const void *vgan::anchor_euka_kernels() { return (const void *)&vgan::euka_clear_kernel; }
