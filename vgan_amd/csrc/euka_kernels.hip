// euka per-read kernel for gfx950: the body of readGAM3's per-alignment lambda (reference
// src/readGAM_Euka.h:67-577) with Baseshift::baseshift_calc (src/baseshift.cpp:57-88).
//
// One wave per read, one lane per alignment column.  The two quantities the reference carries serially along the
// read -- the damage position n (advanced on every non-gap read column, readGAM_Euka.h:457-461) and the softclip
// counter (:269) -- are prefix popcounts of wave ballots, so the columns are independent.  Model 1 per column is
// pre[4] (divergence) x the 4x4 damage matrix selected from the 5'/3' tables, marginalised over the sequencing
// error: log(sum_b post[b] * w[b]) -- one log instead of the reference's four logs folded with oplusInitnatl
// (identical value; a fold whose running value is exactly 0 cannot occur since every weight is < 1).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "euka_device.h"

namespace vgan {

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

__global__ __launch_bounds__(EK_WAVES * 64, 4) void euka_read_kernel(EukaDev d, EukaBatchDev b, EukaOutDev o) {
    __shared__ double qs_s[100];
    for (int i = threadIdx.x; i < 100; i += blockDim.x) qs_s[i] = d.qscore[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // this workgroup's replica of the per-clade accumulators
    const uint32_t rep = blockIdx.x % EUKA_REPLICAS;
    int32_t *const clade_count = o.clade_count + (size_t)rep * d.n_clades;
    uint32_t *const baseshift = o.baseshift + (size_t)rep * d.n_clades * 2 * (d.ltp > 0 ? d.ltp : 1) * 16;
    double *const bin_cov = o.bin_cov + (size_t)rep * o.n_bins;

    for (uint32_t r = blockIdx.x * EK_WAVES + wave; r < b.n_reads; r += gridDim.x * EK_WAVES) {
        const uint32_t col0 = b.read_col_off[r];
        const uint32_t G = b.read_gseq_len[r], A = b.read_rseq_len[r];
        const uint32_t q0 = b.read_qual_off[r], QL = b.read_qual_off[r + 1] - q0;
        const uint32_t Lseq = b.read_seq_len[r];
        const int32_t mapq = b.read_mapq[r];
        const bool rev = b.read_rev[r] != 0;
        const uint32_t m0 = b.read_map_off[r], m1 = b.read_map_off[r + 1];
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
        const double pair_dist = d.clade_dist[c_n];

        double lik = 0.0, lik2 = 0.0;
        uint32_t carry_n = 0, carry_sc = 0;
        bool bad = false;
        for (uint32_t base = 0; base < G; base += 64) {
            const uint32_t m = base + lane;
            const bool active = m < G;
            const uint32_t gc = active ? b.graph_seq[col0 + m] : 0u;
            const uint32_t rc = (active && m < A) ? b.read_seq[col0 + m] : 0u;
            const uint64_t nongap = __builtin_amdgcn_ballot_w64(active && rc != '-');
            const uint32_t n_before = carry_n + (uint32_t)__builtin_popcountll(nongap & lt_mask);
            const uint32_t n = rev ? (Lseq - 1u - n_before) : n_before; // unsigned wrap as in the reference
            const bool isN = gc == 'N' || rc == 'N';
            const bool isgap = gc == '-' || rc == '-';
            const bool israre = is_rare(gc) || is_rare(rc);
            const bool isS = gc == 'S' || rc == 'S';
            const bool sc_col = active && !isN && !isgap && !israre && isS;
            const uint64_t scb = __builtin_amdgcn_ballot_w64(sc_col);
            const uint32_t sc_index = carry_sc + (uint32_t)__builtin_popcountll(scb & lt_mask) + 1u; // ++softclip_count
            int q = m < QL ? (int)(int8_t)b.qual[q0 + m] : 0; // Q15
            q = q < 0 ? 0 : (q > 99 ? 99 : q);
            const double qs = qs_s[q];
            // model 1 = c1 + log(a1), model 2 = l2; the branches only pick a1 / c1 / l2 so that one log serves all of them
            double a1 = 1.0, c1 = 0.0, l2 = 0.0;
            if (active) {
                if (isN) { // :236-241
                    c1 = l2 = base_freq_log(rc);
                } else if (isgap) { // :244-249
                    c1 = -6.214608098422191;  // log(0.002)
                    l2 = -1.6094379124341003; // log(0.2)
                } else if (israre) { // :252-257
                    a1 = (1.0 - pair_dist) * 0.001;
                    l2 = -6.907755278982137; // log(0.001)
                } else if (isS) { // :263-280
                    a1 = (sc_index % 3u == 0u) ? 1.0 - qs : qs / 3.0;
                    l2 = -1.3862943611198906; // log(0.25)
                } else {
                    if (n >= Lseq) bad = true; // subDeamDiNuc[Lseq][n] out of range in the reference
                    const int gi = acgt_index(gc), ri = acgt_index(rc);
                    const uint32_t nn = min(n, Lseq - 1u);
                    const double *e = d.dmg_pair + 20u * (min(nn, d.n5 - 1u) * d.n3 + min(Lseq - 1u - nn, d.n3 - 1u));
                    // p = sum_o pre[o] * sum_b M[o][b] * w[b] (:337-340, :385-394), w = w_miss except w[read base] = w_hit
                    const double w_hit = 1.0 - qs, w_miss = qs / 3.0;
                    const double2 rs01 = *reinterpret_cast<const double2 *>(e + 16);
                    const double2 rs23 = *reinterpret_cast<const double2 *>(e + 18);
                    const double *col = e + 4 * max(ri, 0); // M[.][read base]
                    const double2 c01 = *reinterpret_cast<const double2 *>(col);
                    const double2 c23 = *reinterpret_cast<const double2 *>(col + 2);
                    const double dw = ri < 0 ? 0.0 : w_hit - w_miss; // a read base outside ACGT matches no column
                    const double dot[4] = {w_miss * rs01.x + dw * c01.x, w_miss * rs01.y + dw * c01.y,
                                           w_miss * rs23.x + dw * c23.x, w_miss * rs23.y + dw * c23.y};
                    double p = 0.0;
#pragma unroll
                    for (int bpo = 0; bpo < 4; ++bpo) {
                        // :312-318; a graph base outside ACGT has no t_T_ratio entry (0)
                        const double pre = gi < 0 ? 0.0 : (bpo == gi ? 1.0 - pair_dist : pair_dist * tT_ratio(gi, bpo));
                        p += pre * dot[bpo];
                    }
                    a1 = p;
                    l2 = gc == rc ? -0.2948543988682102 /* log(1-0.25536) */ : -1.3650809647206932 /* log(0.25536) */;
                }
            }
            const double l1 = c1 + log_pos(a1); // log_pos(1) == 0 exactly
            lik += l1;
            lik2 += l2;
            carry_n += (uint32_t)__builtin_popcountll(nongap);
            carry_sc += (uint32_t)__builtin_popcountll(scb);
        }
        const double in = wave_sum(lik), out = wave_sum(lik2);
        bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (bad) {
            if (lane == 0) {
                o.clade[r] = -1;
                o.in_lik[r] = o.out_lik[r] = o.like[r] = o.not_like[r] = 0.0;
                o.pass[r] = 0;
                atomicAdd(o.n_bad, 1ull);
            }
            continue;
        }
        // Baseshift::baseshift_calc: first / last lengthToProf columns (baseshift.cpp:57-88)
        if (lane < 2 * d.ltp) {
            const int p = lane;
            const int64_t gi = p < d.ltp ? p : (int64_t)G - 2 * d.ltp + p;
            const int64_t ri = p < d.ltp ? p : (int64_t)A - 2 * d.ltp + p;
            if (gi >= 0 && ri >= 0 && gi < (int64_t)G && ri < (int64_t)A) {
                uint32_t gb = b.graph_seq[col0 + gi], rb = b.read_seq[col0 + ri];
                gb = (gb >= 'a' && gb <= 'z') ? gb - 32u : gb;
                rb = (rb >= 'a' && rb <= 'z') ? rb - 32u : rb;
                const int g4 = acgt_index(gb), r4 = acgt_index(rb);
                if (g4 >= 0 && r4 >= 0) atomicAdd(&baseshift[((size_t)c_n * 2 * d.ltp + p) * 16 + g4 * 4 + r4], 1u);
            }
        }
        // clade_like / clade_not_like (:485-492)
        // 1 - 10^(-mapq/10): table for 0..255; beyond it the subtrahend is below 1 ulp of 1 (negative mapq cannot be encoded
        // by a well-formed GAM, it is evaluated with exp for completeness)
        const double map_q = mapq >= 256 ? 1.0 : (mapq >= 0 ? d.mapq_ok[mapq] : 1.0 - exp(-0.1 * mapq * 2.302585092994046));
        double lse;
        if (in == 0.0) lse = out; // oplusInitnatl: a running value of 0 means "empty"
        else lse = fmax(in, out) + log1p(exp(-fabs(in - out)));
        const double like = map_q * exp(in - lse);
        const bool pass = (in - out > 1.0) && ((uint32_t)mapq > d.min_mapq); // :504-510 (unsigned compare)
        if (lane == 0) {
            o.clade[r] = c_n;
            o.in_lik[r] = in;
            o.out_lik[r] = out;
            o.like[r] = like;
            o.not_like[r] = 1.0 - like;
            o.pass[r] = pass ? 1 : 0;
            if (pass) atomicAdd(&clade_count[c_n], 1);
        }
        if (pass) { // bin coverage: every mapping's node adds 1/#mappings to each bin of the clade holding it (:520-546)
            const uint32_t b0 = d.bin_off[c_n], b1 = d.bin_off[c_n + 1];
            const double inv = 1.0 / (double)(m1 - m0);
            for (uint32_t jb = b0; jb < b1; jb += 64) { // lane j holds bin jb + j's bounds; the loop below reads them by lane
                const uint32_t nb = min(64u, b1 - jb);
                const int32_t my_lo = (uint32_t)lane < nb ? d.bin_lo[jb + lane] : 1;
                const int32_t my_hi = (uint32_t)lane < nb ? d.bin_hi[jb + lane] : 0;
                for (uint32_t mb = m0; mb < m1; mb += 64) {
                    const uint32_t mi = mb + lane;
                    const int32_t node = mi < m1 ? (int32_t)b.map_node[mi] : -1;
                    for (uint32_t j = 0; j < nb; ++j) {
                        const int32_t lo = __builtin_amdgcn_readlane(my_lo, (int)j), hi = __builtin_amdgcn_readlane(my_hi, (int)j);
                        const uint64_t hit = __builtin_amdgcn_ballot_w64(mi < m1 && node >= lo && node <= hi);
                        if (hit && lane == 0) unsafeAtomicAdd(&bin_cov[jb + j], (double)__builtin_popcountll(hit) * inv);
                    }
                }
            }
        }
    }
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

void launch_euka_reduce(const EukaOutDev &o, uint32_t n_clades, int32_t ltp, hipStream_t st) {
    const uint32_t n_shift = n_clades * 2 * (ltp > 0 ? ltp : 1) * 16;
    const uint32_t n = n_shift > o.n_bins ? n_shift : o.n_bins;
    hipLaunchKernelGGL(euka_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, o, n_clades, n_shift, o.n_bins);
}

void launch_euka_reads(const EukaDev &d, const EukaBatchDev &b, const EukaOutDev &o, hipStream_t st) {
    if (b.n_reads == 0) return;
    const uint32_t blocks = (uint32_t)((b.n_reads + EK_WAVES - 1) / EK_WAVES);
    hipLaunchKernelGGL(euka_read_kernel, dim3(blocks < 256u * 8u ? blocks : 256u * 8u), dim3(EK_WAVES * 64), 0, st, d, b, o);
}

} // namespace vgan
