// HaploCart per-read likelihood kernels for gfx950 (MI355X, wave64).
//
// What the reference computes (src/update_likelihood.cpp:19-53, src/process_mapping.cpp:26-91,
// src/get_p_obs_base.cpp:3-69): for every read r, mapping ("segment") m on node n_m and path p
//     ll_r[p] += pathsgo[n_m][p] ? S_m : U_m
// where S_m (sum over the segment's valid columns of log((1-pcm)*bg + pcm*match*(1-eps))) and U_m (sum of
// log p_err(Q) over the read's |algnseq|-long quality window starting at the segment, process_mapping.cpp:4-24)
// do not depend on p.  final[p] = sum_r ll_r[p]                                   (src/HaploCart.cpp:420).
//
// Kernels:
//   hc_segment_kernel   S_m, U_m per segment: one wave per read, quality window prefix sums in LDS,
//                       wave shuffles for the scan / reductions.
//   hc_sweep_kernel     the per-path update.  Accumulates  acc[p] += D_m  for every path NOT supported by the
//                       node (D_m = S_m - U_m >= 0), so that final[p] = sum_m S_m - acc[p] has no cancellation.
//                       One wave owns a tile of mask words; a 64-bit mask word is moved into EXEC so one
//                       v_add_f64 updates the 64 paths of that word.  Used per segment (PER_READ mode, streams
//                       one mask row per segment) and per node (NODE_WEIGHTS mode, D = W[node], once).
//   hc_read_loglik_kernel  literal per-read x per-path vectors (debug / parity aid).
//   hc_finish_kernel    final[p] = Stot - acc[p].
//   hc_posterior_kernel log-sum-exp over path sets (src/get_posterior.cpp:78-127).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "hc_device.h"

namespace vgan {

// ---------------------------------------------------------------------------------------------- helpers
__device__ __forceinline__ bool is_acgt(uint32_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

__device__ __forceinline__ double bg_freq(uint32_t c) { // src/haplocart_functions.cpp:81-98
    return c == 'A' ? 0.27532 : c == 'C' ? 0.30044 : c == 'G' ? 0.16644 : c == 'T' ? 0.25780 : 0.25;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------- segments
// One wave per read.  LDS: tables (lq 256, qscore 100) + per-wave prefix sums of the quality window.
constexpr int SEG_WAVES = 4;
constexpr int SEG_MAXQ = 1024;

__global__ __launch_bounds__(SEG_WAVES * 64) void hc_segment_kernel(HcGraphDev g, HcBatchDev b, HcParamsDev prm,
                                                                     double *__restrict__ segS,
                                                                     double *__restrict__ segU,
                                                                     double *__restrict__ segD,
                                                                     double *__restrict__ nodeW,
                                                                     double *__restrict__ totals) {
    __shared__ double lq_s[256];
    __shared__ double qs_s[100];
    __shared__ double ps_s[SEG_WAVES][SEG_MAXQ + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 256; i += blockDim.x) lq_s[i] = g.lq[i];
    for (int i = tid; i < 100; i += blockDim.x) qs_s[i] = g.qscore[i];
    __syncthreads();
    const double lq0 = lq_s[0];
    double *ps = ps_s[wave];
    double sumS = 0.0, sumU = 0.0;

    for (uint32_t r = blockIdx.x * SEG_WAVES + wave; r < b.n_reads; r += gridDim.x * SEG_WAVES) {
        const uint32_t seg0 = b.read_seg_off[r], seg1 = b.read_seg_off[r + 1];
        const uint32_t col0 = b.read_col_off[r];
        const uint32_t q0 = b.read_qual_off[r];
        const uint32_t QL = b.read_qual_off[r + 1] - q0;
        const uint32_t A = b.read_algn_len[r];
        const double pinc = g.incmap[b.read_mapq[r]];
        const bool use_lds = QL <= SEG_MAXQ;
        // quality window prefix sums + first Q >= 90 (update_likelihood.cpp:40-44)
        uint32_t first90 = 0xFFFFFFFFu;
        double carry = 0.0;
        for (uint32_t base = 0; base < QL; base += 64) {
            const uint32_t j = base + lane;
            const uint32_t qb = j < QL ? b.qual[q0 + j] : 0u;
            if (j < QL && (int)(int8_t)qb >= 90) first90 = min(first90, j);
            if (use_lds) {
                const double v = j < QL ? lq_s[qb] : 0.0;
                const double s = wave_incl_scan(v, lane);
                if (j < QL) ps[j + 1] = carry + s;
                carry += __shfl(s, 63, 64);
            }
        }
        if (lane == 0) ps[0] = 0.0;
        first90 = wave_min_u32(first90);
        // LDS writes of this wave are read back by this wave only; wave64 executes in lockstep but the
        // compiler must not reorder the ds ops:
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        for (uint32_t sb = seg0; sb < seg1; sb += 64) {
            const uint32_t s = sb + lane;
            if (s < seg1) {
                const uint32_t node = b.seg_node[s];
                const uint32_t start = b.seg_start[s];
                const uint32_t len = b.seg_len[s];
                // U_m: window [start, start+A) of the quality string, zero padded (Q5)
                const uint32_t lo = min(start, QL), hi = min(start + A, QL);
                double U;
                if (use_lds) {
                    U = (ps[hi] - ps[lo]) + (double)(A - (hi - lo)) * lq0;
                } else {
                    U = 0.0;
                    for (uint32_t j = lo; j < hi; ++j) U += lq_s[b.qual[q0 + j]];
                    U += (double)(A - (hi - lo)) * lq0;
                }
                const bool use_bep = prm.use_bep || first90 < hi; // sticky within the read (:42)
                const HcNodeDev nd = g.node_tab[node];
                const double pcm = (1.0 - pinc) * nd.mappability; // process_mapping.cpp:41
                double S = 0.0;
                for (uint32_t j = 0; j < len; ++j) {
                    const uint32_t gc = b.graph_seq[col0 + start + j];
                    const uint32_t rc = j < A ? b.algnseq[col0 + j] : 0u; // Q4: read bases from the read start
                    if (!is_acgt(gc) || !is_acgt(rc)) continue;           // process_mapping.cpp:62-63
                    int q = (start + j) < QL ? (int)(int8_t)b.qual[q0 + start + j] : 0;
                    q = q < 0 ? 0 : (q > 99 ? 99 : q);
                    const double e = use_bep ? prm.bep : qs_s[q];
                    const double eps = gc == rc ? e : 1.0 - e;          // get_p_obs_base.cpp:3-27
                    const double pobs = nd.match * (1.0 - eps);         // get_p_obs_base.cpp:67 with tv = ts = 0
                    const double x = prm.consensus ? (1.0 - prm.bep) * pobs : (1.0 - pcm) * bg_freq(rc) + pcm * pobs;
                    S += log(x);
                }
                if (segS) segS[s] = S;
                if (segU) segU[s] = U;
                if (segD) segD[s] = S - U;
                if (nodeW) unsafeAtomicAdd(&nodeW[node], S - U);
                sumS += S;
                sumU += U;
            }
        }
    }
    // totals[0] += sum of S over all segments (the value every path would get if it supported every node)
    sumS = wave_sum(sumS);
    sumU = wave_sum(sumU);
    if (lane == 0 && totals) {
        unsafeAtomicAdd(&totals[0], sumS);
        unsafeAtomicAdd(&totals[1], sumU);
    }
}

// ---------------------------------------------------------------------------------------------- sweep
// acc[p] += D[i] for all p with umask[node_i][p] set.  Wave = (chunk of items, tile of SWEEP_TW mask words).
constexpr int SWEEP_TW = 16;

struct Words16 {
    uint64_t w[SWEEP_TW];
};

__device__ __forceinline__ void masked_add16(double (&acc)[SWEEP_TW], const Words16 &m, double d) {
    uint64_t save;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b64 exec, %[m0]\n\tv_add_f64 %[a0], %[a0], %[d]\n\t"
                 "s_mov_b64 exec, %[m1]\n\tv_add_f64 %[a1], %[a1], %[d]\n\t"
                 "s_mov_b64 exec, %[m2]\n\tv_add_f64 %[a2], %[a2], %[d]\n\t"
                 "s_mov_b64 exec, %[m3]\n\tv_add_f64 %[a3], %[a3], %[d]\n\t"
                 "s_mov_b64 exec, %[m4]\n\tv_add_f64 %[a4], %[a4], %[d]\n\t"
                 "s_mov_b64 exec, %[m5]\n\tv_add_f64 %[a5], %[a5], %[d]\n\t"
                 "s_mov_b64 exec, %[m6]\n\tv_add_f64 %[a6], %[a6], %[d]\n\t"
                 "s_mov_b64 exec, %[m7]\n\tv_add_f64 %[a7], %[a7], %[d]\n\t"
                 "s_mov_b64 exec, %[m8]\n\tv_add_f64 %[a8], %[a8], %[d]\n\t"
                 "s_mov_b64 exec, %[m9]\n\tv_add_f64 %[a9], %[a9], %[d]\n\t"
                 "s_mov_b64 exec, %[m10]\n\tv_add_f64 %[a10], %[a10], %[d]\n\t"
                 "s_mov_b64 exec, %[m11]\n\tv_add_f64 %[a11], %[a11], %[d]\n\t"
                 "s_mov_b64 exec, %[m12]\n\tv_add_f64 %[a12], %[a12], %[d]\n\t"
                 "s_mov_b64 exec, %[m13]\n\tv_add_f64 %[a13], %[a13], %[d]\n\t"
                 "s_mov_b64 exec, %[m14]\n\tv_add_f64 %[a14], %[a14], %[d]\n\t"
                 "s_mov_b64 exec, %[m15]\n\tv_add_f64 %[a15], %[a15], %[d]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(save), [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]),
                   [a4] "+v"(acc[4]), [a5] "+v"(acc[5]), [a6] "+v"(acc[6]), [a7] "+v"(acc[7]), [a8] "+v"(acc[8]),
                   [a9] "+v"(acc[9]), [a10] "+v"(acc[10]), [a11] "+v"(acc[11]), [a12] "+v"(acc[12]),
                   [a13] "+v"(acc[13]), [a14] "+v"(acc[14]), [a15] "+v"(acc[15])
                 : [d] "s"(d), [m0] "s"(m.w[0]), [m1] "s"(m.w[1]), [m2] "s"(m.w[2]), [m3] "s"(m.w[3]), [m4] "s"(m.w[4]),
                   [m5] "s"(m.w[5]), [m6] "s"(m.w[6]), [m7] "s"(m.w[7]), [m8] "s"(m.w[8]), [m9] "s"(m.w[9]),
                   [m10] "s"(m.w[10]), [m11] "s"(m.w[11]), [m12] "s"(m.w[12]), [m13] "s"(m.w[13]), [m14] "s"(m.w[14]),
                   [m15] "s"(m.w[15]));
}

__device__ __forceinline__ Words16 load_row_tile(const uint64_t *__restrict__ p) {
    // wave-uniform address -> scalar loads (s_load_dwordx16 x2)
    Words16 r;
#pragma unroll
    for (int k = 0; k < SWEEP_TW; ++k) r.w[k] = p[k];
    return r;
}

__device__ __forceinline__ uint64_t any_bits(const Words16 &m) {
    uint64_t o = 0;
#pragma unroll
    for (int k = 0; k < SWEEP_TW; ++k) o |= m.w[k];
    return o;
}

// items [0, n_items): item i has node = item_node ? item_node[i] : i, weight D[i].
// umask rows have stride row_words (multiple of SWEEP_TW, zero padded).
__global__ __launch_bounds__(256) void hc_sweep_kernel(const uint64_t *__restrict__ umask, uint32_t row_words,
                                                        const uint32_t *__restrict__ item_node,
                                                        const double *__restrict__ D, uint32_t n_items,
                                                        uint32_t items_per_chunk, uint32_t n_tiles, int skip_zero,
                                                        double *__restrict__ acc_out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wid = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t tile = wid % n_tiles;
    const uint32_t chunk = wid / n_tiles;
    const uint64_t i0 = (uint64_t)chunk * items_per_chunk;
    if (i0 >= n_items) return;
    const uint32_t i1 = (uint32_t)min((uint64_t)n_items, i0 + items_per_chunk);
    const uint64_t *__restrict__ base = umask + (size_t)tile * SWEEP_TW;
    double acc[SWEEP_TW];
#pragma unroll
    for (int k = 0; k < SWEEP_TW; ++k) acc[k] = 0.0;

    uint32_t i = (uint32_t)i0;
    uint32_t node = item_node ? item_node[i] : i;
    Words16 cur = load_row_tile(base + (size_t)node * row_words);
    for (; i < i1; ++i) {
        const double d = D[i];
        Words16 nxt = cur;
        if (i + 1 < i1) { // prefetch the next row while this one is applied
            const uint32_t nn = item_node ? item_node[i + 1] : i + 1;
            nxt = load_row_tile(base + (size_t)nn * row_words);
        }
        if (!skip_zero || (d != 0.0 && any_bits(cur) != 0)) masked_add16(acc, cur, d);
        cur = nxt;
    }
#pragma unroll
    for (int k = 0; k < SWEEP_TW; ++k) {
        if (acc[k] != 0.0) unsafeAtomicAdd(&acc_out[((size_t)tile * SWEEP_TW + k) * 64 + lane], acc[k]);
    }
}

// final[p] = Stot - acc[p]
__global__ void hc_finish_kernel(const double *__restrict__ totals, const double *__restrict__ acc_seg,
                                 const double *__restrict__ acc_node, uint32_t n_paths, double *__restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_paths) out[p] = totals[0] - (acc_seg[p] + acc_node[p]);
}

// ---------------------------------------------------------------------------------------------- per-read dump
// out[r*P + p] = sum_m (supported ? S_m : U_m): the vector Haplocart::update returns (debug / parity aid).
__global__ void hc_read_loglik_kernel(HcGraphDev g, HcBatchDev b, const double *__restrict__ segS,
                                      const double *__restrict__ segU, double *__restrict__ out) {
    const uint32_t r = blockIdx.y;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.n_paths) return;
    double ll = 0.0;
    for (uint32_t s = b.read_seg_off[r]; s < b.read_seg_off[r + 1]; ++s) {
        const uint64_t w = g.umask[(size_t)b.seg_node[s] * g.row_words + (p >> 6)];
        ll += ((w >> (p & 63)) & 1) ? segU[s] : segS[s];
    }
    out[(size_t)r * g.n_paths + p] = ll;
}

// ---------------------------------------------------------------------------------------------- posterior
// One block per set: conf[set] = exp(LSE(final[p] : p in set) - LSE(final[all])).
// libgab's oplusInitnatl treats a running value of exactly 0 as "empty" (SURVEY.md Q11): zeros in front of
// the first non-zero term are skipped, both here and in the reference's sequential fold.
__device__ double block_lse(const double *__restrict__ v, const uint64_t *__restrict__ set, uint32_t n, double *sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    // first non-zero member
    uint32_t first = 0xFFFFFFFFu;
    for (uint32_t p = tid; p < n; p += blockDim.x) {
        const bool in = !set || ((set[p >> 6] >> (p & 63)) & 1);
        if (in && v[p] != 0.0) first = min(first, p);
    }
    first = wave_min_u32(first);
    __shared__ uint32_t shu[16];
    if (lane == 0) shu[wave] = first;
    __syncthreads();
    first = 0xFFFFFFFFu;
    for (int w = 0; w < nw; ++w) first = min(first, shu[w]);
    __syncthreads();
    if (first == 0xFFFFFFFFu) return 0.0; // all members are exactly 0 (or the set is empty)
    double mx = -INFINITY;
    for (uint32_t p = tid; p < n; p += blockDim.x) {
        const bool in = !set || ((set[p >> 6] >> (p & 63)) & 1);
        if (in && p >= first) mx = fmax(mx, v[p]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) sh[wave] = mx;
    __syncthreads();
    mx = -INFINITY;
    for (int w = 0; w < nw; ++w) mx = fmax(mx, sh[w]);
    __syncthreads();
    double s = 0.0;
    for (uint32_t p = tid; p < n; p += blockDim.x) {
        const bool in = !set || ((set[p >> 6] >> (p & 63)) & 1);
        if (in && p >= first) s += exp(v[p] - mx);
    }
    s = wave_sum(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    s = 0.0;
    for (int w = 0; w < nw; ++w) s += sh[w];
    __syncthreads();
    return mx + log(s);
}

__global__ __launch_bounds__(256) void hc_posterior_kernel(const double *__restrict__ final_vec, uint32_t n_paths,
                                                            const uint64_t *__restrict__ sets, uint32_t set_words,
                                                            double *__restrict__ conf) {
    __shared__ double sh[16];
    const double total = block_lse(final_vec, nullptr, n_paths, sh);
    const double part = block_lse(final_vec, sets + (size_t)blockIdx.x * set_words, n_paths, sh);
    if (threadIdx.x == 0) conf[blockIdx.x] = exp(part - total);
}

// ---------------------------------------------------------------------------------------------- launchers
void launch_hc_segments(const HcGraphDev &g, const HcBatchDev &b, const HcParamsDev &prm, double *segS, double *segU,
                        double *segD, double *nodeW, double *totals, hipStream_t st) {
    if (b.n_reads == 0) return;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)b.n_reads + SEG_WAVES - 1) / SEG_WAVES, 256u * 8u);
    hipLaunchKernelGGL(hc_segment_kernel, dim3(blocks), dim3(SEG_WAVES * 64), 0, st, g, b, prm, segS, segU, segD, nodeW,
                       totals);
}

void launch_hc_sweep(const HcGraphDev &g, const uint32_t *item_node, const double *D, uint32_t n_items, int skip_zero,
                     double *acc, hipStream_t st) {
    if (n_items == 0) return;
    const uint32_t n_tiles = g.row_words / SWEEP_TW;
    // aim for ~8 waves per SIMD over the whole chip
    const uint32_t target_waves = 256u * 4u * 8u;
    uint32_t n_chunks = std::max(1u, target_waves / n_tiles);
    uint32_t per = (n_items + n_chunks - 1) / n_chunks;
    per = std::max(per, 64u);
    n_chunks = (n_items + per - 1) / per;
    const uint64_t waves = (uint64_t)n_chunks * n_tiles;
    const uint32_t blocks = (uint32_t)((waves + 3) / 4);
    hipLaunchKernelGGL(hc_sweep_kernel, dim3(blocks), dim3(256), 0, st, g.umask, g.row_words, item_node, D, n_items, per,
                       n_tiles, skip_zero, acc);
}

void launch_hc_finish(const double *totals, const double *acc_seg, const double *acc_node, uint32_t n_paths, double *out,
                      hipStream_t st) {
    hipLaunchKernelGGL(hc_finish_kernel, dim3((n_paths + 255) / 256), dim3(256), 0, st, totals, acc_seg, acc_node,
                       n_paths, out);
}

void launch_hc_read_loglik(const HcGraphDev &g, const HcBatchDev &b, const double *segS, const double *segU, double *out,
                           hipStream_t st) {
    if (b.n_reads == 0) return;
    hipLaunchKernelGGL(hc_read_loglik_kernel, dim3((g.n_paths + 255) / 256, b.n_reads), dim3(256), 0, st, g, b, segS,
                       segU, out);
}

void launch_hc_posterior(const double *final_vec, uint32_t n_paths, const uint64_t *sets, uint32_t set_words,
                         uint32_t n_sets, double *conf, hipStream_t st) {
    if (n_sets == 0) return;
    hipLaunchKernelGGL(hc_posterior_kernel, dim3(n_sets), dim3(256), 0, st, final_vec, n_paths, sets, set_words, conf);
}

} // namespace vgan
