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
//   hc_segment_tile_kernel   S_m, U_m per segment.  A 256-thread workgroup takes a tile of up to 8 reads:
//                       (1) one wave per read: quality-window prefix sums by a DPP wave scan into LDS, then one
//                       lane per segment computes U_m and marks its columns; (2) one lane per alignment column,
//                       flat over the tile: the log term, reduced per segment with LDS fp64 atomics; (3) one lane
//                       per segment: D_m = S_m - U_m streamed out.  hc_segment_general_kernel is the same arithmetic
//                       for reads of any length (one wave per read).
//   hc_nodeacc_kernel   NODE_WEIGHTS mode: W[node] += D_m with a workgroup-private W in LDS (ds_add_f64),
//                       flushed once per workgroup with coalesced global atomics.
//   hc_sweep_kernel     the per-path update acc[p] += D for every path NOT supported by the node
//                       (final[p] = sum_m S_m - acc[p]: no cancellation).  Mask rows are stored bit-transposed
//                       (hc_device.h): lane l loads ONE 16-bit entry holding its path's bit for each of the tile's
//                       words; v_add_co_u32 m,m,m peels the top bit of every lane into an SGPR pair, which becomes
//                       EXEC for one v_add_f64 updating the 64 paths of that word.  Tile = blockIdx % 8, so each
//                       XCD's L2 holds one eighth of the table.  Used per segment (PER_READ modes) and per node
//                       (NODE_WEIGHTS mode, D = W[node], once per finalize).
//   hc_read_loglik_kernel  literal per-read x per-path vectors (debug / parity aid).
//   hc_finish_kernel    final[p] = Stot - acc[p].
//   hc_posterior_kernel log-sum-exp over path sets (src/get_posterior.cpp:78-127).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_math.h"
#include "hc_device.h"

namespace vgan {

// one column's log term: src/process_mapping.cpp:59-77 with get_p_obs_base.cpp:67 (tv = ts = 0)
__device__ __forceinline__ double column_term(uint32_t gc, uint32_t rc, double e, double match, double pcm,
                                              const HcParamsDev &prm) {
    const double eps = gc == rc ? e : 1.0 - e;
    const double pobs = match * (1.0 - eps);
    const double x = prm.consensus ? (1.0 - prm.bep) * pobs : (1.0 - pcm) * bg_freq(rc) + pcm * pobs;
    return log_pos(x);
}

// ---------------------------------------------------------------------------------------------- segments (general)
// One wave per read, any read length.  LDS: tables + per-wave prefix sums of the quality window.
constexpr int SEG_WAVES = 4;
constexpr int SEG_MAXQ = 1024;

__global__ __launch_bounds__(SEG_WAVES * 64) void hc_segment_general_kernel(HcGraphDev g, HcBatchDev b, HcParamsDev prm,
                                                                             double *__restrict__ segS,
                                                                             double *__restrict__ segU,
                                                                             double *__restrict__ segD,
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
                const double s = wave_incl_scan(v);
                if (j < QL) ps[j + 1] = carry + s;
                carry += __shfl(s, 63, 64);
            }
        }
        if (lane == 0) ps[0] = 0.0;
        first90 = wave_min_u32(first90);
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
                    S += column_term(gc, rc, use_bep ? prm.bep : qs_s[q], nd.match, pcm, prm);
                }
                if (segS) segS[s] = S;
                if (segU) segU[s] = U;
                if (segD) segD[s] = S - U;
                sumS += S;
                sumU += U;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    sumS = wave_sum(sumS);
    sumU = wave_sum(sumU);
    if (lane == 0 && totals) {
        unsafeAtomicAdd(&totals[0], sumS);
        unsafeAtomicAdd(&totals[1], sumU);
    }
}

// ---------------------------------------------------------------------------------------------- segments (tiled)
constexpr int ST_THREADS = 256;
constexpr int ST_WAVES = ST_THREADS / 64;
constexpr int ST_READS = 8;   // reads per tile (at most)
constexpr int ST_COLS = 1280; // LDS capacity per tile: alignment columns,
constexpr int ST_QUAL = 1280; //                        quality bytes,
constexpr int ST_SEGS = 512;  //                        segments
constexpr int ST_MAXQ = HC_TILE_MAX_READ_QUAL;

struct alignas(16) StSeg {
    double pcm;      // (1 - incorrect_mapping_vec[mapq]) * mappability[node]   (process_mapping.cpp:41)
    double match;    // pow(1 - mu(node), 8)                                     (get_p_obs_base.cpp:64)
    uint32_t cstart; // tile-local column of the segment's first base
    uint32_t rbase;  // tile-local column of the read's first base (Q4: read bases are taken from the read start)
    uint32_t A;      // |algnseq| of the read
    uint32_t use_bep;
};

// Copies bytes [g0, g0+n) of src into dst (LDS) as aligned dwords; dst[i + (g0 & 3)] = src[g0 + i].
// n <= 1280, so every thread moves at most two dwords (fixed trip count: no loop bookkeeping).
__device__ __forceinline__ void stage_bytes(uint8_t *dst, const uint8_t *__restrict__ src, uint32_t g0, uint32_t n, int tid) {
    static_assert(ST_COLS <= 2 * 4 * ST_THREADS - 8 && ST_QUAL <= 2 * 4 * ST_THREADS - 8, "two dwords per thread");
    const uint32_t a0 = g0 & ~3u;
    const uint32_t nd = (g0 + n - a0 + 3u) >> 2;
    const uint32_t *__restrict__ s32 = reinterpret_cast<const uint32_t *>(src + a0);
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst);
    const uint32_t i0 = tid, i1 = tid + ST_THREADS;
    const uint32_t v0 = i0 < nd ? s32[i0] : 0u;
    const uint32_t v1 = i1 < nd ? s32[i1] : 0u;
    if (i0 < nd) d32[i0] = v0;
    if (i1 < nd) d32[i1] = v1;
}

__global__ __launch_bounds__(ST_THREADS) void hc_segment_tile_kernel(HcGraphDev g, HcBatchDev b, HcParamsDev prm,
                                                                      uint32_t reads_per_block,
                                                                      double *__restrict__ segD_out,
                                                                      double *__restrict__ totals) {
    __shared__ double lq_s[256];
    __shared__ double qs_s[100];
    __shared__ double bg_s[4];
    __shared__ double ps_s[ST_WAVES][ST_MAXQ + 1];
    __shared__ double segS_s[ST_SEGS];
    __shared__ StSeg segpm_s[ST_SEGS];
    __shared__ uint16_t colseg_s[ST_COLS];
    __shared__ __attribute__((aligned(16))) uint8_t qcol_s[ST_COLS + 8]; // clamped quality per tile column (0 past the read's qualities)
    __shared__ __attribute__((aligned(16))) uint8_t gseq_s[ST_COLS + 8];
    __shared__ __attribute__((aligned(16))) uint8_t rseq_s[ST_COLS + 8];
    __shared__ __attribute__((aligned(16))) uint8_t qual_s[ST_QUAL + 8];
    __shared__ uint32_t off_s[3][ST_READS + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 256; i += ST_THREADS) lq_s[i] = g.lq[i];
    for (int i = tid; i < 100; i += ST_THREADS) qs_s[i] = g.qscore[i];
    if (tid < 4) bg_s[tid] = tid == 0 ? 0.27532 : tid == 1 ? 0.30044 : tid == 2 ? 0.25780 : 0.16644; // A C T G by (c>>1)&3
    const double lq0 = g.lq[0];
    double sumT = 0.0, sumU = 0.0; // sum of the column terms (= sum of S_m) and of U_m, each without cancellation

    const uint32_t rb0 = blockIdx.x * reads_per_block;
    const uint32_t rb1 = min(b.n_reads, rb0 + reads_per_block);
    uint32_t r0 = rb0;
    while (r0 < rb1) {
        // ---- tile header: offsets of up to ST_READS reads
        if (tid <= ST_READS) {
            const uint32_t r = min(r0 + tid, rb1);
            off_s[0][tid] = b.read_seg_off[r];
            off_s[1][tid] = b.read_col_off[r];
            off_s[2][tid] = b.read_qual_off[r];
        }
        __syncthreads();
        const uint32_t seg_base = off_s[0][0], col_base = off_s[1][0], q_base = off_s[2][0];
        uint32_t n = 1; // one read always fits: the host selects this kernel only for reads within the per-read limits
        while (n < (uint32_t)ST_READS && r0 + n < rb1 && off_s[0][n + 1] - seg_base <= (uint32_t)ST_SEGS &&
               off_s[1][n + 1] - col_base <= (uint32_t)ST_COLS && off_s[2][n + 1] - q_base <= (uint32_t)ST_QUAL)
            ++n;
        const uint32_t n_seg = off_s[0][n] - seg_base, n_col = off_s[1][n] - col_base, n_q = off_s[2][n] - q_base;
        const uint32_t cshift = col_base & 3u, qshift = q_base & 3u;
        // ---- stage the tile's read / graph / quality windows in LDS (coalesced dword loads)
        stage_bytes(gseq_s, b.graph_seq, col_base, n_col, tid);
        stage_bytes(rseq_s, b.algnseq, col_base, n_col, tid);
        stage_bytes(qual_s, b.qual, q_base, n_q, tid);
        {
            uint32_t *cs32 = reinterpret_cast<uint32_t *>(colseg_s);
            uint32_t *qc32 = reinterpret_cast<uint32_t *>(qcol_s);
            const uint32_t nc2 = (n_col + 1) >> 1, nc4 = (n_col + 3) >> 2;
#pragma unroll
            for (int it = 0; it < (ST_COLS / 2 + ST_THREADS - 1) / ST_THREADS; ++it) {
                const uint32_t i = tid + it * ST_THREADS;
                if (i < nc2) cs32[i] = 0xFFFFFFFFu;
                if (i < nc4) qc32[i] = 0u;
            }
        }
        __syncthreads();

        // ---- phase 1: one wave per read: quality prefix sums (DPP scan), then one lane per segment
        for (uint32_t k = wave; k < n; k += ST_WAVES) {
            const uint32_t r = r0 + k;
            const uint32_t qoff = off_s[2][k] - q_base, QL = off_s[2][k + 1] - off_s[2][k];
            const uint32_t A = b.read_algn_len[r];
            const uint32_t colbase = off_s[1][k] - col_base;
            const uint32_t ncols_k = off_s[1][k + 1] - off_s[1][k];
            const double pinc = g.incmap[b.read_mapq[r]];
            double *ps = ps_s[wave];
            // prefix sums of log p_err over the read's quality bytes: each lane owns E consecutive bytes
            // (E <= 4 since QL <= 256), one DPP wave scan over the lane totals
            const uint32_t E = (QL + 63u) >> 6;
            const uint32_t jb = lane * E;
            double loc[4];
            double tot = 0.0;
            uint32_t hot = 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                const uint32_t j = jb + e;
                const bool in = e < E && j < QL;
                const uint32_t qb = in ? qual_s[qoff + qshift + j] : 0u;
                if (in && (int)(int8_t)qb >= 90) hot = min(hot, j);
                if (in && j < ncols_k) { // quality by alignment column for phase 2, clamped as qscore_vec's index
                    const int qi = (int)(int8_t)qb;
                    qcol_s[colbase + j] = (uint8_t)(qi < 0 ? 0 : (qi > 99 ? 99 : qi));
                }
                tot += in ? lq_s[qb] : 0.0;
                loc[e] = tot;
            }
            const double base_sum = wave_incl_scan(tot) - tot;
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                const uint32_t j = jb + e;
                if (e < E && j < QL) ps[j + 1] = base_sum + loc[e];
            }
            const uint64_t hot_lanes = __builtin_amdgcn_ballot_w64(hot != 0xFFFFFFFFu);
            const uint32_t first90 =
                hot_lanes ? (uint32_t)__builtin_amdgcn_readlane((int)hot, (int)__builtin_ctzll(hot_lanes)) : 0xFFFFFFFFu;
            if (lane == 0) ps[0] = 0.0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t s0 = off_s[0][k], s1 = off_s[0][k + 1];
            for (uint32_t sb = s0; sb < s1; sb += 64) {
                const uint32_t s = sb + lane;
                if (s < s1) {
                    const uint32_t ls = s - seg_base;
                    const uint32_t start = b.seg_start[s], len = b.seg_len[s];
                    const HcNodeDev nd = g.node_tab[b.seg_node[s]];
                    const uint32_t lo = min(start, QL), hi = min(start + A, QL);
                    const double U = (ps[hi] - ps[lo]) + (double)(A - (hi - lo)) * lq0; // Q5 zero padding
                    segS_s[ls] = -U; // phase 2 adds the column terms: the slot ends as D_m = S_m - U_m
                    sumU += U;
                    const uint32_t use_bep = (prm.use_bep || first90 < hi) ? 1u : 0u; // update_likelihood.cpp:42
                    segpm_s[ls] = StSeg{(1.0 - pinc) * nd.mappability, nd.match, colbase + start, colbase, A, use_bep};
                    const uint32_t cend = min(colbase + start + len, n_col);
                    for (uint32_t c = colbase + start; c < cend; ++c) colseg_s[c] = (uint16_t)ls;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // ps is rewritten by this wave's next read
        }
        __syncthreads();

        // ---- phase 2: one lane per alignment column, flat over the tile; everything comes from LDS.
        // Loads are issued up front and the arithmetic is unconditional (no early exits): the dependency chain is
        // column -> {segment id, graph base, quality} -> segment record -> read base.
        for (uint32_t c = tid; c < n_col; c += ST_THREADS) {
            const uint32_t ls_raw = colseg_s[c];
            const uint32_t gc = gseq_s[c + cshift];
            const uint32_t q = qcol_s[c];
            const bool in_seg = ls_raw != 0xFFFFu; // columns no mapping scores (Q6 tail) carry 0xFFFF
            const uint32_t ls = in_seg ? ls_raw : 0u;
            const StSeg sg = segpm_s[ls];
            const double qsv = qs_s[q];
            const uint32_t j = c - sg.cstart;
            uint32_t rc = rseq_s[min(sg.rbase + j, n_col - 1) + cshift]; // Q4: read bases from the read start
            rc = j < sg.A ? rc : 0u;
            const bool valid = in_seg && is_acgt(gc) && is_acgt(rc); // process_mapping.cpp:62-63
            const double e = sg.use_bep ? prm.bep : qsv;
            const double eps = gc == rc ? e : 1.0 - e;          // get_p_obs_base.cpp:3-27
            const double pobs = sg.match * (1.0 - eps);         // get_p_obs_base.cpp:67 with tv = ts = 0
            const double bgv = bg_s[(rc >> 1) & 3u];
            const double x = prm.consensus ? (1.0 - prm.bep) * pobs : (1.0 - sg.pcm) * bgv + sg.pcm * pobs;
            const double t = log_pos(valid ? x : 1.0); // log(1) = 0 for the lanes that do not count
            sumT += t;
            if (valid) unsafeAtomicAdd(&segS_s[ls], t);
        }
        __syncthreads();

        // ---- phase 3: one lane per segment
        for (uint32_t ls = tid; ls < n_seg; ls += ST_THREADS) {
            if (segD_out) segD_out[seg_base + ls] = segS_s[ls];
        }
        r0 += n;
        // the next tile's barriers order phase 3 against the next phase 1 (phase 3 touches segS/segU only)
    }
    sumT = wave_sum(sumT);
    sumU = wave_sum(sumU);
    if (lane == 0 && totals) {
        unsafeAtomicAdd(&totals[0], sumT);
        unsafeAtomicAdd(&totals[1], sumU);
    }
}

// ---------------------------------------------------------------------------------------------- node accumulate
// W[node] += D_m with a workgroup-private copy of W in LDS.
__global__ __launch_bounds__(1024) void hc_nodeacc_lds_kernel(const uint32_t *__restrict__ seg_node,
                                                               const double *__restrict__ segD, uint32_t n_items,
                                                               uint32_t rows, double *__restrict__ nodeW) {
    extern __shared__ double w_s[];
    for (uint32_t j = threadIdx.x; j < rows; j += blockDim.x) w_s[j] = 0.0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x)
        unsafeAtomicAdd(&w_s[seg_node[i]], segD[i]);
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < rows; j += blockDim.x) {
        const double v = w_s[j];
        if (v != 0.0) unsafeAtomicAdd(&nodeW[j], v);
    }
}

__global__ void hc_nodeacc_global_kernel(const uint32_t *__restrict__ seg_node, const double *__restrict__ segD,
                                         uint32_t n_items, double *__restrict__ nodeW) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x)
        unsafeAtomicAdd(&nodeW[seg_node[i]], segD[i]);
}

// ---------------------------------------------------------------------------------------------- sweep
#define VG_X(k) "v_add_co_u32 %[m], %[p" #k "], %[m], %[m]\n\t"
#define VG_A(k) "s_mov_b64 exec, %[p" #k "]\n\tv_add_f64 %[a" #k "], %[a" #k "], %[d]\n\t"

// Each helper peels the next B top bits of m (all lanes) into SGPR pairs and applies them as EXEC masks.
// Precondition: EXEC is all ones (block size is a multiple of 64 and control flow is wave uniform here).
__device__ __forceinline__ void sweep_apply1(double &a0, uint32_t &m, double d) {
    uint64_t p0;
    asm volatile(VG_X(0) VG_A(0) "s_mov_b64 exec, -1" : [m] "+v"(m), [p0] "=&s"(p0), [a0] "+v"(a0) : [d] "s"(d));
}
__device__ __forceinline__ void sweep_apply2(double *a, uint32_t &m, double d) {
    uint64_t p0, p1;
    asm volatile(VG_X(0) VG_X(1) VG_A(0) VG_A(1) "s_mov_b64 exec, -1"
                 : [m] "+v"(m), [p0] "=&s"(p0), [p1] "=&s"(p1), [a0] "+v"(a[0]), [a1] "+v"(a[1])
                 : [d] "s"(d));
}
__device__ __forceinline__ void sweep_apply4(double *a, uint32_t &m, double d) {
    uint64_t p0, p1, p2, p3;
    asm volatile(VG_X(0) VG_X(1) VG_X(2) VG_X(3) VG_A(0) VG_A(1) VG_A(2) VG_A(3) "s_mov_b64 exec, -1"
                 : [m] "+v"(m), [p0] "=&s"(p0), [p1] "=&s"(p1), [p2] "=&s"(p2), [p3] "=&s"(p3), [a0] "+v"(a[0]),
                   [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3])
                 : [d] "s"(d));
}
__device__ __forceinline__ void sweep_apply8(double *a, uint32_t &m, double d) {
    uint64_t p0, p1, p2, p3, p4, p5, p6, p7;
    asm volatile(VG_X(0) VG_X(1) VG_X(2) VG_X(3) VG_X(4) VG_X(5) VG_X(6) VG_X(7) VG_A(0) VG_A(1) VG_A(2) VG_A(3) VG_A(4)
                     VG_A(5) VG_A(6) VG_A(7) "s_mov_b64 exec, -1"
                 : [m] "+v"(m), [p0] "=&s"(p0), [p1] "=&s"(p1), [p2] "=&s"(p2), [p3] "=&s"(p3), [p4] "=&s"(p4),
                   [p5] "=&s"(p5), [p6] "=&s"(p6), [p7] "=&s"(p7), [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]),
                   [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]), [a6] "+v"(a[6]), [a7] "+v"(a[7])
                 : [d] "s"(d));
}
#undef VG_X
#undef VG_A

template <int B> __device__ __forceinline__ void sweep_apply(double *a, uint32_t &m, double d) {
    if constexpr (B >= 8) {
        sweep_apply8(a, m, d);
        sweep_apply<B - 8>(a + 8, m, d);
    } else if constexpr (B >= 4) {
        sweep_apply4(a, m, d);
        sweep_apply<B - 4>(a + 4, m, d);
    } else if constexpr (B >= 2) {
        sweep_apply2(a, m, d);
        sweep_apply<B - 2>(a + 2, m, d);
    } else if constexpr (B == 1) {
        sweep_apply1(a[0], m, d);
    }
}

constexpr int SWEEP_UNROLL = 8;

struct alignas(32) U32x8 {
    uint32_t v[8];
};
struct alignas(64) F64x8 {
    double v[8];
};

// items [0, n_items): item i has node = IDENT ? i : item_node[i] and weight D[i].
// Wave = (chunk of items, tile); tile t owns words [tile_word0[t], tile_word0[t+1]): TB or TB+1 of them.
// items_per_wave is a multiple of SWEEP_UNROLL and the arrays are 64-byte aligned, so a block of 8 node ids / weights
// is one aligned scalar load.
template <int TB, bool IDENT>
__global__ __launch_bounds__(256) void hc_sweep_kernel(const uint16_t *__restrict__ umaskT, uint32_t row_entries,
                                                        const uint32_t *__restrict__ item_node,
                                                        const double *__restrict__ D, uint32_t n_items,
                                                        uint32_t items_per_wave, uint32_t n_tiles,
                                                        const uint16_t *__restrict__ tile_word0, int skip_zero,
                                                        double *__restrict__ acc_out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile = blockIdx.x % n_tiles;
    const uint32_t chunk = (blockIdx.x / n_tiles) * 4 + wave;
    const uint32_t w0 = tile_word0[tile];
    const uint32_t tw = tile_word0[tile + 1] - w0;
    const uint64_t i0_64 = (uint64_t)chunk * items_per_wave;
    if (tw == 0 || i0_64 >= n_items) return;
    const uint32_t i0 = (uint32_t)i0_64;
    const uint32_t i1 = (uint32_t)min((uint64_t)n_items, i0_64 + items_per_wave);
    const uint32_t full = i0 + (i1 - i0) / SWEEP_UNROLL * SWEEP_UNROLL;
    const bool extra = tw > (uint32_t)TB;
    const uint32_t lane_off = tile * 64 + lane;

    double acc[TB + 1];
#pragma unroll
    for (int k = 0; k <= TB; ++k) acc[k] = 0.0;

    auto load_block = [&](uint32_t i, uint32_t(&m)[SWEEP_UNROLL]) {
        U32x8 nd;
        if (!IDENT) nd = *reinterpret_cast<const U32x8 *>(item_node + i);
#pragma unroll
        for (int u = 0; u < SWEEP_UNROLL; ++u) {
            const uint32_t node = IDENT ? i + u : nd.v[u];
            const uint16_t *row = umaskT + (size_t)node * row_entries;
            m[u] = row[lane_off];
        }
    };
    auto apply = [&](uint32_t entry, double d) {
        uint32_t m = entry << 16;
        if (skip_zero && (d == 0.0 || __builtin_amdgcn_ballot_w64(m != 0) == 0)) return;
        sweep_apply<TB>(acc, m, d);
        if (extra) sweep_apply1(acc[TB], m, d);
    };

    uint32_t cur[SWEEP_UNROLL], nxt[SWEEP_UNROLL];
    if (i0 < full) load_block(i0, cur);
    for (uint32_t i = i0; i < full; i += SWEEP_UNROLL) {
        const bool more = i + SWEEP_UNROLL < full;
        if (more) load_block(i + SWEEP_UNROLL, nxt); // in flight while this block is applied
        const F64x8 d = *reinterpret_cast<const F64x8 *>(D + i);
#pragma unroll
        for (int u = 0; u < SWEEP_UNROLL; ++u) apply(cur[u], d.v[u]);
        if (more) {
#pragma unroll
            for (int u = 0; u < SWEEP_UNROLL; ++u) cur[u] = nxt[u];
        }
    }
    for (uint32_t i = full; i < i1; ++i) { // tail of the last chunk
        const uint32_t node = IDENT ? i : item_node[i];
        apply(umaskT[(size_t)node * row_entries + lane_off], D[i]);
    }
#pragma unroll
    for (int k = 0; k <= TB; ++k) {
        if ((uint32_t)k < tw && acc[k] != 0.0) unsafeAtomicAdd(&acc_out[(size_t)(w0 + k) * 64 + lane], acc[k]);
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
        const uint64_t w = g.umask[(size_t)b.seg_node[s] * g.mask_words + (p >> 6)];
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
void launch_hc_segments(const HcGraphDev &g, const HcBatchDev &b, const HcParamsDev &prm, bool tiled, double *segS,
                        double *segU, double *segD, double *totals, hipStream_t st) {
    if (b.n_reads == 0) return;
    if (tiled && !segS && !segU) { // the tiled kernel produces D_m only; separate S_m / U_m (debug API) come from the general one
        // ~6 resident workgroups per CU; contiguous read ranges per workgroup
        const uint32_t want_blocks = 256u * 6u * 2u;
        uint32_t per = (b.n_reads + want_blocks - 1) / want_blocks;
        per = std::max(per, (uint32_t)ST_READS);
        const uint32_t blocks = (b.n_reads + per - 1) / per;
        hipLaunchKernelGGL(hc_segment_tile_kernel, dim3(blocks), dim3(ST_THREADS), 0, st, g, b, prm, per, segD, totals);
    } else {
        const uint32_t blocks =
            (uint32_t)std::min<uint64_t>(((uint64_t)b.n_reads + SEG_WAVES - 1) / SEG_WAVES, 256u * 8u);
        hipLaunchKernelGGL(hc_segment_general_kernel, dim3(blocks), dim3(SEG_WAVES * 64), 0, st, g, b, prm, segS, segU,
                           segD, totals);
    }
}

int launch_hc_nodeacc(const uint32_t *seg_node, const double *segD, uint32_t n_items, uint32_t rows, double *nodeW,
                      hipStream_t st) {
    if (n_items == 0) return 0;
    const size_t lds = (size_t)rows * sizeof(double);
    if (lds <= 150u * 1024u) {
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void *)hc_nodeacc_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024) != hipSuccess)
                return -1;
            attr_set = true;
        }
        const uint32_t blocks = (uint32_t)std::min<uint64_t>(256, ((uint64_t)n_items + 1023) / 1024);
        hipLaunchKernelGGL(hc_nodeacc_lds_kernel, dim3(blocks), dim3(1024), lds, st, seg_node, segD, n_items, rows, nodeW);
    } else {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>(256u * 8u, ((uint64_t)n_items + 255) / 256);
        hipLaunchKernelGGL(hc_nodeacc_global_kernel, dim3(blocks), dim3(256), 0, st, seg_node, segD, n_items, nodeW);
    }
    return 0;
}

template <int TB>
static void launch_sweep_tb(const HcGraphDev &g, const uint32_t *item_node, const double *D, uint32_t n_items,
                            uint32_t per, uint32_t blocks, int skip_zero, double *acc, hipStream_t st) {
    if (item_node)
        hipLaunchKernelGGL((hc_sweep_kernel<TB, false>), dim3(blocks), dim3(256), 0, st, g.umaskT, g.row_entries, item_node, D,
                           n_items, per, g.n_tiles, g.tile_word0, skip_zero, acc);
    else
        hipLaunchKernelGGL((hc_sweep_kernel<TB, true>), dim3(blocks), dim3(256), 0, st, g.umaskT, g.row_entries, item_node, D,
                           n_items, per, g.n_tiles, g.tile_word0, skip_zero, acc);
}

void launch_hc_sweep(const HcGraphDev &g, const uint32_t *item_node, const double *D, uint32_t n_items, int skip_zero,
                     double *acc, hipStream_t st) {
    if (n_items == 0) return;
    // chunk-blocks of 4 waves per tile; ~2 full residencies of 8 waves/SIMD for tail balance
    uint32_t chunk_blocks = std::max(1u, (256u * 4u * 8u * 2u) / (4u * g.n_tiles));
    uint32_t per = (n_items + chunk_blocks * 4 - 1) / (chunk_blocks * 4);
    per = std::max(per, 64u);
    per = (per + SWEEP_UNROLL - 1) / SWEEP_UNROLL * SWEEP_UNROLL;
    chunk_blocks = (n_items + per * 4 - 1) / (per * 4);
    const uint32_t blocks = chunk_blocks * g.n_tiles;
    switch (g.tile_base_words) {
#define VG_CASE(TB)                                                                     \
    case TB:                                                                            \
        launch_sweep_tb<TB>(g, item_node, D, n_items, per, blocks, skip_zero, acc, st); \
        break;
        VG_CASE(0)
        VG_CASE(1)
        VG_CASE(2)
        VG_CASE(3)
        VG_CASE(4)
        VG_CASE(5)
        VG_CASE(6)
        VG_CASE(7)
        VG_CASE(8)
        VG_CASE(9)
        VG_CASE(10)
        VG_CASE(11)
        VG_CASE(12)
        VG_CASE(13)
        VG_CASE(14)
        VG_CASE(15)
#undef VG_CASE
    default:
        break; // unreachable: tile_base_words <= 15 by construction (hc_capi.hip)
    }
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
