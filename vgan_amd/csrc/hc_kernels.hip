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
//   hc_segment_tile_kernel   D_m = S_m - U_m per segment and, in NODE_WEIGHTS mode, W[node] += D_m through an LDS
//                       window over the node ids of the workgroup's reads (the batch is sorted by node id), flushed to
//                       HBM a few times per launch.  A 256-thread workgroup takes a tile of up to 8 (24 for short)
//                       reads staged in LDS and works flat over it (phases B..E at the kernel); a tile's data leaves HBM
//                       one tile ahead.  hc_segment_general_kernel is the same arithmetic for reads that do not fit a
//                       tile (one wave per read, any length).
//   hc_sweep_kernel     the per-path update acc[p] += D for every path NOT supported by the node
//                       (final[p] = sum_m S_m - acc[p]: no cancellation).  Mask rows are stored bit-transposed
//                       (hc_device.h): lane l loads ONE 16-bit entry holding its path's bit for each of the tile's
//                       words; v_add_co_u32 m,m,m peels the top bit of every lane into an SGPR pair, which becomes
//                       EXEC for one v_add_f64 updating the 64 paths of that word.  Tile = blockIdx % 8, so each
//                       XCD's L2 holds one eighth of the table.  Used per segment (PER_READ modes) and per node
//                       (NODE_WEIGHTS mode, D = W[node], once per finalize).
//   hc_read_loglik_kernel  literal per-read x per-path vectors (debug / parity aid).
//   hc_finish_kernel    final[p] = Stot - acc[p]; leaves the node-pass accumulators zero.
//   hc_posterior_kernel log-sum-exp over path sets (src/get_posterior.cpp:78-127).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

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
                                                                             uint32_t r_begin,
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

    for (uint32_t r = r_begin + blockIdx.x * SEG_WAVES + wave; r < b.n_reads; r += gridDim.x * SEG_WAVES) {
        const uint32_t seg0 = b.read_seg_off[r], seg1 = b.read_seg_off[r + 1];
        const uint32_t col0 = b.read_col_off[r];
        const uint32_t q0 = b.read_qual_off[r];
        // (the batch contract bounds a quality string at 65535 bytes; clamped so that a caller's error cannot leave ps_s)
        const uint32_t QL = min(b.read_qual_off[r + 1] - q0, (uint32_t)SEG_MAXQ * 64u);
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
            const double v = j < QL ? lq_s[qb] : 0.0;
            const double s = wave_incl_scan(v);
            if (use_lds) {
                if (j < QL) ps[j + 1] = carry + s; // every prefix
            } else if (lane == 63) {
                ps[(base >> 6) + 1] = carry + s;   // longer quality strings: one prefix per 64 bytes (QL <= 65535)
            }
            carry += __shfl(s, 63, 64);
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
                } else { // coarse prefix + at most 63 terms at either end
                    double Phi = ps[hi >> 6], Plo = ps[lo >> 6];
                    for (uint32_t j = hi & ~63u; j < hi; ++j) Phi += lq_s[b.qual[q0 + j]];
                    for (uint32_t j = lo & ~63u; j < lo; ++j) Plo += lq_s[b.qual[q0 + j]];
                    U = (Phi - Plo) + (double)(A - (hi - lo)) * lq0;
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
                if (nodeW) unsafeAtomicAdd(&nodeW[node], S - U); // the few reads outside the tile contract: W in HBM directly
                sumS += S;
                sumU += U;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    sumS = wave_sum(sumS);
    sumU = wave_sum(sumU);
    if (lane == 0 && totals) {
        double *t = totals + ((blockIdx.x * SEG_WAVES + wave) % HC_TOTAL_SLOTS) * HC_TOTAL_STRIDE;
        unsafeAtomicAdd(&t[0], sumS);
        unsafeAtomicAdd(&t[1], sumU);
    }
}

// ---------------------------------------------------------------------------------------------- segments (tiled)
// Everything is flat over a tile of up to ST_READS reads staged in LDS -- no per-read wave work:
//   top  the tile's graph / read / quality bytes go from registers to LDS (they were requested from HBM during the
//        previous tile's phase D, like the segment records and, a tile earlier still, the read offsets)
//   B  lane per 5 quality bytes: log p_err prefix sums (thread-local + one DPP wave scan; wave totals folded in by the
//      consumers), so that U_m is a difference of two prefix values
//   C  lane per segment: U_m, the segment record {kappa, lw, column bounds, index shifts}; a head bit at its first column
//   D  lane per alignment column: owner segment = number of head bits at or below the column - 1 (the segments of a
//      tile cover ascending column ranges), the column's log term from two table reads and a short log1p series,
//      LDS fp64 atomic into the segment's sum
//   E  lane per segment: D_m = S_m - U_m added to W[node] through the workgroup's LDS window over node ids (and / or
//      streamed out)
//
// The column term (process_mapping.cpp:59-77, get_p_obs_base.cpp:67 with tv = ts = 0) is
//     log(x),  x = wbg * bg(read base) + wobs * om,   om = (graph base == read base) ? 1 - e(Q) : e(Q)
// with wbg = 1 - pcm, wobs = pcm * match, pcm = (1 - p_inc(mapq)) * mappability(node).  Factorised:
//     log(x) = log(wobs) + log(om) + log1p(rho),   rho = (wbg / wobs) * bg / om
// log(wobs) = lw is a sum of a per-read and a per-node table value (both taken on the host), log(om) and 1/om come from a
// 101 x 2 table built per workgroup, and rho is ~1e-7 for a confidently mapped read, so log1p is a degree-8 series; a
// column with rho >= 2^-6 (low mapping quality, mismatch at a very high base quality) or a segment with wobs = 0
// (mapq 0) takes the table-driven log of x itself.  No log, no division per column or segment.
constexpr int ST_THREADS = 256;
constexpr int ST_WAVES = ST_THREADS / 64;
constexpr int ST_READS_LONG = 8;   // reads per tile (at most) for batches of long reads (compare-sum read lookup),
constexpr int ST_READS_SHORT = 24; // ... of short reads: 40-column reads (ancient DNA) still fill most of a tile
constexpr int ST_READS_LOG2 = 5; // steps of the segment -> read search (2^5 >= ST_READS)
constexpr int ST_COLS = 1280; // LDS capacity per tile: alignment columns,
constexpr int ST_QUAL = 1280; //                        quality bytes,
constexpr int ST_SEGS = 512;  //                        segments
constexpr int ST_WIN = 448;   // node ids covered by the workgroup's W window (what 40 KB of LDS per workgroup leave)
constexpr int ST_WIN_SHORT = 288; // ... in the short-read variant
constexpr int ST_QB = ST_QUAL / ST_THREADS;         // quality bytes per lane in phase B
constexpr int ST_QW = ST_QB * 64;                   // quality bytes per wave in phase B
constexpr int ST_SEG_ITERS = ST_SEGS / ST_THREADS;  // segments per lane in phases C and E
constexpr int ST_FWORDS = ST_COLS / 32;             // head-bit words
static_assert((ST_SEGS & (ST_SEGS - 1)) == 0, "segment index mask");
static_assert((1 << ST_READS_LOG2) >= ST_READS_SHORT && ST_READS_SHORT < 64, "read search / header lanes");
static_assert(HC_TILE_MAX_READ_COLS <= (uint32_t)ST_COLS && HC_TILE_MAX_READ_QUAL <= (uint32_t)ST_QUAL &&
                  HC_TILE_MAX_READ_SEGS <= (uint32_t)ST_SEGS,
              "a tileable read fits one tile");
static_assert(ST_QB * ST_THREADS == ST_QUAL && ST_SEG_ITERS * ST_THREADS == ST_SEGS && ST_WAVES == 4,
              "tile shape");
static_assert(ST_COLS + 16 <= 8 * ST_THREADS && ST_QUAL == ST_COLS, "byte windows: one 8-byte word per thread");
static_assert(ST_FWORDS <= 64 && ST_SEGS / 32 <= 64, "head words / read marks fit one wave");

// per segment: {kappa = wbg / wobs, lw = log(wobs)}; a segment with wobs = 0: {+inf, wbg}
struct alignas(16) StSegKL {
    double kappa, lw;
};
// Per-segment column bound and index shifts, tile-local, so that a column c of the segment needs one compare or add
// for each: it is scored when c < cend; its read base is rseq_s[c + rshift] (Q4: the read bases are taken from the
// read start, so rshift = read column 0 - segment column 0; a tileable read has |algnseq| = its column count, so the
// index stays inside the read); its quality is qual_s[c + qshift] while c < qend (Q5: zero beyond the read's quality
// string).
struct alignas(8) StSegGeo {
    uint16_t cend_bep; // first column past the segment | use_bep << 15
    int16_t rshift;
    int16_t qshift;
    uint16_t qend;
};
struct alignas(16) StLom { // per (error-rate index, base match): log(om), 1 / om
    double lom, iom;
};
struct alignas(32) StRead { // per read of the tile
    double omp, lp, ip; // 1 - p_inc, its log (log(1 - bep) for a consensus FASTA), its reciprocal
    uint32_t a_ql;      // |algnseq| columns | quality string length << 16
    uint32_t col_q;     // first column | first quality byte << 16, both relative to the tile's
};
constexpr double ST_RHO_MAX = 0.015625; // 2^-6: the series' next term rho^9 / 9 is below 2^-57 relative to log1p's first

__device__ const LogTabEntry hc_log_table[64] = {VGAN_LOG_TABLE};

#ifdef VGAN_PHASE_TIMING // developer aid: cycles per phase of the tile kernel (thread 0 of every workgroup)
__device__ unsigned long long hc_phase_cycles[48]; // [wave][slot]
#define PT_MARK(slot)                                                  \
    do {                                                               \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        pt_acc[slot] += now_ - pt_last;                                \
        pt_last = now_;                                                \
    } while (0)
#else
#define PT_MARK(slot)
#endif

struct StTile { // extents of one tile (wave uniform)
    uint32_t n, seg_base, n_seg, col_base, n_col, q_base, n_q, short_qual;
};

// a byte window [g0, g0+n) of src as aligned 8-byte words: word i of the window goes to word i of the LDS array, so that
// lds[i + (g0 & 7)] = src[g0 + i].  One load per thread (162 of the 256 threads for a full tile).
struct StWindow {
    uint2 v;
};
__device__ __forceinline__ StWindow window_request(const uint8_t *__restrict__ src, uint32_t g0, uint32_t n, int tid) {
    const uint32_t a0 = g0 & ~7u;
    const uint32_t nd = (g0 + n - a0 + 7u) >> 3;
    const uint2 *__restrict__ s64 = reinterpret_cast<const uint2 *>(src + a0);
    StWindow w;
    w.v = (uint32_t)tid < nd ? s64[tid] : uint2{0u, 0u};
    return w;
}
__device__ __forceinline__ void window_store(uint8_t *dst, const StWindow &w, int tid) {
    if (tid < (ST_COLS + 16) / 8) reinterpret_cast<uint2 *>(dst)[tid] = w.v; // words past the window carry zeros
}

struct StLoads { // one tile's HBM data in flight
    StWindow gseq, rseq, qual;
    uint32_t start[ST_SEG_ITERS], len[ST_SEG_ITERS], node[ST_SEG_ITERS];
    double mapp[ST_SEG_ITERS], ln_w[ST_SEG_ITERS], inv_mm[ST_SEG_ITERS]; // of the node (HcNodeDev)
};

// wave64 inclusive prefix sum of a 32-bit count (DPP, no LDS traffic)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
}
// A C T G -> 0 8 16 24 (byte offsets into the background table), anything else 32
__device__ __forceinline__ uint32_t base_code8(uint32_t b) {
    const uint32_t d = b - 65u;
    const bool ok = d < 20u && ((0x80045u >> d) & 1u);
    return ok ? ((b << 2) & 24u) : 32u;
}

// a * b + c as the three-address v_fma_f64: with a constant c the compiler prefers a copy of c + v_fmac_f64 (two instructions)
__device__ __forceinline__ double fma3(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int ST_READS>
__global__ __launch_bounds__(ST_THREADS, 4) void hc_segment_tile_kernel(HcGraphDev g, HcBatchDev b, HcParamsDev prm,
                                                                      uint32_t n_tileable, uint32_t reads_per_block,
                                                                      double *__restrict__ segD_out,
                                                                      double *__restrict__ nodeW,
                                                                      double *__restrict__ totals) {
    __shared__ StLom lom_s[101][2];   // [qscore index, 100 = background error rate][mismatch, match]
    __shared__ double bg_s[5];
    // log p_err of a quality byte is log(10^(-Q/10)) = -Q ln(10)/10 for Q > 2 and log(0.25) otherwise (src/miscfunc.h:180-188 on
    // int(char)): its prefix sums are kept as INTEGERS, sum of the Q above 2 in bits 11.. and the count of the others in bits
    // 0..10 (1280 bytes of at most 127: 18 + 11 bits), wave-local; U_m is one multiply-add per segment on exact differences.
    __shared__ uint32_t ps_s[ST_QUAL + 1];
    __shared__ uint32_t wsum_s[ST_WAVES]; // each wave's total
    __shared__ double segS_s[ST_SEGS];
    __shared__ StSegKL segkl_s[ST_SEGS];
    __shared__ StSegGeo seggeo_s[ST_SEGS];
    __shared__ uint32_t flags_s[ST_FWORDS]; // bit c: a segment starts at tile column c
    __shared__ __attribute__((aligned(16))) uint8_t gseq_s[ST_COLS + 16];
    __shared__ __attribute__((aligned(16))) uint8_t rseq_s[ST_COLS + 16];
    __shared__ __attribute__((aligned(16))) uint8_t qual_s[ST_QUAL + 16];
    // per-read header, double buffered: the next tile's is written while this tile's is in use
    __shared__ uint32_t off_s[2][3][ST_READS + 1];
    __shared__ StRead rd_s[2][ST_READS];
    // which read a segment belongs to: .x = marks at the last segment of every read but the tile's last, 32 segments per
    // word; .y = marks in the words below.  Segment ls is of read  popcount(x & bits below ls) + y.
    __shared__ uint2 rmark_s[2][ST_SEGS / 32];
    __shared__ uint32_t first90_s[ST_READS];
    __shared__ StTile tile_s[2];
    __shared__ uint32_t tilebits_s[4]; // [0] bit 0: a segment takes the background error rate on its own; [1] lowest node id
                                       // (when the window is to be placed); [2] a quality byte >= 90 in the tile;
                                       // [3] the tile left the window
    constexpr int WIN = ST_READS <= 8 ? ST_WIN : ST_WIN_SHORT; // (the 24-read variant holds more header: a smaller window keeps it at 4 workgroups per CU)
    __shared__ double win_s[WIN];      // W[winbase .. winbase + WIN) of this workgroup's reads

    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double sumT = 0.0, sumU = 0.0; // sum of the column terms (= sum of S_m) and of U_m, each without cancellation
    uint32_t winbase = 0xFFFFFFFFu; // no window yet (workgroup uniform)
    bool need_min = true;           // the next tile places the window (workgroup uniform)

    const uint32_t rb0 = blockIdx.x * reads_per_block;
    const uint32_t rb1 = min(n_tileable, rb0 + reads_per_block);
    if (rb0 >= rb1) return;

    // Header values travel in registers of lanes 0..ST_READS of the LAST wave until they are published in LDS: that wave has
    // the fewest column chunks in phase D (5 5 5 4 for a full tile) and publishes in the time the others still compute.
    constexpr int HW = ST_WAVES - 1;
    uint32_t h_seg = 0, h_col = 0, h_q = 0, h_A = 0, h_mapq = 0;
    double h_omp = 0.0, h_lp = 0.0, h_ip = 0.0;
    auto header_request = [&](uint32_t first) {
        if (wave == HW && lane <= ST_READS) {
            const uint32_t r = min(first + (uint32_t)lane, rb1);
            h_seg = b.read_seg_off[r];
            h_col = b.read_col_off[r];
            h_q = b.read_qual_off[r];
            const uint32_t rr = min(r, b.n_reads - 1);
            h_A = b.read_algn_len[rr];
            h_mapq = b.read_mapq[rr];
        }
    };
    // the read's share of pcm by mapping quality: dependent loads, issued a phase after the request so that nothing waits
    auto header_resolve = [&]() {
        if (wave == HW && lane <= ST_READS) {
            const double *t = g.rdtab + 3u * min(h_mapq, 99u);
            h_omp = t[0];
            h_lp = t[1];
            h_ip = t[2];
        }
    };
    // Wave 0 publishes the header and the tile's extents: reads [first, first+n) with n the largest count whose
    // segments, columns and quality bytes fit the LDS tile (one read always fits: the host selects this kernel only
    // for reads within the per-read limits).  The offsets ascend, so "read t-1 still fits" is a prefix property and
    // n is a popcount.
    auto header_publish = [&](uint32_t buf, uint32_t first) {
        if (wave == HW) {
            const uint32_t sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)h_seg);
            const uint32_t cb = (uint32_t)__builtin_amdgcn_readfirstlane((int)h_col);
            const uint32_t qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)h_q);
            const bool fits = lane >= 1 && lane <= ST_READS && first + (uint32_t)lane <= rb1 && h_seg - sb <= (uint32_t)ST_SEGS &&
                              h_col - cb <= (uint32_t)ST_COLS && h_q - qb <= (uint32_t)ST_QUAL;
            const uint32_t n = max(1u, (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(fits)));
            // a read whose quality string is shorter than its |algnseq| columns needs the column phase's bound check (Q5)
            const uint32_t ql_next = (uint32_t)__shfl_down((int)h_q, 1, 64) - h_q;
            const bool shortq = (uint32_t)lane < n && ql_next < h_A;
            const uint32_t any_short = __builtin_amdgcn_ballot_w64(shortq) != 0 ? 1u : 0u;
            if (lane <= ST_READS) {
                off_s[buf][0][lane] = h_seg;
                off_s[buf][1][lane] = h_col;
                off_s[buf][2][lane] = h_q;
                if (lane < ST_READS)
                    rd_s[buf][lane] = StRead{h_omp, h_lp, h_ip, (uint32_t)(h_A | (min(ql_next, 0xFFFFu) << 16)),
                                             (uint32_t)(min(h_col - cb, (uint32_t)ST_COLS) | (min(h_q - qb, (uint32_t)ST_QUAL) << 16))};
            }
            // the read marks (LDS operations of one wave complete in order: zero, mark, count)
            if (lane < ST_SEGS / 32) rmark_s[buf][lane].x = 0u;
            if (lane >= 1 && (uint32_t)lane < n) {
                const uint32_t pos = min(h_seg - sb, (uint32_t)ST_SEGS) - 1u; // the previous read's last segment
                if (pos < (uint32_t)ST_SEGS) atomicOr(&rmark_s[buf][pos >> 5].x, 1u << (pos & 31u));
            }
            {
                const uint32_t pc = lane < ST_SEGS / 32 ? (uint32_t)__builtin_popcount(rmark_s[buf][lane & (ST_SEGS / 32 - 1)].x) : 0u;
                const uint32_t below = wave_incl_scan_u32(pc) - pc;
                if (lane < ST_SEGS / 32) rmark_s[buf][lane].y = below;
            }
            // clamped: a read that breaks the tile contract (a caller's error) must not index past the LDS arrays
            if ((uint32_t)lane == n)
                tile_s[buf] = StTile{n, sb, (uint32_t)min(h_seg - sb, (uint32_t)ST_SEGS), cb,
                                     (uint32_t)min(h_col - cb, (uint32_t)ST_COLS), qb, (uint32_t)min(h_q - qb, (uint32_t)ST_QUAL),
                                     any_short};
        }
    };
    auto tile_extents = [&](uint32_t buf) { return tile_s[buf]; };
    auto tile_request = [&](const StTile &t, StLoads &L) {
#pragma unroll
        for (int it = 0; it < ST_SEG_ITERS; ++it) {
            const uint32_t ls = tid + it * ST_THREADS;
            const uint32_t s = t.seg_base + min(ls, t.n_seg ? t.n_seg - 1 : 0u);
            const bool on = ls < t.n_seg;
            L.node[it] = on ? b.seg_node[s] : 0u;
            L.start[it] = on ? b.seg_start[s] : 0u;
            L.len[it] = on ? b.seg_len[s] : 0u;
        }
        L.gseq = window_request(b.graph_seq, t.col_base, t.n_col, tid);
        L.rseq = window_request(b.algnseq, t.col_base, t.n_col, tid);
        L.qual = window_request(b.qual, t.q_base, t.n_q, tid);
    };
    auto tile_gather = [&](const StTile &t, StLoads &L) { // needs L.node: issued well after tile_request
#pragma unroll
        for (int it = 0; it < ST_SEG_ITERS; ++it) {
            const uint32_t ls = tid + it * ST_THREADS;
            const HcNodeDev *nd = g.node_tab + (ls < t.n_seg ? L.node[it] : 0u);
            const double2 lw_inv = *reinterpret_cast<const double2 *>(&nd->ln_w); // one 16-byte load
            L.ln_w[it] = lw_inv.x;
            L.inv_mm[it] = lw_inv.y;
            L.mapp[it] = nd->mappability;
        }
    };
    // the window's content goes to W in HBM (the slots stay zero otherwise: no atomic for them)
    auto window_flush = [&]() {
        for (uint32_t j = tid; j < (uint32_t)WIN; j += ST_THREADS) {
            const double v = win_s[j];
            if (v != 0.0) {
                unsafeAtomicAdd(&nodeW[winbase + j], v);
                win_s[j] = 0.0;
            }
        }
    };

#ifdef VGAN_PHASE_TIMING
    unsigned long long pt_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pt_last = __builtin_readcyclecounter();
#endif
    // ---- prologue: the first tile's header and data, the second tile's header
    uint32_t r0 = rb0, cur = 0;
    header_request(r0);
    // the workgroup's tables, built while the first header is on its way
    for (int i = tid; i < 202; i += ST_THREADS) {
        const int qi = i >> 1;
        const double e = (qi == 100 || prm.use_bep) ? prm.bep : g.qscore[qi];
        const double om = (i & 1) ? 1.0 - e : 1.0 - (1.0 - e); // (a mismatch: 1 - (1 - e) in double, get_p_obs_base.cpp:21,67)
        lom_s[qi][i & 1] = StLom{log_pos(om), 1.0 / om};
    }
    for (int i = tid; i < WIN; i += ST_THREADS) win_s[i] = 0.0;
    if (tid < 5) bg_s[tid] = tid == 0 ? 0.27532 : tid == 1 ? 0.30044 : tid == 2 ? 0.25780 : tid == 3 ? 0.16644 : 0.25; // A C T G by (c>>1)&3
    header_resolve();
    __syncthreads(); // the tables are in place
    header_publish(cur, r0);
    __syncthreads();
    StTile T = tile_extents(cur);
    StLoads L;
    tile_request(T, L);
    tile_gather(T, L);
    if (r0 + T.n < rb1) {
        header_request(r0 + T.n);
        header_resolve();
        header_publish(cur ^ 1u, r0 + T.n); // (the first tile's barriers come before anyone reads it)
    }
    while (true) {
        // per-thread addresses are cheap to rebuild and expensive to keep: nothing derived from the thread id is to be
        // hoisted out of the tile loop (the register allocator would spill it)
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint32_t lanemask = 0xFFFFFFFFu >> (31u - ((uint32_t)tid & 31u)); // head bits at or below this lane's column
        const bool has_next = r0 + T.n < rb1; // wave (and workgroup) uniform
        const uint32_t cshift = T.col_base & 7u, qshift = T.q_base & 7u;
        // ---- top: byte windows into LDS, per-tile state reset
        window_store(gseq_s, L.gseq, tid);
        window_store(rseq_s, L.rseq, tid);
        window_store(qual_s, L.qual, tid);
        if (tid < ST_FWORDS) flags_s[tid] = 0u;
        if (tid < ST_READS) first90_s[tid] = 0xFFFFFFFFu;
        if (tid < 3) tilebits_s[tid] = tid == 1 ? 0xFFFFFFFFu : 0u;
        PT_MARK(0);
        __syncthreads();
        PT_MARK(1);

        // ---- B: prefix sums of log p_err over the tile's quality bytes.  No bounds checks: slots past the tile's
        // T.n_q bytes receive sums nobody reads (prefix values at or below T.n_q do not depend on what follows).
        {
            const uint32_t i0 = tid * ST_QB;
            uint32_t qb[ST_QB];
            bool hot = false;
            { // the lane's 5 bytes lie in two consecutive dwords: one LDS read instead of five
                static_assert(ST_QB == 5, "two dwords hold the lane's bytes");
                const uint32_t o = qshift + i0;
                const uint32_t *w = reinterpret_cast<const uint32_t *>(qual_s + (o & ~3u)); // (ds_read2_b32)
                const uint64_t bytes = (((uint64_t)w[1] << 32) | w[0]) >> (8u * (o & 3u));
#pragma unroll
                for (int e = 0; e < ST_QB; ++e) { // one rare branch for the lot
                    qb[e] = (uint32_t)(bytes >> (8 * e)) & 0xFFu;
                    hot |= (int)(int8_t)qb[e] >= 90;
                }
            }
            if (hot) { // rare: Q >= 90 switches the read to the background error rate
                tilebits_s[2] = 1u; // (phase C looks at first90_s only then)
                for (int e = 0; e < ST_QB; ++e) {
                    const uint32_t i = i0 + e, gq = T.q_base + i;
                    if ((int)(int8_t)qb[e] >= 90 && i < T.n_q) {
                        uint32_t k = 0;
                        for (uint32_t t = 1; t < T.n; ++t) k += gq >= off_s[cur][2][t] ? 1u : 0u;
                        atomicMin(&first90_s[k], gq - off_s[cur][2][k]);
                    }
                }
            }
            uint32_t loc[ST_QB];
            uint32_t run = 0u;
#pragma unroll
            for (int e = 0; e < ST_QB; ++e) {
                const int Q = (int)(int8_t)qb[e];
                run += Q > 2 ? (uint32_t)Q << 11 : 1u;
                loc[e] = run;
            }
            const uint32_t incl = wave_incl_scan_u32(run);
            const uint32_t before = incl - run;
#pragma unroll
            for (int e = 0; e < ST_QB; ++e) ps_s[i0 + e + 1] = before + loc[e];
            if (lane == 63) wsum_s[wave] = incl;
            if (tid == 0) ps_s[0] = 0u;
            need_min = need_min || tilebits_s[3] != 0u;
        }
        PT_MARK(2);
        __syncthreads();
        PT_MARK(3);

        // ---- C: one lane per segment
        double segU[ST_SEG_ITERS];
        {
            const uint32_t ws0 = wsum_s[0], ws1 = wsum_s[1], ws2 = wsum_s[2];
            const bool any_hot = __builtin_amdgcn_readfirstlane((int)tilebits_s[2]) != 0; // a quality >= 90 somewhere in the tile
            uint32_t nmin = 0xFFFFFFFFu;
            bool own_bep = false;
            if (tid == 0) tilebits_s[3] = 0u;
#pragma unroll
            for (int it = 0; it < ST_SEG_ITERS; ++it) {
                const uint32_t ls = tid + it * ST_THREADS;
                segU[it] = 0.0;
                if (ls < T.n_seg) {
                    // read of the segment: the marks below it
                    const uint2 mk = rmark_s[cur][ls >> 5];
                    const uint32_t k = min((uint32_t)__builtin_popcount(mk.x & (lanemask >> 1)) + mk.y, (uint32_t)ST_READS - 1u);
                    const StRead rd = rd_s[cur][k];
                    const uint32_t A = rd.a_ql & 0xFFFFu, QL = rd.a_ql >> 16;
                    const uint32_t colbase = rd.col_q & 0xFFFFu, qoff = rd.col_q >> 16;
                    const uint32_t start = L.start[it], len = L.len[it];
                    const uint32_t lo = min(start, QL), hi = min(start + A, QL);
                    const uint32_t ilo = qoff + lo, ihi = qoff + hi;
                    uint32_t pk = ps_s[ihi] - ps_s[ilo]; // (wave-local sums: whole waves in between are added back)
                    pk += (ilo <= 1u * ST_QW && ihi > 1u * ST_QW) ? ws0 : 0u;
                    pk += (ilo <= 2u * ST_QW && ihi > 2u * ST_QW) ? ws1 : 0u;
                    pk += (ilo <= 3u * ST_QW && ihi > 3u * ST_QW) ? ws2 : 0u;
                    // Q5: the bytes beyond the quality string count as Q = 0, i.e. among the "others"
                    const uint32_t n_low = (pk & 2047u) + (A - (hi - lo));
                    const double U = fma((double)(pk >> 11), -0.23025850929940457 /* ln(10) / 10 */, (double)n_low * -1.3862943611198906 /* log(0.25) */);
                    segU[it] = U;
                    sumU += U;
                    const bool sticky = any_hot && first90_s[k] < hi; // update_likelihood.cpp:42
                    own_bep |= sticky && !prm.use_bep;
                    const uint32_t use_bep = (prm.use_bep || sticky) ? 1u : 0u;
                    const uint32_t cs = colbase + start;
                    const uint32_t cl = cs < T.n_col ? min(len, T.n_col - cs) : 0u;
                    // wbg = 1 - pcm, wobs = pcm * match (process_mapping.cpp:41,66-75; a consensus FASTA: 0 and (1 - bep) * match)
                    const double pcm = rd.omp * L.mapp[it];
                    const double wbg = prm.consensus ? 0.0 : 1.0 - pcm;
                    double kappa = prm.consensus ? 0.0 : wbg * (rd.ip * L.inv_mm[it]);
                    double lw = rd.lp + L.ln_w[it];
                    if (!(kappa < 1e300)) { // wobs = 0 (mapping quality 0, mappability 0): the column is log(wbg * bg)
                        kappa = INFINITY;
                        lw = wbg;
                    }
                    segkl_s[ls] = StSegKL{kappa, lw};
                    seggeo_s[ls] = StSegGeo{(uint16_t)((cs + cl) | (use_bep << 15)), (int16_t)((int)colbase - (int)cs + (int)cshift),
                                            (int16_t)((int)qoff - (int)colbase + (int)qshift), (uint16_t)(colbase + QL)};
                    segS_s[ls] = 0.0;
                    if (cl) atomicOr(&flags_s[cs >> 5], 1u << (cs & 31u));
                    nmin = min(nmin, L.node[it]);
                }
            }
            if (nodeW && need_min) { // the window is (re)placed at this tile's lowest node id
                nmin = wave_min_u32(nmin);
                if (lane == 0) atomicMin(&tilebits_s[1], nmin);
            }
            if (__builtin_amdgcn_ballot_w64(own_bep) != 0 && lane == 0) atomicOr(&tilebits_s[0], 1u);
        }
        PT_MARK(4);
        __syncthreads();
        PT_MARK(5);

        // the next tile's data leaves HBM now and lands during phase D
        StTile Tn = T;
        StLoads Ln;
        if (has_next) {
            Tn = tile_extents(cur ^ 1u);
            tile_request(Tn, Ln);
            if (r0 + T.n + Tn.n < rb1) header_request(r0 + T.n + Tn.n);
        }
        PT_MARK(6);

        // ---- D: one lane per alignment column, flat over the tile; everything comes from LDS
        const uint32_t tb0 = tilebits_s[0], node_lo = tilebits_s[1];
        // Lane w keeps head word w and the number of heads in the words below it: a wave's 64 columns lie in two words, so the
        // owner of a column is four v_readlane and a masked popcount away -- no LDS round trip.
        const uint32_t flagw = lane < ST_FWORDS ? flags_s[lane] : 0u;
        uint32_t wordbase;
        {
            const uint32_t pc = (uint32_t)__builtin_popcount(flagw);
            wordbase = wave_incl_scan_u32(pc) - pc - 1u; // (- 1: the owner's index is the head count less one)
        }
        // A chunk of 64 columns per wave, stage by stage (three dependent rounds of LDS reads, then the series).  U > 1 chunks at
        // a time interleave the chains: measured -1.2 % at U = 2, but only with the series constants in scalar registers, which
        // alone costs 3.5 % (SGPRs spill to VGPR lanes); with the constants in VGPRs U = 2 spills.  So U = 1.
        auto column_group = [&](auto general_tag, auto width_tag, const uint32_t c0) {
            constexpr bool GEN = decltype(general_tag)::value;
            constexpr int U = decltype(width_tag)::value;
            uint32_t c[U], ls[U], gbyte[U];
            bool owned[U];
            uint2 geo[U];
            StSegKL kl[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t cu0 = c0 + (uint32_t)u * ST_THREADS; // wave uniform; below ST_COLS + ST_THREADS
                const int w0 = (int)(cu0 >> 5);
                // heads at or below lane l of the chunk = head bit 0 + (bits 1..l) = bit 0 + v_mbcnt of the chunk's 64 head
                // bits shifted down by one; everything but the two v_mbcnt is scalar
                const uint64_t heads = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)flagw, w0) |
                                       ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)flagw, w0 + 1) << 32);
                const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)wordbase, w0) + (uint32_t)(heads & 1u);
                const uint64_t above0 = heads >> 1;
                c[u] = cu0 + (uint32_t)lane;
                const uint32_t own = __builtin_amdgcn_mbcnt_hi((uint32_t)(above0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)above0, base));
                owned[u] = (int)own >= 0; // columns no mapping scores (Q6 tail) have no owner
                ls[u] = own & ((uint32_t)ST_SEGS - 1u);
                // every LDS read is unconditional (indices in range, results selected afterwards)
                gbyte[u] = gseq_s[min(c[u] + cshift, (uint32_t)ST_COLS + 15u)];
                geo[u] = *reinterpret_cast<const uint2 *>(&seggeo_s[ls[u]]);
                kl[u] = segkl_s[ls[u]];
            }
            uint32_t rbyte[U];
            int q[U];
            bool in_seg[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t cend = geo[u].x & 0x7FFFu;
                const int rshift = (int)geo[u].x >> 16, qsh = (int)(int16_t)(geo[u].y & 0xFFFFu);
                in_seg[u] = owned[u] && c[u] < cend;
                const uint32_t ri = min((uint32_t)((int)c[u] + rshift), (uint32_t)ST_COLS + 7u);
                const uint32_t qi = min((uint32_t)((int)c[u] + qsh), (uint32_t)ST_QUAL + 7u);
                rbyte[u] = rseq_s[ri];
                q[u] = (int)(int8_t)qual_s[qi];
            }
            uint32_t match[U];
            bool valid[U];
            StLom lo[U];
            double bgv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t gcode = base_code8(gbyte[u]), rcode = base_code8(rbyte[u]);
                if constexpr (GEN) q[u] = c[u] < (geo[u].y >> 16) ? q[u] : 0; // Q5: zero beyond the quality string
                q[u] = q[u] < 0 ? 0 : (q[u] > 99 ? 99 : q[u]);                  // qscore_vec's index
                if constexpr (GEN) q[u] = (geo[u].x & 0x8000u) ? 100 : q[u];    // slot 100 holds the background error rate
                valid[u] = in_seg[u] && (gcode | rcode) < 32u; // process_mapping.cpp:62-63
                match[u] = gcode == rcode ? 1u : 0u;           // 1 - eps: get_p_obs_base.cpp:3-27, :67 with tv = ts = 0
                lo[u] = lom_s[q[u]][match[u]];
                bgv[u] = *reinterpret_cast<const double *>(reinterpret_cast<const uint8_t *>(bg_s) + rcode);
            }
            double t[U];
            bool rare = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double rho = kl[u].kappa * bgv[u] * lo[u].iom;
                double p = fma3(rho, 1.0 / 8.0, -1.0 / 7.0);
                p = fma3(rho, p, 1.0 / 6.0);
                p = fma3(rho, p, -0.2);
                p = fma3(rho, p, 0.25);
                p = fma3(rho, p, -1.0 / 3.0);
                p = fma(rho, p, 0.5);
                p = fma(rho, -p, 1.0);
                t[u] = fma(rho, p, lo[u].lom + kl[u].lw);
                rare |= valid[u] && !(rho < ST_RHO_MAX);
            }
            if (__builtin_expect(rare, 0)) { // see the head of this section
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double rho = kl[u].kappa * bgv[u] * lo[u].iom;
                    if (valid[u] && !(rho < ST_RHO_MAX)) {
                        const double e = (q[u] == 100 || prm.use_bep) ? prm.bep : g.qscore[q[u]];
                        const double om = match[u] ? 1.0 - e : 1.0 - (1.0 - e);
                        const bool deg = !(kl[u].kappa < 1e300); // wobs = 0: {inf, wbg}
                        // (the table log by hand: log_tab()'s series for values outside the normal range would park its
                        // constants in scratch for the whole kernel)
                        double x = deg ? kl[u].lw * bgv[u] : fma(kl[u].kappa, bgv[u], om);
                        double adj = 0.0;
                        if (x < 2.2250738585072014e-308 && x > 0.0) { // subnormal
                            x *= 18014398509481984.0; // 2^54
                            adj = -37.429947750237048; // -54 ln 2
                        }
                        const double lx = x > 0.0 ? (x <= 1.7976931348623157e308 ? log_tab_eval(x, hc_log_table) + adj : x)
                                                  : (x == 0.0 ? -INFINITY : __builtin_nan(""));
                        t[u] = deg ? lx : kl[u].lw + lx;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (valid[u]) unsafeAtomicAdd(&segS_s[ls[u]], t[u]);
        };
        // (whole waves: the owner lookup reads other lanes' registers, and every index stays inside its array for the
        // columns past the tile's end, which no segment claims)
        auto columns = [&](auto general_tag) {
            uint32_t c0 = (uint32_t)wave * 64u;
            for (; c0 < T.n_col; c0 += ST_THREADS) column_group(general_tag, std::integral_constant<int, 1>{}, c0);
        };
        if ((tb0 | T.short_qual) != 0u) columns(std::true_type{});
        else columns(std::false_type{});
        PT_MARK(7);
        if (has_next) {
            tile_gather(Tn, Ln); // the node ids arrived during D,
            // ... and so did the header of the tile after the next, requested before D: it takes this tile's header slots, which
            // nobody reads after phase C (the last wave gets here first, so its dependent table loads cost nobody a wait)
            if (r0 + T.n + Tn.n < rb1) {
                header_resolve();
                header_publish(cur, r0 + T.n + Tn.n);
            }
        }
        PT_MARK(8);
        __syncthreads();
        PT_MARK(9);

        // ---- E: one lane per segment
        if (nodeW && need_min && T.n_seg) {
            // The window sits at the lowest node id of the workgroup's first tile: the batch is sorted by the reads' lowest
            // node id (vgan_hc_flatten), so a workgroup's reads stay above it and move through the node ids slowly.  Any
            // other order is still correct -- a segment outside the window adds to W in HBM directly -- and a tile that
            // leaves the window with many segments has the next one place it anew.
            if (winbase != 0xFFFFFFFFu) {
                window_flush();
                __syncthreads();
            }
            winbase = node_lo;
            need_min = false;
        }
        uint32_t n_outside = 0;
#pragma unroll
        for (int it = 0; it < ST_SEG_ITERS; ++it) {
            const uint32_t ls = tid + it * ST_THREADS;
            if (ls < T.n_seg) {
                const double S = segS_s[ls];
                sumT += S;
                const double D = S - segU[it];
                if (segD_out) segD_out[T.seg_base + ls] = D;
                if (nodeW) {
                    const uint32_t slot = L.node[it] - winbase;
                    if (slot < (uint32_t)WIN) {
                        unsafeAtomicAdd(&win_s[slot], D);
                    } else {
                        unsafeAtomicAdd(&nodeW[L.node[it]], D);
                        n_outside++;
                    }
                }
            }
        }
        if (nodeW && __builtin_popcountll(__builtin_amdgcn_ballot_w64(n_outside > 0)) > 16 && lane == 0) atomicOr(&tilebits_s[3], 1u);
        PT_MARK(10);
        if (!has_next) break;
        // the next tile's barriers order E against its C (E reads segS_s only)
        r0 += T.n;
        T = Tn;
        L = Ln;
        cur ^= 1u;
    }
    if (nodeW && winbase != 0xFFFFFFFFu) {
        __syncthreads();
        window_flush();
    }
#ifdef VGAN_PHASE_TIMING
    if (lane == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(&hc_phase_cycles[wave * 12 + i], pt_acc[i]);
#endif
    sumT = wave_sum(sumT);
    sumU = wave_sum(sumU);
    if (lane == 0 && totals) {
        double *t = totals + ((blockIdx.x * ST_WAVES + wave) % HC_TOTAL_SLOTS) * HC_TOTAL_STRIDE;
        unsafeAtomicAdd(&t[0], sumT);
        unsafeAtomicAdd(&t[1], sumU);
    }
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
    if (tw == 0) return; // the whole workgroup: the tile is uniform in it
    if (!IDENT && i0_64 >= n_items) return;
    // (the node pass keeps an item-less wave: it takes part in the workgroup's final reduction with zeros)
    const uint32_t i0 = (uint32_t)min((uint64_t)n_items, i0_64);
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
    if constexpr (IDENT) {
        // the node pass is short and ends in atomics on the same P accumulators from every wave: same-address device
        // atomics serialise (~0.1 us each), so the four waves of a workgroup are summed in LDS first
        __shared__ double red_s[3][TB + 1][64];
        if (wave > 0) {
#pragma unroll
            for (int k = 0; k <= TB; ++k) red_s[wave - 1][k][lane] = acc[k];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k <= TB; ++k) {
                const double v = ((acc[k] + red_s[0][k][lane]) + red_s[1][k][lane]) + red_s[2][k][lane];
                if ((uint32_t)k < tw && v != 0.0) unsafeAtomicAdd(&acc_out[(size_t)(w0 + k) * 64 + lane], v);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k <= TB; ++k) {
            if ((uint32_t)k < tw && acc[k] != 0.0) unsafeAtomicAdd(&acc_out[(size_t)(w0 + k) * 64 + lane], acc[k]);
        }
    }
}

// final[p] = Stot - acc[p]
// (acc_node is scratch of one finalize: the kernel leaves it zero for the next, so finalize needs no memset of its own; out2,
// when given, receives a second copy -- the caller's device buffer -- instead of a copy engine pass behind the kernel)
__global__ void hc_finish_kernel(const double *__restrict__ totals, const double *__restrict__ acc_seg,
                                 double *__restrict__ acc_node, uint32_t n_paths, uint32_t n_slots, double *__restrict__ out,
                                 double *__restrict__ out2) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_slots) return;
    const double an = acc_node[p];
    acc_node[p] = 0.0;
    if (p < n_paths) {
        double stot = 0.0; // (the partial sums of hc_device.h, in slot order: the same value in every thread)
        for (uint32_t s = 0; s < HC_TOTAL_SLOTS; ++s) stot += totals[s * HC_TOTAL_STRIDE];
        const double v = stot - (acc_seg[p] + an);
        out[p] = v;
        if (out2) out2[p] = v;
    }
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
// One block per list: conf[list] = exp(LSE(final[idx[i]] : i in list) - LSE(final[all])).
// A list is the reference's all_top vector (src/get_posterior.cpp:51-76): the path indices of every recursion level in
// turn, so a path reachable at two depths of children.txt appears -- and is counted -- twice.
// libgab's oplusInitnatl treats a running value of exactly 0 as "empty" (SURVEY.md Q11): zeros in front of
// the first non-zero term (in list order) are skipped, both here and in the reference's sequential fold.
__device__ double block_lse(const double *__restrict__ v, const uint32_t *__restrict__ idx, uint32_t n, double *sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    uint32_t first = 0xFFFFFFFFu;
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        if (v[idx ? idx[i] : i] != 0.0) first = min(first, i);
    }
    first = wave_min_u32(first);
    __shared__ uint32_t shu[16];
    if (lane == 0) shu[wave] = first;
    __syncthreads();
    first = 0xFFFFFFFFu;
    for (int w = 0; w < nw; ++w) first = min(first, shu[w]);
    __syncthreads();
    if (first == 0xFFFFFFFFu) return 0.0; // all members are exactly 0, or the list is empty (oracle.h: sum of nothing = 0)
    double mx = -INFINITY;
    for (uint32_t i = first + tid; i < n; i += blockDim.x) mx = fmax(mx, v[idx ? idx[i] : i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) sh[wave] = mx;
    __syncthreads();
    mx = -INFINITY;
    for (int w = 0; w < nw; ++w) mx = fmax(mx, sh[w]);
    __syncthreads();
    double s = 0.0;
    for (uint32_t i = first + tid; i < n; i += blockDim.x) s += exp(v[idx ? idx[i] : i] - mx);
    s = wave_sum(s);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    s = 0.0;
    for (int w = 0; w < nw; ++w) s += sh[w];
    __syncthreads();
    return mx + log(s);
}

__global__ __launch_bounds__(256) void hc_posterior_kernel(const double *__restrict__ final_vec, uint32_t n_paths,
                                                            const uint32_t *__restrict__ list_off,
                                                            const uint32_t *__restrict__ list_idx,
                                                            double *__restrict__ conf) {
    __shared__ double sh[16];
    const double total = block_lse(final_vec, nullptr, n_paths, sh);
    const uint32_t o0 = list_off[blockIdx.x], o1 = list_off[blockIdx.x + 1];
    const double part = block_lse(final_vec, list_idx + o0, o1 - o0, sh);
    if (threadIdx.x == 0) conf[blockIdx.x] = exp(part - total);
}

#ifdef VGAN_PHASE_TIMING
extern "C" int vgan_hc_debug_phase_cycles(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hc_phase_cycles), sizeof(hc_phase_cycles)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[48] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(hc_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// ---------------------------------------------------------------------------------------------- launchers
void launch_hc_segments(const HcGraphDev &g, const HcBatchDev &b, const HcParamsDev &prm, uint32_t n_tileable,
                        uint32_t mean_cols_per_read, double *segS, double *segU, double *segD, double *nodeW, double *totals,
                        hipStream_t st) {
    if (b.n_reads == 0) return;
    uint32_t nt = std::min(n_tileable, b.n_reads);
    if (segS || segU) nt = 0; // the tiled kernel produces D_m only; separate S_m / U_m (debug API) come from the general one
    if (nt) {
        // ~6 workgroups per CU and launch round; contiguous read ranges per workgroup
        // 16 workgroups per CU: contiguous read ranges (the LDS window wants them), short enough for the dispatcher to even
        // out the CUs (1024 persistent workgroups: 1.20 ms per 1M x 150 bp; 3072: 1.01; 4096: 1.00; 8192: 1.08)
        const uint32_t want_blocks = 256u * 16u;
        uint32_t per = (nt + want_blocks - 1) / want_blocks;
        per = std::max(per, (uint32_t)ST_READS_SHORT);
        const uint32_t blocks = (nt + per - 1) / per;
        // short reads: more of them per tile, or the tile's lanes idle (the estimate uses all reads of the batch)
        if (mean_cols_per_read < 110u)
            hipLaunchKernelGGL(hc_segment_tile_kernel<ST_READS_SHORT>, dim3(blocks), dim3(ST_THREADS), 0, st, g, b, prm, nt, per, segD, nodeW, totals);
        else
            hipLaunchKernelGGL(hc_segment_tile_kernel<ST_READS_LONG>, dim3(blocks), dim3(ST_THREADS), 0, st, g, b, prm, nt, per, segD, nodeW, totals);
    }
    launch_hc_segments_general(g, b, prm, nt, segS, segU, segD, nodeW, totals, st);
}

void launch_hc_segments_general(const HcGraphDev &g, const HcBatchDev &b, const HcParamsDev &prm, uint32_t r_begin, double *segS,
                                double *segU, double *segD, double *nodeW, double *totals, hipStream_t st) {
    if (r_begin >= b.n_reads) return;
    const uint32_t rest = b.n_reads - r_begin;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)rest + SEG_WAVES - 1) / SEG_WAVES, 256u * 8u);
    hipLaunchKernelGGL(hc_segment_general_kernel, dim3(blocks), dim3(SEG_WAVES * 64), 0, st, g, b, prm, r_begin, segS, segU, segD,
                       nodeW, totals);
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

void launch_hc_finish(const double *totals, const double *acc_seg, double *acc_node, uint32_t n_paths, uint32_t n_slots,
                      double *out, double *out2, hipStream_t st) {
    hipLaunchKernelGGL(hc_finish_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, st, totals, acc_seg, acc_node, n_paths,
                       n_slots, out, out2);
}

void launch_hc_read_loglik(const HcGraphDev &g, const HcBatchDev &b, const double *segS, const double *segU, double *out,
                           hipStream_t st) {
    if (b.n_reads == 0) return;
    hipLaunchKernelGGL(hc_read_loglik_kernel, dim3((g.n_paths + 255) / 256, b.n_reads), dim3(256), 0, st, g, b, segS,
                       segU, out);
}

void launch_hc_posterior(const double *final_vec, uint32_t n_paths, const uint32_t *list_off, const uint32_t *list_idx,
                         uint32_t n_lists, double *conf, hipStream_t st) {
    if (n_lists == 0) return;
    hipLaunchKernelGGL(hc_posterior_kernel, dim3(n_lists), dim3(256), 0, st, final_vec, n_paths, list_off, list_idx, conf);
}

} // namespace vgan
#include "module_anchor.h"
const void *vgan::anchor_hc_kernels() { return (const void *)&vgan::hc_finish_kernel; }
