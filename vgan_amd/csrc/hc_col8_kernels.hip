// HaploCart segment kernel, "eight columns to a lane" form (gfx950, wave64): W[node] += S_m - U_m and the totals for the
// packed reads of a batch (node-weights accumulation).  Same arithmetic as hc_segment_wave_kernel (hc_wave_kernels.hip;
// reference: src/process_mapping.cpp:4-91, src/update_likelihood.cpp:19-53, src/get_p_obs_base.cpp:3-69), built on what that
// kernel's counters said -- its launch is bound by the COUNT of vector and scalar instructions it issues and by LDS traffic,
// not by HBM -- so this one changes what a column and a segment cost:
//
//   * a lane owns EIGHT CONSECUTIVE alignment columns of the tile (two 16-byte loads per lane and tile instead of eight
//     4-byte ones).  The index of a column's mapping is then a running byte sum over the lane's own records (the head flag
//     is byte 3 = 4: the sum is the index times four) plus ONE wave scan per tile -- no ballot, no scalar bit counting per 64
//     columns -- and neighbouring lanes sit eight columns apart, so the fp64 LDS adds of one wave instruction hardly ever share
//     an address (a mapping's columns used to sit in neighbouring lanes);
//   * the quality prefix sums of get_log_lik_if_unsupported (process_mapping.cpp:4-24) are taken from byte 2 of those same
//     records (a lane holds eight consecutive quality bytes): no second copy of the quality strings is read;
//   * U_m stays an INTEGER pair {sum of Q above 2, count of the others} all the way: it is added into a 64-bit integer window
//     beside the fp64 one and turned into -ln(10)/10 * sum + log(1/4) * count once per window slot, at the flush
//     (exact sums, two roundings per slot instead of two per segment);
//   * the column term log(wbg * bg + wobs * om) is a pure function of {mapping quality, node class, quality, match, read base}
//     where "node class" = the node's {mappability, match probability} pair: a graph has a handful of them (get_p_obs_base.cpp:
//     44-59 knows six mutation rates; mappability is 1 nearly everywhere).  For the reads of mapping quality
//     VGAN_HC_MAPQ_MAJOR (the flatten step puts them first) every workgroup builds the table of those terms for the
//     C8_NMEMO most frequent node classes WITH THE KERNEL'S OWN col_term() -- the sums do not depend on which path a tile takes
//     -- and a tile whose mappings all fall into it costs one LDS read and one LDS add per column: no fp64 arithmetic at all.
//     Any other tile (another mapping quality, a rare node class, a quality byte outside [0, C8_QMAX), Q >= 90, a mapping
//     outside the W window, the background-error-rate modes) computes every column as the wave kernel does.
//
// One workgroup of 1024 threads per CU (the table is shared by its 16 waves; the waves share nothing else and never meet at
// a barrier after the set-up), work in units of reads from a ticket counter, everything a tile reads requested one tile ahead.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "device_math.h"
#include "hc_device.h"

namespace vgan {
namespace c8 {

#ifndef C8_ROTATE
#define C8_UNROLL2
#endif

// A variant of the kernel: columns per lane (a multiple of 8: the tile holds 64 * CPL columns), segments per tile (whole passes
// of 64), node ids covered by a wave's W window, waves of the one workgroup a CU holds (what its LDS can carry).
template <int CPL_, int CAPS_, int WIN_, int WAVES_> struct C8Cfg {
    static constexpr int CPL = CPL_, CAPC = 64 * CPL_, CAPS = CAPS_, SPASS = CAPS_ / 64, WIN = WIN_, WAVES = WAVES_, THREADS = 64 * WAVES_;
    static_assert(CPL_ % 8 == 0 && CAPS_ % 64 == 0, "whole groups and passes");
};
using C8Short = C8Cfg<8, 192, 160, 16>;  // three 150 bp reads to a tile (a read spans 106 node ids on the hcfiles graph, 140 at most)
using C8Mid = C8Cfg<16, 384, 256, 8>;    // three 300 bp reads (212 node ids)
using C8Long = C8Cfg<24, 512, 512, 4>;   // two 600 bp reads (424 node ids); a read up to the tile contract's 1280 columns / 512 segments
constexpr int C8_NR = 8;                // reads per tile at most (the segment records carry the read's index & 7)
constexpr int C8_QMAX = 48;             // quality values the table of column terms covers: [0, C8_QMAX)
constexpr int C8_NMEMO = 16;            // node classes it covers (the most frequent ones)
constexpr uint32_t C8_CLS_BYTES = C8_QMAX * 64u; // a class's part of the table: [quality][match][read base] doubles
constexpr uint32_t C8_CLS2_BYTES = HC_EXT_QMAX * 64u; // ... of the wide table (every class, every quality below HC_EXT_QMAX)
static_assert(C8_NMEMO == (int)HC_MEMO_CLASSES && C8_CLS_BYTES == HC_MEMO_CLASS_BYTES, "hc_capi.hip writes node_hi[] with these");
constexpr uint32_t C8_BUF_FLAGS = 0x00020000u;   // raw buffer descriptor, 32-bit data format (gfx9)
constexpr uint32_t C8_BLOCK = 32u;               // reads per block of the work queue
constexpr uint32_t C8_OUTSIDE = 0xFFFFu;         // info.lo: the mapping's node lies outside the W window

struct alignas(16) C8KL { // per segment of a general tile: kappa = wbg / wobs (sign set: sticky Q >= 90), lw = log(wobs)
    double kappa, lw;
};
struct alignas(16) C8Lom { // per (error-rate index, base match): log(om), 1 / om
    double lom, iom;
};

template <class K> struct C8Slice { // one wave's LDS
    static constexpr int C8_CAPC = K::CAPC, C8_CAPS = K::CAPS, C8_WIN = K::WIN;
    union {
        uint32_t ps[C8_CAPC + 8]; // ps[4 + j]: prefix through tile column j (ps[3] = 0: the empty prefix); dead behind phase C
        C8KL kl[C8_CAPS];         // a general tile's {kappa, lw} per segment (written behind phase C)
    };
    uint32_t info[C8_CAPS + 4]; // [1 + segment]: window slot byte offset (or C8_OUTSIDE) | table byte offset of the class << 16
    uint2 rdA[C8_NR];           // per read of the tile: {|algnseq| | quality length << 16, first column in the tile | mapping quality << 16 | major << 23}
    double rdB[C8_NR][4];       // general tiles: {1 - p_inc, its log, its reciprocal} of the read's mapping quality
    uint32_t first90[C8_NR];
    double win_d[C8_WIN];             // sum of the column terms per window slot
    unsigned long long win_i[C8_WIN]; // {count of Q <= 2 (low word), sum of Q above 2 (high word)} per window slot
};

template <class K> struct C8Lds {
    double memo[C8_NMEMO * C8_QMAX * 8]; // first: its byte offsets fit the 16 bits info[] gives them
    C8Lom lom[101][2];                   // [qscore index, 100 = background error rate][mismatch, match]
    double2 bg[4];                       // A C T G by (base >> 1) & 3: {frequency, frequency / 6}
    LogTabEntry logtab[64];
    HcNodeDev cls[HC_MAX_NODE_CLASSES];  // the node classes' scalars
    uint32_t ticket;                     // the workgroup's work queue: the next ticket
    uint32_t pad_[3];
    C8Slice<K> slice[K::WAVES];
};
static_assert(sizeof(C8Lds<C8Short>) <= 163840 && sizeof(C8Lds<C8Mid>) <= 163840 && sizeof(C8Lds<C8Long>) <= 163840, "one workgroup per CU: all of its LDS");

__device__ const LogTabEntry c8_log_table[64] = {VGAN_LOG_TABLE};

using lds_u32p = __attribute__((address_space(3))) uint32_t *;
using lds_u64p = __attribute__((address_space(3))) unsigned long long *;
using lds_f64p = __attribute__((address_space(3))) double *;
typedef double c8_v2d __attribute__((ext_vector_type(2)));
using lds_v2dp = __attribute__((address_space(3))) c8_v2d *;
template <class T> __device__ __forceinline__ uint32_t lds_addr(T *p) { // byte address in LDS of a __shared__ object
    return (uint32_t)(size_t)(__attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ uint32_t lds_ld32(uint32_t a) { return *(lds_u32p)(size_t)a; }
__device__ __forceinline__ double lds_ld64(uint32_t a) { return *(lds_f64p)(size_t)a; }
__device__ __forceinline__ void lds_fadd(uint32_t a, double v) { __builtin_amdgcn_ds_atomic_fadd_f64((lds_f64p)(size_t)a, v); }
__device__ __forceinline__ void lds_iadd(uint32_t a, unsigned long long v) {
    (void)__hip_atomic_fetch_add((lds_u64p)(size_t)a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

__device__ __forceinline__ double c8_fma3s(double a, double b, double c) { // v_fma_f64 with the addend in a scalar register pair
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
__device__ __forceinline__ uint32_t c8_scan_u32(uint32_t v) { // wave64 inclusive prefix sum (DPP)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t c8_wave_min(uint32_t v) { // wave64 minimum, in every lane's hands through lane 63 (DPP: no LDS round trips)
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t c8_readlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t c8_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

using c8_rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ c8_rsrc c8_make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, (int)C8_BUF_FLAGS);
}
__device__ __forceinline__ uint32_t c8_load_u16(c8_rsrc r, uint32_t off) { return (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, (int)off, 0, 0); }
#ifndef C8_STREAM_AUX
#define C8_STREAM_AUX 0 // cache policy of the loads that stream the batch through (records read once): 2 = non-temporal
#endif
__device__ __forceinline__ uint32_t c8_load1(c8_rsrc r, uint32_t off) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, C8_STREAM_AUX); }
__device__ __forceinline__ uint2 c8_load2(c8_rsrc r, uint32_t off) {
    using v2 = __attribute__((__vector_size__(2 * sizeof(int)))) int;
    const v2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    return uint2{(uint32_t)v[0], (uint32_t)v[1]};
}
__device__ __forceinline__ uint4 c8_load4(c8_rsrc r, uint32_t off) {
    using v4 = __attribute__((__vector_size__(4 * sizeof(int)))) int;
    const v4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, C8_STREAM_AUX);
    return uint4{(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]};
}

// One column's log term (process_mapping.cpp:59-77 with get_p_obs_base.cpp:67, tv = ts = 0), factorised:
//   log(wbg * bg + wobs * om) = lw + log(om) + log1p(kappa * bg / om),   kappa = wbg / wobs, lw = log(wobs)
// log1p by six terms below 2^-8 (the next, rho^7 / 7, is under 2e-18 there), through the table log of 1 + rho beyond; a
// segment with wobs = 0 ({+inf, wbg}: mapping quality 0, mappability 0) scores log(wbg * bg).  The ONE place the term is
// computed: the table of a workgroup and the columns of a general tile both come from here.
#if !defined(__HIP_DEVICE_COMPILE__)
__device__ double c8_col_term(double kappa, double lw, C8Lom lo, double2 bg, const LogTabEntry *logtab); // (host pass: a name only)
#else
__device__ __forceinline__ double c8_col_term(double kappa, double lw, C8Lom lo, double2 bg, const LogTabEntry *logtab) {
    const double ki = fabs(kappa) * lo.iom;
    const double rho = ki * bg.x;
    double p = c8_fma3s(ki, bg.y, -0.2); // rho / 6 - 1 / 5
    p = c8_fma3s(rho, p, 0.25);
    p = c8_fma3s(rho, p, -1.0 / 3.0);
    p = fma(rho, p, 0.5);
    p = fma(rho, -p, 1.0);
    const double l0 = lo.lom + lw;
    double t = fma(rho, p, l0);
    if (!(rho < 0.00390625)) {
        if (rho < 1e290) {
            const double x = 1.0 + rho;
            const double corr = (rho - (x - 1.0)) * __builtin_amdgcn_rcp(x);
            t = l0 + (log_tab_eval_s(x, logtab) + corr);
        } else {
            const bool deg = !(fabs(kappa) < 1e300);
            double x = deg ? lw * bg.x : 1.0 + rho;
            const double corr = deg ? 0.0 : (rho - (x - 1.0)) * __builtin_amdgcn_rcp(x);
            double adj = 0.0;
            if (x < 2.2250738585072014e-308 && x > 0.0) { // subnormal
                x *= 18014398509481984.0;                   // 2^54
                adj = -37.429947750237048;                  // -54 ln 2
            }
            const double lx = x > 0.0 ? (x <= 1.7976931348623157e308 ? log_tab_eval_s(x, logtab) + adj : x) : (x == 0.0 ? -INFINITY : __builtin_nan(""));
            t = deg ? lx : l0 + (lx + corr);
        }
    }
    return t;
}
#endif
// a segment's {kappa, lw} from its read's and its node class's scalars (process_mapping.cpp:41,66-75; a consensus FASTA: wbg = 0)
__device__ __forceinline__ C8KL c8_seg_kl(double omp, double lp, double ip, const HcNodeDev &nd, bool consensus) {
    const double pcm = omp * nd.mappability;
    const double wbg = consensus ? 0.0 : 1.0 - pcm;
    double kappa = consensus ? 0.0 : wbg * (ip * nd.inv_mm);
    double lw = lp + nd.ln_w;
    if (!(kappa < 1e300)) { // wobs = 0: the column is log(wbg * bg)
        kappa = INFINITY;
        lw = wbg;
    }
    return C8KL{kappa, lw};
}

struct C8Args {
    const uint4 *rhdr;
    const uint32_t *srec;
    const uint32_t *crec;
    const uint16_t *node_hi;  // per node: its class's byte offset in the table, or 0xE000 | class for a class outside it
    const HcNodeDev *cls_tab; // per node class: {ln_w, inv_mm, mappability, match}
    const double *qscore;
    const double *rdtab;
    const double *gmemo;      // [100 mapping qualities][C8_NMEMO classes][C8_QMAX][match][read base] column terms (hc_col8_memo_kernel)
    const double *gmemo2;     // [100 mapping qualities][n_cls classes][HC_EXT_QMAX][match][read base]: the wide table (hc_col8_memo2_kernel)
    double *nodeW;
    double *totals;
    double bep;
    uint32_t n_reads, rows, n_cls;
    uint64_t n_cols4, n_segs8; // bytes of crec / srec
    uint32_t use_bep, consensus;
};

struct C8Tile { // one tile's extents (wave uniform)
    uint32_t r, n, s_base, n_seg, c_base, n_col;
    uint32_t w1; // end of the work unit the tile lies in
};

#ifdef C8_PHASES // developer aid: where a wave's time goes (shader clock between the phases of the tile loop, summed over the waves)
__device__ unsigned long long c8_phase_cycles[12]; // top, reads + Q, C, general C2, D fast, D general, tiles, not-fast tiles, end (classes of the next tile), gfast tiles
#define C8_MARK(i)                                                    \
    do {                                                              \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        ph_acc[i] += now_ - ph_t;                                     \
        ph_t = now_;                                                  \
    } while (0)
#else
#define C8_MARK(i)
#endif
#ifdef C8_STATS // developer aid: tiles, reads, general tiles, placed windows, tiles with a mapping outside the window
__device__ unsigned long long c8_stats[8];
#define C8_COUNT(slot, n)                                                  \
    do {                                                                   \
        if (lane == 0) atomicAdd(&c8_stats[slot], (unsigned long long)(n)); \
    } while (0)
#else
#define C8_COUNT(slot, n)
#endif

template <class K> __global__ __launch_bounds__(K::THREADS) void hc_segment_col8_kernel(C8Args a) {
    constexpr int C8_CPL = K::CPL, C8_CAPC = K::CAPC, C8_CAPS = K::CAPS, C8_SPASS = K::SPASS, C8_WIN = K::WIN, C8_WAVES = K::WAVES, C8_THREADS = K::THREADS;
    constexpr int C8_NG = C8_CPL / 4; // 16-byte groups of column records per lane
    using C8Slice = c8::C8Slice<K>;
    __shared__ C8Lds<K> S;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- the workgroup's tables
    for (int i = tid; i < 202; i += C8_THREADS) {
        const int qi = i >> 1;
        const double e = (qi == 100 || a.use_bep) ? a.bep : a.qscore[qi];
        // (a mismatch: the reference pushes 1 - e into a vector as a double and get_p_obs_base takes 1 - that: 1 - (1 - e), which is e
        // only to the double's rounding of 1 - e; get_p_obs_base.cpp:21,67)
        const double om = (i & 1) ? 1.0 - e : 1.0 - (1.0 - e);
        S.lom[qi][i & 1] = C8Lom{log_pos(om), 1.0 / om};
    }
    if (tid < 4) {
        const double f = tid == 0 ? 0.27532 : tid == 1 ? 0.30044 : tid == 2 ? 0.25780 : 0.16644;
        S.bg[tid] = double2{f, f * (1.0 / 6.0)};
    }
    if (tid >= 64 && tid < 128) S.logtab[tid - 64] = c8_log_table[tid - 64];
    if (tid == 0) S.ticket = C8_WAVES; // (a wave's first ticket is its own number)
    if (tid >= 128 && tid < 128 + (int)HC_MAX_NODE_CLASSES) S.cls[tid - 128] = a.cls_tab[min((uint32_t)(tid - 128), a.n_cls - 1u)];
    C8Slice &L = S.slice[wave];
    for (int i = lane; i < C8_WIN; i += 64) {
        L.win_d[i] = 0.0;
        L.win_i[i] = 0ull;
    }
    if (lane < 4) L.ps[lane] = 0u;
    if (lane == 0) L.info[0] = 0u;
    // the table of column terms of the VGAN_HC_MAPQ_MAJOR reads: the context's table (hc_col8_memo_kernel, built once with the
    // same c8_col_term) has it as one contiguous block
    for (int i = tid; i < C8_NMEMO * C8_QMAX * 8; i += C8_THREADS) S.memo[i] = a.gmemo[(size_t)VGAN_HC_MAPQ_MAJOR * (C8_NMEMO * C8_QMAX * 8) + i];
    __syncthreads(); // from here on every wave is on its own

    // Work: the batch in blocks of C8_BLOCK reads, dealt to the workgroups round robin (workgroup b owns blocks b, b + G, b + 2G,
    // ...: every workgroup sees the same mix of cheap and dear reads, and no counter is shared between CUs -- one global ticket
    // counter hands out ~80 M tickets a second, which this kernel's ~70 k tiles a second per CU would saturate), and handed to
    // the workgroup's waves by a ticket counter in LDS.  A ticket is a whole block through the first three quarters of the
    // workgroup's blocks, then half a block, then a quarter: the waves of a workgroup end within a few tiles of each other.
    const uint32_t n_blocks = (a.n_reads + C8_BLOCK - 1u) / C8_BLOCK;
    const uint32_t nb_wg = blockIdx.x < n_blocks ? (n_blocks - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0u; // this workgroup's blocks
    const uint32_t t_whole = nb_wg * 3u / 4u, t_half = 2u * (nb_wg * 3u / 20u), t_all = t_whole + t_half + 4u * (nb_wg - t_whole - t_half / 2u);
    auto unit_of = [&](uint32_t t, uint32_t &first, uint32_t &w1) { // ticket -> reads [first, w1); false: the queue is empty
        if (t >= t_all) {
            first = w1 = a.n_reads;
            return false;
        }
        uint32_t blk, sub, len;
        if (t < t_whole) {
            blk = t, sub = 0u, len = C8_BLOCK;
        } else if (t < t_whole + t_half) {
            const uint32_t u = t - t_whole;
            blk = t_whole + (u >> 1), sub = (u & 1u) * (C8_BLOCK / 2u), len = C8_BLOCK / 2u;
        } else {
            const uint32_t u = t - t_whole - t_half;
            blk = t_whole + (t_half >> 1) + (u >> 2), sub = (u & 3u) * (C8_BLOCK / 4u), len = C8_BLOCK / 4u;
        }
        first = min((blk * gridDim.x + blockIdx.x) * C8_BLOCK + sub, a.n_reads);
        w1 = min(a.n_reads, first + len);
        return first < w1;
    };
    auto grab_ticket = [&]() {
        uint32_t t = 0u;
        if (lane == 0) t = __hip_atomic_fetch_add(&S.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return c8_first(t);
    };
    const c8_rsrc rs_hdr = c8_make_rsrc(a.rhdr, (a.n_reads + 1u) * 16u);
    const c8_rsrc rs_nhi = c8_make_rsrc(a.node_hi, a.rows * 2u);
    const uint32_t lane4 = (uint32_t)lane * 4u, laneC = (uint32_t)lane * (uint32_t)C8_CPL, laneC4 = laneC * 4u;
    const uint32_t memo_base = lds_addr(&S.memo[0]);
    const uint32_t info_base = lds_addr(&L.info[0]), wind_base = lds_addr(&L.win_d[0]), wini_base = lds_addr(&L.win_i[0]);

    auto header_load = [&](uint32_t first, uint32_t w1) { // lane t holds rhdr[first + t] for t = 0..C8_NR
        const uint32_t rr = min(first + min((uint32_t)lane, (uint32_t)C8_NR), w1);
        return c8_load4(rs_hdr, rr * 16u);
    };
    auto tile_form = [&](const uint4 &h, uint32_t r, uint32_t w1) { // reads [r, r + n): as many as fit the tile
        const uint32_t hs0 = c8_first(h.x), hc0 = c8_first(h.z);
        const bool fits = lane >= 1 && lane <= C8_NR && r + (uint32_t)lane <= w1 && h.x - hs0 <= (uint32_t)C8_CAPS && h.z - hc0 <= (uint32_t)C8_CAPC;
        const uint32_t n = max(1u, (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(fits)));
        const uint32_t n_seg = min(c8_readlane(h.x, n) - hs0, (uint32_t)C8_CAPS), n_col = min(c8_readlane(h.z, n) - hc0, (uint32_t)C8_CAPC);
        return C8Tile{r, n, hs0, n_seg, hc0, n_col, w1};
    };
    // a tile's loads are issued unconditionally (`live` false: descriptors of length zero, nothing is fetched)
    auto request_cols = [&](const C8Tile &t, bool live, uint4 (&rq)[C8_NG]) { // the lane's C8_CPL consecutive columns
        const c8_rsrc rs_c = c8_make_rsrc(a.crec + t.c_base, live ? t.n_col * 4u : 0u);
#pragma unroll
        for (int g = 0; g < C8_NG; ++g) rq[g] = c8_load4(rs_c, laneC4 + (uint32_t)g * 16u);
    };
    auto request_segs = [&](const C8Tile &t, bool live, uint32_t (&sr)[C8_SPASS]) { // (VGAN_HC_SREC: node | seg_start << 18 | read index & 7 << 29)
        const c8_rsrc rs_s = c8_make_rsrc(a.srec + t.s_base, live ? t.n_seg * 4u : 0u);
#pragma unroll
        for (int k = 0; k < C8_SPASS; ++k) sr[k] = c8_load1(rs_s, lane4 + (uint32_t)k * 256u);
    };

    double sumT = 0.0;                      // sum of the column terms (the W window's fp64 part passes through here at a flush)
    unsigned long long accQ = 0, accN = 0;  // sum over the segments of {sum of Q above 2, count of the others} (taken at the window's flush)
    uint32_t winbase = 0xFFFFFFFFu;         // no window yet (wave uniform)
    bool need_place = true;
    auto window_flush = [&](uint32_t wb) {
        for (uint32_t j = lane; j < (uint32_t)C8_WIN; j += 64) {
            const double d = L.win_d[j];
            const unsigned long long u = L.win_i[j];
            if (d != 0.0 || u != 0ull) {
                sumT += d;
                accQ += u >> 32;
                accN += (uint32_t)u;
                // - U_m = ln(10)/10 * (sum of Q above 2) + log(4) * (count of the others): miscfunc.h:180-188
                const double mu = fma((double)(uint32_t)(u >> 32), 0.23025850929940457, (double)(uint32_t)u * 1.3862943611198906);
                unsafeAtomicAdd(&a.nodeW[wb + j], d + mu);
                L.win_d[j] = 0.0;
                L.win_i[j] = 0ull;
            }
        }
    };
    auto next_first = [&](const C8Tile &t, uint32_t &first, uint32_t &w1, bool &fresh) {
        first = t.r + t.n;
        w1 = t.w1;
        fresh = false;
        if (first >= t.w1) {
            (void)unit_of(grab_ticket(), first, w1); // (an empty queue, or the empty tail of the batch's last block: first = w1 = n_reads)
            fresh = true;
        }
    };

    // ---- prologue.  In the loop everything a tile reads was requested a whole tile earlier (its header two tiles earlier): a
    // fast tile is ~2.5 us of a wave's time, about one trip to HBM under load.
    auto request_classes = [&](const uint32_t (&srx)[C8_SPASS], uint32_t (&nh)[C8_SPASS]) { // (a lane without a segment reads node 0)
#pragma unroll
        for (int k = 0; k < C8_SPASS; ++k) nh[k] = c8_load_u16(rs_nhi, min(srx[k] & VGAN_HC_SREC_MAX_NODE, a.rows - 1u) * 2u);
    };
    uint32_t u0, u1;
    if (!unit_of((uint32_t)wave, u0, u1)) return;
    uint4 Hn = header_load(u0, u1);
    C8Tile T = tile_form(Hn, u0, u1);
    uint32_t h_q = Hn.y, h_c = Hn.z, h_am = Hn.w; // of the tile's reads (lane t: read r + t)
    uint32_t fn, wn;
    bool fresh_n, fresh = true;
    next_first(T, fn, wn, fresh_n);
    Hn = header_load(fn, wn);
    uint4 rq[C8_NG], rqN[C8_NG]; // the tile's column records (the lane's columns, four to a register group), the next tile's
    uint32_t sr[C8_SPASS], srN[C8_SPASS];
    uint32_t nhi[C8_SPASS], nhiN[C8_SPASS]; // the mappings' node classes
    request_segs(T, true, sr);
    request_cols(T, true, rq);
    request_classes(sr, nhi);

#ifdef C8_PHASES
    unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_readcyclecounter();
#endif
    // The tile loop is a step taken twice, the second time with the records' registers in each other's roles: what the step loads for the
    // next tile is where the next step reads its own from, and nothing is moved at a step's end (14 register moves a tile: 1.3 % of the
    // launch; -DC8_ROTATE keeps the one-step loop that moved them, for A/B runs).
#ifdef C8_UNROLL2
    auto tile_step = [&](uint4(&rq)[C8_NG], uint4(&rqN)[C8_NG], uint32_t(&sr)[C8_SPASS], uint32_t(&srN)[C8_SPASS], uint32_t(&nhi)[C8_SPASS],
                         uint32_t(&nhiN)[C8_SPASS]) __attribute__((always_inline)) -> bool {
#else
    while (true) {
#endif
        // ---- the next tile: formed from its header; its segment and column records, its reads' scalars and the header after it requested
        const bool has_next = fn < a.n_reads;
        const C8Tile Tn = tile_form(Hn, fn, wn);
        const uint32_t hn_q = Hn.y, hn_c = Hn.z, hn_am = Hn.w;
        uint32_t f2 = a.n_reads, w2 = a.n_reads;
        bool fresh_2 = false;
        if (has_next) next_first(Tn, f2, w2, fresh_2);
        request_segs(Tn, has_next, srN);
        request_cols(Tn, has_next, rqN);
        Hn = header_load(f2, w2);
#ifdef C8_L2_PREFETCH // (measured: 0.437 against 0.361 ms -- more requests in flight make every one of them slower; kept for A/B runs)
        // the tile after the next one: its records' cache lines are touched now (one dword a line, the value is not used), so that
        // the loads a tile from now find them on the chip -- a wave's tile is shorter than a trip to HBM under this kernel's load
        uint32_t pf_touch;
        {
            const uint64_t c2 = (uint64_t)Tn.c_base + Tn.n_col, s2 = (uint64_t)Tn.s_base + Tn.n_seg;
            const uint8_t *pc = reinterpret_cast<const uint8_t *>(a.crec + c2) + (uint32_t)lane * 128u;
            const uint8_t *ps2 = reinterpret_cast<const uint8_t *>(a.srec + s2) + ((uint32_t)lane - 16u) * 128u;
            const uint8_t *pp = lane < 16 ? pc : ps2;
            const bool onp = has_next && (lane < 16 ? c2 * 4u + (uint32_t)lane * 128u < a.n_cols4 : (lane < 28 && s2 * 8u + ((uint32_t)lane - 16u) * 128u < a.n_segs8));
            pf_touch = onp ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(pp)) : 0u;
        }
#endif
        if (fresh) need_place = true;
        C8_COUNT(0, 1);
        C8_COUNT(1, T.n);

        C8_MARK(0);
        // ---- the tile's reads
        {
            const uint32_t q_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h_q, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
            if ((uint32_t)lane < T.n) {
                const uint32_t A = h_am & 0xFFFFu, QL = min(q_next - h_q, 0xFFFFu);
                const uint32_t mq = min(h_am >> 16, 99u), major = mq == (uint32_t)VGAN_HC_MAPQ_MAJOR ? 0x80000000u : 0u; // (the sign bit: one compare tells)
                L.rdA[lane] = uint2{A | (QL << 16), min(h_c - T.c_base, (uint32_t)C8_CAPC) | (mq << 16) | major};
            }
        }
        uint32_t rec[C8_CPL];
#pragma unroll
        for (int g = 0; g < C8_NG; ++g) {
            rec[4 * g] = rq[g].x;
            rec[4 * g + 1] = rq[g].y;
            rec[4 * g + 2] = rq[g].z;
            rec[4 * g + 3] = rq[g].w;
        }

        // ---- Q: prefix sums of {Q above 2: Q << 11, else 1} over the tile's columns, from byte 2 of the records
        bool q_plain, q_ext, tile_hot; // every quality byte of the tile in [0, C8_QMAX); one at 90 or above (bytes are signed, as the reference reads them)
        {
            uint32_t loc[C8_CPL], run = 0u, mxu = 255u; // mxu: the largest quality byte read as UNSIGNED (a negative one is 128 and above)
            // (the short variant only: with two waves or one to a SIMD -- the long variants -- the compiler's own schedule of the plain
            // form is faster: 0.59 against 0.68 ms at 300 bp)
            if constexpr (C8_CPL <= 8) {
                mxu = 0u;
#pragma unroll
                for (int e = 0; e < C8_CPL; ++e) {
                    // byte 2 taken where it is used (SDWA): Q << 11, Q > 2, max -- no extraction of its own
                    uint32_t q11, term;
                    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(q11) : "v"(11u), "v"(rec[e]));
                    uint64_t gt2; // (a mask of its own per column: through vcc the columns' compares and selects would queue up behind each other)
                    asm("v_cmp_lt_u32_sdwa %0, %1, %2 src0_sel:DWORD src1_sel:BYTE_2" : "=s"(gt2) : "v"(2u), "v"(rec[e]));
                    asm("v_cndmask_b32 %0, 1, %1, %2" : "=v"(term) : "v"(q11), "s"(gt2));
                    asm("v_max_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(mxu) : "v"(rec[e]), "v"(mxu));
                    run += term;
                    loc[e] = run;
                }
            }
            int mn = 0, mx = (int)mxu;
            if (C8_CPL > 8 || __builtin_expect(__builtin_amdgcn_ballot_w64(mxu >= 90u) != 0, 0)) { // a byte at 90 or above, or a negative one: the signed form
                run = 0u;
                mn = 127, mx = -128;
#pragma unroll
                for (int e = 0; e < C8_CPL; ++e) {
                    const int Q = __builtin_amdgcn_sbfe((int)rec[e], 16u, 8u);
                    run += Q > 2 ? (uint32_t)Q << 11 : 1u;
                    loc[e] = run;
                    mn = min(mn, Q);
                    mx = max(mx, Q);
                }
            }
            const uint32_t incl = c8_scan_u32(run);
            const uint32_t before = incl - run;
            uint4 *dst = reinterpret_cast<uint4 *>(&L.ps[4 + lane * C8_CPL]);
#pragma unroll
            for (int g = 0; g < C8_NG; ++g) dst[g] = uint4{before + loc[4 * g], before + loc[4 * g + 1], before + loc[4 * g + 2], before + loc[4 * g + 3]};
            // (columns past the tile's come back as zero records: quality 0)
            q_plain = __builtin_amdgcn_ballot_w64(mn < 0 || mx >= C8_QMAX) == 0;
            q_ext = __builtin_amdgcn_ballot_w64(mn < 0 || mx >= 90) == 0; // (the wide table's range, short of the sticky qualities)
            tile_hot = __builtin_amdgcn_ballot_w64(mx >= 90) != 0; // switches the rest of a read to the background error rate (update_likelihood.cpp:40-44)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (__builtin_expect(tile_hot, 0)) {
            if ((uint32_t)lane < T.n) L.first90[lane] = 0xFFFFFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma nounroll
            for (int e = 0; e < C8_CPL; ++e) { // first90[k] = index of the first such byte in read k's quality string
                const int Q = __builtin_amdgcn_sbfe((int)rec[e], 16u, 8u);
                const uint32_t col = laneC + (uint32_t)e;
                if (Q >= 90 && col < T.n_col) {
                    uint32_t kk = 0;
                    for (uint32_t t = 1; t < T.n; ++t) kk += col >= (L.rdA[t].y & 0xFFFFu) ? 1u : 0u;
                    const uint32_t i = col - (L.rdA[kk].y & 0xFFFFu);
                    if (i < (L.rdA[kk].x >> 16)) atomicMin(&L.first90[kk], i);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        C8_MARK(1);
        // ---- C: one lane per segment
        if (need_place) { // the window sits at the lowest node id of the tile (the batch is sorted by the reads' lowest node id)
            uint32_t nmin = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < C8_SPASS; ++k)
                if ((uint32_t)k * 64u + (uint32_t)lane < T.n_seg) nmin = min(nmin, sr[k] & VGAN_HC_SREC_MAX_NODE);
            nmin = c8_wave_min(nmin);
            if (winbase != 0xFFFFFFFFu) window_flush(winbase);
            winbase = c8_first(nmin);
            need_place = false;
            C8_COUNT(3, 1);
        }
        bool all_cls = true;   // every mapping of the tile sits on a node class the tables of column terms cover (wave uniform)
        bool all_major = true; // ... and belongs to a read of mapping quality VGAN_HC_MAPQ_MAJOR: the workgroup's table in LDS
        bool tile_out = false; // a mapping of the tile lies outside the W window (wave uniform)
        bool tile_bep = false; // a mapping of the tile takes the background error rate on its own (wave uniform)
        uint64_t sticky_m[C8_SPASS];
        uint2 rd[C8_SPASS]; // the mappings' reads (rdA)
        {
            // (the three passes side by side, stage by stage: their LDS round trips overlap; a pass beyond the tile's segments works
            // on zero records and changes nothing)
            uint32_t kr[C8_SPASS], hi[C8_SPASS], gap[C8_SPASS], p_lo[C8_SPASS], p_hi[C8_SPASS];
            // lanes with a mapping outside the window / on a node class outside the tables / of a read of another mapping quality: the
            // compares' own masks, combined on the scalar unit (the vector unit is what this kernel waits for)
            uint64_t out_m = 0, cls_m = 0, minor_m = 0;
            uint32_t seg_flags = 0u; // (the long variants: per lane, counts of its mappings outside the window, bits 0-7 / on a class outside the tables, 8-15 / of a minor read, 16-23)
#pragma unroll
            for (int k = 0; k < C8_SPASS; ++k) {
                kr[k] = ((sr[k] >> 29) - T.r) & 7u;
                rd[k] = L.rdA[kr[k]];
            }
#pragma unroll
            for (int k = 0; k < C8_SPASS; ++k) {
                const uint32_t start = (sr[k] >> 18) & 0x7FFu;
                const uint32_t A = rd[k].x & 0xFFFFu, QL = rd[k].x >> 16, coff = rd[k].y & 0xFFFFu;
                const uint32_t lo = min(start, QL);
                hi[k] = min(start + A, QL);
                gap[k] = A - (hi[k] - lo); // Q5: the bytes beyond the quality string count as Q = 0, i.e. among the "others"
                p_hi[k] = L.ps[min(coff + hi[k], (uint32_t)C8_CAPC) + 3u];
                p_lo[k] = L.ps[min(coff + lo, (uint32_t)C8_CAPC) + 3u];
            }
#pragma unroll
            for (int k = 0; k < C8_SPASS; ++k) {
                sticky_m[k] = 0;
                const uint32_t ls = (uint32_t)k * 64u + (uint32_t)lane;
                const uint32_t pkd = p_hi[k] - p_lo[k];
                const uint32_t n_low = (pkd & 2047u) + gap[k], sq = pkd >> 11;
                const uint32_t node = sr[k] & VGAN_HC_SREC_MAX_NODE, sl = node - winbase;
                bool on, inside;
                if constexpr (C8_SPASS <= 3) {
                    uint64_t on_m, in_m, c_m, mj_m;
                    asm("v_cmp_lt_u32 %0, %1, %2" : "=s"(on_m) : "v"(ls), "s"(T.n_seg));
                    asm("v_cmp_lt_u32 %0, %1, %2" : "=s"(in_m) : "v"(sl), "s"((uint32_t)C8_WIN));
                    asm("v_cmp_le_u32 %0, %2, %1" : "=s"(c_m) : "v"(nhi[k]), "s"(0xE000u));
                    asm("v_cmp_gt_i32 %0, 0, %1" : "=s"(mj_m) : "v"(rd[k].y));
                    on = __builtin_amdgcn_inverse_ballot_w64(on_m), inside = __builtin_amdgcn_inverse_ballot_w64(in_m);
                    out_m |= on_m & ~in_m;
                    cls_m |= on_m & c_m;
                    minor_m |= on_m & ~mj_m;
                } else { // (six or eight passes: that many masks do not fit the scalar registers -- counted per lane, three ballots a tile)
                    on = ls < T.n_seg, inside = sl < (uint32_t)C8_WIN;
                    const uint32_t bits = (inside ? 0u : 1u) + (nhi[k] >= 0xE000u ? 0x100u : 0u) + ((int32_t)rd[k].y < 0 ? 0u : 0x10000u);
                    seg_flags += on ? bits : 0u;
                }

                L.info[ls + 1u] = (inside ? sl * 8u : C8_OUTSIDE) | ((nhi[k] + memo_base) << 16);
                if (on) {
                    if (inside) {
                        lds_iadd(wini_base + sl * 8u, ((unsigned long long)sq << 32) | n_low);
                    } else { // (rare: the sums that otherwise pass through the window's flush)
                        accQ += sq;
                        accN += n_low;
                        unsafeAtomicAdd(&a.nodeW[min(node, a.rows - 1u)], fma((double)sq, 0.23025850929940457, (double)n_low * 1.3862943611198906));
                    }
                }
                if (__builtin_expect(tile_hot, 0)) {
                    const bool sticky = on && !a.use_bep && L.first90[kr[k]] < hi[k]; // update_likelihood.cpp:42
                    sticky_m[k] = __builtin_amdgcn_ballot_w64(sticky);
                    tile_bep = tile_bep || sticky_m[k] != 0;
                }
            }
            if constexpr (C8_SPASS > 3) {
                static_assert(C8_SPASS <= 255, "the counts' fields");
                cls_m = __builtin_amdgcn_ballot_w64((seg_flags & 0xFF00u) != 0u);
                minor_m = __builtin_amdgcn_ballot_w64((seg_flags & 0xFF0000u) != 0u);
                out_m = __builtin_amdgcn_ballot_w64((seg_flags & 0xFFu) != 0u);
            }
            all_cls = cls_m == 0;
            all_major = minor_m == 0;
            const uint32_t n_out = (uint32_t)__builtin_popcountll(out_m);
            tile_out = n_out != 0;
            if (n_out > 8) need_place = true; // (the next tile places the window anew)
        }
        // fast: every column term comes from the workgroup's table; gfast: from the context's (a read of another mapping quality)
        // wide: from the context's wide table -- a node class beyond the sixteen, a quality byte from C8_QMAX up to 89
        const bool narrow = all_cls && !tile_out && q_plain, wide = !narrow && !tile_out && q_ext && a.gmemo2 != nullptr;
        const bool tabled = narrow || wide, gfast = narrow && !all_major;
#ifdef C8_PHASES
        const bool fast = tabled && all_major;
#endif
        C8_MARK(2);
        if (gfast) {
#pragma unroll
            for (int k = 0; k < C8_SPASS; ++k) // byte offset of the read's mapping quality in the context's table (over the dead prefix sums; ps[3] = 0 stays)
                L.ps[4 + k * 64 + lane] = ((rd[k].y >> 16) & 0x7Fu) * (uint32_t)(C8_NMEMO * C8_CLS_BYTES);
        }
        if (__builtin_expect(wide, 0)) {
            // the same for the wide table; a mapping's info[] gets its class's offset there in units of 64 bytes (a class's part is
            // HC_EXT_QMAX of them), where the narrow tiles keep the LDS address of the class's part of the workgroup's table
#pragma unroll
            for (int k = 0; k < C8_SPASS; ++k) {
                const uint32_t ls = (uint32_t)k * 64u + (uint32_t)lane;
                const uint32_t cls = nhi[k] < 0xE000u ? nhi[k] / C8_CLS_BYTES : min(nhi[k] - 0xE000u, a.n_cls - 1u);
                L.ps[4 + ls] = ((rd[k].y >> 16) & 0x7Fu) * (a.n_cls * C8_CLS2_BYTES);
                L.info[ls + 1u] = (L.info[ls + 1u] & 0xFFFFu) | ((cls * HC_EXT_QMAX) << 16);
            }
        }
        if (__builtin_expect(!tabled, 0)) {
            // a general tile: {kappa, lw} per segment over the (now dead) prefix sums
            C8_COUNT(2, 1);
            if (tile_out) C8_COUNT(4, 1);
            if ((uint32_t)lane < T.n) { // (a rare tile: its reads' scalars are fetched here, not ahead)
                const uint32_t mq = min(h_am >> 16, 99u);
                L.rdB[lane][0] = a.rdtab[3 * mq];
                L.rdB[lane][1] = a.rdtab[3 * mq + 1];
                L.rdB[lane][2] = a.rdtab[3 * mq + 2];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < C8_SPASS; ++k) {
                if ((uint32_t)k * 64u < T.n_seg) {
                    const uint32_t ls = (uint32_t)k * 64u + (uint32_t)lane;
                    const uint32_t kr = ((sr[k] >> 29) - T.r) & 7u;
                    const uint32_t cls = nhi[k] < 0xE000u ? nhi[k] / C8_CLS_BYTES : min(nhi[k] - 0xE000u, a.n_cls - 1u);
                    const HcNodeDev nd = cls < HC_MAX_NODE_CLASSES ? S.cls[cls] : a.cls_tab[cls]; // (the rarer classes' scalars: the context's array)
                    C8KL kl = c8_seg_kl(L.rdB[kr][0], L.rdB[kr][1], L.rdB[kr][2], nd, a.consensus != 0);
                    if ((sticky_m[k] >> lane) & 1ull) kl.kappa = -kl.kappa;
                    L.kl[ls] = kl;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        C8_MARK(3);

        // ---- D: the lane's columns, eight at a time.  Index of a column's mapping * 4 = running sum of the head flags (byte 3 = 4).
        {
            uint32_t own[C8_CPL];
            own[0] = rec[0] >> 24;
#pragma unroll
            for (int e = 1; e < C8_CPL; ++e)
                asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(own[e]) : "v"(rec[e]), "v"(own[e - 1]));
            const uint32_t incl = c8_scan_u32(own[C8_CPL - 1]);
            const uint32_t excl = incl - own[C8_CPL - 1];
            const uint32_t ibase = excl + info_base; // (info[0] stands for "no mapping yet")
            if (__builtin_expect(tabled, 1)) {
                auto steps = [&](auto mode, const uint32_t *rec8, const uint32_t *own8) { // 0: the workgroup's table; 1: the context's; 2: its wide one
                    constexpr bool G = decltype(mode)::value != 0, W2 = decltype(mode)::value == 2;
                    double t[8];
                    uint64_t valid[8];
                    uint32_t inf[8], rbo[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) inf[e] = lds_ld32(ibase + own8[e]);
                    if constexpr (G) {
                        // the read's part of the context's table, less the LDS address info[] carries (the "no mapping yet" slot reads ps[3] = 0)
                        const uint32_t rbase = ibase + (lds_addr(&L.ps[3]) - info_base);
#pragma unroll
                        for (int e = 0; e < 8; ++e) rbo[e] = lds_ld32(rbase + own8[e]) - (W2 ? 0u : memo_base);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        // both bases among A C G T (process_mapping.cpp:62-63): the codes (b >> 1) & 3 select their letters out of
                        // "A.C.T.G." and the pair of letters is compared with the pair of bytes
                        uint32_t sel, want;
                        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(sel) : "v"(rec8[e]), "s"(0x0606u), "v"(0x0C0C0000u)); // (one scalar operand an instruction)
                        want = __builtin_amdgcn_perm(0x00470054u, 0x00430041u, sel);
                        asm("v_cmp_eq_u32_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:WORD_0" : "=s"(valid[e]) : "v"(want), "v"(rec8[e]));
                        const uint32_t q = __builtin_amdgcn_ubfe(rec8[e], 16u, 8u);
                        uint32_t row; // 2 q + (graph base == read base): the compare writes VCC, the add takes it as carry
                        asm("v_cmp_eq_u32_sdwa vcc, %1, %1 src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_addc_co_u32 %0, vcc, %2, %2, vcc" : "=v"(row) : "v"(rec8[e]), "v"(q) : "vcc");
                        const uint32_t rc = __builtin_amdgcn_ubfe(rec8[e], 9u, 2u);
                        const uint32_t ad = (rc << 3) + ((row << 5) + (W2 ? (inf[e] >> 16) << 6 : inf[e] >> 16));
                        if constexpr (W2) t[e] = *reinterpret_cast<const double *>(reinterpret_cast<const uint8_t *>(a.gmemo2) + (ad + rbo[e]));
                        else if constexpr (G) t[e] = *reinterpret_cast<const double *>(reinterpret_cast<const uint8_t *>(a.gmemo) + (ad + rbo[e]));
#if defined(C8_EXP) && (C8_EXP & 2) // (developer pricing run: the table read without bank conflicts -- wrong sums)
                        else {
                            asm volatile("" ::"v"(ad)); // (the address is still computed: only the conflicts go)
                            t[e] = lds_ld64(memo_base + (uint32_t)lane * 8u + (uint32_t)e * 512u);
                        }
#else
                        else t[e] = lds_ld64(ad);
#endif
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        uint32_t sa;
                        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(sa) : "v"(inf[e]), "s"(wind_base));
#if defined(C8_EXP) && (C8_EXP & 1) // (developer pricing run: the window adds without conflicts -- wrong sums)
                        sa = wind_base + (uint32_t)lane * 8u + (sa & 0u);
#endif
                        if (__builtin_amdgcn_inverse_ballot_w64(valid[e])) lds_fadd(sa, t[e]);
                    }
                };
#pragma unroll
                for (int g8 = 0; g8 < C8_CPL; g8 += 8) {
                    if (__builtin_expect(wide, 0)) steps(std::integral_constant<int, 2>{}, &rec[g8], &own[g8]);
                    else if (gfast) steps(std::integral_constant<int, 1>{}, &rec[g8], &own[g8]);
                    else steps(std::integral_constant<int, 0>{}, &rec[g8], &own[g8]);
                }
            } else {
                const uint32_t klbase = lds_addr(&L.kl[0]) - 4u * 4u; // (own counts from 4: segment 0)
#pragma unroll
                for (int g8 = 0; g8 < C8_CPL; g8 += 8) {
                    uint32_t inf[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) inf[e] = lds_ld32(ibase + own[g8 + e]);
#pragma unroll 1
                    for (int e = 0; e < 8; ++e) {
                        const uint32_t r_ = rec[g8 + e];
                        const uint32_t want = __builtin_amdgcn_perm(0x47544341u, 0x47544341u, (r_ >> 1) & 0x0303u);
                        const bool valid = (want & 0xFFFFu) == (r_ & 0xFFFFu);
                        int q = __builtin_amdgcn_sbfe((int)r_, 16u, 8u);
                        q = q < 0 ? 0 : (q > 99 ? 99 : q); // qscore_vec's index
                        uint32_t row = 2u * (uint32_t)q + ((r_ & 0xFFu) == ((r_ >> 8) & 0xFFu) ? 1u : 0u);
                        const uint32_t own_seg4 = excl + own[g8 + e]; // (index + 1) * 4
                        const c8_v2d klv = *(lds_v2dp)(size_t)(klbase + own_seg4 * 4u);
                        const C8KL kl{klv.x, klv.y};
                        if (tile_bep && __double2hiint(kl.kappa) < 0) row = 200u + (row & 1u);
                        const C8Lom lo = *reinterpret_cast<const C8Lom *>(reinterpret_cast<const uint8_t *>(S.lom) + (row << 4));
                        const double2 bg = S.bg[(r_ >> 9) & 3u];
                        const double tt = c8_col_term(kl.kappa, kl.lw, lo, bg, S.logtab);
                        if (valid && own_seg4 != 0u) {
                            const uint32_t so = inf[e] & 0xFFFFu;
                            if (so == C8_OUTSIDE) {
                                sumT += tt;
                                unsafeAtomicAdd(&a.nodeW[min(a.srec[T.s_base + (own_seg4 >> 2) - 1u] & VGAN_HC_SREC_MAX_NODE, a.rows - 1u)], tt);
                            } else {
                                lds_fadd(wind_base + so, tt);
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane == 0) L.ps[3] = 0u; // (the segments' {kappa, lw} lay over the prefix sums' first words)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef C8_PHASES
        C8_MARK(fast ? 4 : 5);
        ph_acc[6] += 1;
        ph_acc[7] += fast ? 0 : 1;
        ph_acc[9] += gfast ? 1 : 0;
#endif
#ifdef C8_UNROLL2
        if (!has_next) return false;
#else
        if (!has_next) break;
#endif
        request_classes(srN, nhiN); // (the next tile's segment records have been on their way for a whole tile; these land during its Q)
#ifdef C8_L2_PREFETCH
        asm volatile("" ::"v"(pf_touch)); // (the touch's register is its own until here)
#endif
        C8_MARK(8);
        T = Tn;
#ifndef C8_UNROLL2
#pragma unroll
        for (int k = 0; k < C8_SPASS; ++k) {
            sr[k] = srN[k];
            nhi[k] = nhiN[k];
        }
#endif

        h_q = hn_q;
        h_c = hn_c;
        h_am = hn_am;
        fn = f2;
        wn = w2;
        fresh = fresh_n;
        fresh_n = fresh_2;
#ifdef C8_UNROLL2
        return true;
    };
    while (true) {
        if (!tile_step(rq, rqN, sr, srN, nhi, nhiN)) break;
        if (!tile_step(rqN, rq, srN, sr, nhiN, nhi)) break;
    }
#else
#pragma unroll
        for (int g = 0; g < C8_NG; ++g) rq[g] = rqN[g];
    }
#endif
    if (winbase != 0xFFFFFFFFu) window_flush(winbase);
#ifdef C8_PHASES
    if (lane == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(&c8_phase_cycles[i], ph_acc[i]);
#endif
    sumT = wave_sum(sumT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        accQ += (unsigned long long)__shfl_xor((long long)accQ, o, 64);
        accN += (unsigned long long)__shfl_xor((long long)accN, o, 64);
    }
    if (lane == 0 && a.totals) {
        double *t = a.totals + ((blockIdx.x * C8_WAVES + (uint32_t)wave) % HC_TOTAL_SLOTS) * HC_TOTAL_STRIDE;
        unsafeAtomicAdd(&t[0], sumT);
        unsafeAtomicAdd(&t[1], -fma((double)accQ, 0.23025850929940457, (double)accN * 1.3862943611198906));
    }
}

// The context's table of column terms: entry [mapping quality][node class][quality][match][read base] = the term of a column
// of a mapping of a read of that mapping quality on a node of that class -- c8_col_term of c8_seg_kl, the code every other
// column goes through.  A pure function of the context (its graph's node classes, its error-rate parameters): built once, at
// vgan_hc_create.  One workgroup per mapping quality.
__global__ __launch_bounds__(256) void hc_col8_memo_kernel(const double *__restrict__ qscore, const double *__restrict__ rdtab,
                                                           const HcNodeDev *__restrict__ cls_tab, uint32_t n_cls, double bep, uint32_t use_bep,
                                                           uint32_t consensus, double *__restrict__ out) {
    __shared__ C8Lom lom[101][2];
    __shared__ double2 bg[4];
    __shared__ LogTabEntry logtab[64];
    const int tid = threadIdx.x;
    for (int i = tid; i < 202; i += 256) {
        const int qi = i >> 1;
        const double e = (qi == 100 || use_bep) ? bep : qscore[qi];
        const double om = (i & 1) ? 1.0 - e : 1.0 - (1.0 - e); // (get_p_obs_base.cpp:21,67: see hc_segment_col8_kernel)
        lom[qi][i & 1] = C8Lom{log_pos(om), 1.0 / om};
    }
    if (tid < 4) {
        const double f = tid == 0 ? 0.27532 : tid == 1 ? 0.30044 : tid == 2 ? 0.25780 : 0.16644;
        bg[tid] = double2{f, f * (1.0 / 6.0)};
    }
    if (tid >= 64 && tid < 128) logtab[tid - 64] = c8_log_table[tid - 64];
    __syncthreads();
    const uint32_t mq = blockIdx.x;
    const double omp = rdtab[3 * mq], lp = rdtab[3 * mq + 1], ip = rdtab[3 * mq + 2];
    const int n = (int)min(n_cls, (uint32_t)C8_NMEMO) * C8_QMAX * 8;
    double *dst = out + (size_t)mq * (C8_NMEMO * C8_QMAX * 8);
    for (int i = tid; i < C8_NMEMO * C8_QMAX * 8; i += 256) {
        double v = 0.0;
        if (i < n) {
            const int cls = i / (C8_QMAX * 8), rem = i - cls * (C8_QMAX * 8);
            const C8KL kl = c8_seg_kl(omp, lp, ip, cls_tab[cls], consensus != 0);
            v = c8_col_term(kl.kappa, kl.lw, lom[rem >> 3][(rem >> 2) & 1], bg[rem & 3], logtab);
        }
        dst[i] = v;
    }
}

// The wide table: [mapping quality][class][quality below HC_EXT_QMAX][match][read base], every class of the graph, with the same
// c8_seg_kl / c8_col_term as the tiles that compute their columns -- a tile that reads it gets the bits a general tile would compute.
// A workgroup per (mapping quality, class).
__global__ __launch_bounds__(256) void hc_col8_memo2_kernel(const double *__restrict__ qscore, const double *__restrict__ rdtab, const HcNodeDev *__restrict__ cls_tab,
                                                            uint32_t n_cls, double bep, uint32_t use_bep, uint32_t consensus, double *__restrict__ out) {
    __shared__ C8Lom lom[HC_EXT_QMAX][2];
    __shared__ double2 bg[4];
    __shared__ LogTabEntry logtab[64];
    const int tid = threadIdx.x;
    for (int i = tid; i < (int)HC_EXT_QMAX * 2; i += 256) {
        const int qi = i >> 1;
        const double e = use_bep ? bep : qscore[qi];
        const double om = (i & 1) ? 1.0 - e : 1.0 - (1.0 - e); // (get_p_obs_base.cpp:21,67: see hc_segment_col8_kernel)
        lom[qi][i & 1] = C8Lom{log_pos(om), 1.0 / om};
    }
    if (tid < 4) {
        const double f = tid == 0 ? 0.27532 : tid == 1 ? 0.30044 : tid == 2 ? 0.25780 : 0.16644;
        bg[tid] = double2{f, f * (1.0 / 6.0)};
    }
    if (tid >= 64 && tid < 128) logtab[tid - 64] = c8_log_table[tid - 64];
    __syncthreads();
    const uint32_t mq = blockIdx.x, cls = blockIdx.y;
    const C8KL kl = c8_seg_kl(rdtab[3 * mq], rdtab[3 * mq + 1], rdtab[3 * mq + 2], cls_tab[cls], consensus != 0);
    double *dst = out + ((size_t)mq * n_cls + cls) * (HC_EXT_QMAX * 8);
    for (int i = tid; i < (int)HC_EXT_QMAX * 8; i += 256) dst[i] = c8_col_term(kl.kappa, kl.lw, lom[i >> 3][(i >> 2) & 1], bg[i & 3], logtab);
}

} // namespace c8
using namespace c8;

size_t hc_col8_memo_doubles() { return (size_t)100 * C8_NMEMO * C8_QMAX * 8; }
size_t hc_col8_memo2_doubles(uint32_t n_cls) { return (size_t)100 * n_cls * HC_EXT_QMAX * 8; }
void launch_hc_col8_memo2(const HcGraphDev &g, const HcParamsDev &prm, double *out, hipStream_t st) {
    if (g.n_cls == 0) return;
    hipLaunchKernelGGL(hc_col8_memo2_kernel, dim3(100, g.n_cls), dim3(256), 0, st, g.qscore, g.rdtab, g.cls_tab, g.n_cls, prm.bep, prm.use_bep ? 1u : 0u,
                       prm.consensus ? 1u : 0u, out);
}
void launch_hc_col8_memo(const HcGraphDev &g, const HcParamsDev &prm, double *out, hipStream_t st) {
    hipLaunchKernelGGL(hc_col8_memo_kernel, dim3(100), dim3(256), 0, st, g.qscore, g.rdtab, g.cls_tab, g.n_cls, prm.bep, prm.use_bep ? 1u : 0u,
                       prm.consensus ? 1u : 0u, out);
}

#ifdef C8_PHASES
extern "C" int vgan_hc_debug_col8_phases(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(c8_phase_cycles), sizeof(c8_phase_cycles)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[12] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(c8_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#ifdef C8_STATS
extern "C" int vgan_hc_debug_col8_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(c8_stats), sizeof(c8_stats)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(c8_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// Which batches the kernel takes: node-weights accumulation of reads that fit one of its variants' tiles and W window, on a graph
// whose nodes fall into at most HC_MAX_NODE_CLASSES classes.  VGAN_HC_KERNEL=wave / tile keep the older kernels (A/B runs).
namespace {
template <class K> bool c8_variant_holds(const HcPackedDev &pk, uint32_t span) {
    return pk.max_read_segs <= (uint32_t)K::CAPS && pk.max_read_cols <= (uint32_t)K::CAPC && span < (uint32_t)K::WIN;
}
// 0 none, 1 short, 2 mid, 3 long: the smallest variant that holds every read and, where a larger one is needed for that, the larger
int c8_variant(const HcGraphDev &g, const HcPackedDev &pk) {
    const char *e = getenv("VGAN_HC_KERNEL");
    if (e && (strcmp(e, "wave") == 0 || strcmp(e, "tile") == 0)) return 0;
    if (g.n_cls == 0 || !g.node_hi || !g.cls_tab || !g.col_memo || pk.n_reads == 0) return 0;
    if (pk.max_read_qual > pk.max_read_cols || pk.qual_excess) return 0; // (a quality string that outruns its read's columns: not in the records)
    const uint32_t mean_segs = pk.n_segments / pk.n_reads, mean_cols = (uint32_t)(pk.n_cols / pk.n_reads);
    const uint32_t span = pk.max_read_node_span ? pk.max_read_node_span : mean_cols * 3u / 4u;
    // (at least two mean reads to a tile where a variant offers that: a lone read leaves half of a tile's lanes idle)
    if (c8_variant_holds<C8Short>(pk, span) && (2u * mean_segs <= (uint32_t)C8Short::CAPS && 2u * mean_cols <= (uint32_t)C8Short::CAPC)) return 1;
    if (c8_variant_holds<C8Mid>(pk, span) && (2u * mean_segs <= (uint32_t)C8Mid::CAPS && 2u * mean_cols <= (uint32_t)C8Mid::CAPC)) return 2;
    if (c8_variant_holds<C8Long>(pk, span)) return 3;
    return 0;
}
} // namespace
bool hc_col8_kernel_fits(const HcGraphDev &g, const HcPackedDev &pk) { return c8_variant(g, pk) != 0; }

void launch_hc_segments_col8(const HcGraphDev &g, const HcPackedDev &pk, const HcParamsDev &prm, double *nodeW, double *totals, hipStream_t st) {
    if (pk.n_reads == 0) return;
    static int n_cu_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (n_cu_dev[dev] <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n_cu_dev[dev] = n;
    }
    const int variant = c8_variant(g, pk);
    const uint32_t waves = variant == 1 ? C8Short::WAVES : variant == 2 ? C8Mid::WAVES : C8Long::WAVES;
    const uint32_t n_blocks = (pk.n_reads + C8_BLOCK - 1u) / C8_BLOCK;
    const uint32_t blocks = std::min<uint32_t>((uint32_t)n_cu_dev[dev], (n_blocks + waves - 1) / waves); // (a workgroup per CU: all of its LDS)
    C8Args a{};
    a.rhdr = pk.rhdr;
    a.srec = pk.srec;
    a.crec = pk.crec;
    a.node_hi = g.node_hi;
    a.cls_tab = g.cls_tab;
    a.qscore = g.qscore;
    a.rdtab = g.rdtab;
    a.gmemo = g.col_memo;
    a.gmemo2 = g.col_memo2;
    a.nodeW = nodeW;
    a.totals = totals;
    a.bep = prm.bep;
    a.n_reads = pk.n_reads;
    a.rows = g.rows;
    a.n_cls = g.n_cls;
    a.n_cols4 = pk.n_cols * 4u;
    a.n_segs8 = (uint64_t)pk.n_segments * 4u;
    a.use_bep = prm.use_bep ? 1u : 0u;
    a.consensus = prm.consensus ? 1u : 0u;
    if (variant == 1) hipLaunchKernelGGL(hc_segment_col8_kernel<C8Short>, dim3(blocks), dim3(C8Short::THREADS), 0, st, a);
    else if (variant == 2) hipLaunchKernelGGL(hc_segment_col8_kernel<C8Mid>, dim3(blocks), dim3(C8Mid::THREADS), 0, st, a);
    else if (variant == 3) hipLaunchKernelGGL(hc_segment_col8_kernel<C8Long>, dim3(blocks), dim3(C8Long::THREADS), 0, st, a);
}

} // namespace vgan
#include "module_anchor.h"
const void *vgan::anchor_hc_col8() { return (const void *)&vgan::c8::hc_col8_memo_kernel; }
