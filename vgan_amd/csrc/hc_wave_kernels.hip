// HaploCart segment kernel, wave-owned form (gfx950, wave64): S_m, U_m and W[node] += S_m - U_m for the tileable reads
// of a batch.  Same arithmetic as hc_segment_tile_kernel (hc_kernels.hip; reference: src/process_mapping.cpp:4-91,
// src/update_likelihood.cpp:19-53, src/get_p_obs_base.cpp:3-69), different data path:
//
//   * the batch is read in the packed layout of HcPackedDev (hc_device.h): one 32-bit record per alignment column
//     {graph byte, read byte, quality byte, head-of-segment bit}, loaded by the column's own lane straight into a register
//     (256 coalesced bytes per wave and chunk) -- no byte windows in LDS, no index arithmetic per column;
//   * a WAVE owns its reads: a tile of up to WV_NR consecutive reads (as many as fit CAPS segments / CAPQ quality bytes of
//     wave-private LDS) is taken through  quality prefix sums -> lane per segment -> lane per column -> lane per segment
//     by the one wave, so nothing in the tile loop needs a workgroup barrier (LDS operations of a wave execute in order);
//     the waves of a workgroup share only the read-only tables built before the single __syncthreads();
//   * W[node] is kept in a wave-private LDS window over the node ids of the wave's reads (the batch is sorted by node id).
//
// Per tile:
//   Q  lane per 8 quality bytes: integer prefix sums (sum of Q above 2, count of the others: log p_err is -Q ln10/10 or
//      log 0.25, src/miscfunc.h:180-188), one DPP wave scan per 512 bytes
//   C  lane per segment: U_m from two prefix values, {kappa, lw} of the factorised column term into LDS
//   D  lane per column: owner segment = running count of head bits (ballot + v_mbcnt), the column's term
//      log(wobs) + log(om) + log1p(kappa * bg / om) from two table reads and a short series, LDS fp64 add into S_m
//   E  lane per segment: D_m = S_m - U_m into the window (or straight to HBM outside it), or streamed out
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "device_math.h"
#include "hc_device.h"

namespace vgan {
namespace wv {

#ifndef WV_CAPQ
#define WV_CAPQ 480 // quality bytes and
#endif
#ifndef WV_CAPC
#define WV_CAPC 512 // alignment columns per tile of the small variant
#endif
#ifndef WV_CAPS
#define WV_CAPS 192 // segments per tile of the small variant (three 150 bp reads on the hcfiles graph: 182 on average)
#endif
#ifndef WV_GROUP
#define WV_GROUP 1 // column chunks whose LDS reads are issued together (2: measured 3 % slower)
#endif
#ifndef WV_OCC
#define WV_OCC 4 // waves per SIMD the small variant is compiled for (registers; LDS: WV_OCC * 4 waves' slices + tables per CU)
#endif
#ifndef WV_WG_THREADS
#define WV_WG_THREADS 256 // a workgroup shares nothing but the read-only tables: its size only sets how often they are built
#endif
#ifndef WV_SUB
#define WV_SUB 2 // accumulator sub-slots per window entry (lane parity picks one: halves the same-address LDS adds of a segment)
#endif
constexpr int WV_THREADS_SMALL = WV_WG_THREADS; // workgroup of the small variant (several reads per tile)
constexpr int WV_THREADS_LARGE = 256;           // ... of the large one (a tile of one long read, a wave per SIMD)
#ifdef WV_MAX_VGPR
#define WV_VGPR_ATTR __attribute__((amdgpu_num_vgpr(WV_MAX_VGPR)))
#else
#define WV_VGPR_ATTR
#endif
#ifndef WV_NR_MAX
#define WV_NR_MAX 8
#endif
constexpr int WV_NR = WV_NR_MAX;    // reads per tile at most (their headers travel in lanes 0..WV_NR of the wave)
#ifndef WV_WIN_SLOTS
#define WV_WIN_SLOTS 160 // (a 150 bp read spans 106 node ids on the hcfiles graph, 140 at most: three sorted reads fit)
#endif
constexpr int WV_WIN = WV_WIN_SLOTS; // node ids covered by a wave's W window
constexpr uint32_t WV_BUF_FLAGS = 0x00020000u; // raw buffer descriptor, 32-bit data format (gfx9)


struct alignas(16) WvKL { // per segment: kappa = wbg / wobs, lw = log(wobs); a segment with wobs = 0: {+inf, wbg}
    double kappa, lw;     // (sign of kappa set: the segment takes the background error rate, update_likelihood.cpp:42)
};
struct alignas(16) WvLom { // per (error-rate index, base match): log(om), 1 / om
    double lom, iom;
};
struct alignas(32) WvRead { // per read of the tile
    double omp, lp, ip; // 1 - p_inc, its log (log(1 - bep) for a consensus FASTA), its reciprocal
    uint32_t a_ql;      // |algnseq| | quality string length << 16
    uint32_t qoff;      // first quality byte, relative to the tile's quality window
};

// DIRECT (node-weights accumulation only): a column's term goes straight into the W window slot of its mapping's node --
// no per-segment sum S, no pass over the segments behind the column loop.  slot[]: the segment's byte offset into win[], or
// 0xFFFF for a node outside the window (its id is then read from the segment record in HBM: the rare path).
template <int CAPS, int CAPQ, bool DIRECT> struct WvSlice { // one wave's LDS
    WvKL kl[CAPS];
    double S[DIRECT ? 1 : CAPS];
    uint16_t slot[DIRECT ? CAPS : 2];
    uint32_t ps[CAPQ + 16]; // ps[4 + i]: prefix through byte i of the quality window (ps[3] = 0: the empty prefix)
    WvRead rd[WV_NR];
    uint32_t first90[WV_NR];
    double win[WV_WIN * (DIRECT ? WV_SUB : 1)];
};

__device__ const LogTabEntry wv_log_table[64] = {VGAN_LOG_TABLE};

#ifdef WV_STATS // developer aid: how often the column loop leaves its short path (one count per wave and event)
__device__ unsigned long long wv_stats[8]; // tiles, reads, chunk groups, far groups, rare groups, segment passes, bep tiles, windows placed
#define WV_COUNT(slot, n)                                     \
    do {                                                      \
        if (lane == 0) atomicAdd(&wv_stats[slot], (unsigned long long)(n)); \
    } while (0)
#else
#define WV_COUNT(slot, n)
#endif

#ifdef WV_SPANS // developer aid: the launch's timeline wave by wave (tools/wave_spans.py)
__device__ unsigned long long wv_wave_span[4 * 16384]; // per wave: when it started / left (100 MHz clock), tiles, when it took its last unit
#endif
#ifdef WV_PHASES // developer aid: where a wave's time goes (shader clock between the phases of the tile loop, summed over the waves)
__device__ unsigned long long wv_phase_cycles[8]; // gather issue, next tile + header, reads, Q, C, D, E, tiles
#define WV_MARK(i)                                                   \
    do {                                                             \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        ph_acc[i] += now_ - ph_t;                                    \
        ph_t = now_;                                                 \
    } while (0)
#else
#define WV_MARK(i)
#endif

__device__ __forceinline__ double wv_fma3(double a, double b, double c) { // three-address v_fma_f64 (see hc_kernels.hip)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// the same with the addend in a scalar register pair: a series' constants need no vector registers and no copies into them
__device__ __forceinline__ double wv_fma3s(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
__device__ __forceinline__ uint32_t wv_scan_u32(uint32_t v) { // wave64 inclusive prefix sum (DPP)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t wv_readlane(uint32_t v, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
__device__ __forceinline__ uint32_t wv_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

using wv_rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ wv_rsrc wv_make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, (int)WV_BUF_FLAGS);
}
__device__ __forceinline__ uint32_t wv_load1(wv_rsrc r, uint32_t off) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
}
__device__ __forceinline__ uint2 wv_load2(wv_rsrc r, uint32_t off) {
    using v2 = __attribute__((__vector_size__(2 * sizeof(int)))) int;
    const v2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    return uint2{(uint32_t)v[0], (uint32_t)v[1]};
}
__device__ __forceinline__ uint4 wv_load4(wv_rsrc r, uint32_t off) {
    using v4 = __attribute__((__vector_size__(4 * sizeof(int)))) int;
    const v4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return uint4{(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]};
}
// fp64 add into LDS through a pointer the compiler KNOWS is LDS (a select between an LDS and a global destination otherwise
// ends as one flat atomic: out of order, counted in vmcnt and lgkmcnt both)
using wv_lds_dptr = __attribute__((address_space(3))) double *;
__device__ __forceinline__ void wv_lds_add(double *p, double v) { __builtin_amdgcn_ds_atomic_fadd_f64((wv_lds_dptr)p, v); }
__device__ __forceinline__ double wv_dbl(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }

// Q >= 90 switches the rest of the read to the background error rate (update_likelihood.cpp:40-44): first90[k] = index of the
// first such byte in read k's quality string.  Out of line (real data holds no such quality).  qv: the lane's 8 window bytes
// at window position pos0; reads: rd / first90 of the tile's n reads; the tile's bytes are window positions [lo, hi).
__device__ __attribute__((noinline)) void wv_first90(uint2 qv, uint32_t pos0, uint32_t lo, uint32_t hi, uint32_t n, const WvRead *rd,
                                                     uint32_t *first90) {
#pragma nounroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t w = e < 4 ? qv.x : qv.y;
        const int Q = (int)(int8_t)(w >> (8 * (e & 3)));
        const uint32_t pos = pos0 + (uint32_t)e;
        if (Q >= 90 && pos >= lo && pos < hi) {
            uint32_t kk = 0;
            for (uint32_t t = 1; t < n; ++t) kk += pos >= rd[t].qoff ? 1u : 0u;
            atomicMin(&first90[kk], pos - rd[kk].qoff);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ the kernel
struct WvArgs { // only what the kernel reads (every pointer costs two scalar registers for the whole launch)
    const uint4 *rhdr;
    const uint32_t *srec;
    const uint32_t *crec;
    const uint8_t *qualp;
    const HcNodeDev *node_tab;
    const double *qscore;
    const double *rdtab;
    double *segD_out;
    double *nodeW;
    double *totals;
    double bep;
    uint32_t *work;          // ticket counter of the launch's work queue (scalar atomic)
    uint32_t work_base;      // its value when the launch starts
    uint32_t n_reads, rows, unit_reads; // reads per work unit
    uint32_t qual_bytes; // readable bytes of qualp
    uint32_t use_bep, consensus;
};

struct WvTile { // one tile's extents (wave uniform)
    uint32_t r, n, s_base, n_seg, q_base, n_q, c_base, n_col;
    uint32_t w1; // end of the work unit the tile lies in
};
struct WvRdTab { // rdtab row of a mapping quality: 1 - p_inc, its log, its reciprocal
    double omp, lp, ip;
};
template <int SPASS, int QCH, int NCH> struct WvData { // one tile's HBM data, in flight or arrived
    uint2 qv[QCH];
    uint32_t sr[SPASS]; // VGAN_HC_SREC: node | seg_start << 18 | read index & 7 << 29
    uint32_t rec[NCH];
};

// A tile holds at most CAPS segments, CAPQ quality bytes and CAPC alignment columns: everything a tile reads from HBM is
// requested as a fixed set of loads (bounded by the buffer descriptors: what lies beyond the tile comes back as zeros) one
// tile ahead.  No load in the tile loop is conditional -- s_waitcnt vmcnt counts in order, and the compiler can only wait for
// exactly the loads it needs when it knows how many were issued after them.
template <int CAPS, int CAPQ, int CAPC, bool DIRECT, int WV_THREADS, int OCC>
__global__ __launch_bounds__(WV_THREADS, OCC) WV_VGPR_ATTR void hc_segment_wave_kernel(WvArgs a) {
    constexpr int WV_WAVES = WV_THREADS / 64;
    constexpr int SPASS = (CAPS + 63) / 64;     // segment passes per tile at most
    constexpr int QCH = (CAPQ + 8 + 511) / 512; // quality chunks (512 bytes: 8 per lane) per tile at most
    constexpr int NCH = CAPC / 64;              // column chunks per tile at most
    static_assert(CAPS >= 64 && CAPC % 64 == 0, "whole chunks");
    using Slice = WvSlice<CAPS, CAPQ, DIRECT>;
    constexpr int SUB = DIRECT ? WV_SUB : 1;
    using Data = WvData<SPASS, QCH, NCH>;
    __shared__ WvLom lom_s[101][2]; // [qscore index, 100 = background error rate][mismatch, match]
    __shared__ double2 bg_s[4];     // A C T G by (base >> 1) & 3: {frequency, frequency / 6} (the series' leading coefficient rides along)
    __shared__ WvRdTab rdtab_s[100]; // the read's share of pcm by mapping quality (process_mapping.cpp:41)
    __shared__ LogTabEntry logtab_s[64]; // log_tab.h
    __shared__ Slice slice_s[WV_WAVES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 202; i += WV_THREADS) {
        const int qi = i >> 1;
        const double e = (qi == 100 || a.use_bep) ? a.bep : a.qscore[qi];
        // (a mismatch: the reference pushes 1 - e into a vector as a double and get_p_obs_base takes 1 - that: 1 - (1 - e), which is e
        // only to the double's rounding of 1 - e -- 1e-10 relative at Q = 60; get_p_obs_base.cpp:21,67)
        const double om = (i & 1) ? 1.0 - e : 1.0 - (1.0 - e);
        lom_s[qi][i & 1] = WvLom{log_pos(om), 1.0 / om};
    }
    if (tid < 4) {
        const double f = tid == 0 ? 0.27532 : tid == 1 ? 0.30044 : tid == 2 ? 0.25780 : 0.16644;
        bg_s[tid] = double2{f, f * (1.0 / 6.0)};
    }
    if (tid < 64) logtab_s[tid] = wv_log_table[tid];
    if (tid < 100) rdtab_s[tid] = WvRdTab{a.rdtab[3 * tid], a.rdtab[3 * tid + 1], a.rdtab[3 * tid + 2]};
    Slice &L = slice_s[wave];
    for (int i = lane; i < WV_WIN * SUB; i += 64) L.win[i] = 0.0;
    if (lane < 4) L.ps[lane] = 0u;
    __syncthreads(); // the only one: from here on every wave is on its own

    // Work comes in units of a.unit_reads consecutive reads, handed out by a ticket counter: a wave that finishes early takes
    // more (CUs do not run at one speed, reads do not cost the same), and the launch ends when the last unit does instead of
    // when the slowest of a fixed set of ranges does.  The ticket is a SCALAR atomic: it returns into a scalar register and
    // counts in lgkmcnt, so the vector-memory pipeline of the tile loop (counted s_waitcnt vmcnt) never sees it.
    const uint32_t n_units = (a.n_reads + a.unit_reads - 1u) / a.unit_reads;
    auto grab_unit = [&]() {
        uint32_t t = 1u;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(a.work) : "memory");
        return t - a.work_base; // >= n_units: the queue is empty (every wave with work draws exactly one such ticket)
    };
    // (a wave's first unit is its own number: thousands of tickets drawn in the launch's first microsecond would queue)
    uint32_t unit = blockIdx.x * WV_WAVES + (uint32_t)wave;
    if (unit >= n_units) return;
    const wv_rsrc rs_hdr = wv_make_rsrc(a.rhdr, (a.n_reads + 1u) * 16u);
    const wv_rsrc rs_node = wv_make_rsrc(a.node_tab, a.rows * 32u);
    const uint32_t lane4 = (uint32_t)lane * 4u, lane8 = (uint32_t)lane * 8u;

    // a tile's header: lane t holds rhdr[first + t] for t = 0..WV_NR (the entry behind the last read gives its end)
    auto header_load = [&](uint32_t first, uint32_t w1) {
        const uint32_t rr = min(first + min((uint32_t)lane, (uint32_t)WV_NR), w1);
        return wv_load4(rs_hdr, rr * 16u);
    };
    // reads [r, r + n), n the largest count whose segments and quality bytes fit the wave's LDS (the offsets ascend, so
    // "read t still fits" is a prefix property and n is a popcount); one read always fits (the launcher's choice of variant)
    auto tile_form = [&](const uint4 &h, uint32_t r, uint32_t w1) {
        const uint32_t hs0 = wv_first(h.x), hq0 = wv_first(h.y), hc0 = wv_first(h.z);
        const bool fits = lane >= 1 && lane <= WV_NR && r + (uint32_t)lane <= w1 && h.x - hs0 <= (uint32_t)CAPS && h.y - hq0 <= (uint32_t)CAPQ &&
                          h.z - hc0 <= (uint32_t)CAPC;
        const uint32_t n = max(1u, (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(fits)));
        // (clamped: a caller's contract violation must not index past the LDS arrays)
        const uint32_t n_seg = min(wv_readlane(h.x, n) - hs0, (uint32_t)CAPS), n_q = min(wv_readlane(h.y, n) - hq0, (uint32_t)CAPQ);
        const uint32_t n_col = min(wv_readlane(h.z, n) - hc0, (uint32_t)CAPC);
        return WvTile{r, n, hs0, n_seg, hq0, n_q, hc0, n_col, w1};
    };
    // Every load of a tile is issued unconditionally (`live` false: descriptors of length zero, nothing is fetched), each
    // into the register the same tile position was just consumed from: a slot is refilled for the NEXT tile right behind its
    // last use, so no second set of registers and no copy between the sets exists.
    auto request_qual = [&](const WvTile &t, bool live, Data &d) {
        const uint32_t a0 = t.q_base & ~7u; // the quality window: aligned 8-byte words
        const wv_rsrc rs_q = wv_make_rsrc(a.qualp + a0, live ? t.n_q + (t.q_base & 7u) + 8u : 0u); // (whole words: qualp is padded)
#pragma unroll
        for (int k = 0; k < QCH; ++k) d.qv[k] = wv_load2(rs_q, lane8 + (uint32_t)k * 512u);
    };
    auto request_segs = [&](const WvTile &t, bool live, Data &d) {
        const wv_rsrc rs_s = wv_make_rsrc(a.srec + t.s_base, live ? t.n_seg * 4u : 0u);
#pragma unroll
        for (int k = 0; k < SPASS; ++k) d.sr[k] = wv_load1(rs_s, lane4 + (uint32_t)k * 256u); // (VGAN_HC_SREC words, taken apart where they are used)
    };
    auto rsrc_cols = [&](const WvTile &t, bool live) { return wv_make_rsrc(a.crec + t.c_base, live ? t.n_col * 4u : 0u); };
    // the nodes' scalars (dependent on the segment records; a lane without a segment reads node 0)
    auto node_gather = [&](const Data &d, double (&nd_lw)[SPASS], double (&nd_inv)[SPASS], double (&nd_mapp)[SPASS]) {
#pragma unroll
        for (int k = 0; k < SPASS; ++k) {
            const uint32_t o = min(d.sr[k] & VGAN_HC_SREC_MAX_NODE, a.rows - 1u) << 5;
            const uint4 v = wv_load4(rs_node, o);
            const uint2 m = wv_load2(rs_node, o + 16u);
            nd_lw[k] = wv_dbl(v.x, v.y);
            nd_inv[k] = wv_dbl(v.z, v.w);
            nd_mapp[k] = wv_dbl(m.x, m.y);
        }
    };
    auto window_flush = [&](uint32_t winbase) {
        for (uint32_t j = lane; j < (uint32_t)WV_WIN; j += 64) {
            double v = L.win[j * SUB];
            if constexpr (SUB == 2) v += L.win[j * SUB + 1];
            if (v != 0.0) {
#ifndef WV_NOFLUSH // (developer aid: what the flush's global atomics cost)
                unsafeAtomicAdd(&a.nodeW[winbase + j], v);
#endif
                L.win[j * SUB] = 0.0;
                if constexpr (SUB == 2) L.win[j * SUB + 1] = 0.0;
            }
        }
    };
    // The window sits at the lowest node id of the tile: the batch is sorted by the reads' lowest node id (vgan_hc_flatten), so
    // the wave's reads stay above it and move through the node ids slowly.  Any other order is still correct -- a segment
    // outside the window adds to W in HBM directly -- and a tile that leaves the window with many segments has the next one
    // place it anew.
    auto window_place = [&](const uint32_t (&nodes)[SPASS], const WvTile &t, uint32_t &winbase) {
        uint32_t nmin = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < SPASS; ++k)
            if ((uint32_t)k * 64u + (uint32_t)lane < t.n_seg) nmin = min(nmin, nodes[k]);
        nmin = wave_min_u32(nmin);
        if (winbase != 0xFFFFFFFFu) window_flush(winbase);
        winbase = wv_first(nmin);
        WV_COUNT(7, 1);
    };

    const uint32_t sub8 = SUB == 2 ? ((uint32_t)lane & 1u) * 8u : 0u; // the lane's accumulator sub-slot
    double sumT = 0.0, sumU = 0.0;  // sum of S_m and of U_m, each without cancellation
    uint32_t winbase = 0xFFFFFFFFu; // no window yet (wave uniform)
    bool need_place = true;

    // the tile behind `t`: in the same unit, or at the head of the next unit drawn from the queue (first = w1 = n_reads when
    // the queue is empty: no such tile)
    auto next_first = [&](const WvTile &t, uint32_t &first, uint32_t &w1, bool &fresh) {
        first = t.r + t.n;
        w1 = t.w1;
        fresh = false;
        if (first >= t.w1) {
            unit = a.work ? grab_unit() : n_units; // (no queue: one unit per wave, the fixed partition)
#ifdef WV_SAMEUNIT // (developer aid: every unit re-reads the first units' data -- what the launch costs without HBM latency)
            first = unit < n_units ? (unit % 64u) * a.unit_reads : a.n_reads;
#else
            first = unit < n_units ? unit * a.unit_reads : a.n_reads;
#endif
            w1 = min(a.n_reads, first + a.unit_reads);
            fresh = true;
        }
    };

    // ---- prologue: the first tile's header and data, the second tile's header.  In the loop a tile's data was requested a
    // whole tile earlier and its header two tiles earlier, so nothing in it waits for HBM.
    const uint32_t u0 = unit * a.unit_reads, u1 = min(a.n_reads, u0 + a.unit_reads);
    uint4 Hn = header_load(u0, u1);
    WvTile T = tile_form(Hn, u0, u1);
    uint32_t h_q = Hn.y, h_am = Hn.w; // of the tile's reads (lane t: read r + t): first quality byte, |algnseq| | mapq << 16
    uint32_t fn, wn;       // where the next tile starts and its unit ends
    bool fresh_n, fresh = true; // the (next) tile opens a unit: the W window is placed anew
    next_first(T, fn, wn, fresh_n);
    Hn = header_load(fn, wn); // (behind the last tile: the entry at n_reads, a header nobody uses)
    Data D;
    request_qual(T, true, D);
    request_segs(T, true, D);
    {
        const wv_rsrc rs_c = rsrc_cols(T, true);
#pragma unroll
        for (int k = 0; k < NCH; ++k) D.rec[k] = wv_load1(rs_c, lane4 + (uint32_t)k * 256u);
    }

#ifdef WV_PHASES
    unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_readcyclecounter();
#endif
#ifdef WV_SPANS
    const unsigned long long span_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long span_last = span_t0, span_tiles = 0;
#endif
    while (true) {
        // the nodes' scalars: the segment records arrived during the tile before, these land during Q
        double nd_lw[SPASS], nd_inv[SPASS], nd_mapp[SPASS];
        node_gather(D, nd_lw, nd_inv, nd_mapp);
        WV_MARK(0);
        // ---- the next tile: formed from its header (here since the tile before), the header after it requested
        const bool has_next = fn < a.n_reads;
        const WvTile Tn = tile_form(Hn, fn, wn); // (meaningless behind the last tile, and unused)
        const uint32_t hn_q = Hn.y, hn_am = Hn.w;
        uint32_t f2 = a.n_reads, w2 = a.n_reads;
        bool fresh_2 = false;
        if (has_next) next_first(Tn, f2, w2, fresh_2);
        Hn = header_load(f2, w2);
        if (fresh) need_place = true;
#ifdef WV_SPANS
        span_tiles += 1;
        if (fresh_2) span_last = __builtin_amdgcn_s_memrealtime();
#endif
        const uint32_t a0 = T.q_base & ~7u, qshift = T.q_base & 7u, n_qw = T.n_q + qshift;
        WV_COUNT(0, 1);
        WV_COUNT(1, T.n);
        WV_COUNT(5, (T.n_seg + 63u) / 64u);

        WV_MARK(1);
        // ---- the tile's reads: one record each
        {
            const uint32_t q_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h_q, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
            if ((uint32_t)lane < T.n) {
                const uint32_t A = h_am & 0xFFFFu, QL = min(q_next - h_q, 0xFFFFu);
                const WvRdTab rt = rdtab_s[min(h_am >> 16, 99u)];
                L.rd[lane] = WvRead{rt.omp, rt.lp, rt.ip, A | (QL << 16), h_q - a0};
                L.first90[lane] = 0xFFFFFFFFu;
            }
        }

        WV_MARK(2);
        // ---- Q: prefix sums over the quality window.  Slots past the tile's bytes receive sums nobody reads.
        bool hot = false;
        {
            uint32_t carry = 0u;
#pragma unroll
            for (int k = 0; k < QCH; ++k) {
                if ((uint32_t)k * 512u < n_qw) {
                    uint32_t loc[8], run = 0u;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const uint32_t w = e < 4 ? D.qv[k].x : D.qv[k].y;
                        const int Q = (int)(int8_t)(w >> (8 * (e & 3)));
                        run += Q > 2 ? (uint32_t)Q << 11 : 1u;
                        loc[e] = run;
                        hot |= Q >= 90;
                    }
                    const uint32_t incl = wv_scan_u32(run);
                    const uint32_t before = incl - run + carry;
                    carry += wv_readlane(incl, 63);
                    uint4 *dst = reinterpret_cast<uint4 *>(&L.ps[4 + k * 512 + lane * 8]);
                    if (k * 512 + lane * 8 < CAPQ + 8) { // (the window holds at most CAPQ + 7 bytes)
                        dst[0] = uint4{before + loc[0], before + loc[1], before + loc[2], before + loc[3]};
                        dst[1] = uint4{before + loc[4], before + loc[5], before + loc[6], before + loc[7]};
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool tile_hot = __builtin_amdgcn_ballot_w64(hot) != 0; // a quality byte >= 90 in (or just beside) the window
        if (__builtin_expect(tile_hot, 0)) {
            // rare: Q >= 90 switches the rest of the read to the background error rate (update_likelihood.cpp:40-44);
            // first90[k] = index of the first such byte in read k's quality string
#pragma unroll
            for (int k = 0; k < QCH; ++k) wv_first90(D.qv[k], (uint32_t)k * 512u + lane8, qshift, n_qw, T.n, L.rd, L.first90);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        WV_MARK(3);
        request_qual(Tn, has_next, D); // the next tile's quality bytes take the registers: a whole tile to arrive

        // ---- C: one lane per segment
        double segU[SPASS];
        uint32_t segnode[SPASS];
        bool tile_bep = false; // a segment of the tile takes the background error rate on its own (wave uniform)
        bool tile_out = false; // DIRECT: a segment of the tile lies outside the W window (wave uniform)
        if constexpr (DIRECT) {
            if (need_place) {
#pragma unroll
                for (int k = 0; k < SPASS; ++k) segnode[k] = D.sr[k] & VGAN_HC_SREC_MAX_NODE;
                window_place(segnode, T, winbase);
                need_place = false;
            }
        }
#pragma unroll
        for (int k = 0; k < SPASS; ++k) {
            segU[k] = 0.0;
            segnode[k] = 0u;
            if ((uint32_t)k * 64u < T.n_seg) {
                const uint32_t ls = (uint32_t)k * 64u + (uint32_t)lane;
                const bool on = ls < T.n_seg;
                const uint32_t start = (D.sr[k] >> 18) & 0x7FFu;
                const uint32_t kr = ((D.sr[k] >> 29) - T.r) & 7u;
                static_assert(WV_NR == 8, "the segment records carry the read's index & 7");
                const WvRead rd = L.rd[kr];
                const uint32_t A = rd.a_ql & 0xFFFFu, QL = rd.a_ql >> 16;
                const uint32_t lo = min(start, QL), hi = min(start + A, QL);
                const uint32_t ilo = min(rd.qoff + lo, (uint32_t)CAPQ + 8u), ihi = min(rd.qoff + hi, (uint32_t)CAPQ + 8u);
                const uint32_t pkd = L.ps[ihi + 3u] - L.ps[ilo + 3u];
                // Q5: the bytes beyond the quality string count as Q = 0, i.e. among the "others"
                const uint32_t n_low = (pkd & 2047u) + (A - (hi - lo));
                const double U = fma((double)(pkd >> 11), -0.23025850929940457 /* ln(10) / 10 */,
                                     (double)n_low * -1.3862943611198906 /* log(0.25) */);
                segU[k] = U;
                segnode[k] = D.sr[k] & VGAN_HC_SREC_MAX_NODE;
                if (on) sumU += U;
                // wbg = 1 - pcm, wobs = pcm * match (process_mapping.cpp:41,66-75; a consensus FASTA: 0 and (1 - bep) * match)
                const double pcm = rd.omp * nd_mapp[k];
                const double wbg = a.consensus ? 0.0 : 1.0 - pcm;
                double kappa = a.consensus ? 0.0 : wbg * (rd.ip * nd_inv[k]);
                double lw = rd.lp + nd_lw[k];
                if (!(kappa < 1e300)) { // wobs = 0 (mapping quality 0, mappability 0): the column is log(wbg * bg)
                    kappa = INFINITY;
                    lw = wbg;
                }
                if (__builtin_expect(tile_hot, 0)) {
                    const bool sticky = on && !a.use_bep && L.first90[kr] < hi; // update_likelihood.cpp:42
                    if (sticky) kappa = -kappa;
                    tile_bep = tile_bep || __builtin_amdgcn_ballot_w64(sticky) != 0;
                }
                if (CAPS % 64 == 0 || on) { // (the last pass of a capacity that is not whole passes)
                    L.kl[ls] = WvKL{kappa, lw};
                    if constexpr (!DIRECT) L.S[ls] = 0.0;
                }
                if constexpr (DIRECT) {
                    // -U_m goes to the node's slot here, the columns' terms follow in D
                    const uint32_t node = D.sr[k] & VGAN_HC_SREC_MAX_NODE, sl = node - winbase;
                    const bool inside = sl < (uint32_t)WV_WIN;
                    if (CAPS % 64 == 0 || on) L.slot[ls] = (uint16_t)(inside ? sl * (8u * SUB) : 0xFFFFu);
                    if (on) {
                        if (inside) wv_lds_add(&L.win[sl * SUB], -U);
                        else unsafeAtomicAdd(&a.nodeW[node], -U);
                    }
                    const uint32_t n_out = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(on && !inside));
                    tile_out = tile_out || n_out != 0;
                    if (n_out > 16) need_place = true; // (the next tile places the window anew)
                }
            }
        }
        request_segs(Tn, has_next, D); // (the node ids this tile's E needs are in segnode)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        WV_MARK(4);
        // ---- D: one lane per alignment column, 64 columns per step
        {
            uint32_t segs_before = 0u; // heads in the chunks before this one (scalar)
            // N chunks at a time: every LDS read of the group is issued before the first result is used, so the group pays one
            // LDS round trip (the chains of the chunks interleave); a group shares the branches to the longer series
            auto chunks = [&](auto n_tag, const uint32_t *recs) {
                constexpr int N = decltype(n_tag)::value;
                const WvKL *klp[N];
                const uint16_t *slp[N];
                uint32_t sgl[N]; // (the chunk's first segment, as an index into the batch's segment records: the rare path)
                double *Sp[N];
                uint32_t own[N], row[N], rc8[N];
                uint64_t valid[N], farm[N]; // lane masks (kept as masks: they only ever gate branches)
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    const uint32_t rec = recs[u];
                    const uint64_t heads = __builtin_amdgcn_ballot_w64(rec >= VGAN_HC_CREC_HEAD); // (byte 3 holds nothing else)
                    // owner = heads at or below the lane - 1 = (head bit 0 + heads before the chunk - 1: scalar) + (bits 1..l:
                    // v_mbcnt over the head bits shifted down by one); the scalar part goes into the LDS address
                    uint32_t sbase = segs_before + (uint32_t)(heads & 1u) - 1u;
                    asm volatile("" : "+s"(sbase)); // (kept whole: the -1 otherwise travels into a vector add per chunk)
                    klp[u] = L.kl + sbase;
                    slp[u] = L.slot + (DIRECT ? sbase : 0u);
                    sgl[u] = T.s_base + sbase;
                    Sp[u] = L.S + (DIRECT ? 0u : sbase);
                    const uint64_t above0 = heads >> 1;
                    own[u] = __builtin_amdgcn_mbcnt_hi((uint32_t)(above0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)above0, 0u));
                    segs_before += (uint32_t)__builtin_popcountll(heads);
                    // bases: code (b >> 1) & 3 = A C T G -> 0 1 2 3; a byte is one of the four iff it equals its code's letter.
                    // Both bytes at once: the two codes select their letters out of "ACTG" (v_perm_b32), and the pair of
                    // letters is compared with the pair of bytes (process_mapping.cpp:62-63)
                    rc8[u] = (rec >> 5) & 0x30u;
                    const uint32_t letters = 0x47544341u; // "ACTG"
                    const uint32_t want = __builtin_amdgcn_perm(letters, letters, (rec >> 1) & 0x0303u);
                    uint64_t vmask; // (the compiler has no 16-bit compare of two registers' low halves; SDWA does it in one)
                    asm("v_cmp_eq_u32_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:WORD_0" : "=s"(vmask) : "v"(want), "v"(rec));
                    valid[u] = vmask;
                    int q = __builtin_amdgcn_sbfe((int)rec, 16u, 8u);
                    q = q < 0 ? 0 : (q > 99 ? 99 : q); // qscore_vec's index
                    // table row 2 q + (graph base == read base): the compare writes VCC, the add takes it as carry
                    // (1 - eps on a match: get_p_obs_base.cpp:3-27, :67 with tv = ts = 0)
                    asm("v_cmp_eq_u32_sdwa vcc, %1, %1 src0_sel:BYTE_0 src1_sel:BYTE_1\n\tv_addc_co_u32 %0, vcc, %2, %2, vcc"
                        : "=v"(row[u])
                        : "v"(rec), "v"(q)
                        : "vcc");
                }
                WvKL kl[N];
                uint32_t slv[N];
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    kl[u] = klp[u][own[u]];
                    if constexpr (DIRECT) slv[u] = slp[u][own[u]];
                }
                if (__builtin_expect(tile_bep, 0)) { // a segment behind a quality >= 90: row 100 holds the background error rate
#pragma unroll
                    for (int u = 0; u < N; ++u)
                        if (__double2hiint(kl[u].kappa) < 0) row[u] = 200u + (row[u] & 1u);
                }
                WvLom lo[N];
                double bgv[N], bg6[N], rho[N], l0[N], t[N];
                bool far = false;
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    lo[u] = *reinterpret_cast<const WvLom *>(reinterpret_cast<const uint8_t *>(lom_s) + (row[u] << 4));
                    const double2 b2 = *reinterpret_cast<const double2 *>(reinterpret_cast<const uint8_t *>(bg_s) + rc8[u]);
                    bgv[u] = b2.x;
                    bg6[u] = b2.y;
                }
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    const double ki = fabs(kl[u].kappa) * lo[u].iom;
                    rho[u] = ki * bgv[u];
                    // log1p(rho) by six terms below 2^-8 (the next, rho^7 / 7, is under 2e-18 there).  On HaploCart's own
                    // pairing of graph and read bases (update_likelihood.cpp:46) two columns in three are mismatches, i.e.
                    // rho = kappa * bg / e(Q) ~ 1e-3 for a confidently mapped read: this IS the common case.
                    double p = wv_fma3s(ki, bg6[u], -0.2); // rho / 6 - 1 / 5
                    p = wv_fma3s(rho[u], p, 0.25);
                    p = wv_fma3s(rho[u], p, -1.0 / 3.0);
                    p = fma(rho[u], p, 0.5);
                    p = fma(rho[u], -p, 1.0);
                    l0[u] = lo[u].lom + kl[u].lw;
                    t[u] = fma(rho[u], p, l0[u]);
                    farm[u] = valid[u] & __builtin_amdgcn_ballot_w64(!(rho[u] < 0.00390625));
                    far |= farm[u] != 0;
                }
                WV_COUNT(2, 1);
                if (far) {
                    // Beyond (mapping quality below ~50, low mappability): log1p(rho) = log(u) + (rho - (u - 1)) / u with
                    // u = 1 + rho rounded, the log from the table in LDS (log_tab.h); a segment with wobs = 0 ({inf, wbg}:
                    // mapping quality 0) scores log(wbg * bg)
                    WV_COUNT(3, 1);
#pragma unroll
                    for (int u = 0; u < N; ++u) {
                        // (the lanes of a segment with wobs = 0, or whose rho left the doubles' range, apart: the branch around
                        // their special cases is taken by whole chunks, almost always)
                        const uint64_t odd = farm[u] & __builtin_amdgcn_ballot_w64(!(rho[u] < 1e290));
                        if (__builtin_amdgcn_inverse_ballot_w64(farm[u] & ~odd)) {
                            const double x = 1.0 + rho[u];
                            const double corr = (rho[u] - (x - 1.0)) * __builtin_amdgcn_rcp(x);
                            t[u] = l0[u] + (log_tab_eval_s(x, logtab_s) + corr);
                        }
                        if (odd != 0) {
                            if (__builtin_amdgcn_inverse_ballot_w64(odd)) {
                                const bool deg = !(fabs(kl[u].kappa) < 1e300);
                                double x = deg ? kl[u].lw * bgv[u] : 1.0 + rho[u];
                                const double corr = deg ? 0.0 : (rho[u] - (x - 1.0)) * __builtin_amdgcn_rcp(x);
                                double adj = 0.0;
                                if (x < 2.2250738585072014e-308 && x > 0.0) { // subnormal
                                    x *= 18014398509481984.0;                   // 2^54
                                    adj = -37.429947750237048;                  // -54 ln 2
                                }
                                const double lx = x > 0.0 ? (x <= 1.7976931348623157e308 ? log_tab_eval_s(x, logtab_s) + adj : x)
                                                          : (x == 0.0 ? -INFINITY : __builtin_nan(""));
                                t[u] = deg ? lx : l0[u] + (lx + corr);
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    if (__builtin_amdgcn_inverse_ballot_w64(valid[u])) {
                        if constexpr (DIRECT) {
                            sumT += t[u];
                            double *dst = reinterpret_cast<double *>(reinterpret_cast<uint8_t *>(L.win) + slv[u] + sub8);
                            if (__builtin_expect(tile_out, 0)) {
                                if (slv[u] == 0xFFFFu) unsafeAtomicAdd(&a.nodeW[min(a.srec[sgl[u] + own[u]] & VGAN_HC_SREC_MAX_NODE, a.rows - 1u)], t[u]);
                                else wv_lds_add(dst, t[u]);
                            } else {
                                wv_lds_add(dst, t[u]);
                            }
                        } else {
                            wv_lds_add(&Sp[u][own[u]], t[u]);
                        }
                    }
                }
            };
            static_assert(NCH % WV_GROUP == 0, "whole groups");
            const wv_rsrc rs_cn = rsrc_cols(Tn, has_next);
#pragma unroll
            for (int k = 0; k < NCH; k += WV_GROUP) {
                if constexpr (WV_GROUP == 2) {
                    if (((uint32_t)k + 1u) * 64u < T.n_col) chunks(std::integral_constant<int, 2>{}, &D.rec[k]);
                    else if ((uint32_t)k * 64u < T.n_col) chunks(std::integral_constant<int, 1>{}, &D.rec[k]);
                } else {
                    if ((uint32_t)k * 64u < T.n_col) chunks(std::integral_constant<int, 1>{}, &D.rec[k]);
                }
                // the slots just used take the next tile's columns
#pragma unroll
                for (int u = 0; u < WV_GROUP; ++u) D.rec[k + u] = wv_load1(rs_cn, lane4 + (uint32_t)(k + u) * 256u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        WV_MARK(5);
        // ---- E: one lane per segment (DIRECT: nothing is left to do)
        if (!DIRECT && a.nodeW && need_place) {
            window_place(segnode, T, winbase);
            need_place = false;
        }
#pragma unroll
        for (int k = 0; k < (DIRECT ? 0 : SPASS); ++k) {
            if ((uint32_t)k * 64u < T.n_seg) {
                const uint32_t ls = (uint32_t)k * 64u + (uint32_t)lane;
                const bool on = ls < T.n_seg;
                const double S = L.S[ls];
                const double Dm = S - segU[k];
                bool outside = false;
                if (on) {
                    sumT += S;
                    if (a.segD_out) a.segD_out[T.s_base + ls] = Dm;
                    if (a.nodeW) {
                        const uint32_t slot = segnode[k] - winbase;
                        if (slot < (uint32_t)WV_WIN) {
                            wv_lds_add(&L.win[slot], Dm);
                        } else {
                            unsafeAtomicAdd(&a.nodeW[segnode[k]], Dm);
                            outside = true;
                        }
                    }
                }
                if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(outside)) > 16) need_place = true;
            }
        }
        WV_MARK(6);
#ifdef WV_PHASES
        ph_acc[7] += 1;
#endif
        if (!has_next) break;
        T = Tn;
        h_q = hn_q;
        h_am = hn_am;
        fn = f2;
        wn = w2;
        fresh = fresh_n;
        fresh_n = fresh_2;
    }
    if ((DIRECT || a.nodeW) && winbase != 0xFFFFFFFFu) window_flush(winbase);
#ifdef WV_PHASES
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&wv_phase_cycles[i], ph_acc[i]);
#endif
#ifdef WV_SPANS
    if (lane == 0) {
        const uint32_t gw = (blockIdx.x * WV_WAVES + (uint32_t)wave) & 16383u;
        wv_wave_span[4 * gw] = span_t0;
        wv_wave_span[4 * gw + 1] = __builtin_amdgcn_s_memrealtime();
        wv_wave_span[4 * gw + 2] = span_tiles;
        wv_wave_span[4 * gw + 3] = span_last;
    }
#endif
    sumT = wave_sum(sumT);
    sumU = wave_sum(sumU);
    if (lane == 0 && a.totals) {
        double *t = a.totals + ((blockIdx.x * WV_WAVES + (uint32_t)wave) % HC_TOTAL_SLOTS) * HC_TOTAL_STRIDE;
        unsafeAtomicAdd(&t[0], sumT);
        unsafeAtomicAdd(&t[1], sumU);
    }
}

// ------------------------------------------------------------------------------------------------------------ layout pass
// One wave per read.  Lanes over its segments: the segment records, and for every column a segment scores its seg_start + 1
// in LDS; then lanes over its columns: the column records, read and written in coalesced runs.
constexpr int PK_COLS = 1280; // a tileable read's columns at most (hc_device.h: HC_TILE_MAX_READ_COLS)
__global__ __launch_bounds__(256) void hc_pack_kernel(HcBatchDev b, uint32_t n_pack, uint4 *__restrict__ rhdr,
                                                      uint32_t *__restrict__ srec, uint32_t *__restrict__ crec,
                                                      uint32_t *__restrict__ maxima) {
    __shared__ uint16_t own_s[4][PK_COLS];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *own = own_s[wave];
    // the batch's maxima: kept per wave and sent once (an atomic per read on the same three words serialises the whole launch:
    // 3 M same-address atomics at ~14 ns were 34 of the pass's 35 ms)
    uint32_t mx_s = 0u, mx_q = 0u, mx_c = 0u, mx_x = 0u;
    for (uint32_t r = blockIdx.x * 4u + wave; r <= n_pack; r += gridDim.x * 4u) { // (a wave per read would be a million waves)
        if (r == n_pack) { // the end offsets
            if (lane == 0) rhdr[r] = uint4{b.read_seg_off[r], b.read_qual_off[r], b.read_col_off[r], 0u};
            break;
        }
        const uint32_t s0 = b.read_seg_off[r], s1 = b.read_seg_off[r + 1];
        const uint32_t c0 = b.read_col_off[r], c1 = b.read_col_off[r + 1];
        const uint32_t q0 = b.read_qual_off[r], q1 = b.read_qual_off[r + 1];
        const uint32_t A = b.read_algn_len[r];
        if (lane == 0) rhdr[r] = uint4{s0, q0, c0, A | ((uint32_t)b.read_mapq[r] << 16)};
        mx_s = max(mx_s, s1 - s0);
        mx_q = max(mx_q, q1 - q0);
        mx_c = max(mx_c, c1 - c0);
        mx_x = max(mx_x, (q1 - q0) > (c1 - c0) ? (q1 - q0) - (c1 - c0) : 0u);
        const uint32_t cols = min(c1 - c0, (uint32_t)PK_COLS), QL = q1 - q0; // (a read beyond the tile contract: a caller's error)
        for (uint32_t c = lane; c < cols; c += 64u) own[c] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (uint32_t s = s0 + lane; s < s1; s += 64u) {
            const uint32_t start = b.seg_start[s], len = b.seg_len[s];
            srec[s] = VGAN_HC_SREC(b.seg_node[s], start, r);
            const uint32_t cl = start < cols ? min(len, cols - start) : 0u;
            for (uint32_t j = 0; j < cl; ++j) own[start + j] = (uint16_t)(start + 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t c = lane; c < cols; c += 64u) {
            const uint32_t o = own[c];
            uint32_t rec = (c < QL ? (uint32_t)b.qual[q0 + c] : 0u) << 16; // the quality byte on every column (zero beyond the string)
            if (o) { // (a column no segment scores keeps that byte alone)
                const uint32_t j = c - (o - 1u);
                const uint32_t gb = b.graph_seq[c0 + c];
                const uint32_t rb = j < A ? b.algnseq[c0 + j] : 0u; // read bases from the read start (update_likelihood.cpp:46)
                rec |= gb | (rb << 8) | (j == 0 ? VGAN_HC_CREC_HEAD : 0u);
            }
            crec[c0 + c] = rec;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the next read's marks come after these reads of own[])
    }
    if (maxima && lane == 0) {
        if (mx_s) atomicMax(&maxima[0], mx_s);
        if (mx_q) atomicMax(&maxima[1], mx_q);
        if (mx_c) atomicMax(&maxima[2], mx_c);
        if (mx_x) atomicMax(&maxima[3], mx_x);
    }
}

__global__ __launch_bounds__(256) void hc_srec_nodes_kernel(const uint32_t *__restrict__ srec, uint32_t n, uint32_t *__restrict__ out) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) out[i] = srec[i] & VGAN_HC_SREC_MAX_NODE;
}

} // namespace wv
using namespace wv;

#ifdef WV_SPANS
extern "C" int vgan_hc_debug_wave_spans(unsigned long long *out /* 4 * 16384 */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(wv_wave_span), sizeof(wv_wave_span)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef WV_PHASES
extern "C" int vgan_hc_debug_wave_phases(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(wv_phase_cycles), sizeof(wv_phase_cycles)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(wv_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef WV_STATS
extern "C" int vgan_hc_debug_wave_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(wv_stats), sizeof(wv_stats)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(wv_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// Which reads the wave kernel takes.  Its tile holds WV_CAPS segments / WV_CAPQ quality bytes / WV_CAPC columns of at most
// WV_NR reads: the kernel is built for tiles of several reads (150 bp: three, 75 bp: six).  A batch whose reads are so long
// that a tile holds one (300 bp) or so short that WV_NR of them leave the tile half empty (40 bp) runs faster on the LDS-tiled
// kernel (measured, 1M reads: 300 bp 1.18 against 0.87 ms, 40 bp 0.39 against 0.29 ms; 150 bp 0.57 against 0.89, 75 bp 0.40
// against 0.48), and so does one holding a read beyond the tile's capacity.  VGAN_HC_KERNEL=wave takes the kernel whenever a
// variant of it can hold the batch's reads (developer aid; the tests use it to force the large variant).
bool hc_wave_kernel_fits(uint32_t max_read_segs, uint32_t max_read_qual, uint32_t max_read_cols, uint32_t mean_read_segs,
                         uint32_t mean_read_cols) {
    if (max_read_segs > 512u || max_read_qual > 1280u || max_read_cols > 1280u) return false;
    const char *e = getenv("VGAN_HC_KERNEL");
    if (e && strcmp(e, "wave") == 0) return true;
    if (max_read_segs > (uint32_t)WV_CAPS || max_read_qual > (uint32_t)WV_CAPQ || max_read_cols > (uint32_t)WV_CAPC) return false;
    return 2u * mean_read_segs <= (uint32_t)WV_CAPS && 2u * mean_read_cols <= (uint32_t)WV_CAPC && mean_read_cols * (uint32_t)WV_NR >= 384u;
}

void launch_hc_pack(const HcBatchDev &b, uint32_t n_tileable, uint64_t n_cols, uint64_t n_qual, uint4 *rhdr, uint32_t *srec,
                    uint32_t *crec, uint8_t *qualp, uint32_t *maxima, hipStream_t st) {
    const uint32_t n = std::min(n_tileable, b.n_reads);
    if (maxima) (void)hipMemsetAsync(maxima, 0, 16, st);
    if (n_qual) (void)hipMemcpyAsync(qualp, b.qual, n_qual, hipMemcpyDeviceToDevice, st);
    (void)hipMemsetAsync(qualp + n_qual, 0, 32, st);
    hipLaunchKernelGGL(hc_pack_kernel, dim3(std::min<uint32_t>((n + 1 + 3) / 4, 8192u)), dim3(256), 0, st, b, n, rhdr, srec, crec, maxima);
}

void launch_hc_srec_nodes(const uint32_t *srec, uint32_t n_segments, uint32_t *out, hipStream_t st) {
    if (n_segments == 0) return;
    hipLaunchKernelGGL(hc_srec_nodes_kernel, dim3(std::min<uint32_t>((n_segments + 255u) / 256u, 4096u)), dim3(256), 0, st, srec, n_segments, out);
}

void launch_hc_segments_wave(const HcGraphDev &g, const HcPackedDev &pk, const HcParamsDev &prm, double *segD, double *nodeW,
                             double *totals, uint32_t *work_ctr, uint32_t *work_base, hipStream_t st) {
    if (pk.n_reads == 0) return;
    const bool small = pk.max_read_segs <= (uint32_t)WV_CAPS && pk.max_read_qual <= (uint32_t)WV_CAPQ && pk.max_read_cols <= (uint32_t)WV_CAPC;
    // a persistent grid: as many workgroups as the chip holds at once, fed by the work queue
    // (keyed on the device the launch goes to: contexts of one process may sit on different GPUs)
    static int n_cu_dev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (n_cu_dev[dev] <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n_cu_dev[dev] = n;
    }
    const int n_cu = n_cu_dev[dev];
    constexpr int OCC_S = WV_OCC, TS = WV_THREADS_SMALL, TL = WV_THREADS_LARGE;
    using KS = void (*)(WvArgs);
    const KS k_small_direct = hc_segment_wave_kernel<WV_CAPS, WV_CAPQ, WV_CAPC, true, TS, OCC_S>;
    const KS k_small = hc_segment_wave_kernel<WV_CAPS, WV_CAPQ, WV_CAPC, false, TS, OCC_S>;
    const KS k_large_direct = hc_segment_wave_kernel<512, 1280, 1280, true, TL, 1>;
    const KS k_large = hc_segment_wave_kernel<512, 1280, 1280, false, TL, 1>;
    // Workgroups a CU really holds at once.  The occupancy API overstates it on gfx950: LDS is handed out in granules of
    // 1280 bytes (128 per CU), so five workgroups fit only at 32 000 bytes each and below, four at 40 960 (measured with
    // tools/dev/census.hip: 32 256 bytes per workgroup run four to a CU where the API says five).
    auto resident = [](KS kern, int threads, int fallback) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, 0) != hipSuccess || n <= 0) n = fallback;
        hipFuncAttributes fa{};
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)) == hipSuccess && fa.sharedSizeBytes > 0)
            n = std::min<int>(n, 128 / (int)((fa.sharedSizeBytes + 1279) / 1280));
        return std::max(1, n);
    };
    static const int occ_small = [&] {
        int n = resident(k_small_direct, TS, OCC_S * 256 / TS);
        if (const char *e = getenv("VGAN_WV_BLOCKS_PER_CU")) n = std::max(1, atoi(e)); // developer aid
        return n;
    }();
    static const int occ_large = [&] { return resident(k_large, TL, 1); }();
    const int WV_WAVES = (small ? TS : TL) / 64;
    uint32_t unit = small ? 32u : 8u; // reads per work unit: ~16 tiles -- the price of a ticket against the length of the launch's tail
    if (const char *e = getenv("VGAN_WV_UNIT")) unit = (uint32_t)std::max(1, atoi(e)); // developer aid
    uint32_t blocks = (uint32_t)(n_cu * (small ? occ_small : occ_large));
    if (const char *e = getenv("VGAN_WV_WG_PER_CU")) { // developer aid: the fixed partition into that many workgroups per CU, no queue
        blocks = (uint32_t)(n_cu * std::max(1, atoi(e)));
        unit = std::max(8u, (pk.n_reads + blocks * WV_WAVES - 1) / (blocks * WV_WAVES));
        work_ctr = nullptr;
    }
    const uint32_t n_units = (pk.n_reads + unit - 1) / unit;
    blocks = std::min(blocks, (n_units + WV_WAVES - 1) / WV_WAVES);
    const uint32_t n_waves = blocks * WV_WAVES;
    if (work_ctr && *work_base > 0xC0000000u) { // (the counter runs on from launch to launch; long before it wraps it starts over)
        (void)hipMemsetAsync(work_ctr, 0, 4, st);
        *work_base = 0;
    }
    WvArgs a{};
    a.rhdr = pk.rhdr;
    a.srec = pk.srec;
    a.crec = pk.crec;
    a.qualp = pk.qualp;
    a.node_tab = g.node_tab;
    a.qscore = g.qscore;
    a.rdtab = g.rdtab;
    a.segD_out = segD;
    a.nodeW = nodeW;
    a.totals = totals;
    a.bep = prm.bep;
    a.work = work_ctr;
    // tickets: the first n_waves units are the waves' own, every further unit is one ticket, and every wave draws one more to
    // learn that the queue is empty (the grid never has more waves than units)
    a.work_base = *work_base - n_waves;
    if (work_ctr) *work_base += n_units;
    a.n_reads = pk.n_reads;
    a.rows = g.rows;
    a.unit_reads = unit;
    a.qual_bytes = (uint32_t)std::min<uint64_t>(0xFFFFFFF0u, pk.n_qual + 32u);
    a.use_bep = prm.use_bep ? 1u : 0u;
    a.consensus = prm.consensus ? 1u : 0u;
    // node-weights accumulation alone: the columns add straight into the W window (no per-segment sums); D_m streamed out
    // (the per-read modes, the test aids): the per-segment form
    // ... when the reads fit the W window (a read spanning more node ids than the window has slots sends its outlying
    // segments' COLUMNS to HBM one by one there; the per-segment form sends one sum per outlying segment: 300 bp reads, 212 ids:
    // 3.4 against 1.2 ms per 500 k).  The span as the flatten step measured it, else judged by the mean read length.
    const uint32_t span = pk.max_read_node_span ? pk.max_read_node_span : (uint32_t)(pk.n_cols / std::max<uint32_t>(1, pk.n_reads)) * 3u / 4u;
    const bool direct = nodeW && !segD && span < (uint32_t)WV_WIN && !getenv("VGAN_WV_NO_DIRECT");
    const KS kern = small ? (direct ? k_small_direct : k_small) : (direct ? k_large_direct : k_large);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(small ? TS : TL), 0, st, a);
}

} // namespace vgan
#include "module_anchor.h"
const void *vgan::anchor_hc_wave() { return (const void *)&vgan::wv::hc_srec_nodes_kernel; }
