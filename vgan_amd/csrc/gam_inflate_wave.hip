// DEFLATE (RFC 1951) of BGZF members in two kernels (reference: src/readGAM.h:20-68 through libvgio's BGZF stream):
//
//   gd_tokens_kernel  a WAVE per member: the Huffman half.  The member's code tables are built by all lanes (direct tables of 2^10 /
//                     2^9 entries in LDS, the canonical compare-against-limits decode of gam_kernels.hip for the entries and for the rare
//                     longer codes); then the 64 lanes decode 64 chunks of the block's bits AT ONCE -- lane 0 from the block's first
//                     code, the others from an arbitrary bit: a Huffman stream synchronises itself (a decode begun inside a code
//                     stands on true code boundaries after ~11 codes on GAM data), so the end of lane i's chunk is almost always right
//                     even when its start was not.  Then every lane restarts from where the lane in front ended, and again until no start
//                     moves (twice, as a rule): by induction from lane 0 every chunk is then decoded from a true boundary.  A last pass
//                     writes the codes out as 32-bit TOKENS (a match {length, distance}, or up to three literals) into a scratch region
//                     claimed with one atomic add per block.
//   gd_lz_kernel      a LANE per member: the LZ77 half over the tokens -- a short uniform loop (a token, or eight bytes of a match) with
//                     the output gathered in registers and stored in aligned 8-byte words, as gd_inflate_kernel's.  (Copies with
//                     sources a few bytes back are a chain through memory that no wave-wide trick shortens for GAM data, whose matches
//                     are ~6 bytes long and mostly reach back one mapping: ~15 bytes.)
//
// gd_inflate_kernel (gam_kernels.hip: a lane per member for both halves, ~350 vector instructions a step with every path of its loop
// taken by some lane) stays as the reference: whatever a member's status is other than GD_OK after these two kernels -- a stored block,
// more blocks than a member has regions for, no room in the scratch, an error -- the member is done again by it, so an error is only
// ever reported by the older kernel.  Byte work: the output is zlib's, bit for bit (tests/test_gamdev_gpu.py, test_gampipe_gpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>

#include "gam_device.h"
#include "gam_object.h"
#include "host/common.h"
#include "vgan_gpu.h"

using namespace vgan;

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(VGAN_ENODEV, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace vgan {
namespace gd {

namespace {
__device__ const uint16_t gw_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t gw_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t gw_dist_base[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                              193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t gw_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t gw_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct __attribute__((packed)) GwU64 {
    uint64_t v;
};
// the stream's bits from bit `p` on (57 of them at least), LSB first
__device__ __forceinline__ uint64_t gw_bits(const uint8_t *pay, uint32_t p) { return reinterpret_cast<const GwU64 *>(pay + (p >> 3))->v >> (p & 7u); }

constexpr uint32_t GW_TB = 9, GW_DB = 9;   // bits the direct tables are indexed by
constexpr uint32_t GW_MAX_REGIONS = 4;
constexpr uint32_t GW_SPEC = 1280; // bits a speculative decode runs before its end is taken for a true boundary (~11 codes of ~13 bits are the mean)

// One alphabet's canonical code (RFC 1951 3.2.2) in LDS: limit[L] = (first code of length L + their count) << (15 - L), base[L] =
// (symbols of shorter codes) - (first code of length L), the symbols in code order -- gam_kernels.hip: Dec, dec_len, dec_build.
struct GwCode {
    uint32_t lim[16];
    int32_t base[16];
    uint32_t n_coded;
};
// length of the code in front of `pk` (the next 15 bits, first bit most significant): 1..15, or 16: none
__device__ __forceinline__ uint32_t gw_code_len(const GwCode &c, uint32_t pk) {
    uint32_t n = 1;
#pragma unroll
    for (int L = 1; L <= 15; ++L) n += pk >= c.lim[L] ? 1u : 0u;
    return n;
}
// lens[0..n) -> c, sorted[] (the symbols in code order); every lane runs it (n <= 320); false: an over-subscribed set
template <class SymT> __device__ bool gw_build(const uint8_t *lens, uint32_t n, GwCode &c, SymT *sorted, uint32_t lane) {
    uint32_t cnt[16];
#pragma unroll
    for (int L = 0; L < 16; ++L) cnt[L] = 0;
    for (uint32_t s0 = 0; s0 < n; s0 += 64u) {
        const uint32_t s = s0 + lane, my = s < n ? lens[s] : 0u;
#pragma unroll
        for (uint32_t L = 1; L <= 15; ++L) cnt[L] += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(my == L));
    }
    int left = 1;
    uint32_t offs[16], code = 0, idx = 0;
    bool ok = true;
    offs[0] = 0;
#pragma unroll
    for (uint32_t L = 1; L <= 15; ++L) {
        left = (left << 1) - (int)cnt[L];
        if (left < 0) ok = false;
        offs[L] = idx;
        if (lane == 0) {
            c.base[L] = (int32_t)idx - (int32_t)code;
            c.lim[L] = (code + cnt[L]) << (15u - L);
        }
        code = (code + cnt[L]) << 1;
        idx += cnt[L];
    }
    if (lane == 0) c.n_coded = idx;
    if (!ok) return false;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (uint32_t s0 = 0; s0 < n; s0 += 64u) {
        const uint32_t s = s0 + lane, my = s < n ? lens[s] : 0u;
        uint32_t at = 0;
#pragma unroll
        for (uint32_t L = 1; L <= 15; ++L) {
            const uint64_t m = __builtin_amdgcn_ballot_w64(my == L);
            if (my == L) at = offs[L] + (uint32_t)__builtin_popcountll(m & lt);
            offs[L] += (uint32_t)__builtin_popcountll(m);
        }
        if (my) sorted[at] = (SymT)s;
    }
    return true;
}
// Direct tables hold READY entries: everything a code says, in one word.
//   literal / length alphabet:  bits 0-3 the code's length, 4-7 its extra bits, 8-16 the literal byte or the length's base, 20 / 21 / 22: a
//                               literal / the end-of-block code / a length
//   distance alphabet:          bits 0-3 the code's length, 4-7 its extra bits, 8-22 the distance's base, 30: a distance
//   the code-length alphabet:   bits 0-3 the code's length, 8-16 the symbol
// 0: no code begins with these bits; GW_LONG: a code longer than the table's index (the canonical decode makes its entry).
constexpr uint32_t GW_LONG = 0x80000000u, GW_LIT = 1u << 20, GW_EOB = 1u << 21, GW_MATCH = 1u << 22, GW_DIST = 1u << 30;
enum { GW_ALPHA_LIT = 0, GW_ALPHA_DIST = 1, GW_ALPHA_CL = 2 };
template <int ALPHA> __device__ __forceinline__ uint32_t gw_entry(uint32_t sym, uint32_t len, const uint32_t *len_tab, const uint32_t *dist_tab) {
    if (ALPHA == GW_ALPHA_CL) return len | (sym << 8);
    if (ALPHA == GW_ALPHA_DIST) {
        if (sym >= 30u) return 0u;
        const uint32_t dt = dist_tab[sym];
        return len | ((dt >> 16) << 4) | ((dt & 0xFFFFu) << 8) | GW_DIST;
    }
    if (sym < 256u) return len | (sym << 8) | GW_LIT;
    if (sym == 256u) return len | GW_EOB;
    if (sym >= 286u) return 0u;
    const uint32_t lt = len_tab[sym - 257u];
    return len | ((lt >> 16) << 4) | ((lt & 0xFFFFu) << 8) | GW_MATCH;
}
template <int ALPHA, class SymT>
__device__ void gw_fill(const GwCode &c, const SymT *sorted, uint32_t *tab, uint32_t bits, uint32_t lane, const uint32_t *len_tab, const uint32_t *dist_tab) {
    for (uint32_t idx = lane; idx < (1u << bits); idx += 64u) {
        const uint32_t pk = __builtin_bitreverse32(idx) >> 17;
        const uint32_t len = gw_code_len(c, pk);
        uint32_t e = 0;
        if (len <= bits) {
            const uint32_t at = (uint32_t)(c.base[len] + (int32_t)(pk >> (15u - len)));
            e = at < c.n_coded ? gw_entry<ALPHA>((uint32_t)sorted[at], len, len_tab, dist_tab) : 0u;
        } else if (len <= 15u) {
            e = GW_LONG;
        }
        tab[idx] = e;
    }
}
// Codes longer than a table's index: a SECOND table per index whose code is longer, the two of an alphabet's and both alphabets' out of
// one pool.  The first table's entry then says where the second lies and how many more bits index it (GW_LONG | offset << 8 | bits); a
// second table's entries are ready entries (their length: the whole code's).  Canonical codes grow with their left-aligned value, so
// the longest code behind an index is the one in front of the index's LARGEST 15-bit continuation.  All lanes fill one second table at
// a time (64 entries at most: codes are 15 bits at most, the first tables take 9).  False: the pool is full (the older kernel's member).
// (A wave decodes 64 chunks in step: a path for long codes that ONE lane takes is a path the wave takes, in most steps -- it must cost
// a table look-up, not a canonical decode.)
constexpr uint32_t GW_POOL = 768;
template <int ALPHA, class SymT>
__device__ bool gw_fill_long(const GwCode &c, const SymT *sorted, uint32_t *tab, uint32_t bits, uint32_t *pool, uint32_t &pool_n, uint32_t lane, const uint32_t *len_tab,
                             const uint32_t *dist_tab) {
    for (uint32_t i0 = 0; i0 < (1u << bits); i0 += 64u) {
        uint64_t todo = __builtin_amdgcn_ballot_w64(tab[i0 + lane] == GW_LONG);
        while (todo) {
            const uint32_t idx = i0 + (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t pk0 = __builtin_bitreverse32(idx) >> 17; // (its low 15 - bits bits are zero)
            const uint32_t lmax = min(gw_code_len(c, pk0 | ((1u << (15u - bits)) - 1u)), 15u), sb = lmax - bits;
            if (pool_n + (1u << sb) > GW_POOL) return false;
            if (lane < (1u << sb)) {
                const uint32_t pk = __builtin_bitreverse32(idx | (lane << bits)) >> 17;
                const uint32_t len = gw_code_len(c, pk);
                uint32_t e = 0;
                if (len <= lmax) {
                    const uint32_t at = (uint32_t)(c.base[len] + (int32_t)(pk >> (15u - len)));
                    if (at < c.n_coded) e = gw_entry<ALPHA>((uint32_t)sorted[at], len, len_tab, dist_tab);
                }
                pool[pool_n + lane] = e;
            }
            if (lane == 0) tab[idx] = GW_LONG | (pool_n << 8) | sb;
            pool_n += 1u << sb;
        }
    }
    return true;
}
// the ready entry of the code at the low bits of w (two look-ups for a long code)
__device__ __forceinline__ uint32_t gw_lookup(const uint32_t *tab, uint32_t bits, const uint32_t *pool, uint32_t w) {
    uint32_t e = tab[w & ((1u << bits) - 1u)];
    if (e & GW_LONG) e = pool[((e >> 8) & 0xFFFFu) + __builtin_amdgcn_ubfe(w, bits, e & 15u)];
    return e;
}

struct GwHeader { // what a block's header is read with: done with when the two alphabets' codes are built
    uint8_t lens[320];
    uint8_t cl[20];
    uint8_t csym[20];
    GwCode cc;
    uint32_t ct[128];
};
struct GwShared {
    union { // (the literal / length table is filled when the header's arrays are done with)
        GwHeader h;
        uint32_t lt[1u << GW_TB];
    };
    uint32_t dt[1u << GW_DB];
    uint32_t pool[GW_POOL]; // the second tables of both alphabets
    uint16_t lsym[288];
    uint8_t dsym[32];
    GwCode lc, dc;
    uint32_t len_tab[32], dist_tab[32]; // base | extra bits << 16
};
static_assert(sizeof(GwHeader) <= sizeof(uint32_t) << GW_TB, "the header's arrays borrow the literal / length table");

// The stream's bits through a window of three aligned 8-byte words, the third asked for two words ahead of the bits in use: a code's
// bits are in registers when the code before it is done (the address of a load that fetched them only then would depend on that code).
struct GwWin {
    const uint64_t *q; // the word behind nx
    uint64_t lo, hi, nx;
    int32_t base; // the bit position (in the payload) of lo's bit 0
};
__device__ __forceinline__ void gw_win_init(GwWin &b, const uint8_t *pay, uint32_t p) {
    const uintptr_t a = (uintptr_t)pay + (p >> 3), a8 = a & ~(uintptr_t)7;
    const uint64_t *q = reinterpret_cast<const uint64_t *>(a8);
    b.lo = q[0];
    b.hi = q[1];
    b.nx = q[2];
    b.q = q + 3;
    b.base = (int32_t)(((intptr_t)a8 - (intptr_t)pay) * 8);
}
// 57 bits and more from bit p on (p within 64 bits of the window's base, as it is behind gw_win_to)
__device__ __forceinline__ uint64_t gw_win_peek(const GwWin &b, uint32_t p) {
    const uint32_t sft = (uint32_t)((int32_t)p - b.base);
    return (b.lo >> sft) | ((b.hi << 1) << (63u - sft));
}
__device__ __forceinline__ void gw_win_to(GwWin &b, uint32_t p) { // p moved on by less than 64 bits
    if ((int32_t)p - b.base >= 64) {
        b.lo = b.hi;
        b.hi = b.nx;
        b.nx = *b.q++;
        b.base += 64;
    }
}

// One chunk of the block's codes, from bit t to the first code boundary at or behind `limit`, or behind the end-of-block code when one
// comes first (flags & 1).  A chunk begun at an arbitrary bit meets bits that are no code (an alphabet that is not complete): it goes on
// one bit further, as any other start would do -- flags & 2 says so, and counts only once the chunk's start is known to be true.  EMIT:
// the tokens go to out[]; either way their number (a token per match, one per three literals of a run), and the bytes they stand for.
template <bool EMIT>
__device__ __forceinline__ uint32_t gw_chunk(const GwShared &sh, const uint8_t *pay, uint32_t t, uint32_t limit, uint32_t end_bits, uint32_t *out, uint32_t &n_tok,
                                             uint32_t &n_bytes, uint32_t &flags) {
    uint32_t p = t, ntok = 0, nb = 0, run = 0, run_n = 0, fl = 0;
    const uint32_t stop = min(limit, end_bits); // (bits beyond the payload: nothing of this block)
    GwWin br;
    gw_win_init(br, pay, min(t, end_bits));
    while (p < stop) {
        const uint64_t w = gw_win_peek(br, p);
        const uint32_t w0 = (uint32_t)w;
        const uint32_t e = gw_lookup(sh.lt, GW_TB, sh.pool, w0);
        const uint32_t len = e & 15u, xb = (e >> 4) & 15u, val = (e >> 8) & 511u;
        uint32_t used = len + xb;
        bool ok = (e & (GW_LIT | GW_EOB | GW_MATCH)) != 0u;
        const bool lit = (e & GW_LIT) != 0u;
        if (e & GW_MATCH) {
            // (the length's extra bits lie within the first 20 bits; the distance code and its extra bits, 28 at most, behind them)
            const uint32_t L = val + __builtin_amdgcn_ubfe(w0, len, xb);
            const uint32_t w2 = (uint32_t)(w >> used);
            const uint32_t de = gw_lookup(sh.dt, GW_DB, sh.pool, w2);
            const uint32_t dl = de & 15u, dxb = (de >> 4) & 15u;
            ok = (de & GW_DIST) != 0u;
            if (ok) {
                if (EMIT) {
                    if (run_n) { // the literals that wait go first
                        out[ntok] = run | (run_n << 30);
                        ntok += 1;
                        run = 0;
                    }
                    const uint32_t D = ((de >> 8) & 0x7FFFu) + __builtin_amdgcn_ubfe(w2, dl, dxb);
                    out[ntok] = L | ((D - 1u) << 9);
                }
                run_n = 0;
                ntok += 1;
                nb += L;
                used += dl + dxb;
            }
        }
        if (!ok) { // no code here: one bit further
            fl |= 2u;
            used = 1;
        } else if (lit) {
            if (EMIT) {
                run |= val << (8u * run_n);
                if (run_n == 2u) {
                    out[ntok] = run | (3u << 30);
                    ntok += 1;
                    run = 0;
                }
            } else {
                ntok += run_n == 0u ? 1u : 0u; // (a run's token is counted when the run opens)
            }
            run_n = run_n == 2u ? 0u : run_n + 1u;
            nb += 1;
        }
        p += used;
        gw_win_to(br, p);
        if (e & GW_EOB) {
            fl |= 1u;
            break;
        }
    }
    if (p >= end_bits && !(fl & 1u)) fl |= 4u;
    if (EMIT && run_n) {
        out[ntok] = run | (run_n << 30);
        ntok += 1;
    }
    n_tok = ntok;
    n_bytes = nb;
    flags = fl;
    return p;
}

__device__ __forceinline__ uint32_t gw_scan_incl(uint32_t v, uint32_t lane) {
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d);
        if (lane >= d) v += o;
    }
    return v;
}
} // namespace

// status[b]: GD_OK, GD_PUNT (the older kernel does the member), or what is wrong with it.  reg[b * 4 + k] = {first token, tokens} of the
// member's k-th block, n_reg[b] their number.
__global__ __launch_bounds__(64, 8) void gd_tokens_kernel(const uint8_t *__restrict__ in, const GdBlock *__restrict__ blocks, uint32_t n_blocks, uint32_t *__restrict__ tok,
                                                       uint32_t tok_cap, uint32_t *__restrict__ cursor, uint2 *__restrict__ reg, uint32_t *__restrict__ n_reg,
                                                       uint32_t *__restrict__ status) {
    __shared__ GwShared sh;
    const uint32_t lane = threadIdx.x, b = blockIdx.x;
    if (b >= n_blocks) return;
    if (lane < 29) sh.len_tab[lane] = gw_len_base[lane] | ((uint32_t)gw_len_extra[lane] << 16);
    if (lane < 30) sh.dist_tab[lane] = gw_dist_base[lane] | ((uint32_t)gw_dist_extra[lane] << 16);
    const GdBlock bl = blocks[b];
    const uint8_t *pay = in + bl.in_off;
    const uint32_t end_bits = bl.in_size * 8u;
    uint32_t pos = 0, st = GD_OK, n_regions = 0, total_bytes = 0, rounds = 0;
    bool last = false;
    while (!last && st == GD_OK) {
        if (pos + 3u > end_bits) {
            st = GD_OVERRUN_IN;
            break;
        }
        uint64_t w = gw_bits(pay, pos);
        last = (w & 1u) != 0;
        const uint32_t type = (uint32_t)(w >> 1) & 3u;
        pos += 3;
        if (type == 0u || type == 3u || n_regions >= GW_MAX_REGIONS) { // stored blocks, many blocks: the older kernel's
            st = GD_PUNT;
            break;
        }
        uint32_t nlen = 288, ndist = 30;
        __syncthreads();
        if (type == 1u) {
            for (uint32_t s = lane; s < 320u; s += 64u) sh.h.lens[s] = s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : s < 288u ? 8 : s < 318u ? 5 : 0;
        } else {
            w = gw_bits(pay, pos);
            nlen = ((uint32_t)w & 31u) + 257u;
            ndist = ((uint32_t)(w >> 5) & 31u) + 1u;
            const uint32_t ncode = ((uint32_t)(w >> 10) & 15u) + 4u;
            pos += 14;
            if (nlen > 286u || ndist > 30u || pos + 3u * ncode > end_bits) {
                st = GD_BAD_BLOCK;
                break;
            }
            if (lane < 19) sh.h.cl[lane] = 0;
            __syncthreads();
            if (lane < ncode) sh.h.cl[gw_clen_order[lane]] = (uint8_t)(gw_bits(pay, pos + 3u * lane) & 7u);
            pos += 3u * ncode;
            __syncthreads();
            if (!gw_build(sh.h.cl, 19u, sh.h.cc, sh.h.csym, lane)) {
                st = GD_BAD_BLOCK;
                break;
            }
            __syncthreads();
            gw_fill<GW_ALPHA_CL>(sh.h.cc, sh.h.csym, sh.h.ct, 7u, lane, sh.len_tab, sh.dist_tab);
            __syncthreads();
            // the code lengths of the two alphabets: a serial walk (every lane the same), runs written by all lanes
            uint32_t idx = 0, prev = 0;
            const uint32_t want = nlen + ndist;
            GwWin cw;
            gw_win_init(cw, pay, pos);
            while (idx < want) {
                if (pos > end_bits) break;
                gw_win_to(cw, pos);
                const uint64_t v = gw_win_peek(cw, pos);
                const uint32_t e = sh.h.ct[(uint32_t)v & 127u], cl = e & 15u, s = (e >> 8) & 511u;
                if (cl == 0u || (e & GW_LONG)) {
                    st = GD_BAD_CODE;
                    break;
                }
                pos += cl;
                if (s < 16u) {
                    if (lane == 0) sh.h.lens[idx] = (uint8_t)s;
                    prev = s;
                    idx += 1;
                    continue;
                }
                uint32_t rep, val = 0;
                if (s == 16u) {
                    if (idx == 0) {
                        st = GD_BAD_BLOCK;
                        break;
                    }
                    val = prev;
                    rep = 3u + ((uint32_t)(v >> cl) & 3u);
                    pos += 2;
                } else if (s == 17u) {
                    rep = 3u + ((uint32_t)(v >> cl) & 7u);
                    pos += 3;
                } else {
                    rep = 11u + ((uint32_t)(v >> cl) & 127u);
                    pos += 7;
                }
                if (idx + rep > want) {
                    st = GD_BAD_BLOCK;
                    break;
                }
                for (uint32_t k = lane; k < rep; k += 64u) sh.h.lens[idx + k] = (uint8_t)val;
                prev = val;
                idx += rep;
            }
            if (st != GD_OK) break;
            if (idx < want || pos > end_bits) {
                st = GD_OVERRUN_IN;
                break;
            }
            __syncthreads();
            if (sh.h.lens[256] == 0) {
                st = GD_BAD_BLOCK;
                break;
            }
            // the distance lengths behind the literal / length ones, at a fixed place
            uint8_t dl0 = 0, dl1 = 0; // (lanes carry them across the move: the two ranges may overlap)
            if (lane < 32u) dl0 = lane < ndist ? sh.h.lens[nlen + lane] : 0;
            (void)dl1;
            __syncthreads();
            if (lane < 32u) sh.h.lens[288 + lane] = dl0;
            for (uint32_t s = nlen + lane; s < 288u; s += 64u) sh.h.lens[s] = 0;
            __syncthreads();
            nlen = 288;
            ndist = 30;
        }
        __syncthreads();
        if (!gw_build(sh.h.lens, nlen, sh.lc, sh.lsym, lane) || !gw_build(sh.h.lens + 288, ndist, sh.dc, sh.dsym, lane)) {
            st = GD_BAD_BLOCK;
            break;
        }
        __syncthreads();
        gw_fill<GW_ALPHA_LIT>(sh.lc, sh.lsym, sh.lt, GW_TB, lane, sh.len_tab, sh.dist_tab);
        gw_fill<GW_ALPHA_DIST>(sh.dc, sh.dsym, sh.dt, GW_DB, lane, sh.len_tab, sh.dist_tab);
        __syncthreads();
        {
            uint32_t pool_n = 0;
            const bool fits = gw_fill_long<GW_ALPHA_LIT>(sh.lc, sh.lsym, sh.lt, GW_TB, sh.pool, pool_n, lane, sh.len_tab, sh.dist_tab) &&
                              gw_fill_long<GW_ALPHA_DIST>(sh.dc, sh.dsym, sh.dt, GW_DB, sh.pool, pool_n, lane, sh.len_tab, sh.dist_tab);
            if (!fits) {
                st = GD_PUNT;
                break;
            }
        }
        __syncthreads();
        // ---- the block's codes: 64 chunks at once, starts moved to true boundaries until none moves
        const uint32_t rem = end_bits - pos;
        const uint32_t C = max((rem + 63u) / 64u, 256u);
        const uint32_t limit = lane == 63u ? 0x7FFFFFF0u : pos + (lane + 1u) * C;
        // (the first round only has to find where each chunk ENDS: it starts GW_SPEC bits in front of that end -- far enough for the decode to
        // stand on true boundaries when it gets there, a third of a chunk of a full member; the last lane's end is nobody's start)
        uint32_t t = lane == 63u ? end_bits : max(pos + lane * C, limit > GW_SPEC ? limit - GW_SPEC : 0u), e = 0, ntok = 0, nbytes = 0, fl = 0;
        bool need = true;
        for (uint32_t it = 0; it < 70u; ++it) {
            rounds += 1;
            if (need) e = gw_chunk<false>(sh, pay, t, limit, end_bits, nullptr, ntok, nbytes, fl);
            const uint32_t pe = (uint32_t)__shfl_up((int)e, 1);
            const uint32_t nt = lane == 0 ? pos : pe;
            need = nt != t;
            t = nt;
            if (!__builtin_amdgcn_ballot_w64(need)) break;
        }
        // every chunk now starts where the one in front ended: the chain from the block's first code is the block's, up to the first
        // end-of-block code on it (the chunks behind that one hold the next block's bits, or none)
        const uint64_t eob = __builtin_amdgcn_ballot_w64((fl & 1u) != 0);
        if (__builtin_amdgcn_ballot_w64(need) || !eob) { // (no fixed point in 70 rounds cannot be; no end-of-block code in the payload)
            st = GD_OVERRUN_IN;
            break;
        }
        const uint32_t el = (uint32_t)__builtin_ctzll(eob);
        if (__builtin_amdgcn_ballot_w64(lane <= el && (fl & 2u))) { // bits that are no code, on the block's own chain
            st = GD_BAD_CODE;
            break;
        }
        const uint32_t after = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)el);
        const uint32_t mine = lane <= el ? ntok : 0u;
        const uint32_t incl = gw_scan_incl(mine, lane);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t nbs = lane <= el ? nbytes : 0u;
#pragma unroll
        for (uint32_t d = 32; d >= 1u; d >>= 1) nbs += (uint32_t)__shfl_xor((int)nbs, d);
        total_bytes += nbs;
        uint32_t base = 0;
        if (lane == 0) base = total ? atomicAdd(cursor, total) : 0u;
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if ((uint64_t)base + total > tok_cap) {
            st = GD_PUNT;
            break;
        }
        if (mine) {
            uint32_t n2 = 0, b2 = 0, f2 = 0;
            (void)gw_chunk<true>(sh, pay, t, limit, end_bits, tok + base + (incl - mine), n2, b2, f2);
        }
        if (lane == 0) reg[b * GW_MAX_REGIONS + n_regions] = make_uint2(base, total);
        n_regions += 1;
        pos = after;
    }
    if (st == GD_OK && total_bytes != bl.out_size) st = GD_OVERRUN_OUT;
    if (lane == 0) {
        n_reg[b] = n_regions | (rounds << 8); // (the rounds the chunks' starts took to settle: a developer's figure, kept in the upper bits)
        status[b] = st;
    }
}

// The LZ77 half, a lane per member (members whose status is not GD_OK are left alone): tokens -> bytes.  One loop steps the wave: a step is
// a token, or eight bytes of a match under way (gam_kernels.hip: gd_inflate_kernel, whose output path this is).
__global__ __launch_bounds__(64) void gd_lz_kernel(const GdBlock *__restrict__ blocks, uint32_t n_blocks, const uint32_t *__restrict__ tok, const uint2 *__restrict__ reg,
                                                   const uint32_t *__restrict__ n_reg, uint8_t *out, uint32_t *__restrict__ status) {
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= n_blocks || status[b] != GD_OK) return;
    const GdBlock bl = blocks[b];
    uint8_t *o = out + bl.out_off;
    const uint32_t o_cap = bl.out_size, nr = n_reg[b] & 255u;
    uint32_t pos = 0, err = GD_OK;
    uint64_t ob_w = 0, h_lo = 0, h_hi = 0;
    uint32_t ob_fill = (uint32_t)((uintptr_t)o & 7u), ob_hole = ob_fill;
    auto ob_store = [&](uint8_t *word_at) {
        if (ob_fill == 8u && ob_hole == 0u) {
            *reinterpret_cast<uint64_t *>(word_at) = ob_w;
        } else {
            for (uint32_t k = ob_hole; k < ob_fill; ++k) word_at[k] = (uint8_t)(ob_w >> (8u * k));
        }
    };
    auto put = [&](uint64_t v, uint32_t n) { // n in 1..8 bytes (the bytes of v above them zero) behind what is there
        if (n == 8u) {
            h_lo = h_hi;
            h_hi = v;
        } else {
            const uint32_t shb = 8u * n;
            h_lo = (h_lo >> shb) | (h_hi << (64u - shb));
            h_hi = (h_hi >> shb) | (v << (64u - shb));
        }
        const uint32_t f = ob_fill;
        ob_w |= v << (8u * f);
        if (f + n >= 8u) {
            ob_fill = 8u;
            ob_store(o + pos - f);
            ob_hole = 0u;
            ob_w = f ? v >> (8u * (8u - f)) : 0ull;
            ob_fill = f + n - 8u;
        } else {
            ob_fill = f + n;
        }
        pos += n;
    };
    uint32_t r = 0, k = 0, n_k = 0;
    const uint32_t *tk = tok;
    if (nr) {
        const uint2 rg = reg[b * GW_MAX_REGIONS];
        tk = tok + rg.x;
        n_k = rg.y;
    }
    // (the tokens are asked for two steps ahead of their use: the scratch has room behind its last token)
    uint32_t t_a = n_k ? tk[0] : 0u, t_b = n_k > 1u ? tk[1] : 0u;
    uint32_t cp_len = 0, cp_dist = 0, sp_n = 0;
    uint64_t sp0 = 0, sp1 = 0;
    bool cp_short = false;
    for (;;) {
        if (cp_len) { // ---- eight bytes of a match
            if (!cp_short) { // its source lies sixteen bytes and more behind: all of it is in memory
                const uint32_t n = min(cp_len, 8u);
                uint64_t v = reinterpret_cast<const GwU64 *>(o + pos - cp_dist)->v;
                if (n < 8u) v &= (1ull << (8u * n)) - 1ull;
                put(v, n);
                cp_len -= n;
            } else { // a short period: out of the registers that hold it (a whole number of periods a step: the phase stays 0)
                const uint32_t n0 = min(cp_len, min(sp_n, 8u));
                put(n0 < 8u ? sp0 & ((1ull << (8u * n0)) - 1ull) : sp0, n0);
                cp_len -= n0;
                if (sp_n > 8u && cp_len) {
                    const uint32_t n1 = min(cp_len, sp_n - 8u); // < 8
                    put(sp1 & ((1ull << (8u * n1)) - 1ull), n1);
                    cp_len -= n1;
                }
            }
            continue;
        }
        if (k >= n_k) { // the member's next block of tokens, or its end
            r += 1;
            if (r >= nr) break;
            const uint2 rg = reg[b * GW_MAX_REGIONS + r];
            tk = tok + rg.x;
            n_k = rg.y;
            k = 0;
            t_a = n_k ? tk[0] : 0u;
            t_b = n_k > 1u ? tk[1] : 0u;
            continue;
        }
        const uint32_t t = t_a;
        t_a = t_b;
        t_b = k + 2u < n_k ? tk[k + 2u] : 0u;
        k += 1;
        const uint32_t kind = t >> 30;
        if (kind) { // ---- one to three literals
            if (pos + kind > o_cap) {
                err = GD_OVERRUN_OUT;
                break;
            }
            put((uint64_t)(t & 0xFFFFFFu), kind);
            continue;
        }
        const uint32_t len = t & 511u, dist = ((t >> 9) & 0x7FFFu) + 1u;
        if (dist > pos) { // (BGZF members carry no preset dictionary: nothing lies before the member's own output)
            err = GD_BAD_CODE;
            break;
        }
        if (pos + len > o_cap) {
            err = GD_OVERRUN_OUT;
            break;
        }
        cp_len = len;
        cp_dist = dist;
        cp_short = dist < 16u;
        if (cp_short) { // the period's bytes: the last `dist` of the sixteen kept in registers
            if (dist >= 8u) {
                const uint32_t shb = 8u * (16u - dist); // 8..64
                sp0 = shb == 64u ? h_hi : (h_lo >> shb) | (h_hi << (64u - shb));
                sp1 = shb == 64u ? 0ull : h_hi >> shb; // (its first dist - 8 bytes are used)
                sp_n = dist;
            } else {
                const uint64_t v0 = h_hi >> (8u * (8u - dist));
                uint64_t ext = v0 & ((1ull << (8u * dist)) - 1ull);
                ext |= ext << (8u * dist);                 // 2 periods (dist < 8: the shifts stay below 64)
                if (dist < 4u) ext |= ext << (16u * dist); // 4
                if (dist < 2u) ext |= ext << 32;           // 8
                sp0 = ext;
                sp1 = 0;
                sp_n = (8u / dist) * dist;
            }
        }
    }
    ob_store(o + pos - ob_fill); // what waits goes out as bytes
    if (err == GD_OK && pos != o_cap) err = GD_OVERRUN_OUT; // (ISIZE says how long the member's output is)
    if (err != GD_OK) status[b] = err;
}

// The LZ77 half by a WAVE per member: no lane waits for another's bytes.  A window of the member's output (up to GW_W bytes, a few
// hundred tokens) is laid out in LDS as 16-bit REFERENCES, one per byte: a literal says its byte (0x8000 | byte), a byte of a match says
// where in the window its source lies -- or, for a source before the window, its byte, fetched from the output already written.  Then every
// reference is replaced by the one it points at, again and again (pointer jumping): chains of copies of copies -- GAM data is little
// else: a mapping is the mapping before it with two bytes changed -- halve with every round, ~7 rounds for a window.  The bytes then leave
// as coalesced dwords.  (A lane per member walks the same chains one L2 round trip at a time: 29 ms for any number of members up to the
// chip's 64 x waves, which is what a piece of the file waited for.)
constexpr uint32_t GW_W = 4096, GW_TPL = 10; // window bytes; tokens a lane takes per window
__global__ __launch_bounds__(64) void gd_lzw_kernel(const GdBlock *__restrict__ blocks, uint32_t n_blocks, const uint32_t *__restrict__ tok, const uint2 *__restrict__ reg,
                                                    const uint32_t *__restrict__ n_reg, uint8_t *out, uint32_t *__restrict__ status) {
    __shared__ __attribute__((aligned(8))) uint16_t ref[GW_W + 8];
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (b >= n_blocks || status[b] != GD_OK) return;
    const GdBlock bl = blocks[b];
    uint8_t *o = out + bl.out_off;
    const uint32_t o_cap = bl.out_size, nr = n_reg[b] & 255u;
    uint32_t pos = 0, err = GD_OK;
    for (uint32_t r = 0; r < nr && err == GD_OK; ++r) {
        const uint2 rg = reg[b * GW_MAX_REGIONS + r];
        const uint32_t *tk = tok + rg.x;
        const uint32_t n = rg.y;
        uint32_t k = 0;
        while (k < n && err == GD_OK) {
            // ---- the window's tokens: GW_TPL consecutive ones per lane, as many of them as end within GW_W bytes
            uint32_t t[GW_TPL], sum = 0;
            const uint32_t i0 = k + lane * GW_TPL;
#pragma unroll
            for (uint32_t j = 0; j < GW_TPL; ++j) {
                t[j] = i0 + j < n ? tk[i0 + j] : 0u;
                sum += i0 + j < n ? ((t[j] >> 30) ? (t[j] >> 30) : (t[j] & 511u)) : 0u;
            }
            uint32_t x = gw_scan_incl(sum, lane) - sum, taken = 0, w_end = 0;
            bool bad = false;
            // the bytes a match takes from BEFORE the window (written out already): asked for now, all of the lane's tokens' at once -- eight
            // bytes a match, which is most matches whole -- and put into the references below (a load per byte inside that loop was a trip
            // to L2 per byte, one after the other: most of this kernel's time)
            uint64_t pre[GW_TPL];
            {
                uint32_t xx = x;
#pragma unroll
                for (uint32_t j = 0; j < GW_TPL; ++j) {
                    const uint32_t kind = t[j] >> 30, len = i0 + j < n ? (kind ? kind : (t[j] & 511u)) : 0u;
                    const uint32_t dist = ((t[j] >> 9) & 0x7FFFu) + 1u;
                    pre[j] = 0;
                    if (!kind && len && xx + len <= GW_W && dist > xx && dist <= pos + xx) pre[j] = reinterpret_cast<const GwU64 *>(o + (pos + xx - dist))->v;
                    xx += len;
                }
            }
            __syncthreads(); // (the window before is written out)
#pragma unroll
            for (uint32_t j = 0; j < GW_TPL; ++j) {
                if (i0 + j >= n) break;
                const uint32_t kind = t[j] >> 30, len = kind ? kind : (t[j] & 511u);
                if (x + len > GW_W) break; // (this token and every one behind it: the next window's)
#if defined(LZW_EXP) && (LZW_EXP & 1) // (developer pricing run: what the references' filling costs -- wrong output)
                if (true) {
                    if (len) ref[x] = (uint16_t)(0x8000u | (t[j] & 255u));
                } else
#endif
                if (kind) {
                    for (uint32_t q = 0; q < kind; ++q) ref[x + q] = (uint16_t)(0x8000u | ((t[j] >> (8u * q)) & 255u));
                } else {
                    const uint32_t dist = ((t[j] >> 9) & 0x7FFFu) + 1u;
                    if (dist > pos + x) { // (BGZF members carry no preset dictionary: nothing lies before the member's own output)
                        bad = true;
                        break;
                    }
                    const uint32_t n_ext = dist > x ? min(len, dist - x) : 0u; // bytes whose source lies before the window
                    for (uint32_t q = 0; q < min(n_ext, 8u); ++q) ref[x + q] = (uint16_t)(0x8000u | ((uint32_t)(pre[j] >> (8u * q)) & 255u));
                    for (uint32_t q = 8u; q < n_ext; ++q) ref[x + q] = (uint16_t)(0x8000u | o[pos + x + q - dist]); // (a long match across the window's start)
                    for (uint32_t q = n_ext; q < len; ++q) ref[x + q] = (uint16_t)(x + q - dist);
                }
                x += len;
                taken += 1;
                w_end = x;
            }
            if (__builtin_amdgcn_ballot_w64(bad)) {
                err = GD_BAD_CODE;
                break;
            }
            uint32_t n_taken = taken;
#pragma unroll
            for (uint32_t d = 32; d >= 1u; d >>= 1) {
                n_taken += (uint32_t)__shfl_xor((int)n_taken, d);
                w_end = max(w_end, (uint32_t)__shfl_xor((int)w_end, d));
            }
            if (pos + w_end > o_cap) {
                err = GD_OVERRUN_OUT;
                break;
            }
            __syncthreads();
            // ---- pointer jumping, 256 references at a time from the window's front (a lane owns four consecutive ones of them): a reference
            // points backwards, so what lies in front of the 256 at hand is bytes already -- ONE look-up settles a reference into it -- and only
            // the references into the 256 themselves (copies of copies a few bytes back) take more rounds, over 256 entries, not 4 096.  (Every
            // round over the whole window, until no reference anywhere was open: ~7 rounds x 16 quads a lane, three quarters of this
            // kernel's instructions.)
            const uint32_t n_quads = (w_end + 3u) / 4u;
#if defined(LZW_EXP) && (LZW_EXP & 2) // (developer pricing run: without the pointer jumping -- wrong output)
            for (uint32_t qb = 0; qb < 0u; qb += 64u) {
#else
            for (uint32_t qb = 0; qb < n_quads; qb += 64u) {
#endif
                const uint32_t qd = qb + lane;
                for (uint32_t round = 0; round < 16u; ++round) {
                    bool open = false;
                    if (qd < n_quads) {
                        const uint64_t v = *reinterpret_cast<const uint64_t *>(&ref[qd * 4u]);
                        if ((v & 0x8000800080008000ull) != 0x8000800080008000ull) {
                            uint64_t nv = 0;
#pragma unroll
                            for (uint32_t c = 0; c < 4u; ++c) {
                                uint32_t e = (uint32_t)(v >> (16u * c)) & 0xFFFFu;
                                if (!(e & 0x8000u) && qd * 4u + c < w_end) {
                                    e = ref[e];
                                    open |= !(e & 0x8000u);
                                }
                                nv |= (uint64_t)e << (16u * c);
                            }
                            *reinterpret_cast<uint64_t *>(&ref[qd * 4u]) = nv;
                        }
                    }
                    __syncthreads();
                    if (!__builtin_amdgcn_ballot_w64(open)) break;
                }
            }
            // ---- the window's bytes: four to a lane and store
            for (uint32_t qd = lane; qd < n_quads; qd += 64u) {
                const uint64_t v = *reinterpret_cast<const uint64_t *>(&ref[qd * 4u]);
                const uint32_t w4 = ((uint32_t)v & 255u) | (((uint32_t)(v >> 16) & 255u) << 8) | (((uint32_t)(v >> 32) & 255u) << 16) | (((uint32_t)(v >> 48) & 255u) << 24);
                uint8_t *dst = o + pos + qd * 4u;
                if (qd * 4u + 4u <= w_end) {
                    struct __attribute__((packed)) U32 {
                        uint32_t v;
                    };
                    reinterpret_cast<U32 *>(dst)->v = w4;
                } else {
                    for (uint32_t c = 0; qd * 4u + c < w_end; ++c) dst[c] = (uint8_t)(w4 >> (8u * c));
                }
            }
            pos += w_end;
            k += n_taken;
            if (n_taken == 0) { // (cannot be: a token is at most 258 bytes)
                err = GD_BAD_BLOCK;
                break;
            }
        }
    }
    if (err == GD_OK && pos != o_cap) err = GD_OVERRUN_OUT; // (ISIZE says how long the member's output is)
    if (err != GD_OK && lane == 0) status[b] = err;
}

// The members' CRC-32 (RFC 1952 8: the gzip trailer's first word), a wave per member: lane j takes the j-th kilobyte counted from the
// member's END (the first chunk is the short one); the chunks' registers are then folded front to back -- processing B from register r
// equals (r moved across |B| zero bytes) xor (B processed from 0), and moving across GW_CRC_CHUNK zero bytes is four look-ups in a table
// made for that length.  A lane takes its kilobyte sixteen bytes to a load and four bytes to a step ("slicing by four": the register
// xor the next word, then one look-up per byte in the tables of a byte followed by 3, 2, 1, 0 zero bytes -- four look-ups that do not wait
// for each other).  (First form: a byte to a load and to a step -- lanes a kilobyte apart, so every byte load of a wave touched 64 lines
// that the other waves of the CU had pushed out of L1 since: 3.2 ms per 1 M reads' members, more than half of what inflating them takes.)
// A member whose bytes do not give the trailer's value gets GD_BAD_CRC: the host pipeline's zlib / libdeflate refuse such a file, and so
// does this one (the member goes through the older kernel first, as every status but GD_OK does).
// tabs (GW_CRC_TABS words): [256] the CRC table, [4][256] the zero-bytes operator, [3][256] the table moved across 1, 2, 3 zero bytes.
constexpr uint32_t GW_CRC_CHUNK = 1024;
__global__ __launch_bounds__(64) void gd_crc_kernel(const uint8_t *__restrict__ out, const GdBlock *__restrict__ blocks, uint32_t n_blocks, const uint32_t *__restrict__ want,
                                                    const uint32_t *__restrict__ tabs, uint32_t *__restrict__ status) {
    __shared__ uint32_t t[GAMDEV_CRC_TABS];
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (b >= n_blocks || status[b] != GD_OK) return;
    for (uint32_t i = lane; i < GAMDEV_CRC_TABS; i += 64u) t[i] = tabs[i];
    __syncthreads();
    const GdBlock bl = blocks[b];
    const uint8_t *o = out + bl.out_off;
    const uint32_t n = bl.out_size, n_chunks = (n + GW_CRC_CHUNK - 1u) / GW_CRC_CHUNK; // <= 64: a BGZF member holds 64 KB at most
    if (n_chunks > 64u) return; // (not BGZF's: left unchecked)
    const uint32_t first_len = n - (n_chunks ? (n_chunks - 1u) * GW_CRC_CHUNK : 0u);
    const uint32_t *t1 = t + 1280u, *t2 = t + 1536u, *t3 = t + 1792u;
    uint32_t r = 0;
    if (lane < n_chunks) {
        const uint32_t at = lane == 0 ? 0u : first_len + (lane - 1u) * GW_CRC_CHUNK, len = lane == 0 ? first_len : GW_CRC_CHUNK;
        r = lane == 0 ? 0xFFFFFFFFu : 0u;
        struct __attribute__((packed)) U128 { // (a member's output starts at any byte)
            uint4 v;
        };
        const uint8_t *p = o + at;
        uint32_t k = 0;
        if (len >= 16u) {
            uint4 nx = reinterpret_cast<const U128 *>(p)->v;
            for (; k + 16u <= len; k += 16u) {
                const uint4 w = nx;
                if (k + 32u <= len) nx = reinterpret_cast<const U128 *>(p + k + 16u)->v; // (asked for a step ahead)
                r ^= w.x;
                r = t3[r & 255u] ^ t2[(r >> 8) & 255u] ^ t1[(r >> 16) & 255u] ^ t[r >> 24];
                r ^= w.y;
                r = t3[r & 255u] ^ t2[(r >> 8) & 255u] ^ t1[(r >> 16) & 255u] ^ t[r >> 24];
                r ^= w.z;
                r = t3[r & 255u] ^ t2[(r >> 8) & 255u] ^ t1[(r >> 16) & 255u] ^ t[r >> 24];
                r ^= w.w;
                r = t3[r & 255u] ^ t2[(r >> 8) & 255u] ^ t1[(r >> 16) & 255u] ^ t[r >> 24];
            }
        }
        for (; k < len; ++k) r = t[(r ^ p[k]) & 255u] ^ (r >> 8);
    }
    uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)r, 0);
    for (uint32_t j = 1; j < n_chunks; ++j) {
        const uint32_t rj = (uint32_t)__shfl((int)r, (int)j);
        total = t[256u + (total & 255u)] ^ t[512u + ((total >> 8) & 255u)] ^ t[768u + ((total >> 16) & 255u)] ^ t[1024u + (total >> 24)] ^ rj;
    }
    if (lane == 0 && (n ? (total ^ 0xFFFFFFFFu) : 0u) != want[b]) status[b] = GD_BAD_CRC;
}

} // namespace gd

// Inflates n_blocks members: the two kernels above on the stream.  The scratch belongs to the caller (tok: tok_cap tokens shared by all
// launches that use `cursor`, which the caller zeroes once per piece; reg / n_reg: per member).  Members left with a status other than
// GD_OK are the caller's to send through gamdev_inflate (the older kernel): gamdev_inflate_redo.
int gamdev_inflate_wave(const uint8_t *d_in, const GdBlock *d_blocks, uint32_t n_blocks, uint8_t *d_out, uint32_t *d_status, uint32_t *d_tok, uint32_t tok_cap,
                        uint32_t *d_cursor, void *d_reg, uint32_t *d_n_reg, hipStream_t st) {
    if (n_blocks == 0) return VGAN_OK;
    hipLaunchKernelGGL(gd::gd_tokens_kernel, dim3(n_blocks), dim3(64), 0, st, d_in, d_blocks, n_blocks, d_tok, tok_cap, d_cursor, (uint2 *)d_reg, d_n_reg, d_status);
    static const bool lane_lz = getenv("VGAN_GAMDEV_LZ") && !strcmp(getenv("VGAN_GAMDEV_LZ"), "lane"); // (developer aid: the LZ77 half a lane per member)
    if (lane_lz) hipLaunchKernelGGL(gd::gd_lz_kernel, dim3((n_blocks + 63) / 64), dim3(64), 0, st, d_blocks, n_blocks, d_tok, (const uint2 *)d_reg, d_n_reg, d_out, d_status);
    else hipLaunchKernelGGL(gd::gd_lzw_kernel, dim3(n_blocks), dim3(64), 0, st, d_blocks, n_blocks, d_tok, (const uint2 *)d_reg, d_n_reg, d_out, d_status);
    HIPCHK(hipGetLastError());
    return VGAN_OK;
}

// [256] the CRC-32 table (reflected polynomial 0xEDB88320), then [4][256]: byte k of a register moved across GW_CRC_CHUNK zero bytes, then
// [3][256]: the table's entries moved across 1, 2, 3 zero bytes (slicing by four)
const uint32_t *gamdev_crc_tables() {
    static uint32_t tabs[GAMDEV_CRC_TABS];
    static const bool made = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            tabs[i] = c;
        }
        for (uint32_t k = 0; k < 4; ++k)
            for (uint32_t v = 0; v < 256; ++v) {
                uint32_t r = v << (8 * k);
                for (uint32_t z = 0; z < gd::GW_CRC_CHUNK; ++z) r = tabs[r & 255u] ^ (r >> 8);
                tabs[256 + 256 * k + v] = r;
            }
        for (uint32_t k = 1; k < 4; ++k)
            for (uint32_t v = 0; v < 256; ++v) {
                const uint32_t prev = k == 1 ? tabs[v] : tabs[1280 + 256 * (k - 2) + v];
                tabs[1280 + 256 * (k - 1) + v] = tabs[prev & 255u] ^ (prev >> 8);
            }
        return true;
    }();
    (void)made;
    return tabs;
}
int gamdev_crc(const uint8_t *d_out, const GdBlock *d_blocks, uint32_t n_blocks, const uint32_t *d_want, const uint32_t *d_tabs, uint32_t *d_status, hipStream_t st) {
    if (n_blocks == 0) return VGAN_OK;
    hipLaunchKernelGGL(gd::gd_crc_kernel, dim3(n_blocks), dim3(64), 0, st, d_out, d_blocks, n_blocks, d_want, d_tabs, d_status);
    HIPCHK(hipGetLastError());
    return VGAN_OK;
}

} // namespace vgan
#include "module_anchor.h"
const void *vgan::anchor_gam_inflate_wave() { return (const void *)&vgan::gd::gd_crc_kernel; }
