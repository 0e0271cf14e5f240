// DEFLATE of BGZF members, a LANE per member (reference: src/readGAM.h:20-68 through libvgio's BGZF stream): round 5's
// inflate kernel, and since round 6 the fallback of gam_inflate_wave.hip -- whatever the two kernels there do not finish (a stored block,
// more blocks than a member has regions for, no room in their scratch) or call wrong is done again here, so an error is only ever
// reported by this kernel.
//
//   gd_inflate_kernel   blocks are independent members of at most 64 KB.  A lane decodes serially as a CPU would, a wave steps its 64
//                       lanes through one loop (a step = a block header, a symbol, or eight bytes of a match).  A code's length comes
//                       from fifteen compares against limits kept in registers (no loop over its bits), its symbol from the lane's
//                       column of the wave's tables in LDS; output gathers in a register and leaves in aligned 8-byte words, a match
//                       is copied eight bytes to a load.  A lane is slow -- a 64 KB block takes ~50 ms whatever the number of
//                       blocks up to the chip's 98 k -- which is why it is the fallback.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// ------------------------------------------------------------------------------------------------------------------ inflate
// (the GD_* status codes: gam_object.h)

// LSB-first bit reader over 4-byte words, one word ahead: the word that refills `buf` was requested a refill earlier.  (The payload is
// followed by the member's 8-byte trailer: a word read past the payload's end stays inside the file; past that nothing is read and zero
// bits are fed -- bits_overran says whether any were consumed.)  A ring of the lane's input in LDS, topped up by all lanes together every
// other step, was tried against the wait for these loads and measured SLOWER (110 against 76 ms for a wave alone): a wave alone spends its
// time issuing instructions, ~350 a step with every path of the loop taken by some lane, not waiting.
struct Bits {
    const uint8_t *p; // the next word to request
    const uint8_t *e; // the payload's end
    uint64_t buf;
    uint32_t cnt, nxt; // valid bits in buf; the word that goes in next
};
__device__ __forceinline__ uint32_t bits_word(Bits &b) {
    uint32_t w = 0;
    if (b.p + 4 <= b.e + 8) {
        w = *reinterpret_cast<const uint32_t *>(b.p);
        const int64_t left = b.e - b.p;
        if (left < 4) w = left <= 0 ? 0u : w & ((1u << (8 * left)) - 1u);
    }
    b.p += 4;
    return w;
}
__device__ __forceinline__ void bits_init(Bits &b, const uint8_t *p, const uint8_t *e) {
    b.p = p;
    b.e = e;
    b.buf = 0;
    b.cnt = 0;
    while (((uintptr_t)b.p & 3u) && b.p < b.e) { // up to the first aligned word
        b.buf |= (uint64_t)*b.p++ << b.cnt;
        b.cnt += 8;
    }
    if ((uintptr_t)b.p & 3u) { // a payload that ended before it: the bytes up to the word are fed as zero bits (bits_overran counts from p)
        const uint32_t skip = 4u - (uint32_t)((uintptr_t)b.p & 3u);
        b.p += skip;
        b.cnt += 8u * skip;
    }
    b.nxt = bits_word(b);
}
__device__ __forceinline__ void bits_fill(Bits &b) { // at least 32 bits
    if (b.cnt < 32) {
        b.buf |= (uint64_t)b.nxt << b.cnt;
        b.cnt += 32;
        b.nxt = bits_word(b);
    }
}
__device__ __forceinline__ uint32_t bits_get(Bits &b, uint32_t n) { // n <= 16
    bits_fill(b);
    const uint32_t v = (uint32_t)b.buf & ((1u << n) - 1u);
    b.buf >>= n;
    b.cnt -= n;
    return v;
}
__device__ __forceinline__ void bits_drop(Bits &b, uint32_t n) { // behind a bits_fill
    b.buf >>= n;
    b.cnt -= n;
}
// bits consumed beyond the payload's end?  (fed so far: everything below p; not consumed: cnt and the 32 of nxt)
__device__ __forceinline__ bool bits_overran(const Bits &b) { return (int64_t)(b.e - b.p) * 8 + (int64_t)b.cnt + 32 < 0; }

// counts per code length 0..15, packed four to a 64-bit word
struct Counts {
    uint64_t w[4];
};
__device__ __forceinline__ uint32_t cnt_get(const Counts &c, uint32_t len) {
    const uint64_t w = len < 8 ? (len < 4 ? c.w[0] : c.w[1]) : (len < 12 ? c.w[2] : c.w[3]);
    return (uint32_t)(w >> ((len & 3u) * 16u)) & 0xFFFFu;
}
__device__ __forceinline__ void cnt_add(Counts &c, uint32_t len, uint32_t v) {
    const uint64_t add = (uint64_t)v << ((len & 3u) * 16u);
    if (len < 4) c.w[0] += add;
    else if (len < 8) c.w[1] += add;
    else if (len < 12) c.w[2] += add;
    else c.w[3] += add;
}

// Canonical Huffman decoding without a loop over the code's bits.  With the next 15 stream bits as a number `peek` whose most significant
// bit is the first bit read (codes are packed that way round, RFC 1951 3.1.1), the codes of length L are the numbers in
// [limit[L-1], limit[L]) where limit[L] = (first code of length L + their count) << (15 - L): the length of the code in front is one more
// than the number of limits not above peek -- fifteen compares against registers, the same for every lane -- and its symbol is
// sym[base[L] + (peek >> (15 - L))] with base[L] = (symbols of shorter codes) - (first code of length L).  (The first version
// walked the code bit by bit, as puff.c does: a literal of nine bits was nine rounds of the loop for the whole wave, 80 ms per block.)
struct Dec {
    uint32_t lim[8]; // limit[1..15], two to a word: limit[2i + 1] in the low half of lim[i], limit[2i + 2] in the high one (lim[7]: 0xFFFF)
};
__device__ __forceinline__ uint32_t dec_len(const Dec &d, uint32_t peek) { // 1..15, or 16: no code
    uint32_t n = 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        n += peek >= (d.lim[i] & 0xFFFFu) ? 1u : 0u;
        n += peek >= (d.lim[i] >> 16) ? 1u : 0u;
    }
    return n;
}
// {limits, bases} of the code whose counts are c (base[L] into base_t[(L - 1) * 64], the lane's column of the wave's table); returns the
// number of coded symbols
__device__ __forceinline__ uint32_t dec_build(const Counts &c, Dec &d, uint16_t *base_t) {
    uint32_t code = 0, idx = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d.lim[i] = 0xFFFF0000u;
#pragma unroll
    for (uint32_t len = 1; len <= 15; ++len) {
        const uint32_t cnt = cnt_get(c, len);
        base_t[(len - 1u) * 64u] = (uint16_t)(idx - code);
        code += cnt;
        idx += cnt;
        const uint32_t lim = code << (15u - len); // <= 2^15 for a set that is not over-subscribed
        if ((len - 1u) & 1u) d.lim[(len - 1u) >> 1] = (d.lim[(len - 1u) >> 1] & 0xFFFFu) | (lim << 16);
        else d.lim[(len - 1u) >> 1] = (d.lim[(len - 1u) >> 1] & 0xFFFF0000u) | lim;
        code <<= 1;
    }
    return idx;
}
__device__ __forceinline__ uint32_t bits_peek15(const Bits &b) { return __builtin_bitreverse32((uint32_t)b.buf) >> 17; }

// counts per length of `n` code lengths; false for an over-subscribed set (an incomplete one is allowed where RFC 1951 allows it: a
// single distance code); offs[L] = symbols of shorter codes
__device__ __forceinline__ bool huff_counts(const uint8_t *lengths, int n, Counts &c, uint16_t (&offs)[16]) {
    c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
    for (int s = 0; s < n; ++s) cnt_add(c, lengths[s], 1u);
    int left = 1;
    for (uint32_t len = 1; len <= 15; ++len) {
        left <<= 1;
        left -= (int)cnt_get(c, len);
        if (left < 0) return false;
    }
    offs[0] = 0;
    offs[1] = 0;
    for (uint32_t len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + cnt_get(c, len));
    c.w[0] &= ~0xFFFFull; // (codes of length 0 are not codes: their count must not take part in decoding)
    return true;
}

__device__ const uint16_t gd_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t gd_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t gd_dist_base[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                              193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t gd_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t gd_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// eight bytes at any address (the hardware takes unaligned global accesses; the compiler is told so by the packed type)
struct __attribute__((packed)) GdU64 {
    uint64_t v;
};
__device__ __forceinline__ uint64_t gd_load8(const uint8_t *p) { return reinterpret_cast<const GdU64 *>(p)->v; }

// The lane's output goes out in aligned 8-byte words: bytes gather in a register, a full word is ONE store.  (A byte store per literal
// and eight byte loads + eight byte stores per piece of a match were ten vector-memory instructions per symbol, each to 64 different
// cache lines -- 64 cycles of a CU's address unit each: with every lane of the chip busy that, not the decoding, set the 250 ms a
// 10 M-read file took.)  `hole`: bytes of the first word that belong to the member in front (another lane's).
struct OutBuf {
    uint64_t w;
    uint32_t fill, hole; // bytes of w that are decided (the hole included); of those, the first `hole` are not ours
};
__device__ __forceinline__ void ob_store(OutBuf &ob, uint8_t *word_at) { // a full word, or what there is of it (hole .. fill)
    if (ob.fill == 8u && ob.hole == 0u) {
        *reinterpret_cast<uint64_t *>(word_at) = ob.w;
    } else {
        for (uint32_t k = ob.hole; k < ob.fill; ++k) word_at[k] = (uint8_t)(ob.w >> (8u * k));
    }
}

// One lane per BGZF member, the wave stepped by ONE loop: a step is a block header (with the code tables: the cold part, a few per
// member), one symbol, or eight bytes of a match -- a lane in the middle of a long match or at a block header holds the other 63 up for a
// step of its own kind and no longer.  The symbol tables (canonical order; low byte + a bit for the symbols from 256 on) and the bases are
// in LDS, one column per lane (26 KB per wave: six waves to a CU); the fifteen limits of either code in registers.
constexpr uint32_t GD_LIT_ROWS = 288, GD_DIST_ROWS = 30;
__global__ __launch_bounds__(64) void gd_inflate_kernel(const uint8_t *__restrict__ in, const GdBlock *__restrict__ blocks, uint32_t n_blocks,
                                                        uint8_t *out, uint32_t *__restrict__ status) {
    __shared__ uint8_t lit_s[GD_LIT_ROWS][64];
    __shared__ uint32_t hi_s[(GD_LIT_ROWS + 31) / 32][64];
    __shared__ uint8_t dsym_s[GD_DIST_ROWS][64];
    __shared__ uint16_t base_s[2][15][64];
    __shared__ uint32_t len_s[29], dist_s[30]; // base | extra bits << 16
    const uint32_t lane = threadIdx.x, b = blockIdx.x * 64u + lane;
    if (lane < 29) len_s[lane] = gd_len_base[lane] | ((uint32_t)gd_len_extra[lane] << 16);
    if (lane < 30) dist_s[lane] = gd_dist_base[lane] | ((uint32_t)gd_dist_extra[lane] << 16);
    __syncthreads();
    if (b >= n_blocks) return;
    const GdBlock bl = blocks[b];
    uint8_t *lit = &lit_s[0][lane], *dsy = &dsym_s[0][lane];
    uint32_t *hi = &hi_s[0][lane];
    uint16_t *lbase = &base_s[0][0][lane], *dbase = &base_s[1][0][lane];
    uint8_t *o = out + bl.out_off;
    const uint32_t o_cap = bl.out_size;
    uint32_t pos = 0, err = GD_OK; // pos: bytes decoded (those waiting in ob included)
    OutBuf ob;
    ob.hole = ob.fill = (uint32_t)((uintptr_t)o & 7u);
    ob.w = 0;
    Bits br;
    bits_init(br, in + bl.in_off, in + bl.in_off + bl.in_size);
    Dec ld, dd;
#pragma unroll
    for (int i = 0; i < 8; ++i) ld.lim[i] = dd.lim[i] = 0u;
    uint32_t n_lit = 0, n_dist = 0;
    bool in_block = false, last = false;
    uint32_t cp_len = 0, cp_dist = 0, m_len = 0, sp_n = 0; // a match under way: bytes to go, distance; the length waiting for its distance; bytes a step of a short period
    uint64_t sp0 = 0, sp1 = 0;                              // a short period's bytes
    bool cp_short = false, want_dist = false;
    // the word the next byte goes into starts at o + pos - ob.fill (pos counts the bytes that wait in ob; fill counts the hole too)
    // the last 16 bytes of output, newest last (byte 7 of h_hi): a match whose source begins less than 16 bytes back takes its period from
    // here -- some of those bytes are still waiting in ob, and sending them out as single bytes first (and the rest of their word as
    // single bytes later) was most of the kernel's store instructions
    uint64_t h_lo = 0, h_hi = 0;
    auto put = [&](uint64_t v, uint32_t n) { // n in 1..8 bytes (the bytes of v above them zero) behind what is there
        if (n == 8u) {
            h_lo = h_hi;
            h_hi = v;
        } else {
            const uint32_t sh = 8u * n; // 8..56
            h_lo = (h_lo >> sh) | (h_hi << (64u - sh));
            h_hi = (h_hi >> sh) | (v << (64u - sh));
        }
        const uint32_t f = ob.fill;
        ob.w |= v << (8u * f);
        if (f + n >= 8u) {
            ob.fill = 8u;
            ob_store(ob, o + pos - f);
            ob.hole = 0u;
            ob.w = f ? v >> (8u * (8u - f)) : 0ull;
            ob.fill = f + n - 8u;
        } else {
            ob.fill = f + n;
        }
        pos += n;
    };
    auto flush = [&]() { // what waits goes out as bytes; the word goes on from there
        ob_store(ob, o + pos - ob.fill);
        ob.hole = ob.fill; // (those bytes are in memory: not to be written again)
    };
    for (;;) {
        if (!in_block) { // ---- a block header (cold)
            if (last || err != GD_OK) break;
            last = bits_get(br, 1) != 0;
            const uint32_t type = bits_get(br, 2);
            if (type == 0) { // stored
                const uint32_t drop = br.cnt & 7u;
                bits_drop(br, drop);
                const uint32_t len = bits_get(br, 16), nlen = bits_get(br, 16);
                if ((len ^ 0xFFFFu) != nlen) {
                    err = GD_BAD_STORED;
                    continue;
                }
                if (pos + len > o_cap) {
                    err = GD_OVERRUN_OUT;
                    continue;
                }
                for (uint32_t i = 0; i < len; ++i) put(bits_get(br, 8), 1u);
                if (bits_overran(br)) err = GD_OVERRUN_IN;
                continue;
            }
            if (type == 3) {
                err = GD_BAD_BLOCK;
                continue;
            }
            uint8_t lengths[GD_LIT_ROWS + GD_DIST_ROWS + 4]; // (scratch)
            uint32_t nlen = 288, ndist = 30;
            if (type == 1) { // fixed codes
                for (int s = 0; s < 144; ++s) lengths[s] = 8;
                for (int s = 144; s < 256; ++s) lengths[s] = 9;
                for (int s = 256; s < 280; ++s) lengths[s] = 7;
                for (int s = 280; s < 288; ++s) lengths[s] = 8; // (286 and 287 never occur but take part in the code's construction)
                for (int s = 0; s < 30; ++s) lengths[nlen + s] = 5;
            } else { // dynamic codes
                nlen = bits_get(br, 5) + 257;
                ndist = bits_get(br, 5) + 1;
                const uint32_t ncode = bits_get(br, 4) + 4;
                if (nlen > 286 || ndist > 30) {
                    err = GD_BAD_BLOCK;
                    continue;
                }
                uint8_t cl[19];
                for (int s = 0; s < 19; ++s) cl[s] = 0;
                for (uint32_t i = 0; i < ncode; ++i) cl[gd_clen_order[i]] = (uint8_t)bits_get(br, 3);
                Counts cc;
                uint16_t offs[16];
                if (!huff_counts(cl, 19, cc, offs)) {
                    err = GD_BAD_BLOCK;
                    continue;
                }
                for (int s = 0; s < 19; ++s) // (the code-length code's symbols borrow the distance table's column)
                    if (cl[s] != 0) dsy[(uint32_t)(offs[cl[s]]++) * 64u] = (uint8_t)s;
                Dec cd;
                (void)dec_build(cc, cd, dbase);
                uint32_t idx = 0;
                while (idx < nlen + ndist && err == GD_OK) {
                    bits_fill(br);
                    const uint32_t pk = bits_peek15(br), cl_len = dec_len(cd, pk);
                    const uint32_t ci = cl_len > 15u ? 0xFFFFu : ((uint32_t)dbase[(cl_len - 1u) * 64u] + (pk >> (15u - cl_len))) & 0xFFFFu;
                    if (ci >= 19u) {
                        err = GD_BAD_CODE;
                        break;
                    }
                    const uint32_t s = dsy[ci * 64u];
                    bits_drop(br, cl_len);
                    if (s < 16u) {
                        lengths[idx++] = (uint8_t)s;
                    } else {
                        uint32_t rep, val = 0;
                        if (s == 16u) {
                            if (idx == 0) {
                                err = GD_BAD_BLOCK;
                                break;
                            }
                            val = lengths[idx - 1];
                            rep = 3 + bits_get(br, 2);
                        } else if (s == 17u) {
                            rep = 3 + bits_get(br, 3);
                        } else {
                            rep = 11 + bits_get(br, 7);
                        }
                        if (idx + rep > nlen + ndist) {
                            err = GD_BAD_BLOCK;
                            break;
                        }
                        while (rep--) lengths[idx++] = (uint8_t)val;
                    }
                }
                if (err != GD_OK) continue;
                if (lengths[256] == 0) {
                    err = GD_BAD_BLOCK;
                    continue;
                }
            }
            Counts lc, dc;
            uint16_t offs[16];
            if (!huff_counts(lengths, (int)nlen, lc, offs)) {
                err = GD_BAD_BLOCK;
                continue;
            }
            for (uint32_t k = 0; k < (GD_LIT_ROWS + 31) / 32; ++k) hi[k * 64u] = 0u;
            for (uint32_t s = 0; s < nlen; ++s)
                if (lengths[s] != 0) {
                    const uint32_t at = offs[lengths[s]]++;
                    lit[at * 64u] = (uint8_t)s;
                    if (s >= 256u) hi[(at >> 5) * 64u] |= 1u << (at & 31u);
                }
            n_lit = dec_build(lc, ld, lbase);
            if (!huff_counts(lengths + nlen, (int)ndist, dc, offs)) {
                err = GD_BAD_BLOCK;
                continue;
            }
            for (uint32_t s = 0; s < ndist; ++s)
                if (lengths[nlen + s] != 0) dsy[(uint32_t)(offs[lengths[nlen + s]]++) * 64u] = (uint8_t)s;
            n_dist = dec_build(dc, dd, dbase);
            in_block = true;
            continue;
        }
        if (cp_len) { // ---- eight bytes of a match
            if (!cp_short) { // its source lies sixteen bytes and more behind: all of it is in memory
                const uint32_t n = min(cp_len, 8u);
                uint64_t v = gd_load8(o + pos - cp_dist);
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
        // ---- a code: of the literal / length alphabet, or -- the step after a length -- of the distance alphabet, through the same
        // instructions (a lane that stood at a distance code used to run them a second time with the other 63 waiting)
        bits_fill(br);
        const uint32_t peek = bits_peek15(br);
        Dec cur;
#pragma unroll
        for (int i = 0; i < 8; ++i) cur.lim[i] = want_dist ? dd.lim[i] : ld.lim[i];
        const uint32_t cl = dec_len(cur, peek);
        const uint32_t ci = cl > 15u ? 0xFFFFu : ((uint32_t)(want_dist ? dbase : lbase)[(cl - 1u) * 64u] + (peek >> (15u - cl))) & 0xFFFFu;
        if (ci >= (want_dist ? n_dist : n_lit)) { // no code (or one of a set that holds fewer)
            err = GD_BAD_CODE;
            in_block = false;
            continue;
        }
        const uint32_t sym = (want_dist ? dsy : lit)[ci * 64u];
        const uint32_t is_hi = want_dist ? 0u : (hi[(ci >> 5) * 64u] >> (ci & 31u)) & 1u;
        bits_drop(br, cl); // (at least 17 bits are left: the extra bits of a length, 5 at most, or of a distance, 13 at most, need no refill)
        if (want_dist) { // ---- the distance: the match begins
            want_dist = false;
            if (sym > 29u) {
                err = GD_BAD_CODE;
                in_block = false;
                continue;
            }
            const uint32_t dt = dist_s[sym], xb = dt >> 16;
            const uint32_t dist = (dt & 0xFFFFu) + ((uint32_t)br.buf & ((1u << xb) - 1u));
            bits_drop(br, xb);
            if (dist > pos) { // (BGZF members carry no preset dictionary: nothing lies before the member's own output)
                err = GD_BAD_CODE;
                in_block = false;
                continue;
            }
            if (pos + m_len > o_cap) {
                err = GD_OVERRUN_OUT;
                in_block = false;
                continue;
            }
            cp_len = m_len;
            cp_dist = dist;
            cp_short = dist < 16u;
            if (cp_short) { // the period's bytes: the last `dist` of the sixteen kept in registers
                if (dist >= 8u) {
                    const uint32_t sh = 8u * (16u - dist); // 8..64
                    sp0 = sh == 64u ? h_hi : (h_lo >> sh) | (h_hi << (64u - sh));
                    sp1 = sh == 64u ? 0ull : h_hi >> sh; // (its first dist - 8 bytes are used)
                    sp_n = dist;
                } else {
                    const uint64_t v0 = h_hi >> (8u * (8u - dist));
                    uint64_t ext = v0 & ((1ull << (8u * dist)) - 1ull);
                    ext |= ext << (8u * dist);                  // 2 periods (dist < 8: the shifts stay below 64)
                    if (dist < 4u) ext |= ext << (16u * dist);  // 4
                    if (dist < 2u) ext |= ext << 32;            // 8
                    sp0 = ext;
                    sp1 = 0;
                    sp_n = (8u / dist) * dist;
                }
            }
            continue;
        }
        if (!is_hi) { // ---- a literal
            if (pos >= o_cap) {
                err = GD_OVERRUN_OUT;
                in_block = false;
                continue;
            }
            put(sym, 1u);
            continue;
        }
        if (sym == 0u) { // 256: end of block
            in_block = false;
            if (bits_overran(br)) err = GD_OVERRUN_IN;
            continue;
        }
        if (sym > 29u) { // (286, 287: in the fixed code, never in a stream)
            err = GD_BAD_CODE;
            in_block = false;
            continue;
        }
        { // ---- a length: its distance code is the next step's
            const uint32_t lt = len_s[sym - 1u], xb = lt >> 16;
            m_len = (lt & 0xFFFFu) + ((uint32_t)br.buf & ((1u << xb) - 1u));
            bits_drop(br, xb);
            want_dist = true;
        }
    }
    flush();
    if (err == GD_OK && bits_overran(br)) err = GD_OVERRUN_IN;
    if (err == GD_OK && pos != o_cap) err = GD_OVERRUN_OUT; // (ISIZE says how long the member's output is)
    status[b] = err;
}

} // namespace gd
} // namespace vgan

using namespace vgan::gd;

namespace vgan {

int gamdev_inflate(const uint8_t *d_in, const GdBlock *d_blocks, uint32_t n_blocks, uint8_t *d_out, uint32_t *d_status, hipStream_t st) {
    if (n_blocks == 0) return VGAN_OK;
    hipLaunchKernelGGL(gd_inflate_kernel, dim3((n_blocks + 63) / 64), dim3(64), 0, st, d_in, d_blocks, n_blocks, d_out, d_status);
    HIPCHK(hipGetLastError());
    return VGAN_OK;
}

} // namespace vgan
