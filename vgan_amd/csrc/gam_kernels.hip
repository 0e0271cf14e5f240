// The GAM front end on the device (reference: src/readGAM.h:20-68, which goes through libvgio's BGZF stream and
// protobuf's parser, serially): a BGZF file's bytes go up as they are, and
//
//   gd_inflate_kernel   DEFLATE (RFC 1951), one LANE per BGZF block: blocks are independent members of at most 64 KB.  A lane
//                       decodes serially as a CPU would (canonical Huffman decoding bit by bit: the counts per code length
//                       sit in registers, the symbol tables of the lane in LDS); a match is copied from the lane's own
//                       output in groups of eight independent loads.  A lane is slow -- a 64 KB block takes milliseconds --
//                       but a 10 M-read file is 75 000 blocks: every lane of the chip has one.
//   gd_anchor_kernel,   the framing of libvgio's stream ({count, count x (length, bytes)} groups, every group vg writes opened by the
//   gd_frame_kernel     item "GAM"), one lane per SEGMENT of the inflated bytes: a lane finds the first group tag in its segment, walks
//                       the items from there (csrc/host/gam.cpp: frame_segment) and goes on into the next segments until it
//                       stands exactly on the tag the next anchored segment started from -- every walk is then the true one, or the
//                       launch says so (a tag-like byte pattern inside a message: the host pipeline takes the file).
//   gd_count_kernel,    protobuf wire walk of vg.Alignment (csrc/host/gam.cpp: parse_alignment, field numbers SURVEY 8b), one lane
//   gd_fill_kernel      per message: sizes first, then -- behind exclusive sums -- the arrays hc_flatten_kernels.hip reads
//                       (a DfSlice: 32-bit offsets, node ids, edit lengths, quality and substitution bytes) written in place.
//
// Integer / byte work throughout: every array is bit for bit what the host pipeline (gam.cpp + the narrowing of
// vgan_hc_devflat_run) hands the device flatten for the same file (tests/test_gamdev_gpu.py).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "gam_device.h"
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
constexpr int INF_LIT = 288, INF_DIST = 30, INF_ROW = 322; // symbols per lane: 288 + 30, padded to an odd number of dwords (161)
enum : uint32_t { GD_OK = 0, GD_BAD_BLOCK = 1, GD_BAD_CODE = 2, GD_OVERRUN_IN = 3, GD_OVERRUN_OUT = 4, GD_BAD_STORED = 5 };

struct Bits { // LSB-first bit reader over 4-byte words (the payload is followed by the member's 8-byte trailer: a word read past
    const uint8_t *p; // the payload's end stays inside the file)
    const uint8_t *e;
    uint64_t buf;
    uint32_t cnt;
    bool over;
};
__device__ __forceinline__ void bits_init(Bits &b, const uint8_t *p, const uint8_t *e) {
    b.p = p;
    b.e = e;
    b.buf = 0;
    b.cnt = 0;
    b.over = false;
    while (((uintptr_t)b.p & 3u) && b.p < b.e) { // up to the first aligned word
        b.buf |= (uint64_t)*b.p++ << b.cnt;
        b.cnt += 8;
    }
}
__device__ __forceinline__ void bits_fill(Bits &b) { // at least 32 bits, or what is left
    if (b.cnt < 32) {
        if (b.p + 4 <= b.e + 8) { // (the trailer is readable)
            uint32_t w = *reinterpret_cast<const uint32_t *>(b.p);
            const int64_t left = b.e - b.p;
            if (left < 4) {
                if (left <= 0) {
                    w = 0;
                    b.over = b.over || b.cnt == 0; // (nothing real left at all)
                } else {
                    w &= (1u << (8 * left)) - 1u;
                }
            }
            b.buf |= (uint64_t)w << b.cnt;
            b.p += 4;
            b.cnt += 32;
        } else {
            b.cnt += 32;
            b.over = true;
        }
    }
}
__device__ __forceinline__ uint32_t bits_get(Bits &b, uint32_t n) { // n <= 16
    bits_fill(b);
    const uint32_t v = (uint32_t)b.buf & ((1u << n) - 1u);
    b.buf >>= n;
    b.cnt -= n;
    return v;
}
// bits consumed beyond the payload's end? (cnt counts phantom zero bits once p passed e)
__device__ __forceinline__ bool bits_overran(const Bits &b) {
    const int64_t avail = (int64_t)(b.e - b.p) * 8 + (int64_t)b.cnt;
    return b.over || avail < 0;
}

// counts per code length 0..15, packed four to a 64-bit word (registers: the decode loop reads them without touching memory)
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

// canonical Huffman decode, one bit at a time (RFC 1951 3.2.2): -1 on an invalid code
__device__ __forceinline__ int huff_decode(Bits &b, const Counts &c, const uint16_t *sym) {
    bits_fill(b);
    uint32_t code = 0, first = 0, index = 0;
    uint64_t bitbuf = b.buf;
    uint32_t left = b.cnt;
    for (uint32_t len = 1; len <= 15; ++len) {
        if (left == 0) return -1; // (fill gives >= 32 bits: not reached for codes of <= 15 bits)
        code |= (uint32_t)bitbuf & 1u;
        bitbuf >>= 1;
        left -= 1;
        const uint32_t count = cnt_get(c, len);
        if (code < first + count) {
            b.buf = bitbuf;
            b.cnt = left;
            return (int)sym[index + (code - first)];
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// builds {counts, symbols in canonical order} from code lengths; returns false for an over-subscribed set (an incomplete one
// is allowed where RFC 1951 allows it: a single distance code)
__device__ bool huff_build(const uint8_t *lengths, int n, Counts &c, uint16_t *sym) {
    c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
    for (int s = 0; s < n; ++s) cnt_add(c, lengths[s], 1u);
    int left = 1;
    for (uint32_t len = 1; len <= 15; ++len) {
        left <<= 1;
        left -= (int)cnt_get(c, len);
        if (left < 0) return false;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (uint32_t len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + cnt_get(c, len));
    for (int s = 0; s < n; ++s)
        if (lengths[s] != 0) sym[offs[lengths[s]]++] = (uint16_t)s;
    // (codes of length 0 are not codes: their count must not take part in decoding)
    c.w[0] &= ~0xFFFFull;
    return true;
}

__device__ const uint16_t gd_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t gd_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t gd_dist_base[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                              193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t gd_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t gd_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// the lane's own earlier output, read past the vector L1 (a line the lane loaded before it stored into it may sit there)
__device__ __forceinline__ uint8_t out_byte(const uint8_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(64) void gd_inflate_kernel(const uint8_t *__restrict__ in, const GdBlock *__restrict__ blocks, uint32_t n_blocks,
                                                        uint8_t *out, uint32_t *__restrict__ status) {
    __shared__ uint16_t sym_s[64][INF_ROW];
    const uint32_t lane = threadIdx.x, b = blockIdx.x * 64u + lane;
    if (b >= n_blocks) return;
    const GdBlock bl = blocks[b];
    uint16_t *lsym = sym_s[lane], *dsym = lsym + INF_LIT;
    uint8_t *o = out + bl.out_off;
    const uint32_t o_cap = bl.out_size;
    uint32_t pos = 0, err = GD_OK;
    Bits br;
    bits_init(br, in + bl.in_off, in + bl.in_off + bl.in_size);
    uint8_t lengths[INF_LIT + INF_DIST + 2]; // (scratch)
    for (bool last = false; !last && err == GD_OK;) {
        last = bits_get(br, 1) != 0;
        const uint32_t type = bits_get(br, 2);
        if (type == 0) { // stored
            const uint32_t drop = br.cnt & 7u;
            br.buf >>= drop;
            br.cnt -= drop;
            const uint32_t len = bits_get(br, 16), nlen = bits_get(br, 16);
            if ((len ^ 0xFFFFu) != nlen) {
                err = GD_BAD_STORED;
                break;
            }
            if (pos + len > o_cap) {
                err = GD_OVERRUN_OUT;
                break;
            }
            for (uint32_t i = 0; i < len; ++i) o[pos + i] = (uint8_t)bits_get(br, 8);
            pos += len;
            continue;
        }
        if (type == 3) {
            err = GD_BAD_BLOCK;
            break;
        }
        Counts lc, dc;
        if (type == 1) { // fixed codes
            for (int s = 0; s < 144; ++s) lengths[s] = 8;
            for (int s = 144; s < 256; ++s) lengths[s] = 9;
            for (int s = 256; s < 280; ++s) lengths[s] = 7;
            for (int s = 280; s < 288; ++s) lengths[s] = 8;
            (void)huff_build(lengths, 288, lc, lsym);
            for (int s = 0; s < 30; ++s) lengths[s] = 5;
            (void)huff_build(lengths, 30, dc, dsym);
        } else { // dynamic codes
            const uint32_t nlen = bits_get(br, 5) + 257, ndist = bits_get(br, 5) + 1, ncode = bits_get(br, 4) + 4;
            if (nlen > 286 || ndist > 30) {
                err = GD_BAD_BLOCK;
                break;
            }
            for (int s = 0; s < 19; ++s) lengths[s] = 0;
            for (uint32_t i = 0; i < ncode; ++i) lengths[gd_clen_order[i]] = (uint8_t)bits_get(br, 3);
            Counts cc;
            if (!huff_build(lengths, 19, cc, lsym)) { // (the code-length code borrows the literal table's room)
                err = GD_BAD_BLOCK;
                break;
            }
            uint32_t idx = 0;
            while (idx < nlen + ndist) {
                const int s = huff_decode(br, cc, lsym);
                if (s < 0) {
                    err = GD_BAD_CODE;
                    break;
                }
                if (s < 16) {
                    lengths[idx++] = (uint8_t)s;
                } else {
                    uint32_t rep, val = 0;
                    if (s == 16) {
                        if (idx == 0) {
                            err = GD_BAD_BLOCK;
                            break;
                        }
                        val = lengths[idx - 1];
                        rep = 3 + bits_get(br, 2);
                    } else if (s == 17) {
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
            if (err != GD_OK) break;
            if (lengths[256] == 0 || !huff_build(lengths, (int)nlen, lc, lsym) || !huff_build(lengths + nlen, (int)ndist, dc, dsym)) {
                err = GD_BAD_BLOCK;
                break;
            }
        }
        // ---- the block's symbols
        for (;;) {
            const int s = huff_decode(br, lc, lsym);
            if (s < 0) {
                err = GD_BAD_CODE;
                break;
            }
            if (s < 256) {
                if (pos >= o_cap) {
                    err = GD_OVERRUN_OUT;
                    break;
                }
                o[pos++] = (uint8_t)s;
                continue;
            }
            if (s == 256) break;
            if (s > 285) {
                err = GD_BAD_CODE;
                break;
            }
            const uint32_t len = gd_len_base[s - 257] + bits_get(br, gd_len_extra[s - 257]);
            const int ds = huff_decode(br, dc, dsym);
            if (ds < 0 || ds > 29) {
                err = GD_BAD_CODE;
                break;
            }
            const uint32_t dist = gd_dist_base[ds] + bits_get(br, gd_dist_extra[ds]);
            if (dist > pos) { // (BGZF members carry no preset dictionary: nothing lies before the member's own output)
                err = GD_BAD_CODE;
                break;
            }
            if (pos + len > o_cap) {
                err = GD_OVERRUN_OUT;
                break;
            }
            const uint8_t *src = o + pos - dist;
            uint8_t *dst = o + pos;
            if (dist >= 8) { // groups of eight independent loads (a group never reaches into bytes it writes itself)
                uint32_t i = 0;
                for (; i + 8 <= len; i += 8) {
                    uint8_t t[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = out_byte(src + i + k);
#pragma unroll
                    for (int k = 0; k < 8; ++k) dst[i + k] = t[k];
                }
                for (; i < len; ++i) dst[i] = out_byte(src + i);
            } else { // a short period: the pattern is read once and repeated out of registers
                uint8_t pat[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pat[k] = k < (int)dist ? out_byte(src + k) : 0;
                uint32_t k = 0;
                for (uint32_t i = 0; i < len; ++i) {
                    uint8_t v = pat[0];
#pragma unroll
                    for (int j = 1; j < 8; ++j) v = k == (uint32_t)j ? pat[j] : v;
                    dst[i] = v;
                    k = k + 1 == dist ? 0 : k + 1;
                }
            }
            pos += len;
        }
        if (err == GD_OK && bits_overran(br)) err = GD_OVERRUN_IN;
    }
    if (err == GD_OK && pos != o_cap) err = GD_OVERRUN_OUT; // (ISIZE says how long the member's output is)
    status[b] = err;
}

} // namespace gd
} // namespace vgan

using namespace vgan::gd;

// ------------------------------------------------------------------------------------------------------------ host side
namespace vgan {

int gamdev_inflate(const uint8_t *d_in, const GdBlock *d_blocks, uint32_t n_blocks, uint8_t *d_out, uint32_t *d_status, hipStream_t st) {
    if (n_blocks == 0) return VGAN_OK;
    hipLaunchKernelGGL(gd_inflate_kernel, dim3((n_blocks + 63) / 64), dim3(64), 0, st, d_in, d_blocks, n_blocks, d_out, d_status);
    HIPCHK(hipGetLastError());
    return VGAN_OK;
}

} // namespace vgan

// Developer / test entry: inflates a BGZF file's bytes on the device and copies the result back (the host's bgzf_index says where the
// members lie).  Returns VGAN_EIO when a member does not inflate to its stated size.
extern "C" int vgan_gamdev_inflate_bytes(const void *bytes, uint64_t n, void *out, uint64_t out_cap, uint64_t *out_size, double *kernel_ms) {
    if (!bytes || !out_size) return fail(VGAN_EINVAL, "vgan_gamdev_inflate_bytes: null argument");
    std::vector<BgzfBlock> blocks;
    if (!bgzf_index((const unsigned char *)bytes, (size_t)n, blocks)) return fail(VGAN_EIO, "vgan_gamdev_inflate_bytes: not a BGZF stream");
    std::vector<GdBlock> gb;
    uint64_t total = 0;
    for (const BgzfBlock &b : blocks) {
        const unsigned char *p = (const unsigned char *)bytes + b.in_off;
        const size_t xlen = p[10] | (p[11] << 8), hdr = 12 + xlen;
        gb.push_back(GdBlock{(uint64_t)(b.in_off + hdr), (uint64_t)b.out_off, (uint32_t)(b.in_size - hdr - 8), (uint32_t)b.out_size});
        total = b.out_off + b.out_size;
    }
    *out_size = total;
    if (!out) return VGAN_OK;
    if (out_cap < total) return fail(VGAN_EINVAL, "vgan_gamdev_inflate_bytes: the output buffer is too small");
    uint8_t *d_in = nullptr, *d_out = nullptr;
    GdBlock *d_b = nullptr;
    uint32_t *d_s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = VGAN_OK;
    auto cleanup = [&] {
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        if (d_b) (void)hipFree(d_b);
        if (d_s) (void)hipFree(d_s);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define GDCHK(expr)                                                                       \
    do {                                                                                  \
        if ((expr) != hipSuccess) {                                                       \
            cleanup();                                                                    \
            return fail(VGAN_ENODEV, "vgan_gamdev_inflate_bytes: %s failed", #expr);      \
        }                                                                                 \
    } while (0)
    GDCHK(hipMalloc((void **)&d_in, n + 16));
    GDCHK(hipMalloc((void **)&d_out, total + 16));
    GDCHK(hipMalloc((void **)&d_b, gb.size() * sizeof(GdBlock) + 16));
    GDCHK(hipMalloc((void **)&d_s, gb.size() * 4 + 16));
    GDCHK(hipMemcpy(d_in, bytes, n, hipMemcpyHostToDevice));
    GDCHK(hipMemcpy(d_b, gb.data(), gb.size() * sizeof(GdBlock), hipMemcpyHostToDevice));
    GDCHK(hipEventCreate(&e0));
    GDCHK(hipEventCreate(&e1));
    GDCHK(hipEventRecord(e0, nullptr));
    rc = gamdev_inflate(d_in, d_b, (uint32_t)gb.size(), d_out, d_s, nullptr);
    GDCHK(hipEventRecord(e1, nullptr));
    GDCHK(hipDeviceSynchronize());
    if (rc == VGAN_OK) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (kernel_ms) *kernel_ms = ms;
        std::vector<uint32_t> stt(gb.size());
        GDCHK(hipMemcpy(stt.data(), d_s, gb.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < stt.size(); ++i)
            if (stt[i] != GD_OK) {
                cleanup();
                return fail(VGAN_EIO, "vgan_gamdev_inflate_bytes: BGZF member %zu does not inflate (code %u)", i, stt[i]);
            }
        GDCHK(hipMemcpy(out, d_out, total, hipMemcpyDeviceToHost));
    }
#undef GDCHK
    cleanup();
    return rc;
}
