// The GAM front end on the device (reference: src/readGAM.h:20-68, which goes through libvgio's BGZF stream and
// protobuf's parser, serially): a BGZF file's bytes go up as they are (whole, or in pieces: gam_pipe.hip runs the per-piece functions
// below as a pipeline), and
//
//   (inflate            gam_inflate_wave.hip: a wave per BGZF member, Huffman half and LZ77 half; gam_inflate_lane.hip: round 5's
//                       lane-per-member kernel, its fallback)
//   gd_anchor_kernel,   the framing of libvgio's stream ({count, count x (length, bytes)} groups, every group vg writes opened by the
//   gd_frame_kernel     item "GAM"), one lane per SEGMENT of the inflated bytes: a lane finds the first group tag in its segment, walks
//                       the items from there (csrc/host/gam.cpp: frame_segment) and goes on into the next segments until it
//                       stands exactly on the tag the next anchored segment started from -- every walk is then the true one, or the
//                       launch says so (a tag-like byte pattern inside a message: the host pipeline takes the file).
//   gd_count_kernel,    protobuf wire walk of vg.Alignment (csrc/host/gam.cpp: parse_alignment, field numbers SURVEY 8b), one lane
//   gd_fill_kernel,     per message: sizes first, then -- behind exclusive sums -- what is per read and a record per mapping (where its
//   gd_fill_maps_kernel bytes lie, where its edits go); then one lane per MAPPING writes the mappings' and edits' arrays
//                       hc_flatten_kernels.hip reads (a DfSlice: 32-bit offsets, node ids, edit lengths, substitution bytes).
//
// Integer / byte work throughout: every array is bit for bit what the host pipeline (gam.cpp + the narrowing of
// vgan_hc_devflat_run) hands the device flatten for the same file (tests/test_gamdev_gpu.py).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

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

// (the inflate kernels: gam_inflate_wave.hip -- a wave per member -- and gam_inflate_lane.hip, the older lane-per-member kernel it falls back to)
// eight bytes at any address (the hardware takes unaligned global accesses; the compiler is told so by the packed type)
struct __attribute__((packed)) GdU64 {
    uint64_t v;
};
__device__ __forceinline__ uint64_t gd_load8(const uint8_t *p) { return reinterpret_cast<const GdU64 *>(p)->v; }

// ------------------------------------------------------------------------------------------------------------------ framing
// libvgio's stream: groups {count, count x (length, bytes)}; the first item of every group vg writes is the tag "GAM".
constexpr uint32_t GD_TAG = 0x4D414703u; // the bytes 03 'G' 'A' 'M' as they sit in memory
constexpr uint64_t GD_NO_ANCHOR = ~0ull;
// (the GF_* status codes: gam_object.h)

__device__ __forceinline__ uint32_t gd_u32_at(const uint8_t *u, uint64_t n, uint64_t p) { // four bytes at any offset (zeros beyond n)
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) v |= (p + k < n ? (uint32_t)u[p + k] : 0u) << (8 * k);
    return v;
}
// 1: read (q is past it); 0: the bytes end inside it; -1: longer than ten bytes   (csrc/host/gam.cpp: get_varint)
__device__ __forceinline__ int gd_varint(const uint8_t *u, uint64_t e, uint64_t p, uint64_t &v, uint64_t &q) {
    v = 0;
    for (int shift = 0; shift <= 63; shift += 7) {
        if (p >= e) return 0;
        const uint8_t b = u[p++];
        v |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) {
            q = p;
            return 1;
        }
    }
    return -1;
}

// The same with the eight bytes at p in one load when the stream has them (a byte to a load, a message was three dependent loads and
// more: the walk of a 1 MiB segment is ~800 messages, one after the other); `word` / `have`: those bytes, for the caller to look at what
// follows the varint (a group's tag) without another load.
__device__ __forceinline__ int gd_varint_w(const uint8_t *u, uint64_t e, uint64_t p, uint64_t &v, uint64_t &q, uint64_t &word, bool &have) {
    have = p + 8 <= e;
    if (!have) return gd_varint(u, e, p, v, q);
    word = reinterpret_cast<const GdU64 *>(u + p)->v;
    // the first byte without its continuation bit ends the varint: bit 7 of byte k is bit 8k + 7
    const uint64_t stop = ~word & 0x8080808080808080ull;
    if (!stop) { // (eight bytes and more: lengths and counts never are; the byte loop says what it is)
        have = false;
        return gd_varint(u, e, p, v, q);
    }
    const uint32_t nb = ((uint32_t)__builtin_ctzll(stop) >> 3) + 1u; // bytes of the varint: 1..8
    uint64_t x = nb == 8u ? word : word & ((1ull << (8u * nb)) - 1ull), r = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) r |= ((x >> (8u * k)) & 0x7Full) << (7u * k);
    v = r;
    q = p + nb;
    return 1;
}
// the bytes [q, q + 3) "GAM", out of the word loaded at p when it holds them
__device__ __forceinline__ bool gd_is_gam(const uint8_t *u, uint64_t p, uint64_t q, uint64_t word, bool have) {
    if (have && q + 3 <= p + 8) return ((uint32_t)(word >> (8u * (uint32_t)(q - p))) & 0xFFFFFFu) == 0x4D4147u;
    return u[q] == 'G' && u[q + 1] == 'A' && u[q + 2] == 'M';
}

// One WAVE per segment: the first tag in [seg start (or 1), seg end); a match may begin in the segment and end beyond it.
// the first tag in [from, s1) (a wave searches: every lane gets the result), GD_NO_ANCHOR when there is none
__device__ __forceinline__ uint64_t gd_find_tag(const uint8_t *__restrict__ u, uint64_t n, uint64_t from, uint64_t s1, uint32_t lane) {
    // (four kilobytes to a round -- four loads a lane asked for before any is looked at: a round is a trip to memory, and a tag lies some
    // hundred kilobytes into the segment)
    constexpr int GD_FT = 4;
    for (uint64_t base = from & ~15ull; base < s1; base += GD_FT * 64u * 16u) {
        uint32_t w[GD_FT][5];
#pragma unroll
        for (int b = 0; b < GD_FT; ++b) {
            const uint64_t at = base + (uint64_t)b * 1024u + (uint64_t)lane * 16u;
            if (at + 20 <= n) {
                const uint4 v = *reinterpret_cast<const uint4 *>(u + at);
                w[b][0] = v.x, w[b][1] = v.y, w[b][2] = v.z, w[b][3] = v.w;
                w[b][4] = *reinterpret_cast<const uint32_t *>(u + at + 16);
            } else {
#pragma unroll
                for (int k = 0; k < 5; ++k) w[b][k] = gd_u32_at(u, n, at + 4u * k);
            }
        }
#pragma unroll
        for (int b = 0; b < GD_FT; ++b) {
            const uint64_t bb = base + (uint64_t)b * 1024u, at = bb + (uint64_t)lane * 16u;
            if (bb >= s1) break;
            uint32_t hit = 16;
#pragma unroll
            for (int i = 15; i >= 0; --i) {
                const uint32_t lo = w[b][i >> 2], hi = w[b][(i >> 2) + 1];
                const uint32_t win = (i & 3) ? (uint32_t)(((uint64_t)hi << 32 | lo) >> (8 * (i & 3))) : lo;
                const uint64_t pos = at + (uint64_t)i;
                if (win == GD_TAG && pos >= from && pos < s1 && pos >= 1) hit = (uint32_t)i;
            }
            const uint64_t m = __builtin_amdgcn_ballot_w64(hit < 16);
            if (m) {
                const uint32_t l0 = (uint32_t)__builtin_ctzll(m);
                const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)hit, (int)l0);
                return bb + (uint64_t)l0 * 16u + h0;
            }
        }
    }
    return GD_NO_ANCHOR;
}
__global__ __launch_bounds__(256) void gd_anchor_kernel(const uint8_t *__restrict__ u, uint64_t n, uint64_t seg_bytes, uint32_t n_segs,
                                                        uint64_t *__restrict__ anchor) {
    const uint32_t lane = threadIdx.x & 63u, seg = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (seg >= n_segs) return;
    const uint64_t s0 = (uint64_t)seg * seg_bytes, s1 = min(n, s0 + seg_bytes);
    if (seg == 0) { // the stream's own start: the walk begins in header state at offset 0, no tag is looked for
        if (lane == 0) anchor[0] = 0;
        return;
    }
    const uint64_t found = gd_find_tag(u, n, s0, s1, lane);
    if (lane == 0) anchor[seg] = found;
}
// One segment's anchor again, from behind the one it had: the walk in front of it did not arrive there -- those four bytes were not a
// group's tag but a message's own (a stream of random bytes holds them every 4 GB; a segment begins with ~300 KB that are not a tag)
__global__ __launch_bounds__(64) void gd_reanchor_kernel(const uint8_t *__restrict__ u, uint64_t n, uint64_t seg_bytes, uint32_t seg,
                                                         uint64_t *__restrict__ anchor) {
    const uint64_t s0 = (uint64_t)seg * seg_bytes, s1 = min(n, s0 + seg_bytes);
    const uint64_t old = anchor[seg];
    const uint64_t found = old == GD_NO_ANCHOR ? GD_NO_ANCHOR : gd_find_tag(u, n, max(s0, old + 1), s1, threadIdx.x);
    if (threadIdx.x == 0) anchor[seg] = found;
}

// One LANE per anchored segment: the walk from its tag to the next anchored segment's tag (or the stream's end).  emit = false:
// count the messages; true: write {offset, length} from msg_base[seg].
// A stream in pieces (gam_object.h: GdCarryState): segment 0 takes up the state the piece before left (`in`); with `open_end` (another
// piece follows) the walk that reaches the stream's end -- the one whose stop_tag is n -- ends wherever an item is cut by it and leaves its
// state in `carry`; a walk that runs into the end with a tag still in front of it has missed that tag.
struct GdWalkOut {
    uint32_t n_msg, status;
};
// A walk is a chain of loads, each at an address the one before gave: a hop is a trip to HBM (~2.3 us under this kernel's load; a segment's
// ~800 hops were 1.9 ms).  So a whole WAVE walks a segment -- every lane the same walk, the same addresses: one request -- and the lanes
// touch the stream ahead of it, a line each: the bytes of the next 16-32 KB are on their way to the L2 before the walk asks for them, and a
// hop is a trip there.  (The touches' values are folded into a word nobody reads: they are asked for at one touch and looked at at the
// next, when they have long arrived -- loads come back in the order they were sent, and a hop sent behind a touch waits for it.)
constexpr uint64_t GD_AHEAD_STEP = 16384;
struct GdAhead {
    uint64_t front = 0; // bytes below it have been touched
    uint64_t sink = 0, p0 = 0, p1 = 0;
    uint32_t lane = 0;
    bool on = false, all_write = false; // all_write: the walk's state goes to a place of each lane's own
};
__device__ __forceinline__ void gd_touch(GdAhead &h, const uint8_t *u, uint64_t n, uint64_t p) {
    if (!h.on || p + GD_AHEAD_STEP <= h.front || h.front >= n) return; // (the same in every lane)
    if (h.front < p) h.front = p & ~127ull;
    h.sink ^= h.p0 ^ h.p1;
    const uint64_t a0 = h.front + (uint64_t)h.lane * 128u, a1 = a0 + GD_AHEAD_STEP / 2;
    h.p0 = a0 + 8 <= n ? gd_load8(u + a0) : 0;
    h.p1 = a1 + 8 <= n ? gd_load8(u + a1) : 0;
    h.front += GD_AHEAD_STEP;
}
__device__ __forceinline__ void gd_touch_end(GdAhead &h) { // (the touches are loads the compiler may not drop)
    const uint64_t v = h.sink ^ h.p0 ^ h.p1;
    asm volatile("" ::"v"((uint32_t)v), "v"((uint32_t)(v >> 32)));
}
// (ahead: null -- one lane walks alone, as the kernels that walk once do -- or the wave's: every lane walks, lane 0 writes)
template <bool EMIT>
__device__ GdWalkOut gd_walk_segment(const uint8_t *u, uint64_t n, uint64_t start, uint32_t mode0, uint64_t rem0, uint32_t first0, uint64_t stop_tag,
                                     uint64_t *msg_off, uint32_t *msg_len, uint64_t out_base, bool open_end, GdCarryState *carry, GdAhead *ahead = nullptr) {
    uint64_t p = start, cnt = 0;
    uint64_t rem = rem0;
    bool in_group = mode0 == 1u, first = first0 != 0u;
    GdAhead none;
    GdAhead &ah = ahead ? *ahead : none;
    const bool writer = !ahead || ahead->lane == 0 || ahead->all_write;
    // the stream ends inside the item at p (or exactly in front of it)
    auto cut = [&](uint32_t mode) -> GdWalkOut {
        if (!open_end) return GdWalkOut{(uint32_t)cnt, GF_TRUNCATED};
        if (stop_tag != n) return GdWalkOut{(uint32_t)cnt, GF_MISSED};
        if (writer) *carry = GdCarryState{p, rem, mode, first ? 1u : 0u};
        return GdWalkOut{(uint32_t)cnt, GF_OK};
    };
    if (mode0 == 2u) {
        // the anchor's own group: its count lies before the tag and cannot be told from the previous message's last bytes, so its end
        // is recognised instead -- a count followed by the tag (no message starts with field number 0)   (gam.cpp: frame_segment)
        for (;;) {
            uint64_t v, q;
            gd_touch(ah, u, n, p);
            if (p >= n) {
                if (open_end) return cut(2u);
                return GdWalkOut{(uint32_t)cnt, stop_tag == n && p == n ? GF_OK : GF_TRUNCATED};
            }
            uint64_t word = 0;
            bool have = false;
            const int r = gd_varint_w(u, n, p, v, q, word, have);
            if (r < 0) return GdWalkOut{(uint32_t)cnt, GF_BAD_VARINT};
            if (r == 0) return cut(2u);
            if (open_end && q + 4 > n) return cut(2u); // (whether a tag follows cannot be told from these bytes alone)
            const uint32_t four = have && q + 4 <= p + 8 ? (uint32_t)(word >> (8u * (uint32_t)(q - p))) : gd_u32_at(u, n, q);
            if (four == GD_TAG && q + 4 <= n) { // p is a group header
                if (q == stop_tag) return GdWalkOut{(uint32_t)cnt, GF_OK};
                if (q > stop_tag) return GdWalkOut{(uint32_t)cnt, GF_MISSED};
                rem = v;
                first = true;
                in_group = v != 0;
                p = q;
                break;
            }
            if (v > n - q) return cut(2u);
            if (q + v > stop_tag) return GdWalkOut{(uint32_t)cnt, GF_MISSED}; // (the next anchor lies inside this item: one of the two tags is no tag)
            if (EMIT && writer) {
                msg_off[out_base + cnt] = q;
                msg_len[out_base + cnt] = (uint32_t)v;
            }
            if (v > 0xFFFFFFFFull) return GdWalkOut{(uint32_t)cnt, GF_BAD_MESSAGE};
            cnt += 1;
            p = q + v;
        }
    }
    for (;;) { // the walk proper (gam.cpp: walk)
        uint64_t v, q;
        gd_touch(ah, u, n, p);
        if (!in_group) {
            if (p == n) {
                if (open_end) return cut(0u);
                return GdWalkOut{(uint32_t)cnt, stop_tag == n ? GF_OK : GF_MISSED};
            }
            const int r = gd_varint(u, n, p, v, q);
            if (r < 0) return GdWalkOut{(uint32_t)cnt, GF_BAD_VARINT};
            if (r == 0) return cut(0u);
            rem = v;
            first = true;
            in_group = v != 0;
            p = q;
            // the next anchored segment starts on this group's tag (a piece that ends behind a group's count: the walk goes on, into the cut)
            if (in_group && p == stop_tag && !(open_end && stop_tag == n)) return GdWalkOut{(uint32_t)cnt, GF_OK};
            if (p > stop_tag) return GdWalkOut{(uint32_t)cnt, GF_MISSED};
            continue;
        }
        uint64_t word = 0;
        bool have = false;
        const int r = gd_varint_w(u, n, p, v, q, word, have);
        if (r < 0) return GdWalkOut{(uint32_t)cnt, GF_BAD_VARINT};
        if (r == 0 || v > n - q) return cut(1u);
        const bool tag = first && v == 3 && gd_is_gam(u, p, q, word, have);
        if (!tag) {
            if (v > 0xFFFFFFFFull) return GdWalkOut{(uint32_t)cnt, GF_BAD_MESSAGE};
            if (EMIT && writer) {
                msg_off[out_base + cnt] = q;
                msg_len[out_base + cnt] = (uint32_t)v;
            }
            cnt += 1;
        }
        first = false;
        p = q + v;
        if (p > stop_tag) return GdWalkOut{(uint32_t)cnt, GF_MISSED};
        if (--rem == 0) in_group = false;
    }
}

template <bool EMIT>
__global__ __launch_bounds__(64) void gd_frame_kernel(const uint8_t *__restrict__ u, uint64_t n, uint32_t n_segs, const uint64_t *__restrict__ anchor,
                                                      const uint64_t *__restrict__ next_anchor, uint32_t *__restrict__ seg_msgs,
                                                      const uint64_t *__restrict__ msg_base, uint64_t *__restrict__ msg_off, uint32_t *__restrict__ msg_len,
                                                      uint32_t *__restrict__ seg_status, GdCarryState in, int open_end, GdCarryState *__restrict__ carry) {
    const uint32_t seg = blockIdx.x, lane = threadIdx.x; // a wave per segment (gd_touch)
    if (seg >= n_segs) return;
    const uint64_t a = anchor[seg];
    if (a == GD_NO_ANCHOR) { // (its bytes are walked from the anchored segment before it)
        if (!EMIT && lane == 0) {
            seg_msgs[seg] = 0;
            seg_status[seg] = GF_OK;
        }
        return;
    }
    GdAhead ah;
    ah.lane = lane;
    ah.on = true;
    // segment 0 starts in the state the stream starts in (a file: before a group's count); the others on a tag, inside its group
    const GdWalkOut w = seg == 0 ? gd_walk_segment<EMIT>(u, n, a, in.mode, in.rem, in.first, next_anchor[seg], msg_off, msg_len, EMIT ? msg_base[seg] : 0,
                                                         open_end != 0, carry, &ah)
                                 : gd_walk_segment<EMIT>(u, n, a + 4, 2u, 0, 0u, next_anchor[seg], msg_off, msg_len, EMIT ? msg_base[seg] : 0, open_end != 0, carry, &ah);
    gd_touch_end(ah);
    if (!EMIT && lane == 0) {
        seg_msgs[seg] = w.n_msg;
        seg_status[seg] = w.status;
    }
}

// next_anchor[seg] = the tag position of the first anchored segment behind seg, or n (one thread: a few thousand segments)
__global__ void gd_next_anchor_kernel(const uint64_t *__restrict__ anchor, uint32_t n_segs, uint64_t n, uint64_t *__restrict__ next_anchor) {
    if (blockIdx.x || threadIdx.x) return;
    uint64_t nxt = n;
    for (uint32_t s = n_segs; s-- > 0;) {
        next_anchor[s] = nxt;
        if (anchor[s] != GD_NO_ANCHOR) nxt = anchor[s];
    }
}

// What a piece leaves to the next one, ahead of the piece's own framing: the walk from the LAST group tag of the stream to its end (one
// wave: tags are searched segment by segment from the back; the first lane walks).  The next piece's framing waits for this state and
// for nothing else of the piece -- the whole framing (two passes of walks a group long, a lane each) was what 25 pieces waited for one
// after the other: 6-12 ms each, half of `vgan euka`'s time on a 5 M-read file.  A tag is not taken on sight: the walk starts from the
// tag of the tagged segment BEFORE the last one and must arrive exactly on the last one's (tag-like bytes inside a message -- read names
// can hold anything -- send a walk anywhere but there); if it does not, nothing is said ahead (mode 0xFFFFFFFF) and the next piece waits
// for the piece's own framing, whose walks must meet all along, as before.  That framing has the last word in either case
// (gd_piece_parse compares the two states; a difference fails the run, and the host pipeline takes the file).
__global__ __launch_bounds__(64) void gd_tail_walk_kernel(const uint8_t *__restrict__ u, uint64_t n, uint64_t seg_bytes, uint32_t n_segs, GdCarryState in,
                                                          GdCarryState *__restrict__ carry) {
    const uint32_t lane = threadIdx.x;
    uint64_t a2 = GD_NO_ANCHOR, a1 = GD_NO_ANCHOR;
    for (uint32_t seg = n_segs; seg-- > 1u;) {
        const uint64_t s0 = (uint64_t)seg * seg_bytes, s1 = min(n, s0 + seg_bytes);
        const uint64_t a = gd_find_tag(u, n, s0, s1, lane);
        if (a == GD_NO_ANCHOR) continue;
        if (a2 == GD_NO_ANCHOR) {
            a2 = a;
        } else {
            a1 = a;
            break;
        }
    }
    // (every lane walks -- the same walk --, touching the stream ahead of it: gd_touch; the states are each lane's own copies)
    GdCarryState out{~0ull, 0, 0xFFFFFFFFu, 0}, none{};
    GdAhead ah;
    ah.lane = lane;
    ah.on = true;
    ah.all_write = true; // (the states are each lane's own: every lane writes its copy)
    if (a2 == GD_NO_ANCHOR) { // a short piece: the walk of all of it, from the state it starts in
        const GdWalkOut w = gd_walk_segment<false>(u, n, 0, in.mode, in.rem, in.first, n, nullptr, nullptr, 0, true, &out, &ah);
        if (w.status != GF_OK) out.mode = 0xFFFFFFFFu;
    } else {
        // from the tag before (or, with one tagged segment only, from the piece's start) the walk must stand on the last tag
        const GdWalkOut w1 = a1 == GD_NO_ANCHOR ? gd_walk_segment<false>(u, n, 0, in.mode, in.rem, in.first, a2, nullptr, nullptr, 0, true, &none, &ah)
                                                : gd_walk_segment<false>(u, n, a1 + 4, 2u, 0, 0u, a2, nullptr, nullptr, 0, true, &none, &ah);
        if (w1.status == GF_OK) {
            const GdWalkOut w2 = gd_walk_segment<false>(u, n, a2 + 4, 2u, 0, 0u, n, nullptr, nullptr, 0, true, &out, &ah);
            if (w2.status != GF_OK) out.mode = 0xFFFFFFFFu;
        }
    }
    gd_touch_end(ah);
    if (lane == 0) *carry = out;
}

// ------------------------------------------------------------------------------------------------------------------ parsing
// protobuf wire walk of vg.Alignment, statement for statement csrc/host/gam.cpp (Cur, parse_alignment, parse_mapping, parse_edit):
// which occurrence of a repeated scalar wins, which fields are appended, what is skipped and what makes a message malformed.
struct GdCur {
    const uint8_t *p, *e;
    bool ok;
};
// The walk's bytes come through a window of GD_WIN bytes per lane, kept in LDS: bytes [base, base + GD_WIN) of the inflated stream (which
// has 64 bytes of slack behind it), four 16-byte loads when a byte outside it is asked for.  (A byte to a load, every load of the wave
// touched 64 cache lines and the next byte waited for it: a message of 60 mappings was ~1000 dependent loads.  Eight bytes to a load --
// a window in a register pair, the form before this one -- every 128-byte line was still asked for sixteen times by its lane, and what
// the lanes of an XCD hold open at a time -- 32 CUs x 2 048 lanes x 128 bytes = 8 MB -- is twice its L2: the walks ran at what HBM gives
// for sixteen times their bytes, 1.8 and 2.4 ms for the two passes over 500 k messages.)
constexpr uint32_t GD_WIN = 64;
typedef uint32_t gd_v4u __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) GdU128 {
    gd_v4u v;
};
using gd_lds_u8p = __attribute__((address_space(3))) uint8_t *;
using gd_lds_u128p = __attribute__((address_space(3))) gd_v4u *;
struct GdWin {
    const uint8_t *base;
    gd_lds_u8p row; // this lane's GD_WIN bytes of LDS (16-byte aligned)
};
#define GD_WIN_ROWS(name) __shared__ uint4 name[256][GD_WIN / 16] // a workgroup's rows: one per thread
__device__ __forceinline__ gd_lds_u8p gd_win_row(uint4 (*rows)[GD_WIN / 16]) { return (gd_lds_u8p)(__attribute__((address_space(3))) void *)&rows[threadIdx.x][0]; }
__device__ __forceinline__ uint32_t gc_byte(GdWin &win, const uint8_t *p) {
    uint64_t d = (uint64_t)(p - win.base);
    if (d >= GD_WIN) {
        win.base = p;
        const GdU128 *src = reinterpret_cast<const GdU128 *>(p);
        const gd_v4u a0 = src[0].v, a1 = src[1].v, a2 = src[2].v, a3 = src[3].v;
        gd_lds_u128p dst = (gd_lds_u128p)win.row;
        dst[0] = a0, dst[1] = a1, dst[2] = a2, dst[3] = a3;
        d = 0;
    }
    return win.row[d];
}
__device__ __forceinline__ bool gc_done(const GdCur &c) { return c.p >= c.e; }
__device__ __forceinline__ uint64_t gc_varint(GdCur &c, GdWin &win) {
    if (c.p < c.e) {
        const uint32_t b0 = gc_byte(win, c.p);
        if (!(b0 & 0x80u)) {
            c.p += 1;
            return b0;
        }
    }
    uint64_t v = 0;
    int shift = 0;
    while (c.p < c.e) {
        const uint32_t b = gc_byte(win, c.p);
        c.p += 1;
        v |= (uint64_t)(b & 0x7fu) << shift;
        if (!(b & 0x80u)) return v;
        shift += 7;
        if (shift > 63) break;
    }
    c.ok = false;
    return 0;
}
__device__ __forceinline__ GdCur gc_sub(GdCur &c, GdWin &win) {
    const uint64_t n = gc_varint(c, win);
    if (!c.ok || n > (uint64_t)(c.e - c.p)) {
        c.ok = false;
        return GdCur{c.p, c.p, false};
    }
    GdCur s{c.p, c.p + n, true};
    c.p += n;
    return s;
}
__device__ __forceinline__ void gc_skip(GdCur &c, int wt, GdWin &win) {
    switch (wt) {
    case 0: (void)gc_varint(c, win); break;
    case 1:
        if (c.e - c.p >= 8) c.p += 8;
        else c.ok = false;
        break;
    case 2: (void)gc_sub(c, win); break;
    case 5:
        if (c.e - c.p >= 4) c.p += 4;
        else c.ok = false;
        break;
    default: c.ok = false;
    }
}

struct GdSizes { // per message
    uint32_t n_map, n_edit, eseq, qual, seq;
};
struct GdOut { // the arrays of one DfSlice (hc_flatten_kernels.hip) and what the host's duplicate marks need
    uint32_t *map_off, *qual_off, *edit_off, *e_seq_off, *m_node;
    int32_t *m_offset, *mapq, *e_len;
    uint8_t *unmapped, *m_rev, *e_seq, *qual;
    int64_t *first_node, *first_offset; // of the read's first mapping (-1, 0: no mapping): src/rmdup.cpp's key
    uint32_t *seq_len;                  // |Alignment.sequence| (euka, soibean: the damage tables' Lseq)
};

// (GdMapRec -- what the message pass leaves per mapping for the lane that fills its arrays: gam_object.h)

// One mapping's bytes (csrc/host/gam.cpp: parse_mapping, parse_edit): position, edits.  STORE false: its node / offset / strand and the
// counts of its edits and their sequence bytes; true: the edits' arrays too, from (e_at, s_at) on.  False when it is malformed.
template <bool STORE>
__device__ __forceinline__ bool gd_walk_mapping(GdCur mc, GdWin &win, int64_t &node, int64_t &off, uint8_t &rev, uint32_t &n_edit, uint32_t &n_eseq,
                                                const GdOut &o, uint32_t e_at, uint32_t s_at) {
    node = 0, off = 0, rev = 0, n_edit = 0, n_eseq = 0;
    while (!gc_done(mc) && mc.ok) {
        const uint64_t k3 = gc_varint(mc, win);
        const int f3 = (int)(k3 >> 3), w3 = (int)(k3 & 7);
        if (f3 == 1 && w3 == 2) { // position
            GdCur pc = gc_sub(mc, win);
            while (!gc_done(pc) && pc.ok) {
                const uint64_t k4 = gc_varint(pc, win);
                const int f4 = (int)(k4 >> 3), w4 = (int)(k4 & 7);
                if (f4 == 1 && w4 == 0) node = (int64_t)gc_varint(pc, win);
                else if (f4 == 2 && w4 == 0) off = (int64_t)gc_varint(pc, win);
                else if (f4 == 4 && w4 == 0) rev = gc_varint(pc, win) != 0;
                else gc_skip(pc, w4, win);
            }
            if (!pc.ok) return false;
        } else if (f3 == 2 && w3 == 2) { // an edit
            GdCur ec = gc_sub(mc, win);
            if (!mc.ok) return false;
            int32_t from = 0, to = 0;
            const uint8_t *sb = nullptr, *se = nullptr;
            while (!gc_done(ec) && ec.ok) {
                const uint64_t k4 = gc_varint(ec, win);
                const int f4 = (int)(k4 >> 3), w4 = (int)(k4 & 7);
                if (f4 == 1 && w4 == 0) from = (int32_t)gc_varint(ec, win);
                else if (f4 == 2 && w4 == 0) to = (int32_t)gc_varint(ec, win);
                else if (f4 == 3 && w4 == 2) {
                    GdCur sc = gc_sub(ec, win);
                    sb = sc.p;
                    se = sc.e;
                } else gc_skip(ec, w4, win);
            }
            if (!ec.ok) return false;
            const uint32_t nb = sb ? (uint32_t)(se - sb) : 0u;
            if (STORE) {
                o.e_len[e_at + n_edit] = from == to && from >= 0 ? from : -1;
                for (uint32_t k = 0; k < nb; ++k) o.e_seq[s_at + n_eseq + k] = sb[k];
                o.e_seq_off[e_at + n_edit + 1] = s_at + n_eseq + nb;
            }
            n_edit += 1;
            n_eseq += nb;
        } else gc_skip(mc, w3, win);
    }
    return mc.ok;
}

// FILL false: the message's sizes, its keep flag (identity != 0 or keep_unmapped) and whether it parses; true: what is per READ, written at
// the places the exclusive sums give it, and a GdMapRec per mapping -- the mappings' and edits' arrays are filled by a lane per MAPPING
// (gd_fill_maps_kernel: neighbouring lanes write neighbouring elements; a lane per message wrote six arrays at 64 places 240 bytes apart,
// every line of them filled over thirty stores and long gone from the L2 by then: 100 ms for the 10 M-read file around a walk of 26).
// WALK true: every mapping is walked here (its edits counted into sz.n_edit / sz.eseq); false: a mapping is hopped over -- its tag, its
// length -- and its inside is the per-mapping passes' (gd_map_count_kernel, gd_fill_maps_kernel), where the lanes of a wave sit at the same
// place of neighbouring mappings instead of at 64 different places of 64 messages' nested fields.
template <bool FILL, bool WALK>
__device__ bool gd_parse_message(const uint8_t *u, const uint8_t *mp, uint32_t mlen, GdSizes &sz, double &identity, int32_t &mapq, const GdOut &o,
                                 GdMapRec *recs, uint32_t m0, uint32_t e0, uint32_t s0, uint32_t q0, int64_t &first_node, int64_t &first_off,
                                 const uint8_t *&q_src, uint32_t &q_n, gd_lds_u8p row) {
    GdCur c{mp, mp + mlen, true};
    GdWin win{mp - GD_WIN, row}; // (nothing loaded yet: the first byte asked for moves it)
    q_src = nullptr; // FILL: the message's first quality string is left to the caller (the wave copies its lanes' strings together)
    q_n = 0;
    sz = GdSizes{0, 0, 0, 0, 0};
    identity = 0.0;
    mapq = 0;
    first_node = -1;
    first_off = 0;
    while (!gc_done(c) && c.ok) {
        const uint64_t key = gc_varint(c, win);
        const int f = (int)(key >> 3), wt = (int)(key & 7);
        if (f == 2 && wt == 2) {
            GdCur path = gc_sub(c, win);
            while (!gc_done(path) && path.ok) {
                const uint64_t k2 = gc_varint(path, win);
                const int f2 = (int)(k2 >> 3), w2 = (int)(k2 & 7);
                if (f2 == 2 && w2 == 2) { // a mapping
                    GdCur mc = gc_sub(path, win);
                    if (!path.ok || (uint64_t)(mc.e - mc.p) >= (1ull << 24)) return false; // (GdMapRec holds a mapping's length in 24 bits)
                    if (WALK || (FILL && sz.n_map == 0)) { // (the read's first mapping says its duplicate key: src/rmdup.cpp)
                        int64_t node, off;
                        uint8_t rev;
                        uint32_t ne, ns;
                        // (a first mapping that does not parse: the message pass that only hops goes on -- every mapping's record must be
                        // written -- and the per-mapping pass reports it)
                        if (!gd_walk_mapping<false>(mc, win, node, off, rev, ne, ns, o, 0, 0) && WALK) return false;
                        if (sz.n_map == 0) {
                            first_node = node;
                            first_off = off;
                        }
                        sz.n_edit += ne;
                        sz.eseq += ns;
                    }
                    if (FILL) recs[m0 + sz.n_map] = GdMapRec{(uint64_t)(mc.p - u) | (uint64_t)(mc.e - mc.p) << 40, 0u, 0u};
                    sz.n_map += 1;
                } else gc_skip(path, w2, win);
            }
            if (!path.ok) return false;
        } else if (f == 4 && wt == 2) {
            GdCur sc = gc_sub(c, win);
            const uint32_t nb = (uint32_t)(sc.e - sc.p);
            if (FILL && sc.ok) {
                if (!q_src && sz.qual == 0) {
                    q_src = sc.p;
                    q_n = nb;
                } else { // (a second quality field: appended, as the host parser does)
                    for (uint32_t k = 0; k < nb; ++k) o.qual[q0 + sz.qual + k] = sc.p[k];
                }
            }
            sz.qual += sc.ok ? nb : 0u;
        } else if (f == 5 && wt == 0) {
            mapq = (int32_t)gc_varint(c, win);
        } else if (f == 16 && wt == 1) {
            if (c.e - c.p < 8) return false;
            const uint64_t bits = gd_load8(c.p);
            identity = __longlong_as_double((long long)bits);
            c.p += 8;
        } else if (f == 1 && wt == 2) { // sequence: its length alone (a second one is appended, as the host parser does)
            const GdCur sc = gc_sub(c, win);
            sz.seq += sc.ok ? (uint32_t)(sc.e - sc.p) : 0u;
        } else if (f == 3 && wt == 2) { // name: not needed on the device
            (void)gc_sub(c, win);
        } else gc_skip(c, wt, win);
    }
    return c.ok;
}

// a lane per mapping, first pass: how many edits it has and how many bytes their sequences take (exclusive sums of the two say where every
// mapping's edits go: the first of them IS edit_off[]); a mapping that does not parse counts as a malformed message
__global__ __launch_bounds__(256) void gd_map_count_kernel(const uint8_t *__restrict__ u, const GdMapRec *__restrict__ recs, uint32_t n_maps, uint32_t *__restrict__ n_edit,
                                                           uint32_t *__restrict__ n_eseq, uint32_t *__restrict__ bad) {
    GD_WIN_ROWS(rows);
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m > n_maps) return;
    if (m == n_maps) { // (the sums' last input: their output there is the total)
        n_edit[m] = 0;
        n_eseq[m] = 0;
        return;
    }
    const GdMapRec r = recs[m];
    const uint8_t *p = u + (r.pos & ((1ull << 40) - 1ull));
    GdCur mc{p, p + (r.pos >> 40), true};
    GdWin win{p - GD_WIN, gd_win_row(rows)};
    int64_t node, off;
    uint8_t rev;
    uint32_t ne = 0, ns = 0;
    if (!gd_walk_mapping<false>(mc, win, node, off, rev, ne, ns, GdOut{}, 0, 0)) {
        atomicAdd(bad, 1u);
        ne = ns = 0;
    }
    n_edit[m] = ne;
    n_eseq[m] = ns;
}

// a lane per mapping, second pass: its node / offset / strand, its edits' lengths and sequence bytes, the offsets behind them.  s_at[m] (the
// mapping's first edit-sequence byte) lies where m_offset[m] will: read before it is written
__global__ __launch_bounds__(256) void gd_fill_maps_kernel(const uint8_t *__restrict__ u, const GdMapRec *__restrict__ recs, uint32_t n_maps, const uint32_t *s_at, GdOut o) {
    GD_WIN_ROWS(rows);
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m == 0) o.e_seq_off[0] = 0;
    if (m >= n_maps) return;
    const GdMapRec r = recs[m];
    const uint8_t *p = u + (r.pos & ((1ull << 40) - 1ull));
    GdCur mc{p, p + (r.pos >> 40), true};
    GdWin win{p - GD_WIN, gd_win_row(rows)};
    int64_t node, off;
    uint8_t rev;
    uint32_t ne, ns;
    const uint32_t e_at = o.edit_off[m], sa = s_at[m];
    (void)gd_walk_mapping<true>(mc, win, node, off, rev, ne, ns, o, e_at, sa);
    o.m_node[m] = node < 0 || node > 0xFFFFFFFEll ? 0xFFFFFFFFu : (uint32_t)node;
    o.m_offset[m] = off != (int64_t)(int32_t)off || (int32_t)off == INT32_MIN ? INT32_MIN : (int32_t)off;
    o.m_rev[m] = rev;
}

__global__ __launch_bounds__(256) void gd_count_kernel(const uint8_t *__restrict__ u, const uint64_t *__restrict__ msg_off, const uint32_t *__restrict__ msg_len,
                                                       uint32_t n_msg, int keep_unmapped, uint32_t *__restrict__ keep, uint32_t *__restrict__ n_map,
                                                       uint32_t *__restrict__ n_qual, uint32_t *__restrict__ bad) {
    GD_WIN_ROWS(rows);
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_msg) return;
    GdSizes sz;
    double identity;
    int32_t mapq;
    int64_t fn, fo;
    const uint8_t *qs;
    uint32_t qn;
    bool ok = gd_parse_message<false, false>(u, u + msg_off[i], msg_len[i], sz, identity, mapq, GdOut{}, nullptr, 0, 0, 0, 0, fn, fo, qs, qn, gd_win_row(rows));
    const bool kp = ok && (keep_unmapped || identity != 0.0); // readGAM.h:47: "Discard unmapped reads"
    // (a message that is dropped is never seen by the per-mapping passes: its mappings are walked here -- the host parser refuses a file
    // with a malformed mapping whether or not the read is kept)
    if (ok && !kp && sz.n_map) ok = gd_parse_message<false, true>(u, u + msg_off[i], msg_len[i], sz, identity, mapq, GdOut{}, nullptr, 0, 0, 0, 0, fn, fo, qs, qn, gd_win_row(rows));
    if (!ok) atomicAdd(bad, 1u);
    keep[i] = kp ? 1u : 0u;
    n_map[i] = kp ? sz.n_map : 0u;
    n_qual[i] = kp ? sz.qual : 0u;
}

__global__ __launch_bounds__(256) void gd_fill_kernel(const uint8_t *__restrict__ u, const uint64_t *__restrict__ msg_off, const uint32_t *__restrict__ msg_len,
                                                      uint32_t n_msg, const uint32_t *__restrict__ keep, const uint32_t *__restrict__ r_at,
                                                      const uint32_t *__restrict__ m_at, const uint32_t *__restrict__ q_at, GdOut o, GdMapRec *__restrict__ recs) {
    GD_WIN_ROWS(rows);
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) { // the offsets' leading zeros
        o.map_off[0] = 0;
        o.qual_off[0] = 0;
    }
    const uint8_t *q_src = nullptr;
    uint32_t q_n = 0, q_dst = 0;
    if (i < n_msg && keep[i]) {
        GdSizes sz;
        double identity;
        int32_t mapq;
        int64_t fn, fo;
        const uint32_t r = r_at[i];
        (void)gd_parse_message<true, false>(u, u + msg_off[i], msg_len[i], sz, identity, mapq, o, recs, m_at[i], 0, 0, q_at[i], fn, fo, q_src, q_n, gd_win_row(rows));
        q_dst = q_at[i];
        o.map_off[r + 1] = m_at[i] + sz.n_map;
        o.qual_off[r + 1] = q_at[i] + sz.qual;
        o.mapq[r] = mapq;
        o.unmapped[r] = identity < 1e-10 ? 1 : 0; // HaploCart.cpp:410
        o.first_node[r] = fn;
        o.first_offset[r] = fo;
        o.seq_len[r] = sz.seq;
    }
    // the quality strings of the wave's 64 messages, one after the other with all lanes: 64 bytes to a load (a lane copying its own string
    // byte by byte is a load and a store to 64 different cache lines per byte: two thirds of this kernel's time)
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t has = __builtin_amdgcn_ballot_w64(q_n != 0u);
    while (has) {
        const int l = __builtin_ctzll(has);
        has &= has - 1;
        const uint64_t sp = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)(uintptr_t)q_src >> 32), l) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uintptr_t)q_src, l);
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)q_n, l), d = (uint32_t)__builtin_amdgcn_readlane((int)q_dst, l);
        const uint8_t *src = reinterpret_cast<const uint8_t *>((uintptr_t)sp);
        for (uint32_t k = lane; k < n; k += 64u) o.qual[d + k] = src[k];
    }
}

// ------------------------------------------------------------------------------------------------------------------ duplicates
// src/rmdup.cpp:20-41,68-110 (single-end): a read is a duplicate when an EARLIER read has the same (node id, offset) in its first
// mapping.  Two stable radix sorts (by offset, then by node id) bring equal keys together in input order: every read of a run but
// its first is a duplicate.
__global__ __launch_bounds__(256) void gd_dup_keys_kernel(const int64_t *__restrict__ v, const uint32_t *__restrict__ perm, uint32_t n, uint64_t *__restrict__ key) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) key[i] = (uint64_t)v[perm ? perm[i] : i] ^ 0x8000000000000000ull; // (signed order, not that it matters: equal keys must meet)
}
__global__ __launch_bounds__(256) void gd_dup_mark_kernel(const uint32_t *__restrict__ perm, const int64_t *__restrict__ node, const int64_t *__restrict__ off,
                                                          const uint32_t *__restrict__ map_off, uint32_t n, uint8_t *__restrict__ dup, uint32_t *__restrict__ n_dup) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = perm[i];
    uint8_t d = 0;
    if (i > 0 && map_off[r + 1] > map_off[r]) { // (a read without mappings is no duplicate and no earlier read to anyone)
        const uint32_t q = perm[i - 1];
        d = map_off[q + 1] > map_off[q] && node[q] == node[r] && off[q] == off[r] ? 1 : 0;
        // (reads without mappings carry the key (-1, 0): they sort together, in front, and are skipped here)
    }
    dup[r] = d;
    if (d) atomicAdd(n_dup, 1u);
}

// A file in pieces: a read is a duplicate when an earlier read OF ANY PIECE SO FAR has its key.  `seen`: the keys of the pieces before,
// ascending (node first, signed -- the order the two sorts above give).  new_flag[i] (i: place in the piece's sorted order): the read
// brings a key nobody had.
__device__ __forceinline__ bool gd_key_less(int64_t an, int64_t ao, int64_t bn, int64_t bo) { return an < bn || (an == bn && ao < bo); }
__device__ __forceinline__ uint64_t gd_lower_bound(const int64_t *__restrict__ sn, const int64_t *__restrict__ so, uint64_t n, int64_t kn, int64_t ko) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (gd_key_less(sn[mid], so[mid], kn, ko)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint64_t gd_upper_bound(const int64_t *__restrict__ sn, const int64_t *__restrict__ so, uint64_t n, int64_t kn, int64_t ko) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (!gd_key_less(kn, ko, sn[mid], so[mid])) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(256) void gd_dup_mark_seen_kernel(const uint32_t *__restrict__ perm, const int64_t *__restrict__ node, const int64_t *__restrict__ off,
                                                               const uint32_t *__restrict__ map_off, uint32_t n, const int64_t *__restrict__ seen_node,
                                                               const int64_t *__restrict__ seen_off, uint64_t n_seen, uint8_t *__restrict__ dup,
                                                               uint32_t *__restrict__ new_flag, uint32_t *__restrict__ n_dup) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = perm[i];
    uint8_t d = 0;
    uint32_t nf = 0;
    if (map_off[r + 1] > map_off[r]) {
        if (i > 0) {
            const uint32_t q = perm[i - 1];
            d = map_off[q + 1] > map_off[q] && node[q] == node[r] && off[q] == off[r] ? 1 : 0;
        }
        if (!d) {
            const uint64_t lb = gd_lower_bound(seen_node, seen_off, n_seen, node[r], off[r]);
            d = lb < n_seen && seen_node[lb] == node[r] && seen_off[lb] == off[r] ? 1 : 0;
            nf = d ? 0u : 1u;
        }
    }
    dup[r] = d;
    new_flag[i] = nf;
    if (d) atomicAdd(n_dup, 1u);
}
__global__ __launch_bounds__(256) void gd_new_keys_kernel(const uint32_t *__restrict__ perm, const int64_t *__restrict__ node, const int64_t *__restrict__ off,
                                                          const uint32_t *__restrict__ new_flag, const uint32_t *__restrict__ new_at, uint32_t n,
                                                          int64_t *__restrict__ out_node, int64_t *__restrict__ out_off) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || !new_flag[i]) return;
    const uint32_t r = perm[i];
    out_node[new_at[i]] = node[r];
    out_off[new_at[i]] = off[r];
}
// two ascending key lists into one: an element's place is its own index plus the elements of the other list in front of it
__global__ __launch_bounds__(256) void gd_merge_keys_kernel(const int64_t *__restrict__ an, const int64_t *__restrict__ ao, uint64_t na, const int64_t *__restrict__ bn,
                                                            const int64_t *__restrict__ bo, uint64_t nb, int64_t *__restrict__ out_node, int64_t *__restrict__ out_off) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t < na) {
        const uint64_t at = t + gd_lower_bound(bn, bo, nb, an[t], ao[t]);
        out_node[at] = an[t];
        out_off[at] = ao[t];
    } else if (t < na + nb) {
        const uint64_t j = t - na, at = j + gd_upper_bound(an, ao, na, bn[j], bo[j]);
        out_node[at] = bn[j];
        out_off[at] = bo[j];
    }
}

// ------------------------------------------------------------------------------------------------------ messages back to the host
// The messages of the reads a mask names, one after the other in a buffer of their own (the reads the device flatten leaves to the
// host: their bytes go down, the host parses and flattens them as it always did).
__global__ __launch_bounds__(256) void gd_pick_kernel(const uint32_t *__restrict__ keep, const uint32_t *__restrict__ r_at, const uint8_t *__restrict__ read_mask,
                                                      const uint32_t *__restrict__ msg_len, uint32_t n_msg, uint32_t *__restrict__ pick, uint32_t *__restrict__ bytes) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_msg) return;
    const bool on = keep[i] && read_mask[r_at[i]];
    pick[i] = on ? 1u : 0u;
    bytes[i] = on ? msg_len[i] : 0u;
}
// the picked messages' numbers, in order (pick's exclusive sums say where)
__global__ __launch_bounds__(256) void gd_pick_list_kernel(const uint32_t *__restrict__ pick, const uint32_t *__restrict__ k_at, uint32_t n_msg, uint32_t *__restrict__ list) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n_msg && pick[i]) list[k_at[i]] = i;
}
// a wave per PICKED message (a wave per message of the file, most of them leaving at once, was 48 ms of launches for 10 M messages)
__global__ __launch_bounds__(256) void gd_gather_kernel(const uint8_t *__restrict__ u, const uint64_t *__restrict__ msg_off, const uint32_t *__restrict__ msg_len,
                                                        const uint32_t *__restrict__ list, const uint32_t *__restrict__ b_at, uint32_t n_picked,
                                                        uint8_t *__restrict__ out, uint64_t *__restrict__ out_off) {
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (wave == 0 && lane == 0) out_off[0] = 0;
    if (wave >= n_picked) return;
    const uint32_t i = list[wave], len = msg_len[i];
    const uint8_t *src = u + msg_off[i];
    uint8_t *dst = out + b_at[i];
    for (uint32_t k = lane; k < len; k += 64u) dst[k] = src[k];
    if (lane == 0) out_off[wave + 1] = (uint64_t)b_at[i] + len;
}

} // namespace gd
} // namespace vgan

using namespace vgan::gd;

// ------------------------------------------------------------------------------------------------------------ host side

namespace {
int gd_check_inflate(const uint8_t *d_in, const GdBlock *d_blocks, size_t n_blocks, uint8_t *d_out, uint32_t *d_status, hipStream_t st, uint64_t *n_redone,
                     const uint32_t *h_want, const uint32_t *d_tabs);
}
// Developer / test entry: inflates a BGZF file's bytes on the device and copies the result back (the host's bgzf_index says where the
// members lie).  Returns VGAN_EIO when a member does not inflate to its stated size.
extern "C" int vgan_gamdev_inflate_bytes(const void *bytes, uint64_t n, void *out, uint64_t out_cap, uint64_t *out_size, double *kernel_ms) {
    if (!bytes || !out_size) return fail(VGAN_EINVAL, "vgan_gamdev_inflate_bytes: null argument");
    std::vector<BgzfBlock> blocks;
    if (!bgzf_index((const unsigned char *)bytes, (size_t)n, blocks)) return fail(VGAN_EIO, "vgan_gamdev_inflate_bytes: not a BGZF stream");
    std::vector<GdBlock> gb;
    std::vector<uint32_t> want;
    uint64_t total = 0;
    for (const BgzfBlock &b : blocks) {
        const unsigned char *p = (const unsigned char *)bytes + b.in_off;
        const size_t xlen = p[10] | (p[11] << 8), hdr = 12 + xlen;
        gb.push_back(GdBlock{(uint64_t)(b.in_off + hdr), (uint64_t)b.out_off, (uint32_t)(b.in_size - hdr - 8), (uint32_t)b.out_size});
        const unsigned char *t = p + b.in_size - 8;
        want.push_back((uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24));
        total = b.out_off + b.out_size;
    }
    *out_size = total;
    if (!out) return VGAN_OK;
    if (out_cap < total) return fail(VGAN_EINVAL, "vgan_gamdev_inflate_bytes: the output buffer is too small");
    uint8_t *d_in = nullptr, *d_out = nullptr;
    GdBlock *d_b = nullptr;
    uint32_t *d_s = nullptr, *d_tok = nullptr, *d_nreg = nullptr, *d_cur = nullptr, *d_want = nullptr;
    void *d_reg = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = VGAN_OK;
    auto cleanup = [&] {
        for (void *q : {(void *)d_tok, (void *)d_nreg, (void *)d_cur, d_reg, (void *)d_want})
            if (q) (void)hipFree(q);
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
    GDCHK(hipMalloc((void **)&d_in, n + 64));
    GDCHK(hipMalloc((void **)&d_out, total + 16));
    GDCHK(hipMalloc((void **)&d_b, gb.size() * sizeof(GdBlock) + 16));
    GDCHK(hipMalloc((void **)&d_s, gb.size() * 4 + 16));
    GDCHK(hipMemcpy(d_in, bytes, n, hipMemcpyHostToDevice));
    GDCHK(hipMemcpy(d_b, gb.data(), gb.size() * sizeof(GdBlock), hipMemcpyHostToDevice));
    GDCHK(hipEventCreate(&e0));
    GDCHK(hipEventCreate(&e1));
    // (as gd_piece_upload_inflate: the two-kernel inflate, its leftovers through the older kernel; VGAN_GAMDEV_INFLATE=lane: the older one alone)
    const bool lane_inflate = getenv("VGAN_GAMDEV_INFLATE") && !strcmp(getenv("VGAN_GAMDEV_INFLATE"), "lane");
    const uint32_t tok_cap = (uint32_t)std::min<uint64_t>(0xFFFFFFF0ull, total * 3 / 10 + 65536);
    if (!lane_inflate) {
        GDCHK(hipMalloc((void **)&d_tok, (size_t)tok_cap * 4 + 16));
        GDCHK(hipMalloc((void **)&d_reg, gb.size() * 32 + 16));
        GDCHK(hipMalloc((void **)&d_nreg, gb.size() * 4 + 16));
        GDCHK(hipMalloc((void **)&d_cur, 16));
        GDCHK(hipMemset(d_cur, 0, 16));
    }
    GDCHK(hipEventRecord(e0, nullptr));
    rc = lane_inflate ? gamdev_inflate(d_in, d_b, (uint32_t)gb.size(), d_out, d_s, nullptr)
                      : gamdev_inflate_wave(d_in, d_b, (uint32_t)gb.size(), d_out, d_s, d_tok, tok_cap, d_cur, d_reg, d_nreg, nullptr);
    GDCHK(hipEventRecord(e1, nullptr));
    GDCHK(hipMalloc((void **)&d_want, gb.size() * 4 + 16 + GAMDEV_CRC_TABS * 4));
    GDCHK(hipMemcpy(d_want, want.data(), gb.size() * 4, hipMemcpyHostToDevice));
    uint32_t *d_tabs = d_want + ((gb.size() + 3) & ~(size_t)3);
    GDCHK(hipMemcpy(d_tabs, gamdev_crc_tables(), GAMDEV_CRC_TABS * 4, hipMemcpyHostToDevice));
    if (rc == VGAN_OK) rc = gamdev_crc(d_out, d_b, (uint32_t)gb.size(), d_want, d_tabs, d_s, nullptr);
    GDCHK(hipDeviceSynchronize());
    if (rc == VGAN_OK) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (kernel_ms) *kernel_ms = ms;
        uint64_t redone = 0;
        if ((rc = gd_check_inflate(d_in, d_b, gb.size(), d_out, d_s, nullptr, &redone, want.data(), d_tabs)) < 0) {
            cleanup();
            return rc;
        }
        if (getenv("VGAN_TIMING")) {
            double rounds = 0;
            if (d_nreg) {
                std::vector<uint32_t> nr(gb.size());
                GDCHK(hipMemcpy(nr.data(), d_nreg, gb.size() * 4, hipMemcpyDeviceToHost));
                for (uint32_t v : nr) rounds += v >> 8;
            }
            fprintf(stderr, "[vgan timing] vgan_gamdev_inflate_bytes: %zu members, %llu of them through the older kernel; %.2f rounds per member for the chunks' starts to settle\n",
                    gb.size(), (unsigned long long)redone, gb.empty() ? 0.0 : rounds / gb.size());
        }
        GDCHK(hipMemcpy(out, d_out, total, hipMemcpyDeviceToHost));
    }
#undef GDCHK
    cleanup();
    return rc;
}

// ------------------------------------------------------------------------------------------------------------ the C-ABI object
// (struct vgan_gamdev: gam_object.h)
namespace vgan {
namespace gd {
double g_alloc_ms = 0;
}
} // namespace vgan
static constexpr int GD_PIECES = vgan_gamdev::GD_PIECES;

extern "C" int vgan_gamdev_create(int device, void *hip_stream, vgan_gamdev **out) {
    if (!out) return fail(VGAN_EINVAL, "vgan_gamdev_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(VGAN_ENODEV, "vgan_gamdev_create: no HIP device is visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(VGAN_EINVAL, "vgan_gamdev_create: device %d out of range", device);
    HIPCHK(hipSetDevice(device));
    auto g = new vgan_gamdev();
    g->device = device;
    if (hip_stream) {
        g->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) {
            delete g;
            return fail(VGAN_ENODEV, "vgan_gamdev_create: hipStreamCreate failed");
        }
        g->own_stream = true;
    }
    *out = g;
    return VGAN_OK;
}

extern "C" void vgan_gamdev_free(vgan_gamdev *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    g->release_all();
    for (hipStream_t ps : g->piece_stream)
        if (ps) (void)hipStreamDestroy(ps);
    if (g->own_stream && g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

extern "C" int vgan_gamdev_drop_bytes(vgan_gamdev *g, int what) {
    if (!g || what < 1 || what > 2) return fail(VGAN_EINVAL, "vgan_gamdev_drop_bytes: null object or what not 1 / 2");
    HIPCHK(hipSetDevice(g->device));
    g->in.release();
    if (what >= 2) {
        g->infl.release();
        g->u = nullptr;
    }
    return VGAN_OK;
}

namespace {
int exclusive_sum(vgan_gamdev *g, const uint32_t *in, uint32_t *out, size_t n) {
    size_t tmp = 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, in, out, (int)n, g->stream) != hipSuccess) return fail(VGAN_ENODEV, "vgan_gamdev_parse: scan sizing failed");
    int rc;
    if ((rc = g->cub_tmp.reserve(tmp + 16))) return rc;
    if (hipcub::DeviceScan::ExclusiveSum(g->cub_tmp.p, tmp, in, out, (int)n, g->stream) != hipSuccess) return fail(VGAN_ENODEV, "vgan_gamdev_parse: scan failed");
    return VGAN_OK;
}
double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
} // namespace

namespace {
// the host's share of a parse: where the BGZF members lie (no HIP call: vgan_gamdev_open makes it while another thread brings the runtime up)
bool gd_index(const void *bytes, uint64_t n, std::vector<GdBlock> &gb, std::vector<uint32_t> &crcs, uint64_t &total) {
    std::vector<BgzfBlock> blocks;
    if (!bgzf_index((const unsigned char *)bytes, (size_t)n, blocks)) return false;
    gb.clear();
    crcs.clear();
    gb.reserve(blocks.size());
    for (const BgzfBlock &b : blocks) {
        const unsigned char *p = (const unsigned char *)bytes + b.in_off;
        const size_t xlen = p[10] | (p[11] << 8), hdr = 12 + xlen;
        if (b.out_size == 0) continue; // (the end-of-file member, empty members: nothing to write)
        gb.push_back(GdBlock{(uint64_t)(b.in_off + hdr), (uint64_t)b.out_off, (uint32_t)(b.in_size - hdr - 8), (uint32_t)b.out_size});
        const unsigned char *t = p + b.in_size - 8;
        crcs.push_back((uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24));
    }
    total = blocks.empty() ? 0 : blocks.back().out_off + blocks.back().out_size;
    return true;
}
int gd_parse_indexed(vgan_gamdev *g, const void *bytes, uint64_t n, const std::vector<GdBlock> &gb, const std::vector<uint32_t> &crcs, uint64_t total, int keep_unmapped);
} // namespace

// A BGZF GAM file's bytes -> the parser's arrays on the device (what vgan_gam_stream + the narrowing of vgan_hc_devflat_run make of the
// same file on the host).  VGAN_EIO: not BGZF, a member that does not inflate, a stream the segment walks cannot frame consistently
// (the caller takes the host pipeline), a malformed message.
extern "C" int vgan_gamdev_parse(vgan_gamdev *g, const void *bytes, uint64_t n, int keep_unmapped) {
    if (!g || (!bytes && n)) return fail(VGAN_EINVAL, "vgan_gamdev_parse: null argument");
    std::vector<GdBlock> gb;
    std::vector<uint32_t> crcs;
    uint64_t total = 0;
    if (!gd_index(bytes, n, gb, crcs, total)) return fail(VGAN_EIO, "vgan_gamdev_parse: not a BGZF stream");
    return gd_parse_indexed(g, bytes, n, gb, crcs, total, keep_unmapped);
}

// vgan_gamdev_create + vgan_gamdev_parse with the member index made first: a thread that calls this while another one makes the
// process's first HIP call has the index (a walk over every member's header, ~0.5 us each) ready when the runtime is.
extern "C" int vgan_gamdev_open(int device, void *hip_stream, const void *bytes, uint64_t n, int keep_unmapped, vgan_gamdev **out) {
    if (!out || (!bytes && n)) return fail(VGAN_EINVAL, "vgan_gamdev_open: null argument");
    *out = nullptr;
    std::vector<GdBlock> gb;
    std::vector<uint32_t> crcs;
    uint64_t total = 0;
    if (!gd_index(bytes, n, gb, crcs, total)) return fail(VGAN_EIO, "vgan_gamdev_open: not a BGZF stream");
    vgan_gamdev *g = nullptr;
    int rc = vgan_gamdev_create(device, hip_stream, &g);
    if (rc < 0) return rc;
    if ((rc = gd_parse_indexed(g, bytes, n, gb, crcs, total, keep_unmapped)) < 0) {
        vgan_gamdev_free(g);
        return rc;
    }
    *out = g;
    return VGAN_OK;
}

namespace {
// The members' statuses after the inflate kernels.  Whatever the two-kernel inflate did not finish (stored blocks, many blocks, no room in
// its scratch -- and whatever it calls an error) goes through the older kernel, a lane per member: an error is that kernel's to report.
int gd_check_inflate(const uint8_t *d_in, const GdBlock *d_blocks, size_t n_blocks, uint8_t *d_out, uint32_t *d_status, hipStream_t st, uint64_t *n_redone,
                     const uint32_t *h_want, const uint32_t *d_tabs) {
    if (n_blocks == 0) return VGAN_OK;
    std::vector<uint32_t> stt(n_blocks);
    HIPCHK(hipMemcpyAsync(stt.data(), d_status, n_blocks * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<uint32_t> again;
    for (size_t i = 0; i < n_blocks; ++i)
        if (stt[i] != GD_OK) again.push_back((uint32_t)i);
    if (again.empty()) return VGAN_OK;
    if (n_redone) *n_redone += again.size();
    if (getenv("VGAN_TIMING")) { // (developer aid: why the two-kernel inflate left them)
        size_t h[16] = {};
        for (uint32_t i : again) h[std::min<uint32_t>(stt[i], 15u)] += 1;
        fprintf(stderr, "[vgan timing] inflate: %zu of %zu members go through the older kernel; by status:", again.size(), n_blocks);
        for (int k = 0; k < 16; ++k)
            if (h[k]) fprintf(stderr, " %d: %zu", k, h[k]);
        fprintf(stderr, "\n");
    }
    std::vector<GdBlock> all(n_blocks), sub(again.size());
    HIPCHK(hipMemcpy(all.data(), d_blocks, n_blocks * sizeof(GdBlock), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < again.size(); ++k) sub[k] = all[again[k]];
    GdBlock *d_sub = nullptr;
    uint32_t *d_st = nullptr; // (statuses, then -- with a CRC check -- the trailers' values)
    if (hipMalloc((void **)&d_sub, sub.size() * sizeof(GdBlock)) != hipSuccess || hipMalloc((void **)&d_st, sub.size() * 8) != hipSuccess) {
        if (d_sub) (void)hipFree(d_sub);
        return fail(VGAN_ENOMEM, "vgan_gamdev_parse: no device memory for the members to inflate again");
    }
    std::vector<uint32_t> st2(sub.size(), 0xFFu);
    int rc = VGAN_OK;
    if (hipMemcpy(d_sub, sub.data(), sub.size() * sizeof(GdBlock), hipMemcpyHostToDevice) != hipSuccess) rc = fail(VGAN_ENODEV, "vgan_gamdev_parse: upload failed");
    if (rc == VGAN_OK) rc = gamdev_inflate(d_in, d_sub, (uint32_t)sub.size(), d_out, d_st, st);
    if (rc == VGAN_OK && h_want && d_tabs) {
        std::vector<uint32_t> w2(sub.size());
        for (size_t k = 0; k < again.size(); ++k) w2[k] = h_want[again[k]];
        if (hipMemcpy(d_st + sub.size(), w2.data(), sub.size() * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fail(VGAN_ENODEV, "vgan_gamdev_parse: upload failed");
        if (rc == VGAN_OK) rc = gamdev_crc(d_out, d_sub, (uint32_t)sub.size(), d_st + sub.size(), d_tabs, d_st, st);
    }
    if (rc == VGAN_OK && (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(st2.data(), d_st, sub.size() * 4, hipMemcpyDeviceToHost) != hipSuccess))
        rc = fail(VGAN_ENODEV, "vgan_gamdev_parse: the members inflated again: no status");
    (void)hipFree(d_sub);
    (void)hipFree(d_st);
    if (rc < 0) return rc;
    for (size_t k = 0; k < st2.size(); ++k)
        if (st2[k] != GD_OK) return fail(VGAN_EIO, "vgan_gamdev_parse: BGZF member %u does not inflate (code %u)", again[k], st2[k]);
    return VGAN_OK;
}

// a whole file as one piece that is the stream's first and last
int gd_parse_indexed(vgan_gamdev *g, const void *bytes, uint64_t n, const std::vector<GdBlock> &gb, const std::vector<uint32_t> &crcs, uint64_t total, int keep_unmapped) {
    // every offset the parse leaves is 32 bits wide and every array it fills is a subset of the inflated bytes (a mapping, an edit, a
    // message take at least a byte each): a stream below 2^32 bytes cannot wrap any of them.  Longer files go through in pieces
    // (vgan_gampipe_*: gam_pipe.hip), or through the host pipeline.
    if (total > 0xFFFFFFF0ull) return fail(VGAN_ERANGE, "vgan_gamdev_parse: %llu inflated bytes are beyond the parse's 32-bit offsets; parse the file in pieces", (unsigned long long)total);
    int rc;
    if ((rc = gd::gd_piece_upload_inflate(g, (const uint8_t *)bytes, n, gb.data(), crcs.data(), gb.size(), total, 0))) return rc;
    return gd::gd_piece_parse(g, GdCarry{}, true, nullptr, keep_unmapped, nullptr, nullptr);
}
} // namespace

int vgan::gd::gd_piece_upload_inflate(vgan_gamdev *g, const uint8_t *bytes, uint64_t n, const GdBlock *gb, const uint32_t *crcs, size_t n_gb, uint64_t total,
                                      uint64_t tail_cap) {
    HIPCHK(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    g->n_inflated = g->n_messages = g->R = g->M = g->E = g->S = g->Q = 0;
    g->n_inflated = total;
    g->n_blocks = n_gb;
    g->tail_cap = tail_cap;
    g->u = nullptr;
    g->n_stream = 0;
    g->inflate_checked = false;
    int rc;
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = g->in.reserve(n + 64)) || (rc = g->infl.reserve(tail_cap + total + 64)) || (rc = g->blocks.reserve(n_gb + 1)) || (rc = g->status.reserve(n_gb + 1))) return rc;
    uint8_t *d_out = g->infl.p + tail_cap;
    // the two-kernel inflate's scratch: ~0.22 tokens per output byte on GAM data (room for 0.3; a member that finds none goes to the older
    // kernel); the tokens borrow the parse's per-mapping records, which are written long after the inflate is done with them
    static const bool lane_inflate = getenv("VGAN_GAMDEV_INFLATE") && !strcmp(getenv("VGAN_GAMDEV_INFLATE"), "lane"); // (developer aid: the older kernel alone)
    const uint32_t tok_cap = (uint32_t)std::min<uint64_t>(0xFFFFFFF0ull, total * 3 / 10 + 65536);
    if (!lane_inflate) {
        if ((rc = g->map_rec.reserve(((size_t)tok_cap + 3) / 4)) || (rc = g->tok_reg.reserve(n_gb * 4 + 4)) || (rc = g->tok_nreg.reserve(n_gb + 1)) ||
            (rc = g->tok_cursor.reserve(4)))
            return rc;
        HIPCHK(hipMemsetAsync(g->tok_cursor.p, 0, 4, st));
    }
    if (n_gb) HIPCHK(hipMemcpyAsync(g->blocks.p, gb, n_gb * sizeof(GdBlock), hipMemcpyHostToDevice, st));
    g->h_crc.clear();
    if (crcs && n_gb) { // the trailers' CRC-32 and the kernel's tables
        g->h_crc.assign(crcs, crcs + n_gb);
        if ((rc = g->crc_want.reserve(n_gb + 1))) return rc;
        HIPCHK(hipMemcpyAsync(g->crc_want.p, g->h_crc.data(), n_gb * 4, hipMemcpyHostToDevice, st));
        if (!g->crc_tab.p) {
            if ((rc = g->crc_tab.reserve(GAMDEV_CRC_TABS))) return rc;
            HIPCHK(hipMemcpyAsync(g->crc_tab.p, gamdev_crc_tables(), GAMDEV_CRC_TABS * 4, hipMemcpyHostToDevice, st));
        }
    }
    // The bytes go up in a few parts, each on a stream of its own with the inflate of its members behind it: a part's kernel runs BESIDE
    // the next part's copy and the other parts' kernels, not before them (parts on ONE stream ran one after the other).
    g->ms_upload = 0;
    HIPCHK(hipStreamSynchronize(st)); // (the member list is up)
    uint64_t piece = std::max<uint64_t>(64ull << 20, n / GD_PIECES + 1);
    if (const char *e = getenv("VGAN_GAMDEV_PIECE")) piece = std::max<uint64_t>(1, strtoull(e, nullptr, 10)); // (test aid: many small parts)
    size_t b0 = 0, k = 0;
    uint64_t sent = 0;
    while (sent < n) {
        size_t b1 = b0;
        while (b1 < n_gb && gb[b1].in_off + gb[b1].in_size + 8 <= sent + piece) ++b1;
        if (b1 == b0 && b0 < n_gb) b1 = b0 + 1; // (a part below a member's size: one member at a time)
        const uint64_t upto = b1 < n_gb ? std::min<uint64_t>(n, gb[b1 - 1].in_off + gb[b1 - 1].in_size + 8) : n;
        hipStream_t ps = st;
        if (upto < n || k) { // (a file of one part stays on the object's stream)
            const size_t slot = k % GD_PIECES;
            if (!g->piece_stream[slot]) HIPCHK(hipStreamCreateWithFlags(&g->piece_stream[slot], hipStreamNonBlocking));
            ps = g->piece_stream[slot];
        }
        const auto tc = std::chrono::steady_clock::now();
        if (upto > sent) HIPCHK(hipMemcpyAsync(g->in.p + sent, bytes + sent, upto - sent, hipMemcpyHostToDevice, ps));
        g->ms_upload += ms_since(tc);
        sent = upto;
        if (b1 > b0) {
            if (lane_inflate) rc = gamdev_inflate(g->in.p, g->blocks.p + b0, (uint32_t)(b1 - b0), d_out, g->status.p + b0, ps);
            else rc = gamdev_inflate_wave(g->in.p, g->blocks.p + b0, (uint32_t)(b1 - b0), d_out, g->status.p + b0, reinterpret_cast<uint32_t *>(g->map_rec.p), tok_cap,
                                          g->tok_cursor.p, g->tok_reg.p + (size_t)b0 * 4, g->tok_nreg.p + b0, ps);
            if (rc) return rc;
            if (!g->h_crc.empty() && (rc = gamdev_crc(d_out, g->blocks.p + b0, (uint32_t)(b1 - b0), g->crc_want.p + b0, g->crc_tab.p, g->status.p + b0, ps))) return rc;
        }
        b0 = b1;
        ++k;
    }
    g->ms_inflate = ms_since(t0) - g->ms_upload; // (what the calls took: gd_piece_parse adds its wait)
    return VGAN_OK;
}

int vgan::gd::gd_piece_inflated(vgan_gamdev *g) {
    if (g->inflate_checked) return VGAN_OK;
    HIPCHK(hipSetDevice(g->device));
    const auto t0 = std::chrono::steady_clock::now();
    for (hipStream_t ps : g->piece_stream)
        if (ps) HIPCHK(hipStreamSynchronize(ps));
    int rc;
    if ((rc = gd_check_inflate(g->in.p, g->blocks.p, g->n_blocks, g->infl.p + g->tail_cap, g->status.p, g->stream, &g->n_redone, g->h_crc.empty() ? nullptr : g->h_crc.data(),
                               g->crc_tab.p)) < 0)
        return rc;
    g->ms_inflate += ms_since(t0);
    g->inflate_checked = true;
    return VGAN_OK;
}

int vgan::gd::gd_piece_parse(vgan_gamdev *g, const GdCarry &cin, bool last_piece, GdCarry *cout, int keep_unmapped, void (*frame_done)(void *), void *user) {
    HIPCHK(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    int rc;
    auto t0 = std::chrono::steady_clock::now();
    const uint64_t n_tail = cin.tail.size(), total = g->n_inflated + n_tail;
    struct Done { // the next piece's framing waits for this piece's: whatever way this call ends, it is told -- a failure by a state no walk leaves
        void (*fn)(void *);
        void *user;
        GdCarry *out;
        bool called = false;
        void operator()() {
            if (fn && !called) fn(user);
            called = true;
        }
        ~Done() {
            if (!called && out) out->st.mode = 0xFFFFFFFFu;
            (*this)();
        }
    } done{frame_done, user, cout};
    if (!last_piece && !cout) return fail(VGAN_EINVAL, "gd_piece_parse: a piece that is not the last needs a place for what it leaves over");
    if (n_tail > g->tail_cap) return fail(VGAN_ERANGE, "the GAM front end on the device: %llu bytes of an item are left over from the piece before, room was kept for %llu",
                                          (unsigned long long)n_tail, (unsigned long long)g->tail_cap);
    if (total > 0xFFFFFFF0ull) return fail(VGAN_ERANGE, "the GAM front end on the device: a piece of %llu inflated bytes is beyond the parse's 32-bit offsets", (unsigned long long)total);
    uint8_t *u = g->infl.p + (g->tail_cap - n_tail);
    if (n_tail) HIPCHK(hipMemcpyAsync(u, cin.tail.data(), n_tail, hipMemcpyHostToDevice, st));
    if ((rc = gd_piece_inflated(g)) < 0) return rc; // (done already by a caller that waits for its inflate before its turn)
    g->u = u;
    g->n_stream = total;
    if (cout) {
        cout->st = GdCarryState{total, 0, 0, 0};
        cout->tail.clear();
    }
    if (total == 0) {
        if (cout) cout->st = cin.st, cout->st.p = 0;
        done();
        return VGAN_OK;
    }
    // ---- framing
    t0 = std::chrono::steady_clock::now();
    const uint64_t seg_bytes = 1u << 20;
    const uint32_t n_segs = (uint32_t)((total + seg_bytes - 1) / seg_bytes);
    if ((rc = g->anchor.reserve(n_segs)) || (rc = g->next_anchor.reserve(n_segs)) || (rc = g->seg_msgs.reserve(n_segs)) || (rc = g->seg_status.reserve(n_segs)) ||
        (rc = g->msg_base.reserve(n_segs)) || (rc = g->carry.reserve(2)))
        return rc;
    const int open_end = last_piece ? 0 : 1;
    GdCarryState in = cin.st;
    in.p = 0;
    GdCarryState ahead{}; // what the next piece is told before this piece's own framing has run
    if (open_end) {
        hipLaunchKernelGGL(gd_tail_walk_kernel, dim3(1), dim3(64), 0, st, u, total, seg_bytes, n_segs, in, g->carry.p + 1);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&ahead, g->carry.p + 1, sizeof ahead, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (ahead.mode <= 2u && ahead.p <= total) {
            cout->st = ahead;
            cout->tail.resize((size_t)(total - ahead.p));
            if (total > ahead.p) HIPCHK(hipMemcpy(cout->tail.data(), u + ahead.p, (size_t)(total - ahead.p), hipMemcpyDeviceToHost));
            done(); // (the next piece's framing can start: it needs nothing else of this one)
        } // (else: the whole framing below finds out what is wrong with the stream, and says so)
    } else {
        done();
    }
    hipLaunchKernelGGL(gd_anchor_kernel, dim3((n_segs + 3) / 4), dim3(256), 0, st, u, total, seg_bytes, n_segs, g->anchor.p);
    std::vector<uint32_t> seg_n(n_segs), seg_st(n_segs);
    std::vector<uint64_t> nxt(n_segs);
    // The walks must meet: segment 0's walk is the true one, so the FIRST walk that misses the next anchored segment's tag says that
    // tag is not a group's -- that segment takes the next tag-like bytes behind it (or none: the walk in front goes on through it),
    // and the walks are counted again.  A walk that breaks (not: misses) is a malformed stream.
    for (uint32_t again = 0;; ++again) {
        hipLaunchKernelGGL(gd_next_anchor_kernel, dim3(1), dim3(1), 0, st, g->anchor.p, n_segs, total, g->next_anchor.p);
        hipLaunchKernelGGL(gd_frame_kernel<false>, dim3(n_segs), dim3(64), 0, st, u, total, n_segs, g->anchor.p, g->next_anchor.p, g->seg_msgs.p,
                           (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, g->seg_status.p, in, open_end, g->carry.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(seg_n.data(), g->seg_msgs.p, n_segs * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(seg_st.data(), g->seg_status.p, n_segs * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        uint32_t f = 0;
        while (f < n_segs && seg_st[f] == GF_OK) ++f;
        if (f == n_segs) break;
        if (seg_st[f] != GF_MISSED || again >= 256u)
            return fail(VGAN_EIO, "vgan_gamdev_parse: the stream cannot be framed from segment %u (code %u: %s)", f, seg_st[f],
                        seg_st[f] == GF_MISSED ? "a segment's walk does not meet the next one's group tag" : "malformed or truncated");
        HIPCHK(hipMemcpy(nxt.data(), g->next_anchor.p, n_segs * 8, hipMemcpyDeviceToHost));
        if (nxt[f] >= total) return fail(VGAN_EIO, "vgan_gamdev_parse: the stream cannot be framed from segment %u (its walk passes the stream's end)", f);
        const uint32_t suspect = (uint32_t)(nxt[f] / seg_bytes);
        hipLaunchKernelGGL(gd_reanchor_kernel, dim3(1), dim3(64), 0, st, u, total, seg_bytes, suspect, g->anchor.p);
        g->n_reanchored += 1;
    }
    if (open_end) { // what the piece leaves over: the state of the walk that reached its end -- the one the next piece was told ahead
        GdCarryState cs{};
        HIPCHK(hipMemcpy(&cs, g->carry.p, sizeof cs, hipMemcpyDeviceToHost));
        if (cs.p > total || cs.mode > 2u) return fail(VGAN_EIO, "vgan_gamdev_parse: the framing left no state at the piece's end");
        if (!done.called) { // (the walk ahead broke where the walks that meet did not: they have the last word)
            cout->st = cs;
            cout->tail.resize((size_t)(total - cs.p));
            if (total > cs.p) HIPCHK(hipMemcpy(cout->tail.data(), u + cs.p, (size_t)(total - cs.p), hipMemcpyDeviceToHost));
            done();
        } else if (cs.p != ahead.p || cs.mode != ahead.mode || (cs.mode == 1u && (cs.rem != ahead.rem || cs.first != ahead.first))) {
            return fail(VGAN_EIO, "vgan_gamdev_parse: the piece's last group tag (byte %llu) is no group's -- tag-like bytes inside a message at a piece's end; "
                                  "the next piece was framed from a state that does not hold",
                        (unsigned long long)ahead.p);
        }
    }
    std::vector<uint64_t> base(n_segs);
    uint64_t n_msg = 0;
    for (uint32_t s = 0; s < n_segs; ++s) {
        base[s] = n_msg;
        n_msg += seg_n[s];
    }
    if (n_msg > 0xFFFFFFF0ull) return fail(VGAN_ERANGE, "vgan_gamdev_parse: more than 2^32 messages");
    g->n_messages = n_msg;
    if (n_msg == 0) {
        g->ms_frame = ms_since(t0);
        return VGAN_OK;
    }
    if ((rc = g->msg_off.reserve(n_msg)) || (rc = g->msg_len.reserve(n_msg))) return rc;
    HIPCHK(hipMemcpyAsync(g->msg_base.p, base.data(), n_segs * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gd_frame_kernel<true>, dim3(n_segs), dim3(64), 0, st, u, total, n_segs, g->anchor.p, g->next_anchor.p, g->seg_msgs.p,
                       g->msg_base.p, g->msg_off.p, g->msg_len.p, g->seg_status.p, in, open_end, g->carry.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    g->ms_frame = ms_since(t0);
    // ---- parsing
    t0 = std::chrono::steady_clock::now();
    const uint32_t NM = (uint32_t)n_msg;
    for (auto *b : {&g->keep, &g->n_map, &g->n_edit, &g->n_eseq, &g->n_qual, &g->r_at, &g->m_at, &g->e_at, &g->s_at, &g->q_at})
        if ((rc = b->reserve((size_t)NM + 1))) return rc;
    if ((rc = g->bad.reserve(4))) return rc;
    HIPCHK(hipMemsetAsync(g->bad.p, 0, 4, st));
    hipLaunchKernelGGL(gd_count_kernel, dim3((NM + 255) / 256), dim3(256), 0, st, u, g->msg_off.p, g->msg_len.p, NM, keep_unmapped, g->keep.p, g->n_map.p,
                       g->n_qual.p, g->bad.p);
    HIPCHK(hipGetLastError());
    if ((rc = exclusive_sum(g, g->keep.p, g->r_at.p, NM)) || (rc = exclusive_sum(g, g->n_map.p, g->m_at.p, NM)) || (rc = exclusive_sum(g, g->n_qual.p, g->q_at.p, NM)))
        return rc;
    uint32_t last[6], bad = 0;
    {
        const uint32_t *srcs[6] = {g->keep.p, g->n_map.p, g->n_qual.p, g->r_at.p, g->m_at.p, g->q_at.p};
        for (int k = 0; k < 6; ++k) HIPCHK(hipMemcpyAsync(&last[k], srcs[k] + (NM - 1), 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&bad, g->bad.p, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    if (bad) return fail(VGAN_EIO, "vgan_gamdev_parse: %u malformed alignment messages", bad);
    // (32-bit sums: every array is a subset of the stream's bytes, which are fewer than 2^32 -- checked above -- so none of them wraps)
    g->R = (uint64_t)last[0] + last[3];
    g->M = (uint64_t)last[1] + last[4];
    g->Q = (uint64_t)last[2] + last[5];
    g->E = g->S = 0;
    if ((rc = g->map_off.reserve(g->R + 1)) || (rc = g->qual_off.reserve(g->R + 1)) || (rc = g->edit_off.reserve(g->M + 2)) || (rc = g->m_node.reserve(g->M + 1)) ||
        (rc = g->m_offset.reserve(g->M + 2)) || (rc = g->mapq.reserve(g->R + 1)) || (rc = g->unmapped.reserve(g->R + 1)) || (rc = g->m_rev.reserve(g->M + 1)) ||
        (rc = g->qual.reserve(g->Q + 1)) || (rc = g->first_node.reserve(g->R + 1)) || (rc = g->first_offset.reserve(g->R + 1)) || (rc = g->map_rec.reserve(g->M + 1)) ||
        (rc = g->seq_len.reserve(g->R + 1)))
        return rc;
    GdOut o{g->map_off.p, g->qual_off.p, g->edit_off.p, nullptr, g->m_node.p, g->m_offset.p, g->mapq.p, nullptr, g->unmapped.p, g->m_rev.p, nullptr,
            g->qual.p, g->first_node.p, g->first_offset.p, g->seq_len.p};
    // per message: what is per read, and where every mapping's bytes lie
    hipLaunchKernelGGL(gd_fill_kernel, dim3((NM + 255) / 256), dim3(256), 0, st, u, g->msg_off.p, g->msg_len.p, NM, g->keep.p, g->r_at.p, g->m_at.p, g->q_at.p, o,
                       g->map_rec.p);
    HIPCHK(hipGetLastError());
    // per mapping: its edits counted (into edit_off[] itself and, for their sequence bytes, into the words m_offset[] will take), the two
    // summed in place -- edit_off[] is then final --, the totals read, the edits' arrays asked for and filled
    uint32_t *s_at_m = reinterpret_cast<uint32_t *>(g->m_offset.p);
    const uint32_t NMAP = (uint32_t)g->M;
    hipLaunchKernelGGL(gd_map_count_kernel, dim3((NMAP + 1 + 255) / 256), dim3(256), 0, st, u, g->map_rec.p, NMAP, g->edit_off.p, s_at_m, g->bad.p);
    HIPCHK(hipGetLastError());
    if ((rc = exclusive_sum(g, g->edit_off.p, g->edit_off.p, (size_t)NMAP + 1)) || (rc = exclusive_sum(g, s_at_m, s_at_m, (size_t)NMAP + 1))) return rc;
    {
        uint32_t tot[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(&tot[0], g->edit_off.p + NMAP, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&tot[1], s_at_m + NMAP, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&bad, g->bad.p, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (bad) return fail(VGAN_EIO, "vgan_gamdev_parse: %u malformed alignment messages", bad);
        g->E = tot[0];
        g->S = tot[1];
    }
    if ((rc = g->e_seq_off.reserve(g->E + 1)) || (rc = g->e_len.reserve(g->E + 1)) || (rc = g->e_seq.reserve(g->S + 1))) return rc;
    o.e_seq_off = g->e_seq_off.p;
    o.e_len = g->e_len.p;
    o.e_seq = g->e_seq.p;
    hipLaunchKernelGGL(gd_fill_maps_kernel, dim3((NMAP + 1 + 255) / 256), dim3(256), 0, st, u, g->map_rec.p, NMAP, s_at_m, o);
    HIPCHK(hipGetLastError());
    // what vgan_gamdev_pick will want, asked for now: device memory asked for while another thread asks for the packed batch's 12 GB
    // and the device is busy took 180 ms there (its four arrays per message are the count arrays above, done with by then)
    if ((rc = g->picked_bytes.reserve(std::max<uint64_t>(g->R, total / 32 + 1))) || (rc = g->picked_off.reserve(NM / 32 + 2))) return rc;
    HIPCHK(hipStreamSynchronize(st));
    g->ms_parse = ms_since(t0);
    if (getenv("VGAN_TIMING") && g->tail_cap == 0) fprintf(stderr, "[vgan timing] gamdev: device memory asked for so far took %.1f ms\n", g_alloc_ms);
    return VGAN_OK;
}

// sizes[8]: inflated bytes, messages, reads, mappings, edits, edit-sequence bytes, quality bytes, 0;  ms[4]: upload, inflate, framing, parsing (wall)
extern "C" int vgan_gamdev_sizes(const vgan_gamdev *g, uint64_t sizes[8], double ms[4]) {
    if (!g) return fail(VGAN_EINVAL, "vgan_gamdev_sizes: null argument");
    if (sizes) {
        const uint64_t v[8] = {g->u ? g->n_stream : g->n_inflated, g->n_messages, g->R, g->M, g->E, g->S, g->Q, g->n_reanchored};
        memcpy(sizes, v, sizeof v);
    }
    if (ms) {
        ms[0] = g->ms_upload, ms[1] = g->ms_inflate, ms[2] = g->ms_frame, ms[3] = g->ms_parse;
    }
    return VGAN_OK;
}

// Copies one of the arrays of the last parse to the host (test aid).  which: 0 map_off[R+1] 1 qual_off[R+1] 2 edit_off[M+1] 3 e_seq_off[E+1]
// 4 m_node[M] 5 m_offset[M] 6 mapq[R] 7 e_len[E] 8 unmapped[R] 9 m_rev[M] 10 e_seq[S] 11 qual[Q] 12 first_node[R] 13 first_offset[R]
// 14 the inflated bytes 15 msg_off[messages] (uint64) 16 msg_len[messages] 17 seq_len[R] (|Alignment.sequence|)
extern "C" int vgan_gamdev_download(const vgan_gamdev *g, int which, void *dst) {
    if (!g || !dst) return fail(VGAN_EINVAL, "vgan_gamdev_download: null argument");
    HIPCHK(hipSetDevice(g->device));
    const void *src = nullptr;
    size_t bytes = 0;
    switch (which) {
    case 0: src = g->map_off.p, bytes = (g->R + 1) * 4; break;
    case 1: src = g->qual_off.p, bytes = (g->R + 1) * 4; break;
    case 2: src = g->edit_off.p, bytes = (g->M + 1) * 4; break;
    case 3: src = g->e_seq_off.p, bytes = (g->E + 1) * 4; break;
    case 4: src = g->m_node.p, bytes = g->M * 4; break;
    case 5: src = g->m_offset.p, bytes = g->M * 4; break;
    case 6: src = g->mapq.p, bytes = g->R * 4; break;
    case 7: src = g->e_len.p, bytes = g->E * 4; break;
    case 8: src = g->unmapped.p, bytes = g->R; break;
    case 9: src = g->m_rev.p, bytes = g->M; break;
    case 10: src = g->e_seq.p, bytes = g->S; break;
    case 11: src = g->qual.p, bytes = g->Q; break;
    case 12: src = g->first_node.p, bytes = g->R * 8; break;
    case 13: src = g->first_offset.p, bytes = g->R * 8; break;
    case 14: src = g->u, bytes = g->n_stream; break; // (a piece: what the piece before left over, then its own inflated bytes)
    case 15: src = g->msg_off.p, bytes = g->n_messages * 8; break;
    case 16: src = g->msg_len.p, bytes = g->n_messages * 4; break;
    case 17: src = g->seq_len.p, bytes = g->R * 4; break;
    default: return fail(VGAN_EINVAL, "vgan_gamdev_download: no array %d", which);
    }
    if (g->R == 0 && which <= 3) { // (an empty parse: the offsets' leading zero)
        memset(dst, 0, 4);
        return VGAN_OK;
    }
    if (bytes && src) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return VGAN_OK;
}

bool vgan::gamdev_slice(const vgan_gamdev *g, GamdevSlice *o) {
    if (!g || !o) return false;
    *o = GamdevSlice{g->map_off.p, g->qual_off.p, g->edit_off.p, g->e_seq_off.p, g->m_node.p, g->m_offset.p, g->mapq.p, g->e_len.p, g->unmapped.p, g->m_rev.p,
                     g->e_seq.p, g->qual.p, g->first_node.p, g->first_offset.p, g->seq_len.p, g->R, g->device};
    return true;
}

// Duplicate marks of the last parse's reads (keep-first by the first mapping's (node id, offset): src/rmdup.cpp), left on the device for
// vgan_hc_devflat_run_gamdev (vgan_gamdev_dup_marks: the device pointer) and counted.
namespace {
// two ascending lists on the device -> seen's other pair of buffers, which becomes the current one
int gd_seen_take(vgan_gamdev *g, GdSeen &seen, const int64_t *d_node, const int64_t *d_off, uint64_t nk) {
    if (nk == 0) return VGAN_OK;
    hipStream_t st = g->stream;
    const int cur = seen.cur, oth = cur ^ 1;
    const uint64_t total = seen.n + nk;
    int rc;
    if (total > seen.node[oth].cap || !seen.node[oth].p) { // (room for twice what is needed: a set that grows piece by piece is re-allocated log times)
        if ((rc = seen.node[oth].reserve(total * 2)) || (rc = seen.off[oth].reserve(total * 2))) return rc;
    }
    hipLaunchKernelGGL(gd_merge_keys_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, seen.node[cur].p, seen.off[cur].p, seen.n, d_node, d_off, nk,
                       seen.node[oth].p, seen.off[oth].p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st)); // (the lists given may go when this returns)
    seen.cur = oth;
    seen.n = total;
    return VGAN_OK;
}
} // namespace

int vgan::gd::gd_seen_merge(vgan_gamdev *g, GdSeen &seen, const GdKeyList &keys) {
    const uint64_t nk = keys.node.size();
    if (nk == 0) return VGAN_OK;
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = g->new_node.reserve(nk)) || (rc = g->new_off.reserve(nk))) return rc;
    HIPCHK(hipMemcpyAsync(g->new_node.p, keys.node.data(), nk * 8, hipMemcpyHostToDevice, g->stream));
    HIPCHK(hipMemcpyAsync(g->new_off.p, keys.off.data(), nk * 8, hipMemcpyHostToDevice, g->stream));
    return gd_seen_take(g, seen, g->new_node.p, g->new_off.p, nk);
}

int vgan::gd::gd_piece_mark_duplicates(vgan_gamdev *g, GdSeen &seen, GdKeyList *added, int64_t *n_dup) {
    if (n_dup) *n_dup = 0;
    if (added) added->node.clear(), added->off.clear();
    if (g->R == 0) return VGAN_OK;
    HIPCHK(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    const uint32_t R = (uint32_t)g->R;
    int rc;
    if ((rc = g->dup.reserve(R)) || (rc = g->sort_key.reserve(R)) || (rc = g->sort_key2.reserve(R)) || (rc = g->perm_a.reserve(R)) || (rc = g->perm_b.reserve(R)) ||
        (rc = g->bad.reserve(4)) || (rc = g->new_flag.reserve(R)) || (rc = g->new_at.reserve(R)))
        return rc;
    // perm_a = 0, 1, 2, ... (an exclusive sum of ones)
    {
        size_t tmp = 0;
        auto ones = hipcub::ConstantInputIterator<uint32_t>(1u);
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, ones, g->perm_a.p, (int)R, st) != hipSuccess) return fail(VGAN_ENODEV, "vgan_gamdev_mark_duplicates: scan sizing failed");
        if ((rc = g->cub_tmp.reserve(tmp + 16))) return rc;
        if (hipcub::DeviceScan::ExclusiveSum(g->cub_tmp.p, tmp, ones, g->perm_a.p, (int)R, st) != hipSuccess) return fail(VGAN_ENODEV, "vgan_gamdev_mark_duplicates: scan failed");
    }
    auto sort_by = [&](const int64_t *values, const uint32_t *perm_in, uint32_t *perm_out) -> int {
        hipLaunchKernelGGL(gd_dup_keys_kernel, dim3((R + 255) / 256), dim3(256), 0, st, values, perm_in, R, g->sort_key.p);
        size_t tmp = 0;
        if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, g->sort_key.p, g->sort_key2.p, perm_in, perm_out, (int)R, 0, 64, st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_gamdev_mark_duplicates: sort sizing failed");
        int rc2;
        if ((rc2 = g->cub_tmp.reserve(tmp + 16))) return rc2;
        if (hipcub::DeviceRadixSort::SortPairs(g->cub_tmp.p, tmp, g->sort_key.p, g->sort_key2.p, perm_in, perm_out, (int)R, 0, 64, st) != hipSuccess)
            return fail(VGAN_ENODEV, "vgan_gamdev_mark_duplicates: sort failed");
        return VGAN_OK;
    };
    if ((rc = sort_by(g->first_offset.p, g->perm_a.p, g->perm_b.p)) || (rc = sort_by(g->first_node.p, g->perm_b.p, g->perm_a.p))) return rc;
    HIPCHK(hipMemsetAsync(g->bad.p, 0, 4, st));
    hipLaunchKernelGGL(gd_dup_mark_seen_kernel, dim3((R + 255) / 256), dim3(256), 0, st, g->perm_a.p, g->first_node.p, g->first_offset.p, g->map_off.p, R,
                       seen.node[seen.cur].p, seen.off[seen.cur].p, seen.n, g->dup.p, g->new_flag.p, g->bad.p);
    HIPCHK(hipGetLastError());
    if ((rc = exclusive_sum(g, g->new_flag.p, g->new_at.p, R))) return rc;
    uint32_t nd = 0, last[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(&nd, g->bad.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&last[0], g->new_flag.p + (R - 1), 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&last[1], g->new_at.p + (R - 1), 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (n_dup) *n_dup = nd;
    const uint64_t nk = (uint64_t)last[0] + last[1];
    if (nk == 0) return VGAN_OK;
    if ((rc = g->new_node.reserve(nk)) || (rc = g->new_off.reserve(nk))) return rc;
    hipLaunchKernelGGL(gd_new_keys_kernel, dim3((R + 255) / 256), dim3(256), 0, st, g->perm_a.p, g->first_node.p, g->first_offset.p, g->new_flag.p, g->new_at.p, R,
                       g->new_node.p, g->new_off.p);
    HIPCHK(hipGetLastError());
    if (added) {
        added->node.resize(nk), added->off.resize(nk);
        HIPCHK(hipMemcpyAsync(added->node.data(), g->new_node.p, nk * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(added->off.data(), g->new_off.p, nk * 8, hipMemcpyDeviceToHost, st));
    }
    return gd_seen_take(g, seen, g->new_node.p, g->new_off.p, nk);
}

extern "C" int vgan_gamdev_mark_duplicates(vgan_gamdev *g, int64_t *n_dup) {
    if (!g) return fail(VGAN_EINVAL, "vgan_gamdev_mark_duplicates: null argument");
    GdSeen none; // (a file in one go: nothing was seen before it)
    const int rc = gd::gd_piece_mark_duplicates(g, none, nullptr, n_dup);
    (void)hipSetDevice(g->device);
    none.release();
    return rc;
}

extern "C" const uint8_t *vgan_gamdev_dup_marks(const vgan_gamdev *g) { return g ? g->dup.p : nullptr; }

// The messages of the reads read_mask names (host, per read of the last parse) are put one after the other on the device; sizes
// come back here, the bytes and their offsets ([n + 1]) through vgan_gamdev_picked.
extern "C" int vgan_gamdev_pick(vgan_gamdev *g, const uint8_t *read_mask, uint64_t *n_msgs, uint64_t *n_bytes) {
    if (!g || !read_mask || !n_msgs || !n_bytes) return fail(VGAN_EINVAL, "vgan_gamdev_pick: null argument");
    *n_msgs = *n_bytes = 0;
    g->n_picked = g->n_picked_bytes = 0;
    if (g->R == 0 || g->n_messages == 0) return VGAN_OK;
    if (!g->infl.p || !g->u) return fail(VGAN_EINVAL, "vgan_gamdev_pick: the inflated bytes were given back (vgan_gamdev_drop_bytes)");
    HIPCHK(hipSetDevice(g->device));
    hipStream_t st = g->stream;
    const uint32_t NM = (uint32_t)g->n_messages;
    // the parse's four count arrays (one word per message, done with when the parse ends) hold the flags, the lengths and their sums
    uint32_t *pick = g->n_map.p, *pick_bytes = g->n_edit.p, *k_at = g->n_eseq.p, *b_at = g->n_qual.p;
    int rc;
    if ((rc = g->picked_bytes.reserve(g->R))) return rc;
    uint8_t *d_mask = g->picked_bytes.p; // (borrowed for the mask until the bytes' size is known)
    HIPCHK(hipMemcpyAsync(d_mask, read_mask, g->R, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gd_pick_kernel, dim3((NM + 255) / 256), dim3(256), 0, st, g->keep.p, g->r_at.p, d_mask, g->msg_len.p, NM, pick, pick_bytes);
    if ((rc = exclusive_sum(g, pick, k_at, NM)) || (rc = exclusive_sum(g, pick_bytes, b_at, NM))) return rc;
    uint32_t last[4];
    HIPCHK(hipMemcpyAsync(&last[0], pick + (NM - 1), 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&last[1], k_at + (NM - 1), 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&last[2], pick_bytes + (NM - 1), 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&last[3], b_at + (NM - 1), 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    g->n_picked = (uint64_t)last[0] + last[1];
    g->n_picked_bytes = (uint64_t)last[2] + last[3];
    *n_msgs = g->n_picked;
    *n_bytes = g->n_picked_bytes;
    if (g->n_picked == 0) return VGAN_OK;
    // (the mask's buffer is read by nothing any more: the pick flags hold what it said; the lengths' array takes the list)
    if ((rc = g->picked_off.reserve(g->n_picked + 1))) return rc;
    if ((rc = g->picked_bytes.reserve(std::max<uint64_t>(g->n_picked_bytes, g->R)))) return rc;
    uint32_t *list = pick_bytes;
    hipLaunchKernelGGL(gd_pick_list_kernel, dim3((NM + 255) / 256), dim3(256), 0, st, pick, k_at, NM, list);
    hipLaunchKernelGGL(gd_gather_kernel, dim3((uint32_t)((g->n_picked * 64 + 255) / 256)), dim3(256), 0, st, g->u, g->msg_off.p, g->msg_len.p, list, b_at,
                       (uint32_t)g->n_picked, g->picked_bytes.p, g->picked_off.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return VGAN_OK;
}

extern "C" int vgan_gamdev_picked(const vgan_gamdev *g, uint64_t *offsets, uint8_t *bytes) {
    if (!g || !offsets || !bytes) return fail(VGAN_EINVAL, "vgan_gamdev_picked: null argument");
    if (g->n_picked == 0) {
        offsets[0] = 0;
        return VGAN_OK;
    }
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(hipMemcpy(offsets, g->picked_off.p, (g->n_picked + 1) * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(bytes, g->picked_bytes.p, g->n_picked_bytes, hipMemcpyDeviceToHost));
    return VGAN_OK;
}
#include "module_anchor.h"
const void *vgan::anchor_gam_kernels() { return (const void *)&vgan::gd::gd_next_anchor_kernel; }
