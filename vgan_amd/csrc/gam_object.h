// The GAM front end's device object (gam_kernels.hip makes and fills it; gam_pipe.hip runs several of them as a pipeline over the
// file's pieces).  Private to the library: the C-ABI sees an opaque vgan_gamdev.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>

#include "gam_device.h"
#include "host/common.h"
#include "vgan_gpu.h"

namespace vgan {
namespace gd {

enum : uint32_t { GD_OK = 0, GD_BAD_BLOCK = 1, GD_BAD_CODE = 2, GD_OVERRUN_IN = 3, GD_OVERRUN_OUT = 4, GD_BAD_STORED = 5, GD_BAD_CRC = 6, GD_PUNT = 7 /* gd_tokens_kernel leaves the member to gd_inflate_kernel */ };
enum : uint32_t { GF_OK = 0, GF_BAD_VARINT = 1, GF_MISSED = 2, GF_TRUNCATED = 3, GF_BAD_MESSAGE = 4 };

struct GdMapRec { // what the message pass leaves per mapping for the lane that fills its arrays
    uint64_t pos;   // the mapping's bytes: offset in the inflated stream | length << 40
    uint32_t e_at;  // its first edit's index
    uint32_t s_at;  // its first edit-sequence byte's index
};
static_assert(sizeof(GdMapRec) == 16, "one 16-byte store per mapping");

// Where a piece's framing walk stood when the piece's bytes ended (libvgio's stream: groups {count, count x (length, bytes)}): the
// bytes from `p` on -- an item the piece holds the beginning of -- go in front of the next piece's inflated bytes, whose walk takes up
// the state.  mode 0: before a group's count; 1: inside a group, `rem` items to go (`first`: the next one may be the group's tag);
// 2: inside a group whose count the walk never saw (it started from a tag found in the bytes: that group ends where a count is followed by
// a tag).
struct GdCarryState {
    uint64_t p, rem;
    uint32_t mode, first;
};
struct GdCarry {
    GdCarryState st{0, 0, 0, 0};
    std::vector<uint8_t> tail; // the stream's bytes from st.p on (host)
};

extern double g_alloc_ms; // (VGAN_TIMING: what hipMalloc took, summed)
template <class T> struct GBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap && p) return VGAN_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const auto t_alloc = std::chrono::steady_clock::now();
        struct Acc {
            std::chrono::steady_clock::time_point t0;
            ~Acc() { g_alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
        } acc{t_alloc};
        const size_t want = n + std::min<size_t>(n / 8, ((size_t)16 << 20) / sizeof(T)) + 64;
        const hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            return fail(e == hipErrorOutOfMemory ? VGAN_ENOMEM : VGAN_ENODEV, "the GAM front end on the device: hipMalloc of %zu bytes failed: %s", want * sizeof(T),
                        hipGetErrorString(e));
        }
        cap = want;
        // (test aid: fresh device memory is often zero in a young process and someone's old data in an old one -- a kernel that leaves
        // an entry unwritten passes every test but the one that runs late)
        static const bool poison = getenv("VGAN_POISON_ALLOCS") != nullptr;
        if (poison) {
            if (hipMemset(p, 0xA5, want * sizeof(T)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) // (the fill runs on the null stream, the kernels on others)
                return fail(VGAN_ENODEV, "the GAM front end on the device: hipMemset failed");
        }
        return VGAN_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// The (node id, offset) keys of the first mappings seen so far, in ascending order (signed, node first): what makes a read of a later
// piece a duplicate of a read of an earlier one (src/rmdup.cpp keeps the FIRST read of a key).  One per pipeline lane, on its device.
struct GdSeen {
    GBuf<int64_t> node[2], off[2];
    int cur = 0;
    uint64_t n = 0;
    size_t merged = 0; // how many of the pipeline's key lists (one per piece, kept on the host) this set holds
    void release() {
        for (int k = 0; k < 2; ++k) node[k].release(), off[k].release();
        n = 0;
    }
};
struct GdKeyList { // the keys a piece added (host copy: the other lanes' sets take them up)
    int lane = -1;
    std::vector<int64_t> node, off;
};

} // namespace gd
} // namespace vgan

#ifndef GD_PIECES_N
#define GD_PIECES_N 4
#endif

struct vgan_gamdev {
    static constexpr int GD_PIECES = GD_PIECES_N;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    vgan::gd::GBuf<uint8_t> in, infl, cub_tmp;
    vgan::gd::GBuf<vgan::GdBlock> blocks;
    vgan::gd::GBuf<uint32_t> status, seg_msgs, seg_status, msg_len, keep, n_map, n_edit, n_eseq, n_qual, r_at, m_at, e_at, s_at, q_at, bad;
    vgan::gd::GBuf<uint64_t> anchor, next_anchor, msg_base, msg_off;
    vgan::gd::GBuf<vgan::gd::GdCarryState> carry;
    vgan::gd::GBuf<uint64_t> tok_reg;              // gd_tokens_kernel -> gd_lz_kernel: {first token, tokens} per block of a member (the tokens themselves borrow map_rec)
    vgan::gd::GBuf<uint32_t> tok_nreg, tok_cursor; // blocks per member; the scratch's fill
    vgan::gd::GBuf<uint32_t> crc_want, crc_tab;    // the members' CRC-32 as their trailers say; gd_crc_kernel's tables
    std::vector<uint32_t> h_crc;                   // (host copy of crc_want: the members the older kernel does again are checked again)
    vgan::gd::GBuf<vgan::gd::GdMapRec> map_rec; // per mapping: where its bytes lie, where its edits go (gd_fill_kernel -> gd_fill_maps_kernel)
    // one DfSlice's arrays (hc_flatten_kernels.hip) of the file's reads
    vgan::gd::GBuf<uint32_t> map_off, qual_off, edit_off, e_seq_off, m_node, seq_len;
    vgan::gd::GBuf<int32_t> m_offset, mapq, e_len;
    vgan::gd::GBuf<uint8_t> unmapped, m_rev, e_seq, qual;
    vgan::gd::GBuf<int64_t> first_node, first_offset;
    vgan::gd::GBuf<uint8_t> dup, picked_bytes;           // duplicate marks per read; the messages handed back to the host
    vgan::gd::GBuf<uint64_t> sort_key, sort_key2, picked_off;
    vgan::gd::GBuf<uint32_t> perm_a, perm_b, new_flag, new_at;
    vgan::gd::GBuf<int64_t> new_node, new_off;
    hipStream_t piece_stream[GD_PIECES] = {}; // the file's (or a piece's) bytes go up in a few parts, each copied and inflated on a stream of its own
    // the stream the framing and the parse read: `u` = infl.p + tail_cap - (bytes carried over from the piece before), n_stream bytes
    uint64_t tail_cap = 0;
    const uint8_t *u = nullptr;
    uint64_t n_stream = 0;
    size_t n_blocks = 0;
    uint64_t n_redone = 0;     // (test aid) members the two-kernel inflate left to the older kernel
    bool inflate_checked = false; // gd_piece_inflated has waited for this piece's inflate and looked at its members' states
    uint64_t n_reanchored = 0; // (test aid) tag-like bytes the framing of the parses so far took for a group's tag and gave up again
    uint64_t n_picked = 0, n_picked_bytes = 0;
    uint64_t n_inflated = 0, n_messages = 0, R = 0, M = 0, E = 0, S = 0, Q = 0;
    double ms_inflate = 0, ms_frame = 0, ms_parse = 0, ms_upload = 0;
    size_t device_bytes() const { // what the object holds of the device's memory
        size_t b = in.cap + infl.cap + cub_tmp.cap + blocks.cap * sizeof(vgan::GdBlock) + carry.cap * sizeof(vgan::gd::GdCarryState) + map_rec.cap * sizeof(vgan::gd::GdMapRec);
        for (auto *x : {&status, &seg_msgs, &seg_status, &msg_len, &keep, &n_map, &n_edit, &n_eseq, &n_qual, &r_at, &m_at, &e_at, &s_at, &q_at, &bad, &map_off, &qual_off,
                        &edit_off, &e_seq_off, &m_node, &seq_len, &perm_a, &perm_b, &new_flag, &new_at})
            b += x->cap * 4;
        for (auto *x : {&anchor, &next_anchor, &msg_base, &msg_off, &sort_key, &sort_key2, &picked_off, &tok_reg}) b += x->cap * 8;
        b += (tok_nreg.cap + tok_cursor.cap + crc_want.cap + crc_tab.cap) * 4;
        for (auto *x : {&m_offset, &mapq, &e_len}) b += x->cap * 4;
        for (auto *x : {&unmapped, &m_rev, &e_seq, &qual, &dup, &picked_bytes}) b += x->cap;
        for (auto *x : {&first_node, &first_offset, &new_node, &new_off}) b += x->cap * 8;
        return b;
    }
    void release_all() {
        in.release(), infl.release(), cub_tmp.release(), blocks.release();
        for (auto *b : {&status, &seg_msgs, &seg_status, &msg_len, &keep, &n_map, &n_edit, &n_eseq, &n_qual, &r_at, &m_at, &e_at, &s_at, &q_at, &bad, &map_off,
                        &qual_off, &edit_off, &e_seq_off, &m_node, &seq_len})
            b->release();
        for (auto *b : {&anchor, &next_anchor, &msg_base, &msg_off}) b->release();
        carry.release();
        tok_reg.release(), tok_nreg.release(), tok_cursor.release(), crc_want.release(), crc_tab.release();
        map_rec.release();
        for (auto *b : {&m_offset, &mapq, &e_len}) b->release();
        for (auto *b : {&unmapped, &m_rev, &e_seq, &qual}) b->release();
        first_node.release(), first_offset.release();
        dup.release(), picked_bytes.release(), sort_key.release(), sort_key2.release(), picked_off.release();
        for (auto *b : {&perm_a, &perm_b, &new_flag, &new_at}) b->release();
        new_node.release(), new_off.release();
    }
};

namespace vgan {
namespace gd {
// ---- a file in pieces (gam_kernels.hip): what vgan_gamdev_parse does in one go, cut at BGZF member boundaries
// The piece's bytes go up and its members are inflated behind them (asynchronous: gd_piece_parse waits).  blocks: the piece's members,
// in_off relative to `bytes`, out_off relative to the piece's first inflated byte; tail_cap: room kept in front of the inflated bytes
// for what the piece before leaves over; crcs: the members' CRC-32 as their trailers say (checked on the device: GD_BAD_CRC).
int gd_piece_upload_inflate(vgan_gamdev *g, const uint8_t *bytes, uint64_t n_bytes, const GdBlock *blocks, const uint32_t *crcs, size_t n_blocks, uint64_t total_out,
                            uint64_t tail_cap);
// Framing from the state `in` (its tail in front of the inflated bytes) + the protobuf walk: the piece's arrays as vgan_gamdev_parse
// leaves a file's.  `last`: the stream ends with this piece (a walk that ends inside an item is then a truncated file); otherwise `out`
// takes the state and the bytes left over.  frame_done (or null) is called once `out` is final: the next piece's framing may start then.
// Waits for the piece's inflate and checks its members (a member the two kernels left is done again by the older one; a wrong CRC-32 is an
// error).  Needs nothing of the piece before: a caller makes it BEFORE it takes its turn at the framing's hand-over (gd_piece_parse calls
// it if the caller did not).
int gd_piece_inflated(vgan_gamdev *g);
int gd_piece_parse(vgan_gamdev *g, const GdCarry &in, bool last, GdCarry *out, int keep_unmapped, void (*frame_done)(void *), void *user);
// Duplicate marks of the piece's reads against the reads of the piece itself and the keys of every piece before it (`seen`, updated);
// added (or null): the keys the piece adds, for the other lanes' sets.
int gd_piece_mark_duplicates(vgan_gamdev *g, GdSeen &seen, GdKeyList *added, int64_t *n_dup);
// the keys another lane's piece added go into this lane's set
int gd_seen_merge(vgan_gamdev *g, GdSeen &seen, const GdKeyList &keys);

// ---- the pipeline over a file's pieces (gam_pipe.hip)
// What a subcommand does with a parsed piece (its arrays on the device: gamdev_slice): called on the piece's slot thread, pieces of
// different slots at the same time -- the consumer serialises what it must.  d_dup: the piece's duplicate marks (device, per read), or null.
struct GamConsumer {
    virtual ~GamConsumer() {}
    virtual int consume(int lane, vgan_gamdev *g, uint64_t read_base, const uint8_t *d_dup, int64_t piece) = 0;
    virtual void aborted() {} // the pipeline failed elsewhere: a consume that waits must give up
};
int gampipe_run(const void *bytes, uint64_t n, const std::vector<int> &lane_devices, const vgan_gampipe_opts &opts, GamConsumer &consumer,
                vgan_gampipe_stats *stats);
// the options a run takes when the caller leaves them open (piece size from the input's size, slots, tail room; VGAN_GAMPIPE_* environment)
vgan_gampipe_opts gampipe_defaults(const vgan_gampipe_opts *o, uint64_t n_bytes, int n_lanes);
} // namespace gd
} // namespace vgan
struct vgan_hc_devflat;
namespace vgan {
size_t hc_devflat_device_bytes(const vgan_hc_devflat *f); // hc_flatten_kernels.hip
}
