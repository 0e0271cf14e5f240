// The GAM front end on the device as a pipeline over the file's pieces (reference: src/readGAM.h:20-68, one serial loop over
// libvgio's stream that feeds HaploCart.cpp:383-421, readGAM_Euka.h:581 and getLCAfromGAM.h:31-45).
//
//   file bytes --cut at BGZF member boundaries--> pieces of <= piece_bytes
//   piece i -> lane i mod N (a lane = one device context), slot (i / N) mod S of that lane (a slot = one vgan_gamdev: buffers, a stream,
//   a host thread):  upload + inflate | framing <- the state piece i-1 left | protobuf walk | duplicate marks <- the keys so far | consume
//
// Only the framing's hand-over, the index of a piece's first read and the duplicate keys are serial (a few hundred bytes and a
// few milliseconds per piece); everything else of a piece runs beside the other slots' pieces.  Device memory is S sets of one piece's
// buffers per lane, whatever the file's size.  No kernel lives here: the kernels and the per-piece functions are gam_kernels.hip's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "euka_device.h"
#include "gam_device.h"
#include "gam_object.h"
#include "hc_device.h"
#include "host/common.h"
#include "vgan_gpu.h"

using namespace vgan;
using namespace vgan::gd;

namespace {

double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

// piece_bytes = 0: a piece is a twenty-fourth of a lane's share of the file, within 32..192 MB -- a short file still gives every slot
// several pieces to overlap (a 1.5 GB file in pieces of 192 MB was three rounds of three slots: 1.1 s against 0.9 in pieces of 64 MB),
// a long one is not cut finer than the per-piece costs (a dozen stream synchronisations, the framing's hand-over) are worth
vgan_gampipe_opts with_defaults(const vgan_gampipe_opts *o, uint64_t n_bytes = 0, int n_lanes = 1) {
    vgan_gampipe_opts r{};
    if (o) r = *o;
    if (const char *e = getenv("VGAN_GAMPIPE_PIECE")) r.piece_bytes = strtoull(e, nullptr, 10);
    if (const char *e = getenv("VGAN_GAMPIPE_SLOTS")) r.slots = atoi(e);
    if (r.slots <= 0) r.slots = 3;
    if (r.piece_bytes == 0)
        r.piece_bytes = n_bytes ? std::min<uint64_t>(192ull << 20, std::max<uint64_t>(32ull << 20, n_bytes / (uint64_t)(8 * r.slots * std::max(1, n_lanes)))) : 192ull << 20;
    if (r.slots > 16) r.slots = 16;
    if (r.tail_bytes == 0) r.tail_bytes = 8ull << 20;
    return r;
}

// ---------------------------------------------------------------------------------------------------------------- the pieces
struct PiecePlan {
    uint64_t in0 = 0, in1 = 0;       // the piece's bytes in the file: its first member's header .. its last member's trailer
    const GdBlock *blocks = nullptr; // its members: in_off relative to in0, out_off relative to the piece's first inflated byte
    const uint32_t *crcs = nullptr;  // ... and the CRC-32 their trailers state
    size_t n_blocks = 0;
    uint64_t out_bytes = 0;
};
// Walks the BGZF member headers as pieces are asked for (SAM spec 4.1; the same checks as host/util.cpp: bgzf_index): the first piece
// is known after a few thousand headers, not after the file's two hundred thousand.
struct Cutter {
    const uint8_t *p;
    uint64_t n, piece_bytes, max_out;
    uint64_t off = 0;
    bool bad = false, at_end = false;
    std::deque<std::vector<GdBlock>> blocks;
    std::deque<std::vector<uint32_t>> crcs;
    std::deque<PiecePlan> pieces;
    std::mutex mu;
    Cutter(const void *bytes, uint64_t n_, uint64_t piece, uint64_t max_out_) : p((const uint8_t *)bytes), n(n_), piece_bytes(piece), max_out(max_out_) {}

    bool member_at(uint64_t o, uint64_t &bsize, uint64_t &hdr, uint64_t &isize, uint32_t &crc) const {
        if (n - o < 18 || p[o] != 0x1f || p[o + 1] != 0x8b || p[o + 2] != 8 || !(p[o + 3] & 4)) return false;
        const uint64_t xlen = p[o + 10] | (p[o + 11] << 8);
        if (n - o < 12 + xlen) return false;
        bsize = 0;
        for (uint64_t x = o + 12; x + 4 <= o + 12 + xlen;) {
            const uint64_t slen = p[x + 2] | (p[x + 3] << 8);
            if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= o + 12 + xlen) bsize = (uint64_t)(p[x + 4] | (p[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || bsize > n - o) return false;
        const uint8_t *t = p + o + bsize - 4;
        isize = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
        crc = (uint32_t)t[-4] | ((uint32_t)t[-3] << 8) | ((uint32_t)t[-2] << 16) | ((uint32_t)t[-1] << 24);
        hdr = 12 + xlen;
        return true;
    }
    void cut_one() {
        PiecePlan pc;
        pc.in0 = off;
        std::vector<GdBlock> gb;
        std::vector<uint32_t> cr;
        uint64_t out = 0, end = off;
        while (off < n) {
            uint64_t bsize, hdr, isize;
            uint32_t crc;
            if (!member_at(off, bsize, hdr, isize, crc)) {
                bad = true;
                break;
            }
            if (isize == 0) { // (the end-of-file member, empty members: nothing to inflate; in front of a piece they are not sent up either)
                off += bsize;
                if (gb.empty()) pc.in0 = off;
                continue;
            }
            if (!gb.empty() && (off + bsize - pc.in0 > piece_bytes || out + isize > max_out)) break;
            gb.push_back(GdBlock{off + hdr - pc.in0, out, (uint32_t)(bsize - hdr - 8), (uint32_t)isize});
            cr.push_back(crc);
            out += isize;
            off += bsize;
            end = off;
        }
        if (off >= n || bad) at_end = true;
        if (gb.empty()) return;
        blocks.push_back(std::move(gb));
        crcs.push_back(std::move(cr));
        pc.in1 = end;
        pc.blocks = blocks.back().data();
        pc.crcs = crcs.back().data();
        pc.n_blocks = blocks.back().size();
        pc.out_bytes = out;
        pieces.push_back(pc);
    }
    // piece i, or false: the file has fewer (is_bad() then says whether the walk met bytes that are no BGZF member before the file's end)
    bool get(size_t i, PiecePlan &out) {
        std::lock_guard<std::mutex> lk(mu);
        while (pieces.size() <= i && !at_end) cut_one();
        if (pieces.size() <= i) return false;
        out = pieces[i];
        return true;
    }
    bool is_bad() {
        std::lock_guard<std::mutex> lk(mu);
        return bad;
    }
};

// inflated / compressed bytes of the file's first members (what a piece of c compressed bytes will take on the device)
double first_ratio(const void *bytes, uint64_t n) {
    Cutter c(bytes, n, 4ull << 20, ~0ull);
    PiecePlan pc;
    if (!c.get(0, pc) || pc.in1 <= pc.in0) return 3.0;
    return std::max(1.0, (double)pc.out_bytes / (double)(pc.in1 - pc.in0));
}

// ---------------------------------------------------------------------------------------------------------------- the pipeline
struct Lane {
    int device = 0;
    std::vector<vgan_gamdev *> slots;
    GdSeen seen;
};
struct Pipe {
    Cutter cut;
    vgan_gampipe_opts o;
    int N, S;
    std::vector<Lane> lanes;
    GamConsumer &consumer;
    std::mutex mu;
    std::condition_variable cv;
    size_t framed = 0, counted = 0, deduped = 0; // pieces [0, x) have passed the turn
    GdCarry carry;                                // what piece `framed - 1` left: the state in front of piece `framed`
    uint64_t next_base = 0;                       // index of piece `counted`'s first read
    std::deque<GdKeyList> keys;                   // (touched inside the duplicate turn only)
    int err_code = 0;
    std::string err;
    vgan_gampipe_stats st{};
    Pipe(const void *bytes, uint64_t n, const vgan_gampipe_opts &opts, int n_lanes, GamConsumer &c)
        : cut(bytes, n, opts.piece_bytes, 0xE0000000ull - opts.tail_bytes), o(opts), N(n_lanes), S(opts.slots), lanes((size_t)n_lanes), consumer(c) {}

    void fail_with(int code, const std::string &why) {
        bool first = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (err_code == 0) {
                err_code = code;
                err = why;
                first = true;
            }
        }
        cv.notify_all();
        if (first) consumer.aborted();
    }
    bool failed() {
        std::lock_guard<std::mutex> lk(mu);
        return err_code != 0;
    }
    // waits until `turn` reaches i; false when the pipeline has failed meanwhile
    bool wait_turn(const size_t &turn, size_t i) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return turn == i || err_code != 0; });
        return err_code == 0;
    }
    void pass_turn(size_t &turn) {
        {
            std::lock_guard<std::mutex> lk(mu);
            turn += 1;
        }
        cv.notify_all();
    }

    std::chrono::steady_clock::time_point t_start = std::chrono::steady_clock::now();
    bool timing = getenv("VGAN_TIMING") != nullptr;

    void slot_thread(int lane, int slot) {
        Lane &L = lanes[(size_t)lane];
        vgan_gamdev *g = nullptr;
        if (vgan_gamdev_create(L.device, nullptr, &g) < 0) {
            fail_with(VGAN_ENODEV, last_error());
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            L.slots[(size_t)slot] = g;
        }
        for (size_t j = (size_t)slot;; j += (size_t)S) {
            const size_t i = j * (size_t)N + (size_t)lane;
            PiecePlan pc, nxt;
            if (failed() || !cut.get(i, pc)) break;
            const bool last = !cut.get(i + 1, nxt);
            if (last && cut.is_bad()) {
                fail_with(VGAN_EIO, "the GAM front end on the device: the file is not a BGZF stream (or ends inside a member)");
                break;
            }
            int rc;
            double ts[8] = {ms_since(t_start)}; // (VGAN_TIMING: when the piece's stages ended, from the pipeline's start)
            if ((rc = gd_piece_upload_inflate(g, cut.p + pc.in0, pc.in1 - pc.in0, pc.blocks, pc.crcs, pc.n_blocks, pc.out_bytes, o.tail_bytes)) < 0) {
                fail_with(rc, last_error());
                break;
            }
            ts[1] = ms_since(t_start);
            // (the piece's own inflate is waited for BEFORE its turn: taken first, the turn held the pieces behind it for as long as this
            // one's inflate still had to run -- 9 ms a piece where the hand-over itself takes 3)
            if ((rc = gd_piece_inflated(g)) < 0) {
                fail_with(rc, last_error());
                break;
            }
            // ---- the framing turn: the state the piece before left, and what this one leaves
            if (!wait_turn(framed, i)) break;
            ts[2] = ms_since(t_start);
            GdCarry cin, cout;
            {
                std::lock_guard<std::mutex> lk(mu);
                cin = std::move(carry);
                carry = GdCarry{};
            }
            if (cin.st.mode > 2u) { // (the piece before failed inside its framing: its thread says why)
                fail_with(VGAN_EIO, "the GAM front end on the device: the piece before left no framing state");
                break;
            }
            struct Turn {
                Pipe *p;
                GdCarry *out;
                static void done(void *u) {
                    auto *t = static_cast<Turn *>(u);
                    {
                        std::lock_guard<std::mutex> lk(t->p->mu);
                        t->p->carry = std::move(*t->out);
                        t->p->framed += 1;
                    }
                    t->p->cv.notify_all();
                }
            } turn{this, &cout};
            rc = gd_piece_parse(g, cin, last, &cout, o.keep_unmapped, &Turn::done, &turn);
            if (rc < 0) {
                fail_with(rc, last_error());
                break;
            }
            ts[3] = ms_since(t_start);
            // ---- the index of the piece's first read
            if (!wait_turn(counted, i)) break;
            uint64_t base;
            {
                std::lock_guard<std::mutex> lk(mu);
                base = next_base;
                next_base += g->R;
                counted += 1;
                st.n_pieces += 1;
                st.compressed_bytes += pc.in1 - pc.in0;
                st.inflated_bytes += pc.out_bytes;
                st.n_messages += g->n_messages;
                st.n_reads += g->R;
                st.ms_upload += g->ms_upload;
                st.ms_inflate += g->ms_inflate;
                st.ms_frame += g->ms_frame;
                st.ms_parse += g->ms_parse;
            }
            cv.notify_all();
            // ---- duplicate marks: against the piece's own reads and the keys of every piece before it
            const uint8_t *d_dup = nullptr;
            if (o.mark_duplicates) {
                if (!wait_turn(deduped, i)) break;
                const auto t0 = std::chrono::steady_clock::now();
                rc = VGAN_OK;
                for (size_t k = L.seen.merged; k < keys.size() && rc == VGAN_OK; ++k)
                    if (keys[k].lane != lane) rc = gd_seen_merge(g, L.seen, keys[k]);
                L.seen.merged = keys.size();
                int64_t nd = 0;
                keys.emplace_back();
                keys.back().lane = lane;
                if (rc == VGAN_OK) rc = gd_piece_mark_duplicates(g, L.seen, N > 1 ? &keys.back() : nullptr, &nd);
                L.seen.merged = keys.size();
                if (rc < 0) {
                    fail_with(rc, last_error());
                    break;
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    st.n_duplicates += (uint64_t)nd;
                    st.ms_dedup += ms_since(t0);
                    deduped += 1;
                }
                cv.notify_all();
                d_dup = g->R ? vgan_gamdev_dup_marks(g) : nullptr;
            }
            ts[4] = ms_since(t_start);
            // ---- the subcommand's share
            const auto t0 = std::chrono::steady_clock::now();
            if (g->R && (rc = consumer.consume(lane, g, base, d_dup, (int64_t)i)) < 0) {
                fail_with(rc, last_error());
                break;
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                st.ms_consume += ms_since(t0);
            }
            if (timing)
                fprintf(stderr, "[vgan timing] gampipe piece %zu (lane %d slot %d, %llu reads): from %.0f ms: sent %.0f, its framing turn at %.0f, parsed %.0f (inflate wait %.0f, framing %.0f, walk %.0f), marked %.0f, consumed %.0f\n",
                        i, lane, slot, (unsigned long long)g->R, ts[0], ts[1], ts[2], ts[3], g->ms_inflate, g->ms_frame, g->ms_parse, ts[4], ms_since(t_start));
        }
    }
};

} // namespace

vgan_gampipe_opts vgan::gd::gampipe_defaults(const vgan_gampipe_opts *o, uint64_t n_bytes, int n_lanes) { return with_defaults(o, n_bytes, n_lanes); }

int vgan::gd::gampipe_run(const void *bytes, uint64_t n, const std::vector<int> &lane_devices, const vgan_gampipe_opts &opts_in, GamConsumer &consumer,
                          vgan_gampipe_stats *stats) {
    if (stats) memset(stats, 0, sizeof *stats);
    if ((!bytes && n) || lane_devices.empty()) return fail(VGAN_EINVAL, "gampipe_run: null argument");
    const int N = (int)lane_devices.size();
    vgan_gampipe_opts o = with_defaults(&opts_in, n, N);
    const auto t0 = std::chrono::steady_clock::now();
    // ---- what the lanes will hold of their devices' memory against what is free there: smaller pieces, fewer slots, or not at all
    // (S sets of {a piece's bytes, their inflated form, the parse's arrays ~1.1 x that} + one flattened piece ~1.3 x per lane)
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(VGAN_ENODEV, "gampipe_run: no HIP device is visible (this library has no CPU path)");
        for (int d : lane_devices)
            if (d < 0 || d >= ndev) return fail(VGAN_EINVAL, "gampipe_run: device %d out of range", d);
        const double ratio = first_ratio(bytes, n);
        for (;;) {
            bool fits = true;
            const uint64_t c = std::min<uint64_t>(o.piece_bytes, n);
            for (int d = 0; d < ndev && fits; ++d) {
                int lanes_here = 0;
                for (int x : lane_devices) lanes_here += x == d;
                if (!lanes_here) continue;
                size_t fr = 0, tot = 0;
                if (hipSetDevice(d) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) return fail(VGAN_ENODEV, "gampipe_run: hipMemGetInfo failed on device %d", d);
                const double f = (double)c * ratio;
                const double need = lanes_here * (o.slots * ((double)c + (double)o.tail_bytes + 2.2 * f) + 1.4 * f) + (256 << 20);
                if (need > 0.92 * (double)fr) fits = false;
            }
            if (fits) break;
            if (o.piece_bytes > (32ull << 20)) o.piece_bytes /= 2;
            else if (o.slots > 1) o.slots -= 1;
            else return fail(VGAN_ENOMEM, "gampipe_run: the device has no room for the front end's buffers (a piece of %llu bytes in one slot)", (unsigned long long)o.piece_bytes);
        }
    }
    Pipe pipe(bytes, n, o, N, consumer);
    for (int l = 0; l < N; ++l) {
        pipe.lanes[(size_t)l].device = lane_devices[(size_t)l];
        pipe.lanes[(size_t)l].slots.assign((size_t)o.slots, nullptr);
    }
    {
        PiecePlan first;
        if (!pipe.cut.get(0, first) && pipe.cut.is_bad()) return fail(VGAN_EIO, "gampipe_run: not a BGZF stream");
    }
    std::vector<std::thread> th;
    for (int l = 0; l < N; ++l)
        for (int s = 0; s < o.slots; ++s) th.emplace_back([&pipe, l, s] { pipe.slot_thread(l, s); });
    for (auto &t : th) t.join();
    for (Lane &L : pipe.lanes) {
        size_t held = 0;
        for (vgan_gamdev *g : L.slots)
            if (g) {
                held += g->device_bytes();
                pipe.st.n_reanchored += g->n_reanchored;
            }
        held += (L.seen.node[0].cap + L.seen.node[1].cap + L.seen.off[0].cap + L.seen.off[1].cap) * 8;
        pipe.st.device_bytes += held;
        (void)hipSetDevice(L.device);
        L.seen.release();
        for (vgan_gamdev *g : L.slots) vgan_gamdev_free(g);
    }
    pipe.st.ms_wall = ms_since(t0);
    if (stats) *stats = pipe.st;
    if (pipe.err_code) return fail(pipe.err_code, "%s", pipe.err.c_str());
    return VGAN_OK;
}

// ------------------------------------------------------------------------------------------------------------ test aids
extern "C" int64_t vgan_gampipe_plan(const void *bytes, uint64_t n, const vgan_gampipe_opts *opts, uint64_t *piece_in_off, uint64_t *piece_in_bytes,
                                     uint64_t *piece_out_bytes, int64_t cap) {
    if (!bytes && n) return fail(VGAN_EINVAL, "vgan_gampipe_plan: null argument");
    const vgan_gampipe_opts o = with_defaults(opts, n, 1);
    Cutter c(bytes, n, o.piece_bytes, 0xE0000000ull - o.tail_bytes);
    int64_t k = 0;
    PiecePlan pc;
    for (; c.get((size_t)k, pc); ++k)
        if (k < cap) {
            if (piece_in_off) piece_in_off[k] = pc.in0;
            if (piece_in_bytes) piece_in_bytes[k] = pc.in1 - pc.in0;
            if (piece_out_bytes) piece_out_bytes[k] = pc.out_bytes;
        }
    if (c.is_bad()) return fail(VGAN_EIO, "vgan_gampipe_plan: not a BGZF stream");
    return k;
}

struct vgan_gampipe_carry {
    GdCarry c;
};
extern "C" void vgan_gampipe_carry_free(vgan_gampipe_carry *c) { delete c; }
extern "C" int vgan_gampipe_parse_piece(vgan_gamdev *g, const void *bytes, uint64_t n, const vgan_gampipe_opts *opts, int64_t piece, vgan_gampipe_carry **carry) {
    if (!g || (!bytes && n) || !carry || piece < 0) return fail(VGAN_EINVAL, "vgan_gampipe_parse_piece: null argument");
    const vgan_gampipe_opts o = with_defaults(opts, n, 1);
    Cutter c(bytes, n, o.piece_bytes, 0xE0000000ull - o.tail_bytes);
    PiecePlan pc, nxt;
    if (!c.get((size_t)piece, pc)) return fail(c.is_bad() ? VGAN_EIO : VGAN_EINVAL, "vgan_gampipe_parse_piece: the file has no piece %lld", (long long)piece);
    const bool last = !c.get((size_t)piece + 1, nxt);
    if (last && c.is_bad()) return fail(VGAN_EIO, "vgan_gampipe_parse_piece: not a BGZF stream");
    int rc;
    if ((rc = gd_piece_upload_inflate(g, c.p + pc.in0, pc.in1 - pc.in0, pc.blocks, pc.crcs, pc.n_blocks, pc.out_bytes, o.tail_bytes)) < 0) return rc;
    if (!*carry) *carry = new vgan_gampipe_carry();
    GdCarry out;
    if ((rc = gd_piece_parse(g, (*carry)->c, last, &out, o.keep_unmapped, nullptr, nullptr)) < 0) return rc;
    (*carry)->c = std::move(out);
    return VGAN_OK;
}

// ------------------------------------------------------------------------------------------------------------ HaploCart
// vgan haplocart's consumer: device flatten of the piece into the packed batch + the segment kernel, on the lane's context; the reads the
// device flatten leaves (indels, soft clips, long reads) come back as their messages and go through the host's parser and flatten,
// beside the device's write pass.
namespace {
struct LeftJob { // the reads of one piece that the device flatten leaves to the host
    std::thread t;
    std::vector<uint8_t> mask;
    uint64_t n = 0;
    int rc = VGAN_OK;
    std::string err;
    std::mutex m;
    std::condition_variable cv;
    bool dev_done = false; // the messages are down: the piece's device object is free
};
} // namespace
struct vgan_hc_gamrun : GamConsumer {
    std::vector<std::shared_ptr<LeftJob>> bg;
    std::vector<int> devices;
    const void *bytes = nullptr;
    uint64_t n = 0;
    vgan_gampipe_opts opts{};
    std::thread coord;
    int rc = VGAN_OK;
    std::string err;
    vgan_gampipe_stats pst{};
    // what attach brings
    std::mutex mu;
    std::condition_variable cv;
    bool attached = false, gave_up = false;
    std::vector<vgan_hc_ctx *> ctx;
    const vgan_graph *graph = nullptr;
    std::vector<vgan_hc_devflat *> df;
    std::deque<std::mutex> lane_mu;
    vgan_hc_flatten_stats fst{};
    uint64_t n_host_reads = 0, n_device_reads = 0;
    size_t df_bytes = 0;
    double ms_wait_contexts = 0;
    int host_threads = 2;

    void aborted() override {
        {
            std::lock_guard<std::mutex> lk(mu);
            gave_up = true;
        }
        cv.notify_all();
    }
    int hand_over(vgan_hc_ctx *cx, vgan_hc_host_batch *hb) {
        vgan_hc_batch b;        // the reads outside the tile contract (long reads, ...): the general kernel
        vgan_hc_packed_view pk; // everything else, in the segment kernel's own layout as the flatten step wrote it
        int r;
        if ((r = vgan_hc_host_batch_get(hb, &b)) < 0 || (r = vgan_hc_host_batch_get_packed(hb, &pk)) < 0) return r;
        if ((r = vgan_hc_accumulate_packed(cx, &pk)) < 0 || (b.n_reads && (r = vgan_hc_accumulate(cx, &b)) < 0)) return r;
        return VGAN_OK;
    }
    int consume(int lane, vgan_gamdev *g, uint64_t read_base, const uint8_t *d_dup, int64_t) override {
        {
            const auto t0 = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return attached || gave_up; });
            if (!attached) return fail(VGAN_ESTATE, "vgan_hc_gam: no contexts were attached");
            ms_wait_contexts = std::max(ms_wait_contexts, ms_since(t0));
        }
        vgan_hc_ctx *cx = ctx[(size_t)lane];
        uint64_t sz[8];
        (void)vgan_gamdev_sizes(g, sz, nullptr);
        const uint64_t R = sz[2];
        if (read_base + R > 0xFFFFFFF0ull) return fail(VGAN_ERANGE, "vgan_hc_gam: more than 2^32 reads");
        // the reads the device flatten leaves: their messages come down (on a thread of its own from the moment the mask is known, beside
        // the device's offsets and write pass) -- the piece's slot waits for that much, the object's buffers are free then -- and the
        // host parses, flattens and hands them over behind the slot's back, while it already takes its next piece
        auto job = std::make_shared<LeftJob>();
        job->mask.assign((size_t)R, 0);
        vgan_hc_packed_view pk;
        vgan_hc_flatten_stats st{};
        vgan_hc_gamrun *self = this;
        std::function<void()> host_left = [self, job, g, cx, lane] {
            auto device_done = [&] {
                {
                    std::lock_guard<std::mutex> lk(job->m);
                    job->dev_done = true;
                }
                job->cv.notify_all();
            };
            for (uint8_t m : job->mask) job->n += m;
            uint64_t nm = 0, nb = 0;
            std::vector<uint64_t> offs;
            std::vector<uint8_t> msgs;
            if (job->n) {
                if ((job->rc = vgan_gamdev_pick(g, job->mask.data(), &nm, &nb)) >= 0) {
                    offs.resize((size_t)nm + 1);
                    msgs.resize((size_t)std::max<uint64_t>(nb, 1));
                    job->rc = vgan_gamdev_picked(g, offs.data(), msgs.data());
                }
                if (job->rc < 0) job->err = last_error();
            }
            device_done();
            if (job->n == 0 || job->rc < 0) return;
            vgan_alnparts *parts = nullptr;
            vgan_hc_host_batch *hb = nullptr;
            vgan_hc_flatten_stats sh{};
            if ((job->rc = vgan_alnparts_from_messages(msgs.data(), offs.data(), (int64_t)nm, self->opts.keep_unmapped, self->host_threads, &parts)) >= 0) {
                job->rc = vgan_hc_flatten_parts_packed(self->graph, parts, 0, vgan_alnparts_count(parts), nullptr, self->host_threads, &hb, &sh);
                vgan_alnparts_free(parts);
            }
            if (job->rc >= 0 && hb) {
                std::lock_guard<std::mutex> lk(self->lane_mu[(size_t)lane]);
                job->rc = self->hand_over(cx, hb); // (the batch's host arrays are copied inside these calls: it may go when they return)
            }
            if (job->rc < 0) job->err = last_error();
            vgan_hc_host_batch_free(hb);
            std::lock_guard<std::mutex> lk(self->mu);
            self->fst.n_bad += sh.n_bad;
            self->fst.n_unmapped += sh.n_unmapped;
            self->fst.n_clamped += sh.n_clamped;
            self->fst.n_out += sh.n_out;
            self->fst.n_segments += sh.n_segments;
            self->fst.n_cols += sh.n_cols;
        };
        struct Hook {
            LeftJob *job;
            std::function<void()> *fn;
            static void go(void *u) {
                auto *h = static_cast<Hook *>(u);
                h->job->t = std::thread(*h->fn);
            }
        } hook{job.get(), &host_left};
        {
            std::lock_guard<std::mutex> lk(mu); // (from here on the run joins the thread, whatever way this call ends)
            bg.push_back(job);
        }
        int r;
        const auto tc0 = std::chrono::steady_clock::now();
        double t_lock = 0, t_flat = 0;
        {
            std::lock_guard<std::mutex> lk(lane_mu[(size_t)lane]); // (the lane's context, its stream and its flatten object: one piece at a time)
            t_lock = ms_since(tc0);
            if (!df[(size_t)lane] && (r = vgan_hc_devflat_create(cx, graph, &df[(size_t)lane])) < 0) return r;
            r = vgan_hc_devflat_run_gamdev_cb(df[(size_t)lane], g, d_dup, d_dup ? 1 : 0, (uint32_t)read_base, &pk, job->mask.data(), &st, &Hook::go, &hook);
            if (r >= 0) r = vgan_hc_accumulate_packed(cx, &pk);
            t_flat = ms_since(tc0);
        }
        if (job->t.joinable()) { // (started when the mask was final: the object's buffers are the thread's until it says so)
            std::unique_lock<std::mutex> lk(job->m);
            job->cv.wait(lk, [&] { return job->dev_done; });
        }
        if (r < 0) return r;
        if (job->rc < 0) return fail(job->rc, "the reads left to the host: %s", job->err.c_str());
        if (getenv("VGAN_TIMING"))
            fprintf(stderr, "[vgan timing] hc consume: waited %.1f ms for the lane, flattened + handed over at %.1f, the messages of the %llu reads left to the host down at %.1f\n", t_lock,
                    t_flat, (unsigned long long)job->n, ms_since(tc0));
        std::lock_guard<std::mutex> lk(mu);
        fst.n_in += st.n_in;
        fst.n_unmapped += st.n_unmapped;
        fst.n_clamped += st.n_clamped;
        fst.n_out += st.n_out;
        fst.n_segments += st.n_segments;
        fst.n_cols += st.n_cols;
        n_host_reads += job->n;
        n_device_reads += (uint64_t)st.n_out;
        return VGAN_OK;
    }
};

extern "C" int vgan_hc_gam_start(const int *devices, int n_lanes, const void *bytes, uint64_t n, const vgan_gampipe_opts *opts, vgan_hc_gamrun **out) {
    if (!devices || n_lanes <= 0 || (!bytes && n) || !out) return fail(VGAN_EINVAL, "vgan_hc_gam_start: null argument");
    auto *r = new vgan_hc_gamrun();
    r->devices.assign(devices, devices + n_lanes);
    r->bytes = bytes;
    r->n = n;
    r->opts = with_defaults(opts, n, n_lanes);
    r->df.assign((size_t)n_lanes, nullptr);
    r->lane_mu.resize((size_t)n_lanes);
    const int cpus = r->opts.n_threads > 0 ? r->opts.n_threads : (int)usable_cpus();
    r->host_threads = std::max(1, std::min(8, cpus / std::max(1, n_lanes * r->opts.slots)));
    r->coord = std::thread([r] {
        r->rc = gampipe_run(r->bytes, r->n, r->devices, r->opts, *r, &r->pst);
        if (r->rc < 0) r->err = last_error();
    });
    *out = r;
    return VGAN_OK;
}

extern "C" int vgan_hc_gam_attach(vgan_hc_gamrun *r, vgan_hc_ctx *const *ctxs, int n_ctx, const vgan_graph *graph) {
    if (!r || !ctxs || !graph) return fail(VGAN_EINVAL, "vgan_hc_gam_attach: null argument");
    if (n_ctx != (int)r->devices.size()) return fail(VGAN_EINVAL, "vgan_hc_gam_attach: %d contexts for %zu lanes", n_ctx, r->devices.size());
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if (r->attached) return fail(VGAN_ESTATE, "vgan_hc_gam_attach: called twice");
        r->ctx.assign(ctxs, ctxs + n_ctx);
        r->graph = graph;
        r->attached = true;
    }
    r->cv.notify_all();
    return VGAN_OK;
}

extern "C" int vgan_hc_gam_finish(vgan_hc_gamrun *r, vgan_hc_flatten_stats *stats, vgan_gampipe_stats *pstats) {
    if (!r) return fail(VGAN_EINVAL, "vgan_hc_gam_finish: null argument");
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if (!r->attached) r->gave_up = true; // (nobody will bring contexts any more: the pieces that wait for them give up)
    }
    r->cv.notify_all();
    if (r->coord.joinable()) r->coord.join();
    for (auto &j : r->bg) { // the host's share of the pieces (parse, flatten, hand-over of the reads the device left)
        if (j->t.joinable()) j->t.join();
        if (j->rc < 0 && r->rc >= 0) {
            r->rc = j->rc;
            r->err = "the reads left to the host: " + j->err;
        }
    }
    for (size_t l = 0; l < r->df.size(); ++l) {
        if (r->df[l] && l < r->ctx.size()) (void)vgan_hc_synchronize(r->ctx[l]); // (the segment kernel reads the flatten object's buffers)
        r->df_bytes += hc_devflat_device_bytes(r->df[l]);
        vgan_hc_devflat_free(r->df[l]);
    }
    r->pst.device_bytes += r->df_bytes;
    r->pst.n_host_reads = r->n_host_reads;
    r->pst.n_device_reads = r->n_device_reads;
    r->pst.ms_wait_contexts = r->ms_wait_contexts;
    if (stats) *stats = r->fst;
    if (pstats) *pstats = r->pst;
    const int rc = r->rc;
    const std::string err = r->err;
    delete r;
    if (rc < 0) return fail(rc, "%s", err.c_str());
    return VGAN_OK;
}

extern "C" int vgan_hc_accumulate_gam_bytes(vgan_hc_ctx *const *ctxs, int n_ctx, const vgan_graph *graph, const void *bytes, uint64_t n,
                                            const vgan_gampipe_opts *opts, vgan_hc_flatten_stats *stats, vgan_gampipe_stats *pstats) {
    if (!ctxs || n_ctx <= 0 || !graph) return fail(VGAN_EINVAL, "vgan_hc_accumulate_gam_bytes: null argument");
    std::vector<int> dev((size_t)n_ctx);
    for (int i = 0; i < n_ctx; ++i) {
        if (!ctxs[i]) return fail(VGAN_EINVAL, "vgan_hc_accumulate_gam_bytes: null context");
        dev[(size_t)i] = hc_ctx_info(ctxs[i]).device;
    }
    vgan_hc_gamrun *r = nullptr;
    int rc;
    if ((rc = vgan_hc_gam_start(dev.data(), n_ctx, bytes, n, opts, &r)) < 0) return rc;
    if ((rc = vgan_hc_gam_attach(r, ctxs, n_ctx, graph)) < 0) {
        (void)vgan_hc_gam_finish(r, nullptr, nullptr);
        return rc;
    }
    return vgan_hc_gam_finish(r, stats, pstats);
}

// ------------------------------------------------------------------------------------------------------------ euka
// vgan euka's consumer (reference: src/readGAM_Euka.h:581 -- readGAM3's loop over the stream -- feeding the lambda of :67-577): device
// flatten of the piece into a vgan_euka_batch in HBM + the read kernel on the lane's context; the per-read results (clade, pass, |sequence|:
// what the abundance chain and the report read, MCMC.cpp:1192-1193) come down with the index of their read in the file.  The reads the
// device flatten leaves go through the host's parser, vgan_euka_flatten and the same context, behind the slot's back.
namespace {
struct DevArr { // a device array that grows
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap && p) return VGAN_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipMalloc(&p, want) != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            return fail(VGAN_ENOMEM, "vgan_euka_gam: hipMalloc of %zu bytes failed", want);
        }
        cap = want;
        return VGAN_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
struct EukaLeftJob {
    std::thread t;
    int rc = VGAN_OK;
    std::string err;
};
} // namespace

struct vgan_euka_gamrun : GamConsumer {
    std::vector<int> devices;
    const void *bytes = nullptr;
    uint64_t n = 0;
    vgan_gampipe_opts opts{};
    std::thread coord;
    int rc = VGAN_OK;
    std::string err;
    vgan_gampipe_stats pst{};
    std::mutex mu;
    std::condition_variable cv;
    bool attached = false, gave_up = false;
    std::vector<vgan_euka_ctx *> ctx;
    const vgan_graph *graph = nullptr;
    std::vector<vgan_euka_devflat *> df;
    std::vector<DevArr> o_clade, o_d, o_pass;
    std::deque<std::mutex> lane_mu;
    std::vector<std::shared_ptr<EukaLeftJob>> bg;
    // per processed read, in the order the pieces came (sorted by `idx` at the end)
    std::vector<uint32_t> idx;
    std::vector<int32_t> clade;
    std::vector<uint8_t> pass;
    std::vector<uint16_t> len;
    int64_t n_bad = 0;
    uint64_t n_host_reads = 0, n_device_reads = 0;
    size_t df_bytes = 0;
    double ms_wait_contexts = 0;
    int host_threads = 2;

    void aborted() override {
        {
            std::lock_guard<std::mutex> lk(mu);
            gave_up = true;
        }
        cv.notify_all();
    }
    int consume(int lane, vgan_gamdev *g, uint64_t read_base, const uint8_t *, int64_t) override {
        {
            const auto t0 = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return attached || gave_up; });
            if (!attached) return fail(VGAN_ESTATE, "vgan_euka_gam: no contexts were attached");
            ms_wait_contexts = std::max(ms_wait_contexts, ms_since(t0));
        }
        vgan_euka_ctx *cx = ctx[(size_t)lane];
        uint64_t sz[8];
        (void)vgan_gamdev_sizes(g, sz, nullptr);
        const uint64_t R = sz[2];
        if (read_base + R > 0xFFFFFFF0ull) return fail(VGAN_ERANGE, "vgan_euka_gam: more than 2^32 reads");
        std::vector<uint8_t> mask((size_t)R, 0);
        std::vector<int32_t> h_clade;
        std::vector<uint8_t> h_pass;
        int r;
        {
            std::lock_guard<std::mutex> lk(lane_mu[(size_t)lane]); // (the lane's context, its stream and its flatten object: one piece at a time)
            const size_t l = (size_t)lane;
            if (!df[l] && (r = vgan_euka_devflat_create(cx, graph, &df[l])) < 0) return r;
            vgan_euka_batch b;
            vgan_euka_flatten_stats st{};
            if ((r = vgan_euka_devflat_run_gamdev(df[l], g, (uint32_t)read_base, &b, mask.data(), &st)) < 0) return r;
            if (b.n_reads) {
                const size_t nr = b.n_reads;
                if ((r = o_clade[l].reserve(nr * 4)) || (r = o_d[l].reserve(nr * 32)) || (r = o_pass[l].reserve(nr))) return r;
                double *d = (double *)o_d[l].p;
                vgan_euka_read_out out{(int32_t *)o_clade[l].p, d, d + nr, d + 2 * nr, d + 3 * nr, (uint8_t *)o_pass[l].p};
                if ((r = vgan_euka_accumulate(cx, &b, &out)) < 0) return r;
                if ((r = vgan_euka_synchronize(cx)) < 0) return r;
                h_clade.resize(nr);
                h_pass.resize(nr);
                if (hipMemcpy(h_clade.data(), o_clade[l].p, nr * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(h_pass.data(), o_pass[l].p, nr, hipMemcpyDeviceToHost) != hipSuccess)
                    return fail(VGAN_ENODEV, "vgan_euka_gam: the per-read results did not come down");
                const uint32_t *src = nullptr;
                const uint16_t *sl = nullptr;
                (void)vgan_euka_devflat_host_arrays(df[l], &src, &sl);
                std::lock_guard<std::mutex> lk2(mu);
                idx.insert(idx.end(), src, src + nr);
                len.insert(len.end(), sl, sl + nr);
                clade.insert(clade.end(), h_clade.begin(), h_clade.end());
                pass.insert(pass.end(), h_pass.begin(), h_pass.end());
                n_device_reads += nr;
            }
        }
        // ---- the reads left to the host: their messages down now (the object's buffers are the next piece's after this call), the rest behind
        std::vector<uint32_t> where;
        for (uint64_t i = 0; i < R; ++i)
            if (mask[(size_t)i]) where.push_back((uint32_t)(read_base + i));
        if (where.empty()) return VGAN_OK;
        uint64_t nm = 0, nb = 0;
        if ((r = vgan_gamdev_pick(g, mask.data(), &nm, &nb)) < 0) return r;
        if (nm != where.size()) return fail(VGAN_ESTATE, "vgan_euka_gam: %llu messages picked for %zu reads", (unsigned long long)nm, where.size());
        auto offs = std::make_shared<std::vector<uint64_t>>((size_t)nm + 1);
        auto msgs = std::make_shared<std::vector<uint8_t>>((size_t)std::max<uint64_t>(nb, 1));
        if ((r = vgan_gamdev_picked(g, offs->data(), msgs->data())) < 0) return r;
        auto job = std::make_shared<EukaLeftJob>();
        vgan_euka_gamrun *self = this;
        auto whr = std::make_shared<std::vector<uint32_t>>(std::move(where));
        job->t = std::thread([self, job, offs, msgs, whr, cx, lane] {
            vgan_alnparts *parts = nullptr;
            vgan_alnset merged;
            vgan_euka_host_batch *hb = nullptr;
            vgan_euka_flatten_stats st{};
            // (keep_unmapped: the parse on the device dropped identity == 0 already; every message handed back is a read)
            if ((job->rc = vgan_alnparts_from_messages(msgs->data(), offs->data(), (int64_t)whr->size(), 1, self->host_threads, &parts)) >= 0) {
                merge_alnsets(parts->parts, merged);
                vgan_alnparts_free(parts);
                if (merged.n_reads() != (int64_t)whr->size()) job->rc = fail(VGAN_ESTATE, "vgan_euka_gam: %lld reads parsed of %zu messages", (long long)merged.n_reads(), whr->size());
            }
            if (job->rc >= 0) job->rc = vgan_euka_flatten(self->graph, &merged, 0, merged.n_reads(), self->host_threads, &hb, &st);
            std::vector<int32_t> c;
            std::vector<uint8_t> p;
            vgan_euka_batch b{};
            if (job->rc >= 0 && (job->rc = vgan_euka_host_batch_get(hb, &b)) >= 0 && b.n_reads) {
                const size_t nr = b.n_reads;
                c.resize(nr);
                p.resize(nr);
                std::vector<double> d(4 * nr);
                vgan_euka_read_out out{c.data(), d.data(), d.data() + nr, d.data() + 2 * nr, d.data() + 3 * nr, p.data()};
                std::lock_guard<std::mutex> lk(self->lane_mu[(size_t)lane]);
                job->rc = vgan_euka_accumulate(cx, &b, &out);
            }
            if (job->rc < 0) {
                job->err = last_error();
            } else {
                std::lock_guard<std::mutex> lk(self->mu);
                for (size_t i = 0; i < b.n_reads; ++i) {
                    self->idx.push_back((*whr)[b.read_src[i]]);
                    self->len.push_back(b.read_seq_len[i]);
                    self->clade.push_back(c[i]);
                    self->pass.push_back(p[i]);
                }
                self->n_bad += st.n_bad;
                self->n_host_reads += whr->size();
            }
            vgan_euka_host_batch_free(hb);
        });
        std::lock_guard<std::mutex> lk(mu);
        bg.push_back(job);
        return VGAN_OK;
    }
};

extern "C" int vgan_euka_gam_start(const int *devices, int n_lanes, const void *bytes, uint64_t n, const vgan_gampipe_opts *opts, vgan_euka_gamrun **out) {
    if (!devices || n_lanes <= 0 || (!bytes && n) || !out) return fail(VGAN_EINVAL, "vgan_euka_gam_start: null argument");
    auto *r = new vgan_euka_gamrun();
    r->devices.assign(devices, devices + n_lanes);
    r->bytes = bytes;
    r->n = n;
    r->opts = with_defaults(opts, n, n_lanes);
    r->opts.keep_unmapped = 0;   // readGAM_Euka.h:72: identity == 0 is no fragment of any clade (the messages are counted all the same)
    r->opts.mark_duplicates = 0; // (euka removes no duplicates)
    r->df.assign((size_t)n_lanes, nullptr);
    r->o_clade.resize((size_t)n_lanes), r->o_d.resize((size_t)n_lanes), r->o_pass.resize((size_t)n_lanes);
    r->lane_mu.resize((size_t)n_lanes);
    const int cpus = r->opts.n_threads > 0 ? r->opts.n_threads : (int)usable_cpus();
    r->host_threads = std::max(1, std::min(8, cpus / std::max(1, n_lanes * r->opts.slots)));
    r->coord = std::thread([r] {
        r->rc = gampipe_run(r->bytes, r->n, r->devices, r->opts, *r, &r->pst);
        if (r->rc < 0) r->err = last_error();
    });
    *out = r;
    return VGAN_OK;
}

extern "C" int vgan_euka_gam_attach(vgan_euka_gamrun *r, vgan_euka_ctx *const *ctxs, int n_ctx, const vgan_graph *graph) {
    if (!r || !ctxs || !graph) return fail(VGAN_EINVAL, "vgan_euka_gam_attach: null argument");
    if (n_ctx != (int)r->devices.size()) return fail(VGAN_EINVAL, "vgan_euka_gam_attach: %d contexts for %zu lanes", n_ctx, r->devices.size());
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if (r->attached) return fail(VGAN_ESTATE, "vgan_euka_gam_attach: called twice");
        r->ctx.assign(ctxs, ctxs + n_ctx);
        r->graph = graph;
        r->attached = true;
    }
    r->cv.notify_all();
    return VGAN_OK;
}

extern "C" int vgan_euka_gam_finish(vgan_euka_gamrun *r, vgan_euka_gam_result *res, vgan_gampipe_stats *pstats) {
    if (!r) return fail(VGAN_EINVAL, "vgan_euka_gam_finish: null argument");
    if (res) memset(res, 0, sizeof *res);
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if (!r->attached) r->gave_up = true;
    }
    r->cv.notify_all();
    if (r->coord.joinable()) r->coord.join();
    for (auto &j : r->bg) {
        if (j->t.joinable()) j->t.join();
        if (j->rc < 0 && r->rc >= 0) {
            r->rc = j->rc;
            r->err = "the reads left to the host: " + j->err;
        }
    }
    r->bg.clear();
    for (size_t l = 0; l < r->df.size(); ++l) {
        if (l < r->ctx.size() && r->ctx[l]) (void)vgan_euka_synchronize(r->ctx[l]);
        r->df_bytes += euka_devflat_device_bytes(r->df[l]) + r->o_clade[l].cap + r->o_d[l].cap + r->o_pass[l].cap;
        vgan_euka_devflat_free(r->df[l]);
        r->df[l] = nullptr;
        if (l < r->devices.size()) (void)hipSetDevice(r->devices[l]);
        r->o_clade[l].release(), r->o_d[l].release(), r->o_pass[l].release();
    }
    r->pst.device_bytes += r->df_bytes;
    r->pst.n_host_reads = r->n_host_reads;
    r->pst.n_device_reads = r->n_device_reads;
    r->pst.ms_wait_contexts = r->ms_wait_contexts;
    if (pstats) *pstats = r->pst;
    if (r->rc < 0) return fail(r->rc, "%s", r->err.c_str());
    { // the per-read lists in the order of the file: every index below n_mapped occurs at most once -- a placement, not a sort
      // (std::sort through an index array took 0.4 s for 5 M reads: as long as the whole pipeline)
        const size_t nr = r->idx.size(), nm = (size_t)r->pst.n_reads;
        std::vector<uint32_t> at(nm, 0xFFFFFFFFu);
        for (size_t i = 0; i < nr; ++i)
            if (r->idx[i] < nm) at[r->idx[i]] = (uint32_t)i;
        std::vector<uint32_t> i2;
        std::vector<int32_t> c2;
        std::vector<uint8_t> p2;
        std::vector<uint16_t> l2;
        i2.reserve(nr), c2.reserve(nr), p2.reserve(nr), l2.reserve(nr);
        for (size_t k = 0; k < nm; ++k) {
            const uint32_t i = at[k];
            if (i == 0xFFFFFFFFu) continue;
            i2.push_back((uint32_t)k);
            c2.push_back(r->clade[i]);
            p2.push_back(r->pass[i]);
            l2.push_back(r->len[i]);
        }
        if (i2.size() != nr) return fail(VGAN_ESTATE, "vgan_euka_gam_finish: %zu results for %zu distinct reads", nr, i2.size());
        r->idx.swap(i2), r->clade.swap(c2), r->pass.swap(p2), r->len.swap(l2);
    }
    if (res) {
        res->n_messages = (int64_t)r->pst.n_messages;
        res->n_mapped = (int64_t)r->pst.n_reads;
        res->n_bad = r->n_bad;
        res->n_reads = (int64_t)r->idx.size();
        res->read_index = r->idx.data();
        res->read_clade = r->clade.data();
        res->read_pass = r->pass.data();
        res->read_seq_len = r->len.data();
    }
    return VGAN_OK;
}

extern "C" void vgan_euka_gam_free(vgan_euka_gamrun *r) {
    if (!r) return;
    if (r->coord.joinable()) (void)vgan_euka_gam_finish(r, nullptr, nullptr);
    delete r;
}
