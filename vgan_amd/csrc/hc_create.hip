// The HaploCart context: what it holds of the graph on the device (mask tiles, node scalars and classes, the tables of column
// terms, tables of the error model; reference: src/load.cpp, src/get_p_obs_base.cpp:44-64, src/miscfunc.h:180-212), its making and its end.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <chrono>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "hc_ctx.h"
#include "module_anchor.h"
#include "hc_device.h"
#include "host/common.h"
#include "vgan_gpu.h"

using namespace vgan;

#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(VGAN_ENODEV, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

namespace {
inline bool in_range(unsigned lo, unsigned hi, unsigned x) { return lo <= x && x <= hi; }

// src/get_p_obs_base.cpp:44-64 (Q1/Q2: the protein-coding rate is 0 by integer division)
double match_prob(int pangenome_base) {
    const unsigned b = (unsigned)pangenome_base;
    double mu;
    if (in_range(57, 372, b)) mu = 1.64273e-7;
    else if (in_range(1, 56, b) || in_range(373, 576, b)) mu = 2.29640e-8;
    else if (in_range(16384, 16569, b)) mu = 1.54555e-8;
    else if (in_range(3307, 4262, b) || in_range(4470, 5511, b) || in_range(5904, 7445, b) || in_range(7586, 8269, b) ||
             in_range(8366, 9990, b) || in_range(10059, 10403, b) || in_range(10470, 12137, b) ||
             in_range(12337, 14673, b) || in_range(14747, 15886, b))
        mu = 0.0;
    else if (in_range(577, 647, b) || in_range(1602, 1670, b) || in_range(3230, 3304, b) || in_range(4263, 4400, b) ||
             in_range(4402, 4469, b) || in_range(5512, 5579, b) || in_range(5587, 5654, b) || in_range(5657, 5728, b) ||
             in_range(5761, 5891, b) || in_range(7446, 7514, b) || in_range(7518, 7585, b) || in_range(8295, 8364, b) ||
             in_range(15888, 15953, b) || in_range(15956, 16023, b))
        mu = 6.91285e-9;
    else if (in_range(648, 1601, b) || in_range(1671, 3229, b)) mu = 6.91285e-9;
    else mu = 2.48537e-8;
    mu *= 30;
    return pow((1 - mu), 8);
}


void parse_relatives(const char *txt, std::unordered_map<std::string, std::vector<std::string>> &rel) {
    // src/load.cpp:303-345: "name tok tok ...", tokens containing '[' dropped, first insertion wins
    std::istringstream in(txt ? txt : "");
    std::string line, tok;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::vector<std::string> t;
        while (ls >> tok) t.push_back(tok);
        if (t.empty()) continue;
        std::vector<std::string> v;
        for (size_t j = 1; j < t.size(); ++j)
            if (t[j].find('[') == std::string::npos) v.push_back(t[j]);
        rel.emplace(t[0], std::move(v));
    }
}
} // namespace

extern "C" int vgan_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The first HIP call of a process loads the runtime and the library's code objects (~0.25 s): a front end calls this on a
// thread of its own at start-up, while it reads its graph, so that the context creation finds the device ready.
extern "C" int vgan_device_warmup(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(VGAN_ENODEV, "vgan_device_warmup: no HIP device %d", device);
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(nullptr));
    return VGAN_OK;
}

// The code objects of the kernels a run will launch, loaded now and beside each other (a thread per translation unit): left to the first
// launch of each, they load one after the other along the first piece's way through the stages -- upload, inflate, framing, protobuf walk,
// flatten each waited 40-70 ms for the next one's on the 10 M-read file, ~0.25 s in all.  what: VGAN_PRELOAD_* bits.
extern "C" int vgan_device_preload(int device, unsigned what) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(VGAN_ENODEV, "vgan_device_preload: no HIP device %d", device);
    std::vector<std::pair<const char *, const void *>> fns;
    // (the runtime loads them one after the other whatever the threads do, roughly in the order they ask: what the contexts' creation
    // launches first -- small --, then the stages of a piece in the order it meets them)
    if (what & VGAN_PRELOAD_HC) fns.push_back({"hc col8", anchor_hc_col8()}), fns.push_back({"hc sweep", anchor_hc_kernels()});
    if (what & VGAN_PRELOAD_EUKA) fns.push_back({"euka", anchor_euka_kernels()});
    if (what & VGAN_PRELOAD_SB) fns.push_back({"soibean", anchor_sb_kernels()});
    if (what & VGAN_PRELOAD_GAM) fns.push_back({"inflate", anchor_gam_inflate_wave()}), fns.push_back({"framing + protobuf", anchor_gam_kernels()});
    if (what & VGAN_PRELOAD_HC) fns.push_back({"hc flatten", anchor_hc_flatten()}), fns.push_back({"hc wave", anchor_hc_wave()});
    if (what & VGAN_PRELOAD_EUKA) fns.push_back({"euka flatten", anchor_euka_flatten()});
    if (what & VGAN_PRELOAD_SB) fns.push_back({"soibean flatten", anchor_sb_flatten()});
    const bool timing = getenv("VGAN_TIMING") != nullptr;
    std::vector<std::thread> ts;
    std::vector<double> ms(fns.size(), 0.0);
    for (size_t i = 0; i < fns.size(); ++i) {
        if (i) std::this_thread::sleep_for(std::chrono::microseconds(200)); // (so that they ask in this order)
        ts.emplace_back([&, i] {
            const auto t0 = std::chrono::steady_clock::now();
            hipFuncAttributes a;
            ms[i] = -1.0; // (until the kernel has been found)
            const bool ok = hipSetDevice(device) == hipSuccess && hipFuncGetAttributes(&a, fns[i].second) == hipSuccess;
            (void)hipGetLastError();
            if (ok) ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        });
    }
    for (auto &t : ts) t.join();
    for (size_t i = 0; i < fns.size(); ++i)
        if (ms[i] < 0) return fail(VGAN_ENODEV, "vgan_device_preload: the runtime does not find the kernels of \"%s\"", fns[i].first);
    if (timing) {
        std::string line;
        for (size_t i = 0; i < fns.size(); ++i) line += (i ? ", " : "") + std::string(fns[i].first) + " " + std::to_string((int)(ms[i] + 0.5));
        fprintf(stderr, "[vgan timing] code objects loaded beside each other (ms each, from the common start): %s\n", line.c_str());
    }
    return VGAN_OK;
}

extern "C" int vgan_hc_create(const vgan_graph_view *gv, const vgan_hc_params *params, int device, vgan_hc_ctx **out) {
    if (!gv || !params || !out) return fail(VGAN_EINVAL, "vgan_hc_create: null argument");
    if (gv->n_paths == 0 || gv->max_id < 0 || !gv->mask || !gv->pangenome_base || !gv->mappability)
        return fail(VGAN_EINVAL, "vgan_hc_create: incomplete graph view");
    if (!(params->background_error_prob >= 0.0 && params->background_error_prob <= 1.0))
        return fail(VGAN_EINVAL, "Error: background error probability must be between 0 and 1"); // HaploCart.cpp:107-113
    // (what follows until the first device call is host work -- the mask's transposition, the node classes, the tables, the names: ~0.1 s
    // on the hcfiles graph.  A caller that starts this beside the runtime's start-up, vgan_device_warmup on another thread, has it done
    // when the runtime is: the device is first asked for below, where the arrays go up.)
    auto c = new vgan_hc_ctx();
    c->device = device;
    c->P = gv->n_paths;
    c->W = (gv->n_paths + 63) / 64;
    c->rows = (uint32_t)gv->max_id + 1;
    c->n_tiles = 8 * ((c->W + 126) / 127); // at most 16 words per tile (one bit each in a 16-bit entry): floor(W / n_tiles) <= 15
    c->prm.bep = params->background_error_prob;
    c->prm.use_bep = params->use_background_error_prob != 0;
    c->prm.consensus = params->is_consensus_fasta != 0;
    int rc = VGAN_OK;
    auto bail = [&](int code) {
        vgan_hc_destroy(c);
        return code;
    };
    // unsupported-path mask: plain rows + the per-tile bit-transposed copy the sweep reads (hc_device.h)
    std::vector<uint64_t> um((size_t)c->rows * c->W, 0);
    for (uint32_t r = 0; r < c->rows; ++r) {
        const uint64_t *src = gv->mask + (size_t)r * c->W;
        uint64_t *dst = &um[(size_t)r * c->W];
        for (uint32_t w = 0; w < c->W; ++w) {
            uint64_t valid = ~0ull;
            if (w == c->W - 1 && (c->P & 63)) valid = (1ull << (c->P & 63)) - 1;
            dst[w] = ~src[w] & valid;
        }
    }
    const uint32_t tile_base = c->W / c->n_tiles, tile_rem = c->W % c->n_tiles;
    if (tile_base > 15) return bail(fail(VGAN_ERANGE, "vgan_hc_create: internal tile split out of range (%u words per tile)", tile_base));
    std::vector<uint16_t> tw0(c->n_tiles + 1, 0);
    for (uint32_t t = 0; t < c->n_tiles; ++t) tw0[t + 1] = (uint16_t)(tw0[t] + tile_base + (t < tile_rem ? 1 : 0));
    const uint32_t row_entries = c->n_tiles * 64;
    std::vector<uint16_t> umT((size_t)c->rows * row_entries, 0);
    {
        auto rows_range = [&](uint32_t r0, uint32_t r1) {
            for (uint32_t r = r0; r < r1; ++r) {
                const uint64_t *src = &um[(size_t)r * c->W];
                uint16_t *dst = &umT[(size_t)r * row_entries];
                for (uint32_t t = 0; t < c->n_tiles; ++t) {
                    for (uint32_t k = 0; k < (uint32_t)(tw0[t + 1] - tw0[t]); ++k) {
                        uint64_t bits = src[tw0[t] + k];
                        while (bits) {
                            const int l = __builtin_ctzll(bits);
                            bits &= bits - 1;
                            dst[t * 64 + l] |= (uint16_t)(1u << (15 - k));
                        }
                    }
                }
            }
        };
        const uint32_t nth = std::min<uint32_t>(16, std::max<uint32_t>(1, std::min<uint32_t>(std::thread::hardware_concurrency(), c->rows / 256)));
        if (nth <= 1) {
            rows_range(0, c->rows);
        } else {
            std::vector<std::thread> th;
            for (uint32_t t = 0; t < nth; ++t)
                th.emplace_back(rows_range, (uint32_t)((uint64_t)c->rows * t / nth), (uint32_t)((uint64_t)c->rows * (t + 1) / nth));
            for (auto &t : th) t.join();
        }
    }
    std::vector<HcNodeDev> nt(c->rows);
    const bool consensus = params->is_consensus_fasta != 0;
    for (uint32_t r = 0; r < c->rows; ++r) {
        const int32_t pb = gv->pangenome_base[r];
        double mp = 0.0, mt = 1.0;
        if (pb >= 0 && (uint64_t)pb < gv->n_mappability) {
            mp = gv->mappability[pb];
            mt = match_prob(pb);
        }
        // log(0) = -inf and 1/0 = inf are meant: the kernel scores such a segment from wbg alone (hc_kernels.hip)
        const long double mm = (long double)mp * (long double)mt;
        nt[r] = {(double)logl(consensus ? (long double)mt : mm), (double)(1.0L / mm), mp, mt};
    }
    // node classes: the distinct {ln_w, inv_mm, mappability} triples, most frequent first (hc_col8_kernels.hip keeps a table of
    // column terms for the first HC_MEMO_CLASSES of them in LDS and one for all of them in HBM; a graph with more than
    // HC_EXT_NODE_CLASSES -- a mappability track with that many distinct values -- takes the older kernels)
    std::vector<HcNodeDev> cls;
    std::vector<uint16_t> nhi(c->rows, 0);
    {
        auto same = [](const HcNodeDev &x, const HcNodeDev &y) { return memcmp(&x, &y, 3 * sizeof(double)) == 0; };
        std::vector<uint32_t> of(c->rows), cnt;
        bool many = false;
        for (uint32_t r = 0; r < c->rows && !many; ++r) {
            uint32_t k = 0;
            while (k < cls.size() && !same(cls[k], nt[r])) ++k;
            if (k == cls.size()) {
                if (cls.size() == HC_EXT_NODE_CLASSES) {
                    many = true;
                    break;
                }
                cls.push_back(nt[r]);
                cnt.push_back(0);
            }
            of[r] = k;
            cnt[k]++;
        }
        if (many) {
            cls.clear();
        } else {
            std::vector<uint32_t> ord(cls.size()), rank(cls.size());
            for (uint32_t k = 0; k < ord.size(); ++k) ord[k] = k;
            std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return cnt[x] > cnt[y]; });
            std::vector<HcNodeDev> sorted(cls.size());
            for (uint32_t i = 0; i < ord.size(); ++i) {
                rank[ord[i]] = i;
                sorted[i] = cls[ord[i]];
            }
            cls.swap(sorted);
            for (uint32_t r = 0; r < c->rows; ++r) {
                const uint32_t k = rank[of[r]];
                nhi[r] = (uint16_t)(k < HC_MEMO_CLASSES ? k * HC_MEMO_CLASS_BYTES : 0xE000u | k);
            }
        }
    }
    std::vector<double> tb(756);
    for (int bte = 0; bte < 256; ++bte) { // src/miscfunc.h:180-188 on int(char)
        const int Q = (int)(int8_t)bte;
        tb[bte] = log(Q > 2 ? pow(10, ((-1 * Q) * 0.1)) : 0.25);
    }
    for (int Q = 0; Q < 100; ++Q) { // src/miscfunc.h:199-212, src/haplocart_functions.cpp:101-107
        tb[256 + Q] = Q > 2 ? pow(10, ((-1 * Q) * 0.1)) : 0.25;
        tb[356 + Q] = pow(10, ((-1 * Q) * 0.1));
        const double om = 1.0 - tb[356 + Q]; // process_mapping.cpp:41: pcm = (1 - incorrect_mapping_vec[mapq]) * mappability
        tb[456 + 3 * Q] = om;
        tb[456 + 3 * Q + 1] = consensus ? (double)logl(1.0L - (long double)params->background_error_prob) : (double)logl((long double)om);
        tb[456 + 3 * Q + 2] = (double)(1.0L / (long double)om);
    }
    // posterior side tables
    {
        std::istringstream in(gv->path_names ? gv->path_names : "");
        std::string line;
        while (std::getline(in, line)) {
            std::istringstream ls(line);
            std::string tok;
            if (!(ls >> tok)) continue;
            c->path_index.emplace(tok, (uint32_t)c->path_names.size());
            c->path_names.push_back(tok);
        }
        parse_relatives(gv->parents_txt, c->parents);
        parse_relatives(gv->children_txt, c->children);
    }
    // ---- the device
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            return bail(fail(VGAN_ENODEV, "vgan_hc_create: no HIP device is visible (this library has no CPU path)"));
        if (device < 0 || device >= ndev) return bail(fail(VGAN_EINVAL, "vgan_hc_create: device %d out of range (%d visible)", device, ndev));
        if (hipSetDevice(device) != hipSuccess) return bail(fail(VGAN_ENODEV, "vgan_hc_create: hipSetDevice(%d) failed", device));
        if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(VGAN_ENODEV, "hipStreamCreate failed"));
        c->stream = c->own_stream;
    }
    const size_t accn = (size_t)c->W * 64;
    if ((rc = c->umask.reserve(um.size())) || (rc = c->umaskT.reserve(umT.size())) ||
        (rc = c->tile_word0.reserve(tw0.size())) || (rc = c->node_tab.reserve(nt.size())) || (rc = c->tables.reserve(756)) ||
        (rc = c->node_hi.reserve(nhi.size())) || (rc = c->cls_tab.reserve(std::max<size_t>(1, cls.size()))) ||
        (rc = c->accum.reserve(((size_t)c->rows + 7) / 8 * 8 + 2 * accn + HC_TOTAL_SLOTS * HC_TOTAL_STRIDE)) || (rc = c->final_vec.reserve(c->P)))
        return bail(rc);
    { // sub-ranges of the accumulator block (sweep kernels read 64-byte aligned blocks of weights: keep 64-byte offsets)
        const size_t nW = ((size_t)c->rows + 7) / 8 * 8;
        c->nodeW.p = c->accum.p;
        c->acc_seg.p = c->accum.p + nW;
        c->acc_node.p = c->acc_seg.p + accn;
        c->totals.p = c->acc_node.p + accn;
        c->accum_n = nW + 2 * accn + HC_TOTAL_SLOTS * HC_TOTAL_STRIDE;
    }
    if (hipMemcpy(c->umask.p, um.data(), um.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->umaskT.p, umT.data(), umT.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->tile_word0.p, tw0.data(), tw0.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->node_tab.p, nt.data(), nt.size() * sizeof(HcNodeDev), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->tables.p, tb.data(), tb.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->node_hi.p, nhi.data(), nhi.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        (!cls.empty() && hipMemcpy(c->cls_tab.p, cls.data(), cls.size() * sizeof(HcNodeDev), hipMemcpyHostToDevice) != hipSuccess))
        return bail(fail(VGAN_ENODEV, "vgan_hc_create: upload failed"));
    c->g.umask = c->umask.p;
    c->g.umaskT = c->umaskT.p;
    c->g.tile_word0 = c->tile_word0.p;
    c->g.node_tab = c->node_tab.p;
    c->g.lq = c->tables.p;
    c->g.qscore = c->tables.p + 256;
    c->g.incmap = c->tables.p + 356;
    c->g.rdtab = c->tables.p + 456;
    c->g.node_hi = c->node_hi.p;
    c->g.cls_tab = c->cls_tab.p;
    c->g.n_cls = (uint32_t)cls.size();
    c->g.col_memo = nullptr;
    c->g.col_memo2 = nullptr;
    if (c->g.n_cls) { // the table of column terms (a pure function of the graph's node classes and the error-rate parameters)
        if ((rc = c->col_memo.reserve(hc_col8_memo_doubles()))) return bail(rc);
        launch_hc_col8_memo(c->g, c->prm, c->col_memo.p, c->stream);
        if (hipGetLastError() != hipSuccess) return bail(fail(VGAN_ENODEV, "vgan_hc_create: the table of column terms could not be built"));
        c->g.col_memo = c->col_memo.p;
        if ((rc = c->col_memo2.reserve(hc_col8_memo2_doubles(c->g.n_cls)))) return bail(rc);
        launch_hc_col8_memo2(c->g, c->prm, c->col_memo2.p, c->stream);
        if (hipGetLastError() != hipSuccess) return bail(fail(VGAN_ENODEV, "vgan_hc_create: the wide table of column terms could not be built"));
        c->g.col_memo2 = c->col_memo2.p;
    }
    c->g.rows = c->rows;
    c->g.mask_words = c->W;
    c->g.row_entries = row_entries;
    c->g.n_tiles = c->n_tiles;
    c->g.tile_base_words = tile_base;
    c->g.n_paths = c->P;
    if ((rc = vgan_hc_reset(c))) return bail(rc);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(fail(VGAN_ENODEV, "vgan_hc_create: sync failed"));
    *out = c;
    return VGAN_OK;
}

extern "C" void vgan_hc_destroy(vgan_hc_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    c->umask.release();
    c->umaskT.release();
    c->tile_word0.release();
    c->node_tab.release();
    c->cls_tab.release();
    c->col_memo.release();
    c->col_memo2.release();
    c->node_hi.release();
    c->tables.release();
    c->accum.release();
    c->final_vec.release();
    c->segD.release();
    c->segS.release();
    c->segU.release();
    c->dump.release();
    c->s_u32.release();
    c->s_u16.release();
    c->s_u8.release();
    c->scratch_pack.release();
    c->work_ctr.release();
    c->lists.release();
    c->conf.release();
    for (auto &t : c->timed) {
        (void)hipEventDestroy(t.a);
        (void)hipEventDestroy(t.b);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}
