// vgan -- C++ host driver keeping the reference's subcommand surface for the GPU hot path.
//
//   vgan haplocart  -g reads.gam --hc-files DIR [-e P] [-o OUT] [-s NAME] [-np] [-pf FILE] [-q] [-t N] [-d] ...
//
// Mirrors Haplocart::run (reference src/HaploCart.cpp:58-488): same flags, same validation messages, same
// output lines.  What differs, and why:
//   * the graph is read from DIR/graph.gfa or DIR/graph.og (+ the hcfiles sidecars) by this build's own readers;
//   * FASTQ / FASTA inputs need vg giraffe in-process (src/map_giraffe.cpp), which is not available: map with vg
//     and pass the sorted GAM with -g (for a consensus FASTA mapped that way add -f NAME to get the reference's
//     consensus arithmetic);
//   * the likelihood loop runs on the GPU through the C-ABI (include/vgan_gpu.h); -t only sizes the host front end.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <deque>
#include <condition_variable>
#include <iostream>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include <unistd.h>

#include "vgan_gpu.h"

#include <chrono>
#include <thread>

#include "cli_util.h"

int euka_main(int argc, char **argv);    // vgan_euka_main.cpp
int soibean_main(int argc, char **argv); // vgan_soibean_main.cpp

namespace {
using namespace vgan_cli;

std::string haplocart_usage() {
    return "\n vgan haplocart [options]\n\n"
           " Predict the mitochondrial haplogroup of a sample on the GPU (MI355X).\n\n"
           " Input:\n"
           "   -g  [STR]        GAM input (sorted; FASTQ/FASTA input needs vg giraffe: map first)\n"
           "   -f  [STR]        treat the GAM as a mapped consensus FASTA (consensus arithmetic)\n"
           "   --hc-files [STR] HaploCart graph directory: graph.gfa + path_supports, parsed_pangenome_mapping,\n"
           "                    mappability.tsv, graph_paths, parents.txt, children.txt (plain or .gz)\n"
           " Options:\n"
           "   -e  [FLOAT]      background error probability for FASTA input (default 0.0001)\n"
           "   -o  [STR]        output file (default: stdout)\n"
           "   -s  [STR]        sample name\n"
           "   -np              do not compute clade-level posteriors\n"
           "   -pf [STR]        posterior output file (default: stdout)\n"
           "   -t  [INT]        host threads (-1 for all available)\n"
           "   -d               write per-haplogroup log-likelihoods to <out>.loglik.tsv\n"
           "   -j -jf [STR]     write every alignment as a line of JSON to the file (protobuf's JSON mapping as vg's pb2json\n"
           "                    configures it: the same fields and values, not verified byte for byte against vg's output)\n"
           "   -q               quiet\n"
           "   --keep-duplicates   skip duplicate removal\n"
           "   --per-read       stream the path-membership mask per read (the reference's loop order)\n"
           "   --device [INT]   GPU index (default 0)\n"
           "   --gpus [LIST]    GPU indices, comma separated, or `all`: one device context each (default: the one of --device;\n"
           "                    the environment's VGAN_GPUS is read the same way); the reads are dealt to the contexts chunk by\n"
           "                    chunk and the per-haplogroup log-likelihoods are reduced once at the end (RCCL between distinct\n"
           "                    GPUs; VGAN_HC_REDUCE=host sums on the host)\n";
}

int haplocart(int argc, char **argv) {
    bool debug = false, quiet = false, compute_posteriors = true, rmdup = true, per_read = false, webapp = false;
    std::string posteriorfilename = "/dev/stdout", outputfilename = "/dev/stdout", hcfiledir = "../share/vgan/hcfiles/";
    std::string gamfilename, fastafilename, fastq1, fastq2, samplename, jsonfilename;
    bool dump_json = false;
    bool invoked_samplename = false;
    double background_error_prob = 0.0001;
    int n_threads = 1, device = 0;
    std::string gpu_spec;
    std::vector<int> gpu_list;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char *flag) -> std::string {
            if (i + 1 >= argc) die(std::string("[HaploCart] Error, option ") + flag + " needs a value");
            return argv[++i];
        };
        if (a == "-" || a == "-h" || a == "--help") {
            std::cerr << haplocart_usage() << std::endl;
            return 0;
        } else if (a == "-d") debug = true;
        else if (a == "--hc-files") {
            hcfiledir = need("--hc-files");
            if (hcfiledir.back() != '/') hcfiledir += '/';
        } else if (a == "-e") {
            background_error_prob = parse_double(need("-e"), "-e", "[HaploCart]");
            if (background_error_prob < 0 || background_error_prob > 1)
                die("[HaploCart] Error, option -e is not a valid probability."); // HaploCart.cpp:107-113
        } else if (a == "-g") gamfilename = need("-g");
        else if (a == "-f") fastafilename = need("-f");
        else if (a == "-fq1") fastq1 = need("-fq1");
        else if (a == "-fq2") fastq2 = need("-fq2");
        else if (a == "-i") die("[HaploCart] interleaved FASTQ input needs vg giraffe; map with vg and pass -g");
        else if (a == "-j") dump_json = true;           // HaploCart.cpp:146-149
        else if (a == "-jf") jsonfilename = need("-jf"); // HaploCart.cpp:151-154
        else if (a == "-o") outputfilename = need("-o");
        else if (a == "-np") compute_posteriors = false;
        else if (a == "-pf") posteriorfilename = need("-pf");
        else if (a == "-q") quiet = true;
        else if (a == "-s") {
            samplename = need("-s");
            invoked_samplename = true;
        } else if (a == "-t") {
            n_threads = parse_int(need("-t"), "-t", "[HaploCart]");
            if (n_threads == 0 || n_threads < -1)
                die("[HaploCart] Error, invalid number of threads"); // HaploCart.cpp:183-194
            // host threads only, as in the reference: a command line written for the CPU build (`-t 32`) must not grab
            // every GPU of a shared node -- several GPUs are asked for by name (--gpus, or VGAN_GPUS in the environment)
            if (n_threads == -1) n_threads = 0;                      // all hardware threads
        } else if (a == "--gpus") {
            gpu_spec = need("--gpus");
        } else if (a == "-w") webapp = true;
        else if (a == "-z") (void)need("-z");
        else if (a == "--keep-duplicates") rmdup = false;
        else if (a == "--per-read") per_read = true;
        else if (a == "--device") {
            device = parse_int(need("--device"), "--device", "[HaploCart]");
            if (device < 0) die("[HaploCart] Error, --device needs a non-negative GPU index");
        }
        else die("[HaploCart] Error, unrecognized option " + a);
    }
    if (webapp) die("[HaploCart] webapp mode is not part of the GPU path");
    if (!fastq1.empty() || !fastq2.empty())
        die("[HaploCart] FASTQ input needs vg giraffe in-process, which this build does not have; map with vg and pass -g");
    if (dump_json && jsonfilename.empty()) die("[HaploCart] Error, cannot invoke -j without -jf"); // HaploCart.cpp:231
    if (gamfilename.empty()) die("[HaploCart] Error, no input file given (use -g)");
    if (!std::ifstream(gamfilename)) die("[HaploCart] Error, GAM input file " + gamfilename + " does not exist");
    if (!invoked_samplename) samplename = !fastafilename.empty() ? fastafilename : gamfilename;

    PhaseTimer pt("haplocart");
    const auto t_start = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) { // VGAN_TIMING: time since the subcommand started
        if (getenv("VGAN_TIMING"))
            fprintf(stderr, "[vgan timing] haplocart @ %.0f ms: %s\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), what);
    };
    // the GAM is inflated, framed and parsed behind this thread (vgan_gam_stream) while it loads the graph and brings the
    // device up; chunks are then flattened and sent to the device as they become available
    struct StreamCloser {
        vgan_gam_stream *s = nullptr;
        ~StreamCloser() { vgan_gam_stream_close(s); }
    } stream;
    // A long BGZF input: the front end runs ON THE DEVICE (vgan_hc_gam_*: the file in pieces through inflate, framing, protobuf walk,
    // duplicate marks and flatten as kernels; csrc/gam_pipe.hip, gam_kernels.hip) -- the file's bytes go up as they are and the host
    // parses only the reads the device flatten leaves (indels, soft clips: a percent or two).  With --gpus LIST piece i goes to context
    // i mod n: every device inflates and parses its own pieces.  The host pipeline costs ~3 us of CPU per read, which on a CPU quota
    // is what a long input's wall time follows.  VGAN_HC_DEVICE_GAM=0 / 1: never / whenever the input is a regular file;
    // anything the device refuses (not BGZF, a stream its segment walks cannot frame, no memory) goes through the host pipeline.
    struct FileMap {
        const uint8_t *p = nullptr;
        size_t n = 0;
        ~FileMap() {
            if (p) munmap(const_cast<uint8_t *>(p), n);
        }
    } gam_map;
    bool device_gam = false;
    {
        const char *e = getenv("VGAN_HC_DEVICE_GAM");
        struct stat sb;
        if (!(e && e[0] == '0') && !per_read && fastafilename.empty() && stat(gamfilename.c_str(), &sb) == 0 && S_ISREG(sb.st_mode) &&
            ((e && e[0] == '1') || (uint64_t)sb.st_size >= (384ull << 20)) && sb.st_size > 28) {
            const int fd = open(gamfilename.c_str(), O_RDONLY);
            if (fd >= 0) {
                void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                close(fd);
                if (m != MAP_FAILED) {
                    gam_map.p = (const uint8_t *)m;
                    gam_map.n = (size_t)sb.st_size;
                    device_gam = gam_map.p[0] == 0x1f && gam_map.p[1] == 0x8b && gam_map.p[2] == 8 && (gam_map.p[3] & 4); // (BGZF members carry an extra field)
                    if (device_gam) (void)madvise(m, gam_map.n, MADV_WILLNEED);
                }
            }
        }
    }
    if (!device_gam) check(vgan_gam_stream_open(gamfilename.c_str(), 0, &stream.s), "reading GAM");
    // the HIP runtime comes up on a thread of its own while the graph is read, the run's code objects behind it (cli_util.h: DeviceWarm)
    DeviceWarm warm;
    {
        int warm_dev = device; // (the first GPU the run will use: the runtime's start-up creates its context there)
        const std::string spec = !gpu_spec.empty() ? gpu_spec : (getenv("VGAN_GPUS") ? std::string(getenv("VGAN_GPUS")) : std::string());
        if (!spec.empty() && spec != "all" && isdigit((unsigned char)spec[0])) warm_dev = atoi(spec.c_str());
        warm.on_up = [&stamp] { stamp("HIP runtime up"); };
        warm.start(warm_dev, VGAN_PRELOAD_HC | (device_gam ? VGAN_PRELOAD_GAM : 0u));
    }
    // which GPUs: --gpus LIST, the environment's VGAN_GPUS, or the one of --device
    // (the HIP runtime is still coming up: only "--gpus all" has to wait for it here, to know how many contexts there will
    // be; otherwise the first chunks are decoded and flattened beside it and the context thread below is the one that waits)
    if (gpu_spec.empty())
        if (const char *e = getenv("VGAN_GPUS")) gpu_spec = e;
    if (gpu_spec == "all") {
        warm.wait_runtime();
        const int n_visible = vgan_device_count();
        if (n_visible <= 0) die("[HaploCart] no HIP device is visible: the likelihood path runs on the GPU only");
        for (int d = 0; d < n_visible; ++d) gpu_list.push_back(d);
    } else if (!gpu_spec.empty()) {
        size_t p0 = 0;
        while (p0 <= gpu_spec.size()) {
            size_t c1 = gpu_spec.find(',', p0);
            if (c1 == std::string::npos) c1 = gpu_spec.size();
            const int d = parse_int(gpu_spec.substr(p0, c1 - p0), "--gpus", "[HaploCart]");
            if (d < 0) die("[HaploCart] Error, --gpus needs non-negative GPU indices");
            gpu_list.push_back(d);
            p0 = c1 + 1;
        }
    }
    if (gpu_list.empty()) gpu_list.push_back(device);
    // (the device front end needs neither graph nor contexts for its first stages: it starts now -- the member index of the first piece,
    // then, as soon as the HIP runtime is up, upload, inflate, framing and parse -- beside the graph load and the contexts' set-up)
    struct GdRun {
        vgan_hc_gamrun *r = nullptr;
        ~GdRun() {
            if (r) (void)vgan_hc_gam_finish(r, nullptr, nullptr); // (an error's unwinding: the pieces give up, the threads are joined)
        }
    } gd;
    if (device_gam) {
        vgan_gampipe_opts po{};
        po.mark_duplicates = rmdup ? 1 : 0;
        po.n_threads = n_threads;
        if (vgan_hc_gam_start(gpu_list.data(), (int)gpu_list.size(), gam_map.p, gam_map.n, &po, &gd.r) < 0) {
            device_gam = false;
            check(vgan_gam_stream_open(gamfilename.c_str(), 0, &stream.s), "reading GAM");
        }
    }
    // -j -jf FILE: every alignment of the GAM as a line of JSON (readGAM.h:37-38 writes them while it reads; here a pass of
    // its own, on a thread beside the run).  The reference opens -jf FILE even without -j (and leaves it empty).
    struct JsonDump {
        std::thread t;
        int rc = 0;
        std::string err;
        ~JsonDump() {
            if (t.joinable()) t.join();
        }
    } json;
    if (dump_json) {
        json.t = std::thread([&] {
            json.rc = vgan_gam_dump_json(gamfilename.c_str(), jsonfilename.c_str(), nullptr);
            if (json.rc < 0) json.err = vgan_last_error();
        });
    } else if (!jsonfilename.empty()) {
        std::ofstream touch(jsonfilename);
    }
    vgan_graph *graph = nullptr;
    // graph.gfa when there is one, else the hcfiles' own graph.og (read natively: node sequences, path names, path membership)
    const std::string graphfile = hcfiledir + (std::ifstream(hcfiledir + "graph.gfa") ? "graph.gfa" : "graph.og");
    check(vgan_graph_load(graphfile.c_str(), hcfiledir.c_str(), &graph), "loading graph");
    pt.lap("graph load");
    stamp("graph loaded");
    vgan_graph_view gv;
    check(vgan_graph_view_get(graph, &gv), "graph view");
    std::vector<std::string> path_names;
    {
        std::string all = gv.path_names ? gv.path_names : "";
        size_t p = 0;
        while (p < all.size()) {
            size_t nl = all.find('\n', p);
            if (nl == std::string::npos) nl = all.size();
            if (nl > p) path_names.emplace_back(all, p, nl - p);
            p = nl + 1;
        }
    }
    if (path_names.size() != gv.n_paths) die("[HaploCart] graph_paths does not name every path of path_supports");

    vgan_hc_params prm;
    prm.background_error_prob = background_error_prob;
    prm.use_background_error_prob = !fastafilename.empty(); // HaploCart.cpp:397-400
    prm.is_consensus_fasta = !fastafilename.empty();
    struct Contexts { // one device context per entry of the list (an index may repeat: several contexts on one GPU)
        std::vector<vgan_hc_ctx *> v;
        ~Contexts() {
            for (auto c : v) vgan_hc_destroy(c);
        }
    } ctxs;
    struct GdStop { // (an error's unwinding: the front end's threads use the contexts -- they are joined before the contexts go)
        vgan_hc_gamrun *&r;
        ~GdStop() {
            if (r) (void)vgan_hc_gam_finish(r, nullptr, nullptr);
            r = nullptr;
        }
    } gd_stop{gd.r};
    // the contexts come up on a thread of their own (mask transposition, uploads: ~0.2 s) while this one already flattens
    // the first chunk of reads, which needs the graph only
    std::atomic<bool> contexts_up{false}; // (the lanes flatten on the host until the device can: see the chunk loop)
    struct Creator {
        std::thread t;
        std::string err;
        ~Creator() {
            if (t.joinable()) t.join();
        }
    } creator;
    creator.t = std::thread([&] {
        // (not waited for: vgan_hc_create's host work -- the mask's transposition, the tables -- runs beside the runtime's start-up and asks
        // for the device when it is done)
        if (getenv("VGAN_HC_CREATE_AFTER_RUNTIME")) warm.wait_runtime(); // (developer A/B)
        for (int d : gpu_list) {
            vgan_hc_ctx *c = nullptr;
            if (vgan_hc_create(&gv, &prm, d, &c) < 0 ||
                vgan_hc_set_mode(c, per_read ? VGAN_HC_MODE_PER_READ : VGAN_HC_MODE_NODE_WEIGHTS) < 0) {
                creator.err = std::string("[vgan] creating the device context: ") + vgan_last_error();
                if (c) vgan_hc_destroy(c);
                return;
            }
            ctxs.v.push_back(c);
        }
        stamp("device contexts ready");
        contexts_up.store(true);
    });
    std::once_flag creator_joined;
    auto contexts_ready_quiet = [&] { // from any thread
        std::call_once(creator_joined, [&] {
            if (creator.t.joinable()) creator.t.join();
        });
    };
    auto contexts_ready = [&] {
        contexts_ready_quiet();
        if (!creator.err.empty()) die(creator.err);
    };
    const size_t n_ctx = gpu_list.size();

    // reads per device batch: small enough that the first one is decoded ~30 ms after the start and every array of the loop
    // is recycled a few chunks later (the resident set, which the kernel takes apart at ~80 ms per GB when the process ends,
    // follows the chunk size), large enough for one work item of 2048 reads per flatten thread (65536 reads: ~8 ms on 32
    // threads) and a full grid on the device.  With several GPUs the batches are dealt round-robin.
    int64_t BATCH = n_ctx > 1 ? std::max<int64_t>(50000, 500000 / (int64_t)n_ctx) : 65536;
    if (const char *e = getenv("VGAN_HC_BATCH")) BATCH = std::max<int64_t>(1024, atoll(e)); // developer aid
    size_t n_chunks = 0;
    struct DedupCloser {
        vgan_dedup *d = nullptr;
        ~DedupCloser() { vgan_dedup_free(d); }
    } dedup;
    // HaploCart.cpp:389-390: remove_duplicates_internal runs whatever the input was; only its message depends on -f
    if (rmdup) check(vgan_dedup_create(&dedup.d), "duplicate removal");
    int64_t n_in = 0, n_dup = 0;
    vgan_hc_flatten_stats tot{};
    // decoded chunks -> device, in order, on a thread of its own.  The front half of the hot path -- reconstruct_graph_sequence,
    // the slicing into mappings, the segment kernel's layout -- runs ON THE DEVICE for every read whose edits are matches or
    // substitutions (vgan_hc_devflat: the parser's arrays go up as they are); only the reads it flags (indels, soft clips,
    // reads beyond the tile contract, reads the reference would terminate on) pass through the host flatten.
    // VGAN_HC_HOST_FLATTEN=1: every read through the host flatten (the lanes then flatten, as before).
    // The device contexts are ready 0.3-0.45 s after the start (the HIP runtime comes up beside a hundred busy threads); until
    // then the chunks wait here -- up to `depth` of them, after which the loop, and behind it the decoder, wait too.
    const bool host_flatten = getenv("VGAN_HC_HOST_FLATTEN") != nullptr;
    int64_t device_after = 32; // chunks (2M reads) flattened on the host before the device takes over
    if (const char *e = getenv("VGAN_HC_DEVICE_AFTER")) device_after = std::max<int64_t>(0, atoll(e)); // developer aid / tests
    struct Item {
        vgan_alnparts *chunk = nullptr;   // device flatten: the decoded chunk itself
        std::vector<uint8_t> dup;         // ... and its duplicate marks (empty: none)
        vgan_hc_host_batch *hb = nullptr; // host flatten: the flattened chunk
    };
    struct Uploader {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Item> q;
        size_t depth = 16;
        bool closed = false;
        std::string err;
        std::thread t;
        int64_t next = 0; // batches enter in input order, whichever lane made them
        static void drop(Item &it) {
            if (it.chunk) vgan_alnparts_free(it.chunk);
            if (it.hb) vgan_hc_host_batch_free(it.hb);
            it.chunk = nullptr;
            it.hb = nullptr;
        }
        void push(int64_t seq, Item &&it) {
            std::unique_lock<std::mutex> lk(mu);
            // (a decoded chunk holds ~2.7 KB of parser arrays per read, a flattened one 1.25 KB: fewer of the former may wait)
            const size_t room = it.chunk ? std::max<size_t>(2, depth / 4) : depth;
            cv.wait(lk, [&] { return (seq == next && q.size() < room) || !err.empty(); });
            if (!err.empty()) {
                lk.unlock();
                drop(it);
                return;
            }
            q.push_back(std::move(it));
            ++next;
            cv.notify_all();
        }
        void fail(const std::string &why) { // a lane that failed: the others must not wait for its turn
            std::unique_lock<std::mutex> lk(mu);
            if (err.empty()) err = why;
            cv.notify_all();
        }
        std::string error() {
            std::lock_guard<std::mutex> lk(mu);
            return err;
        }
        void close() {
            {
                std::lock_guard<std::mutex> lk(mu);
                closed = true;
            }
            cv.notify_all();
            if (t.joinable()) t.join();
        }
        ~Uploader() {
            close();
            for (auto &it : q) drop(it);
        }
    } uploader;
    if (const char *e = getenv("VGAN_HC_QUEUE")) uploader.depth = (size_t)std::max(1, atoi(e)); // developer aid
    std::mutex stat_mu;
    struct DevFlats {
        std::vector<vgan_hc_devflat *> v;
        ~DevFlats() {
            for (auto f : v) vgan_hc_devflat_free(f);
        }
    } devflats;
    double t_devflat = 0, t_hostflat_rest = 0;
    int64_t n_host_reads = 0;
    auto since_ms = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    // one chunk on the device: packed part + the other reads of a host batch
    auto hand_over = [&](vgan_hc_ctx *cx, vgan_hc_host_batch *hb) -> std::string {
        vgan_hc_batch b;        // the reads outside the tile contract (long reads, ...): the general kernel
        vgan_hc_packed_view pk; // everything else, in the segment kernel's own layout as the flatten step wrote it
        if (vgan_hc_host_batch_get(hb, &b) < 0 || vgan_hc_host_batch_get_packed(hb, &pk) < 0) return std::string("[vgan] batch: ") + vgan_last_error();
        if (vgan_hc_accumulate_packed(cx, &pk) < 0 || (b.n_reads && vgan_hc_accumulate(cx, &b) < 0)) return std::string("[vgan] accumulate: ") + vgan_last_error();
        return std::string();
    };
    uploader.t = std::thread([&] {
        std::vector<uint8_t> mask, skip2;
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(uploader.mu);
                uploader.cv.wait(lk, [&] { return !uploader.q.empty() || uploader.closed; });
                if (uploader.q.empty()) return;
                it = std::move(uploader.q.front());
                uploader.q.front() = Item();
            }
            std::string err;
            contexts_ready_quiet();
            const size_t which = n_chunks++ % std::max<size_t>(1, ctxs.v.size());
            vgan_hc_ctx *cx = creator.err.empty() && !ctxs.v.empty() ? ctxs.v[which] : nullptr;
            if (!cx) {
                err = creator.err.empty() ? "[vgan] no device context" : creator.err;
            } else if (it.hb) {
                err = hand_over(cx, it.hb);
            } else {
                if (devflats.v.size() < ctxs.v.size()) devflats.v.resize(ctxs.v.size(), nullptr);
                if (!devflats.v[which] && vgan_hc_devflat_create(cx, graph, &devflats.v[which]) < 0) err = std::string("[vgan] device flatten: ") + vgan_last_error();
                const int64_t nr = vgan_alnparts_n_reads(it.chunk);
                mask.assign((size_t)std::max<int64_t>(1, nr), 0);
                vgan_hc_packed_view pk;
                vgan_hc_flatten_stats st{};
                auto t0 = std::chrono::steady_clock::now();
                if (err.empty() && vgan_hc_devflat_run(devflats.v[which], it.chunk, it.dup.empty() ? nullptr : it.dup.data(), &pk, mask.data(), &st) < 0)
                    err = std::string("[vgan] device flatten: ") + vgan_last_error();
                if (err.empty() && vgan_hc_accumulate_packed(cx, &pk) < 0) err = std::string("[vgan] accumulate: ") + vgan_last_error();
                const double td = since_ms(t0);
                int64_t n_host = 0;
                for (int64_t r = 0; r < nr; ++r) n_host += mask[(size_t)r];
                vgan_hc_flatten_stats sh{};
                double th = 0;
                if (err.empty() && n_host) { // the reads the device left: the host's general walk, then both of its parts
                    t0 = std::chrono::steady_clock::now();
                    skip2.resize((size_t)nr);
                    for (int64_t r = 0; r < nr; ++r) skip2[(size_t)r] = mask[(size_t)r] ? 0 : 1; // (a skipped read is not flagged)
                    vgan_hc_host_batch *hb = nullptr;
                    if (vgan_hc_flatten_parts_packed(graph, it.chunk, 0, vgan_alnparts_count(it.chunk), skip2.data(), 0, &hb, &sh) < 0)
                        err = std::string("[vgan] flattening: ") + vgan_last_error();
                    else err = hand_over(cx, hb);
                    if (hb) vgan_hc_host_batch_free(hb);
                    th = since_ms(t0);
                }
                std::lock_guard<std::mutex> lk(stat_mu);
                tot.n_bad += sh.n_bad;
                tot.n_unmapped += st.n_unmapped;
                tot.n_out += st.n_out + sh.n_out;
                t_devflat += td;
                t_hostflat_rest += th;
                n_host_reads += n_host;
            }
            if (n_chunks <= 3) stamp("chunk handed to the device");
            Uploader::drop(it);
            std::lock_guard<std::mutex> lk(uploader.mu);
            uploader.q.pop_front();
            if (!err.empty() && uploader.err.empty()) uploader.err = err;
            uploader.cv.notify_all();
        }
    });
    if (device_gam) {
        contexts_ready();
        const auto t0 = std::chrono::steady_clock::now();
        vgan_hc_flatten_stats st{};
        vgan_gampipe_stats ps{};
        int rc = vgan_hc_gam_attach(gd.r, ctxs.v.data(), (int)ctxs.v.size(), graph);
        vgan_hc_gamrun *run = gd.r;
        gd.r = nullptr;
        const int rc2 = vgan_hc_gam_finish(run, &st, &ps);
        if (rc >= 0) rc = rc2;
        if (rc < 0) {
            // nothing else has been accumulated: the contexts are cleared and the host pipeline takes the file from its start
            const std::string why = vgan_last_error();
            if (!quiet) std::cerr << "[HaploCart] the device front end does not take this input (" << why << "): the host pipeline does" << std::endl;
            for (auto c : ctxs.v) check(vgan_hc_reset(c), "reset");
            device_gam = false;
            check(vgan_gam_stream_open(gamfilename.c_str(), 0, &stream.s), "reading GAM");
        } else {
            n_in = (int64_t)ps.n_reads;
            n_dup = (int64_t)ps.n_duplicates;
            tot.n_bad += st.n_bad;
            tot.n_unmapped += st.n_unmapped;
            tot.n_out += st.n_out;
            n_host_reads = (int64_t)ps.n_host_reads;
            stamp("the file's pieces inflated, framed, parsed, flattened and accumulated on the device");
            if (getenv("VGAN_TIMING"))
                fprintf(stderr, "[vgan timing] haplocart device front end: %.1f MB -> %.1f MB in %llu pieces on %zu lane(s), %llu messages, %llu reads (%llu duplicates, %llu left to the "
                                "host); %.0f ms from start to finish (%.0f after the contexts were ready; the first pieces waited %.0f ms for them); summed over pieces: upload %.0f, "
                                "inflate %.0f, framing %.0f, protobuf walk %.0f, duplicate marks %.0f, flatten + kernels + host-left reads %.0f ms; %.2f GB of device memory\n",
                        ps.compressed_bytes / 1e6, ps.inflated_bytes / 1e6, (unsigned long long)ps.n_pieces, ctxs.v.size(), (unsigned long long)ps.n_messages,
                        (unsigned long long)ps.n_reads, (unsigned long long)ps.n_duplicates, (unsigned long long)ps.n_host_reads, ps.ms_wall, since_ms(t0), ps.ms_wait_contexts,
                        ps.ms_upload, ps.ms_inflate, ps.ms_frame, ps.ms_parse, ps.ms_dedup, ps.ms_consume, ps.device_bytes / 1e9);
        }
    }
    // The loop: next chunk of decoded reads -> duplicate marks -> (host flatten ->) device queue.  Taking a chunk and marking
    // its duplicates is serial (input order); with the host flatten the flattening of several chunks runs side by side on `lanes`
    // threads (each call spreads over its own share of the host threads), and the queue takes the batches back in input order.
    double t_wait_decode = 0, t_flatten = 0, t_wait_device = 0, t_free = 0; // VGAN_TIMING: where the lanes' time went
    auto since = since_ms;
    const int want_threads = n_threads > 0 ? n_threads : (int)vgan_host_cpus();
    int lanes = std::max(1, std::min(4, want_threads / 64)); // one lane unless the machine is large and all ours
    if (const char *e = getenv("VGAN_HC_LANES")) lanes = std::max(1, std::min(16, atoi(e))); // developer aid
    const int lane_threads = std::max(1, std::min(40, want_threads / lanes));
    std::mutex take_mu;
    int64_t next_seq = 0;
    bool at_end = false;
    std::string lane_err;
    auto lane = [&] {
        try {
            std::vector<uint8_t> dup;
            for (;;) {
                vgan_alnparts *chunk = nullptr;
                int64_t seq = 0;
                const uint8_t *skip = nullptr;
                auto t0 = std::chrono::steady_clock::now();
                double waited = 0;
                {
                    std::lock_guard<std::mutex> lk(take_mu);
                    if (at_end) return;
                    check(vgan_gam_stream_next(stream.s, BATCH, &chunk), "reading GAM");
                    waited = since(t0);
                    if (!chunk) {
                        at_end = true;
                        return;
                    }
                    if (n_in == 0) stamp("first chunk of reads decoded");
                    seq = next_seq++;
                    const int64_t nr = vgan_alnparts_n_reads(chunk);
                    n_in += nr;
                    dup.clear();
                    if (dedup.d) {
                        dup.resize((size_t)nr);
                        int64_t nd = 0;
                        if (vgan_dedup_mark(dedup.d, chunk, dup.data(), &nd) < 0) {
                            vgan_alnparts_free(chunk);
                            check(-1, "duplicate removal");
                        }
                        n_dup += nd;
                        skip = dup.data();
                    }
                }
                Item it;
                double fl = 0, fr = 0;
                vgan_hc_flatten_stats st{};
                // Until the device contexts are up (the HIP runtime takes ~0.3 s to come up) the chunks are flattened here, on the
                // host's threads, as they come: a short input is done by then, a long one switches to the device flatten
                // (the device route costs a few tenths of a second once -- pinned staging, device buffers, their release when the
                // process ends -- and so starts behind the first `device_after` chunks: a 1M-read input never gets there)
                if (host_flatten || !contexts_up.load() || seq < device_after) {
                    t0 = std::chrono::steady_clock::now();
                    const int rc = vgan_hc_flatten_parts_packed(graph, chunk, 0, vgan_alnparts_count(chunk), skip, lane_threads, &it.hb, &st);
                    fl = since(t0);
                    t0 = std::chrono::steady_clock::now();
                    vgan_alnparts_free(chunk);
                    fr = since(t0);
                    check(rc, "flattening");
                } else {
                    it.chunk = chunk; // (freed by the uploader once the device has taken it)
                    it.dup = dup;
                }
                // the copy out of (pageable) host memory completes inside the uploader's calls; the kernels run asynchronously
                t0 = std::chrono::steady_clock::now();
                uploader.push(seq, std::move(it));
                const double wd = since(t0);
                std::lock_guard<std::mutex> lk(stat_mu);
                t_wait_decode += waited;
                t_flatten += fl;
                t_free += fr;
                t_wait_device += wd;
                tot.n_bad += st.n_bad;
                tot.n_unmapped += st.n_unmapped;
                tot.n_out += st.n_out;
                if (!uploader.error().empty()) die(uploader.error());
            }
        } catch (const std::exception &e) {
            uploader.fail(e.what());
            std::lock_guard<std::mutex> lk(stat_mu);
            if (lane_err.empty()) lane_err = e.what();
            std::lock_guard<std::mutex> lk2(take_mu);
            at_end = true;
        }
    };
    if (!device_gam) {
        std::vector<std::thread> lane_threads_v;
        for (int l = 1; l < lanes; ++l) lane_threads_v.emplace_back(lane);
        lane();
        for (auto &t : lane_threads_v) t.join();
    }
    if (!lane_err.empty()) die(lane_err);
    stamp("last chunk flattened");
    if (getenv("VGAN_TIMING"))
        fprintf(stderr, "[vgan timing] haplocart loop (%d lanes x %d threads, summed over the lanes): waiting for decoded reads %.0f ms, flattening %.0f ms, freeing chunks %.0f ms, waiting for the device queue %.0f ms; device flatten + hand-over %.0f ms, host flatten of the %lld reads it left %.0f ms\n",
                lanes, lane_threads, t_wait_decode, t_flatten, t_free, t_wait_device, t_devflat, (long long)n_host_reads, t_hostflat_rest);
    uploader.close();
    if (!uploader.err.empty()) die(uploader.err);
    stamp("last chunk on the device");
    contexts_ready();
    vgan_hc_ctx *ctx = ctxs.v[0];
    if (n_in == 0) die("[HaploCart] Error, no reads mapped"); // HaploCart.cpp:384-385
    int64_t n_reads = n_in - n_dup;
    if (!quiet) {
        std::cerr << "Found " << n_in << " reads." << '\n';
        if (dedup.d && fastafilename.empty()) std::cerr << "PCR duplicates removed." << std::endl;
        if (!fastafilename.empty()) std::cerr << "Using background error probability of " << background_error_prob << '\n';
        else std::cerr << "Computing haplogroup likelihoods from " << n_reads << " reads." << '\n';
    }
    if (tot.n_bad && !quiet)
        std::cerr << "[HaploCart] warning: " << tot.n_bad << " reads skipped (the reference would terminate on them)\n";
    if (tot.n_out == 0) // the reference goes on and reports path 0 from an all-zero vector: said aloud, even with -q
        std::cerr << "[HaploCart] warning: none of the " << n_in << " reads is mapped and usable; the prediction below rests on no evidence\n";
    std::vector<double> final_vec(gv.n_paths);
    if (ctxs.v.size() == 1) {
        check(vgan_hc_finalize(ctx, nullptr, final_vec.data()), "finalize");
    } else { // HaploCart.cpp:419-420 across GPUs: one reduce of the P sums (contexts no chunk reached are left out of it)
        int used_rccl = 0;
        check(vgan_hc_reduce(ctxs.v.data(), (int)ctxs.v.size(), final_vec.data(), &used_rccl), "reduce");
        if (!quiet) {
            char why[256] = "";
            (void)vgan_hc_reduce_why(why, sizeof why);
            std::cerr << "Reduced the log-likelihoods of " << ctxs.v.size() << " device contexts (" << (used_rccl ? "RCCL" : "host") << (why[0] ? ": " : "")
                      << why << ")." << '\n';
            double setup_ms = 0, reduce_ms = 0;
            int n_setups = 0;
            (void)vgan_hc_reduce_last(&reduce_ms, nullptr);
            if (used_rccl && vgan_hc_reduce_info(&setup_ms, &n_setups) == 0 && n_setups)
                std::cerr << "RCCL communicator over " << ctxs.v.size() << " devices set up in " << setup_ms << " ms; the reduce took " << reduce_ms
                          << " ms in all (VGAN_HC_REDUCE=host sums on the host instead)." << '\n';
        }
    }
    pt.lap("flatten + kernels");
    const int maxh = vgan_hc_argmax(final_vec.data(), gv.n_paths); // HaploCart.cpp:423
    const std::string predicted = path_names[(size_t)maxh];

    if (!fastafilename.empty()) n_reads = 1; // HaploCart.cpp:427
    std::replace(samplename.begin(), samplename.end(), ' ', '_');
    {
        std::ofstream out(outputfilename, std::ios::app);
        if (outputfilename == "/dev/stdout" && !quiet) out << "\n\n";
        out << "#sample\tpredicted haplogroup\treads" << std::endl; // HaploCart.cpp:437-438
        out << samplename << '\t' << predicted << '\t' << n_reads << std::endl;
    }
    if (compute_posteriors) { // get_posterior.cpp:4-33 (non-webapp branch)
        std::vector<char> clades(1 << 20);
        std::vector<double> conf(8192);
        const int n = vgan_hc_posterior(ctx, final_vec.data(), predicted.c_str(), clades.data(), (int64_t)clades.size(),
                                        conf.data(), (int32_t)conf.size());
        check(n, "posterior");
        std::ofstream pf(posteriorfilename, std::ios::app);
        pf << "\nClade-level posterior confidence values\n" << samplename << '\t';
        const char *p = clades.data();
        for (int i = 0; i < n; ++i) {
            const char *nl = strchr(p, '\n');
            pf << std::string(p, nl ? (size_t)(nl - p) : strlen(p)) << '\t' << conf[(size_t)i] << '\t' << i << '\t';
            p = nl ? nl + 1 : p + strlen(p);
        }
        pf << "\n" << std::endl;
    }
    if (debug) { // HaploCart.cpp:464-479: names sorted by descending log-likelihood
        std::vector<int> idx(final_vec.size());
        std::iota(idx.begin(), idx.end(), 0);
        std::sort(idx.begin(), idx.end(), [&](int A, int B) { return final_vec[(size_t)A] > final_vec[(size_t)B]; });
        const std::string dbg = (outputfilename == "/dev/stdout" ? std::string("haplocart") : outputfilename) + ".loglik.tsv";
        std::ofstream d(dbg);
        d.precision(10);
        for (int k : idx) d << path_names[(size_t)k] << '\t' << final_vec[(size_t)k] << '\n';
        if (!quiet) std::cerr << "Writing log likelihoods to " << dbg << std::endl;
    }
    pt.lap("posterior + output");
    stamp("output written");
    if (getenv("VGAN_TIMING")) {
        int64_t acc[4] = {0, 0, 0, 0};
        vgan_host_cpu_account(acc);
        timespec pts;
        clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &pts);
        fprintf(stderr, "[vgan timing] haplocart CPU: inflate %.0f ms, frame + parse %.0f ms, flatten %.0f ms, merge (caller) %.0f ms; the process %.0f ms; %d processors usable\n",
                acc[0] * 1e-3, acc[1] * 1e-3, acc[2] * 1e-3, acc[3] * 1e-3, pts.tv_sec * 1e3 + pts.tv_nsec * 1e-6, vgan_host_cpus());
    }
    if (getenv("VGAN_TIMING")) { // resident anonymous memory the kernel has to take apart when the process ends
        if (FILE *f = fopen("/proc/self/status", "r")) {
            char line[256];
            while (fgets(line, sizeof line, f))
                if (!strncmp(line, "VmHWM:", 6) || !strncmp(line, "RssAnon:", 8) || !strncmp(line, "RssFile:", 8) || !strncmp(line, "RssShmem:", 9))
                    fprintf(stderr, "[vgan timing] haplocart memory: %s", line);
            fclose(f);
        }
        if (FILE *f = fopen("/proc/self/smaps_rollup", "r")) {
            char line[256];
            while (fgets(line, sizeof line, f))
                if (!strncmp(line, "Rss:", 4) || !strncmp(line, "AnonHugePages:", 14) || !strncmp(line, "Anonymous:", 10))
                    fprintf(stderr, "[vgan timing] haplocart memory: %s", line);
            fclose(f);
        }
    }
    if (json.t.joinable()) json.t.join();
    if (json.rc < 0) die("[HaploCart] writing " + jsonfilename + ": " + json.err);
    // Every output is written and closed: leave from here.  Returning runs this function's destructors first -- the stream's
    // inflated bytes (1.6 GB per million reads), the recycled arrays, the graph, the device contexts: hundreds of munmap calls,
    // each interrupting every core the process ran on -- which cost 0.2-0.3 s that no one is waiting for; the kernel takes
    // the address space apart in one pass either way (0.3-0.4 s for the ~6 GB a million reads leave resident; handing the
    // recycled arrays back on 32 threads first took 0.25 s and saved 0.15 s of it).  (VGAN_KEEP_TEARDOWN=1: return normally, for leak checkers.)
    if (!getenv("VGAN_KEEP_TEARDOWN")) {
        if (getenv("VGAN_TIMING"))
            fprintf(stderr, "[vgan timing] wall clock at exit: %.6f\n",
                    std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
        EarlyLeave::get().finish(0);
    }
    pt.lap("teardown");
    return 0;
}

std::string usage() {
    return "vgan (MI355X build): per-read likelihood hot path on the GPU\n\n"
           "   vgan haplocart   mitochondrial haplogroup prediction (see: vgan haplocart -h)\n"
           "   vgan euka        abundance estimation of eukaryotic taxa (see: vgan euka -h)\n"
           "   vgan soibean     sources of one taxon and their proportions (see: vgan soibean -h)\n"
           "   vgan version\n\n"
           "The CPU-only subcommands of the reference are not part of this build.\n";
}

} // namespace

int main(int argc, char **argv) {
    // The device front end keeps a dozen streams busy (three slots of a lane: a stream each and four for a piece's parts); the runtime
    // spreads a process's streams over four hardware queues by default, where a piece's short kernels wait behind another piece's
    // one-lane framing walks (2 ms each): with eight, 4 M reads take 235 ms instead of 285.  (Set before the runtime starts; a value
    // in the environment is kept.)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    if (getenv("VGAN_TIMING"))
        fprintf(stderr, "[vgan timing] wall clock at main: %.6f\n", std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
    try {
        if (argc < 2) {
            std::cerr << usage();
            return 1;
        }
        const std::string cmd = argv[1];
        if (cmd == "haplocart" || cmd == "euka" || cmd == "soibean") {
            EarlyLeave::get().start(); // (the parent of the working process leaves from inside this call: cli_util.h)
            int rc;
            try {
                rc = cmd == "haplocart" ? haplocart(argc - 1, argv + 1) : cmd == "euka" ? euka_main(argc - 1, argv + 1) : soibean_main(argc - 1, argv + 1);
            } catch (const std::exception &e) {
                std::cerr << e.what() << std::endl;
                rc = 1;
            }
            // every output file is closed by now: leave without running the exit handlers (the HIP runtime's teardown and the
            // page-by-page release of gigabytes of alignments cost ~0.2 s that no one is waiting for)
            if (getenv("VGAN_KEEP_TEARDOWN")) return rc; // (leak checkers, profilers: their exit handlers run)
            EarlyLeave::get().finish(rc);
        }
        if (cmd == "version") {
            std::cout << "vgan-mi355x ABI " << vgan_abi_version() << std::endl;
            return 0;
        }
        std::cerr << usage();
        return 1;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
}
