// vgan euka -- the reference's subcommand surface (src/Euka.cpp:144-1185) over the GPU per-read pass.
//
//   vgan euka -g reads.gam --euka_dir DIR [--dbprefix euka_db] [--deam5p F --deam3p F] [--no-mcmc] [--iter N] [--burnin N]
//             [--entropy X] [--minBins N] [--maxBins N] [--minMQ N] [--minFrag N] [--outFrag] [--outGroup NAME] [-l N]
//             [-o PREFIX] [--out_dir DIR] [-t N] [--seed N] [--device N]
//
// Same flags, defaults, validation and output files as Euka::run.  What differs, and why:
//   * the graph is read from <dbprefix>.gfa, or <dbprefix>.og by this build's own ODGI reader; a <dbprefix>.gbwt beside the
//     .og is loaded and walked as readOG_Euka.h:36-74 does (its result is not used further there either);
//   * FASTQ input needs vg giraffe in-process (src/map_giraffe.cpp): map with vg and pass the GAM with -g;
//   * readGAM3's per-alignment lambda runs on the GPU (vgan_euka_*), the abundance MCMC in closed form on the host
//     (vgan_euka_report); --seed N makes the chain reproducible (default 0 = std::random_device, as the reference).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <iostream>
#include <thread>
#include <vector>

#include "cli_util.h"

using namespace vgan_cli;

namespace {

std::string euka_usage() {
    return "\n vgan euka [options]\n\n"
           " Abundance estimation of eukaryotic taxa from an environmental DNA sample, per-read pass on the GPU (MI355X).\n\n"
           " Input:\n"
           "   --euka_dir [STR]  euka database location (default: ../share/vgan/euka_dir/)\n"
           "   --dbprefix [STR]  database prefix (default: euka_db): <prefix>.gfa, <prefix>.clade, <prefix>.bins\n"
           "   -g [STR]          GAM input (FASTQ input needs vg giraffe: map first)\n"
           "   -o [STR]          output file prefix (default: euka_output)\n"
           "   -t [INT]          host threads (-1 for all available)\n"
           " Filter options:\n"
           "   --minMQ [INT]     mapping quality minimum for a fragment (default: 29)\n"
           "   --minFrag [INT]   minimum number of fragments per taxon (default: 10)\n"
           "   --entropy [FLOAT] minimum entropy score of a bin (default: 1.17)\n"
           "   --minBins [INT]   minimum number of bins above the entropy threshold (default: 6)\n"
           "   --maxBins [INT]   maximum number of empty bins (default: 0)\n"
           " Damage options:\n"
           "   --deam5p [STR]    5' substitution profile (default: none)\n"
           "   --deam3p [STR]    3' substitution profile (default: none)\n"
           "   -l [INT]          length of the damage profiles written per taxon (default: 5)\n"
           "   --out_dir [STR]   directory created for the profiles when missing\n"
           " MCMC options:\n"
           "   --no-mcmc         report the initial abundance estimates only\n"
           "   --iter [INT]      iterations (default: 10000)\n"
           "   --burnin [INT]    burn-in (default: 100)\n"
           "   --seed [INT]      reproducible chain (default 0: std::random_device)\n"
           " Output options:\n"
           "   --outFrag         write the names of the fragments of every detected taxon\n"
           "   --outGroup [STR]  always write coverage, fragment lengths and profile of this taxon\n"
           "   --device [INT]    GPU index (default 0)\n"
           "   --gpus [LIST]     GPU indices, comma separated (default: the one of --device; VGAN_GPUS in the environment: a list or `all`):\n"
           "                     the fragments are dealt to one device context per entry, the per-clade tables are summed\n";
}

} // namespace

int euka_main(int argc, char **argv) {
    const char *T = "[euka]";
    std::string euka_dir = "../share/vgan/euka_dir/", dbprefix = "euka_db", gam, fq1, fq2, out_prefix = "euka_output";
    std::string deam5, deam3, out_group, out_dir;
    bool interleaved = false, run_mcmc = true, out_frag = false;
    int n_threads = 1, iter = 10000, burnin = 100, ltp = 5, device = 0; // Euka.cpp:171-190
    std::vector<int> gpu_list;
    int min_bins = 6, min_reads = 10, min_mq = 29, max_bins = 0;
    double entropy = 1.17;
    uint64_t seed = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char *flag) -> std::string {
            if (i + 1 >= argc) die(std::string("[euka] Error, option ") + flag + " needs a value");
            return argv[++i];
        };
        auto non_negative = [&](const char *flag) {
            const int v = parse_int(need(flag), flag, T);
            if (v < 0) die(std::string("[euka] Error, option ") + flag + " must not be negative"); // assert(... >= 0)
            return v;
        };
        if (a == "-h" || a == "--help" || a == "-") {
            std::cerr << euka_usage() << std::endl;
            return 0;
        } else if (a == "--euka_dir") {
            euka_dir = need("--euka_dir");
            if (euka_dir.empty() || euka_dir.back() != '/') euka_dir += '/';
        } else if (a == "--dbprefix") dbprefix = need("--dbprefix");
        else if (a == "-fq1") reject_fasta(fq1 = need("-fq1"), T);
        else if (a == "-fq2") reject_fasta(fq2 = need("-fq2"), T);
        else if (a == "-i") {
            interleaved = true;
            if (!fq2.empty()) die("[euka] If interleaved option chosen, Euka expects only one FASTQ file");
        } else if (a == "-g") gam = need("-g");
        else if (a == "-M") (void)need("-M"); // minimizer index prefix: only used by giraffe
        else if (a == "--deam5p") deam5 = need("--deam5p");
        else if (a == "--deam3p") deam3 = need("--deam3p");
        else if (a == "--no-mcmc") run_mcmc = false;
        else if (a == "--iter") iter = non_negative("--iter");
        else if (a == "--burnin") burnin = non_negative("--burnin");
        else if (a == "--entropy") {
            entropy = parse_double(need("--entropy"), "--entropy", T);
            if (entropy < 0) die("[euka] Error, option --entropy must not be negative");
            if (entropy > 5.0) die("[euka] Error, entropy thresold is too stringent"); // Euka.cpp:267
        } else if (a == "--minBins") {
            min_bins = non_negative("--minBins");
            if (min_bins > 20) die("[euka] Error, minimum number of bins exceeds the total number of bins");
        } else if (a == "--maxBins") {
            max_bins = non_negative("--maxBins");
            if (max_bins > 20) die("[euka] Error, maximum number of bins exceeds the total number of bins");
        } else if (a == "--minMQ") {
            min_mq = non_negative("--minMQ");
            if (min_mq > 60) die("[euka] Error, option --minMQ must lie in 0..60");
        } else if (a == "--minFrag") min_reads = non_negative("--minFrag");
        else if (a == "--outFrag") out_frag = true;
        else if (a == "--outGroup") out_group = need("--outGroup");
        else if (a == "-S") die("[euka] SAFARI minimizer mode belongs to the giraffe mapping step, which is not part of the GPU path");
        else if (a == "-t") {
            n_threads = parse_int(need("-t"), "-t", T);
            if (n_threads < -1 || n_threads == 0) die("[euka] Error, invalid number of threads"); // Euka.cpp:298
            const int hw = (int)vgan_host_cpus();
            if (n_threads == -1) n_threads = hw; // (host threads only, as in the reference: GPUs are asked for with --gpus)
            else if (n_threads > hw) {
                std::cerr << "[euka] Warning, specified number of threads is greater than the number available. Using " << hw << " threads\n";
                n_threads = hw;
            }
        } else if (a == "-o") out_prefix = need("-o");
        else if (a == "-z") (void)need("-z");
        else if (a == "-l") {
            ltp = parse_int(need("-l"), "-l", T);
            if (ltp < 0 || ltp > 32) die("[euka] Error, option -l must lie in 0..32");
        } else if (a == "--out_dir") out_dir = need("--out_dir");
        else if (a == "--seed") seed = (uint64_t)std::strtoull(need("--seed").c_str(), nullptr, 10);
        else if (a == "--device") {
            device = parse_int(need("--device"), "--device", T);
            if (device < 0) die("[euka] Error, --device needs a non-negative GPU index");
        } else if (a == "--gpus") { // GPU indices, comma separated: one device context (and host thread) each
            const std::string v = need("--gpus");
            size_t p0 = 0;
            while (p0 <= v.size()) {
                size_t c1 = v.find(',', p0);
                if (c1 == std::string::npos) c1 = v.size();
                const int d = parse_int(v.substr(p0, c1 - p0), "--gpus", T);
                if (d < 0) die("[euka] Error, --gpus needs non-negative GPU indices");
                gpu_list.push_back(d);
                p0 = c1 + 1;
            }
        } else die("[euka] Error, unrecognized option " + a);
    }
    (void)interleaved;
    if (!fq1.empty() || !fq2.empty())
        die("[euka] FASTQ input needs vg giraffe in-process, which this build does not have; map with vg and pass -g");
    const std::string prefix = euka_dir + dbprefix;
    const std::string graph_ext = is_file(prefix + ".gfa") ? ".gfa" : ".og"; // Euka.cpp:373-384 (.og and .gbwt there)
    for (const std::string &ext : {graph_ext, std::string(".clade"), std::string(".bins")})
        if (!is_file(prefix + ext)) die(prefix + ext + " does not exist.");
    if (gam.empty()) die("[euka] Error, no input file given (use -g)");
    if (!is_readable_input(gam)) die("[euka] Error, GAM input file " + gam + " does not exist");
    if (run_mcmc && iter - burnin - 1 <= 0) die("[euka] Error, --iter must exceed --burnin + 1");

    PhaseTimer pt("euka");
    // which GPUs: --gpus LIST, the environment's VGAN_GPUS (`all` or a list), or the one of --device
    if (gpu_list.empty()) {
        const char *e = getenv("VGAN_GPUS");
        if (e && std::string(e) == "all") {
            const int n_visible = vgan_device_count();
            for (int d = 0; d < n_visible; ++d) gpu_list.push_back(d);
        } else if (e && *e) {
            const std::string v = e;
            size_t p0 = 0;
            while (p0 <= v.size()) {
                size_t c1 = v.find(',', p0);
                if (c1 == std::string::npos) c1 = v.size();
                const int d = parse_int(v.substr(p0, c1 - p0), "VGAN_GPUS", T);
                if (d < 0) die("[euka] Error, VGAN_GPUS needs non-negative GPU indices");
                gpu_list.push_back(d);
                p0 = c1 + 1;
            }
        }
        if (gpu_list.empty()) gpu_list.push_back(device);
    }
    // A long BGZF input: the front end runs ON THE DEVICE (vgan_euka_gam_*: the file in pieces through inflate, framing, protobuf walk
    // and euka's flatten as kernels, piece i to context i mod n; csrc/gam_pipe.hip, euka_flatten_kernels.hip); the host parses only the
    // reads the device flatten leaves (indels, soft clips).  VGAN_EUKA_DEVICE_GAM=0 / 1: never / whenever the input is a regular file.
    // --outFrag needs the reads' names, which the device does not keep: the host pipeline's.  Anything the device refuses goes through
    // the host pipeline as well.
    struct FileMap {
        const uint8_t *p = nullptr;
        size_t n = 0;
        ~FileMap() {
            if (p) munmap(const_cast<uint8_t *>(p), n);
        }
    } gam_map;
    struct GdRun {
        vgan_euka_gamrun *r = nullptr;
        ~GdRun() { vgan_euka_gam_free(r); }
    };
    bool device_gam = false;
    {
        const char *e = getenv("VGAN_EUKA_DEVICE_GAM");
        struct stat sb;
        if (!(e && e[0] == '0') && !out_frag && stat(gam.c_str(), &sb) == 0 && S_ISREG(sb.st_mode) && ((e && e[0] == '1') || (uint64_t)sb.st_size >= (128ull << 20)) &&
            sb.st_size > 28) {
            const int fd = open(gam.c_str(), O_RDONLY);
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
    DeviceWarm warm; // (the runtime and the run's code objects, beside the tables' loading)
    warm.start(gpu_list.empty() ? 0 : gpu_list[0], VGAN_PRELOAD_EUKA | (device_gam ? VGAN_PRELOAD_GAM : 0u));
    GamReader reader; // unmapped reads are kept so that they are counted
    GdRun gd; // (started now: upload, inflate, framing and parse need neither tables nor contexts and run beside their set-up)
    if (device_gam) {
        vgan_gampipe_opts po{};
        po.n_threads = n_threads;
        if (vgan_euka_gam_start(gpu_list.data(), (int)gpu_list.size(), gam_map.p, gam_map.n, &po, &gd.r) < 0) device_gam = false;
    }
    if (!device_gam) reader.start(gam, 1);
    Handle<vgan_damage> dmg(vgan_damage_free);
    check(vgan_damage_load(deam5.empty() ? nullptr : deam5.c_str(), deam3.empty() ? nullptr : deam3.c_str(), &dmg.p), "damage profiles");
    std::cerr << "Reading in taxa information ..." << std::endl;
    Handle<vgan_euka_db> db(vgan_euka_db_free);
    check(vgan_euka_db_load((prefix + ".clade").c_str(), (prefix + ".bins").c_str(), &db.p), "clade / bins tables");
    vgan_euka_db_view dv;
    check(vgan_euka_db_view_get(db.p, &dv), "clade table");
    if (dv.n_clades == 0) die("Error: The clade vector is empty. Unable to proceed. Check if the soibean.clade file is not empty.");
    if (dv.bin_off[dv.n_clades] == 0) die("Bins file is empty unable to proceed");
    if (!out_group.empty()) { // Euka.cpp:419-427
        bool found = false;
        const char *p = dv.clade_names;
        for (uint32_t c = 0; c < dv.n_clades && p; ++c) {
            const char *q = strchr(p, '\n');
            found = found || std::string(p, q ? (size_t)(q - p) : strlen(p)) == out_group;
            p = q ? q + 1 : nullptr;
        }
        if (!found) die("[euka] Outgroup not found in reference graph");
    }
    std::cerr << "Reading in variation graph ..." << std::endl;
    Handle<vgan_graph> graph(vgan_graph_free);
    check(vgan_graph_load((prefix + graph_ext).c_str(), nullptr, &graph.p), "loading graph");
    if (graph_ext == ".og") { // Euka.cpp:372,419: the haplotype index beside the graph
        vgan_graph_view ggv;
        check(vgan_graph_view_get(graph.p, &ggv), "graph view");
        check_gbwt_beside(prefix, ggv.max_id - ggv.min_id + 1, ggv.n_paths, "[euka]");
    }
    pt.lap("tables + graph");

    if (vgan_device_count() <= 0) die("[euka] no HIP device is visible: the per-read likelihood pass runs on the GPU only");
    vgan_damage_view dmv;
    check(vgan_damage_view_get(dmg.p, &dmv), "damage view");
    vgan_euka_params prm;
    prm.min_mapq = (uint32_t)min_mq;
    prm.length_to_prof = ltp;
    struct Contexts {
        std::vector<vgan_euka_ctx *> v;
        ~Contexts() {
            for (auto c : v) vgan_euka_destroy(c);
        }
    } ctxs;
    struct GdStop { // (an error's unwinding: the front end's threads use the contexts -- they are joined before the contexts go)
        vgan_euka_gamrun *&r;
        ~GdStop() {
            vgan_euka_gam_free(r);
            r = nullptr;
        }
    } gd_stop{gd.r};
    for (int d : gpu_list) {
        vgan_euka_ctx *c = nullptr;
        check(vgan_euka_create(&dv, &dmv, &prm, d, &c), "creating the device context");
        ctxs.v.push_back(c);
    }
    pt.lap("device context");

    std::cerr << "Estimating clades: Please be patient! Depending on the size of your input file, this process can take some time." << std::endl;
    // per processed read, in input order
    std::vector<int32_t> read_clade;
    std::vector<uint8_t> read_pass;
    std::vector<uint16_t> read_len;
    std::vector<int64_t> name_off{0};
    std::string names;
    int64_t n_mapped = 0, n_bad = 0, n_in_file = 0;
    const size_t K = ctxs.v.size();
    if (device_gam) {
        vgan_euka_gam_result gr{};
        vgan_gampipe_stats ps{};
        int rc = vgan_euka_gam_attach(gd.r, ctxs.v.data(), (int)K, graph.p);
        if (rc >= 0) rc = vgan_euka_gam_finish(gd.r, &gr, &ps);
        if (rc < 0) { // nothing else has been accumulated: the contexts are cleared and the host pipeline takes the file from its start
            std::cerr << "[euka] the device front end does not take this input (" << vgan_last_error() << "): the host pipeline does" << std::endl;
            for (auto c : ctxs.v) check(vgan_euka_reset(c), "reset");
            device_gam = false;
            reader.start(gam, 1);
        } else {
            read_clade.assign(gr.read_clade, gr.read_clade + gr.n_reads);
            read_pass.assign(gr.read_pass, gr.read_pass + gr.n_reads);
            read_len.assign(gr.read_seq_len, gr.read_seq_len + gr.n_reads);
            n_mapped = gr.n_mapped;
            n_bad = gr.n_bad;
            n_in_file = gr.n_messages;
            if (getenv("VGAN_TIMING"))
                fprintf(stderr, "[vgan timing] euka device front end: %.1f MB -> %.1f MB in %llu pieces on %zu lane(s), %llu messages, %llu mapped reads (%llu left to the host); %.0f ms "
                                "from start to finish; summed over pieces: upload %.0f, inflate %.0f, framing %.0f, protobuf walk %.0f, flatten + kernel %.0f ms; %.2f GB of device memory\n",
                        ps.compressed_bytes / 1e6, ps.inflated_bytes / 1e6, (unsigned long long)ps.n_pieces, K, (unsigned long long)ps.n_messages, (unsigned long long)ps.n_reads,
                        (unsigned long long)ps.n_host_reads, ps.ms_wall, ps.ms_upload, ps.ms_inflate, ps.ms_frame, ps.ms_parse, ps.ms_consume, ps.device_bytes / 1e9);
        }
    }
    if (!device_gam) {
    Handle<vgan_alnset> aln(vgan_aln_free);
    aln.p = reader.take();
    vgan_alnset_view av;
    check(vgan_aln_view_get(aln.p, &av), "alignment view");
    n_in_file = av.n_reads;
    pt.lap("GAM decode");

    // Batches of fragments in input order; with several contexts every context has a host thread of its own taking the
    // next batch (vgan_euka_accumulate returns the per-read results, so it is synchronous), and the per-read results go
    // back in batch order: readGAM3 hands them to the abundance chain per read (MCMC.cpp:1192-1193).
    const int64_t BATCH = K > 1 ? std::max<int64_t>(100000, 1000000 / (int64_t)K) : 1000000;
    const int64_t n_batches = (av.n_reads + BATCH - 1) / BATCH;
    struct BatchOut {
        std::vector<int32_t> clade;
        std::vector<uint8_t> pass;
        std::vector<uint16_t> len;
        std::vector<int64_t> src;
        int64_t n_mapped = 0, n_bad = 0;
        std::string err;
    };
    std::vector<BatchOut> outs((size_t)n_batches);
    std::atomic<int64_t> next_batch{0};
    const int flat_threads = std::max(1, n_threads / (int)K);
    auto worker = [&](size_t ci) {
        std::vector<int32_t> o_clade;
        std::vector<double> o_d;
        std::vector<uint8_t> o_pass;
        for (;;) {
            const int64_t bi = next_batch.fetch_add(1);
            if (bi >= n_batches) break;
            BatchOut &o = outs[(size_t)bi];
            const int64_t r0 = bi * BATCH, r1 = std::min(av.n_reads, r0 + BATCH);
            vgan_euka_host_batch *hbp = nullptr;
            vgan_euka_flatten_stats st{};
            if (vgan_euka_flatten(graph.p, aln.p, r0, r1, flat_threads, &hbp, &st) < 0) {
                o.err = std::string("flattening: ") + vgan_last_error();
                break;
            }
            Handle<vgan_euka_host_batch> hb(vgan_euka_host_batch_free);
            hb.p = hbp;
            o.n_mapped = st.n_in - st.n_unmapped;
            o.n_bad = st.n_bad;
            vgan_euka_batch b;
            if (vgan_euka_host_batch_get(hb.p, &b) < 0) {
                o.err = std::string("batch: ") + vgan_last_error();
                break;
            }
            if (b.n_reads == 0) continue;
            const size_t R = b.n_reads;
            o_clade.resize(R);
            o_d.resize(4 * R);
            o_pass.resize(R);
            vgan_euka_read_out out{o_clade.data(), o_d.data(), o_d.data() + R, o_d.data() + 2 * R, o_d.data() + 3 * R, o_pass.data()};
            if (vgan_euka_accumulate(ctxs.v[ci], &b, &out) < 0) {
                o.err = std::string("accumulate: ") + vgan_last_error();
                break;
            }
            // the batch is in node order (vgan_euka_flatten): the per-read lists go on in input order, as the reference's do
            std::vector<int64_t> pos((size_t)(r1 - r0), -1);
            for (size_t i = 0; i < R; ++i) pos[(size_t)((int64_t)b.read_src[i] - r0)] = (int64_t)i;
            o.clade.clear();
            o.pass.clear();
            o.len.clear();
            o.src.clear();
            for (int64_t i : pos) {
                if (i < 0) continue;
                o.clade.push_back(o_clade[(size_t)i]);
                o.pass.push_back(o_pass[(size_t)i]);
                o.len.push_back(b.read_seq_len[i]);
                if (out_frag) o.src.push_back(b.read_src[i]);
            }
        }
    };
    if (K == 1) {
        worker(0);
    } else {
        std::vector<std::thread> th;
        for (size_t ci = 0; ci < K; ++ci) th.emplace_back(worker, ci);
        for (auto &t : th) t.join();
    }
    for (const BatchOut &o : outs) {
        if (!o.err.empty()) die("[vgan] " + o.err);
        n_mapped += o.n_mapped;
        n_bad += o.n_bad;
        read_clade.insert(read_clade.end(), o.clade.begin(), o.clade.end());
        read_pass.insert(read_pass.end(), o.pass.begin(), o.pass.end());
        read_len.insert(read_len.end(), o.len.begin(), o.len.end());
        if (out_frag)
            for (int64_t src : o.src) {
                names.append(av.name + av.name_off[src], (size_t)(av.name_off[src + 1] - av.name_off[src]));
                name_off.push_back((int64_t)names.size());
            }
    }
    } // (the host pipeline)
    std::vector<int32_t> clade_count(dv.n_clades);
    std::vector<uint32_t> baseshift((size_t)dv.n_clades * 2 * std::max(ltp, 1) * 16);
    std::vector<double> bin_cov(dv.bin_off[dv.n_clades]), sum_log_like(dv.n_clades);
    std::vector<int64_t> n_like(dv.n_clades);
    int64_t n_bad_dev = 0;
    check(vgan_euka_reduce(ctxs.v.data(), (int)K, clade_count.data(), baseshift.data(), bin_cov.data(), n_like.data(), sum_log_like.data(),
                           &n_bad_dev), "finalize");
    if (K > 1) std::cerr << "Summed the per-clade tables of " << K << " device contexts." << std::endl;
    pt.lap("flatten + kernels");
    std::cerr << " .. done!" << std::endl;
    int64_t passed = 0;
    for (int32_t c : clade_count) passed += c;
    std::cerr << "Number of fragments in input file: " << n_in_file << std::endl; // readGAM_Euka.h:635-637
    std::cerr << "Number of mapped fragments: " << n_mapped << std::endl;
    std::cerr << "Number of fragments after filtering: " << passed << std::endl;
    if (n_bad + n_bad_dev)
        std::cerr << "[euka] warning: " << n_bad + n_bad_dev << " fragments skipped (the reference would index out of bounds on them)\n";

    vgan_euka_results res{};
    res.db = &dv;
    res.clade_count = clade_count.data();
    res.baseshift = baseshift.data();
    res.bin_cov = bin_cov.data();
    res.n_like = n_like.data();
    res.sum_log_like = sum_log_like.data();
    res.n_reads = (int64_t)read_clade.size();
    res.read_clade = read_clade.data();
    res.read_pass = read_pass.data();
    res.read_seq_len = read_len.data();
    res.name_off = out_frag ? name_off.data() : nullptr;
    res.names = out_frag ? names.data() : nullptr;
    vgan_euka_report_cfg cfg{};
    cfg.detect.min_bins = (uint32_t)min_bins;
    cfg.detect.min_reads = (uint32_t)min_reads;
    cfg.detect.max_zero_bins = max_bins;
    cfg.detect.entropy_threshold = entropy;
    cfg.length_to_prof = ltp;
    cfg.run_mcmc = run_mcmc;
    cfg.iter = iter;
    cfg.burnin = burnin;
    cfg.seed = seed;
    cfg.out_frag = out_frag;
    cfg.out_group = out_group.empty() ? nullptr : out_group.c_str();
    cfg.out_dir = out_dir.empty() ? nullptr : out_dir.c_str();
    int32_t n_detected = 0;
    if (run_mcmc) std::cerr << "Computing MCMC:" << std::endl;
    check(vgan_euka_report(&res, &cfg, out_prefix.c_str(), nullptr, &n_detected, nullptr), "writing the output files");
    pt.lap("abundance + output");
    std::cerr << "Abundance estimation completed! " << std::endl << '\n';
    if (!run_mcmc || n_detected < 2)
        std::cerr << "No MCMC was computed. It was either specified by the user or less than 2 groups were present in the sample." << std::endl;
    const std::string base = out_prefix.substr(out_prefix.find_last_of('/') + 1);
    std::cerr << "You can find all four output files (" << base << "_abundance.tsv, " << base << "_detected.tsv, " << base
              << "_coverage.tsv, and the damage profiles) in your current working directory!" << std::endl;
    return 0;
}
