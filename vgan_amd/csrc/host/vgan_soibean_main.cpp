// vgan soibean -- the reference's subcommand surface (src/soibean.cpp:205-947) over the GPU likelihood path.
//
//   vgan soibean -g reads.gam --dbprefix TAXON [--soibean_dir DIR] [--tree_dir DIR] [--deam5p F --deam3p F] [-k N] [--randStart]
//                [--iter N] [--burnin N] [--chains N] [--no-mcmc] [-P N] [-o PREFIX] [-t N] [--seed N] [--device N]
//
// Same flags, defaults, validation and output files as soibean::run.  What differs, and why:
//   * the graph is read from <soibean_dir>/<dbprefix>.gfa or .og (one embedded path per tree node; the reference takes the
//     paths from the .gbwt, which this build does not read); FASTQ input needs vg giraffe in-process: map with vg and pass -g;
//   * analyse_GAM, the initial estimate and every likelihood refresh of the chains run on the GPU (vgan_sb_*); the chain
//     itself is host control flow (vgan_sb_estimate); --seed N makes it reproducible (0 = std::random_device, as there).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <random>
#include <sstream>
#include <thread>
#include <vector>

#include "cli_util.h"

using namespace vgan_cli;

namespace {

std::string soibean_usage() {
    return "\n vgan soibean [options]\n\n"
           " Identification and abundance estimation of the sources of a taxon in an environmental DNA sample; analyse_GAM,\n"
           " the initial estimate and the likelihood of every MCMC iteration run on the GPU (MI355X).\n\n"
           " Input:\n"
           "   --soibean_dir [STR]  database location (default: ../share/vgan/soibean_dir/)\n"
           "   --tree_dir [STR]     location of <dbprefix>.new.dnd (default: <soibean_dir>/tree_dir/)\n"
           "   --dbprefix [STR]     taxon of interest (required): <dbprefix>.gfa, <dbprefix>.new.dnd, soibean_db.baseFreq\n"
           "   -g [STR]             GAM input (FASTQ input needs vg giraffe: map first)\n"
           "   -o [STR]             output prefix (default: beanOut)\n"
           "   -t [INT]             host threads (-1 for all available)\n"
           " Damage options:\n"
           "   --deam5p / --deam3p [STR]  substitution profiles (default: none)\n"
           " MCMC options:\n"
           "   --no-mcmc            initial estimate only\n"
           "   --iter [INT]         iterations (default: 500000)\n"
           "   --burnin [INT]       burn-in (default: 75000)\n"
           "   --chains [INT]       chains (default: 4)\n"
           "   --randStart          random starting nodes\n"
           "   -k [INT]             number of sources, random start (default: estimated)\n"
           "   -P [INT]             mismatch penalty period for unsupported bases (default: 7)\n"
           "   --seed [INT]         reproducible chains (default 0: std::random_device)\n"
           "   --device [INT]       GPU index (default 0)\n"
           "   --gpus [LIST]        GPU indices, comma separated (default: the one of --device; VGAN_GPUS in the environment: a list or\n"
           "                        `all`): the reads are dealt once to one device context per GPU, every likelihood refresh runs\n"
           "                        on all of them and their sums are added (MCMC.cpp:739's reduction over the reads); the chain\n"
           "                        files are those of one GPU, byte for byte\n";
}

std::vector<std::string> lines_of(const char *joined, size_t n) {
    std::vector<std::string> out;
    std::istringstream s(joined ? joined : "");
    std::string l;
    while (out.size() < n && std::getline(s, l)) out.push_back(l);
    out.resize(n);
    return out;
}

} // namespace

int soibean_main(int argc, char **argv) {
    const char *T = "[soibean]";
    std::string sbdir = "../share/vgan/soibean_dir/", treedir, dbprefix = "soibean_db", gam, fq1, fq2, out_prefix = "beanOut", deam5, deam3;
    bool run_mcmc = true, dbprefix_found = false, rand_start = false, specified_k = false, specified_deam = false;
    int n_threads = 1, iter = 500000, burnin = 75000, chains = 4, k = 1, penalty = 7, device = 0; // soibean.cpp:209-240
    std::vector<int> gpu_list;
    uint64_t seed = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char *flag) -> std::string {
            if (i + 1 >= argc) die(std::string("[soibean] Error, option ") + flag + " needs a value");
            return argv[++i];
        };
        auto non_negative = [&](const char *flag) {
            const int v = parse_int(need(flag), flag, T);
            if (v < 0) die(std::string("[soibean] Error, option ") + flag + " must not be negative");
            return v;
        };
        if (a == "-h" || a == "--help" || a == "-") {
            std::cerr << soibean_usage() << std::endl;
            return 0;
        } else if (a == "--soibean_dir" || a == "--soibean-dir") {
            sbdir = need("--soibean_dir");
            if (sbdir.empty() || sbdir.back() != '/') sbdir += '/';
        } else if (a == "--tree_dir" || a == "--tree-dir") {
            treedir = need("--tree_dir");
            if (treedir.empty() || treedir.back() != '/') treedir += '/';
        } else if (a == "--dbprefix") {
            dbprefix = need("--dbprefix");
            dbprefix_found = true;
        } else if (a == "-fq1") reject_fasta(fq1 = need("-fq1"), T);
        else if (a == "-fq2") reject_fasta(fq2 = need("-fq2"), T);
        else if (a == "-i") {
            if (!fq2.empty()) die("[soibean] If interleaved option chosen, soibean expects only one FASTQ file");
        } else if (a == "-g") gam = need("-g");
        else if (a == "-M" || a == "-z" || a == "-l" || a == "-S" || a == "--SAFARI") (void)need(a.c_str()); // mapping / profiling side
        else if (a == "-t") {
            n_threads = parse_int(need("-t"), "-t", T);
            if (n_threads < -1 || n_threads == 0) die("[soibean] Error, invalid number of threads");
            const int hw = (int)vgan_host_cpus();
            if (n_threads == -1 || n_threads > hw) n_threads = hw;
        } else if (a == "--no-mcmc") run_mcmc = false;
        else if (a == "--iter" || a == "--iterations") iter = non_negative("--iter");
        else if (a == "--burnin") burnin = non_negative("--burnin");
        else if (a == "--chains") chains = non_negative("--chains");
        else if (a == "--deam5p") {
            deam5 = need("--deam5p");
            specified_deam = true;
        } else if (a == "--deam3p") {
            deam3 = need("--deam3p");
            specified_deam = true;
        } else if (a == "-o") out_prefix = need("-o");
        else if (a == "--randStart" || a == "--randstart") rand_start = true;
        else if (a == "-k") {
            k = non_negative("-k");
            specified_k = true;
        } else if (a == "--pathThres") {
            if (parse_int(need("--pathThres"), "--pathThres", T) <= 0) die("[soibean] Error, option --pathThres must be positive");
        } else if (a == "-P") {
            penalty = parse_int(need("-P"), "-P", T);
            if (penalty <= 0) die("[soibean] Error, option -P must be positive");
        } else if (a == "--alignment-detail" || a == "--alignment_detail")
            die("[soibean] the per-base alignment table is not kept on the GPU path (the tables are factorised, include/vgan_gpu.h)");
        else if (a == "--seed") seed = (uint64_t)std::strtoull(need("--seed").c_str(), nullptr, 10);
        else if (a == "--device") {
            device = parse_int(need("--device"), "--device", T);
            if (device < 0) die("[soibean] Error, --device needs a non-negative GPU index");
        } else if (a == "--gpus") {
            parse_gpu_list(need("--gpus"), "--gpus", T, gpu_list);
        } else die("[soibean] Error, unrecognized option " + a);
    }
    if (!fq1.empty() || !fq2.empty())
        die("[soibean] FASTQ input needs vg giraffe in-process, which this build does not have; map with vg and pass -g");
    if (specified_deam && (deam5.empty() || deam3.empty())) die("Error the damage profiles do not exist. Unable to proceed."); // soibean.cpp:418-422
    if (!dbprefix_found)
        die("No database specified. Please choose a taxon of interest and specify it with the --dbprefix option. You can create a graph for "
            "you taxon of interested by using the make_graph_file.sh script."); // :427-429
    if (iter < burnin) die("The number of iterations must be higher than the burn-in period. Unable to proceed."); // :439-441
    if (treedir.empty()) treedir = sbdir + "tree_dir/";
    const std::string gfa = sbdir + dbprefix + (is_file(sbdir + dbprefix + ".gfa") ? ".gfa" : ".og"), treename = treedir + dbprefix + ".new.dnd", freqname = sbdir + "soibean_db.baseFreq";
    if (!is_file(gfa)) die(gfa + " does not exist.");
    if (!is_file(treename)) die(treename + " does not exist.");
    if (gam.empty()) die("[soibean] Error, no input file given (use -g)");
    if (!is_readable_input(gam)) die("[soibean] Error, GAM input file " + gam + " does not exist");

    PhaseTimer pt("soibean");
    if (gpu_list.empty()) {
        if (const char *e = getenv("VGAN_GPUS")) {
            if (std::string(e) == "all")
                for (int d = 0; d < vgan_device_count(); ++d) gpu_list.push_back(d);
            else if (*e) parse_gpu_list(e, "VGAN_GPUS", T, gpu_list);
        }
        if (gpu_list.empty()) gpu_list.push_back(device);
    }
    // A long BGZF input: the front end runs ON THE DEVICE (vgan_sb_gam_*: the file in pieces through inflate, framing, protobuf walk and
    // soibean's flatten as kernels, piece i to context i mod n; csrc/gam_pipe.hip, sb_flatten_kernels.hip, sb_gam_run.hip); the host parses
    // only the reads the device flatten leaves (indels, soft clips).  VGAN_SB_DEVICE_GAM=0 / 1: never / whenever the input is a regular
    // file.  Anything the device refuses goes through the host pipeline.
    struct FileMap {
        const uint8_t *p = nullptr;
        size_t n = 0;
        ~FileMap() {
            if (p) munmap(const_cast<uint8_t *>(p), n);
        }
    } gam_map;
    struct GdRun {
        vgan_sb_gamrun *r = nullptr;
        ~GdRun() { vgan_sb_gam_free(r); }
    };
    bool device_gam = false;
    {
        const char *e = getenv("VGAN_SB_DEVICE_GAM");
        struct stat sb;
        if (!(e && e[0] == '0') && vgan_device_count() > 0 && stat(gam.c_str(), &sb) == 0 && S_ISREG(sb.st_mode) &&
            ((e && e[0] == '1') || (uint64_t)sb.st_size >= (128ull << 20)) && sb.st_size > 28) {
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
        for (int d : gpu_list)
            if (d >= vgan_device_count()) device_gam = false; // (reported below, once the tables are read)
    }
    DeviceWarm warm; // (the runtime and the run's code objects, beside the graph's and the tree's loading)
    warm.start(gpu_list.empty() ? 0 : gpu_list[0], VGAN_PRELOAD_SB | (device_gam ? VGAN_PRELOAD_GAM : 0u));
    GamReader reader;
    GdRun gd; // (started now: upload, inflate, framing and parse need neither graph nor contexts and run beside their set-up)
    if (device_gam) {
        vgan_gampipe_opts po{};
        po.n_threads = n_threads;
        if (vgan_sb_gam_start(gpu_list.data(), (int)gpu_list.size(), gam_map.p, gam_map.n, &po, &gd.r) < 0) device_gam = false;
    }
    if (!device_gam) reader.start(gam, 0);
    Handle<vgan_damage> dmg(vgan_damage_free);
    check(vgan_damage_load(deam5.empty() ? nullptr : deam5.c_str(), deam3.empty() ? nullptr : deam3.c_str(), &dmg.p), "damage profiles");
    std::cerr << "Reading in variation graph ..." << std::endl;
    Handle<vgan_graph> graph(vgan_graph_free);
    check(vgan_graph_load(gfa.c_str(), nullptr, &graph.p), "loading graph");
    vgan_graph_view gv;
    check(vgan_graph_view_get(graph.p, &gv), "graph view");
    if (gv.n_paths == 0) die("Error: The path_names vector is empty. Unable to proceed.");
    if (gfa.size() > 3 && gfa.compare(gfa.size() - 3, 3, ".og") == 0) // soibean.cpp:446: the haplotype index beside the graph
        check_gbwt_beside(sbdir + dbprefix, gv.max_id - gv.min_id + 1, gv.n_paths, "[soibean]");
    const std::vector<std::string> path_names = lines_of(gv.path_names, gv.n_paths);

    std::cerr << "Loading tree ... " << std::endl;
    Handle<vgan_tree> tree(vgan_tree_free);
    check(vgan_tree_load(treename.c_str(), &tree.p), "loading tree");
    vgan_tree_view tv;
    check(vgan_tree_view_get(tree.p, &tv), "tree view");
    const std::vector<std::string> node_names = lines_of(tv.names, tv.n_nodes);
    double shortest = tv.dist[0]; // soibean.cpp:576-602
    for (uint32_t j = 0; j < tv.n_nodes; ++j)
        if (tv.dist[j] < shortest && tv.dist[j] != 0.0) shortest = tv.dist[j];
    const double con = (shortest != 0 && shortest < 1) ? shortest : 0.01;
    std::cerr << " ... done!" << std::endl << "Number of tree nodes " << tv.n_nodes << std::endl;
    if (tv.n_nodes != gv.n_paths) die("The number of tree nodes and paths in the graph is unequal. Unable to proceed. Exiting...");
    std::vector<int32_t> node_path(tv.n_nodes, -1);
    {
        std::map<std::string, int32_t> by_name;
        for (uint32_t p = 0; p < gv.n_paths; ++p) by_name[path_names[p]] = (int32_t)p;
        for (uint32_t v = 0; v < tv.n_nodes; ++v) {
            const auto it = by_name.find(node_names[v]);
            if (it == by_name.end()) die("[soibean] tree node " + node_names[v] + " is not a path of the graph");
            node_path[v] = it->second;
        }
    }
    std::vector<int32_t> path_node(gv.n_paths, -1);
    for (uint32_t v = 0; v < tv.n_nodes; ++v) path_node[(size_t)node_path[v]] = (int32_t)v;

    vgan_sb_estimate_cfg cfg{};
    { // base frequencies of the taxon (soibean.cpp:609-640); all zero when the file has no such line, as there
        std::ifstream f(freqname);
        if (!f) die("Failed to open the base frequency file.");
        std::string line;
        double fa = 0, fc = 0, fg = 0, ft = 0;
        while (std::getline(f, line)) {
            std::istringstream is(line);
            std::string name;
            double x0 = 0, x1 = 0, x2 = 0, x3 = 0;
            is >> name >> x0 >> x1 >> x2 >> x3;
            if (name == dbprefix) {
                fa = x0, fc = x1, fg = x2, ft = x3;
                break;
            }
        }
        const double fr = fa + fg, fy = fc + ft;
        const double fm = 1 / (2 * (((22) * (fa * fg)) + ((22) * (fc * ft)) + (fa * fc + (fa * ft) + (fg * fc + (fg * ft)))));
        const double f7[7] = {fa, fc, fg, ft, fr, fy, fm};
        std::copy(f7, f7 + 7, cfg.freqs7);
    }
    pt.lap("graph + tree");

    if (vgan_device_count() <= 0) die("[soibean] no HIP device is visible: the likelihood path runs on the GPU only");
    vgan_damage_view dmv;
    check(vgan_damage_view_get(dmg.p, &dmv), "damage view");
    vgan_sb_params prm{penalty, 0};
    // one device context per GPU (--gpus / VGAN_GPUS; a GPU may be named twice: two contexts on it), the reads dealt to them
    // once in contiguous shares: MCMC.cpp:739 runs its loop over the reads in parallel with `reduction(+:logLike)`, here the
    // reduction runs over devices -- in integers (vgan_sb_sum), so the number of devices does not change a bit of the result
    struct Contexts {
        std::vector<vgan_sb_ctx *> v;
        vgan_sb_group *group = nullptr;
        ~Contexts() {
            vgan_sb_group_free(group);
            for (auto c : v) vgan_sb_destroy(c);
        }
    } ctxs;
    struct GdStop { // (an error's unwinding: the front end's threads use the contexts -- they are joined before the contexts go)
        vgan_sb_gamrun *&r;
        ~GdStop() {
            vgan_sb_gam_free(r);
            r = nullptr;
        }
    } gd_stop{gd.r};
    for (int d : gpu_list) {
        if (d >= vgan_device_count()) die("[soibean] Error, GPU " + std::to_string(d) + " asked for, " + std::to_string(vgan_device_count()) + " visible");
        vgan_sb_ctx *c = nullptr;
        check(vgan_sb_create(&gv, &dmv, &prm, d, &c), "creating the device context");
        ctxs.v.push_back(c);
    }
    check(vgan_sb_group_create(ctxs.v.data(), (int)ctxs.v.size(), &ctxs.group), "grouping the device contexts");
    vgan_sb_flatten_stats st{};
    int64_t dev_bad = 0;
    const int64_t n_ctx = (int64_t)ctxs.v.size();
    if (device_gam) {
        vgan_sb_gam_result gr{};
        vgan_gampipe_stats ps{};
        int rc = vgan_sb_gam_attach(gd.r, ctxs.v.data(), (int)n_ctx, graph.p);
        if (rc >= 0) rc = vgan_sb_gam_finish(gd.r, &gr, &ps);
        if (rc < 0) { // (vgan_sb_precompute replaces a context's tables: the host pipeline takes the file from its start)
            std::cerr << "[soibean] the device front end does not take this input (" << vgan_last_error() << "): the host pipeline does" << std::endl;
            device_gam = false;
            reader.start(gam, 0);
        } else {
            st.n_bad = gr.n_bad;
            dev_bad = gr.n_dev_bad;
            if (getenv("VGAN_TIMING"))
                fprintf(stderr, "[vgan timing] soibean device front end: %.1f MB -> %.1f MB in %llu pieces on %lld lane(s), %llu messages, %llu mapped reads (%llu left to the host); %.0f ms "
                                "from start to the last piece, analyse_GAM's tables %.0f ms; summed over pieces: upload %.0f, inflate %.0f, framing %.0f, protobuf walk %.0f, flatten %.0f ms; "
                                "%.2f GB of device memory\n",
                        ps.compressed_bytes / 1e6, ps.inflated_bytes / 1e6, (unsigned long long)ps.n_pieces, (long long)n_ctx, (unsigned long long)ps.n_messages, (unsigned long long)ps.n_reads,
                        (unsigned long long)ps.n_host_reads, ps.ms_wall, gr.ms_tables, ps.ms_upload, ps.ms_inflate, ps.ms_frame, ps.ms_parse, ps.ms_consume, ps.device_bytes / 1e9);
        }
        vgan_sb_gam_free(gd.r); // (the batch's memory back: the tables stay in the contexts)
        gd.r = nullptr;
    }
    Handle<vgan_alnset> aln(vgan_aln_free);
    vgan_alnset_view av{};
    if (!device_gam) {
        aln.p = reader.take();
        check(vgan_aln_view_get(aln.p, &av), "alignment view");
    }
    for (int64_t i = 0; i < n_ctx && !device_gam; ++i) {
        const int64_t r0 = av.n_reads * i / n_ctx, r1 = av.n_reads * (i + 1) / n_ctx;
        Handle<vgan_sb_host_batch> hb(vgan_sb_host_batch_free);
        vgan_sb_flatten_stats sti{};
        check(vgan_sb_flatten(graph.p, aln.p, r0, r1, n_threads, &hb.p, &sti), "flattening");
        st.n_in += sti.n_in;
        st.n_out += sti.n_out;
        st.n_unmapped += sti.n_unmapped;
        st.n_bad += sti.n_bad;
        vgan_sb_batch b;
        check(vgan_sb_host_batch_get(hb.p, &b), "batch");
        int64_t bad_i = 0;
        check(vgan_sb_precompute(ctxs.v[(size_t)i], &b, &bad_i), "analyse_GAM");
        dev_bad += bad_i;
    }
    std::vector<int64_t> sig(gv.n_paths);
    int64_t n_ok = 0;
    check(vgan_sb_group_best_paths(ctxs.group, sig.data(), &n_ok), "signature counts");
    if (n_ctx > 1) std::cerr << "[soibean] " << n_ctx << " device contexts, the reads dealt " << (device_gam ? "piece by piece" : "in contiguous shares") << std::endl;
    pt.lap("GAM + analyse_GAM");
    std::cerr << "Number of paths: " << gv.n_paths << std::endl << "Number of reads: " << n_ok << std::endl;
    if (st.n_bad + dev_bad) std::cerr << "[soibean] warning: " << st.n_bad + dev_bad << " reads skipped (the reference would index out of bounds on them)\n";
    if (n_ok == 0) die("[soibean] no usable read in the input");

    // starting nodes (soibean.cpp:642-736)
    cfg.seed = seed;
    auto random_nodes = [&](uint32_t n) { // soibean::generateRandomNumbers, seeded like the chains but from a stream of its own
        std::mt19937 gen(seed ? (uint32_t)(seed * 0x9E3779B97F4A7C15ull >> 32) : std::random_device{}());
        std::uniform_int_distribution<> pick(0, (int)gv.n_paths - 1);
        std::vector<int32_t> v;
        for (uint32_t i = 0; i < n; ++i) v.push_back(pick(gen));
        return v;
    };
    std::vector<int32_t> sig_nodes;
    if (specified_k) {
        if ((uint32_t)k > gv.n_paths) {
            std::cerr << "Number for k cannot be larger than the number of tree nodes present. The tree has " << tv.n_nodes
                      << " nodes. Please adjust k accordingly. Exiting..." << std::endl;
            die("Invalid number of k.");
        }
        if (k == 0) die("Invalid number of k.");
        std::cerr << "User specified number of sources was set to k = " << k << ". A random start is being initiated." << std::endl;
        sig_nodes = random_nodes((uint32_t)k);
    } else {
        std::cerr << "Finding the initial estimate ..." << std::endl;
        std::vector<int32_t> paths(gv.n_paths);
        int32_t n = 0;
        check(vgan_sb_signature_paths(sig.data(), gv.n_paths, n_ok, 0, paths.data(), &n), "signature paths");
        if (n == 0) {
            std::cerr << "Still, no signature node-sets could be identified. Initiating the MCMC with k = 3 and random starting nodes." << std::endl;
            sig_nodes = random_nodes(3);
        } else {
            for (int32_t j = 0; j < n; ++j) {
                sig_nodes.push_back(path_node[(size_t)paths[j]]);
                std::cerr << "Identified signature paths: " << path_names[(size_t)paths[j]] << " with tree node: " << sig_nodes.back() << std::endl;
            }
            if (rand_start) sig_nodes = random_nodes((uint32_t)sig_nodes.size());
        }
        std::cerr << "... done! The Identified signature paths are used as input for the MCMC. " << std::endl;
    }
    if (specified_k || rand_start) {
        std::cerr << "Random starting nodes: ";
        for (int32_t v : sig_nodes) std::cerr << v << " ";
        std::cerr << std::endl;
    }

    cfg.max_iter = (uint32_t)iter;
    cfg.burn = (uint32_t)burnin;
    cfg.chains = (uint32_t)chains;
    cfg.n_paths = gv.n_paths;
    cfg.con = con;
    cfg.run_mcmc = run_mcmc;
    cfg.quiet = 0;
    vgan_sb_engine engine;
    check(vgan_sb_engine_group(ctxs.group, &engine), "engine");
    check(vgan_sb_estimate(&engine, tree.p, node_path.data(), sig_nodes.data(), (uint32_t)sig_nodes.size(), &cfg, out_prefix.c_str()), "estimation");
    pt.lap("chains");
    return 0;
}
