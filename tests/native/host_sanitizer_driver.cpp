// Exercises the host front end (GAM parse incl. corrupted input, graph load/write, flatten for the three paths,
// synthetic generators, duplicate marking) in an AddressSanitizer + UBSan build of the host sources.
// Built and run by tests/test_sanitizers_cpu.py; GPU code is not part of this build (GPU ASan is unavailable here).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <vector>

#include "vgan_gpu.h"

#define REQUIRE(x)                                                         \
    do {                                                                   \
        if (!(x)) {                                                        \
            fprintf(stderr, "REQUIRE failed: %s (%s)\n", #x, vgan_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static std::string slurp(const std::string &p) {
    std::ifstream f(p, std::ios::binary);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string golden = argv[1], tmp = argv[2];
    const uint64_t n_synth = argc > 3 ? strtoull(argv[3], nullptr, 10) : 3000; // reads of the synthetic section
    // graph from GFA, reconstruction KATs
    vgan_graph *g = nullptr;
    REQUIRE(vgan_graph_load((golden + "/reconstruct/target_graph.gfa").c_str(), nullptr, &g) == 0);
    vgan_alnset *a = nullptr;
    REQUIRE(vgan_aln_read_gam((golden + "/reconstruct/test_reads.gam").c_str(), 0, &a) == 0);
    char gs[4096], rs[4096];
    int32_t sizes[4096];
    int64_t lens[3];
    REQUIRE(vgan_reconstruct(g, a, 7, gs, rs, sizes, 4096, lens) == 0);
    REQUIRE(std::string(gs) == "TCTTGCGGTTCTTGGTC------------GACCCTACTCACGGTATAAATGGGGCGCGCTCCAT");
    vgan_hc_host_batch *hb = nullptr;
    vgan_hc_flatten_stats st;
    REQUIRE(vgan_hc_flatten(g, a, 0, 10, 2, &hb, &st) == 0 && st.n_out == 10);
    vgan_hc_host_batch_free(hb);
    vgan_aln_free(a);
    vgan_graph_free(g);

    { // the ODGI reader on the reference's fixture and on damaged copies of it
        vgan_graph *og = nullptr;
        REQUIRE(vgan_graph_load((golden + "/reconstruct/target_graph.og").c_str(), nullptr, &og) == 0);
        vgan_graph_view ov;
        REQUIRE(vgan_graph_view_get(og, &ov) == 0 && ov.n_paths == 5 && ov.min_id == 2 && ov.max_id == 29);
        vgan_graph_free(og);
        const std::string og_raw = slurp(golden + "/reconstruct/target_graph.og");
        std::mt19937 r2(3);
        for (int trial = 0; trial < 300; ++trial) {
            std::string dmg = og_raw;
            if (trial % 3 == 0) dmg.resize(4 + r2() % (dmg.size() - 4));
            else if (trial % 3 == 1)
                for (int k = 0; k < 4; ++k) dmg[r2() % dmg.size()] = (char)(r2() & 0xff);
            else {
                const uint64_t v = ((uint64_t)r2() << 32 | r2()) >> (r2() % 64);
                memcpy(&dmg[4 + r2() % (dmg.size() - 12)], &v, 8);
            }
            std::ofstream(tmp + "/d.og", std::ios::binary).write(dmg.data(), (std::streamsize)dmg.size());
            vgan_graph *x = nullptr;
            if (vgan_graph_load((tmp + "/d.og").c_str(), nullptr, &x) == 0) vgan_graph_free(x);
        }
    }

    // corrupted / truncated GAM payloads
    const std::string raw = slurp(golden + "/alignments/J2a1a1a1.gam");
    std::mt19937 rng(11);
    int n_err = 0;
    for (int trial = 0; trial < 400; ++trial) {
        std::string d = raw;
        if (trial % 2) d.resize(rng() % d.size());
        else
            for (int k = 0; k < 5; ++k) d[rng() % d.size()] = (char)(rng() & 0xff);
        vgan_alnset *x = nullptr;
        const int rc = vgan_aln_parse_gam(d.data(), d.size(), 1, &x);
        if (rc != 0) ++n_err;
        else vgan_aln_free(x);
    }
    REQUIRE(n_err > 0);

    // synthetic graph + reads through every flatten, GAM round trip, duplicate marks, graph round trip
    vgan_synth_graph_cfg gc{5, 1500, 1000, 40};
    REQUIRE(vgan_synth_hc_graph(&gc, &g) == 0);
    vgan_synth_reads_cfg rc{3, n_synth, 100, 0.2, 0.2, 0.1, 1};
    REQUIRE(vgan_synth_hc_reads(g, &rc, &a) == 0);
    REQUIRE(vgan_aln_write_gam(a, (tmp + "/s.gam").c_str(), 100) == 0);
    vgan_alnset *b = nullptr;
    REQUIRE(vgan_aln_read_gam((tmp + "/s.gam").c_str(), 1, &b) == 0); // unmapped reads kept: the round trip is exact
    vgan_alnset_view va, vb;
    vgan_aln_view_get(a, &va);
    vgan_aln_view_get(b, &vb);
    REQUIRE(va.n_reads == vb.n_reads && memcmp(va.seq, vb.seq, (size_t)va.seq_off[va.n_reads]) == 0);
    std::vector<uint8_t> dup((size_t)va.n_reads);
    int64_t nd = 0;
    REQUIRE(vgan_aln_mark_duplicates(a, dup.data(), &nd) == 0);
    vgan_alnset *kept = nullptr;
    REQUIRE(vgan_aln_filter(a, dup.data(), &kept) == 0);
    REQUIRE(vgan_hc_flatten(g, kept, 0, va.n_reads - nd, 3, &hb, &st) == 0);
    vgan_hc_host_batch_free(hb);
    REQUIRE(vgan_hc_flatten_masked(g, a, 0, va.n_reads, dup.data(), 3, &hb, &st) == 0 && st.n_out + st.n_bad + st.n_unmapped == va.n_reads - nd);
    vgan_hc_host_batch_free(hb);
    { // the same GAM as the parser's slices: marks, slice-range flatten, merge
        vgan_alnparts *ps = nullptr;
        REQUIRE(vgan_alnparts_read_gam((tmp + "/s.gam").c_str(), 1, &ps) == 0);
        REQUIRE(vgan_alnparts_n_reads(ps) == vb.n_reads && vgan_alnparts_count(ps) >= 1);
        std::vector<uint8_t> dup2((size_t)vb.n_reads);
        int64_t nd2 = 0;
        REQUIRE(vgan_alnparts_mark_duplicates(ps, dup2.data(), &nd2) == 0 && nd2 == nd && dup2 == dup);
        REQUIRE(vgan_hc_flatten_parts(g, ps, 0, vgan_alnparts_count(ps), dup2.data(), 3, &hb, &st) == 0);
        vgan_hc_host_batch_free(hb);
        vgan_alnset *m = nullptr;
        REQUIRE(vgan_alnparts_merge(ps, &m) == 0);
        vgan_alnset_view vm;
        vgan_aln_view_get(m, &vm);
        REQUIRE(vm.n_reads == vb.n_reads && vgan_alnparts_n_reads(ps) == 0);
        vgan_aln_free(m);
        vgan_alnparts_free(ps);
        // and as a stream of chunks with duplicate marks carried across them
        vgan_gam_stream *gs = nullptr;
        vgan_dedup *dd = nullptr;
        REQUIRE(vgan_gam_stream_open((tmp + "/s.gam").c_str(), 1, &gs) == 0 && vgan_dedup_create(&dd) == 0);
        int64_t seen = 0, nd3 = 0;
        for (;;) {
            vgan_alnparts *ch = nullptr;
            REQUIRE(vgan_gam_stream_next(gs, 9000, &ch) == 0);
            if (!ch) break;
            REQUIRE(vgan_alnparts_base(ch) == seen);
            const int64_t nr = vgan_alnparts_n_reads(ch);
            std::vector<uint8_t> dm((size_t)nr);
            int64_t k = 0;
            REQUIRE(vgan_dedup_mark(dd, ch, dm.data(), &k) == 0);
            REQUIRE(memcmp(dm.data(), dup.data() + seen, (size_t)nr) == 0);
            REQUIRE(vgan_hc_flatten_parts(g, ch, 0, vgan_alnparts_count(ch), dm.data(), 2, &hb, &st) == 0);
            vgan_hc_host_batch_free(hb);
            vgan_alnparts_free(ch);
            seen += nr;
            nd3 += k;
        }
        REQUIRE(seen == vb.n_reads && nd3 == nd);
        vgan_dedup_free(dd);
        vgan_gam_stream_close(gs);
        // a stream abandoned half way is torn down cleanly
        REQUIRE(vgan_gam_stream_open((tmp + "/s.gam").c_str(), 1, &gs) == 0);
        vgan_gam_stream_close(gs);
    }
    vgan_euka_host_batch *eb = nullptr;
    vgan_euka_flatten_stats es;
    REQUIRE(vgan_euka_flatten(g, a, 0, va.n_reads, 3, &eb, &es) == 0);
    vgan_euka_host_batch_free(eb);
    vgan_sb_host_batch *sb = nullptr;
    vgan_sb_flatten_stats ss;
    REQUIRE(vgan_sb_flatten(g, a, 0, va.n_reads, 3, &sb, &ss) == 0);
    vgan_sb_host_batch_free(sb);
    REQUIRE(vgan_graph_write(g, tmp.c_str()) == 0);
    vgan_graph *g2 = nullptr;
    REQUIRE(vgan_graph_load((tmp + "/graph.gfa").c_str(), tmp.c_str(), &g2) == 0);
    vgan_graph_free(g2);
    vgan_aln_free(kept);
    vgan_aln_free(b);
    vgan_aln_free(a);
    vgan_graph_free(g);

    // euka tables + synthetic euka input
    vgan_euka_db *db = nullptr;
    REQUIRE(vgan_euka_db_load((golden + "/euka_dir/euka_db.clade").c_str(), (golden + "/euka_dir/euka_db.bins").c_str(), &db) == 0);
    vgan_euka_db_free(db);
    vgan_damage *dm = nullptr;
    REQUIRE(vgan_damage_load((golden + "/damageProfiles/dhigh5p.prof").c_str(), (golden + "/damageProfiles/dhigh3p.prof").c_str(), &dm) == 0);
    vgan_synth_euka_cfg ec{9, 6, 120, 500, 75, 0, 0};
    REQUIRE(vgan_synth_euka(&ec, dm, &g, &db, &a) == 0);
    REQUIRE(vgan_euka_flatten(g, a, 0, 500, 2, &eb, &es) == 0 && es.n_out > 450);
    vgan_euka_host_batch_free(eb);
    vgan_aln_free(a);
    vgan_euka_db_free(db);
    vgan_graph_free(g);
    vgan_damage_free(dm);
    REQUIRE(vgan_damage_from_text("A>C\tA>G\n0\t0\n", "", &dm) != 0);

    // soibean chain driver over a stand-in engine (the likelihood of a state is a smooth function of its positions), on one
    // of the shipped trees; malformed Newick texts are errors
    {
        vgan_tree *tr = nullptr;
        REQUIRE(vgan_tree_load((golden + "/trees/Ursidae.new.dnd").c_str(), &tr) == 0);
        vgan_tree_view tv;
        REQUIRE(vgan_tree_view_get(tr, &tv) == 0 && tv.n_nodes > 10 && tv.n_leaves > 5);
        std::vector<int32_t> node_path(tv.n_nodes);
        for (uint32_t v = 0; v < tv.n_nodes; ++v) node_path[v] = (int32_t)v;
        vgan_sb_engine eng{}; // refresh_many stays null: the driver asks for one state at a time
        eng.user = nullptr;
        eng.refresh = [](void *, uint32_t k, const vgan_sb_source *src, double, const double *, double *ll, uint64_t *guard) {
            double v = -100.0;
            for (uint32_t y = 0; y < k; ++y) v -= 3.0 * (src[y].pos - 0.3) * (src[y].pos - 0.3) + 0.01 * src[y].child + src[y].theta;
            *ll = v;
            *guard = 0;
            return 0;
        };
        eng.mixture = [](void *, uint32_t n, const int32_t *, double, double *ll) {
            *ll = -50.0 * n;
            return 0;
        };
        vgan_sb_estimate_cfg cfg{};
        cfg.max_iter = 400;
        cfg.burn = 100;
        cfg.chains = 3;
        cfg.n_paths = tv.n_nodes;
        cfg.seed = 12;
        cfg.con = 0.01;
        cfg.run_mcmc = 1;
        cfg.quiet = 1;
        const int32_t start[3] = {1, (int32_t)tv.n_nodes - 1, 0};
        REQUIRE(vgan_sb_estimate(&eng, tr, node_path.data(), start, 3, &cfg, (tmp + "/bean_").c_str()) == 0);
        REQUIRE(!slurp(tmp + "/bean_Diagnostics30.txt").empty() && !slurp(tmp + "/bean_Result31.mcmc").empty());
        cfg.burn = cfg.max_iter;
        REQUIRE(vgan_sb_estimate(&eng, tr, node_path.data(), start, 1, &cfg, (tmp + "/bean2_").c_str()) != 0);
        vgan_tree_free(tr);
        for (const char *bad : {"", "(", "(a,b", "(a:1,b:)c;", "((((((", "a:1e999999999;x"}) {
            vgan_tree *t2 = nullptr;
            if (vgan_tree_parse(bad, &t2) == 0) vgan_tree_free(t2);
        }
    }
    // euka downstream: detected clades, abundance chain and the output files on a made-up result
    {
        REQUIRE(vgan_euka_db_load((golden + "/euka_dir/euka_db.clade").c_str(), (golden + "/euka_dir/euka_db.bins").c_str(), &db) == 0);
        vgan_euka_db_view dv;
        REQUIRE(vgan_euka_db_view_get(db, &dv) == 0);
        const uint32_t C = dv.n_clades, NB = dv.bin_off[C];
        std::vector<int32_t> count(C, 0), rclade;
        std::vector<uint32_t> shift((size_t)C * 10 * 16, 3);
        std::vector<double> cov(NB, 0.0), sll(C, 0.0);
        std::vector<int64_t> nl(C, 0);
        std::vector<uint8_t> rpass;
        std::vector<uint16_t> rlen;
        for (uint32_t c = 0; c < C; c += 7) {
            count[c] = 40 + (int32_t)c;
            nl[c] = 60 + c;
            sll[c] = -0.3 * (double)nl[c];
            for (uint32_t j = dv.bin_off[c]; j < dv.bin_off[c + 1]; ++j) cov[j] = 2.5 + j % 3;
            for (int k = 0; k < 5; ++k) {
                rclade.push_back((int32_t)c);
                rpass.push_back(k != 2);
                rlen.push_back((uint16_t)(60 + k));
            }
        }
        vgan_euka_results res{};
        res.db = &dv;
        res.clade_count = count.data();
        res.baseshift = shift.data();
        res.bin_cov = cov.data();
        res.n_like = nl.data();
        res.sum_log_like = sll.data();
        res.n_reads = (int64_t)rclade.size();
        res.read_clade = rclade.data();
        res.read_pass = rpass.data();
        res.read_seq_len = rlen.data();
        vgan_euka_report_cfg rc2{};
        rc2.detect = {1, 10, 0, 0.0};
        rc2.length_to_prof = 5;
        rc2.run_mcmc = 1;
        rc2.iter = 300;
        rc2.burnin = 30;
        rc2.seed = 4;
        std::vector<int32_t> det(C + 1);
        std::vector<double> est((size_t)(C + 1) * 5);
        int32_t nd2 = 0;
        REQUIRE(vgan_euka_report(&res, &rc2, (tmp + "/euka").c_str(), det.data(), &nd2, est.data()) == 0 && nd2 >= 2);
        rc2.run_mcmc = 0;
        REQUIRE(vgan_euka_report(&res, &rc2, (tmp + "/euka_nm").c_str(), det.data(), &nd2, est.data()) == 0);
        rc2.run_mcmc = 1;
        rc2.iter = 10;
        rc2.burnin = 10;
        REQUIRE(vgan_euka_report(&res, &rc2, (tmp + "/euka_bad").c_str(), det.data(), &nd2, est.data()) != 0);
        vgan_euka_db_free(db);
    }
    puts("host sanitizer driver: ok");
    return 0;
}
