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
    puts("host sanitizer driver: ok");
    return 0;
}
