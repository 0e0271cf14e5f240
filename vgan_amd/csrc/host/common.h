// Host-side data model shared by the front end (GAM/GFA/sidecar readers, flattening, synthetic inputs).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "vgan_gpu.h"

namespace vgan {

void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);
const char *last_error();

// whole file -> bytes; gzip/BGZF members are inflated when the magic is present
bool read_file(const std::string &path, std::string &out, bool inflate_if_gzip = true);
bool gunzip_members(const void *data, size_t n, std::string &out);
bool gzip_bytes(const std::string &in, std::string &out);
bool write_file(const std::string &path, const std::string &bytes);
bool file_exists(const std::string &path);
// opens <path> or <path>.gz
bool read_text_maybe_gz(const std::string &path, std::string &out);

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
};

struct Recon {
    std::string gseq, ps;
    std::vector<int32_t> sizes;
};

} // namespace vgan

struct vgan_graph;
struct vgan_alnset;
namespace vgan {
// reconstruct_graph_sequence (reference src/vgan_utils.h:6-79); 0 or a code for reads the reference dies on
int reconstruct(const vgan_graph &g, const vgan_alnset &a, int64_t r, Recon &o);

} // namespace vgan

struct vgan_graph {
    int64_t min_id = 0, max_id = -1;
    std::vector<int64_t> node_seq_off; // [max_id+2], by node id
    std::string node_seq;
    uint32_t n_paths = 0, mask_words = 0;
    std::vector<uint64_t> mask;          // [(max_id+1)*mask_words]
    std::vector<int32_t> pangenome_base; // [max_id+1]
    std::vector<double> mappability;
    std::string path_names, parents_txt, children_txt;
    std::vector<std::vector<std::pair<int64_t, bool>>> path_steps; // GFA P lines when known (for writing)

    const char *seq_ptr(int64_t id) const { return node_seq.data() + node_seq_off[id]; }
    int64_t seq_len(int64_t id) const { return node_seq_off[id + 1] - node_seq_off[id]; }
    bool has_node(int64_t id) const { return id >= min_id && id <= max_id; }
    void fill_view(vgan_graph_view *v) const;
};

struct vgan_alnset {
    std::vector<int64_t> seq_off{0}, qual_off{0}, name_off{0}, map_off{0}, edit_off{0}, e_seq_off{0};
    std::string seq, qual, name, e_seq;
    std::vector<int32_t> mapq;
    std::vector<double> identity;
    std::vector<int64_t> m_node, m_offset;
    std::vector<uint8_t> m_rev;
    std::vector<int32_t> e_from, e_to;
    int64_t n_reads() const { return (int64_t)mapq.size(); }
    void fill_view(vgan_alnset_view *v) const;
};

namespace vgan {
void merge_alnsets(std::vector<vgan_alnset> &parts, vgan_alnset &out); // parts are consumed
}

struct vgan_hc_host_batch {
    std::vector<uint32_t> read_seg_off{0}, read_col_off{0}, read_qual_off{0};
    std::vector<uint16_t> read_algn_len;
    std::vector<uint8_t> read_mapq;
    std::vector<uint32_t> seg_node;
    std::vector<uint16_t> seg_start, seg_len;
    std::vector<uint8_t> graph_seq, algnseq, qual;
    std::vector<uint32_t> read_src; // index of each batch read in the alignment set
    uint32_t n_tileable = 0;        // reads [0, n_tileable) satisfy the tile contract (include/vgan_gpu.h)
    void fill(vgan_hc_batch *b) const;
};
