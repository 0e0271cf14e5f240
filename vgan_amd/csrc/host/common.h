// Host-side data model shared by the front end (GAM/GFA/sidecar readers, flattening, synthetic inputs).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <functional>
#include <string>
#include <type_traits>
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
struct ByteBuf;
bool gunzip_members(const void *data, size_t n, ByteBuf &out); // large inputs: no zero fill, huge pages
bool gzip_bytes(const std::string &in, std::string &out);
bool write_file(const std::string &path, const std::string &bytes);
bool file_exists(const std::string &path);
// opens <path> or <path>.gz
bool read_text_maybe_gz(const std::string &path, std::string &out);
bool read_bytes_maybe_gz(const std::string &path, ByteBuf &out); // the same for large files (no copy, no zero fill)

// Allocator for the front end's large arrays: default-initialises (resize() does not write, so the pages of a merged
// array are first touched by the threads that fill it, not zeroed serially by the caller) and asks for transparent
// huge pages on big blocks (512x fewer page faults where THP is in madvise mode).
void big_free_bytes(void *p, size_t bytes);
template <class T> struct BigAlloc {
    using value_type = T;
    BigAlloc() = default;
    template <class U> BigAlloc(const BigAlloc<U> &) {}
    T *allocate(size_t n);
    void deallocate(T *p, size_t n) { big_free_bytes(p, n * sizeof(T)); }
    template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const BigAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const BigAlloc<U> &) const { return false; }
};
void *big_alloc_bytes(size_t bytes);         // util.cpp; throws std::bad_alloc
void big_free_bytes(void *p, size_t bytes); // with the size given to big_alloc_bytes (blocks of 64 KB and more are recycled)
void big_pool_release(int n_threads);       // the recycled blocks back to the system
template <class T> T *BigAlloc<T>::allocate(size_t n) { return static_cast<T *>(big_alloc_bytes(n * sizeof(T))); }
template <class T> using BigVec = std::vector<T, BigAlloc<T>>;
struct ByteBuf : BigVec<char> { // the few std::string operations the front end uses on byte arrays
    void append(const char *p, size_t n) { insert(end(), p, p + n); }
    void append(const ByteBuf &src, size_t pos, size_t n) { insert(end(), src.begin() + pos, src.begin() + pos + n); }
    void assign(const char *p, size_t n) { BigVec<char>::assign(p, p + n); }
    ByteBuf &operator+=(const std::string &s) {
        insert(end(), s.begin(), s.end());
        return *this;
    }
};

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

// A file's bytes without a copy where the file can be mapped (util.cpp).
struct MappedFile {
    const void *p = nullptr;
    size_t n = 0;
    bool mapped = false;
    std::string fallback;
    MappedFile() = default;
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    ~MappedFile();
    bool open_path(const std::string &path);
};

// BGZF (SAM spec 4.1): the members of a buffer with their places in the inflated stream; false when the buffer is anything
// else.  inflate_member: one gzip member into exactly out_size bytes.
struct BgzfBlock {
    size_t in_off, in_size, out_off, out_size;
};
bool bgzf_index(const unsigned char *p, size_t n, std::vector<BgzfBlock> &blocks);
bool inflate_member(const unsigned char *in, size_t in_size, unsigned char *out, size_t out_size);

// Block-parallel BGZF inflate that hands out the prefix inflated so far: the consumer (GAM framing) starts while later
// blocks are still being inflated (util.cpp).  start() returns false for anything that is not a BGZF stream.
struct AsyncInflate {
    ByteBuf out;
    AsyncInflate();
    ~AsyncInflate();
    AsyncInflate(const AsyncInflate &) = delete;
    AsyncInflate &operator=(const AsyncInflate &) = delete;
    bool start(const void *data, size_t n);
    bool wait_for(size_t upto); // until bytes [0, min(upto, out.size())) are final; false when a block was corrupt
    bool finish();              // joins the workers; false when a block was corrupt

  private:
    struct Impl;
    Impl *impl;
};

// The processors this process may keep busy: the smaller of its affinity mask and its cgroup's CPU quota (cpu.max; a
// container given 16 CPUs of a 256-thread host is throttled for the rest of every 100 ms period once a hundred threads have
// spent the quota in the first 15 ms of it -- the stages then stall in turn, and the work per read, not the thread count, is
// what the throughput follows).  Thread pools are sized from this, not from hardware_concurrency().
unsigned usable_cpus();
// For a job that ends within a period or two of the quota (a few hundred thousand reads): up to four times as many, the
// unspent quota of the idle time before it is what such a burst runs on.
unsigned burst_cpus();

// fn(0..n-1) on n threads: the caller runs fn(0), the others come from a pool of persistent workers (util.cpp).  A thread
// that is created and destroyed maps and unmaps its stack -- write locks on the address space, which every page fault of
// every other thread waits behind: the chunk loop's ~80 short-lived threads per 65k reads capped the flatten stage at 2 M reads/s
// whatever ran beside it.  Calls may come from several threads at once.
void parallel_run(int n, const std::function<void(int)> &fn);

// Phase timing of the host front end to stderr when VGAN_TIMING is set in the environment (developer aid).
struct PhaseTimer {
    const char *what;
    bool on;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(const char *w) : what(w), on(getenv("VGAN_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void lap(const char *phase) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[vgan timing] %s: %s %.1f ms\n", what, phase, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// CPU time of the calling thread, for the per-stage accounts VGAN_TIMING prints (the container's CPU quota makes the sum of
// these, not any one stage's wall time, what a long input's throughput follows).
inline double thread_cpu_ms() {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}
struct CpuAccount { // process-wide sums, in microseconds
    std::atomic<int64_t> inflate{0}, frame_parse{0}, flatten{0}, merge{0};
};
CpuAccount &cpu_account();

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
    vgan::BigVec<int64_t> seq_off{0}, qual_off{0}, name_off{0}, map_off{0}, edit_off{0}, e_seq_off{0};
    vgan::ByteBuf seq, qual, name, e_seq;
    vgan::BigVec<int32_t> mapq;
    vgan::BigVec<double> identity;
    vgan::BigVec<int64_t> m_node, m_offset;
    vgan::BigVec<uint8_t> m_rev;
    vgan::BigVec<int32_t> e_from, e_to;
    int64_t n_reads() const { return (int64_t)mapq.size(); }
    void fill_view(vgan_alnset_view *v) const;
};

namespace vgan {
void merge_alnsets(std::vector<vgan_alnset> &parts, vgan_alnset &out); // parts are consumed
}

// an alignment set kept as the slices the GAM parser produced, in input order
struct vgan_alnparts {
    std::vector<vgan_alnset> parts;
    std::vector<int64_t> first{0}; // first[i] = index of slice i's first read; first.back() = number of reads
    int64_t base = 0;              // index in the whole input of this object's first read (chunks of a stream)
    void index() {
        first.assign(parts.size() + 1, 0);
        for (size_t i = 0; i < parts.size(); ++i) first[i + 1] = first[i] + parts[i].n_reads();
    }
};

struct vgan_hc_host_batch {
    vgan::BigVec<uint32_t> read_seg_off{0}, read_col_off{0}, read_qual_off{0};
    vgan::BigVec<uint16_t> read_algn_len;
    vgan::BigVec<uint8_t> read_mapq;
    vgan::BigVec<uint32_t> seg_node;
    vgan::BigVec<uint16_t> seg_start, seg_len;
    vgan::BigVec<uint8_t> graph_seq, algnseq, qual;
    vgan::BigVec<uint32_t> read_src; // index of each batch read in the alignment set
    uint32_t n_tileable = 0;        // reads [0, n_tileable) satisfy the tile contract (include/vgan_gpu.h)
    // vgan_hc_flatten*_packed: the reads that satisfy the tile contract in the segment kernel's own layout
    // (vgan_hc_packed_view); the SoA arrays above then hold the other reads alone
    bool is_packed = false;
    vgan::BigVec<uint32_t> pk_rhdr, pk_srec, pk_crec, pk_src;
    vgan::BigVec<uint8_t> pk_qualp;
    uint32_t pk_reads = 0, pk_segments = 0, pk_max_segs = 0, pk_max_qual = 0, pk_max_cols = 0, pk_max_span = 0;
    uint64_t pk_cols = 0, pk_qual = 0;
    void fill(vgan_hc_batch *b) const;
};
