// Error reporting, file and gzip helpers of the host front end.
#include "common.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>

#include <zlib.h>

#include <pthread.h>

#include <algorithm>
#include <exception>
#include <atomic>
#include <functional>
#include <deque>
#include <mutex>
#include <memory>
#include <condition_variable>
#include <cstring>
#include <fstream>
#include <sys/stat.h>
#include <thread>
#include <vector>

namespace vgan {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

const char *last_error() { return g_err; }

bool file_exists(const std::string &path) {
    struct stat st;
    return stat(path.c_str(), &st) == 0 && !S_ISDIR(st.st_mode);
}

// BGZF = gzip members with an extra subfield 'B','C' holding the member size - 1 (SAM spec 4.1).  Returns true and the
// block list when the whole buffer is a sequence of such members.
bool bgzf_index(const unsigned char *p, size_t n, std::vector<BgzfBlock> &blocks) {
    blocks.clear();
    size_t off = 0, out = 0;
    while (off < n) {
        if (n - off < 18 || p[off] != 0x1f || p[off + 1] != 0x8b || p[off + 2] != 8 || !(p[off + 3] & 4)) return false;
        const size_t xlen = p[off + 10] | (p[off + 11] << 8);
        if (n - off < 12 + xlen) return false;
        size_t bsize = 0;
        for (size_t x = off + 12; x + 4 <= off + 12 + xlen;) {
            const size_t slen = p[x + 2] | (p[x + 3] << 8);
            if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= off + 12 + xlen) bsize = (size_t)(p[x + 4] | (p[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || bsize > n - off) return false;
        const unsigned char *tail = p + off + bsize - 4;
        const size_t isize = (size_t)tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
        blocks.push_back({off, bsize, out, isize});
        off += bsize;
        out += isize;
    }
    return true;
}

// libdeflate (what htslib inflates BGZF with: 2-3x zlib's rate on these 64 KB members), when the system has its shared
// object: loaded at run time, by its four stable entry points; zlib does the work otherwise, with the same results.
namespace {
struct FastInflate {
    void *(*alloc)() = nullptr;
    int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    FastInflate() {
        if (getenv("VGAN_NO_LIBDEFLATE")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        gunzip = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_gzip_decompress");
        release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        if (!alloc || !gunzip || !release) alloc = nullptr;
    }
};
const FastInflate &fast_inflate() {
    static const FastInflate f;
    return f;
}
struct ThreadDecompressor {
    void *d = nullptr;
    ~ThreadDecompressor() {
        if (d) fast_inflate().release(d);
    }
};
} // namespace

bool inflate_member(const unsigned char *in, size_t in_size, unsigned char *out, size_t out_size) {
    unsigned char scratch[8]; // an empty member (the BGZF end-of-file block; a whole GAM without reads is just that block) still
                              // needs somewhere to "write": zlib rejects a null next_out and cannot finish with no room at all
    const FastInflate &f = fast_inflate();
    if (f.alloc) {
        static thread_local ThreadDecompressor td;
        if (!td.d) td.d = f.alloc();
        if (td.d) {
            size_t got = 0;
            const int rc = f.gunzip(td.d, in, in_size, out_size ? out : scratch, out_size ? out_size : sizeof scratch, &got);
            return rc == 0 && got == out_size;
        }
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return false;
    zs.next_in = const_cast<unsigned char *>(in);
    zs.avail_in = (uInt)in_size;
    zs.next_out = out_size ? out : scratch;
    zs.avail_out = out_size ? (uInt)out_size : (uInt)sizeof scratch;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == out_size;
    inflateEnd(&zs);
    return ok;
}

// Inflates a concatenation of gzip members.  BGZF input (what vg writes) is inflated block-parallel straight into its
// final position; other gzip streams sequentially.
bool gunzip_members(const void *data, size_t n, ByteBuf &out) {
    const unsigned char *p = (const unsigned char *)data;
    out.clear();
    std::vector<BgzfBlock> blocks;
    if (bgzf_index(p, n, blocks)) {
        const size_t total = blocks.empty() ? 0 : blocks.back().out_off + blocks.back().out_size;
        out.resize(total);
        unsigned nt = usable_cpus();
        nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, blocks.size() / 64));
        std::atomic<bool> ok{true};
        auto work = [&](size_t b0, size_t b1) {
            for (size_t i = b0; i < b1 && ok.load(std::memory_order_relaxed); ++i)
                if (!inflate_member(p + blocks[i].in_off, blocks[i].in_size, (unsigned char *)out.data() + blocks[i].out_off, blocks[i].out_size))
                    ok = false;
        };
        if (nt <= 1) {
            work(0, blocks.size());
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, blocks.size() * t / nt, blocks.size() * (t + 1) / nt);
            for (auto &t : th) t.join();
        }
        return ok;
    }
    out.reserve(n * 4 + 1024);
    std::vector<unsigned char> buf(1 << 20);
    while (n > 0) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return false;
        zs.next_in = const_cast<unsigned char *>(p);
        zs.avail_in = (uInt)std::min<size_t>(n, 0x7fffffffu);
        int rc;
        do {
            zs.next_out = buf.data();
            zs.avail_out = (uInt)buf.size();
            rc = inflate(&zs, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) {
                inflateEnd(&zs);
                return false;
            }
            out.append((const char *)buf.data(), buf.size() - zs.avail_out);
            if (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0) { // truncated member
                inflateEnd(&zs);
                return false;
            }
        } while (rc != Z_STREAM_END);
        const size_t used = (size_t)(zs.next_in - p);
        inflateEnd(&zs);
        p += used;
        n -= used;
    }
    return true;
}

struct AsyncInflate::Impl {
    std::vector<BgzfBlock> blocks;
    const unsigned char *in = nullptr;
    std::unique_ptr<std::atomic<uint8_t>[]> done;
    std::atomic<size_t> next{0};
    std::atomic<size_t> avail{0};
    std::atomic<bool> failed{false};
    size_t prefix = 0; // blocks [0, prefix) are done (under mu)
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::thread> th;
};

AsyncInflate::AsyncInflate() : impl(new Impl()) {}

AsyncInflate::~AsyncInflate() {
    (void)finish();
    delete impl;
}

bool AsyncInflate::start(const void *data, size_t n) {
    Impl &m = *impl;
    m.in = (const unsigned char *)data;
    if (!bgzf_index(m.in, n, m.blocks) || m.blocks.empty()) return false;
    out.resize(m.blocks.back().out_off + m.blocks.back().out_size);
    m.done.reset(new std::atomic<uint8_t>[m.blocks.size()]);
    for (size_t i = 0; i < m.blocks.size(); ++i) m.done[i].store(0, std::memory_order_relaxed);
    unsigned nt = usable_cpus();
    nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, m.blocks.size() / 64));
    auto work = [this]() {
        Impl &m = *impl;
        constexpr size_t RUN = 32; // blocks per grab (~2 MB of output: a thread first-touches whole huge pages), in file order
        for (;;) {
            const size_t i0 = m.next.fetch_add(RUN, std::memory_order_relaxed);
            if (i0 >= m.blocks.size() || m.failed.load(std::memory_order_relaxed)) break;
            const size_t i1 = std::min(m.blocks.size(), i0 + RUN);
            for (size_t i = i0; i < i1; ++i) {
                const BgzfBlock &b = m.blocks[i];
                if (!inflate_member(m.in + b.in_off, b.in_size, (unsigned char *)out.data() + b.out_off, b.out_size)) {
                    m.failed = true;
                    std::lock_guard<std::mutex> lk(m.mu);
                    m.cv.notify_all();
                    return;
                }
                m.done[i].store(1, std::memory_order_release);
            }
            std::lock_guard<std::mutex> lk(m.mu);
            size_t p = m.prefix;
            while (p < m.blocks.size() && m.done[p].load(std::memory_order_acquire)) ++p;
            if (p != m.prefix) {
                m.prefix = p;
                m.avail.store(m.blocks[p - 1].out_off + m.blocks[p - 1].out_size, std::memory_order_release);
                m.cv.notify_all();
            }
        }
    };
    for (unsigned t = 0; t < nt; ++t) m.th.emplace_back(work);
    return true;
}

bool AsyncInflate::wait_for(size_t upto) {
    Impl &m = *impl;
    upto = std::min(upto, out.size());
    if (m.avail.load(std::memory_order_acquire) >= upto) return !m.failed.load();
    std::unique_lock<std::mutex> lk(m.mu);
    m.cv.wait(lk, [&] { return m.avail.load(std::memory_order_acquire) >= upto || m.failed.load(); });
    return !m.failed.load();
}

bool AsyncInflate::finish() {
    Impl &m = *impl;
    for (auto &t : m.th)
        if (t.joinable()) t.join();
    m.th.clear();
    return !m.failed.load();
}

bool gunzip_members(const void *data, size_t n, std::string &out) {
    ByteBuf b;
    if (!gunzip_members(data, n, b)) return false;
    out.assign(b.data(), b.size());
    return true;
}

// BGZF writer: blocks of at most 0xff00 input bytes, deflated in parallel, plus the 28-byte EOF marker block.
bool gzip_bytes(const std::string &in, std::string &out) {
    const size_t BLK = 0xff00;
    const size_t nb = (in.size() + BLK - 1) / BLK;
    std::vector<std::string> parts(nb);
    std::atomic<bool> ok{true};
    auto work = [&](size_t b0, size_t b1) {
        std::vector<unsigned char> buf(BLK + 1024);
        for (size_t b = b0; b < b1; ++b) {
            const size_t off = b * BLK, len = std::min(BLK, in.size() - off);
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
                ok = false;
                return;
            }
            zs.next_in = (unsigned char *)in.data() + off;
            zs.avail_in = (uInt)len;
            zs.next_out = buf.data();
            zs.avail_out = (uInt)buf.size();
            const int rc = deflate(&zs, Z_FINISH);
            const size_t clen = zs.total_out;
            deflateEnd(&zs);
            if (rc != Z_STREAM_END || clen + 26 > 65536) {
                ok = false;
                return;
            }
            std::string &o = parts[b];
            const size_t bsize = clen + 26;
            const unsigned char hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                                           (unsigned char)((bsize - 1) & 0xff), (unsigned char)((bsize - 1) >> 8)};
            o.assign((const char *)hdr, 18);
            o.append((const char *)buf.data(), clen);
            const uLong crc = crc32(crc32(0L, Z_NULL, 0), (const Bytef *)in.data() + off, (uInt)len);
            unsigned char tail[8];
            for (int i = 0; i < 4; ++i) {
                tail[i] = (unsigned char)((crc >> (8 * i)) & 0xff);
                tail[4 + i] = (unsigned char)((len >> (8 * i)) & 0xff);
            }
            o.append((const char *)tail, 8);
        }
    };
    unsigned nt = usable_cpus();
    nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, nb / 16));
    if (nt <= 1) {
        work(0, nb);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, nb * t / nt, nb * (t + 1) / nt);
        for (auto &t : th) t.join();
    }
    if (!ok) return false;
    out.clear();
    size_t tot = 28;
    for (auto &s : parts) tot += s.size();
    out.reserve(tot);
    for (auto &s : parts) out += s;
    static const unsigned char eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    out.append((const char *)eof, 28);
    return true;
}

bool read_file(const std::string &path, std::string &out, bool inflate_if_gzip) {
    const int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    std::string raw;
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) raw.reserve((size_t)st.st_size);
    std::vector<char> buf(1 << 20); // regular files, FIFOs and pipes alike
    for (;;) {
        const ssize_t k = read(fd, buf.data(), buf.size());
        if (k < 0) {
            if (errno == EINTR) continue;
            close(fd);
            return false;
        }
        if (k == 0) break;
        raw.append(buf.data(), (size_t)k);
    }
    close(fd);
    if (inflate_if_gzip && raw.size() >= 2 && (unsigned char)raw[0] == 0x1f && (unsigned char)raw[1] == 0x8b) {
        return gunzip_members(raw.data(), raw.size(), out);
    }
    out.swap(raw);
    return true;
}

CpuAccount &cpu_account() {
    static CpuAccount a;
    return a;
}

unsigned usable_cpus() {
    static const unsigned n = [] {
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::min(hw, (unsigned)std::max(1, CPU_COUNT(&set)));
        double quota = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "max 100000" or "<quota> <period>"
            char q[64];
            long period = 0;
            if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) quota = atof(q) / (double)period;
            fclose(f);
        } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { // cgroup v1
            long q = -1, period = 0;
            if (fscanf(f1, "%ld", &q) != 1) q = -1;
            fclose(f1);
            if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(f2, "%ld", &period) != 1) period = 0;
                fclose(f2);
            }
            if (q > 0 && period > 0) quota = (double)q / (double)period;
        }
        if (quota > 0) hw = std::min(hw, (unsigned)std::max(1.0, quota + 0.5));
        if (const char *e = getenv("VGAN_CPUS")) hw = (unsigned)std::max(1, atoi(e));
        return hw;
    }();
    return n;
}

unsigned burst_cpus() {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::min(hw, (unsigned)std::max(1, CPU_COUNT(&set)));
    if (getenv("VGAN_CPUS")) return usable_cpus();
    return std::min(hw, 4 * usable_cpus());
}

} // namespace vgan
extern "C" int vgan_host_cpus(void) { return (int)vgan::usable_cpus(); }
extern "C" void vgan_host_cpu_account(int64_t out[4]) {
    if (!out) return;
    vgan::CpuAccount &a = vgan::cpu_account();
    out[0] = a.inflate;
    out[1] = a.frame_parse;
    out[2] = a.flatten;
    out[3] = a.merge;
}
namespace vgan {

namespace {
struct WorkerPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    size_t n_threads = 0, idle = 0;
    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            ++idle;
            cv.wait(lk, [&] { return !q.empty(); });
            --idle;
            std::function<void()> f = std::move(q.front());
            q.pop_front();
            lk.unlock();
            f();
            lk.lock();
        }
    }
    void submit(std::function<void()> f) {
        std::lock_guard<std::mutex> lk(mu);
        q.push_back(std::move(f));
        if (idle < q.size() && n_threads < 512) { // one more worker: they stay for the life of the process
            ++n_threads;
            std::thread([this] { worker(); }).detach();
        }
        cv.notify_one();
    }
};
WorkerPool *g_pool = nullptr;
// A child of fork() inherits the pool's counters and none of its threads (Python's multiprocessing forks by default): it
// starts with a pool of its own -- the parent's object is left as it is, its mutex may be held by a thread that is not there.
void pool_after_fork_in_child() { g_pool = new WorkerPool; }
WorkerPool &worker_pool() {
    static const bool once = [] {
        g_pool = new WorkerPool; // never destroyed: its threads outlive static destruction
        (void)pthread_atfork(nullptr, nullptr, pool_after_fork_in_child);
        return true;
    }();
    (void)once;
    return *g_pool;
}
} // namespace

void parallel_run(int n, const std::function<void(int)> &fn) {
    if (n <= 1) {
        fn(0);
        return;
    }
    struct Job {
        std::mutex mu;
        std::condition_variable cv;
        int left;
        std::exception_ptr err; // the first exception of any share: rethrown on the caller once every share has ended
    } job;
    job.left = n - 1;
    WorkerPool &wp = worker_pool();
    for (int t = 1; t < n; ++t)
        wp.submit([&job, &fn, t] {
            std::exception_ptr e;
            try {
                fn(t);
            } catch (...) {
                e = std::current_exception();
            }
            std::lock_guard<std::mutex> lk(job.mu); // held while notifying: the waiter cannot leave (and destroy job) before
            if (e && !job.err) job.err = e;
            if (--job.left == 0) job.cv.notify_one();
        });
    std::exception_ptr mine;
    try {
        fn(0);
    } catch (...) {
        mine = std::current_exception();
    }
    std::unique_lock<std::mutex> lk(job.mu); // (always: the workers hold references to job and fn)
    job.cv.wait(lk, [&] { return job.left == 0; });
    if (mine) std::rethrow_exception(mine);
    if (job.err) std::rethrow_exception(job.err);
}

// Large blocks are recycled, not returned: every munmap interrupts all the cores the process runs on and every fresh mapping
// is faulted in page by page, and the front end allocates a few hundred large arrays per chunk of reads from up to 128 threads
// (freeing one parsed chunk of 262k reads cost 55-110 ms on the 256-core box; freeing it on a thread of its own, beside the
// flatten threads' page faults, made a run take anything from 0.7 to 4 s).  Size classes 2^k and 1.5 * 2^k from 64 KB up;
// a freed block waits in its class for the next request, up to a cap (pool_cap), beyond which it goes back to the system.
namespace {
constexpr size_t POOL_MIN = 64u << 10;
constexpr size_t POOL_MAX = 256u << 20; // larger blocks (a whole inflated input) are one of a kind: not kept
// at most 12 GB and an eighth of the machine's memory wait in the pool
static size_t pool_cap() {
    static const size_t cap = [] {
        const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
        const size_t ram = pages > 0 && psz > 0 ? (size_t)pages * (size_t)psz : (size_t)64 << 30;
        if (const char *e = getenv("VGAN_POOL_CAP_MB")) return (size_t)std::max(64L, atol(e)) << 20; // developer aid
        return std::min((size_t)12 << 30, ram / 8);
    }();
    return cap;
}
struct BlockPool {
    std::mutex mu;
    std::vector<void *> free_blocks[2 * 48];
    size_t held = 0;
};
BlockPool &block_pool() {
    static BlockPool *p = new BlockPool; // never destroyed: blocks are handed back during static destruction too
    return *p;
}
// smallest class >= bytes: index 2k for 2^k, 2k + 1 for 1.5 * 2^k
inline size_t pool_class(size_t bytes, size_t *cls_bytes) {
    size_t k = 16;
    while (((size_t)1 << k) < bytes && k < 47) ++k;
    if (k > 16 && ((size_t)3 << (k - 2)) >= bytes) { // 1.5 * 2^(k-1)
        *cls_bytes = (size_t)3 << (k - 2);
        return 2 * (k - 1) + 1;
    }
    *cls_bytes = (size_t)1 << k;
    return 2 * k;
}
void *raw_big_alloc(size_t bytes) {
    constexpr size_t HUGE = 2u << 20;
    void *p = nullptr;
    if (bytes >= HUGE) {
        const size_t rounded = (bytes + HUGE - 1) / HUGE * HUGE;
        if (posix_memalign(&p, HUGE, rounded) != 0) p = nullptr;
        if (p) (void)madvise(p, rounded, MADV_HUGEPAGE);
    } else {
        p = malloc(bytes ? bytes : 1);
    }
    return p;
}
} // namespace

void *big_alloc_bytes(size_t bytes) {
    void *p = nullptr;
    if (bytes >= POOL_MIN && bytes <= POOL_MAX) {
        size_t cb;
        const size_t c = pool_class(bytes, &cb);
        BlockPool &bp = block_pool();
        {
            std::lock_guard<std::mutex> lk(bp.mu);
            auto &v = bp.free_blocks[c];
            if (!v.empty()) {
                p = v.back();
                v.pop_back();
                bp.held -= cb;
            }
        }
        if (!p) p = raw_big_alloc(cb);
    } else {
        p = raw_big_alloc(bytes);
    }
    if (!p) throw std::bad_alloc();
    return p;
}

// every recycled block back to the system, on n_threads threads (the kernel unmaps disjoint ranges in parallel)
void big_pool_release(int n_threads) {
    std::vector<void *> all;
    {
        BlockPool &bp = block_pool();
        std::lock_guard<std::mutex> lk(bp.mu);
        for (auto &v : bp.free_blocks) {
            all.insert(all.end(), v.begin(), v.end());
            v.clear();
        }
        bp.held = 0;
    }
    n_threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, n_threads), all.size()));
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= all.size()) break;
            free(all[i]);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

void big_free_bytes(void *p, size_t bytes) {
    if (!p) return;
    if (bytes >= POOL_MIN && bytes <= POOL_MAX) {
        size_t cb;
        const size_t c = pool_class(bytes, &cb);
        BlockPool &bp = block_pool();
        std::lock_guard<std::mutex> lk(bp.mu);
        if (bp.held + cb <= pool_cap()) {
            bp.free_blocks[c].push_back(p);
            bp.held += cb;
            return;
        }
    }
    free(p);
}

MappedFile::~MappedFile() {
    if (p && p != MAP_FAILED && mapped) munmap(const_cast<void *>(p), n);
}

// Maps a regular file read-only (no copy); anything else (FIFO, pipe, empty file) is read into `fallback`.
bool MappedFile::open_path(const std::string &path) {
    const int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
            p = m;
            n = (size_t)st.st_size;
            mapped = true;
            close(fd);
            return true;
        }
    }
    close(fd);
    if (!read_file(path, fallback, false)) return false;
    p = fallback.data();
    n = fallback.size();
    mapped = false;
    return true;
}

bool write_file(const std::string &path, const std::string &bytes) {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f.write(bytes.data(), (std::streamsize)bytes.size());
    return (bool)f;
}

// large sidecars: straight into a buffer that is neither zero-filled nor copied (gzip members inflated in parallel)
bool read_bytes_maybe_gz(const std::string &path, ByteBuf &out) {
    std::string p = path;
    if (!file_exists(p)) p += ".gz";
    if (!file_exists(p)) return false;
    MappedFile f;
    if (!f.open_path(p)) return false;
    const unsigned char *b = (const unsigned char *)f.p;
    if (f.n >= 2 && b[0] == 0x1f && b[1] == 0x8b) return gunzip_members(f.p, f.n, out);
    out.assign((const char *)f.p, f.n);
    return true;
}

bool read_text_maybe_gz(const std::string &path, std::string &out) {
    if (file_exists(path)) return read_file(path, out);
    if (file_exists(path + ".gz")) return read_file(path + ".gz", out);
    return false;
}

} // namespace vgan

extern "C" const char *vgan_last_error(void) { return vgan::last_error(); }
extern "C" int vgan_abi_version(void) { return VGAN_ABI_VERSION; }
extern "C" void vgan_host_release_memory(int n_threads) { vgan::big_pool_release(n_threads > 0 ? n_threads : 16); }
