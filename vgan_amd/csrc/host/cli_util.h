// Small helpers shared by the subcommands of the vgan CLI (plain clients of the C-ABI, include/vgan_gpu.h).
#pragma once
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <fcntl.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>
#include <iostream>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>

#include "vgan_gpu.h"

namespace vgan_cli {

[[noreturn]] inline void die(const std::string &msg) { throw std::runtime_error(msg); }

// The HIP runtime comes up on a thread of its own while the tables are read (its failure shows at context creation), and behind it the
// code objects of the run's kernels are loaded beside each other (vgan_device_preload; VGAN_NO_PRELOAD=1: left to their first launches).
struct DeviceWarm {
    std::thread t, pre;
    std::function<void()> on_up; // (called on the warm-up's thread when the runtime is up)
    void start(int device, unsigned what) {
        t = std::thread([this, device, what] {
            const int rc = vgan_device_warmup(device);
            if (on_up) on_up();
            if (rc >= 0 && !getenv("VGAN_NO_PRELOAD")) pre = std::thread([device, what] { (void)vgan_device_preload(device, what); });
        });
    }
    void wait_runtime() { // (the runtime is up -- or has failed; the code objects may still be loading)
        if (t.joinable()) t.join();
    }
    ~DeviceWarm() {
        if (t.joinable()) t.join();
        if (pre.joinable()) pre.join();
    }
};

// ---- leaving without waiting for the teardown ------------------------------------------------------------------------
// A run leaves gigabytes resident and a GPU context behind; the kernel takes them apart inside the process's exit
// (0.2-0.6 s for the 3-4 GB of a million reads), after every output has been written.  So the work runs in a CHILD process
// (forked before anything touches the GPU or starts a thread); when its outputs are flushed it sends its exit code through a
// pipe, lets go of stdin / stdout / stderr and exits -- and the parent, which did nothing but wait for that byte, leaves with
// the code at once while the child's exit is still freeing memory.  (mold, the linker, returns the same way.)  A child that
// dies without a result is waited for and its fate passed on; a parent that dies takes the child with it (PR_SET_PDEATHSIG)
// until the result is out.  OPT-IN (VGAN_EARLY_LEAVE=1): by default the subcommand runs in the calling process, so that the
// exit code arrives when the memory and the GPU context are released and `time` / wait4 account for the worker itself.
// Never done under tools that preload into the process (a profiler's library has initialised the GPU before main: such a
// process must not fork).
struct EarlyLeave {
    int fd = -1; // write end of the result pipe (in the child)
    static EarlyLeave &get() {
        static EarlyLeave e;
        return e;
    }
    // In the parent: does not return (leaves with the child's code).  In the child, or when forking is off: returns.
    void start() {
        const char *want = getenv("VGAN_EARLY_LEAVE");
        if (!want || !*want || strcmp(want, "0") == 0) return;
        if (getenv("VGAN_NO_FORK") || getenv("VGAN_KEEP_TEARDOWN") || getenv("HSA_TOOLS_LIB") || getenv("ROCP_TOOL_LIB") || getenv("ROCPROFILER_LIBRARY_PATH"))
            return;
        if (const char *pre = getenv("LD_PRELOAD")) // a profiler's tool library in the process: the GPU is up before main
            if (strstr(pre, "rocprof") || strstr(pre, "roctracer") || strstr(pre, "rocprofiler")) return;
        int pfd[2];
        if (pipe2(pfd, O_CLOEXEC) != 0) return;
        fflush(nullptr);
        const pid_t parent = getpid(); // (vgan itself may be pid 1 -- `docker run image vgan ...` -- or sit under a sub-reaper)
        const pid_t pid = fork();
        if (pid < 0) {
            close(pfd[0]);
            close(pfd[1]);
            return;
        }
        if (pid == 0) {
            close(pfd[0]);
            (void)prctl(PR_SET_PDEATHSIG, SIGKILL);
            if (getppid() != parent) _exit(1); // the parent is gone already
            fd = pfd[1];
            return;
        }
        close(pfd[1]);
        unsigned char code = 0;
        ssize_t k;
        do k = read(pfd[0], &code, 1);
        while (k < 0 && errno == EINTR);
        if (k == 1) _exit(code);
        int st = 0; // no result: the child was killed or crashed
        while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {
        }
        if (WIFSIGNALED(st)) {
            signal(WTERMSIG(st), SIG_DFL);
            raise(WTERMSIG(st));
        }
        _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 1);
    }
    // Every output is written: flush, hand the code over, leave.
    [[noreturn]] void finish(int code) {
        std::cout.flush();
        std::cerr.flush();
        fflush(nullptr);
        if (fd >= 0) {
            (void)prctl(PR_SET_PDEATHSIG, 0);
            const unsigned char b = (unsigned char)code;
            ssize_t k;
            do k = write(fd, &b, 1);
            while (k < 0 && errno == EINTR);
            close(fd);
            close(0);
            close(1);
            close(2);
        }
        _exit(code);
    }
};

inline int parse_int(const std::string &v, const char *flag, const char *tool);
// "0,1,3" -> GPU indices appended to out (--gpus / VGAN_GPUS)
inline void parse_gpu_list(const std::string &v, const char *flag, const char *tool, std::vector<int> &out) {
    size_t p0 = 0;
    while (p0 <= v.size()) {
        size_t c1 = v.find(',', p0);
        if (c1 == std::string::npos) c1 = v.size();
        const int d = parse_int(v.substr(p0, c1 - p0), flag, tool);
        if (d < 0) die(std::string(tool) + " Error, " + flag + " needs non-negative GPU indices");
        out.push_back(d);
        p0 = c1 + 1;
    }
}

inline void check(int rc, const char *what) {
    if (rc < 0) die(std::string("[vgan] ") + what + ": " + vgan_last_error());
}

// whole-token numeric parses: "12x", "" and out-of-range values are errors with the option named, not an uncaught stoi
inline int parse_int(const std::string &v, const char *flag, const char *tool) {
    errno = 0;
    char *end = nullptr;
    const long x = std::strtol(v.c_str(), &end, 10);
    if (v.empty() || *end != '\0' || errno == ERANGE || x < INT32_MIN || x > INT32_MAX)
        die(std::string(tool) + " Error, option " + flag + " needs an integer, got '" + v + "'");
    return (int)x;
}

inline double parse_double(const std::string &v, const char *flag, const char *tool) {
    errno = 0;
    char *end = nullptr;
    const double x = std::strtod(v.c_str(), &end);
    if (v.empty() || *end != '\0' || errno == ERANGE || !(x == x))
        die(std::string(tool) + " Error, option " + flag + " needs a number, got '" + v + "'");
    return x;
}

inline bool is_file(const std::string &p) {
    struct stat sb;
    return stat(p.c_str(), &sb) == 0 && S_ISREG(sb.st_mode);
}

// an input the caller may also hand over as a FIFO or /dev/stdin (the reference feeds its own GAM reader through a FIFO)
inline bool is_readable_input(const std::string &p) {
    struct stat sb;
    return stat(p.c_str(), &sb) == 0 && !S_ISDIR(sb.st_mode);
}

inline bool ends_with(const std::string &s, const char *suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// the reference's subcommands take FASTQ only (Euka.cpp:213-222, soibean.cpp:286-296)
inline void reject_fasta(const std::string &f, const char *tool) {
    for (const char *suf : {".fa", ".fasta", ".fa.gz", ".fasta.gz"})
        if (ends_with(f, suf)) die(std::string(tool) + " Input file must be FASTQ, not FASTA");
}

// frees a C-ABI object on every exit path
template <class T> struct Handle {
    T *p = nullptr;
    void (*release)(T *);
    explicit Handle(void (*r)(T *)) : release(r) {}
    ~Handle() {
        if (p) release(p);
    }
    Handle(const Handle &) = delete;
    Handle &operator=(const Handle &) = delete;
};

// Decodes a GAM on a thread of its own while the caller loads tables and brings the device up (about 0.2 s of HIP
// initialisation that would otherwise sit in front of the decode).
class GamReader {
  public:
    void start(const std::string &path, int keep_unmapped) {
        th_ = std::thread([this, path, keep_unmapped] {
            rc_ = vgan_aln_read_gam(path.c_str(), keep_unmapped, &aln_);
            if (rc_ < 0) err_ = vgan_last_error(); // the message is thread local: carry it over
        });
    }
    vgan_alnset *take() { // joins; throws what the reader reported
        if (th_.joinable()) th_.join();
        if (rc_ < 0) die("[vgan] reading GAM: " + err_);
        vgan_alnset *a = aln_;
        aln_ = nullptr;
        return a;
    }
    ~GamReader() {
        if (th_.joinable()) th_.join();
        if (aln_) vgan_aln_free(aln_);
    }

  private:
    std::thread th_;
    vgan_alnset *aln_ = nullptr;
    int rc_ = 0;
    std::string err_;
};

// phase times to stderr when VGAN_TIMING is set (developer aid)
struct PhaseTimer {
    const char *tool;
    bool on = getenv("VGAN_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit PhaseTimer(const char *t) : tool(t) {}
    void lap(const char *phase) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[vgan timing] %s: %s %.1f ms\n", tool, phase, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// <prefix>.gbwt beside a graph read from <prefix>.og: the reference loads it and walks every path (readOG_Euka.h:36-74) but
// reads the result only for an emptiness check; an index that is there and cannot be read ends the run, as it does there.
inline void check_gbwt_beside(const std::string &prefix, int64_t n_nodes, int64_t n_paths, const char *tool) {
    const std::string f = prefix + ".gbwt";
    if (!std::ifstream(f)) return;
    vgan_gbwt *gb = nullptr;
    if (vgan_gbwt_load(f.c_str(), &gb) < 0) die(std::string(tool) + " Error, cannot read " + f + ": " + vgan_last_error());
    // The reference fills a nodes x paths matrix here and never reads it (the assignment into NodeInfo::pathsgo is commented
    // out): the walk is what can fail, so the walk is what is repeated -- one thread at a time into a buffer that grows with
    // the longest thread, not a dense matrix (a node count times a path count of bytes: tens of GB on a large graph).
    const int64_t n_seq = std::min<int64_t>(vgan_gbwt_sequences(gb), std::max<int64_t>(0, n_paths));
    std::vector<uint64_t> nodes(1 << 16);
    int64_t rc = 0;
    for (int64_t s = 0; s < n_seq && rc >= 0; ++s) {
        rc = vgan_gbwt_extract(gb, s, nodes.data(), (int64_t)nodes.size());
        if (rc > (int64_t)nodes.size()) { // the thread is longer than the buffer: once more with room for it
            nodes.resize((size_t)rc);
            rc = vgan_gbwt_extract(gb, s, nodes.data(), (int64_t)nodes.size());
        }
    }
    vgan_gbwt_free(gb);
    if (rc < 0) die(std::string(tool) + " Error, cannot walk the paths of " + f + ": " + vgan_last_error());
    if (n_nodes == 0) die("Error: The node_path_matrix is empty. Unable to proceed."); // soibean.cpp:453-455
}

} // namespace vgan_cli
