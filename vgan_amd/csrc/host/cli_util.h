// Small helpers shared by the subcommands of the vgan CLI (plain clients of the C-ABI, include/vgan_gpu.h).
#pragma once
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "vgan_gpu.h"

namespace vgan_cli {

[[noreturn]] inline void die(const std::string &msg) { throw std::runtime_error(msg); }

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

} // namespace vgan_cli
