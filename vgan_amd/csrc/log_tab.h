// Table-driven fp64 natural log shared by the device kernels and the CPU unit test (tests/test_log_tab_cpu.py).
//
// x = 2^k * m, the high word of m in [VGAN_LOG_OFF, VGAN_LOG_OFF + 2^20) (m in [0.707, 1.414)); the top 6 bits of that
// range select {rcp, logc} with ln(m) = logc + log1p(r), r = fma(m, rcp, -1), |r| <= 2^-7; log1p by a degree-7 series
// (next term r^8/8 < 2e-18).  ~20 fp64/int instructions and one 16-byte table read instead of ~45 for the
// division-based series.  Error below 4 ulp, largest just outside the bin centred on 1 (checked against logl in tests/test_log_tab_cpu.py); exact 0 at x = 1.
// Domain: normal positive finite x (callers route everything else to log_pos()).
#pragma once
#include <stdint.h>
#include <string.h>

#include "log_table.h"

#if defined(__HIPCC__)
#define VGAN_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define VGAN_HD inline
#endif

namespace vgan {

struct alignas(16) LogTabEntry {
    double rcp, logc;
};

VGAN_HD bool log_tab_in_domain(double x) { return x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308; }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double log_tab_fma3(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// the same with every constant addend in a scalar register pair (an instruction takes one scalar operand): a kernel short of
// vector registers keeps none of the series' constants in them (hc_wave_kernels.hip at five waves per SIMD)
__device__ __forceinline__ double log_tab_fma3s(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
__device__ __forceinline__ double log_tab_eval_s(double x, const LogTabEntry *tab) {
    uint64_t bits;
    memcpy(&bits, &x, 8);
    const uint32_t lx = (uint32_t)bits;
    const uint32_t h = (uint32_t)(bits >> 32) + (0x3FF00000u - VGAN_LOG_OFF);
    const int k = (int)(h >> 20) - 1023;
    const uint32_t idx = (h >> 14) & 63u;
    const uint64_t mbits = ((uint64_t)((h & 0x000FFFFFu) + VGAN_LOG_OFF) << 32) | lx;
    double m;
    memcpy(&m, &mbits, 8);
    const LogTabEntry e = tab[idx];
    const double r = __builtin_fma(m, e.rcp, -1.0);
    // Horner from the top with one scalar constant per instruction: q = r / 7 - 1 / 6 costs a multiply and an add
    double t = r * (1.0 / 7.0);
    t = t + (-1.0 / 6.0);
    t = log_tab_fma3s(r, t, 0.2);
    t = log_tab_fma3s(r, t, -0.25);
    t = log_tab_fma3s(r, t, 1.0 / 3.0);
    t = __builtin_fma(r, t, -0.5);
    const double p = __builtin_fma(r * r, t, r);
    const double dk = (double)k;
    return __builtin_fma(dk, 6.93147180369123816490e-01, e.logc) + __builtin_fma(dk, 1.90821492927058770002e-10, p);
}
#else
VGAN_HD double log_tab_eval(double x, const LogTabEntry *tab);
inline double log_tab_eval_s(double x, const LogTabEntry *tab) { return log_tab_eval(x, tab); } // (host pass of a .hip file)
#endif

VGAN_HD double log_tab_eval(double x, const LogTabEntry *tab) {
    uint64_t bits;
    memcpy(&bits, &x, 8);
    const uint32_t lx = (uint32_t)bits;
    const uint32_t h = (uint32_t)(bits >> 32) + (0x3FF00000u - VGAN_LOG_OFF);
    const int k = (int)(h >> 20) - 1023;
    const uint32_t idx = (h >> 14) & 63u;
    const uint64_t mbits = ((uint64_t)((h & 0x000FFFFFu) + VGAN_LOG_OFF) << 32) | lx;
    double m;
    memcpy(&m, &mbits, 8);
    const LogTabEntry e = tab[idx];
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = __builtin_fma(m, e.rcp, -1.0);
    // (three-address v_fma_f64 spelled out: with a constant addend the compiler emits a copy of it + v_fmac_f64)
    double q = log_tab_fma3(r, 1.0 / 7.0, -1.0 / 6.0);
    q = log_tab_fma3(r, q, 0.2);
    q = log_tab_fma3(r, q, -0.25);
    q = log_tab_fma3(r, q, 1.0 / 3.0);
    q = __builtin_fma(r, q, -0.5);
    const double p = __builtin_fma(r * r, q, r);
    const double dk = (double)k;
    return __builtin_fma(dk, 6.93147180369123816490e-01, e.logc) + __builtin_fma(dk, 1.90821492927058770002e-10, p);
#else
    const double r = std::fma(m, e.rcp, -1.0);
    double q = std::fma(r, 1.0 / 7.0, -1.0 / 6.0);
    q = std::fma(r, q, 0.2);
    q = std::fma(r, q, -0.25);
    q = std::fma(r, q, 1.0 / 3.0);
    q = std::fma(r, q, -0.5);
    const double p = std::fma(r * r, q, r);
    const double dk = (double)k;
    return std::fma(dk, 6.93147180369123816490e-01, e.logc) + std::fma(dk, 1.90821492927058770002e-10, p);
#endif
}

} // namespace vgan
