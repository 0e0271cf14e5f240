// Device helpers shared by the kernels: wave64 reductions / DPP scan, base classification, a fast fp64 log.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "log_tab.h"

namespace vgan {

// libgab isValidDNA: upper-case A, C, G, T only.  Branch-free: 'A'-'A'=0, 'C'=2, 'G'=6, 'T'=19.
__device__ __forceinline__ bool is_acgt(uint32_t c) {
    const uint32_t d = c - 65u;
    return d < 20u && ((0x80045u >> d) & 1u);
}

__device__ __forceinline__ double bg_freq(uint32_t c) { // src/haplocart_functions.cpp:81-98
    return c == 'A' ? 0.27532 : c == 'C' ? 0.30044 : c == 'G' ? 0.16644 : c == 'T' ? 0.25780 : 0.25;
}

// Natural log in fp64, fdlibm e_log.c scheme (error < 1 ulp): x = 2^k * m, m in [sqrt(1/2), sqrt(2)), log(m) from the
// odd series in s = f/(2+f).  ~35 fp64 instructions instead of the ~100 of the device library's double-double log.
// Special values as log(): 0 -> -inf, negative / NaN -> NaN, +inf -> +inf, subnormals are rescaled.
__device__ __forceinline__ double log_pos(double x) {
    int kadj = 0;
    if (!(x >= 2.2250738585072014e-308)) { // zero, subnormal, negative or NaN
        if (x == 0.0) return -INFINITY;
        if (!(x > 0.0)) return __builtin_nan("");
        x *= 18014398509481984.0; // 2^54
        kadj = -54;
    }
    if (!(x <= 1.7976931348623157e308)) return x; // +inf
    double m = __builtin_amdgcn_frexp_mant(x); // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x) + kadj;
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double den = 2.0 + f;
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double sq = f * r;
    sq = fma(fma(-den, sq, f), r, sq); // s = f / (2 + f) to within an ulp
    const double z = sq * sq, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - fma(sq, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}

// log(1 + exp(-x)) for x >= 0 -- the second term of log-sum-exp -- cheaper than libm's exp and log1p one after the other (which
// also carry every special case of their full domains): exp(-x) = 2^k exp(r), |r| <= ln 2 / 2, by its series to r^13 (the next
// term is below 2e-17 relative); then with u = 1 + t rounded, log1p(t) = log(u) + (t - (u - 1)) / u, log(u) for u in [1, 2] by the
// series of log_pos without its domain checks.  Error ~2 ulp of a value <= ln 2.  x = +inf -> 0, NaN -> NaN.
__device__ __forceinline__ double softplus_neg(double x) {
    const double kd = rint(x * -1.4426950408889634); // k = round(-x / ln 2) <= 0
    double r = fma(kd, -6.93147180369123816490e-01, -x);
    r = fma(kd, -1.90821492927058770002e-10, r);
    double p = fma(r, 1.6059043836821613e-10, 2.08767569878681e-09); // 1/13!, 1/12!
    p = fma(r, p, 2.505210838544172e-08);                            // 1/11!
    p = fma(r, p, 2.755731922398589e-07);                            // 1/10!
    p = fma(r, p, 2.7557319223985893e-06);                           // 1/9!
    p = fma(r, p, 2.48015873015873e-05);                             // 1/8!
    p = fma(r, p, 1.984126984126984e-04);                            // 1/7!
    p = fma(r, p, 1.388888888888889e-03);                            // 1/6!
    p = fma(r, p, 8.333333333333333e-03);                            // 1/5!
    p = fma(r, p, 4.1666666666666664e-02);                           // 1/4!
    p = fma(r, p, 1.6666666666666666e-01);                           // 1/3!
    p = fma(r, p, 0.5);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    double t = __builtin_amdgcn_ldexp(p, (int)kd); // exp(-x); underflows through the subnormals to 0
    t = x > 800.0 ? 0.0 : t;                       // (k beyond the exponent range, +inf)
    // log1p(t), t in [0, 1]
    const double u = 1.0 + t;
    const bool hi = u > 1.4142135623730951; // m = u / 2 in [0.707, 1], k = 1; else m = u, k = 0
    const double m = hi ? 0.5 * u : u;
    const double f = m - 1.0;
    const double den = 2.0 + f;
    double q = __builtin_amdgcn_rcp(den);
    q = fma(fma(-den, q, 1.0), q, q);
    q = fma(fma(-den, q, 1.0), q, q);
    double sq = f * q;
    sq = fma(fma(-den, sq, f), q, sq); // s = f / (2 + f) to within an ulp
    const double z = sq * sq, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = hi ? 1.0 : 0.0;
    const double lg = dk * 6.93147180369123816490e-01 - ((hfsq - fma(sq, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
    // (1 + t was rounded: the part of t that u lost, over u; 1 / u from the reciprocal above when m = u, else refined anew)
    double ru = __builtin_amdgcn_rcp(u);
    ru = fma(fma(-u, ru, 1.0), ru, ru);
    return lg + (t - (u - 1.0)) * ru;
}

// the same out of line: for branches no lane takes in practice (inlined there, the series' constants are hoisted out of the
// caller's loops and held -- or spilled -- for the whole kernel)
__device__ __attribute__((noinline)) double log_pos_cold(double x) { return log_pos(x); }

// log(x) through the table staged in LDS (log_tab.h).  Lanes with `wanted` false may hold anything and get anything
// back; a wanted x outside the normal positive range takes the series above (a branch no lane takes in practice).
__device__ __forceinline__ double log_tab(double x, bool wanted, const LogTabEntry *tab_lds) {
    const bool ok = ((uint32_t)__double2hiint(x) - 0x00100000u) < 0x7FE00000u; // positive, normal, finite
    double r = log_tab_eval(x, tab_lds);
    if (__builtin_expect(wanted && !ok, 0)) r = log_pos_cold(x);
    return r;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

// DPP moves on a double (two 32-bit halves); lanes without a source get 0.0 (BOUND: by bound_ctrl, for a move within all
// rows -- no register to preset; otherwise by presetting the destination, for a move into some rows only).
template <int CTRL, int ROW_MASK, bool BOUND = false> __device__ __forceinline__ double dpp_mov0(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, BOUND);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, BOUND);
    return __hiloint2double(hi, lo);
}

// wave64 inclusive prefix sum with DPP row shifts + row broadcasts (no LDS traffic)
__device__ __forceinline__ double wave_incl_scan(double v) {
    v += dpp_mov0<0x111, 0xf, true>(v); // row_shr:1
    v += dpp_mov0<0x112, 0xf, true>(v); // row_shr:2
    v += dpp_mov0<0x114, 0xf, true>(v); // row_shr:4
    v += dpp_mov0<0x118, 0xf, true>(v); // row_shr:8
    v += dpp_mov0<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
    v += dpp_mov0<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3
    return v;
}


} // namespace vgan
