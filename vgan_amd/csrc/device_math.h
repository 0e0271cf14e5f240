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
