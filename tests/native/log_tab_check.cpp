// CPU check of vgan_amd/csrc/log_tab.h (the table-driven log the HaploCart column kernel evaluates in LDS) against
// logl: prints the worst error in ulps of the result and the worst error relative to max(1, |ln x|); tests/test_log_tab_cpu.py asserts.
#include "log_tab.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

static uint64_t sm_state = 0x76676131ull;
static uint64_t sm() {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double u01() { return (double)(sm() >> 11) * 0x1p-53; }

int main(int argc, char **argv) {
    static const vgan::LogTabEntry tab[64] = {VGAN_LOG_TABLE};
    const long n = argc > 1 ? atol(argv[1]) : 4000000;
    double worst_ulp = 0, worst_abs = 0;
    auto check = [&](double x) {
        if (!vgan::log_tab_in_domain(x)) return;
        const long double ref = logl((long double)x);
        const double got = vgan::log_tab_eval(x, tab);
        const double err = (double)fabsl((long double)got - ref);
        const double ulp = ref == 0 ? (err == 0 ? 0 : 1e9) : err / (std::nextafter(std::fabs((double)ref), INFINITY) - std::fabs((double)ref));
        if (ulp > worst_ulp) worst_ulp = ulp;
        const double scaled = err / std::fmax(1.0, std::fabs((double)ref));
        if (scaled > worst_abs) worst_abs = scaled;
    };
    for (long i = 0; i < n; ++i) {
        check(std::ldexp(1.0 + u01(), (int)(sm() % 40) - 39)); // the probabilities the kernel sees: (1e-12, 1]
        check(1.0 + (u01() - 0.5) * 0.02);                     // next to 1, where the result is small
        check(u01());
        check(std::ldexp(1.0 + u01(), (int)(sm() % 2040) - 1020)); // the whole normal range
    }
    // bin edges and the extremes of the domain
    for (int i = 0; i < 64; ++i)
        for (int d = -2; d <= 2; ++d) {
            uint64_t bits = ((uint64_t)(VGAN_LOG_OFF + (uint32_t)i * 16384u) << 32);
            bits += (uint64_t)(int64_t)d;
            double x;
            memcpy(&x, &bits, 8);
            check(x);
            check(x * 0x1p-300);
            check(x * 0x1p+300);
        }
    check(2.2250738585072014e-308);
    check(1.7976931348623157e308);
    const double one = vgan::log_tab_eval(1.0, tab);
    printf("worst_ulp %.3f worst_abs %.3e log1 %.17g\n", worst_ulp, worst_abs, one);
    return 0;
}
