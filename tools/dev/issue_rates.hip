// developer aid (GPU box): what a wave's instruction stream costs on gfx950 -- cycles per instruction for chains of independent /
// dependent fp64 FMAs, 32-bit VALU, VALU interleaved with SALU, with 1 and with 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/dev/issue_rates.hip -o /tmp/issue_rates && /tmp/issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND> __global__ void k(double *out, int iters, unsigned long long *cyc) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) { // 64 independent fp64 FMAs (8 chains)
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                              "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if constexpr (KIND == 1) { // 64 dependent fp64 FMAs (one chain)
            REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));)
        } else if constexpr (KIND == 2) { // 64 independent fp32 FMAs
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0000001f), "v"(1e-9f));)
        } else if constexpr (KIND == 3) { // 64 dependent fp32 FMAs
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(1.0000001f), "v"(1e-9f));)
        } else if constexpr (KIND == 4) { // 64 independent fp64 FMAs each followed by an independent SALU add
            REP8(asm volatile("v_fma_f64 %0, %0, %12, %13\n s_add_u32 %8, %8, 1\n v_fma_f64 %1, %1, %12, %13\n s_add_u32 %9, %9, 1\n"
                              "v_fma_f64 %2, %2, %12, %13\n s_add_u32 %10, %10, 1\n v_fma_f64 %3, %3, %12, %13\n s_add_u32 %11, %11, 1\n"
                              "v_fma_f64 %4, %4, %12, %13\n s_add_u32 %8, %8, 1\n v_fma_f64 %5, %5, %12, %13\n s_add_u32 %9, %9, 1\n"
                              "v_fma_f64 %6, %6, %12, %13\n s_add_u32 %10, %10, 1\n v_fma_f64 %7, %7, %12, %13\n s_add_u32 %11, %11, 1"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                              : "v"(m), "v"(c) : "scc");)
        } else if constexpr (KIND == 5) { // 64 SALU adds alone
            REP8(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                              "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1"
                              : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
        } else if constexpr (KIND == 6) { // 64 x (32-bit VALU add)
            REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(s0) : "v"(s1));)
        } else if constexpr (KIND == 7) { // 64 independent conversions u32 -> fp64
            REP8(asm volatile("v_cvt_f64_u32 %0, %8\n v_cvt_f64_u32 %1, %8\n v_cvt_f64_u32 %2, %8\n v_cvt_f64_u32 %3, %8\n"
                              "v_cvt_f64_u32 %4, %8\n v_cvt_f64_u32 %5, %8\n v_cvt_f64_u32 %6, %8\n v_cvt_f64_u32 %7, %8"
                              : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(s1));)
        } else if constexpr (KIND == 8) { // 64 independent fp64 reciprocals (a transcendental)
            REP8(asm volatile("v_rcp_f64 %0, %8\n v_rcp_f64 %1, %8\n v_rcp_f64 %2, %8\n v_rcp_f64 %3, %8\n"
                              "v_rcp_f64 %4, %8\n v_rcp_f64 %5, %8\n v_rcp_f64 %6, %8\n v_rcp_f64 %7, %8"
                              : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(m));)
        } else if constexpr (KIND == 9) { // 64 independent 32-bit multiplies (v_mul_lo_u32)
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                              "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(3u));)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + s0 + s1 + s2 + s3;
}

template <int KIND> void run(const char *name, int n_instr) {
    double *out;
    unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 4 * 1024 * 8);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 256, blocks = 256 * waves_per_simd; // a workgroup = a wave per SIMD of one CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 10, cyc);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-44s %d wave(s)/SIMD: %6.2f shader clocks per instruction per wave (%.3f ms, %.2f ns per instruction per SIMD)\n", name, waves_per_simd,
               (double)h / ((double)iters * n_instr), ms, ms * 1e6 / ((double)iters * n_instr * waves_per_simd));
    }
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("fp64 FMA, 8 independent chains", 64);
    run<1>("fp64 FMA, one dependent chain", 64);
    run<2>("fp32 FMA, 8 independent chains", 64);
    run<3>("fp32 FMA, one dependent chain", 64);
    run<4>("fp64 FMA + SALU add alternating (128)", 128);
    run<5>("SALU add alone", 64);
    run<6>("v_add_u32 dependent chain", 64);
    run<7>("v_cvt_f64_u32, independent", 64);
    run<8>("v_rcp_f64, independent", 64);
    run<9>("v_mul_lo_u32, 4 chains", 64);
    return 0;
}
