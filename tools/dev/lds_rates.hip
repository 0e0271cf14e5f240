// developer aid (GPU box): what an LDS instruction costs on gfx950 -- wall time per wave-instruction and CU for reads of 2 / 8 / 16
// bytes per lane, 16-byte writes and fp64 atomic adds (distinct addresses; every pair of lanes on one address; all on one),
// 16 waves per CU.   hipcc --offload-arch=gfx950 -O3 tools/dev/lds_rates.hip -o /tmp/lds_rates && /tmp/lds_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND> __global__ __launch_bounds__(256) void k(double *out, int iters) {
    __shared__ double buf[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 1024; i += 64) buf[wave][i] = i;
    __syncthreads();
    double acc = 0.0;
    unsigned a16 = (unsigned)(size_t)&buf[wave][0] + lane * 16u, a8 = (unsigned)(size_t)&buf[wave][0] + lane * 8u, a2 = (unsigned)(size_t)&buf[wave][0] + lane * 2u;
    unsigned apair = (unsigned)(size_t)&buf[wave][0] + (lane >> 1) * 8u, aone = (unsigned)(size_t)&buf[wave][0];
    double v = 1.0;
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {
            REP64(asm volatile("ds_read_b128 v[10:13], %0\n s_waitcnt lgkmcnt(8)" ::"v"(a16) : "v10", "v11", "v12", "v13", "memory");)
        } else if constexpr (KIND == 1) {
            REP64(asm volatile("ds_read_b64 v[10:11], %0\n s_waitcnt lgkmcnt(8)" ::"v"(a8) : "v10", "v11", "memory");)
        } else if constexpr (KIND == 2) {
            REP64(asm volatile("ds_read_u16 v10, %0\n s_waitcnt lgkmcnt(8)" ::"v"(a2) : "v10", "memory");)
        } else if constexpr (KIND == 3) {
            REP64(asm volatile("ds_write_b128 %0, v[10:13]\n s_waitcnt lgkmcnt(8)" ::"v"(a16) : "memory");)
        } else if constexpr (KIND == 7) {
            REP64(asm volatile("ds_write_b64 %0, v[10:11]\n s_waitcnt lgkmcnt(8)" ::"v"(a8) : "memory");)
        } else if constexpr (KIND == 8) {
            REP64(asm volatile("ds_write_b32 %0, v10\n s_waitcnt lgkmcnt(8)" ::"v"(a8) : "memory");)
        } else if constexpr (KIND == 9) {
            REP64(asm volatile("ds_write2_b64 %0, v[10:11], v[12:13] offset1:1\n s_waitcnt lgkmcnt(8)" ::"v"(a16) : "memory");)
        } else if constexpr (KIND == 10) {
            REP64(asm volatile("ds_read2_b64 v[10:13], %0 offset1:1\n s_waitcnt lgkmcnt(8)" ::"v"(a16) : "v10", "v11", "v12", "v13", "memory");)
        } else if constexpr (KIND == 4) {
            REP64(asm volatile("ds_add_f64 %0, %1\n s_waitcnt lgkmcnt(8)" ::"v"(a8), "v"(v) : "memory");)
        } else if constexpr (KIND == 5) {
            REP64(asm volatile("ds_add_f64 %0, %1\n s_waitcnt lgkmcnt(8)" ::"v"(apair), "v"(v) : "memory");)
        } else if constexpr (KIND == 6) {
            REP64(asm volatile("ds_add_f64 %0, %1\n s_waitcnt lgkmcnt(8)" ::"v"(aone), "v"(v) : "memory");)
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc + buf[wave][lane];
}

template <int KIND> void run(const char *name) {
    double *out;
    hipMalloc(&out, 256 * 4 * 256 * 8);
    const int iters = 500, blocks = 256 * 4; // four workgroups of four waves per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %7.2f ns per wave-instruction and CU (%.3f ms)\n", name, ms * 1e6 / ((double)iters * 64 * 16), ms);
    hipFree(out);
}

int main() {
    run<0>("ds_read_b128, lane-linear");
    run<1>("ds_read_b64, lane-linear");
    run<2>("ds_read_u16, lane-linear");
    run<3>("ds_write_b128, lane-linear");
    run<7>("ds_write_b64, lane-linear");
    run<8>("ds_write_b32, 8-byte stride");
    run<9>("ds_write2_b64 (16 bytes per lane)");
    run<10>("ds_read2_b64 (16 bytes per lane)");
    run<4>("ds_add_f64, an address per lane");
    run<5>("ds_add_f64, two lanes per address");
    run<6>("ds_add_f64, all lanes one address");
    return 0;
}
