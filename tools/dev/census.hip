// Developer aid: how many workgroups of a given shape (threads, LDS bytes, ~VGPRs) does a CU really hold at once, and on which
// SIMDs do their waves land?  Every wave notes its start time, CU and SIMD, then idles for a while; the waves whose start
// lies within a few microseconds of the launch's first start were resident together.
//   hipcc --offload-arch=gfx950 -O2 tools/dev/census.hip -o /tmp/census && /tmp/census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>

struct Rec { unsigned long long t0; unsigned hwid, xcc; };

template <int THREADS, int VGPRS>
__global__ __launch_bounds__(THREADS) void census(Rec *out, unsigned long long spin_ticks) {
    extern __shared__ char lds[];
    // hold VGPRS registers alive
    unsigned acc[VGPRS > 16 ? VGPRS - 16 : 1];
#pragma unroll
    for (int i = 0; i < (VGPRS > 16 ? VGPRS - 16 : 1); ++i) acc[i] = threadIdx.x * (i + 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * (THREADS / 64) + threadIdx.x / 64;
        out[w] = Rec{t0, hwid, xcc};
    }
    lds[threadIdx.x] = (char)threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) {
#pragma unroll
        for (int i = 0; i < (VGPRS > 16 ? VGPRS - 16 : 1); ++i) acc[i] = acc[i] * 1664525u + 1013904223u;
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < (VGPRS > 16 ? VGPRS - 16 : 1); ++i) s ^= acc[i];
    if (s == 0x12345678u) out[0].hwid = s + lds[threadIdx.x];
}

template <int THREADS, int VGPRS> void run(int blocks_per_cu, size_t lds) {
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = ncu * blocks_per_cu, waves = blocks * THREADS / 64;
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * waves);
    hipFuncSetAttribute((const void *)census<THREADS, VGPRS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int api = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, census<THREADS, VGPRS>, THREADS, lds);
    hipLaunchKernelGGL((census<THREADS, VGPRS>), dim3(blocks), dim3(THREADS), lds, 0, d, 20000ull /* 200 us at 100 MHz */);
    hipDeviceSynchronize();
    std::vector<Rec> h(waves);
    hipMemcpy(h.data(), d, sizeof(Rec) * waves, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (auto &r : h) tmin = std::min(tmin, r.t0);
    // per (xcc, se, cu): waves started within 20 us of the first; per SIMD counts
    std::map<unsigned, std::vector<int>> percu;
    int early = 0;
    for (auto &r : h) {
        if (r.t0 - tmin > 2000) continue; // 20 us
        ++early;
        const unsigned cu = (r.hwid >> 8) & 0xF, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 0x7, simd = (r.hwid >> 4) & 0x3;
        const unsigned key = (r.xcc << 16) | (se << 8) | (sh << 4) | cu;
        auto &v = percu[key];
        if (v.empty()) v.assign(4, 0);
        v[simd]++;
    }
    std::map<std::vector<int>, int> shapes;
    for (auto &kv : percu) { auto v = kv.second; shapes[v]++; }
    printf("threads %d vgprs~%d lds %zu: occupancy API %d blocks/CU; %d of %d waves started together on %zu CUs (%.2f waves/CU)\n", THREADS, VGPRS, lds,
           api, early, waves, percu.size(), (double)early / std::max<size_t>(1, percu.size()));
    for (auto &kv : shapes) printf("   SIMD occupancy %d/%d/%d/%d on %d CUs\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
    hipFree(d);
}

int main(int argc, char **argv) {
    if (argc > 1) { // census.bin <lds bytes> ...: 256-thread workgroups of ~84 VGPRs, 6 asked for per CU
        for (int i = 1; i < argc; ++i) run<256, 96>(6, (size_t)atol(argv[i]));
        return 0;
    }
    run<256, 96>(5, 32640);
    run<256, 96>(5, 36000);
    run<256, 120>(4, 36000);
    run<320, 96>(4, 40900);
    run<384, 96>(3, 50528);
    run<640, 96>(2, 79264);
    run<640, 80>(2, 79264);
    run<512, 96>(2, 60000);
    return 0;
}
