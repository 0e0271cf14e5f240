// Developer probe: does s_atomic_add (scalar memory atomic, returning) work on this GPU?  Every wave takes a ticket.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned *ctr, unsigned *out) {
    unsigned v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = v;
}
int main() {
    unsigned *ctr, *out;
    const int blocks = 4096, waves = blocks * 4;
    hipMalloc(&ctr, 4);
    hipMalloc(&out, waves * 4);
    hipMemset(ctr, 0, 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ctr, out);
    std::vector<unsigned> h(waves);
    hipMemcpy(h.data(), out, waves * 4, hipMemcpyDeviceToHost);
    unsigned total;
    hipMemcpy(&total, ctr, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = total == (unsigned)waves;
    for (int i = 0; i < waves; ++i) ok = ok && h[i] == (unsigned)i;
    printf("s_atomic_add: counter %u of %d, tickets %s\n", total, waves, ok ? "unique 0..n-1" : "NOT unique");
    return ok ? 0 : 1;
}
