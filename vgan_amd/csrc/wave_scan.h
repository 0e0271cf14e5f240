// Wave64 prefix sums without LDS round trips (DPP row shifts and broadcasts): what the flatten kernels' per-read scans over mappings take.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vgan {
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) { // inclusive prefix sum over the 64 lanes
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_last_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
#else
__device__ uint32_t wave_incl_scan_u32(uint32_t v); // (host pass: names only)
__device__ uint32_t wave_last_u32(uint32_t v);
#endif
} // namespace vgan
