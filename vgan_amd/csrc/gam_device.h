// Device-side views of the GAM front end on the device (gam_kernels.hip) shared with the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vgan {

struct GdBlock { // one BGZF member: its DEFLATE payload within the file's bytes, its output within the inflated bytes
    uint64_t in_off, out_off;
    uint32_t in_size, out_size;
};

// inflates n_blocks BGZF members (device pointers); d_status[b] = 0 or a GD_* code
int gamdev_inflate(const uint8_t *d_in, const GdBlock *d_blocks, uint32_t n_blocks, uint8_t *d_out, uint32_t *d_status, hipStream_t st);

} // namespace vgan
