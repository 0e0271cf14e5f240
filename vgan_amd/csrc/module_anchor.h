// One kernel of every translation unit of device code: the HIP runtime loads a translation unit's code object when one of its kernels is
// first launched -- or asked about (vgan_device_preload, hc_create.hip: the loads of a run's kernels made at its start, beside each other,
// instead of one by one on the first piece's way through the stages).
#pragma once
namespace vgan {
const void *anchor_gam_inflate_wave();
const void *anchor_gam_kernels();
const void *anchor_hc_flatten();
const void *anchor_hc_col8();
const void *anchor_hc_kernels();
const void *anchor_hc_wave();
const void *anchor_euka_kernels();
const void *anchor_euka_flatten();
const void *anchor_sb_kernels();
const void *anchor_sb_flatten();
} // namespace vgan
