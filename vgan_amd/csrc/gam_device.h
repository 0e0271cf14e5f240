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
// the same in two kernels (gam_inflate_wave.hip: a wave per member for the Huffman half, tokens through a scratch, a lane per member for
// the LZ77 half); members left with a status other than 0 are to be done again by gamdev_inflate
int gamdev_inflate_wave(const uint8_t *d_in, const GdBlock *d_blocks, uint32_t n_blocks, uint8_t *d_out, uint32_t *d_status, uint32_t *d_tok, uint32_t tok_cap,
                        uint32_t *d_cursor, void *d_reg, uint32_t *d_n_reg, hipStream_t st);
// the members' CRC-32 against the trailers' (d_want[b]): GD_BAD_CRC into the status of a member that is GD_OK and does not match; the
// tables (GAMDEV_CRC_TABS words: gamdev_crc_tables, host) on the device
constexpr uint32_t GAMDEV_CRC_TABS = 2048;
const uint32_t *gamdev_crc_tables();
int gamdev_crc(const uint8_t *d_out, const GdBlock *d_blocks, uint32_t n_blocks, const uint32_t *d_want, const uint32_t *d_tabs, uint32_t *d_status, hipStream_t st);

} // namespace vgan
struct vgan_gamdev;
namespace vgan {
// the arrays of the last vgan_gamdev_parse, as hc_flatten_kernels.hip's DfSlice wants them (device pointers)
struct GamdevSlice {
    const uint32_t *map_off, *qual_off, *edit_off, *e_seq_off, *m_node;
    const int32_t *m_offset, *mapq, *e_len;
    const uint8_t *unmapped, *m_rev, *e_seq, *qual;
    const int64_t *first_node, *first_offset;
    const uint32_t *seq_len; // |Alignment.sequence| per read
    uint64_t n_reads;
    int device;
};
bool gamdev_slice(const vgan_gamdev *g, GamdevSlice *out);
} // namespace vgan
