// The HaploCart context and its device buffers (private to the library: the C-ABI sees an opaque vgan_hc_ctx): shared by
// hc_capi.hip (batches in, vectors out), hc_create.hip (what a context holds of the graph) and hc_reduce.hip (several GPUs).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <deque>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "hc_device.h"
#include "host/common.h"
#include "vgan_gpu.h"

#ifndef HIPCHK
#define HIPCHK(expr)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return vgan::fail(VGAN_ENODEV, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)
#endif

namespace vgan {
namespace hcx {

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return VGAN_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 8 + 64;
        HIPCHK(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        return VGAN_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

} // namespace hcx
} // namespace vgan
using vgan::hcx::DevBuf;
using vgan::HcGraphDev;
using vgan::HcNodeDev;
using vgan::HcPackedDev;
using vgan::HcParamsDev;

struct vgan_hc_packed {
    int device = 0;
    DevBuf<uint4> rhdr;
    DevBuf<uint32_t> srec;
    DevBuf<uint32_t> crec;
    DevBuf<uint8_t> qualp;
    DevBuf<uint32_t> maxima;
    HcPackedDev d{};
    void release() {
        rhdr.release();
        srec.release();
        crec.release();
        qualp.release();
        maxima.release();
    }
};

struct vgan_hc_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int mode = VGAN_HC_MODE_NODE_WEIGHTS;
    uint32_t P = 0, W = 0, rows = 0, n_tiles = 0;
    HcGraphDev g{};
    HcParamsDev prm{};
    DevBuf<uint64_t> umask;
    DevBuf<uint16_t> umaskT, tile_word0;
    DevBuf<HcNodeDev> node_tab, cls_tab;
    DevBuf<uint16_t> node_hi;
    DevBuf<double> tables; // lq[256] qscore[100] incmap[100]
    DevBuf<double> col_memo, col_memo2; // hc_col8_kernels.hip: the tables of column terms
    DevBuf<double> accum;                                    // one block: nodeW | acc_seg | acc_node | totals (one memset)
    struct View { double *p = nullptr; } nodeW, acc_seg, acc_node, totals;
    size_t accum_n = 0;
    DevBuf<double> final_vec;
    DevBuf<double> segD, segS, segU, dump;
    // staging for host batches
    DevBuf<uint32_t> s_u32;
    DevBuf<uint16_t> s_u16;
    DevBuf<uint8_t> s_u8;
    vgan_hc_packed scratch_pack; // layout pass output of batches that come without a packed companion
    DevBuf<uint32_t> work_ctr;   // the segment kernel's work queue (hc_wave_kernels.hip)
    uint32_t work_base = 0;
    bool work_dirty = false;     // a launch failed or the stream changed: counter and mirror start over
    bool touched = false;        // something was accumulated since the last reset (vgan_hc_reduce leaves the others out)
    // posterior
    std::vector<std::string> path_names;
    std::unordered_map<std::string, uint32_t> path_index;
    std::unordered_map<std::string, std::vector<std::string>> parents, children;
    DevBuf<uint32_t> lists; // posterior: list offsets, then the path indices
    DevBuf<double> conf;
    // posterior: name -> path indices (built at the first call), and the lists of the last predicted haplotype as they sit on
    // the device (a caller asks about the same prediction again and again: the walk and its upload are done once)
    std::unordered_map<std::string, std::vector<uint32_t>> by_name;
    std::string post_predicted, post_clades;
    uint32_t post_n_off = 0, post_ns = 0;
    // profiling: pairs of events per timed launch, resolved in vgan_hc_profile_read
    int profiling = 0; // 0 off, 1 HIP events around every kernel, 2 around the segment kernel only
    struct Timed {
        int slot;
        hipEvent_t a, b;
    };
    std::vector<Timed> timed;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[VGAN_HC_K_COUNT] = {0, 0, 0, 0, 0};
    uint64_t prof_n[VGAN_HC_K_COUNT] = {0, 0, 0, 0, 0};
};
