#!/usr/bin/env python3
"""Developer aid (GPU box, a -DC8_STATS build): what the tiles of hc_segment_col8_kernel were
(VGAN_LIB=vgan_amd/lib/libvgan_gpu_st.so python3 tools/col8_stats.py [n_reads] [read_len] [seed])."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgan_amd import _native, haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x76676131
g = hc.synth_graph(seed=seed)
a = hc.synth_reads(g, n, seed=seed, read_len=rl)
hb = hc.HostBatch(g, a, packed=True)
ctx = hc.HcContext(g)
db = hc.DeviceBatch(hb)
fn = _native.load().vgan_hc_debug_col8_stats
out = np.zeros(8, np.uint64)
ctx.accumulate(db)
ctx.synchronize()
fn(C.c_void_p(out.ctypes.data), 1)
ctx.reset()
ctx.accumulate(db)
ctx.synchronize()
fn(C.c_void_p(out.ctypes.data), 1)
names = ("tiles", "reads", "general_tiles", "windows_placed", "tiles_with_outside")
print({k: int(v) for k, v in zip(names, out)})
print("reads per tile %.3f  general share %.4f" % (out[1] / max(out[0], 1), out[2] / max(out[0], 1)))
