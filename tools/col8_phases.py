#!/usr/bin/env python3
"""Developer aid (GPU box, a -DC8_PHASES build): shader-clock cycles of a wave per phase of hc_segment_col8_kernel's tile loop
(VGAN_LIB=vgan_amd/lib/libvgan_gpu_ph.so python3 tools/col8_phases.py [n_reads] [read_len])."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgan_amd import _native, haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
g = hc.synth_graph(seed=0x76676131)
a = hc.synth_reads(g, n, seed=0x76676131, read_len=rl)
hb = hc.HostBatch(g, a, packed=True)
ctx = hc.HcContext(g)
db = hc.DeviceBatch(hb)
fn = _native.load().vgan_hc_debug_col8_phases
out = np.zeros(12, np.uint64)
ctx.accumulate(db)
ctx.synchronize()
fn(C.c_void_p(out.ctypes.data), 1)
ctx.reset()
ctx.accumulate(db)
ctx.synchronize()
fn(C.c_void_p(out.ctypes.data), 1)
names = ("top", "reads+Q", "C", "C2(general)", "D fast", "D not fast", "end (next classes)")
idx = (0, 1, 2, 3, 4, 5, 8)
tiles, gen = int(out[6]), int(out[7])
tot = float(sum(out[i] for i in idx))
print("tiles %d not fast %d (of them through the context's table %d); cycles per tile %.0f" % (tiles, gen, int(out[9]), tot / max(tiles, 1)))
for k, v in zip(names, [out[i] for i in idx]):
    print("%-12s %6.1f %%  %8.0f cycles per tile" % (k, 100.0 * float(v) / tot, float(v) / max(tiles, 1)))
if gen:
    print("D general per general tile %.0f, C2 per general tile %.0f, D fast per fast tile %.0f" % (float(out[5]) / gen, float(out[3]) / gen, float(out[4]) / max(tiles - gen, 1)))
