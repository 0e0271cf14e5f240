#!/usr/bin/env python3
"""Developer aid (GPU box): where a wave of the segment kernel spends its time, phase by phase (shader clock).  Needs the
instrumented build: VGAN_BUILD_TAG=_ph VGAN_EXTRA_FLAGS=-DWV_PHASES python -m vgan_amd.build, then
VGAN_LIB=vgan_amd/lib/libvgan_gpu_ph.so python3 tools/wave_phases.py [n_reads] [read_len]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgan_amd import _native, haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
fn = _native.load().vgan_hc_debug_wave_phases
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
g = hc.synth_graph(seed=1)
a = hc.synth_reads(g, n, seed=2, read_len=rl)
hb = hc.HostBatch(g, a, packed=True)
ctx = hc.HcContext(g)
db = hc.DeviceBatch(hb, ctx=ctx)
ctx.accumulate(db)
ctx.synchronize()
out = (ctypes.c_ulonglong * 8)()
fn(out, 1)
ctx.accumulate(db)
ctx.synchronize()
fn(out, 0)
names = ["node gather issued (after the wait for this tile's records)", "next tile formed, header after it requested", "read records", "Q", "C", "D", "E"]
tot = sum(out[:7])
for k, v in zip(names, out[:7]):
    print("%-66s %6.2f %%   %8.0f clocks per tile" % (k, 100.0 * v / tot, v / max(out[7], 1)))
print("tiles %d, clocks per tile %.0f (shader clock, 100 MHz reference on gfx9: s_memtime)" % (out[7], tot / max(out[7], 1)))

