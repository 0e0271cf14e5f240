#!/usr/bin/env python3
"""Developer aid (GPU box): event counts of the wave kernel's column loop.  Needs the counting build:
VGAN_BUILD_TAG=_stats VGAN_EXTRA_FLAGS=-DWV_STATS python -m vgan_amd.build, then
VGAN_LIB=vgan_amd/lib/libvgan_gpu_stats.so python3 tools/wave_stats.py [n_reads] [read_len]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgan_amd import _native, haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
fn = _native.load().vgan_hc_debug_wave_stats
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
g = hc.synth_graph(seed=1)
a = hc.synth_reads(g, n, seed=2, read_len=rl)
hb = hc.HostBatch(g, a, packed=True)
ctx = hc.HcContext(g)
db = hc.DeviceBatch(hb, ctx=ctx)
out = (ctypes.c_ulonglong * 8)()
fn(out, 1)
ctx.accumulate(db)
ctx.synchronize()
fn(out, 0)
names = ["tiles", "reads", "chunk groups", "far groups", "rare groups", "segment passes", "-", "windows placed"]
for k, v in zip(names, out):
    print("%-16s %d" % (k, v))
