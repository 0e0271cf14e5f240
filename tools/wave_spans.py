#!/usr/bin/env python3
"""Developer aid (GPU box): the segment kernel's timeline wave by wave -- when each wave started and left, how many tiles it
took, when it took its last work unit -- i.e. how long the launch's tail is.  Needs the instrumented build:
tools/build_variant.sh sp -DWV_SPANS, then VGAN_LIB=vgan_amd/lib/libvgan_gpu_sp.so python3 tools/wave_spans.py [n_reads] [read_len]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgan_amd import _native, haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
g = hc.synth_graph(seed=1)
a = hc.synth_reads(g, n, seed=2, read_len=rl)
hb = hc.HostBatch(g, a, packed=True)
ctx = hc.HcContext(g)
db = hc.DeviceBatch(hb, ctx=ctx)
for _ in range(3):
    ctx.accumulate(db)
    ctx.synchronize()
# when the waves left: the launch's tail (100 MHz clock)
import numpy as np
sp = _native.load().vgan_hc_debug_wave_spans
buf = np.zeros(4 * 16384, dtype=np.uint64)
sp.argtypes = [ctypes.c_void_p]
if sp(buf.ctypes.data) == 0:
    t0, t1, nt, tl = (buf[k::4].astype(np.int64) for k in range(4))
    on = t1 > 0
    gw = np.nonzero(on)[0]
    t0, t1, nt, tl = t0[on], t1[on], nt[on], tl[on]
    beg, end = t0.min(), t1.max()
    idle = (end - t1) / 100.0
    late = (t0 - beg) / 100.0
    print("waves %d, first start to last end %.1f us; start skew mean %.1f max %.1f us" % (on.sum(), (end - beg) / 100.0, late.mean(), late.max()))
    print("idle behind a wave's end: mean %.1f us, median %.1f, p90 %.1f, p99 %.1f, max %.1f" % (idle.mean(), np.median(idle), np.percentile(idle, 90), np.percentile(idle, 99), idle.max()))
    home = (gw // 4) % 8  # the XCD, as workgroups are dealt round robin
    for h in range(8):
        m = home == h
        print("XCD %d: waves %d, mean end %.0f us, last end %.0f, tiles per wave %.1f, us per tile %.2f, last unit taken at %.0f (max %.0f)" % (
            h, m.sum(), (t1[m] - beg).mean() / 100.0, (t1[m] - beg).max() / 100.0, nt[m].mean(), ((t1[m] - t0[m]) / 100.0 / np.maximum(nt[m], 1)).mean(),
            (tl[m] - beg).mean() / 100.0, (tl[m] - beg).max() / 100.0))
