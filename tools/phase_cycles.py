#!/usr/bin/env python3
"""Developer aid: per-phase cycle split of hc_segment_tile_kernel.

Needs a library built with the phase marks: VGAN_BUILD_TAG=_pt VGAN_EXTRA_FLAGS=-DVGAN_PHASE_TIMING python vgan_amd/build.py
(lands beside the product's library), then VGAN_LIB=vgan_amd/lib/libvgan_gpu_pt.so python tools/phase_cycles.py.  Prints the share of each wave's cycles (lane 0) spent in each phase / barrier wait.
"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from vgan_amd import _native, haplocart as hc  # noqa: E402

NAMES = ["top: windows -> LDS", "barrier 1", "B quality prefix", "barrier 2", "C segments", "barrier 3",
         "next tile: extents + requests", "D columns", "next tile: node gather", "barrier 4", "E store", "-"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    lib = _native.load()  # VGAN_LIB=vgan_amd/lib/libvgan_gpu_pt.so selects the VGAN_BUILD_TAG=_pt build for the whole process
    fn = lib.vgan_hc_debug_phase_cycles
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    g = hc.synth_graph(seed=1)
    a = hc.synth_reads(g, n, seed=2, read_len=150)
    hb = hc.HostBatch(g, a)
    db = hc.DeviceBatch(hb)
    ctx = hc.HcContext(g)
    out = (ctypes.c_ulonglong * 48)()
    for rep in range(3):
        ctx.reset()
        fn(out, 1)
        ctx.accumulate(db)
        ctx.synchronize()
        fn(out, 0)
    v = np.array(list(out), dtype=np.float64).reshape(4, 12)
    print("%-30s %s" % ("share of each wave's cycles", "   ".join("wave %d" % w for w in range(4))))
    for i, name in enumerate(NAMES):
        print("%-30s %s" % (name, "   ".join("%5.2f%%" % (100 * v[w, i] / v[w].sum()) for w in range(4))))
    print("cycles per wave: " + "  ".join("%.3e" % v[w].sum() for w in range(4)))


if __name__ == "__main__":
    main()
