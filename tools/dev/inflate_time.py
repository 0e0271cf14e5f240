#!/usr/bin/env python3
"""GPU box: the inflate kernel alone on a synthetic GAM (python3 tools/dev/inflate_time.py [n_reads]); VGAN_LIB picks a variant build."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vgan_amd import _native as N  # noqa: E402
from vgan_amd import haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
g = hc.synth_graph(seed=3)
with tempfile.TemporaryDirectory(prefix="vgan_it_") as d:
    p = os.path.join(d, "x.gam")
    hc.synth_reads(g, n, seed=4, read_len=150).write_gam(p)
    data = open(p, "rb").read()
buf = np.frombuffer(data, np.uint8)
size, ms = C.c_uint64(0), C.c_double(0)
L = N.lib()
N.check(L.vgan_gamdev_inflate_bytes(buf.ctypes.data, len(data), None, 0, C.byref(size), None))
out = np.zeros(size.value + 16, np.uint8)
for _ in range(3):
    rc = L.vgan_gamdev_inflate_bytes(buf.ctypes.data, len(data), out.ctypes.data, len(out), C.byref(size), C.byref(ms))
    print("%d reads, %.1f MB -> %.1f MB: rc %d, kernel %.2f ms" % (n, len(data) / 1e6, size.value / 1e6, rc, ms.value), flush=True)
