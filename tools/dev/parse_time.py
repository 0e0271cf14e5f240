#!/usr/bin/env python3
"""GPU box: vgan_gamdev_parse alone on a 10 M-read synthetic file, several times (VGAN_LIB picks the build): python3 tools/dev/parse_time.py [n]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vgan_amd import haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
path = "/tmp/vgan_parse_time_%d.gam" % n
if not os.path.exists(path):
    g = hc.synth_graph()
    CH = 1000000
    with open(path, "wb") as f:
        for c0 in range(0, n, CH):
            a = hc.synth_reads(g, min(CH, n - c0), first_read=c0)
            a.write_gam(path + ".part")
            blob = open(path + ".part", "rb").read()
            f.write(blob[:-28] if c0 + CH < n else blob)
            del a
data = open(path, "rb").read()
gd = hc.GamDevice()
for i in range(4):
    t0 = time.perf_counter()
    gd.parse(data)
    dt = time.perf_counter() - t0
    print("parse %.0f ms  %s" % (dt * 1e3, {k: round(v) for k, v in gd.ms.items()}), flush=True)
gd.close()
