#!/usr/bin/env python3
"""Developer aid (GPU box): segment-kernel time of the resident packed batch and its final vector's checksum, for A/B builds
(VGAN_LIB=... python3 tools/wave_time.py [n_reads] [read_len] [reps])."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgan_amd import haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
g = hc.synth_graph(seed=1)
a = hc.synth_reads(g, n, seed=2, read_len=rl)
hb = hc.HostBatch(g, a, packed=True)
ctx = hc.HcContext(g)
db = hc.DeviceBatch(hb)
for _ in range(3):
    ctx.accumulate(db)
ctx.reset()
ctx.profile_enable(True)
for _ in range(reps):
    ctx.accumulate(db)
pr = ctx.profile_read()
ctx.reset()
ctx.accumulate(db)
f = ctx.finalize()
print("%s segment %.4f ms  sum(final) %.15e  min %.15e" % (os.environ.get("VGAN_LIB", "default"), pr["segment"][0] / pr["segment"][1], float(np.sum(f)), float(np.min(f))))
