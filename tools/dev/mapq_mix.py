#!/usr/bin/env python3
"""GPU box: the segment kernel's time on batches of one million reads with 0 %, 10 % and 100 % reads of mapping quality below 60."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vgan_amd import haplocart as hc  # noqa: E402

g = hc.synth_graph(seed=0x76676131)
ctx = hc.HcContext(g)
for rate in (0.0, 0.1, 0.3, 1.0):
    a = hc.synth_reads(g, 1000000, seed=0x76676131, read_len=150, low_mapq_rate=rate)
    hb = hc.HostBatch(g, a, packed=True)
    db = hc.DeviceBatch(hb)
    for _ in range(3):
        ctx.reset()
        ctx.accumulate(db)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.accumulate(db)
    ctx.synchronize()
    print("low-mapq share %.1f: %.3f ms per pass of %d reads" % (rate, (time.perf_counter() - t0) / 10 * 1e3, hb.pk.n_reads), flush=True)
    del db, hb, a
