#!/usr/bin/env python3
"""Host front-end throughput (GAM write / read+parse / flatten) on this machine's cores; not part of bench.py's value."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgan_amd import haplocart as hc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
g = hc.synth_graph()
t = time.time(); a = hc.synth_reads(g, n); print("synth %.2fs" % (time.time() - t))
t = time.time(); a.write_gam("/tmp/x.gam"); print("write_gam %.2fs %.1f MB" % (time.time() - t, os.path.getsize("/tmp/x.gam") / 1e6))
t = time.time(); b = hc.AlnSet.read_gam("/tmp/x.gam"); dt = time.time() - t; print("read_gam %.2fs %.0f reads/s" % (dt, n / dt))
for th in (1, 8, 0):
    t = time.time(); hb = hc.HostBatch(g, b, n_threads=th); dt = time.time() - t
    print("flatten threads=%s %.2fs %.0f reads/s" % (th or "all", dt, n / dt))
print("cores", os.cpu_count())
