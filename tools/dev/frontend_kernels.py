#!/usr/bin/env python3
"""GPU box: the device front end's kernels one at a time (nothing beside them on the GPU) on one piece of N reads: vgan_gamdev_parse +
vgan_hc_devflat_run_gamdev, three times.  Under rocprofv3 --kernel-trace --stats the per-kernel averages are what a piece costs standalone:
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fk -- python3 $GRAFT_REPO_ROOT/tools/dev/frontend_kernels.py 500000"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vgan_amd import haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
g = hc.synth_graph()
with tempfile.TemporaryDirectory(prefix="vgan_fk_") as d:
    p = os.path.join(d, "x.gam")
    hc.synth_reads(g, n).write_gam(p)
    data = open(p, "rb").read()
ctx = hc.HcContext(g)
df = hc.DeviceFlatten(ctx, g)
gd = hc.GamDevice()
for i in range(3):
    t0 = time.perf_counter()
    gd.parse(data)
    t1 = time.perf_counter()
    r = df.run_gamdev(gd)
    t2 = time.perf_counter()
    print("%d reads: parse %.1f ms, flatten %.1f ms (%d taken, %d left to the host)" % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, r.pk.n_reads, int(r.host_mask.sum())), flush=True)
