#!/usr/bin/env python3
"""Developer aid (GPU box): the GAM front end on the device against the host pipeline on one synthetic file
(python3 tools/gamdev_rates.py [n_reads] [read_len])."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgan_amd import haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
g = hc.synth_graph(seed=0x76676131)
ctx = hc.HcContext(g)
df = hc.DeviceFlatten(ctx, g)
with tempfile.TemporaryDirectory(prefix="vgan_gd_") as d:
    p = os.path.join(d, "x.gam")
    hc.synth_reads(g, n, seed=0x76676131, read_len=rl).write_gam(p)
    data = open(p, "rb").read()
    t0 = time.perf_counter()
    parts = hc.AlnParts.read_gam(p)
    t_host_parse = time.perf_counter() - t0
    df.run(parts)
    t0 = time.perf_counter()
    res = df.run(parts)
    t_host_df = time.perf_counter() - t0
gd = hc.GamDevice()
gd.parse(data)
for _ in range(2):
    t0 = time.perf_counter()
    gd.parse(data)
    t_dev_parse = time.perf_counter() - t0
    t0 = time.perf_counter()
    res2 = df.run_gamdev(gd)
    t_dev_df = time.perf_counter() - t0
print("file %.1f MB -> %.1f MB, %d reads" % (len(data) / 1e6, gd.sizes["inflated_bytes"] / 1e6, gd.sizes["reads"]))
print("host: parse %.3f s (%.2f M reads/s), device flatten of its arrays %.3f s" % (t_host_parse, n / t_host_parse / 1e6, t_host_df))
print("device: parse %.3f s (%.2f M reads/s) %s, device flatten in place %.3f s; both %.2f M reads/s" % (
    t_dev_parse, n / t_dev_parse / 1e6, {k: round(v, 1) for k, v in gd.ms.items()}, t_dev_df, n / (t_dev_parse + t_dev_df) / 1e6))
