#!/usr/bin/env python3
"""Developer aid (GPU box): segment-kernel time of a resident batch by read length -- the packed route as the library routes it,
the packed route forced onto the wave kernel, and the LDS-tiled kernel on the SoA form of the same reads (round 2's kernel: the
bar the packed route has to stay under at every length).   python3 tools/len_sweep.py [reads_at_150bp]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgan_amd import haplocart as hc  # noqa: E402

base = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g = hc.synth_graph(seed=1)
ctx = hc.HcContext(g)


def seg_ms(batch, reps=10):
    for _ in range(2):
        ctx.accumulate(batch)
    ctx.reset()
    ctx.profile_enable(True)
    for _ in range(reps):
        ctx.accumulate(batch)
    pr = ctx.profile_read()
    ctx.profile_enable(False)
    ctx.reset()
    return pr["segment"][0] / max(pr["segment"][1], 1)


rows = []
for rl in (40, 75, 150, 300, 600):
    n = base if rl <= 150 else base * 150 // rl
    a = hc.synth_reads(g, n, seed=2, read_len=rl)
    os.environ.pop("VGAN_HC_KERNEL", None)
    dpk = hc.DeviceBatch(hc.HostBatch(g, a, packed=True))
    row = {"read_len": rl, "reads": n, "packed_default_ms": seg_ms(dpk)}
    os.environ["VGAN_HC_KERNEL"] = "wave"
    row["packed_wave_ms"] = seg_ms(dpk)
    del dpk
    os.environ["VGAN_HC_KERNEL"] = "tile"
    dsoa = hc.DeviceBatch(hc.HostBatch(g, a))
    row["soa_tile_ms"] = seg_ms(dsoa)
    del dsoa
    os.environ.pop("VGAN_HC_KERNEL", None)
    rows.append(row)
    print(json.dumps(row), flush=True)
