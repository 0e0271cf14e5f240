#!/usr/bin/env python3
"""GPU box: what a bench step costs beyond its segment kernel -- with and without the HIP events around every kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vgan_amd import haplocart as hc  # noqa: E402

g = hc.synth_graph(seed=0x76676131)
ctx = hc.HcContext(g)
ctx.use_torch_stream() if hasattr(ctx, "use_torch_stream") else None
a = hc.synth_reads(g, 1000000, seed=0x76676131, read_len=150)
db = hc.DeviceBatch(hc.HostBatch(g, a, packed=True))
final_dev = torch.zeros(g.n_paths, dtype=torch.float64, device="cuda:0")


def run(n, what):
    for _ in range(5):
        what()
    torch.cuda.synchronize()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        what()
    ctx.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def step():
    ctx.reset()
    ctx.accumulate(db)
    ctx.finalize_device(final_dev)


def seg_only():
    ctx.accumulate(db)


for prof in (False, True, False, True):
    ctx.profile_enable(prof)
    print("events %-5s step %.4f ms, accumulate alone %.4f ms" % (prof, run(50, step), run(50, seg_only)), flush=True)
