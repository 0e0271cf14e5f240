#!/usr/bin/env python3
"""GPU box: the device front end's pipeline in-process, contexts warm, on a file of N reads (default 4 M), five times -- what the pipeline
itself takes without the process's start and end (python3 tools/dev/pipe_time.py [n_reads]; VGAN_LIB picks a variant build)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vgan_amd import haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
g = hc.synth_graph()
CH = 1000000
with tempfile.TemporaryDirectory(prefix="vgan_pt_", dir="/tmp") as d:
    blobs = []
    for c0 in range(0, n, CH):
        hc.synth_reads(g, min(CH, n - c0), first_read=c0).write_gam(d + "/p.gam")
        b = open(d + "/p.gam", "rb").read()
        blobs.append(b[:-28] if c0 + CH < n else b)
data = b"".join(blobs)
del blobs
if os.environ.get("PIPE_PINNED"):  # (the file's bytes in page-locked memory: what the uploads cost when they are plain DMA)
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).pin_memory()
    data = t.numpy()
n_lanes = int(os.environ.get("PIPE_LANES", "1"))  # (several contexts on the one GPU: does the GPU have room for more than a lane gives it?)
ctxs = [hc.HcContext(g) for _ in range(n_lanes)]
ts = []
for i in range(6):
    for c in ctxs:
        c.reset()
    t0 = time.perf_counter()
    st, ps = hc.accumulate_gam_bytes(ctxs, g, data, n_threads=16)
    ts.append(time.perf_counter() - t0)
    if i == 5:
        print("summed over pieces:", {k: round(v) for k, v in ps.items() if k.startswith("ms_")}, "pieces", ps["n_pieces"])
print("%s: %d reads, %.0f MB: pipeline %s ms (first run: %.0f)" % (os.path.basename(os.environ.get("VGAN_LIB", "libvgan_gpu.so")), n, (data.nbytes if hasattr(data, "nbytes") else len(data)) / 1e6,
                                                                   " ".join("%.0f" % (t * 1e3) for t in ts[1:]), ts[0] * 1e3))
