#!/usr/bin/env python3
"""Developer aid (GPU box): the wave-owned segment kernel against the LDS-tiled one on the same batch.

  python tools/wave_ab.py [n_reads] [read_len]

Prints the largest difference of D_m per segment and of the final vector between the two data paths, the layout pass's
time, and each kernel's time per accumulate of the resident batch (HIP events around the launches).
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vgan_amd import haplocart as hc  # noqa: E402


def kernel_ms(ctx, batch, reps):
    ctx.profile_enable(True)
    for _ in range(reps):
        ctx.accumulate(batch)
    pr = ctx.profile_read()
    ctx.profile_enable(False)
    return {k: m / max(1, c) for k, (m, c) in pr.items()}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    g = hc.synth_graph(seed=1)
    a = hc.synth_reads(g, n, seed=2, read_len=rl)
    hb = hc.HostBatch(g, a)
    ctx = hc.HcContext(g)
    out = {"n_reads": hb.n_reads, "n_segments": hb.n_segments, "n_tileable": hb.n_tileable, "read_len": rl}
    # parity of the two data paths, per segment and through the accumulators
    sub = hc.HostBatch(g, a, 0, min(n, 50000))
    os.environ["VGAN_HC_KERNEL"] = "tile"
    d_tile = ctx.segment_weights(sub)
    os.environ["VGAN_HC_KERNEL"] = "wave"
    d_wave = ctx.segment_weights(sub)
    out["D_max_abs_diff"] = float(np.max(np.abs(d_tile - d_wave)))
    out["D_max_rel_diff"] = float(np.max(np.abs(d_tile - d_wave) / np.maximum(np.abs(d_tile), 1e-300)))
    fin = {}
    for which in ("tile", "wave"):
        os.environ["VGAN_HC_KERNEL"] = which
        ctx.reset()
        ctx.accumulate(hb)
        fin[which] = ctx.finalize()
    out["final_max_rel_diff"] = float(np.max(np.abs(fin["tile"] - fin["wave"]) / np.abs(fin["tile"])))
    print(json.dumps(out), flush=True)
    # timing on the resident batch
    db = hc.DeviceBatch(hb, ctx=ctx)
    out["pack_ms"] = db.pack_ms
    reps = 20
    for which in ("tile", "wave"):
        os.environ["VGAN_HC_KERNEL"] = which
        ctx.reset()
        kernel_ms(ctx, db, 3)
        ms = kernel_ms(ctx, db, reps)
        out["%s_segment_ms" % which] = ms["segment"]
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.accumulate(db)
        ctx.synchronize()
        out["%s_wall_ms" % which] = (time.perf_counter() - t0) * 1e3 / reps
    ctx.reset()
    ctx.accumulate(db)
    f2 = ctx.finalize()
    out["resident_final_max_rel_diff"] = float(np.max(np.abs(fin["tile"] - f2) / np.abs(fin["tile"])))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
