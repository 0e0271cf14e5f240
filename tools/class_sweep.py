#!/usr/bin/env python3
"""GPU box: segment-kernel time of a resident 1 M x 150 bp batch against the number of distinct mappability values of the graph (the
node classes of hc_segment_col8_kernel are the distinct {mappability, match probability} pairs: the workgroup's table covers the 16 most
frequent, the context's wide table all of them up to 256 -- beyond that the wave kernel takes the batch) and against the share of reads
whose quality bytes are 60 (beyond the workgroup's table, inside the wide one).   python3 tools/class_sweep.py [reads] > profiles/roundN_class_sweep.jsonl"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgan_amd import haplocart as hc  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g0 = hc.synth_graph(seed=1)
a = hc.synth_reads(g0, n_reads, seed=2, read_len=150)


def seg_ms(ctx, batch, reps=10):
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


def graph_with(n_values, rng):
    """n_values distinct mappabilities: 1.0 and n_values - 1 others on windows of 40 coordinates (half of the coordinates in all)."""
    mp = np.ones(len(g0.mappability))
    if n_values > 1:
        vals = 0.3 + 0.7 * (np.arange(1, n_values) / n_values)
        k = 0
        for w in range(0, len(mp), 40):
            if (w // 40) % 2 == 0:
                mp[w:w + 40] = vals[k % len(vals)]
                k += 1
    return hc.Graph.from_arrays(g0.min_id, g0.max_id, np.array(g0.node_seq_off), np.array(g0.node_seq).tobytes(), g0.n_paths, np.array(g0.mask),
                                np.array(g0.pangenome_base), mp, "\n".join(g0.path_names) + "\n", g0.parents_txt, g0.children_txt)


def with_q60(hb, share):
    pa = hb.packed_arrays()
    if share <= 0:
        return
    h = pa["rhdr"].reshape(-1, 4)
    step = max(1, int(round(1 / share)))
    for r in range(0, hb.pk.n_reads, step):
        q0, q1, c0 = int(h[r, 1]), int(h[r + 1, 1]), int(h[r, 2])
        pa["qualp"][q0:q1] = 60
        pa["crec"][c0:c0 + (q1 - q0)] = (pa["crec"][c0:c0 + (q1 - q0)] & np.uint32(0xFF00FFFF)) | np.uint32(60 << 16)


rng = np.random.default_rng(5)
base = None
for n_values in (1, 16, 17, 32, 33, 64, 256, 300):
    g = graph_with(n_values, rng)
    ctx = hc.HcContext(g)
    for share in ((0.0, 0.1, 1.0) if n_values in (1, 17, 256) else (0.0,)):
        hb = hc.HostBatch(g, a, packed=True)
        with_q60(hb, share)
        ms = seg_ms(ctx, hc.DeviceBatch(hb))
        row = {"mappability_values": n_values, "q60_share": share, "reads": hb.pk.n_reads, "segment_ms": round(ms, 4)}
        if n_values == 17 and share == 0.0:
            base = ms
        print(json.dumps(row), flush=True)
    ctx.close()
print(json.dumps({"seventeen_values_ms": base}), flush=True)
