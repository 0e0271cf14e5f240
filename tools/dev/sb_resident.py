#!/usr/bin/env python3
"""GPU box: the soibean refresh launched per iteration against the resident kernel (tools/dev: python3 tools/dev/sb_resident.py [reads ...])."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vgan_amd import euka as ek  # noqa: E402
from vgan_amd import haplocart as hc  # noqa: E402
from vgan_amd import soibean as sb  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [20000, 200000, 1000000, 2000000]
g = hc.synth_graph(seed=1, genome_len=16569, n_nodes=11000, n_paths=28)
idx = {n: i for i, n in enumerate(g.path_names)}
pairs = [(idx[t[0]], idx[t[1]]) for t in (ln.split() for ln in g.parents_txt.splitlines()) if len(t) >= 2]
freqs = [0.31, 0.27, 0.13, 0.29, 0.44, 0.56, 0.0012]
dm = ek.Damage.from_text("", "")
for n in sizes:
    alns = hc.synth_reads(g, n, seed=1, read_len=65, indel_rate=0.005, softclip_rate=0.01)
    hb = sb.SbHostBatch(g, alns)
    ctx = sb.SbContext(g, dm)
    ctx.precompute(hb)
    rng = np.random.default_rng(3)
    states = []
    for _ in range(200):
        th = rng.dirichlet([1, 1, 1])
        st = []
        for y in range(3):
            c, p = pairs[rng.integers(len(pairs))]
            st.append((c, p, 0.01 + 0.05 * rng.random(), rng.random() * 0.98 + 0.01, float(th[y])))
        states.append(st)
    res = {}
    for mode in (0, 1, 0, 1):
        ctx.resident(mode)
        for st in states[:20]:
            ctx.refresh(st, 0.01, freqs)
        ctx.kernel_ms()
        t0 = time.perf_counter()
        out = [ctx.refresh(st, 0.01, freqs)[0] for st in states]
        dt = (time.perf_counter() - t0) / len(states)
        km = ctx.kernel_ms()["refresh"]
        res.setdefault(mode, []).append(out)
        print("%8d reads  %s  %.1f us per refresh  (device clock %.1f us over %d)  launches %d" % (
            hb.n_reads, "resident" if mode else "launched", dt * 1e6, km[0] / max(km[1], 1) * 1e3, km[1], ctx.resident_launches()), flush=True)
    import hashlib
    print("   launched results:", hashlib.sha1(np.array(res[0][0]).tobytes()).hexdigest()[:16], flush=True)
    same = all(a == b for a, b in zip(res[0][0], res[1][0])) and res[1][0] == res[1][1]
    print("   bit-identical:", same, flush=True)
    time.sleep(0.05)  # (the kernel leaves by itself after 5 ms)
    ctx.resident(1)
    x = ctx.refresh(states[0], 0.01, freqs)[0]
    print("   after an idle gap: same value", x == res[0][0][0], " launches", ctx.resident_launches(), flush=True)
    ctx.close()
