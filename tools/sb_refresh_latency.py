#!/usr/bin/env python3
"""Latency of one soibean likelihood refresh as the chain driver issues it (vgan_sb_estimate's engine), at several read counts:
wall time per call against the refresh kernel's own time.  usage: sb_refresh_latency.py [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_sb_chain_cpu import _newick_of  # noqa: E402
from vgan_amd import euka as ek  # noqa: E402
from vgan_amd import haplocart as hc  # noqa: E402
from vgan_amd import soibean as sb  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
FREQS = [.31, .25, .15, .29, .46, .54, 0.6]
g = hc.synth_graph(seed=17, genome_len=16000, n_nodes=11000, n_paths=28)
dm = ek.Damage.from_text("", "")
tree = sb.Tree.parse(_newick_of(g))
node_path = tree.node_paths(g.path_names)
for n in (20000, 1000000):
    a = hc.synth_reads(g, n, seed=6, read_len=60)
    ctx = sb.SbContext(g, dm)
    ctx.precompute(sb.SbHostBatch(g, a))
    for k in (1, 3):
        st = [[(3 + y, 1, 0.02, 0.4, 1.0 / k) for y in range(k)]]
        ctx.loglike(st, 0.01, FREQS)
        ctx.kernel_ms()
        t = time.perf_counter()
        for _ in range(calls):
            ctx.loglike(st, 0.01, FREQS)
        dt = (time.perf_counter() - t) / calls
        km = ctx.kernel_ms()["refresh"]
        print("reads %8d k=%d: vgan_sb_loglike %.1f us per call (ctypes included), refresh kernels %.1f us" % (n, k, dt * 1e6, km[0] / max(km[1], 1) * 1e3))
        # the engine's refresh (what vgan_sb_estimate calls): fused kernel + fold into pinned host memory
        import ctypes as C
        from vgan_amd import _native as N
        e = N.SbEngine()
        N.check(N.lib().vgan_sb_engine_gpu(ctx._h, C.byref(e)))
        arr = (N.SbSource * k)(*[N.SbSource(*s_) for s_ in st[0]])
        f7 = (C.c_double * 7)(*FREQS)
        out, gd = C.c_double(0), C.c_uint64(0)
        assert e.refresh(e.user, k, C.cast(arr, C.c_void_p), 0.01, f7, C.byref(out), C.byref(gd)) == 0
        ref, _ = ctx.loglike(st, 0.01, FREQS)
        assert out.value == ref[0] and gd.value == 0, (out.value, ref)
        t = time.perf_counter()
        for _ in range(calls):
            e.refresh(e.user, k, C.cast(arr, C.c_void_p), 0.01, f7, C.byref(out), C.byref(gd))
        dt = (time.perf_counter() - t) / calls
        print("reads %8d k=%d: engine refresh (fused, two launches) %.1f us per call, identical result" % (n, k, dt * 1e6))
    # the chain driver itself (one chain, k = 2): iterations per second including the host side
    import tempfile
    d = tempfile.mkdtemp()
    t = time.perf_counter()
    sb.estimate(ctx, tree, node_path, [3, 5], d + "/b_", g.n_paths, FREQS, iters=calls, burnin=calls // 10, chains=1, seed=3)
    dt = time.perf_counter() - t
    print("reads %8d: vgan_sb_estimate k=1..2, 1 chain x %d iterations: %.1f us per iteration (summaries included)" % (n, calls, dt / (2 * (calls + 1)) * 1e6))
    t = time.perf_counter()
    sb.estimate(ctx, tree, node_path, [3, 5], d + "/c_", g.n_paths, FREQS, iters=calls, burnin=calls // 10, chains=4, seed=3)
    dt = time.perf_counter() - t
    print("reads %8d: vgan_sb_estimate k=1..2, 4 chains advanced together x %d iterations: %.1f us per iteration of all four (summaries included)"
          % (n, calls, dt / (2 * (calls + 1)) * 1e6))
    ctx.close()
