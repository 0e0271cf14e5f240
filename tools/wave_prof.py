#!/usr/bin/env python3
"""Developer aid (GPU box): N accumulates of a resident, packed 1M-read batch -- the program to put behind rocprofv3.

  python3 tools/wave_prof.py [n_reads] [read_len] [reps]      (VGAN_HC_KERNEL=tile selects the LDS-tiled kernel)
"""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vgan_amd import haplocart as hc  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    g = hc.synth_graph(seed=1)
    a = hc.synth_reads(g, n, seed=2, read_len=rl)
    hb = hc.HostBatch(g, a, packed=True)
    ctx = hc.HcContext(g)
    db = hc.DeviceBatch(hb)
    ctx.profile_enable(True)
    for _ in range(reps):
        ctx.accumulate(db)
    pr = ctx.profile_read()
    print("segment kernel: %.4f ms per launch over %d launches" % (pr["segment"][0] / max(1, pr["segment"][1]), pr["segment"][1]))
    ctx.finalize()


if __name__ == "__main__":
    main()
