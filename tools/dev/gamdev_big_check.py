#!/usr/bin/env python3
"""GPU box: the device front end's arrays against the host parser's on a multi-segment, multi-million-read file, array for array
(python3 tools/dev/gamdev_big_check.py [n_reads])."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("VGAN_POISON_ALLOCS", "1")
import numpy as np  # noqa: E402
from vgan_amd import haplocart as hc  # noqa: E402
import test_gamdev_gpu as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
g = hc.synth_graph()
d = tempfile.mkdtemp(dir="/tmp")
CH = 1000000
with open(d + "/r.gam", "wb") as f:
    for c0 in range(0, n, CH):
        a = hc.synth_reads(g, min(CH, n - c0), first_read=c0, indel_rate=0.02, softclip_rate=0.02)
        a.write_gam(d + "/part.gam")
        blob = open(d + "/part.gam", "rb").read()
        f.write(blob[:-28] if c0 + CH < n else blob)
        del a
data = open(d + "/r.gam", "rb").read()
print("GAM of %d reads: %.1f MB" % (n, len(data) / 1e6), flush=True)
gd = T.GamDev()
got = T.check_against_host(gd, data, False)
print("device arrays == host parser's arrays for %d reads (%d mappings, %d edits)" % (got, gd.sizes["M"], gd.sizes["E"]), flush=True)
gd.close()
