#!/bin/bash
# GPU box: the device front end's parse at several upload piece sizes (VGAN_GAMDEV_PIECE), 10 M reads
set -e
python3 - <<'P'
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from vgan_amd import haplocart as hc
n = 10000000
g = hc.synth_graph()
d = tempfile.mkdtemp(dir="/tmp")
g.write(d)
CH = 1000000
with open(d + "/r.gam", "wb") as f:
    for c0 in range(0, n, CH):
        a = hc.synth_reads(g, min(CH, n - c0), first_read=c0)
        a.write_gam(d + "/part.gam")
        blob = open(d + "/part.gam", "rb").read()
        f.write(blob[:-28] if c0 + CH < n else blob)
        del a
for piece in ("6000000000", "2500000000", "1250000000", "700000000", "6000000000", "1250000000"):
    for rep in range(2):
        t = time.time()
        r = subprocess.run(["vgan_amd/bin/vgan", "haplocart", "-g", d + "/r.gam", "--hc-files", d, "-q", "-t", "-1", "--keep-duplicates", "-o", d + "/o.tsv"],
                           capture_output=True, text=True, env=dict(os.environ, VGAN_TIMING="1", VGAN_HC_DEVICE_GAM="1", VGAN_GAMDEV_PIECE=piece))
        dt = time.time() - t
        ln = [x for x in r.stderr.splitlines() if "device front end" in x]
        print(piece, "%.2f s" % dt, ln[-1][ln[-1].index("parse"):ln[-1].index("duplicate")] if ln else r.stderr[-300:], flush=True)
P
