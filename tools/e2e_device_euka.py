#!/usr/bin/env python3
"""GPU box: `vgan euka --no-mcmc` end to end on one synthetic GAM of 75 bp aDNA-like reads (BASELINE configs[3] is 5 M of them), host
pipeline (VGAN_EUKA_DEVICE_GAM=0) against the front end on the device (=1): python3 tools/e2e_device_euka.py [n_reads]"""
import atexit
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from vgan_amd import euka as ek  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
exe = os.path.join(ROOT, "vgan_amd/bin/vgan")
gold = os.path.join(ROOT, "tests/golden/damageProfiles")
p5, p3 = gold + "/dhigh5p.prof", gold + "/dhigh3p.prof"
d = tempfile.mkdtemp(dir="/tmp")
atexit.register(shutil.rmtree, d, True)  # (5 GB per run: a box that is used again would fill up)
dm = ek.Damage.load(p5, p3)
CH = 1000000
t0 = time.time()
with open(d + "/e.gam", "wb") as f:  # (chunks of 1 M reads: BGZF files concatenate)
    for c0 in range(0, n, CH):
        g, db, a = ek.synth_euka(min(CH, n - c0), dm, read_seed=1 + c0)
        if c0 == 0:
            util.write_euka_db(db, g, d)
        a.write_gam(d + "/part.gam")
        blob = open(d + "/part.gam", "rb").read()
        f.write(blob[:-28] if c0 + CH < n else blob)
        del a
print("GAM of %d reads: %.1f MB, written in %.0f s" % (n, os.path.getsize(d + "/e.gam") / 1e6, time.time() - t0), flush=True)
outs = {}
for tag, env in (("host", {"VGAN_EUKA_DEVICE_GAM": "0"}), ("device", {"VGAN_EUKA_DEVICE_GAM": "1"}), ("host2", {"VGAN_EUKA_DEVICE_GAM": "0"}), ("device2", {"VGAN_EUKA_DEVICE_GAM": "1"})):
    t = time.time()
    c0 = os.times()
    r = subprocess.run([exe, "euka", "-g", d + "/e.gam", "--euka_dir", d, "--deam5p", p5, "--deam3p", p3, "-o", d + "/" + tag, "-t", "-1", "--no-mcmc"],
                       capture_output=True, text=True, env=dict(os.environ, VGAN_TIMING="1", **env))
    c1 = os.times()
    dt = time.time() - t
    cpu = (c1.children_user - c0.children_user) + (c1.children_system - c0.children_system)
    print("%-8s rc=%d  %.2f s wall, %.2f s of host CPU, %.2f M reads/s" % (tag, r.returncode, dt, cpu, n / dt / 1e6), flush=True)
    for ln in r.stderr.splitlines():
        if "device front end" in ln or "does not take" in ln or "Number of" in ln or ("[vgan timing] euka:" in ln) or ("gampipe piece" in ln and tag == "device2" and os.environ.get("E2E_PIECES")):
            print("   ", ln[:700])
    if r.returncode:
        print(r.stderr[-1500:])
    outs[tag] = {k: open(d + "/" + tag + k).read() for k in ("_abundance.tsv", "_detected.tsv", "_coverage.tsv")}
print("same detected.tsv and abundance.tsv:", outs["host"]["_detected.tsv"] == outs["device"]["_detected.tsv"], outs["host"]["_abundance.tsv"] == outs["device"]["_abundance.tsv"])
