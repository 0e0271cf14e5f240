#!/usr/bin/env python3
"""GPU box: `vgan haplocart` end to end on one synthetic GAM, host pipeline (VGAN_HC_DEVICE_GAM=0) against the front end on the device
(=1), with VGAN_TIMING's lines (python3 tools/e2e_device_gam.py [n_reads] [keep|dedup])."""
import atexit
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vgan_amd import haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dedup = len(sys.argv) > 2 and sys.argv[2] == "dedup"
g = hc.synth_graph()
d = tempfile.mkdtemp(dir="/tmp")
atexit.register(shutil.rmtree, d, True)  # (5 GB per run: a box that is used again would fill up)
g.write(d)
t0 = time.time()
CH = 1000000
with open(d + "/r.gam", "wb") as f:  # (chunks of 1 M reads: BGZF files concatenate)
    for c0 in range(0, n, CH):
        a = hc.synth_reads(g, min(CH, n - c0), first_read=c0)
        a.write_gam(d + "/part.gam")
        blob = open(d + "/part.gam", "rb").read()
        f.write(blob[:-28] if c0 + CH < n else blob)  # (drop the BGZF end-of-file member of all but the last part)
        del a
print("GAM of %d reads: %.1f MB, written in %.0f s" % (n, os.path.getsize(d + "/r.gam") / 1e6, time.time() - t0), flush=True)
outs = {}
for tag, env in (("host", {"VGAN_HC_DEVICE_GAM": "0"}), ("device", {"VGAN_HC_DEVICE_GAM": "1"}), ("host2", {"VGAN_HC_DEVICE_GAM": "0"}), ("device2", {"VGAN_HC_DEVICE_GAM": "1"})):
    t = time.time()
    cmd = [os.environ.get("VGAN_EXE") or os.path.join(ROOT, "vgan_amd/bin/vgan"), "haplocart", "-g", d + "/r.gam", "--hc-files", d, "-q", "-t", "-1", "-d", "-o", d + "/" + tag + ".tsv"]
    if not dedup:
        cmd.append("--keep-duplicates")
    c0 = os.times()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, VGAN_TIMING="1", **env))
    c1 = os.times()
    dt = time.time() - t
    cpu = (c1.children_user - c0.children_user) + (c1.children_system - c0.children_system)
    print("%-8s rc=%d  %.2f s wall, %.2f s of host CPU (%.2f us per read), %.2f M reads/s" % (tag, r.returncode, dt, cpu, cpu / n * 1e6, n / dt / 1e6), flush=True)
    for ln in r.stderr.splitlines():
        if "device front end" in ln or "code objects" in ln or "HIP runtime up" in ln or "does not take" in ln or (("gampipe piece" in ln or "hc consume" in ln or "hc_devflat" in ln) and tag == "device2") or ("haplocart @" in ln and ("contexts ready" in ln or "on the device" in ln)):
            print("   ", ln[:1200])
    if tag == "device2" and os.environ.get("E2E_FULL"):
        print(r.stderr[-6000:])
    if r.returncode:
        print(r.stderr[-1500:])
    outs[tag] = open(d + "/" + tag + ".tsv").read().splitlines()[-1] if os.path.exists(d + "/" + tag + ".tsv") else None
print(outs["host"], "|", outs["device"])
print("same result line:", outs["host"] == outs["device"])
ll = {}
for tag in ("host", "device"):
    ll[tag] = dict((ln.split("\t")[0], float(ln.split("\t")[1])) for ln in open(d + "/" + tag + ".tsv.loglik.tsv").read().splitlines())
print("max rel diff of the log-likelihoods:", max(abs(ll["host"][k] - ll["device"][k]) / abs(ll["host"][k]) for k in ll["host"]))
if outs["host"] != outs["device"]:  # (paths the reads do not tell apart: the sums' last bits -- the order of the additions -- pick among them)
    a, b = outs["host"].split("\t")[1], outs["device"].split("\t")[1]
    print("predicted %s against %s: their log-likelihoods as printed (10 digits) %r and %r -- %s" % (
        a, b, ll["host"][a], ll["host"][b], "a tie" if ll["host"][a] == ll["host"][b] else "NOT a tie"))
