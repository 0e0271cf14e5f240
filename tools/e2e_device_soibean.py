#!/usr/bin/env python3
"""GPU box: `vgan soibean` end to end on one synthetic GAM (BASELINE configs[4]'s shape: 2 M reads of 65 bp against a 28-path tree, a few
hundred chain iterations so that the front end is what the run's length follows), host pipeline (VGAN_SB_DEVICE_GAM=0) against the front
end on the device (=1): python3 tools/e2e_device_soibean.py [n_reads]"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from vgan_amd import haplocart as hc  # noqa: E402
from test_sb_chain_cpu import _newick_of  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
exe = os.path.join(ROOT, "vgan_amd/bin/vgan")
gold = os.path.join(ROOT, "tests/golden/damageProfiles")
p5, p3 = gold + "/dhigh5p.prof", gold + "/dhigh3p.prof"
d = tempfile.mkdtemp(dir="/tmp")
g = hc.synth_graph(seed=1, genome_len=16569, n_nodes=11000, n_paths=28)
os.makedirs(d + "/db/tree_dir")
g.write(d + "/db")
shutil.move(d + "/db/graph.gfa", d + "/db/Synth.gfa")
open(d + "/db/tree_dir/Synth.new.dnd", "w").write(_newick_of(g) + "\n")
open(d + "/db/soibean_db.baseFreq", "w").write("Synth .31 .25 .15 .29\n")
CH = 1000000
t0 = time.time()
with open(d + "/s.gam", "wb") as f:  # (chunks of 1 M reads: BGZF files concatenate)
    for c0 in range(0, n, CH):
        a = hc.synth_reads(g, min(CH, n - c0), seed=1, read_len=65, indel_rate=0.005, softclip_rate=0.01, first_read=c0)
        a.write_gam(d + "/part.gam")
        blob = open(d + "/part.gam", "rb").read()
        f.write(blob[:-28] if c0 + CH < n else blob)
        del a
print("GAM of %d reads: %.1f MB, written in %.0f s" % (n, os.path.getsize(d + "/s.gam") / 1e6, time.time() - t0), flush=True)


def files(prefix):
    return {os.path.basename(p)[len(os.path.basename(prefix)):]: open(p, "rb").read() for p in sorted(glob.glob(prefix + "*"))}


outs = {}
for tag, env in (("host", {"VGAN_SB_DEVICE_GAM": "0"}), ("device", {"VGAN_SB_DEVICE_GAM": "1"}), ("host2", {"VGAN_SB_DEVICE_GAM": "0"}), ("device2", {"VGAN_SB_DEVICE_GAM": "1"})):
    t = time.time()
    c0 = os.times()
    r = subprocess.run([exe, "soibean", "-g", d + "/s.gam", "--soibean_dir", d + "/db", "--dbprefix", "Synth", "--deam5p", p5, "--deam3p", p3, "-o", d + "/" + tag + "_", "-t", "-1",
                        "--iter", "300", "--burnin", "50", "--chains", "1", "--seed", "3"], capture_output=True, text=True, env=dict(os.environ, VGAN_TIMING="1", **env))
    c1 = os.times()
    dt = time.time() - t
    cpu = (c1.children_user - c0.children_user) + (c1.children_system - c0.children_system)
    print("%-8s rc=%d  %.2f s wall, %.2f s of host CPU, %.2f M reads/s" % (tag, r.returncode, dt, cpu, n / dt / 1e6), flush=True)
    for ln in r.stderr.splitlines():
        if "device front end" in ln or "does not take" in ln or "Number of" in ln or "[vgan timing] soibean" in ln:
            print("   ", ln[:800])
    if r.returncode:
        print(r.stderr[-1500:])
    outs[tag] = files(d + "/" + tag + "_")
same = sorted(outs["host"]) == sorted(outs["device"]) and all(outs["host"][k] == outs["device"][k] for k in outs["host"])
print("same chain files, byte for byte (%d files): %s" % (len(outs["host"]), same))
shutil.rmtree(d, ignore_errors=True)
