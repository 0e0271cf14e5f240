#!/usr/bin/env python3
"""End-to-end times of `vgan euka` and `vgan soibean` on the GPU box (GAM inflate/parse + flatten + GPU + host chains + files).
usage: e2e_rates_euka_soibean.py [n_reads]"""
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
from test_sb_chain_cpu import _newick_of  # noqa: E402
from vgan_amd import euka as ek  # noqa: E402
from vgan_amd import haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
exe = os.path.join(ROOT, "vgan_amd/bin/vgan")
gold = os.path.join(ROOT, "tests/golden/damageProfiles")
p5, p3 = gold + "/dhigh5p.prof", gold + "/dhigh3p.prof"
env = dict(os.environ, VGAN_TIMING="1")

d = tempfile.mkdtemp()
dm = ek.Damage.load(p5, p3)
g, db, a = ek.synth_euka(n, dm)
util.write_euka_db(db, g, d)
a.write_gam(d + "/e.gam")
for extra, what in ((["--seed", "3"], "10000 MCMC iterations"), (["--no-mcmc"], "--no-mcmc")):
    t = time.time()
    r = subprocess.run([exe, "euka", "-g", d + "/e.gam", "--euka_dir", d, "--deam5p", p5, "--deam3p", p3, "-o", d + "/eo", "-t", "-1"] + extra,
                       capture_output=True, text=True, env=env)
    dt = time.time() - t
    print("vgan euka end to end, %s, rc=%d: %.2f s, %.0f reads/s (GAM %.0f MB)" % (what, r.returncode, dt, n / dt, os.path.getsize(d + "/e.gam") / 1e6))
    print("\n".join(l for l in r.stderr.splitlines() if "timing" in l or "Number of" in l))
print(open(d + "/eo_detected.tsv").read()[:600])
shutil.rmtree(d)

d = tempfile.mkdtemp()
g = hc.synth_graph(seed=17, genome_len=16000, n_nodes=11000, n_paths=28)
a = hc.synth_reads(g, n, seed=6, read_len=60, indel_rate=0.1, softclip_rate=0.1)
os.makedirs(d + "/tree_dir")
g.write(d)
os.rename(d + "/graph.gfa", d + "/Synth.gfa")
open(d + "/tree_dir/Synth.new.dnd", "w").write(_newick_of(g))
open(d + "/soibean_db.baseFreq", "w").write("Synth .31 .25 .15 .29\n")
a.write_gam(d + "/s.gam")
for iters, burn in ((20000, 3000),):
    t = time.time()
    r = subprocess.run([exe, "soibean", "-g", d + "/s.gam", "--soibean_dir", d, "--dbprefix", "Synth", "--deam5p", p5, "--deam3p", p3, "-k", "2",
                        "--iter", str(iters), "--burnin", str(burn), "--chains", "4", "--seed", "5", "-o", d + "/bean_", "-t", "-1"],
                       capture_output=True, text=True, env=env)
    dt = time.time() - t
    print("vgan soibean end to end, k=1..2 x 4 chains x %d iterations = %d likelihood refreshes over %d reads, rc=%d: %.2f s"
          % (iters, 2 * 4 * (iters + 1), n, r.returncode, dt))
    print("\n".join(l for l in r.stderr.splitlines() if "timing" in l or "Number of" in l or "Initial" in l))
    if r.returncode:
        print(r.stderr[-1500:])
print(open(d + "/bean_ProportionEstimates2.txt").read()[:800])
shutil.rmtree(d)
