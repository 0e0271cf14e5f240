#!/usr/bin/env python3
"""PCIe-inclusive and end-to-end rates on the GPU box (not bench.py's `value`, which starts with inputs in HBM)."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vgan_amd import haplocart as hc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g = hc.synth_graph()
a = hc.synth_reads(g, n)
hb = hc.HostBatch(g, a)
ctx = hc.HcContext(g)
ctx.accumulate(hb); ctx.finalize()  # warm up (staging buffers allocated)
t = time.time()
for _ in range(5):
    ctx.reset(); ctx.accumulate(hb); ctx.finalize()
dt = (time.time() - t) / 5
print("host-batch accumulate+finalize (H2D over PCIe inclusive): %.2f ms, %.1f M reads/s" % (dt * 1e3, hb.n_reads / dt / 1e6))
d = tempfile.mkdtemp()
g.write(d)
a.write_gam(d + "/r.gam")
t = time.time()
r = subprocess.run([os.path.join(ROOT, "vgan_amd/bin/vgan"), "haplocart", "-g", d + "/r.gam", "--hc-files", d, "-q", "-t", "-1",
                    "--keep-duplicates", "-o", d + "/out.tsv", "-pf", d + "/post.txt"], capture_output=True, text=True)
dt = time.time() - t
print("vgan haplocart end to end (graph load + GAM inflate/parse + flatten + GPU + posterior), rc=%d: %.2f s, %.0f reads/s"
      % (r.returncode, dt, n / dt))
print(open(d + "/out.tsv").read().strip())
t = time.time()
r2 = subprocess.run([os.path.join(ROOT, "vgan_amd/bin/vgan"), "haplocart", "-g", d + "/r.gam", "--hc-files", d, "-q", "-t", "-1",
                     "-o", d + "/out2.tsv", "-pf", d + "/post2.txt"], capture_output=True, text=True)
dt = time.time() - t
print("same with duplicate removal (dedup on, the reference's default), rc=%d: %.2f s, %.0f input reads/s" % (r2.returncode, dt, n / dt))
print(open(d + "/out2.tsv").read().strip().splitlines()[-1])
print("\n".join(l for l in r2.stderr.splitlines() if "haplocart:" in l))
print("\n".join(l for l in r.stderr.splitlines() if "haplocart:" in l or "parse_gam" in l or "warning" in l))
