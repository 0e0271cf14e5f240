#!/usr/bin/env python3
"""Developer aid (GPU box): `vgan haplocart` on an n-read synthetic GAM with VGAN_TIMING=1 -- the stage timeline and the
peak resident set.  usage: python3 tools/e2e_timeline.py [n_reads] [repeats] [ENV=value,ENV=value ...: a second set of runs]"""
import os
import resource
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vgan_amd import haplocart as hc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = hc.synth_graph()
d = tempfile.mkdtemp()
g.write(d)
CH = 1000000
alns = []
for c0 in range(0, n, CH):
    a = hc.synth_reads(g, min(CH, n - c0), first_read=c0)
    a.write_gam(d + "/part%d.gam" % (c0 // CH))
    del a
with open(d + "/r.gam", "wb") as out:  # gzip members concatenate
    for c0 in range(0, n, CH):
        out.write(open(d + "/part%d.gam" % (c0 // CH), "rb").read())
print("GAM: %d reads, %.1f MB" % (n, os.path.getsize(d + "/r.gam") / 1e6), flush=True)
exe = os.path.join(ROOT, "vgan_amd/bin/vgan")
variants = [{}] + [dict(kv.split("=", 1) for kv in v.split(",")) for v in sys.argv[3:]]
for extra, i in [(dict(e), i) for e in variants for i in range(reps)]:
    if i == 0 and extra:
        print("with", extra, flush=True)
    t = time.time()
    dedup = extra.pop("DEDUP", None)  # DEDUP=1: the reference's default (duplicates removed); else --keep-duplicates
    p = subprocess.Popen([exe, "haplocart", "-g", d + "/r.gam", "--hc-files", d, "-q", "-t", "-1"] + ([] if dedup else ["--keep-duplicates"]) +
                         ["-o", d + "/out%d.tsv" % i, "-np"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, VGAN_TIMING="1", **extra))
    peak = {}

    def poll():  # the child's resident set while it runs (anonymous / file-backed / shared), sampled every 20 ms
        while p.poll() is None:
            try:
                for ln in open("/proc/%d/status" % p.pid):
                    k = ln.split(":")[0]
                    if k in ("VmRSS", "RssAnon", "RssFile", "RssShmem"):
                        v = int(ln.split()[1])
                        if v > peak.get(k, (0, 0))[0]:
                            peak[k] = (v, time.time() - t)
            except OSError:
                pass
            time.sleep(0.02)
    import threading
    th = threading.Thread(target=poll)
    th.start()
    _, err = p.communicate()
    dt = time.time() - t
    th.join()
    if peak:
        print("        sampled peaks: " + ", ".join("%s %.2f GB at %.2f s" % (k, v / 1e6, at) for k, (v, at) in sorted(peak.items())), flush=True)
    # (ru_maxrss of the child is of no use here: it starts as a copy of this process, which holds the synthetic reads)
    print("run %d: rc=%d wall %.3f s = %.2f M reads/s" % (i, p.returncode, dt, n / dt / 1e6), flush=True)
    clk = {l.split()[5].rstrip(":"): float(l.split()[-1]) for l in err.splitlines() if "wall clock at" in l}
    if "main" in clk and "exit" in clk:
        print("        spawn -> main %.0f ms, main -> _exit %.0f ms, _exit -> reaped %.0f ms" % (
            (clk["main"] - t) * 1e3, (clk["exit"] - clk["main"]) * 1e3, (t + dt - clk["exit"]) * 1e3), flush=True)
    if i == reps - 1:
        print("\n".join(l for l in err.splitlines() if "timing" in l and "wall clock" not in l))
