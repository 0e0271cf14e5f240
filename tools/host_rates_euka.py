#!/usr/bin/env python3
"""Host front-end throughput of the euka and soibean flatten steps on this machine's cores (developer aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgan_amd import euka as ek
from vgan_amd import haplocart as hc
from vgan_amd import soibean as sb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "damageProfiles")
dm = ek.Damage.load(gold + "/dhigh5p.prof", gold + "/dhigh3p.prof")
t = time.time(); g, db, a = ek.synth_euka(n, dm); print("synth_euka %.2fs" % (time.time() - t))
for th in (1, 0):
    t = time.time(); hb = ek.EukaHostBatch(g, a, n_threads=th); dt = time.time() - t
    print("euka flatten threads=%s %.2fs %.0f reads/s" % (th or "all", dt, n / dt))
g2 = hc.synth_graph(genome_len=16569, n_nodes=11000, n_paths=28)
a2 = hc.synth_reads(g2, n, read_len=65)
for th in (1, 0):
    t = time.time(); hb2 = sb.SbHostBatch(g2, a2, n_threads=th); dt = time.time() - t
    print("soibean flatten threads=%s %.2fs %.0f reads/s" % (th or "all", dt, n / dt))
print("cores", os.cpu_count())
