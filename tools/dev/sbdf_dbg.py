import sys, os, numpy as np, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from vgan_amd import euka as ek, haplocart as hc, soibean as sb
from test_sb_gpu import _soibean_case
g, _, profs, newick = _soibean_case(n_reads=10)
a = hc.synth_reads(g, 30000, seed=6, read_len=60, indel_rate=0.1, softclip_rate=0.1)
d = tempfile.mkdtemp(); gam = d + "/r.gam"; a.write_gam(gam)
data = open(gam, "rb").read()
a2 = hc.AlnSet.read_gam(gam, keep_unmapped=False)
hb = sb.SbHostBatch(g, a2); want = hb.arrays()
dm = ek.Damage.from_text(*(open(p).read() for p in profs))
ctx = sb.SbContext(g, dm, penalty=7)
gd = hc.GamDevice().parse(data, keep_unmapped=False)
df = sb.SbDeviceFlatten(ctx, g)
mask = df.append_gamdev(gd, 0)
x = a2.arrays()
R = a2.n_reads
plain = np.array([np.all(x["e_from"][x["edit_off"][x["map_off"][r]]:x["edit_off"][x["map_off"][r + 1]]] == x["e_to"][x["edit_off"][x["map_off"][r]]:x["edit_off"][x["map_off"][r + 1]]]) for r in range(R)])
diff = np.nonzero((mask == 0) != plain)[0]
print(len(diff), "differences; host n_bad", hb.stats.n_bad, "n_out", hb.stats.n_out, "R", R)
hs = set(int(s) for s in want["read_src"])
for r in diff[:8]:
    m0, m1 = x["map_off"][r], x["map_off"][r + 1]
    e0, e1 = x["edit_off"][m0], x["edit_off"][m1]
    print(r, "mask", mask[r], "plain", plain[r], "in host batch", int(r) in hs, "nm", m1 - m0, "ne", e1 - e0, "from", x["e_from"][e0:e1][:6], "seqlen", x["seq_off"][r + 1] - x["seq_off"][r], "rev", x["m_rev"][m0], "sum from", x["e_from"][e0:e1].sum())
