import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vgan_amd import euka as ek
GOLD = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden")
d = os.path.join(GOLD, "damageProfiles")
texts = (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read())
dm = ek.Damage.from_text(*texts)
g, db, a = ek.synth_euka(1_000_000, dm)
ctx = ek.EukaContext(db, dm)
whole = ek.EukaHostBatch(g, a)
res = []
for rep in range(3):
    ctx.reset()
    got = ctx.accumulate(whole)
    fin = ctx.finalize()
    n, s = ctx.like_sums()
    res.append((n.copy(), s.copy()))
s0 = res[0][1]
print("finite clades", np.isfinite(s0).sum(), "of", len(s0), "nan", np.isnan(s0).sum())
for i in (1, 2):
    f = np.isfinite(s0)
    print("rep", i, "finite sets equal", np.array_equal(np.isfinite(res[i][1]), f), "max rel", np.max(np.abs(res[i][1][f] - s0[f]) / np.abs(s0[f])) if f.any() else None, "n equal", np.array_equal(res[i][0], res[0][0]))
ctx.reset()
half = a.n_reads // 2 + 3
ctx.accumulate(ek.EukaHostBatch(g, a, 0, half)); ctx.accumulate(ek.EukaHostBatch(g, a, half, a.n_reads))
ctx.finalize()
n2, s2 = ctx.like_sums()
f = np.isfinite(s0)
print("shards: finite sets equal", np.array_equal(np.isfinite(s2), f), "n equal", np.array_equal(n2, res[0][0]))
both = f & np.isfinite(s2)
rel = np.abs(s2[both] - s0[both]) / np.abs(s0[both])
print("max rel", rel.max(), "clades differing in finiteness", np.nonzero(np.isfinite(s2) != f)[0][:10], s2[np.isfinite(s2) != f][:5], s0[np.isfinite(s2) != f][:5])
i = np.argmax(rel); idx = np.nonzero(both)[0][i]
print("worst clade", idx, "n", n2[idx], "whole", s0[idx], "shards", s2[idx])
