#!/usr/bin/env python3
"""Device fuzz (developer aid, GPU box): alignments corrupted at random (node ids, edit lengths, offsets, strands,
qualities) that still pass the flatten step, through the euka, HaploCart and soibean device paths against the oracle,
including the counts of reads each side refuses.  usage: gpu_device_fuzz.py [n_seeds]"""

# ---- euka
import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from vgan_amd import _native as N, haplocart as hc, euka as ek, soibean as sb
import orc, util
gold = "tests/golden/damageProfiles"
texts = (open(gold + "/dhigh5p.prof").read(), open(gold + "/dhigh3p.prof").read())
dm = ek.Damage.from_text(*texts)
g, db, a = ek.synth_euka(600, dm, seed=3, n_clades=6, nodes_per_clade=120, read_len_mean=60)
base = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in a.arrays().items()}
og = util.orc_graph_nodes_only(g)
odb = util.orc_euka_db_from_product(db)
worst = 0.0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    rng = np.random.default_rng(seed)
    arr = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in base.items()}
    for _ in range(int(rng.integers(1, 250))):
        w = rng.integers(6)
        if w == 0: arr["m_node"][rng.integers(len(arr["m_node"]))] = rng.integers(1, g.max_id + 1)
        elif w == 1: arr["e_from"][rng.integers(len(arr["e_from"]))] = rng.integers(0, 6)
        elif w == 2: arr["e_to"][rng.integers(len(arr["e_to"]))] = rng.integers(0, 6)
        elif w == 3: arr["m_offset"][rng.integers(len(arr["m_offset"]))] = rng.integers(0, 4)
        elif w == 4: arr["m_rev"][rng.integers(len(arr["m_rev"]))] ^= 1
        elif w == 5: arr["qual"][rng.integers(len(arr["qual"]))] = rng.choice([0, 1, 2, 41, 93, 128, 255])
    oa = orc.AlnSet.from_arrays(**arr)
    v = N.AlnSetView(oa.n_reads, *[getattr(oa, k).ctypes.data for k in ("seq_off", "seq", "qual_off", "qual", "mapq", "identity")],
                     None, None, *[getattr(oa, k).ctypes.data for k in ("map_off", "m_node", "m_offset", "m_rev", "edit_off", "e_from", "e_to", "e_seq_off", "e_seq")])
    h = N.vp(); N.check(N.lib().vgan_aln_from_arrays(v, h)); a2 = hc.AlnSet(h)
    hb = ek.EukaHostBatch(g, a2)
    ctx = ek.EukaContext(db, dm)
    got = ctx.accumulate(hb); fin = ctx.finalize()
    ref = orc.euka_run(og, oa, odb, orc.OrcDamage(*texts))
    src = hb.arrays()["read_src"]
    assert fin["n_bad"] + hb.stats.n_bad == ref["n_bad"], (seed, fin["n_bad"], hb.stats.n_bad, ref["n_bad"])
    assert np.array_equal(got["clade"], ref["clade"][src]), seed
    ok = got["clade"] >= 0
    for k in ("in_lik", "out_lik", "like"):
        gv, rv = got[k][ok], ref[k][src][ok]
        fin_mask = np.isfinite(rv)
        assert np.array_equal(np.isfinite(gv), fin_mask), (seed, k)
        e = util.rel_err(gv[fin_mask], rv[fin_mask]); worst = max(worst, e)
        assert e < 1e-9, (seed, k, e)
    assert np.array_equal(got["pass"], ref["pass"][src]) and np.array_equal(fin["clade_count"], ref["clade_count"]) and np.array_equal(fin["baseshift"], ref["baseshift"])
    assert np.allclose(fin["bin_cov"], ref["bin_cov"], rtol=1e-12, atol=1e-12)
print("euka device fuzz ok, worst rel err %.3g" % worst)

# ---- HaploCart + soibean
g = hc.synth_graph(seed=4, genome_len=900, n_nodes=620, n_paths=28)
a = hc.synth_reads(g, 300, seed=1, read_len=70, indel_rate=0.3, softclip_rate=0.3)
base = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in a.arrays().items()}
og = util.orc_graph_from_product(g)
FREQS = [0.31, 0.27, 0.13, 0.29, 0.44, 0.56, 0.0012]
names = g.path_names; idx = {n: i for i, n in enumerate(names)}
pairs = [(idx[t[0]], idx[t[1]]) for t in (ln.split() for ln in g.parents_txt.splitlines()) if len(t) >= 2]
worst = [0.0, 0.0]
n_irr = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    rng = np.random.default_rng(seed)
    arr = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in base.items()}
    for _ in range(int(rng.integers(1, 200))):
        w = rng.integers(6)
        if w == 0: arr["m_node"][rng.integers(len(arr["m_node"]))] = rng.integers(1, g.max_id + 1)
        elif w == 1: arr["e_from"][rng.integers(len(arr["e_from"]))] = rng.integers(0, 6)
        elif w == 2: arr["e_to"][rng.integers(len(arr["e_to"]))] = rng.integers(0, 6)
        elif w == 3: arr["m_offset"][rng.integers(len(arr["m_offset"]))] = rng.integers(0, 4)
        elif w == 4: arr["m_rev"][rng.integers(len(arr["m_rev"]))] ^= 1
        elif w == 5: arr["qual"][rng.integers(len(arr["qual"]))] = rng.choice([0, 1, 2, 41, 93, 128, 255])
    oa = orc.AlnSet.from_arrays(**arr)
    v = N.AlnSetView(oa.n_reads, *[getattr(oa, k).ctypes.data for k in ("seq_off", "seq", "qual_off", "qual", "mapq", "identity")],
                     None, None, *[getattr(oa, k).ctypes.data for k in ("map_off", "m_node", "m_offset", "m_rev", "edit_off", "e_from", "e_to", "e_seq_off", "e_seq")])
    h = N.vp(); N.check(N.lib().vgan_aln_from_arrays(v, h)); a2 = hc.AlnSet(h)
    b = hc.HostBatch(g, a2)
    n_irr += b.n_reads - b.n_tileable
    _, ref, bad = orc.hc_run(og, oa, n_threads=4, faithful=False)
    assert bad == b.stats.n_bad
    ctx = hc.HcContext(g)
    for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ):
        ctx.reset(); ctx.set_mode(mode); ctx.accumulate(b); got = ctx.finalize()
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(got), fin), (seed, mode)
        e = util.rel_err(got[fin], ref[fin]); worst[0] = max(worst[0], e)
        assert e < 1e-9, (seed, mode, e)
    # soibean on the same reads
    dm = ek.Damage.from_text("", "")
    hb = sb.SbHostBatch(g, a2)
    sctx = sb.SbContext(g, dm)
    dev_bad = sctx.precompute(hb)
    o = orc.SbOracle(og, oa, orc.OrcDamage("", ""))
    assert o.n_bad == hb.stats.n_bad + dev_bad, (seed, o.n_bad, hb.stats.n_bad, dev_bad)
    st = [[(pairs[(seed + y) % len(pairs)][0], pairs[(seed + y) % len(pairs)][1], 0.01 + 0.01 * y, 0.3, 1 / 3) for y in range(3)]]
    gotl, guard = sctx.loglike(st, 0.01, FREQS)
    rc, refl = o.loglike(st[0], 0.01, FREQS)
    if rc == 0 and np.isfinite(refl):
        e = abs(gotl[0] - refl) / abs(refl); worst[1] = max(worst[1], e)
        assert e < 1e-9, (seed, gotl[0], refl)
    else:
        assert guard[0] > 0 or not np.isfinite(gotl[0]), (seed, rc, refl, gotl, guard)
print("hc/soibean device fuzz ok, worst rel err %.3g / %.3g, non-tileable reads seen %d" % (worst[0], worst[1], n_irr))
