"""Parity at BASELINE.json's workload sizes (1M and 10M x 150 bp HaploCart, 1M and 5M euka reads, 2M-read soibean) through properties
that do not depend on the size: modes agree, accumulation is additive over any split of the reads, summaries of disjoint
shards add up, and a random sample of reads matches the oracle.  (The oracle itself finishes only a few hundred reads of
these workloads in seconds.)"""
import os

import numpy as np
import pytest

import orc
import util
from vgan_amd import euka as ek
from vgan_amd import haplocart as hc
from vgan_amd import soibean as sb
from test_euka_cpu import GOLD
from test_sb_cpu import FREQS

pytestmark = pytest.mark.gpu


def _range(a, r0, r1):
    """The reads [r0, r1) of an alignment set as a set of their own."""
    drop = np.ones(a.n_reads, np.uint8)
    drop[r0:r1] = 0
    return a.without(drop)


def test_haplocart_one_million_reads():
    g = hc.synth_graph(seed=0x76676131)
    a = hc.synth_reads(g, 1_000_000, seed=0x76676131, read_len=150)
    whole = hc.HostBatch(g, a)
    assert whole.n_reads > 999_000 and whole.stats.n_bad == 0
    ctx = hc.HcContext(g)
    ctx.accumulate(whole)
    ref = ctx.finalize()
    assert np.all(np.isfinite(ref)) and ref.max() < 0
    # the faithful per-read sweep (mask rows streamed per segment) against the collapsed node-weight form
    ctx.reset()
    ctx.set_mode(hc.MODE_PER_READ)
    ctx.accumulate(whole)
    assert util.rel_err(ctx.finalize(), ref) < 1e-10
    # additive over an uneven three-way split, in any order
    ctx.reset()
    ctx.set_mode(hc.MODE_NODE_WEIGHTS)
    cuts = [0, 123_457, 700_001, a.n_reads]
    for i in (2, 0, 1):
        ctx.accumulate(hc.HostBatch(g, a, cuts[i], cuts[i + 1]))
    assert util.rel_err(ctx.finalize(), ref) < 1e-11
    # every path's total is (sum over reads of S) minus the unsupported penalties: doubling the input doubles it
    ctx.accumulate(whole)
    assert util.rel_err(ctx.finalize(), 2 * ref) < 1e-11
    # a scattered sample of reads against the oracle
    rng = np.random.default_rng(1)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    for r0 in rng.integers(0, a.n_reads - 40, 4):
        _, want, _ = orc.hc_run(og, oa, r0=int(r0), r1=int(r0) + 40, n_threads=8, faithful=False)
        ctx.reset()
        ctx.accumulate(hc.HostBatch(g, a, int(r0), int(r0) + 40))
        assert util.rel_err(ctx.finalize(), want) < 1e-9


def test_euka_one_million_reads():
    d = os.path.join(GOLD, "damageProfiles")
    texts = (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read())
    dm = ek.Damage.from_text(*texts)
    g, db, a = ek.synth_euka(1_000_000, dm)
    ctx = ek.EukaContext(db, dm)
    whole = ek.EukaHostBatch(g, a)
    got = ctx.accumulate(whole)
    fin = ctx.finalize()
    n, s = ctx.like_sums()
    assert fin["clade_count"].sum() == got["pass"].sum() > 300_000 and n.sum() == (got["clade"] >= 0).sum()
    # two shards: counts and baseshift add exactly, coverage and likelihood sums to rounding
    ctx.reset()
    half = a.n_reads // 2 + 3
    parts = [ctx.accumulate(ek.EukaHostBatch(g, a, 0, half)), ctx.accumulate(ek.EukaHostBatch(g, a, half, a.n_reads))]
    fin2 = ctx.finalize()
    n2, s2 = ctx.like_sums()
    assert np.array_equal(fin2["clade_count"], fin["clade_count"]) and np.array_equal(fin2["baseshift"], fin["baseshift"])
    assert np.allclose(fin2["bin_cov"], fin["bin_cov"], rtol=1e-12, atol=1e-9) and np.array_equal(n2, n)
    okc = np.isfinite(s)
    assert np.array_equal(np.isfinite(s2), okc) and util.rel_err(s2[okc], s[okc]) < 1e-12
    # per read, by its index in the input (a batch holds its reads ordered by their first node, not in input order)
    def by_read(res, batch):
        src = batch.arrays()["read_src"]
        out = {k: np.zeros(a.n_reads, res[k].dtype) for k in ("clade", "pass", "like")}
        for k in out:
            out[k][src] = res[k]
        held = np.zeros(a.n_reads, bool)
        held[src] = True
        return out, held
    w, w_held = by_read(got, whole)
    shard_batches = [ek.EukaHostBatch(g, a, 0, half), ek.EukaHostBatch(g, a, half, a.n_reads)]
    seen = np.zeros(a.n_reads, bool)
    for res, hb in zip(parts, shard_batches):
        sub, sel = by_read(res, hb)
        assert not (sel & seen).any() and w_held[sel].all()
        seen |= sel
        for k in ("clade", "pass", "like"):  # like: the same bits whatever the batch
            assert np.array_equal(sub[k][sel], w[k][sel]), k
    assert np.array_equal(seen, w_held)
    # samples of 300 consecutive reads against the oracle
    og, odb = util.orc_graph_nodes_only(g), util.orc_euka_db_from_product(db)
    src = whole.arrays()["read_src"]
    for r0 in (0, 431_007, a.n_reads - 300):
        sub = _range(a, r0, r0 + 300)  # stays alive: the oracle's view borrows its arrays
        ref = orc.euka_run(og, util.orc_alnset_from_product(sub), odb, orc.OrcDamage(*texts), 29, 5)
        idx = np.nonzero((src >= r0) & (src < r0 + 300))[0]
        assert np.array_equal(got["clade"][idx], ref["clade"][src[idx] - r0]) and np.array_equal(got["pass"][idx], ref["pass"][src[idx] - r0])
        ok = got["clade"][idx] >= 0
        assert ok.sum() > 250 and util.rel_err(got["like"][idx][ok], ref["like"][src[idx] - r0][ok]) < 1e-10


def test_soibean_two_million_reads():
    g = hc.synth_graph(seed=0x76676131, genome_len=16569, n_nodes=11000, n_paths=28)
    a = hc.synth_reads(g, 2_000_000, seed=9, read_len=65, indel_rate=0.005, softclip_rate=0.01)
    dm = ek.Damage.from_text("", "")
    idx = {n: i for i, n in enumerate(g.path_names)}
    pairs = [(idx[t[0]], idx[t[1]]) for t in (ln.split() for ln in g.parents_txt.splitlines()) if len(t) >= 2]
    st = [[(pairs[1][0], pairs[1][1], 0.03, 0.35, 0.5), (pairs[7][0], pairs[7][1], 0.011, 0.8, 0.3), (pairs[12][0], pairs[12][1], 0.04, 0.02, 0.2)]]
    ctx = sb.SbContext(g, dm)
    ctx.precompute(sb.SbHostBatch(g, a))
    whole, guard = ctx.loglike(st, 0.01, FREQS)
    fused, g2 = ctx.refresh(st[0], 0.01, FREQS)
    assert guard[0] == 0 and g2 == 0 and fused == whole[0]  # the chain driver's fused path: the same bits
    _, sig, n_ok = ctx.best_paths()
    mix = ctx.mixture_loglike([3, 9, 20], float(np.log(1 / 3)))
    # shards: the log-likelihood of a state, the signature counts and the initial mixture are sums over the reads
    parts, sigs, mixes, oks = 0.0, 0, 0.0, 0
    cuts = [0, 777_777, a.n_reads]
    for i in range(2):
        ctx.precompute(sb.SbHostBatch(g, a, cuts[i], cuts[i + 1]))
        parts += ctx.loglike(st, 0.01, FREQS)[0][0]
        _, s_i, n_i = ctx.best_paths()
        sigs, oks = sigs + s_i, oks + n_i
        mixes += ctx.mixture_loglike([3, 9, 20], float(np.log(1 / 3)))
    assert parts == pytest.approx(whole[0], rel=1e-12) and mixes == pytest.approx(mix, rel=1e-12)
    assert np.array_equal(sigs, sig) and oks == n_ok > 1_990_000
    # a sample of 200 consecutive reads against the oracle
    og = util.orc_graph_from_product(g)
    sub = _range(a, 1_000_000, 1_000_200)  # stays alive: the oracle's view borrows its arrays
    o = orc.SbOracle(og, util.orc_alnset_from_product(sub), orc.OrcDamage("", ""), penalty=7, path_findable=np.ones(g.n_paths, np.uint8))
    ctx.precompute(sb.SbHostBatch(g, a, 1_000_000, 1_000_200))
    rc, ref = o.loglike(st[0], 0.01, FREQS)
    assert rc == 0 and ctx.loglike(st, 0.01, FREQS)[0][0] == pytest.approx(ref, rel=1e-10)


def test_haplocart_ten_million_reads_in_eight_shards():
    """BASELINE configs[2]: 10M x 150 bp as 8 contiguous shards of the seeded read stream, each through a context of its
    own (what 8 ranks do), the eight final vectors summed on the host (what the reduce does) -- against one context that
    takes the same 10M reads in ten batches, and against the oracle on scattered samples of the stream."""
    seed, total, world = 0x76676131, 10_000_000, 8
    g = hc.synth_graph(seed=seed)
    from vgan_amd import distributed as vd
    sharded = np.zeros(g.n_paths)
    n_sharded = 0
    for rank in range(world):
        r0, r1 = vd.shard_bounds(total, rank, world)
        ctx = hc.HcContext(g)
        a = hc.synth_reads(g, r1 - r0, seed=seed, read_len=150, first_read=r0)
        b = hc.HostBatch(g, a)
        n_sharded += b.n_reads
        ctx.accumulate(b)
        sharded += ctx.finalize()
        del ctx, a, b
    one = hc.HcContext(g)
    n_one = 0
    for c0 in range(0, total, 1_000_000):
        a = hc.synth_reads(g, 1_000_000, seed=seed, read_len=150, first_read=c0)
        b = hc.HostBatch(g, a)
        n_one += b.n_reads
        one.accumulate(b)
        del a, b
    whole = one.finalize()
    assert n_one == n_sharded > 9_990_000
    assert np.all(np.isfinite(whole)) and whole.max() < 0 and util.rel_err(sharded, whole) < 1e-11
    # the stream against the oracle, wherever a shard boundary or a batch boundary falls
    og = util.orc_graph_from_product(g)
    ctx = hc.HcContext(g)
    for r0 in (0, 1_249_980, 4_999_990, 9_999_950):
        a = hc.synth_reads(g, 40, seed=seed, read_len=150, first_read=r0)
        _, want, _ = orc.hc_run(og, util.orc_alnset_from_product(a), n_threads=8, faithful=False)
        ctx.reset()
        ctx.accumulate(hc.HostBatch(g, a))
        assert util.rel_err(ctx.finalize(), want) < 1e-9


def test_euka_five_million_reads_in_eight_shards():
    """BASELINE configs[3]: 5M synthetic 75 bp aDNA reads with the dhigh damage profiles as 8 shards of 625k, each in a context
    of its own: per-clade counts and base-shift tables add exactly, coverage and likelihood sums to rounding, against one
    context taking the eight shards in turn; one shard's head against the oracle."""
    d = os.path.join(GOLD, "damageProfiles")
    texts = (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read())
    dm = ek.Damage.from_text(*texts)
    world, per = 8, 625_000
    one = None
    acc = None
    n_reads = 0
    for rank in range(world):
        g, db, a = ek.synth_euka(per, dm, read_seed=1000003 * rank)
        if one is None:
            one = ek.EukaContext(db, dm)
        hb = ek.EukaHostBatch(g, a)
        n_reads += hb.n_reads
        ctx = ek.EukaContext(db, dm)
        got = ctx.accumulate(hb)
        fin = ctx.finalize()
        n, s = ctx.like_sums()
        again = one.accumulate(hb)
        assert np.array_equal(again["like"], got["like"]) and np.array_equal(again["clade"], got["clade"])  # per read: the same bits
        part = {"clade_count": fin["clade_count"].astype(np.int64), "baseshift": fin["baseshift"].astype(np.int64),
                "bin_cov": fin["bin_cov"].astype(np.float64), "n": n.astype(np.int64), "s": np.where(np.isfinite(s), s, 0.0),
                "inf": ~np.isfinite(s) & (n > 0), "passed": int(got["pass"].sum())}
        acc = part if acc is None else {k: acc[k] + part[k] for k in part}
        if rank == 3:  # the head of one shard against the oracle
            sub = _range(a, 0, 300)
            ref = orc.euka_run(util.orc_graph_nodes_only(g), util.orc_alnset_from_product(sub), util.orc_euka_db_from_product(db),
                               orc.OrcDamage(*texts), 29, 5)
            src = hb.arrays()["read_src"]
            idx = np.nonzero(src < 300)[0]
            assert np.array_equal(got["clade"][idx], ref["clade"][src[idx]]) and np.array_equal(got["pass"][idx], ref["pass"][src[idx]])
            ok = got["clade"][idx] >= 0
            assert ok.sum() > 250 and util.rel_err(got["like"][idx][ok], ref["like"][src[idx]][ok]) < 1e-10
        del ctx, g, a, hb
    fin = one.finalize()
    n, s = one.like_sums()
    assert n_reads > 4_900_000 and acc["passed"] == fin["clade_count"].sum() > 1_500_000
    assert np.array_equal(acc["clade_count"], fin["clade_count"]) and np.array_equal(acc["baseshift"], fin["baseshift"])
    assert np.array_equal(acc["n"], n) and np.allclose(acc["bin_cov"], fin["bin_cov"], rtol=1e-12, atol=1e-9)
    okc = np.isfinite(s)
    assert np.array_equal(~okc & (n > 0), acc["inf"] > 0) and util.rel_err(acc["s"][okc], s[okc]) < 1e-11


def test_empty_inputs_on_every_path(tmp_path):
    """No reads at all: every entry point returns zeros / empty results instead of failing (an empty GAM is what a sample
    without a single mapped fragment produces)."""
    d = os.path.join(GOLD, "damageProfiles")
    dm = ek.Damage.from_text(open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read())
    g, db, a = ek.synth_euka(50, dm, n_clades=6, nodes_per_clade=120)
    ectx = ek.EukaContext(db, dm)
    got = ectx.accumulate(ek.EukaHostBatch(g, a, 0, 0))
    fin = ectx.finalize()
    n, s = ectx.like_sums()
    assert len(got["clade"]) == 0 and not fin["clade_count"].any() and not fin["baseshift"].any() and not n.any() and not s.any()
    det, est = ek.report(db, fin, n, s, got["clade"], got["pass"], np.zeros(0, np.uint16), str(tmp_path / "none"), min_bins=1, entropy=0.0)
    assert len(det) == 0 and b"yes" not in open(str(tmp_path / "none_abundance.tsv"), "rb").read()
    g2 = hc.synth_graph(seed=3, genome_len=900, n_nodes=600, n_paths=12)
    a2 = hc.synth_reads(g2, 40, seed=1, read_len=60)
    sctx = sb.SbContext(g2, ek.Damage.from_text("", ""))
    sctx.precompute(sb.SbHostBatch(g2, a2, 0, 0))
    ll, guard = sctx.loglike([[(1, 0, 0.01, 0.5, 1.0)]], 0.01, FREQS)
    best, sig, n_ok = sctx.best_paths()
    assert ll[0] == 0.0 and guard[0] == 0 and sctx.refresh([(1, 0, 0.01, 0.5, 1.0)], 0.01, FREQS) == (0.0, 0)
    assert len(best) == 0 and not sig.any() and n_ok == 0 and sctx.mixture_loglike([0, 1], -0.69) == 0.0
    assert list(sb.signature_paths(sig, n_ok)) == []
    hctx = hc.HcContext(g2)
    hctx.accumulate(hc.HostBatch(g2, a2, 0, 0))
    assert not hctx.finalize().any()


def test_clis_on_a_gam_without_reads(tmp_path):
    import shutil
    import subprocess
    exe = os.path.join(os.path.dirname(GOLD), "..", "vgan_amd", "bin", "vgan")
    g, db, a = ek.synth_euka(20, None, n_clades=5, nodes_per_clade=120)
    util.write_euka_db(db, g, tmp_path)
    gam = str(tmp_path / "none.gam")
    a.without(np.ones(a.n_reads, np.uint8)).write_gam(gam)
    r = subprocess.run([exe, "euka", "-g", gam, "--euka_dir", str(tmp_path), "-o", str(tmp_path / "e")], capture_output=True, text=True)
    assert r.returncode == 0 and "Number of fragments in input file: 0" in r.stderr, r.stderr[-800:]
    assert open(str(tmp_path / "e_abundance.tsv")).read().count("\tno\t0\t0") == 5
    hcdir = tmp_path / "hc"
    hcdir.mkdir()
    g2 = hc.synth_graph(seed=3, genome_len=900, n_nodes=600, n_paths=12)
    g2.write(str(hcdir))
    r = subprocess.run([exe, "haplocart", "-g", gam, "--hc-files", str(hcdir), "-o", str(tmp_path / "h.tsv"), "-np", "-q"], capture_output=True, text=True)
    # HaploCart.cpp:384-385: "[HaploCart] Error, no reads mapped" and a failing exit, no result line
    assert r.returncode != 0 and "[HaploCart] Error, no reads mapped" in r.stderr and not os.path.exists(str(tmp_path / "h.tsv")), r.stderr[-800:]
    shutil.copy(str(hcdir / "graph.gfa"), str(tmp_path / "T.gfa"))
    from test_sb_chain_cpu import _newick_of
    (tmp_path / "tree_dir").mkdir()
    (tmp_path / "tree_dir" / "T.new.dnd").write_text(_newick_of(g2))
    (tmp_path / "soibean_db.baseFreq").write_text("T .3 .2 .2 .3\n")
    r = subprocess.run([exe, "soibean", "-g", gam, "--soibean_dir", str(tmp_path), "--dbprefix", "T", "-o", str(tmp_path / "s_")], capture_output=True, text=True)
    assert r.returncode == 1 and "no usable read" in r.stderr
