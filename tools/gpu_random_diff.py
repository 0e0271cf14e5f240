#!/usr/bin/env python3
"""Randomised differential run (developer aid, GPU box): HaploCart / euka device results against the oracle over many
random graph shapes, read lengths, edit rates and parameter settings.  Prints the worst relative error per path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
import util  # noqa: E402
from vgan_amd import euka as ek  # noqa: E402
from vgan_amd import haplocart as hc  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst_hc = 0.0
for case in range(n_cases):
    L = int(rng.choice([400, 900, 2500, 6000]))
    n_nodes = int(L * rng.uniform(0.05, 0.75)) + 8
    P = int(rng.choice([1, 3, 64, 65, 200, 700, 5179, 8192, 12000]))
    g = hc.synth_graph(seed=int(rng.integers(1 << 30)), genome_len=L, n_nodes=n_nodes, n_paths=P)
    rl = int(rng.choice([30, 60, 100, 150, 250, 400, 1000, 1500]))
    rl = min(rl, L - 10)
    a = hc.synth_reads(g, int(rng.integers(50, 1500)), seed=int(rng.integers(1 << 30)), read_len=rl,
                       indel_rate=float(rng.choice([0, 0.05, 0.5])), softclip_rate=float(rng.choice([0, 0.05, 0.5])),
                       low_mapq_rate=float(rng.choice([0, 0.1, 0.9])))
    kw = [dict(), dict(background_error_prob=0.02, use_background_error_prob=True),
          dict(background_error_prob=0.3, use_background_error_prob=True, is_consensus_fasta=True)][int(rng.integers(3))]
    b = hc.HostBatch(g, a, n_threads=int(rng.integers(1, 5)))
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    p = orc.hc_params(kw.get("background_error_prob", 0.0001), kw.get("use_background_error_prob", False),
                      kw.get("is_consensus_fasta", False))
    _, ref, bad = orc.hc_run(og, oa, p, n_threads=8, faithful=False)
    assert bad == b.stats.n_bad
    ctx = hc.HcContext(g, **kw)
    for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ, hc.MODE_PER_READ_DENSE):
        ctx.reset()
        ctx.set_mode(mode)
        ctx.accumulate(b)
        got = ctx.finalize()
        e = util.rel_err(got, ref)
        worst_hc = max(worst_hc, e)
        assert e < 1e-9, (case, mode, e)
    print("hc case %2d: L=%d nodes=%d P=%d read_len=%d reads=%d tileable=%d kw=%s ok" % (case, L, n_nodes, P, rl, b.n_reads, b.n_tileable, list(kw)), flush=True)
print("haplocart worst rel err %.3g over %d cases" % (worst_hc, n_cases))

gold = os.path.join(ROOT, "tests", "golden", "damageProfiles")
texts = (open(gold + "/dhigh5p.prof").read(), open(gold + "/dhigh3p.prof").read())
worst_ek = 0.0
for case in range(max(4, n_cases // 3)):
    use = texts if case % 2 == 0 else ("", "")
    dm = ek.Damage.from_text(*use)
    g, db, a = ek.synth_euka(int(rng.integers(200, 4000)), dm, seed=int(rng.integers(1 << 30)), n_clades=int(rng.choice([1, 3, 40, 335])),
                             nodes_per_clade=int(rng.choice([60, 200, 400])), read_len_mean=int(rng.choice([40, 75, 120])))
    mm, ltp = int(rng.choice([0, 29, 50])), int(rng.choice([0, 3, 5, 12]))
    hb = ek.EukaHostBatch(g, a)
    ctx = ek.EukaContext(db, dm, min_mapq=mm, length_to_prof=ltp)
    got = ctx.accumulate(hb)
    fin = ctx.finalize()
    ref = orc.euka_run(util.orc_graph_nodes_only(g), util.orc_alnset_from_product(a), util.orc_euka_db_from_product(db),
                       orc.OrcDamage(*use), mm, ltp)
    src = hb.arrays()["read_src"]
    ok = got["clade"] >= 0
    assert np.array_equal(got["clade"], ref["clade"][src]) and np.array_equal(got["pass"], ref["pass"][src])
    for k in ("in_lik", "out_lik", "like"):
        e = util.rel_err(got[k][ok], ref[k][src][ok])
        worst_ek = max(worst_ek, e)
        assert e < 1e-9, (case, k, e)
    assert np.array_equal(fin["clade_count"], ref["clade_count"])
    if ltp > 0:
        assert np.array_equal(fin["baseshift"], ref["baseshift"])
    assert np.allclose(fin["bin_cov"], ref["bin_cov"], rtol=1e-12, atol=1e-12)
    print("euka case %2d ok (%d reads, min_mapq %d, l %d)" % (case, hb.n_reads, mm, ltp), flush=True)
print("euka worst rel err %.3g" % worst_ek)
