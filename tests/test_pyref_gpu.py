"""The HIP path against the committed fixture of the independent Python + mpmath restatement (tools/pyref_hc.py,
tests/golden/hc_pyref/): no oracle/ in this file -- liboracle.so is neither loaded nor needed."""
import json
import os

import numpy as np
import pytest

from vgan_amd import haplocart as hc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(params=["hc_pyref", "hc_pyref_j2", "hc_pyref_two_unique", "hc_pyref_all_the_same", "hc_pyref_all_the_same_reverse"],
                ids=["simulated", "J2a1a1a1", "two_unique", "all_the_same", "all_the_same_reverse"])
def FIX(request):
    return os.path.join(GOLD, request.param)


@pytest.mark.parametrize("key, kw", [("default", {}), ("background", dict(background_error_prob=0.02, use_background_error_prob=True))])
def test_final_vector_per_read_vectors_and_posteriors_against_the_python_restatement(key, kw, FIX):
    want = json.load(open(os.path.join(FIX, "hc_pyref.json")))[key]
    g = hc.Graph.load(os.path.join(FIX, "graph.gfa"), FIX)
    a = hc.AlnSet.read_gam(os.path.join(FIX, "reads.gam"))
    assert a.n_reads == want["n_alignments"]
    # the reads the restatement refuses as undefined in the reference (the product defines them: include/vgan_gpu.h) are left out
    drop = np.zeros(a.n_reads, np.uint8)
    for u in want["undefined_reads"]:
        drop[u["read"]] = 1
    a = a.without(drop)
    b = hc.HostBatch(g, a)
    assert b.n_reads == want["n_used"] and b.stats.n_bad == 0
    fv = np.array([float(x) for x in want["final_vec"]])
    ctx = hc.HcContext(g, **kw)
    for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ, hc.MODE_PER_READ_DENSE):
        ctx.reset()
        ctx.set_mode(mode)
        ctx.accumulate(b)
        got = ctx.finalize()
        assert np.max(np.abs(got - fv) / np.abs(fv)) < 1e-9, (key, mode)
    assert g.path_names[ctx.argmax(got)] == want["predicted"]
    post = ctx.posterior(got, want["predicted"])
    assert [c for c, _, _ in post] == [x["clade"] for x in want["posterior"]]
    for (_, c1, _), x in zip(post, want["posterior"]):
        assert c1 == pytest.approx(float(x["confidence"]), rel=1e-9, abs=1e-300)
    # per-read vectors (the value Haplocart::update returns) of the first reads
    kept = [r for r in range(want["n_alignments"]) if not drop[r]]
    src = list(b.read_src)
    ll = ctx.read_loglik(b)
    for rec in want["first_reads"]:
        k = src.index(kept.index(rec["read"]))
        ref = np.array([float(x) for x in rec["loglik"]])
        assert np.max(np.abs(ll[k] - ref) / np.abs(ref)) < 1e-11, rec["read"]


@pytest.mark.parametrize("key, kw", [("default", {}), ("background", dict(background_error_prob=0.02, use_background_error_prob=True))])
def test_the_reference_shape_against_the_python_restatement(key, kw):
    """tests/golden/hc_pyref_full (tools/pyref_hc.py --make-full): 5 179 paths / 81 mask words, 11 820 nodes, 1 200 reads of ~150
    bases -- the shape BASELINE.json's configs are quoted on -- through every mode, the packed route included."""
    FULL = os.path.join(GOLD, "hc_pyref_full")
    want = json.load(open(os.path.join(FULL, "hc_pyref.json")))[key]
    g = hc.Graph.load(os.path.join(FULL, "graph.gfa"), FULL)
    assert g.n_paths == 5179
    a = hc.AlnSet.read_gam(os.path.join(FULL, "reads.gam"))
    drop = np.zeros(a.n_reads, np.uint8)
    for u in want["undefined_reads"]:
        drop[u["read"]] = 1
    a = a.without(drop)
    fv = np.array([float(x) for x in want["final_vec"]])
    ctx = hc.HcContext(g, **kw)
    for packed in (False, True):
        b = hc.HostBatch(g, a, packed=packed)
        assert b.n_reads == want["n_used"] and b.stats.n_bad == 0
        for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ, hc.MODE_PER_READ_DENSE):
            ctx.reset()
            ctx.set_mode(mode)
            ctx.accumulate(b)
            got = ctx.finalize()
            assert np.max(np.abs(got - fv) / np.abs(fv)) < 1e-9, (key, mode, packed)
    ctx.set_mode(hc.MODE_NODE_WEIGHTS)
    assert g.path_names[ctx.argmax(got)] == want["predicted"]
    post = ctx.posterior(got, want["predicted"])
    assert [c for c, _, _ in post] == [x["clade"] for x in want["posterior"]]
    for (_, c1, _), x in zip(post, want["posterior"]):
        assert c1 == pytest.approx(float(x["confidence"]), rel=1e-9, abs=1e-300)
    kept = [r for r in range(want["n_alignments"]) if not drop[r]]
    b = hc.HostBatch(g, a)
    src = list(b.read_src)
    ll = ctx.read_loglik(b)
    for rec in want["first_reads"]:
        k = src.index(kept.index(rec["read"]))
        ref = np.array([float(x) for x in rec["loglik"]])
        assert np.max(np.abs(ll[k][rec["paths"]] - ref) / np.abs(ref)) < 1e-11, rec["read"]


def test_euka_per_read_models_and_sums_against_the_python_restatement():
    """tools/pyref_euka.py's fixture (tests/golden/euka_pyref/): clade, model 1 / model 2 log-likelihoods, clade_like, the
    detection rule per read; counts, base-shift table and bin coverage per clade."""
    from test_pyref_cpu import _euka_inputs, check_euka_against_fixture, EFIX
    from vgan_amd import euka as ek
    fix = json.load(open(os.path.join(EFIX, "euka_pyref.json")))
    g, db, a, texts = _euka_inputs()
    for key, (mq, ltp) in (("default", (29, 5)), ("other_thresholds", (0, 3))):
        hb = ek.EukaHostBatch(g, a)
        assert hb.stats.n_bad == 0
        ctx = ek.EukaContext(db, ek.Damage.from_text(*texts), min_mapq=mq, length_to_prof=ltp)
        got = ctx.accumulate(hb)
        fin = ctx.finalize()
        assert fin["n_bad"] == 0
        check_euka_against_fixture(got, fin, fix[key], list(hb.arrays()["read_src"]), 1e-9)


def test_euka_on_the_shipped_335_clade_tables_against_the_python_restatement():
    """tests/golden/euka_pyref_full: the shipped euka_db.clade / euka_db.bins, 631 reads over 335 clades' node-id ranges."""
    from test_pyref_cpu import _euka_inputs, check_euka_against_fixture, EFIX_FULL
    from vgan_amd import euka as ek
    fix = json.load(open(os.path.join(EFIX_FULL, "euka_pyref.json")))["default"]
    g, db, a, texts = _euka_inputs(EFIX_FULL)
    hb = ek.EukaHostBatch(g, a)
    assert hb.stats.n_bad == 0
    ctx = ek.EukaContext(db, ek.Damage.from_text(*texts), min_mapq=29, length_to_prof=5)
    got = ctx.accumulate(hb)
    fin = ctx.finalize()
    assert fin["n_bad"] == 0 and len(fin["clade_count"]) == 335
    check_euka_against_fixture(got, fin, fix, list(hb.arrays()["read_src"]), 1e-9)


@pytest.mark.parametrize("which", ["sb_pyref", "sb_pyref_full"])
def test_soibean_tables_and_state_likelihoods_against_the_python_restatement(which):
    """tools/pyref_sb.py's fixtures (tests/golden/sb_pyref/: 12 paths, 120 reads; sb_pyref_full/: the Ursidae tree's 28 paths, 510 reads
    of 50-75 bp, states of one, two and three sources with branch positions 0 and 1 and a branch of length zero): pathMap and the
    (reference, read) pair counts per read and path, and the log-likelihood of the states computed by the restatement from the per-base
    records themselves."""
    from test_pyref_cpu import _sb_inputs, HERE
    from vgan_amd import euka as ek
    from vgan_amd import soibean as sb
    SFIX = os.path.join(HERE, "golden", which)
    fix = json.load(open(os.path.join(SFIX, "sb_pyref.json")))["default"]
    g, a, texts = _sb_inputs(SFIX)
    hb = sb.SbHostBatch(g, a)
    assert hb.stats.n_bad == 0 and hb.n_reads == len(fix["reads"])
    ctx = sb.SbContext(g, ek.Damage.from_text(*texts), penalty=fix["params"]["penalty"])
    assert ctx.precompute(hb) == 0
    pm, cnt, ok = ctx.read_tables()
    assert ok.all()
    recs = {x["read"]: x for x in fix["reads"]}
    for k, r in enumerate(hb.arrays()["read_src"]):
        x = recs[int(r)]
        ref = np.array([float(v) for v in x["pm"]])
        assert np.max(np.abs(pm[:, k] - ref) / np.abs(ref)) < 1e-9, r
        assert np.array_equal(cnt[:, :, k].astype(np.int64), np.array(x["cnt"])), r
    for st in fix["states"]:
        got, guard = ctx.loglike([[tuple(s) for s in st["sources"]]], st["con"], fix["params"]["freqs"])
        assert guard[0] == 0 and got[0] == pytest.approx(float(st["loglike"]), rel=1e-9)
