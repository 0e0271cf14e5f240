"""tools/pyref_hc.py -- the second, independent restatement of the HaploCart path (Python + mpmath, written from the reference's
sources) -- and its committed fixture tests/golden/hc_pyref/: the restatement reproduces the reference's own reconstruction
KATs, and the C++ oracle (the first restatement, long double) agrees with the fixture on the same files.  Two restatements in
two languages and two arithmetics: what the GPU path is held against no longer rests on one of them."""
import json
import os
import sys

import numpy as np
import pytest

import gamio
import orc
import util

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# simulated reads; the reference's four bundled alignment files (test/input_files/*.gam) on graphs covering their node ids
FIX_NAMES = ["hc_pyref", "hc_pyref_j2", "hc_pyref_two_unique", "hc_pyref_all_the_same", "hc_pyref_all_the_same_reverse"]
FIXES = [os.path.join(HERE, "golden", n) for n in FIX_NAMES]


@pytest.fixture(params=FIXES, ids=["simulated", "J2a1a1a1", "two_unique", "all_the_same", "all_the_same_reverse"])
def FIX(request):
    return request.param


def test_the_restatements_take_nothing_from_the_product():
    """tools/pyref_*.py build their graphs and reads themselves (tools/pyref_inputs.py: plain seeded Python): neither the
    arithmetic nor the input distribution of a fixture is the product's."""
    for f in ("pyref_hc.py", "pyref_euka.py", "pyref_sb.py", "pyref_inputs.py"):
        txt = open(os.path.join(ROOT, "tools", f)).read()
        assert "vgan_amd" not in txt and "import orc" not in txt and "liboracle" not in txt, f



def _pyref():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pyref_hc
    return pyref_hc


def test_pyref_reproduces_the_reference_reconstruction_kats():
    _pyref().check_kats()  # src/test.cpp:855-994 (20 strings) + the derived per-edit sizes


def test_pyref_recomputes_its_committed_fixture_on_a_sample(tmp_path, FIX):
    """The committed numbers are what the script computes from the committed inputs (first reads only: mpmath is slow)."""
    p = _pyref()
    fix = json.load(open(os.path.join(FIX, "hc_pyref.json")))
    seqs = p.load_gfa(os.path.join(FIX, "graph.gfa"))
    hcf = p.load_hcfiles(FIX)
    alns = gamio.read_gam(os.path.join(FIX, "reads.gam"))
    for rec in fix["default"]["first_reads"][:4]:
        ll = p.read_loglik(seqs, hcf, alns[rec["read"]], 0.0001, False, False)
        assert [p.mp.nstr(x, 25) for x in ll] == rec["loglik"]


def test_the_cpp_oracle_agrees_with_the_python_restatement(FIX):
    fix = json.load(open(os.path.join(FIX, "hc_pyref.json")))
    og, names, parents, children = util.orc_graph_from_hcfiles(FIX)
    dicts = gamio.read_gam(os.path.join(FIX, "reads.gam"))
    for key, kw in (("default", {}), ("background", dict(params=orc.hc_params(background_error_prob=0.02, use_background_error_prob=True)))):
        want = fix[key]
        undefined = {u["read"] for u in want["undefined_reads"]}
        # the oracle DEFINES the reads the restatement refuses (out-of-range qualities are clamped, oracle.h): left out here
        oa = orc.AlnSet([d for r, d in enumerate(dicts) if r not in undefined])
        ref, _, n_bad = orc.hc_run(og, oa, n_threads=4, faithful=True, **kw)
        assert n_bad == 0
        fv = np.array([float(x) for x in want["final_vec"]])
        assert util.rel_err(np.asarray(ref, np.float64), fv) < 1e-13, key
        pred = names[int(np.argmax(np.asarray(ref, np.float64)))]
        assert pred == want["predicted"]
        post = orc.hc_posterior(np.asarray(ref, np.longdouble), names, parents, children, pred)
        assert [c for c, _, _ in post] == [x["clade"] for x in want["posterior"]]
        for (_, c1, _), x in zip(post, want["posterior"]):
            assert abs(c1 - float(x["confidence"])) <= 1e-12 * max(abs(float(x["confidence"])), 1e-300)
        # per-read vectors of the first reads
        kept = [r for r in range(len(dicts)) if r not in undefined]
        for rec in want["first_reads"]:
            rc, vec, _ = orc.hc_read(og, oa, kept.index(rec["read"]), **kw)
            assert rc == 0 and util.rel_err(vec.astype(np.float64), np.array([float(x) for x in rec["loglik"]])) < 1e-13


# ---- the fixture at the reference's shape: 5 179 paths (81 mask words), 11 820 nodes, 1 200 reads of ~150 bases
FULL = os.path.join(HERE, "golden", "hc_pyref_full")


def test_full_shape_fixture_is_what_the_script_computes_on_a_sample():
    """tools/pyref_hc.py --make-full: per-mapping sums in mpmath, the sum over a read's mappings through the path_supports rows in
    numpy long double, read_loglik's literal loop over the paths beside it on 50 reads (asserted when the fixture is made).  Here:
    the sampled entries of the first reads' vectors are recomputed from the committed inputs."""
    p = _pyref()
    fix = json.load(open(os.path.join(FULL, "hc_pyref.json")))
    assert fix["default"]["literal_reads_checked"] == 200 and len(fix["default"]["first_reads"]) == 50 and len(fix["default"]["final_vec"]) == 5179
    seqs = p.load_gfa(os.path.join(FULL, "graph.gfa"))
    hcf = p.load_hcfiles(FULL, supports_as_numpy=True, supports_lists=False)
    alns = gamio.read_gam(os.path.join(FULL, "reads.gam"))
    sup = hcf["supports_np"]
    for rec in fix["default"]["first_reads"][:3]:
        segs = p.segment_sums(seqs, hcf, alns[rec["read"]], 0.0001, False, False)
        for path, want in list(zip(rec["paths"], rec["loglik"]))[:16]:
            ll = sum((m_ if sup[node, path] else u_) for node, m_, u_ in segs)
            assert abs(ll - p.mp.mpf(want)) <= abs(p.mp.mpf(want)) * p.mp.mpf("1e-22"), (rec["read"], path)


def test_the_cpp_oracle_agrees_with_the_python_restatement_at_the_reference_shape():
    fix = json.load(open(os.path.join(FULL, "hc_pyref.json")))
    og, names, parents, children = util.orc_graph_from_hcfiles(FULL)
    assert len(names) == 5179
    dicts = gamio.read_gam(os.path.join(FULL, "reads.gam"))
    for key, kw in (("default", {}), ("background", dict(params=orc.hc_params(background_error_prob=0.02, use_background_error_prob=True)))):
        want = fix[key]
        undefined = {u["read"] for u in want["undefined_reads"]}
        oa = orc.AlnSet([d for r, d in enumerate(dicts) if r not in undefined])
        # (the hoisted entry: S_m / U_m once per mapping -- bit-identical sums to the literal loops, which take minutes at 5 179 paths;
        # the literal loops run on the first reads below)
        _, ref, n_bad = orc.hc_run(og, oa, n_threads=8, faithful=False, **kw)
        assert n_bad == 0
        fv = np.array([float(x) for x in want["final_vec"]])
        assert util.rel_err(np.asarray(ref, np.float64), fv) < 1e-13, key
        pred = names[int(np.argmax(np.asarray(ref, np.float64)))]
        assert pred == want["predicted"]
        post = orc.hc_posterior(np.asarray(ref, np.longdouble), names, parents, children, pred)
        assert [c for c, _, _ in post] == [x["clade"] for x in want["posterior"]]
        for (_, c1, _), x in zip(post, want["posterior"]):
            assert abs(c1 - float(x["confidence"])) <= 1e-12 * max(abs(float(x["confidence"])), 1e-300)
        kept = [r for r in range(len(dicts)) if r not in undefined]
        for rec in want["first_reads"][:4]:  # the literal per-path loops of the oracle, read by read
            rc, vec, _ = orc.hc_read(og, oa, kept.index(rec["read"]), **kw)
            got = vec.astype(np.float64)[rec["paths"]]
            assert rc == 0 and util.rel_err(got, np.array([float(x) for x in rec["loglik"]])) < 1e-13


# ------------------------------------------------------------------------------------------------------------------- euka
EFIX = os.path.join(HERE, "golden", "euka_pyref")


EFIX_FULL = os.path.join(HERE, "golden", "euka_pyref_full")  # the SHIPPED 335-clade tables (tools/pyref_euka.py --make-full)


def _euka_inputs(d=None):
    from vgan_amd import euka as ek
    from vgan_amd import haplocart as hc
    d = d or EFIX
    g = hc.Graph.load(os.path.join(d, "graph.gfa"))
    db = ek.EukaDb.load(os.path.join(d, "euka_db.clade"), os.path.join(d, "euka_db.bins"))
    a = hc.AlnSet.read_gam(os.path.join(d, "reads.gam"), keep_unmapped=True)
    texts = (open(os.path.join(d, "damage5p.prof")).read(), open(os.path.join(d, "damage3p.prof")).read())
    return g, db, a, texts


def check_euka_against_fixture(got, fin, want, read_index, tol):
    """got: per-read arrays indexed by k, read_index[k] = alignment number; fin: the sums; want: the fixture of one run."""
    recs = {x["read"]: x for x in want["reads"]}
    assert not want["undefined_reads"] and len(recs) == len(read_index)
    for k, r in enumerate(read_index):
        x = recs[int(r)]
        assert int(got["clade"][k]) == x["clade"] and bool(got["pass"][k]) == x["pass"], r
        for key in ("in_lik", "out_lik", "like"):
            ref = float(x[key])
            assert abs(got[key][k] - ref) <= tol * max(abs(ref), 1e-300), (r, key)
        assert abs(got["not_like"][k] - float(x["not_like"])) <= 1e-12
    assert list(fin["clade_count"]) == want["clade_count"]
    assert np.array_equal(np.asarray(fin["baseshift"]), np.array(want["baseshift"], np.uint32))
    flat = np.array([float(v) for b in want["bin_cov"] for v in b])
    assert np.allclose(np.asarray(fin["bin_cov"]), flat, rtol=1e-12, atol=1e-12)


def test_euka_restatement_recomputes_its_fixture_on_a_sample():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pyref_euka as pe
    fix = json.load(open(os.path.join(EFIX, "euka_pyref.json")))["default"]
    seqs = pe.load_gfa(os.path.join(EFIX, "graph.gfa"))
    clades, chunks = pe.load_clades(os.path.join(EFIX, "euka_db.clade")), pe.load_bins(os.path.join(EFIX, "euka_db.bins"))
    dmg = pe.DamageModel(open(os.path.join(EFIX, "damage5p.prof")).read(), open(os.path.join(EFIX, "damage3p.prof")).read())
    alns = gamio.read_gam(os.path.join(EFIX, "reads.gam"))
    for rec in fix["reads"][:6]:
        a = alns[rec["read"]]
        c_n = pe.clade_of(chunks, a["path"]["mapping"][0]["position"]["node_id"])
        gs, rs, _ = pe.reconstruct_graph_sequence(seqs, a["path"])
        i, o = pe.read_models(a, gs, rs, clades[c_n]["dist"], dmg)
        assert c_n == rec["clade"] and pe.mp.nstr(i, 25) == rec["in_lik"] and pe.mp.nstr(o, 25) == rec["out_lik"]


def test_the_cpp_euka_oracle_agrees_with_the_python_restatement():
    fix = json.load(open(os.path.join(EFIX, "euka_pyref.json")))
    g, db, a, texts = _euka_inputs()
    og = util.orc_graph_nodes_only(g)
    oa = util.orc_alnset_from_product(a)
    for key, (mq, ltp) in (("default", (29, 5)), ("other_thresholds", (0, 3))):
        ref = orc.euka_run(og, oa, util.orc_euka_db_from_product(db), orc.OrcDamage(*texts), mq, ltp)
        assert ref["n_bad"] == 0
        idx = [x["read"] for x in fix[key]["reads"]]
        got = {k: ref[k][idx] for k in ("clade", "in_lik", "out_lik", "like", "not_like", "pass")}
        check_euka_against_fixture(got, ref, fix[key], idx, 1e-13)


def test_the_cpp_euka_oracle_agrees_with_the_python_restatement_on_the_shipped_tables():
    """335 clades, the shipped euka_db.clade / euka_db.bins (node ids up to 6.9 million, written as "1836.0": std::stoi reads the
    integer prefix, load.cpp:70-95)."""
    fix = json.load(open(os.path.join(EFIX_FULL, "euka_pyref.json")))["default"]
    g, db, a, texts = _euka_inputs(EFIX_FULL)
    assert len(fix["clade_count"]) == 335
    ref = orc.euka_run(util.orc_graph_nodes_only(g), util.orc_alnset_from_product(a), util.orc_euka_db_from_product(db), orc.OrcDamage(*texts), 29, 5)
    assert ref["n_bad"] == 0
    idx = [x["read"] for x in fix["reads"]]
    got = {k: ref[k][idx] for k in ("clade", "in_lik", "out_lik", "like", "not_like", "pass")}
    # (`like` is exp(a difference of two log-likelihoods of ~ -50): the doubles' 1e-16 on those is 1e-13 on it)
    check_euka_against_fixture(got, ref, fix, idx, 1e-12)


# ---------------------------------------------------------------------------------------------------------------- soibean
SFIX = os.path.join(HERE, "golden", "sb_pyref")
SFIX_FULL = os.path.join(HERE, "golden", "sb_pyref_full")  # the shape of the reference's own soibean test (test.cpp:243-248: Ursidae, 28 paths)
SFIXES = [SFIX, SFIX_FULL]


def _sb_inputs(SFIX=SFIX):
    from vgan_amd import haplocart as hc
    g = hc.Graph.load(os.path.join(SFIX, "graph.gfa"), SFIX)
    a = hc.AlnSet.read_gam(os.path.join(SFIX, "reads.gam"), keep_unmapped=True)
    texts = (open(os.path.join(SFIX, "damage5p.prof")).read(), open(os.path.join(SFIX, "damage3p.prof")).read())
    return g, a, texts


@pytest.mark.parametrize("SFIX", SFIXES, ids=["sb_pyref", "sb_pyref_full"])
def test_soibean_restatement_recomputes_its_fixture_on_a_sample(SFIX):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pyref_sb as ps
    fix = json.load(open(os.path.join(SFIX, "sb_pyref.json")))["default"]
    seqs, names, node_paths, dmg = ps.load_inputs(SFIX)
    alns = gamio.read_gam(os.path.join(SFIX, "reads.gam"))
    for rec in fix["reads"][:3]:
        pm, _ = ps.analyse_read(seqs, node_paths, names, alns[rec["read"]], dmg, 7)
        assert [ps.mp.nstr(x, 25) for x in pm] == rec["pm"]


@pytest.mark.parametrize("SFIX", SFIXES, ids=["sb_pyref", "sb_pyref_full"])
def test_the_cpp_soibean_oracle_agrees_with_the_python_restatement(SFIX):
    fix = json.load(open(os.path.join(SFIX, "sb_pyref.json")))["default"]
    g, a, texts = _sb_inputs(SFIX)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    findable = np.array([len(n) <= 101 for n in g.path_names], np.uint8)
    o = orc.SbOracle(og, oa, orc.OrcDamage(*texts), penalty=fix["params"]["penalty"], path_findable=findable)
    assert o.n_bad == 0 and not fix["undefined_reads"]
    for rec in fix["reads"]:
        r = rec["read"]
        assert o.ok(r)
        # (1e-13 of values as small as -5e-6: sums of a dozen logs of numbers next to 1, where the oracle's long double rounds at 1e-19)
        assert util.rel_err(o.pathmap(r), np.array([float(x) for x in rec["pm"]])) < 3e-13, r
        for p in range(g.n_paths):
            c, _ = o.counts(r, p)
            assert list(c) == rec["cnt"][p], (r, p)
    assert sum(o.ok(r) for r in range(a.n_reads)) == len(fix["reads"])
    for st in fix["states"]:
        rc, ll = o.loglike([tuple(s) for s in st["sources"]], st["con"], fix["params"]["freqs"])
        assert rc == 0 and abs(ll - float(st["loglike"])) <= 1e-12 * abs(float(st["loglike"]))


def test_product_reconstruction_equals_the_python_restatement_on_thousands_of_reads(tmp_path):
    """a1 of every path (vgan_utils.h:6-79) through the product's own GAM reader and `reconstruct` (csrc/host/flatten.cpp: the
    one-walk form for match / substitution reads and the general two-walk form) against tools/pyref_hc.py reading the same file
    with the test-side decoder: graph_seq, the aligned read string and the per-edit sizes, read by read -- and the reads the
    restatement refuses as undefined in the reference are the ones the product reports as such."""
    from vgan_amd import _native as N
    from vgan_amd import haplocart as hc
    p = _pyref()
    g = hc.synth_graph(seed=123, genome_len=2500, n_nodes=1700, n_paths=40)
    a = hc.synth_reads(g, 4000, seed=124, read_len=90, indel_rate=0.3, softclip_rate=0.2)
    f = str(tmp_path / "r.gam")
    a.write_gam(f)
    b = hc.AlnSet.read_gam(f, keep_unmapped=True)
    dicts = gamio.read_gam(f)
    assert b.n_reads == len(dicts) == 4000
    off, seq = g.node_seq_off, g.node_seq
    seqs = {i: bytes(seq[off[i]:off[i + 1]]).decode() for i in range(g.min_id, g.max_id + 1) if off[i + 1] > off[i]}
    n_general = n_undefined = 0
    for r, d in enumerate(dicts):
        try:
            gs, rs, sizes = p.reconstruct_graph_sequence(seqs, d["path"])
        except (p.Undefined, KeyError):
            n_undefined += 1
            with pytest.raises(N.NativeError):
                hc.reconstruct(g, b, r)
            continue
        got_g, got_r, got_s = hc.reconstruct(g, b, r)
        assert got_g == gs.encode() and got_r == rs.encode() and list(got_s) == sizes, r
        n_general += any(e["from_length"] != e["to_length"] for m in d["path"]["mapping"] for e in m["edit"])
    assert n_general > 800 and 4000 - n_general - n_undefined > 1500  # both walks of the product were exercised
