"""tools/pyref_hc.py -- the second, independent restatement of the HaploCart path (Python + mpmath, written from the reference's
sources) -- and its committed fixture tests/golden/hc_pyref/: the restatement reproduces the reference's own reconstruction
KATs, and the C++ oracle (the first restatement, long double) agrees with the fixture on the same files.  Two restatements in
two languages and two arithmetics: what the GPU path is held against no longer rests on one of them."""
import json
import os
import sys

import numpy as np

import gamio
import orc
import util

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FIX = os.path.join(HERE, "golden", "hc_pyref")


def _pyref():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pyref_hc
    return pyref_hc


def test_pyref_reproduces_the_reference_reconstruction_kats():
    _pyref().check_kats()  # src/test.cpp:855-994 (20 strings) + the derived per-edit sizes


def test_pyref_recomputes_its_committed_fixture_on_a_sample(tmp_path):
    """The committed numbers are what the script computes from the committed inputs (first reads only: mpmath is slow)."""
    p = _pyref()
    fix = json.load(open(os.path.join(FIX, "hc_pyref.json")))
    seqs = p.load_gfa(os.path.join(FIX, "graph.gfa"))
    hcf = p.load_hcfiles(FIX)
    alns = gamio.read_gam(os.path.join(FIX, "reads.gam"))
    for rec in fix["default"]["first_reads"][:4]:
        ll = p.read_loglik(seqs, hcf, alns[rec["read"]], 0.0001, False, False)
        assert [p.mp.nstr(x, 25) for x in ll] == rec["loglik"]


def test_the_cpp_oracle_agrees_with_the_python_restatement():
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
