"""euka downstream of the per-read pass: detected clades, abundance MCMC and Euka::run's output files.
Product host code (closed form per clade, vgan_amd/csrc/host/euka_abundance.cpp) against the oracle's literal restatement of
the reference loops (oracle/euka_abundance_oracle.cpp) on the same per-read results and the same seed: every file byte for
byte.  Runs on CPU: this stage is host code; the per-clade sums the product needs come from the oracle's per-read values here
and from the HIP kernel in tests/test_euka_gpu.py."""
import glob
import os

import numpy as np
import pytest

import orc
import util
from vgan_amd import _native as N
from vgan_amd import euka as ek
from test_euka_cpu import GOLD


def _texts(kind="dhigh"):
    d = os.path.join(GOLD, "damageProfiles")
    return (open(d + "/%s5p.prof" % kind).read(), open(d + "/%s3p.prof" % kind).read())


def _per_read(n_reads, seed, n_clades=12, texts=None, read_seed=0, min_mapq=1):
    """Oracle per-read results on a synthetic sample.  min_mapq = 1 lifts the generator's mapq 0 reads: they have
    clade_like == 0, which makes a clade's likelihood -inf and the chain never accept (covered separately)."""
    texts = texts or _texts()
    dm = ek.Damage.from_text(*texts)
    g, db, a = ek.synth_euka(n_reads, dm, seed=seed, n_clades=n_clades, nodes_per_clade=200, read_seed=read_seed)
    og = util.orc_graph_nodes_only(g)
    arr = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in a.arrays().items()}
    arr["mapq"] = np.maximum(arr["mapq"], min_mapq)
    oa = orc.AlnSet.from_arrays(**arr)
    odb = util.orc_euka_db_from_product(db)
    ref = orc.euka_run(og, oa, odb, orc.OrcDamage(*texts), 29, 5)
    arr = a.arrays()
    seq_len = np.diff(arr["seq_off"])
    names = [bytes(arr["name"][arr["name_off"][i]:arr["name_off"][i + 1]]) for i in range(a.n_reads)]
    return db, odb, ref, seq_len, names


def _sums(ref, n_clades):
    ok = ref["clade"] >= 0
    n = np.bincount(ref["clade"][ok], minlength=n_clades).astype(np.int64)
    with np.errstate(divide="ignore"):
        s = np.bincount(ref["clade"][ok], weights=np.log(ref["like"][ok]), minlength=n_clades)
    return n, s


def _files(prefix):
    return {os.path.basename(p)[len(os.path.basename(prefix)):]: open(p, "rb").read() for p in sorted(glob.glob(prefix + "_*"))}


def _both(tmp_path, tag, db, odb, ref, seq_len, names, **kw):
    n, s = _sums(ref, db.n_clades)
    po, pp = str(tmp_path / (tag + "_orc")), str(tmp_path / (tag + "_prod"))
    det_o, est_o = orc.euka_report(odb, db.clade_id, db.clade_names, ref, seq_len, po, names=names, **kw)
    det_p, est_p = ek.report(db, ref, n, s, ref["clade"], ref["pass"], seq_len, pp, names=names, **kw)
    fo, fp = _files(po), _files(pp)
    assert sorted(fo) == sorted(fp), (sorted(fo), sorted(fp))
    for k in fo:
        assert fo[k] == fp[k], (tag, k, fo[k][:300], fp[k][:300])
    assert np.array_equal(det_o, det_p)
    assert np.allclose(est_o, est_p, rtol=1e-12, atol=0)
    return det_p, est_p, fp


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_outputs_match_the_oracle_byte_for_byte(tmp_path, seed):
    db, odb, ref, seq_len, names = _per_read(4000, seed=5 + seed)
    det, est, files = _both(tmp_path, "mcmc", db, odb, ref, seq_len, names, iters=600, burnin=50, seed=seed, min_bins=1, entropy=0.0)
    assert len(det) >= 2 and set(files) >= {"_abundance.tsv", "_detected.tsv", "_coverage.tsv", "_inSize.tsv", "_5p.prof", "_3p.prof"}
    assert abs(est[:, 0].sum() - 1) < 0.2 and np.all(est[:, 3] <= est[:, 1]) and np.all(est[:, 1] <= est[:, 0] + 1e-12)
    assert np.all(est[:, 0] <= est[:, 2] + 1e-12) and np.all(est[:, 2] <= est[:, 4])
    hdr = files["_abundance.tsv"].split(b"\n")[0]
    assert hdr.endswith(b"95%_confidence_interval_higher_bound") and hdr.count(b"\t") == 7
    # a different seed gives a different chain; the same seed the same files
    det2, est2, files2 = _both(tmp_path, "again", db, odb, ref, seq_len, names, iters=600, burnin=50, seed=seed, min_bins=1, entropy=0.0)
    assert files2["_abundance.tsv"] == files["_abundance.tsv"]
    _, est3, _ = _both(tmp_path, "other", db, odb, ref, seq_len, names, iters=600, burnin=50, seed=seed + 100, min_bins=1, entropy=0.0)
    assert not np.array_equal(est3, est)
    n, s = _sums(ref, db.n_clades)
    assert np.all(np.isfinite(s[det]))  # finite likelihoods: the chain does accept and move
    # with the generator's mapq 0 reads left in, some clades have clade_like == 0 entries: log-likelihood -inf, nothing accepted
    db, odb, ref, seq_len, names = _per_read(4000, seed=5 + seed, min_mapq=0)
    assert np.any(np.isneginf(_sums(ref, db.n_clades)[1]))
    _both(tmp_path, "inf", db, odb, ref, seq_len, names, iters=400, burnin=50, seed=seed, min_bins=1, entropy=0.0)


def test_no_mcmc_out_group_and_fragment_names(tmp_path):
    db, odb, ref, seq_len, names = _per_read(3000, seed=9)
    common = dict(min_bins=1, entropy=0.0)
    det, est, files = _both(tmp_path, "nomcmc", db, odb, ref, seq_len, names, run_mcmc=False, **common)
    assert files["_abundance.tsv"].split(b"\n")[0] == b"#Taxa\tdetected\tNumber_of_reads\tproportion_estimate"
    assert abs(est[:, 0].sum() - 1) < 1e-12 and np.all(est[:, 1:] == 0)
    # coverage rows of the no-MCMC branch end in a tab (Euka.cpp:693-702), those of the MCMC branch do not
    rows = files["_coverage.tsv"].split(b"\n")[1:-1]
    assert rows and all(r.endswith(b"\t") for r in rows)
    # thresholds that reject clades; an out group among the rejected ones still gets its coverage / inSize / .prof
    cnt = ref["clade_count"]
    cut = int(np.sort(cnt[cnt > 0])[len(cnt[cnt > 0]) // 2])
    rejected = [n for n, c in zip(db.clade_names, cnt) if c < cut]
    og = rejected[0]
    det2, _, files2 = _both(tmp_path, "og", db, odb, ref, seq_len, names, min_reads=cut, out_group=og, out_frag=True, iters=300,
                            burnin=20, seed=4, **common)
    assert ("_" + og + ".prof") in files2 and "_FragNames.tsv" in files2
    assert det2[-1] == db.clade_id[db.clade_names.index(og)]  # Euka.cpp:561-569: appended to the detected list
    frag = files2["_FragNames.tsv"].split(b"\n")
    assert frag[0].split(b"\t")[0].decode() in db.clade_names and len(frag[0].split(b"\t")) > 5
    _both(tmp_path, "og2", db, odb, ref, seq_len, names, min_reads=cut, out_group=og, run_mcmc=False, **common)
    # everything rejected: no clade rows marked yes, NaN means in the combined profiles, same bytes on both sides
    det3, _, files3 = _both(tmp_path, "none", db, odb, ref, seq_len, names, min_reads=10 ** 9, **common)
    assert len(det3) == 0 and b"yes" not in files3["_abundance.tsv"] and b"nan" in files3["_5p.prof"]
    # default thresholds (entropy 1.17, six bins)
    _both(tmp_path, "dflt", db, odb, ref, seq_len, names, iters=300, burnin=20, seed=2)


def test_detect_and_truncated_coverage_rule():
    db, odb, ref, seq_len, names = _per_read(1500, seed=3, n_clades=6)
    cov = ref["bin_cov"].copy()
    ids = ek.detect(db, ref["clade_count"], cov, min_bins=1, entropy=0.0)
    assert len(ids) >= 2
    # a coverage below 1 in one scored bin counts as an empty bin (vector<int> in the reference) ...
    c = int(ids[0])
    b0 = int(db.bin_off[c])
    cov[b0] = 0.99
    assert c not in ek.detect(db, ref["clade_count"], cov, min_bins=1, entropy=0.0)
    assert c in ek.detect(db, ref["clade_count"], cov, min_bins=1, entropy=0.0, max_zero_bins=1)
    # ... but not in the clade's last bin, which is never scored
    cov = ref["bin_cov"].copy()
    cov[int(db.bin_off[c + 1]) - 1] = 0.0
    assert c in ek.detect(db, ref["clade_count"], cov, min_bins=1, entropy=0.0)
    with pytest.raises(N.NativeError):
        ek.abundance_mcmc([0.5, 0.5], [10, 10], [-3.0, -4.0], iters=10, burnin=9)


def test_chain_statistics_and_degenerate_sums():
    # two clades, 200 / 600 reads with like = 0.9: the likelihood n1 log f1 + n2 log f2 peaks at f = n / sum(n)
    n = np.array([200, 600], np.int64)
    s = n * np.log(0.9)
    est = ek.abundance_mcmc([0.5, 0.5], n, s, iters=4000, burnin=200, seed=11)
    assert np.all(est[:, 3] < est[:, 0]) and np.all(est[:, 0] < est[:, 4])
    assert abs(est[0, 0] - 0.25) < 0.05 and abs(est[1, 0] - 0.75) < 0.05
    # a read with like == 0 makes the clade's sum -inf: no proposal is ever accepted (exp(-inf) == 0), the recorded
    # proposals all come from the start vector
    est0 = ek.abundance_mcmc([0.3, 0.7], n, np.array([-np.inf, s[1]]), iters=2000, burnin=100, seed=5)
    assert abs(est0[0, 0] - 0.3) < 0.03 and abs(est0[1, 0] - 0.7) < 0.03
    # a clade without entries contributes nothing (and no 0 * log 0)
    est1 = ek.abundance_mcmc([0.5, 0.5], np.array([0, 50]), np.array([0.0, -5.0]), iters=500, burnin=50, seed=2)
    assert np.all(np.isfinite(est1))
