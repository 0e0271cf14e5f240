"""GPU tests of euka's front half on the device (csrc/euka_flatten_kernels.hip) and of `vgan euka` over the device front end's pipeline
(csrc/gam_pipe.hip: vgan_euka_gam_*): byte / integer work in front of the read kernel, so the batch is array for array the host flatten's
and the per-read results are the host pipeline's, read for read."""
import os
import subprocess

import numpy as np
import pytest

import util
from vgan_amd import euka as ek
from vgan_amd import haplocart as hc
from test_euka_cpu import GOLD
from test_euka_gpu import _tree

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PER_READ = ("read_gseq_len", "read_rseq_len", "read_seq_len", "read_mapq", "read_rev")


def _damage():
    d = os.path.join(GOLD, "damageProfiles")
    return d + "/dhigh5p.prof", d + "/dhigh3p.prof"


def _by_read(arr, R):
    """A batch's arrays as {read_src: (scalars, graph_seq, read_seq, qual, map_node)}."""
    out = {}
    for i in range(R):
        c0, c1 = int(arr["read_col_off"][i]), int(arr["read_col_off"][i + 1])
        q0, q1 = int(arr["read_qual_off"][i]), int(arr["read_qual_off"][i + 1])
        m0, m1 = int(arr["read_map_off"][i]), int(arr["read_map_off"][i + 1])
        out[int(arr["read_src"][i])] = (tuple(int(arr[k][i]) for k in _PER_READ), arr["graph_seq"][c0:c1].tobytes(), arr["read_seq"][c0:c1].tobytes(),
                                        arr["qual"][q0:q1].tobytes(), arr["map_node"][m0:m1].tobytes())
    return out


def test_the_device_flatten_writes_the_host_flattens_batch(tmp_path):
    """file bytes -> vgan_gamdev_parse -> vgan_euka_devflat_run_gamdev against host parse -> vgan_euka_flatten: every read the device
    takes has the host batch's scalars, strings, quality bytes and node ids; the device batch is in the host batch's order (first
    node id, input order among equals); the reads it leaves are exactly those with an indel or a soft clip; the read kernel gives
    the same per-read results on either batch."""
    p5, p3 = _damage()
    dm = ek.Damage.load(p5, p3)
    g, db, a = ek.synth_euka(30000, dm, seed=41, n_clades=10, nodes_per_clade=180)
    gam = str(tmp_path / "r.gam")
    a.write_gam(gam)
    data = open(gam, "rb").read()
    a2 = hc.AlnSet.read_gam(gam, keep_unmapped=False)  # (identity == 0 is dropped by either parser: readGAM_Euka.h:72)
    hb = ek.EukaHostBatch(g, a2)
    want = hb.arrays()
    ctx = ek.EukaContext(db, dm)
    gd = hc.GamDevice().parse(data, keep_unmapped=False)
    assert gd.sizes["reads"] == a2.n_reads
    df = ek.EukaDeviceFlatten(ctx, g).run_gamdev(gd)
    got = df.download()
    R = df.c.n_reads
    n_left = int(df.mask.sum())
    assert R + n_left == a2.n_reads and 0 < n_left < 0.3 * a2.n_reads
    x = a2.arrays()
    plain = np.array([np.all(x["e_from"][x["edit_off"][x["map_off"][r]]:x["edit_off"][x["map_off"][r + 1]]] ==
                             x["e_to"][x["edit_off"][x["map_off"][r]]:x["edit_off"][x["map_off"][r + 1]]]) for r in range(a2.n_reads)])
    assert np.array_equal(df.mask == 0, plain)  # (the synthetic reads lie on known nodes within the tables' lengths)
    w, gt = _by_read(want, hb.n_reads), _by_read(got, R)
    assert set(gt) <= set(w)
    for src, v in gt.items():
        assert v == w[src], src
    order = [s for s in want["read_src"] if s in gt]
    assert np.array_equal(got["read_src"], np.array(order, np.uint32))
    # the read kernel on the device batch against the host batch's results
    res_h = ctx.accumulate(hb)
    ctx.reset()
    res_d = df.accumulate()
    pos = {int(s): i for i, s in enumerate(want["read_src"])}
    sel = np.array([pos[int(s)] for s in got["read_src"]])
    assert np.array_equal(res_d["clade"], res_h["clade"][sel]) and np.array_equal(res_d["pass"], res_h["pass"][sel])
    assert util.rel_err(res_d["in_lik"], res_h["in_lik"][sel]) < 1e-13
    df.close()
    gd.close()
    ctx.close()


def _host_run(g, db, dm, gam, n_ctx=1):
    a = hc.AlnSet.read_gam(gam, keep_unmapped=True)
    ctx = ek.EukaContext(db, dm)
    hb = ek.EukaHostBatch(g, a)
    out = ctx.accumulate(hb)
    src = hb.arrays()["read_src"]
    order = np.argsort(src, kind="stable")
    fin = ctx.finalize()
    n_like, s_like = ctx.like_sums()
    ident = a.arrays()["identity"]
    kept_rank = np.cumsum(ident != 0) - 1  # a read's index among the file's mapped reads
    return {"n_messages": a.n_reads, "n_mapped": int((ident != 0).sum()), "n_bad": hb.stats.n_bad, "read_index": kept_rank[src[order]].astype(np.uint32),
            "read_clade": out["clade"][order], "read_pass": out["pass"][order], "read_seq_len": hb.arrays()["read_seq_len"][order], "fin": fin,
            "like": (n_like, s_like)}


@pytest.mark.parametrize("n_ctx,piece_bytes,slots", [(1, 300_000, 3), (3, 200_000, 2), (1, 1 << 30, 1)])
def test_the_pipeline_gives_the_host_pipelines_results(tmp_path, n_ctx, piece_bytes, slots):
    p5, p3 = _damage()
    dm = ek.Damage.load(p5, p3)
    g, db, a = ek.synth_euka(60000, dm, seed=43, n_clades=12, nodes_per_clade=180)
    gam = str(tmp_path / "r.gam")
    a.write_gam(gam)
    data = open(gam, "rb").read()
    want = _host_run(g, db, dm, gam)
    ctxs = [ek.EukaContext(db, dm) for _ in range(n_ctx)]
    got, ps = ek.gam_run(ctxs, g, data, piece_bytes=piece_bytes, slots=slots, n_threads=4)
    assert ps["n_pieces"] == len(hc.gampipe_plan(data, piece_bytes)) and (n_ctx == 1 or ps["n_pieces"] > 6)
    for k in ("n_messages", "n_mapped", "n_bad"):
        assert got[k] == want[k], k
    for k in ("read_index", "read_clade", "read_pass", "read_seq_len"):
        assert np.array_equal(got[k], want[k]), k
    assert 0 < ps["n_host_reads"] < 0.3 * want["n_mapped"] and ps["n_device_reads"] + ps["n_host_reads"] >= len(want["read_index"])
    fins = [c.finalize() for c in ctxs]
    assert np.array_equal(sum(f["clade_count"] for f in fins), want["fin"]["clade_count"])
    assert np.array_equal(sum(f["baseshift"].astype(np.int64) for f in fins), want["fin"]["baseshift"].astype(np.int64))
    assert np.allclose(sum(f["bin_cov"] for f in fins), want["fin"]["bin_cov"], rtol=1e-10, atol=1e-10)
    likes = [c.like_sums() for c in ctxs]
    assert np.array_equal(sum(l[0] for l in likes), want["like"][0])
    assert np.allclose(sum(l[1] for l in likes), want["like"][1], rtol=1e-10, atol=1e-9)
    for c in ctxs:
        c.close()


def test_vgan_euka_over_the_device_front_end_writes_the_host_pipelines_files(tmp_path):
    """`vgan euka` with the device front end forced on (one context, and three sharing the GPU) against the host pipeline: the same
    counts on stderr, the same files (the coverage sums are doubles added in another order: compared as numbers); --outFrag keeps the
    host pipeline (the device does not keep the reads' names)."""
    from test_sb_gpu import _same_tables
    p5, p3 = _damage()
    dm = ek.Damage.load(p5, p3)
    g, db, a = ek.synth_euka(120_000, dm, seed=31, n_clades=12, nodes_per_clade=180)
    util.write_euka_db(db, g, tmp_path)
    gam = str(tmp_path / "reads.gam")
    a.write_gam(gam)
    exe = os.path.join(ROOT, "vgan_amd", "bin", "vgan")
    args = ["--entropy", "0", "--minBins", "2", "--minFrag", "40", "-l", "4", "--seed", "5", "--iter", "500", "--burnin", "50", "--minMQ", "20", "-t", "-1"]
    outs = {}
    for tag, extra, env in (("host", [], {"VGAN_EUKA_DEVICE_GAM": "0"}), ("dev", [], {"VGAN_EUKA_DEVICE_GAM": "1", "VGAN_GAMPIPE_PIECE": "400000", "VGAN_TIMING": "1"}),
                            ("dev3", ["--gpus", "0,0,0"], {"VGAN_EUKA_DEVICE_GAM": "1", "VGAN_GAMPIPE_PIECE": "400000", "VGAN_TIMING": "1"}),
                            ("frag", ["--outFrag"], {"VGAN_EUKA_DEVICE_GAM": "1", "VGAN_TIMING": "1"}),
                            # (a piece the device flatten refuses -- its columns beyond the cap the test sets: the contexts are cleared and the host pipeline takes the file)
                            ("refused", [], {"VGAN_EUKA_DEVICE_GAM": "1", "VGAN_GAMPIPE_PIECE": "400000", "VGAN_TIMING": "1", "VGAN_EUKA_DEVFLAT_MAX_COLS": "1000"})):
        r = subprocess.run([exe, "euka", "-g", gam, "--euka_dir", str(tmp_path), "--deam5p", p5, "--deam3p", p3, "-o", str(tmp_path / tag)] + args + extra,
                           capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("euka device front end" in r.stderr) == (tag in ("dev", "dev3")), r.stderr[-1500:]
        assert ("the host pipeline does" in r.stderr and "32-bit offsets" in r.stderr) == (tag == "refused"), r.stderr[-1500:]
        if tag == "dev3":
            assert "on 3 lane(s)" in r.stderr
        outs[tag] = _tree(str(tmp_path / tag))
        outs[tag + "_counts"] = [ln for ln in r.stderr.splitlines() if ln.startswith("Number of")]
    assert outs["host_counts"] == outs["dev_counts"] == outs["dev3_counts"] == outs["refused_counts"] and len(outs["host_counts"]) == 3
    for other in ("dev", "dev3", "refused"):
        assert sorted(outs["host"]) == sorted(outs[other]) and len(outs["host"]) >= 6
        for k in outs["host"]:
            if outs["host"][k] != outs[other][k]:  # only rounding of summed doubles may differ
                _same_tables(outs["host"][k], outs[other][k], 1e-9)
    assert set(outs["host"]) <= set(outs["frag"])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_corrupted_alignments_give_the_host_pipelines_results(tmp_path, seed):
    """Alignments with node ids, offsets, edit lengths and strands changed at random through the pipeline: the same reads processed, the
    same per-read results and counts of refused reads as host parse -> vgan_euka_flatten -> the read kernel."""
    from test_sb_pipe_gpu import _corrupted
    p5, p3 = _damage()
    dm = ek.Damage.load(p5, p3)
    g, db, a0 = ek.synth_euka(4000, dm, seed=50 + seed, n_clades=8, nodes_per_clade=150)
    a = _corrupted(a0, g, seed)
    gam = str(tmp_path / "c.gam")
    a.write_gam(gam)
    data = open(gam, "rb").read()
    want = _host_run(g, db, dm, gam)
    ctx = ek.EukaContext(db, dm)
    got, ps = ek.gam_run([ctx], g, data, piece_bytes=100_000, slots=2, n_threads=4)
    for k in ("n_messages", "n_mapped", "n_bad"):
        assert got[k] == want[k], k
    for k in ("read_index", "read_clade", "read_pass", "read_seq_len"):
        assert np.array_equal(got[k], want[k]), k
    assert want["n_bad"] > 0 and ps["n_device_reads"] > 0 and ps["n_host_reads"] > 0
    fin = ctx.finalize()
    assert np.array_equal(fin["clade_count"], want["fin"]["clade_count"]) and np.array_equal(fin["baseshift"], want["fin"]["baseshift"])
    ctx.close()
