"""GPU tests of soibean's front half on the device (csrc/sb_flatten_kernels.hip) and of `vgan soibean` over the device front end's pipeline
(csrc/sb_gam_run.hip: vgan_sb_gam_*): byte / index work in front of analyse_GAM's kernels, so the batch is array for array the host
flatten's and the tables, the signature counts and every sum over reads are the host pipeline's -- bit for bit (the sums are integers)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from vgan_amd import euka as ek
from vgan_amd import haplocart as hc
from vgan_amd import soibean as sb
from test_sb_cpu import FREQS
from test_sb_gpu import _chain_files, _soibean_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PER_READ = ("read_gseq_len", "read_rseq_len", "read_rev")
_PER_SEG = ("seg_node", "seg_col", "seg_len", "seg_base_ix")


def _by_read(arr):
    """A batch's arrays as {read_src: (scalars, graph_seq, read_seq, qual, segment arrays)}."""
    out = {}
    for i in range(len(arr["read_src"])):
        c0, c1 = int(arr["read_col_off"][i]), int(arr["read_col_off"][i + 1])
        q0, q1 = int(arr["read_qual_off"][i]), int(arr["read_qual_off"][i + 1])
        s0, s1 = int(arr["read_seg_off"][i]), int(arr["read_seg_off"][i + 1])
        out[int(arr["read_src"][i])] = (tuple(int(arr[k][i]) for k in _PER_READ), arr["graph_seq"][c0:c1].tobytes(), arr["read_seq"][c0:c1].tobytes(),
                                        arr["qual"][q0:q1].tobytes(), tuple(arr[k][s0:s1].tobytes() for k in _PER_SEG))
    return out


def _case(n_reads, tmp_path, seed=6):
    g, _, profs, newick = _soibean_case(n_reads=10)
    a = hc.synth_reads(g, n_reads, seed=seed, read_len=60, indel_rate=0.1, softclip_rate=0.1)
    gam = str(tmp_path / "r.gam")
    a.write_gam(gam)
    dm = ek.Damage.from_text(*(open(p).read() for p in profs))
    return g, a, gam, dm, profs, newick


def test_the_device_flatten_writes_the_host_flattens_batch(tmp_path):
    """file bytes -> vgan_gamdev_parse -> vgan_sb_devflat_append_gamdev against host parse -> vgan_sb_flatten: every read the device takes
    has the host batch's scalars, strings, quality bytes and segments, in the host batch's order; a second append goes behind the first;
    the reads it leaves are those with an indel or a soft clip and those the host refuses; a host batch appended behind is the host batch; analyse_GAM's tables of
    the device batch are the host batch's rows."""
    g, a, gam, dm, _, _ = _case(30000, tmp_path)
    data = open(gam, "rb").read()
    a2 = hc.AlnSet.read_gam(gam, keep_unmapped=False)
    hb = sb.SbHostBatch(g, a2)
    want = hb.arrays()
    ctx = sb.SbContext(g, dm, penalty=7)
    gd = hc.GamDevice().parse(data, keep_unmapped=False)
    R = gd.sizes["reads"]
    assert R == a2.n_reads
    df = sb.SbDeviceFlatten(ctx, g)
    mask = df.append_gamdev(gd, 0)
    got = df.download()
    nd = len(got["read_src"])
    n_left = int(mask.sum())
    assert nd + n_left == R and 0 < n_left < 0.4 * R and nd == df.stats.n_out
    x = a2.arrays()
    plain = np.array([np.all(x["e_from"][x["edit_off"][x["map_off"][r]]:x["edit_off"][x["map_off"][r + 1]]] ==
                             x["e_to"][x["edit_off"][x["map_off"][r]]:x["edit_off"][x["map_off"][r + 1]]]) for r in range(R)])
    in_host = np.zeros(R, bool)
    in_host[want["read_src"]] = True  # (the reads vgan_sb_flatten refuses -- here: fewer than 15 bases -- are the host's to refuse)
    assert np.array_equal(mask == 0, plain & in_host) and hb.stats.n_bad > 0
    w, gt = _by_read(want), _by_read(got)
    assert set(gt) <= set(w) and len(gt) == nd
    for src, v in gt.items():
        assert v == w[src], src
    assert np.array_equal(got["read_src"], np.array([s for s in want["read_src"] if s in gt], np.uint32))
    assert got["read_seg_off"][0] == got["read_col_off"][0] == got["read_qual_off"][0] == 0
    assert got["read_rev"].any() and not got["read_rev"].all()
    # a second append: the same rows again behind the first, offsets carried on, read_src from the new base
    mask2 = df.append_gamdev(gd, R)
    assert np.array_equal(mask, mask2)
    two = df.download()
    assert len(two["read_src"]) == 2 * nd
    for k in ("read_gseq_len", "read_rseq_len", "read_rev"):
        assert np.array_equal(two[k][:nd], got[k]) and np.array_equal(two[k][nd:], got[k]), k
    assert np.array_equal(two["read_src"][nd:], got["read_src"] + R)
    for k, tot in (("read_seg_off", "seg_node"), ("read_col_off", "graph_seq"), ("read_qual_off", "qual")):
        assert np.array_equal(two[k][:nd + 1], got[k]) and np.array_equal(two[k][nd:], got[k] + len(got[tot])), k
    for k in _PER_SEG + ("graph_seq", "read_seq", "qual"):
        assert np.array_equal(two[k][:len(got[k])], got[k]) and np.array_equal(two[k][len(got[k]):], got[k]), k
    df.close()
    # the host's batch appended behind the device's rows
    df = sb.SbDeviceFlatten(ctx, g)
    df.append_gamdev(gd, 0)
    df.append_host(hb, np.arange(R, dtype=np.uint32) + 1000)
    both = df.download()
    nh = hb.n_reads
    assert len(both["read_src"]) == nd + nh and np.array_equal(both["read_src"][nd:], want["read_src"] + 1000)
    tail = {k: both[k][nd:] for k in _PER_READ}
    for k in _PER_READ:
        assert np.array_equal(tail[k], want[k]), k
    for k, tot in (("read_seg_off", "seg_node"), ("read_col_off", "graph_seq"), ("read_qual_off", "qual")):
        assert np.array_equal(both[k][nd:], want[k] + len(got[tot])), k
    for k in _PER_SEG + ("graph_seq", "read_seq", "qual"):
        assert np.array_equal(both[k][len(got[k]):], want[k]), k
    # analyse_GAM over the device batch against the host batch's tables
    host_ctx = sb.SbContext(g, dm, penalty=7)
    assert host_ctx.precompute(hb) == 0
    pm_h, cnt_h, ok_h = host_ctx.read_tables()
    db = df.batch()
    assert ctx.precompute(db) == 0
    pm_d, cnt_d, ok_d = ctx.read_tables()
    pos = {int(s): i for i, s in enumerate(want["read_src"])}
    sel = np.array([pos[int(s)] for s in got["read_src"]] + list(range(nh)))
    assert np.array_equal(pm_d, pm_h[:, sel]) and np.array_equal(cnt_d, cnt_h[..., sel]) and np.array_equal(ok_d, ok_h[sel])
    df.close()
    gd.close()
    ctx.close()
    host_ctx.close()


_STATES = [[(1, 0, 0.02, 0.3, 1.0)], [(3, 2, 0.01, 0.7, 1.0)], [(5, 4, 0.0, 0.5, 1.0)]]
_K3 = [[(1, 0, 0.02, 0.3, 0.5), (3, 2, 0.01, 0.7, 0.3), (6, 5, 0.03, 0.2, 0.2)]]


@pytest.mark.parametrize("n_ctx,piece_bytes,slots", [(1, 300_000, 3), (3, 200_000, 2), (1, 1 << 30, 1)])
def test_the_pipeline_leaves_the_contexts_as_the_host_pipeline_does(tmp_path, n_ctx, piece_bytes, slots):
    g, a, gam, dm, _, _ = _case(60000, tmp_path, seed=8)
    data = open(gam, "rb").read()
    a2 = hc.AlnSet.read_gam(gam, keep_unmapped=False)
    hb = sb.SbHostBatch(g, a2)
    want = hb.arrays()
    one = sb.SbContext(g, dm, penalty=7)
    assert one.precompute(hb) == 0
    ctxs = [sb.SbContext(g, dm, penalty=7) for _ in range(n_ctx)]
    got, ps = sb.gam_run(ctxs, g, data, piece_bytes=piece_bytes, slots=slots, n_threads=4, batches=True)
    assert ps["n_pieces"] == len(hc.gampipe_plan(data, piece_bytes)) and (n_ctx == 1 or ps["n_pieces"] > 6)
    assert got["n_messages"] == a.n_reads and got["n_mapped"] == a2.n_reads and got["n_reads"] == hb.n_reads and got["n_bad"] == hb.stats.n_bad
    assert got["n_dev_bad"] == 0 and sum(got["lane_reads"]) == hb.n_reads and all(n > 0 for n in got["lane_reads"])
    assert 0 < ps["n_host_reads"] < 0.4 * a2.n_reads and ps["n_device_reads"] + ps["n_host_reads"] >= hb.n_reads
    # every read of the host batch is in exactly one lane's batch, array for array
    w = _by_read(want)
    seen = {}
    for b in got["batches"]:
        for src, v in _by_read(b).items():
            assert src not in seen and v == w[src], src
            seen[src] = 1
    assert len(seen) == len(w)
    # sums over reads: the contexts' integers add up to the one context's; signature counts likewise
    for sts in (_STATES, _K3):
        whole, _ = one.loglike_sums(sts, 0.01, FREQS)
        split = [c.loglike_sums(sts, 0.01, FREQS) for c in ctxs]
        for e in range(len(sts)):
            assert (sum(s[0][e][0] for s in split), sum(s[0][e][1] for s in split)) == whole[e][:2]
            assert sb.sum_value([s[0][e] for s in split]) == sb.sum_value([whole[e]])
    grp = sb.SbGroup(ctxs)
    _, sig1, n1 = one.best_paths()
    sigg, ng = grp.best_paths()
    assert np.array_equal(sig1, sigg) and n1 == ng == hb.n_reads
    grp.close()
    for c in ctxs + [one]:
        c.close()


def test_vgan_soibean_over_the_device_front_end_writes_the_host_pipelines_files(tmp_path):
    """`vgan soibean` with the device front end forced on (one context, and three sharing the GPU) against the host pipeline: the same
    counts on stderr, the same chain files byte for byte."""
    g, a, gam, dm, profs, newick = _case(40000, tmp_path, seed=12)
    db = tmp_path / "db"
    (db / "tree_dir").mkdir(parents=True)
    g.write(str(db))
    shutil.move(str(db / "graph.gfa"), str(db / "Synth.gfa"))
    (db / "tree_dir" / "Synth.new.dnd").write_text(newick + "\n")
    (db / "soibean_db.baseFreq").write_text("Other .25 .25 .25 .25\nSynth .31 .25 .15 .29\n")
    exe = os.path.join(ROOT, "vgan_amd", "bin", "vgan")
    outs = {}
    for tag, gpus, env in (("host", "0", {"VGAN_SB_DEVICE_GAM": "0"}), ("dev", "0", {"VGAN_SB_DEVICE_GAM": "1", "VGAN_GAMPIPE_PIECE": "300000", "VGAN_TIMING": "1"}),
                           ("dev3", "0,0,0", {"VGAN_SB_DEVICE_GAM": "1", "VGAN_GAMPIPE_PIECE": "300000", "VGAN_TIMING": "1"}),
                           # (a piece the device flatten refuses -- its columns beyond the cap the test sets: the host pipeline takes the file from its start)
                           ("refused", "0", {"VGAN_SB_DEVICE_GAM": "1", "VGAN_GAMPIPE_PIECE": "300000", "VGAN_TIMING": "1", "VGAN_SB_DEVFLAT_MAX_COLS": "1000"})):
        out = str(tmp_path / (tag + "_"))
        r = subprocess.run([exe, "soibean", "-g", gam, "--soibean_dir", str(db), "--dbprefix", "Synth", "--deam5p", profs[0], "--deam3p", profs[1], "--iter", "120",
                            "--burnin", "20", "--chains", "2", "--seed", "7", "-o", out, "--gpus", gpus, "-t", "-1"], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-3000:]
        assert ("soibean device front end" in r.stderr) == (tag in ("dev", "dev3")), r.stderr[-1500:]
        assert ("the host pipeline does" in r.stderr and "32-bit offsets" in r.stderr) == (tag == "refused"), r.stderr[-1500:]
        if tag == "dev3":
            assert "on 3 lane(s)" in r.stderr and "3 device contexts, the reads dealt piece by piece" in r.stderr
        outs[tag] = (_chain_files(out), [l for l in r.stderr.splitlines() if "log-likelihood" in l or "signature" in l or l.startswith("Number of")])
    assert len(outs["host"][0]) >= 7 and len(outs["host"][1]) >= 3
    for other in ("dev", "dev3", "refused"):
        assert sorted(outs["host"][0]) == sorted(outs[other][0])
        for name in outs["host"][0]:
            assert outs["host"][0][name] == outs[other][0][name], name
        assert outs["host"][1] == outs[other][1]


def _corrupted(a, g, seed, n_max=400):
    """The alignment set with node ids, edit lengths, offsets and strands changed at random (what a front half has to judge, not to trust)."""
    import orc
    from vgan_amd import _native as N
    rng = np.random.default_rng(seed)
    arr = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in a.arrays().items()}
    for _ in range(int(rng.integers(50, n_max))):
        w = rng.integers(7)
        if w == 0:
            arr["m_node"][rng.integers(len(arr["m_node"]))] = rng.choice([0, 1, int(g.max_id), int(g.max_id) + 1, int(rng.integers(1, g.max_id + 1))])
        elif w == 1:
            arr["e_from"][rng.integers(len(arr["e_from"]))] = rng.integers(0, 6)
        elif w == 2:
            arr["e_to"][rng.integers(len(arr["e_to"]))] = rng.integers(0, 6)
        elif w == 3:
            arr["m_offset"][rng.integers(len(arr["m_offset"]))] = rng.choice([0, 1, 2, 3, 7, 1 << 20])
        elif w == 4:
            arr["m_rev"][rng.integers(len(arr["m_rev"]))] ^= 1
        elif w == 5:  # (both lengths of an edit, together: still a match, of another length)
            e = rng.integers(len(arr["e_from"]))
            arr["e_from"][e] = arr["e_to"][e] = rng.integers(0, 40)
        else:  # (a whole read onto the other strand)
            r = rng.integers(a.n_reads)
            arr["m_rev"][arr["map_off"][r]:arr["map_off"][r + 1]] ^= 1
    oa = orc.AlnSet.from_arrays(**arr)
    v = N.AlnSetView(oa.n_reads, *[getattr(oa, k).ctypes.data for k in ("seq_off", "seq", "qual_off", "qual", "mapq", "identity")], None, None,
                     *[getattr(oa, k).ctypes.data for k in ("map_off", "m_node", "m_offset", "m_rev", "edit_off", "e_from", "e_to", "e_seq_off", "e_seq")])
    h = N.vp()
    N.check(N.lib().vgan_aln_from_arrays(v, h))
    out = hc.AlnSet(h)
    out._keep = oa
    return out


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_corrupted_alignments_are_taken_or_left_as_the_host_flatten_judges_them(tmp_path, seed):
    """Alignments with node ids, offsets, edit lengths and strands changed at random: every read the device flatten takes has the host
    flatten's rows -- none that the host refuses is taken --, the others are the host's: the batches of the pipeline are the host batch."""
    g, _, profs, _ = _soibean_case(n_reads=10)
    a0 = hc.synth_reads(g, 3000, seed=40 + seed, read_len=50, indel_rate=0.1, softclip_rate=0.1)
    a = _corrupted(a0, g, seed)
    gam = str(tmp_path / "c.gam")
    a.write_gam(gam)
    data = open(gam, "rb").read()
    a2 = hc.AlnSet.read_gam(gam, keep_unmapped=False)
    hb = sb.SbHostBatch(g, a2)
    want = hb.arrays()
    assert hb.stats.n_bad > 0 and hb.n_reads > 1000
    dm = ek.Damage.from_text(*(open(p).read() for p in profs))
    ctx = sb.SbContext(g, dm, penalty=7)
    got, ps = sb.gam_run([ctx], g, data, piece_bytes=100_000, slots=2, n_threads=4, batches=True)
    assert got["n_reads"] == hb.n_reads and got["n_bad"] == hb.stats.n_bad and ps["n_device_reads"] > 0 and ps["n_host_reads"] > 0
    w = _by_read(want)
    gt = _by_read(got["batches"][0])
    assert len(gt) == len(w)
    for src, v in gt.items():
        assert v == w[src], src
    ctx.close()
