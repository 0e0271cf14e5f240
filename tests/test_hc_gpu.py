"""GPU parity tests of the HaploCart path (run with -m gpu on an MI355X).  Everything goes through the C-ABI
(libvgan_gpu.so); the oracle (oracle/liboracle.so) is the checker.  Tolerance: 1e-6 relative is the bar
BASELINE.json's north_star states; these tests hold the device path to 1e-9 (fp64 vs long double)."""
import os

import numpy as np
import pytest

import gamio
import orc
import util
from vgan_amd import haplocart as hc

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def toy(golden_dir):
    d = os.path.join(golden_dir, "reconstruct")
    g = hc.Graph.load(os.path.join(d, "target_graph.gfa"))
    a = hc.AlnSet.read_gam(os.path.join(d, "test_reads.gam"))
    return g, a


def check_final(ctx, batch, og, oa, faithful, modes=(hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ, hc.MODE_PER_READ_DENSE)):
    _, ref, bad = orc.hc_run(og, oa, n_threads=8, faithful=faithful)
    for mode in modes:
        ctx.reset()
        ctx.set_mode(mode)
        ctx.accumulate(batch)
        got = ctx.finalize()
        assert util.rel_err(got, ref) < RTOL, (mode, util.rel_err(got, ref))
    return ref


def test_toy_graph_segments_reads_final(golden_dir):
    g, a = toy(golden_dir)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    b = hc.HostBatch(g, a)
    ctx = hc.HcContext(g)
    S, U = ctx.segment_scalars(b)
    arr = b.arrays()
    src = b.read_src.tolist()  # the batch is ordered for the device (tileable reads by node id first), not as the input
    assert sorted(src) == list(range(a.n_reads))
    for k, r in enumerate(src):
        rc, So, Uo, node = orc.hc_read_segments(og, oa, r)
        assert rc == 0
        s0, s1 = arr["read_seg_off"][k], arr["read_seg_off"][k + 1]
        assert util.rel_err(S[s0:s1], So) < 1e-12 and util.rel_err(U[s0:s1], Uo) < 1e-12
    ll = ctx.read_loglik(b)
    for k, r in enumerate(src):
        rc, ref, _ = orc.hc_read(og, oa, r)  # the literal per-path loops
        assert rc == 0
        assert util.rel_err(ll[k], ref.astype(np.float64)) < 1e-12, r
    check_final(ctx, b, og, oa, faithful=True)


def test_bundled_alignments_as_plumbing(golden_dir):
    """BASELINE config 1: the reference's real-graph GAMs run against a synthetic hcfiles-shaped graph (node ids fit)."""
    g = hc.synth_graph(seed=0x76676131)
    assert (g.max_id, g.n_paths) == (11821, 5179)
    og = util.orc_graph_from_product(g)
    ctx = hc.HcContext(g)
    for f in ["J2a1a1a1.gam", "two_unique.gam", "all_the_same_reverse.gam"]:
        a = hc.AlnSet.read_gam(os.path.join(golden_dir, "alignments", f))
        b = hc.HostBatch(g, a)
        assert b.stats.n_out + b.stats.n_bad == a.n_reads
        oa = util.orc_alnset_from_product(a)
        _, ref, bad = orc.hc_run(og, oa, n_threads=8, faithful=False)
        assert bad == b.stats.n_bad
        ctx.reset()
        ctx.set_mode(hc.MODE_NODE_WEIGHTS)
        ctx.accumulate(b)
        got = ctx.finalize()
        assert util.rel_err(got, ref) < RTOL
        if b.n_reads:  # exactly tied paths may resolve either way in the last bit
            assert ref[ctx.argmax(got)] >= ref.max() - 1e-9 * abs(ref.max())


@pytest.mark.parametrize("kw", [dict(), dict(background_error_prob=0.01, use_background_error_prob=True, is_consensus_fasta=True),
                                dict(background_error_prob=0.3)])
def test_small_synth_all_modes_vs_faithful_oracle(kw, tmp_path):
    g = hc.synth_graph(seed=21, genome_len=1500, n_nodes=1000, n_paths=200)
    # reads for the tiled kernel (up to a whole tile: 1280 columns / 512 mappings) and, beyond, for the general one
    a = util.concat_alnsets(tmp_path, hc.synth_reads(g, 90, seed=5, read_len=150, indel_rate=0.2, softclip_rate=0.2),
                            hc.synth_reads(g, 20, seed=6, read_len=700, indel_rate=0.2, softclip_rate=0.2),
                            hc.synth_reads(g, 20, seed=7, read_len=1400, indel_rate=0.2, softclip_rate=0.2))
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    b = hc.HostBatch(g, a)
    ctx = hc.HcContext(g, **kw)
    p = orc.hc_params(kw.get("background_error_prob", 0.0001), kw.get("use_background_error_prob", False),
                      kw.get("is_consensus_fasta", False))
    _, ref, _ = orc.hc_run(og, oa, p, n_threads=8, faithful=True)
    for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ, hc.MODE_PER_READ_DENSE):
        ctx.reset()
        ctx.set_mode(mode)
        ctx.accumulate(b)
        assert util.rel_err(ctx.finalize(), ref) < RTOL, mode
    ll = ctx.read_loglik(b)
    src = b.read_src  # the batch holds the tileable reads first: map back to the alignment set
    assert 0 < b.n_tileable < b.n_reads and sorted(src.tolist()) == sorted(set(src.tolist()))
    ident = a.arrays()["identity"]
    kept = [r for r in range(a.n_reads) if ident[r] >= 1e-10 and orc.hc_read(og, oa, r, p)[0] == 0]
    assert sorted(src.tolist()) == kept
    for k in range(0, b.n_reads, 5):
        rc, refr, _ = orc.hc_read(og, oa, int(src[k]), p)
        assert rc == 0
        assert util.rel_err(ll[k], refr.astype(np.float64)) < 1e-11, (k, src[k])


@pytest.mark.parametrize("n_nodes,read_len", [(60, 150), (400, 100), (3000, 250)])
def test_tiled_and_general_kernels_agree_per_segment(n_nodes, read_len):
    """D_m from the LDS-tiled kernel (through NODE_WEIGHTS sums per node) against S_m - U_m of the general kernel, on
    graphs with long nodes (owner marks across 32-column words) and short ones; regular reads only."""
    g = hc.synth_graph(seed=77, genome_len=3000, n_nodes=n_nodes, n_paths=96)
    a = hc.synth_reads(g, 4000, seed=9, read_len=read_len, indel_rate=0.0, softclip_rate=0.0)
    b = hc.HostBatch(g, a)
    assert b.n_tileable == b.n_reads > 0
    ctx = hc.HcContext(g)
    S, U = ctx.segment_scalars(b)  # general kernel
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    _, ref, _ = orc.hc_run(og, oa, n_threads=8, faithful=False)
    for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ):
        ctx.reset()
        ctx.set_mode(mode)
        ctx.accumulate(b)  # tiled kernel
        got = ctx.finalize()
        assert util.rel_err(got, ref) < RTOL
    # the same final vector from the general kernel's S and U: final[p] = sum S - sum_{p unsupported} (S - U)
    arr = b.arrays()
    um = util.unsupported_mask(g)  # [rows][P] bool
    D = S - U
    W = np.zeros(um.shape[0])
    np.add.at(W, arr["seg_node"], D)
    want = S.sum() - um.T.astype(np.float64) @ W
    assert util.rel_err(got, want) < 1e-12


def _aln(seq, quals, node_edits, mapq=60):
    return {"sequence": seq, "quality": bytes(quals), "mapping_quality": mapq, "identity": 1.0, "name": b"r",
            "path": {"name": b"", "mapping": [{"position": {"node_id": n, "offset": o, "is_reverse": rv},
                                                 "edit": [{"from_length": f, "to_length": t, "sequence": s} for f, t, s in ed],
                                                 "rank": i + 1} for i, (n, o, rv, ed) in enumerate(node_edits)]}}


def test_closed_form_kats_and_edge_cases():
    """SURVEY.md 8c closed forms through the device path + Q>=90 sticky flag, Q<=2, negative / >=100 qualities,
    a read longer than the LDS quality window, an empty batch."""
    # one node ACGT (id 1) at a protein-coding coordinate; node 2 TTAC at HVS-I; paths: 0 supports both, 1 none
    mask = np.zeros((3, 1), np.uint64)
    mask[1, 0] = 1
    mask[2, 0] = 1
    off = np.array([0, 0, 4, 8], np.int64)
    g = hc.Graph.from_arrays(1, 2, off, b"ACGTTTAC", 2, mask, np.array([-1, 4000, 100], np.int32), np.ones(17000),
                             "a\nb\n", "", "")
    ctx = hc.HcContext(g)
    alns = [
        _aln(b"ACGT", [40] * 4, [(1, 0, False, [(4, 4, b"")])]),
        _aln(b"ACGT", [40, 93, 40, 40], [(1, 0, False, [(4, 4, b"")])]),           # sticky background error prob
        _aln(b"ACGTTTAC", [2, 0, 1, 3, 200, 130, 99, 100], [(1, 0, False, [(4, 4, b"")]), (2, 0, False, [(4, 4, b"")])]),
        _aln(b"TTAC" * 400, [30 + (i % 11) for i in range(1600)], [(2, 0, False, [(4, 4, b"")])] * 400),  # QL > 1024
        _aln(b"GTAAACGT", [35] * 8, [(2, 0, True, [(4, 4, b"")]), (1, 0, True, [(2, 2, b""), (1, 1, b"G"), (1, 1, b"")])]),
    ]
    a = hc.AlnSet.parse_gam(gamio.write_gam(alns))
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    b = hc.HostBatch(g, a)
    assert b.n_reads == len(alns)
    ll = ctx.read_loglik(b)
    src = b.read_src.tolist()  # the 1600-column read is not tileable and sits last in the batch
    assert src == [0, 1, 2, 4, 3] and b.n_tileable == 4
    assert ll[0, 0] == pytest.approx(-4.03019902453559e-4, rel=1e-12)
    assert ll[0, 1] == pytest.approx(-36.8413614879047, rel=1e-13)
    for k, r in enumerate(src):
        rc, ref, _ = orc.hc_read(og, oa, r)
        assert rc == 0
        assert util.rel_err(ll[k], ref.astype(np.float64)) < 1e-12, r
    check_final(ctx, b, og, oa, faithful=True)
    # empty batch and empty accumulators: final = 0, posterior of all-zero vector is 1 (oplusInitnatl quirk Q11)
    empty = hc.HostBatch(g, a, 0, 0)
    ctx.reset()
    ctx.accumulate(empty)
    z = ctx.finalize()
    assert np.all(z == 0.0)
    assert ctx.posterior(z, "a")[0][1] == 1.0


def test_posterior_matches_oracle():
    g = hc.synth_graph(seed=33, genome_len=1200, n_nodes=800, n_paths=300)
    a = hc.synth_reads(g, 400, seed=8, read_len=100)
    b = hc.HostBatch(g, a)
    ctx = hc.HcContext(g)
    ctx.accumulate(b)
    fv = ctx.finalize()
    names = g.path_names
    pred = names[ctx.argmax(fv)]
    for predicted in (pred, names[299], names[0]):
        got = ctx.posterior(fv, predicted)
        ref = orc.hc_posterior(fv.astype(np.longdouble), names, g.parents_txt, g.children_txt, predicted)
        assert [x[0] for x in got] == [x[0] for x in ref]
        assert [x[2] for x in got] == [x[2] for x in ref]
        for (n1, c1, _), (n2, c2, _) in zip(got, ref):
            assert c1 == pytest.approx(c2, rel=1e-9, abs=1e-300), n1
    # flat likelihoods: every clade's confidence = |descendant paths| / P
    flat = np.full(300, -5.0)
    got = ctx.posterior(flat, names[299])
    ref = orc.hc_posterior(flat.astype(np.longdouble), names, g.parents_txt, g.children_txt, names[299])
    for (_, c1, _), (_, c2, _) in zip(got, ref):
        assert c1 == pytest.approx(c2, rel=1e-12)


def _graph_with_relatives(g, parents_txt, children_txt):
    return hc.Graph.from_arrays(g.min_id, g.max_id, g.node_seq_off, g.node_seq.tobytes(), g.n_paths, g.mask, g.pangenome_base,
                                g.mappability, "\n".join(g.path_names) + "\n", parents_txt, children_txt)


def test_posterior_on_a_dag_and_on_childless_ancestors():
    """get_posterior.cpp:51-76 builds a fresh child set per recursion level, so a path reachable at two depths of
    children.txt enters the clade's sum twice; an ancestor without (path-name) descendants sums nothing, which the
    oracle header defines as 0 (the reference reads v[0] of an empty vector)."""
    g0 = hc.synth_graph(seed=5, genome_len=600, n_nodes=400, n_paths=12)
    n = g0.path_names
    # n[0] -> n[1], n[2];  n[1] -> n[3];  n[2] -> n[3], n[4];  n[3] -> n[5];  n[4] -> n[5]: n[5] is reached at depth 3 twice
    # over two parents (one set per level: once) and n[3] at depth 2 over two parents (once); n[6] -> n[1] and n[3]: n[3]
    # sits at depths 1 and 2 of n[6] (twice), n[5] at depths 2 and 3 (twice)
    children = "\n".join([
        "%s %s %s" % (n[0], n[1], n[2]), "%s %s" % (n[1], n[3]), "%s %s %s" % (n[2], n[3], n[4]), "%s %s" % (n[3], n[5]),
        "%s %s" % (n[4], n[5]), "%s %s %s" % (n[6], n[1], n[3]), "%s notapath [x]" % n[7], "%s" % n[8]]) + "\n"
    # the predicted path's ancestors: a repeated entry (Q9), a DAG ancestor, a leafless one, one absent from children.txt
    parents = "%s %s %s %s %s %s %s %s\n" % (n[5], n[3], n[3], n[6], n[0], n[7], n[8], n[9])
    g = _graph_with_relatives(g0, parents, children)
    ctx = hc.HcContext(g)
    rng = np.random.default_rng(3)
    for fv in (-(rng.random(12) * 30 + 1), np.full(12, -2.5), np.array([0, 0, -1.0, 0, -3, 0, -2, 0, 0, 0, -7, 0.0]), np.zeros(12)):
        got = ctx.posterior(fv, n[5])
        ref = orc.hc_posterior(fv.astype(np.longdouble), n, parents, children, n[5])
        assert [(x[0], x[2]) for x in got] == [(x[0], x[2]) for x in ref]
        assert [x[0] for x in got] == [n[5], n[3], n[6], n[0], n[7], n[8], n[9]]
        for (n1, c1, _), (n2, c2, _) in zip(got, ref):
            assert c1 == pytest.approx(c2, rel=1e-9, abs=1e-300), n1
    # closed form on flat likelihoods: confidence = (entries of all_top, multiplicity kept) / P
    got = dict((x[0], x[1]) for x in ctx.posterior(np.full(12, -2.5), n[5]))
    assert got[n[6]] == pytest.approx(5 / 12, rel=1e-12)  # n1 | n3 n3 | n5 n5
    assert got[n[0]] == pytest.approx(5 / 12, rel=1e-12)  # n1 n2 | n3 n4 | n5
    assert got[n[7]] == got[n[8]] == got[n[9]]            # nothing summed: exp(0 - total)
    assert got[n[7]] == pytest.approx(np.exp(2.5) / 12, rel=1e-12)


def test_oracle_fed_without_any_product_loader(tmp_path):
    """The loop most parity tests close -- oracle inputs built from the product's parsed arrays -- opened: 12 000 synthetic
    reads are re-encoded into a GAM by the test-side writer (tests/gamio.py); the oracle reads that file with the test-side
    decoder and the graph with the test-side GFA reader and its own sidecar loaders, the product reads both with its C++
    front end.  A mis-parse shared by product and test views would be invisible elsewhere; here it shows."""
    g0 = hc.synth_graph(seed=77, genome_len=3000, n_nodes=2100, n_paths=400)
    a0 = hc.synth_reads(g0, 12_000, seed=78, read_len=100, indel_rate=0.05, softclip_rate=0.05, low_mapq_rate=0.3)
    hcdir = tmp_path / "hcfiles"
    hcdir.mkdir()
    g0.write(str(hcdir))
    gam = str(tmp_path / "reads.gam")
    open(gam, "wb").write(gamio.write_gam(util.gamio_dicts_from_product(a0), group=300))
    del a0
    # oracle side: nothing of the product
    og, names, parents, children = util.orc_graph_from_hcfiles(str(hcdir))
    dicts = gamio.read_gam(gam)
    assert len(dicts) == 12_000
    oa = orc.AlnSet(dicts)
    _, ref, n_bad = orc.hc_run(og, oa, n_threads=8, faithful=False)
    # product side: its own loaders
    g = hc.Graph.load(str(hcdir / "graph.gfa"), str(hcdir))
    a = hc.AlnSet.read_gam(gam)
    b = hc.HostBatch(g, a)
    assert b.stats.n_bad == n_bad and b.n_reads + b.stats.n_bad + b.stats.n_unmapped == 12_000
    assert g.path_names == names
    ctx = hc.HcContext(g)
    for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ):
        ctx.reset()
        ctx.set_mode(mode)
        ctx.accumulate(b)
        assert util.rel_err(ctx.finalize(), ref) < RTOL, mode
    # per-read vectors against the literal per-path loops, and the posterior from the oracle's own relatives text
    src = b.read_src
    ll = ctx.read_loglik(hc.HostBatch(g, a, 0, 400))
    sub = hc.HostBatch(g, a, 0, 400).read_src
    for k in range(0, len(sub), 9):
        rc, want, _ = orc.hc_read(og, oa, int(sub[k]))
        assert rc == 0 and util.rel_err(ll[k], want.astype(np.float64)) < 1e-11, int(sub[k])
    ctx.set_mode(hc.MODE_NODE_WEIGHTS)
    fv = ctx.finalize()
    pred = names[ctx.argmax(fv)]
    got = ctx.posterior(fv, pred)
    want = orc.hc_posterior(fv.astype(np.longdouble), names, parents, children, pred)
    assert [x[0] for x in got] == [x[0] for x in want] and len(src) == b.n_reads
    for (_, c1, _), (_, c2, _) in zip(got, want):
        assert c1 == pytest.approx(c2, rel=1e-9, abs=1e-300)


def _permuted_batch(arr, order, n_tileable):
    """The reads of a flattened batch in another order (offsets rebuilt), as a hand-built host batch."""
    so, co, qo = arr["read_seg_off"], arr["read_col_off"], arr["read_qual_off"]
    out = {k: [] for k in ("seg_node", "seg_start", "seg_len", "graph_seq", "algnseq", "qual")}
    nso, nco, nqo = [0], [0], [0]
    for r in order:
        for k in ("seg_node", "seg_start", "seg_len"):
            out[k].append(arr[k][so[r]:so[r + 1]])
        for k in ("graph_seq", "algnseq"):
            out[k].append(arr[k][co[r]:co[r + 1]])
        out["qual"].append(arr["qual"][qo[r]:qo[r + 1]])
        nso.append(nso[-1] + int(so[r + 1] - so[r]))
        nco.append(nco[-1] + int(co[r + 1] - co[r]))
        nqo.append(nqo[-1] + int(qo[r + 1] - qo[r]))
    arrays = {k: np.concatenate(v) for k, v in out.items()}
    arrays.update(read_seg_off=nso, read_col_off=nco, read_qual_off=nqo, read_algn_len=arr["read_algn_len"][order], read_mapq=arr["read_mapq"][order])
    return hc.ArrayBatch(arrays, n_tileable=n_tileable)


def test_any_read_order_gives_the_same_sums():
    """The tiled kernel keeps W[node] of a workgroup's reads in an LDS window placed at the lowest node id of its first tile and
    expects the batch sorted by node id (vgan_hc_flatten does that).  A batch in any other order -- shuffled, descending, two
    far-apart regions interleaved (every tile leaves the window: segments go to W in HBM directly and the window is placed
    anew) -- must give the same final vector."""
    g = hc.synth_graph(seed=41, genome_len=9000, n_nodes=6400, n_paths=150)
    a = hc.synth_reads(g, 30_000, seed=2, read_len=150)
    b = hc.HostBatch(g, a)
    assert b.n_tileable == b.n_reads
    arr = {k: np.array(v) for k, v in b.arrays().items() if k != "_owner"}
    ctx = hc.HcContext(g)
    ctx.accumulate(b)
    want = ctx.finalize()
    n = b.n_reads
    rng = np.random.default_rng(9)
    orders = {"shuffled": rng.permutation(n), "descending": np.arange(n)[::-1],
              "interleaved": np.stack([np.arange(n // 2), np.arange(n // 2) + n // 2], 1).ravel()}
    for name, order in orders.items():
        pb = _permuted_batch(arr, order, len(order))
        ctx.validate(pb)
        for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ):
            ctx.reset()
            ctx.set_mode(mode)
            ctx.accumulate(pb)
            got = ctx.finalize()
            if len(order) == n:
                assert util.rel_err(got, want) < 1e-11, (name, mode)
    ctx.set_mode(hc.MODE_NODE_WEIGHTS)


def test_full_size_graph_properties():
    """hcfiles-shaped graph (11821 nodes / 5179 paths), 20k reads of 150 bp: the three device modes agree, the
    accumulation is linear and order independent, device-resident batches equal host batches, and a read subset
    matches the oracle."""
    import torch
    g = hc.synth_graph(seed=0x76676131)
    a = hc.synth_reads(g, 20000, seed=0x76676131, read_len=150)
    b = hc.HostBatch(g, a)
    assert b.stats.n_bad == 0
    ctx = hc.HcContext(g)
    out = {}
    for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ, hc.MODE_PER_READ_DENSE):
        ctx.reset()
        ctx.set_mode(mode)
        ctx.accumulate(b)
        out[mode] = ctx.finalize()
    assert util.rel_err(out[hc.MODE_PER_READ], out[hc.MODE_NODE_WEIGHTS]) < 1e-10
    assert util.rel_err(out[hc.MODE_PER_READ_DENSE], out[hc.MODE_PER_READ]) < 1e-12
    # linearity: the same batch twice doubles every entry; split batches sum to the whole
    ctx.reset()
    ctx.set_mode(hc.MODE_NODE_WEIGHTS)
    ctx.accumulate(b)
    ctx.accumulate(b)
    assert util.rel_err(ctx.finalize(), 2 * out[hc.MODE_NODE_WEIGHTS]) < 1e-12
    ctx.reset()
    for r0 in range(0, 20000, 6000):
        ctx.accumulate(hc.HostBatch(g, a, r0, min(20000, r0 + 6000)))
    assert util.rel_err(ctx.finalize(), out[hc.MODE_NODE_WEIGHTS]) < 1e-11
    # device-resident batch on torch's stream
    db = hc.DeviceBatch(b, "cuda:0")
    ctx.use_torch_stream()
    ctx.reset()
    ctx.accumulate(db)
    dev_out = torch.zeros(g.n_paths, dtype=torch.float64, device="cuda:0")
    got = ctx.finalize(dev_out)
    assert util.rel_err(got, out[hc.MODE_NODE_WEIGHTS]) < 1e-12
    assert np.array_equal(dev_out.cpu().numpy(), got)
    ctx.set_stream(None)
    # oracle on the first 150 reads (hoisted variant, validated against the literal loops in the small tests)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    _, ref, bad = orc.hc_run(og, oa, r0=0, r1=150, n_threads=8, faithful=False)
    ctx.reset()
    ctx.accumulate(hc.HostBatch(g, a, 0, 150))
    assert util.rel_err(ctx.finalize(), ref) < RTOL
    # and the literal reference loops on 4 reads
    _, ref4, _ = orc.hc_run(og, oa, r0=0, r1=4, n_threads=4, faithful=True)
    ctx.reset()
    ctx.set_mode(hc.MODE_PER_READ)
    ctx.accumulate(hc.HostBatch(g, a, 0, 4))
    assert util.rel_err(ctx.finalize(), ref4) < RTOL


def test_cli_end_to_end(tmp_path):
    """vgan haplocart -g ... prints the reference's result lines (HaploCart.cpp:437-438, get_posterior.cpp:9-11)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = hc.synth_graph(seed=5, genome_len=800, n_nodes=560, n_paths=40)
    a = hc.synth_reads(g, 300, seed=1, read_len=100)
    g.write(str(tmp_path))
    a.write_gam(str(tmp_path / "r.gam"))
    out = str(tmp_path / "out.tsv")
    pf = str(tmp_path / "post.txt")
    r = subprocess.run([os.path.join(root, "vgan_amd", "bin", "vgan"), "haplocart", "-g", str(tmp_path / "r.gam"),
                        "--hc-files", str(tmp_path), "-q", "-o", out, "-pf", pf, "-s", "my sample"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    kept = a.without(a.mark_duplicates())
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(kept)
    fld, ref, _ = orc.hc_run(og, oa, n_threads=4, faithful=False)
    pred = g.path_names[int(np.argmax(ref))]
    lines = open(out).read().splitlines()
    assert lines[0] == "#sample\tpredicted haplogroup\treads"
    assert lines[1] == "my_sample\t%s\t%d" % (pred, kept.n_reads)
    post = open(pf).read()
    assert post.startswith("\nClade-level posterior confidence values\nmy_sample\t")
    fields = post.split("\n")[2].split("\t")
    exp = orc.hc_posterior(fld, g.path_names, g.parents_txt, g.children_txt, pred)
    assert fields[1] == exp[0][0] and float(fields[2]) == pytest.approx(exp[0][1], rel=1e-5)
    assert len(fields) // 3 == len(exp)


def test_cli_reads_the_reference_odgi_fixture(tmp_path):
    """`vgan haplocart --hc-files DIR` with only DIR/graph.og (the reference's own layout): the ODGI file of
    test/reconstructInputSeq gives the same prediction, read count and posterior line as the GFA of the same graph."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "tests", "golden", "reconstruct")
    outs = {}
    for kind in ("gfa", "og"):
        hcdir = tmp_path / kind
        hcdir.mkdir()
        shutil.copy(os.path.join(d, "target_graph." + kind), str(hcdir / ("graph." + kind)))
        # the same path order for both (the .og lists its paths in handle order)
        (hcdir / "graph_paths").write_text("seq_1\nseq_2\nseq_3\nseq_4\nseq_5\n")
        out, pf = str(tmp_path / (kind + ".tsv")), str(tmp_path / (kind + ".post"))
        r = subprocess.run([os.path.join(root, "vgan_amd", "bin", "vgan"), "haplocart", "-g", os.path.join(d, "test_reads.gam"),
                            "--hc-files", str(hcdir), "-q", "-np", "-o", out, "-pf", pf, "-s", "kat", "-d"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs[kind] = (open(out).read(), open(out + ".loglik.tsv").read())
    assert outs["gfa"][0] == outs["og"][0] and outs["gfa"][0].splitlines()[1].startswith("kat\tseq_")

    def table(txt):
        return {ln.split("\t")[0]: float(ln.split("\t")[1]) for ln in txt.splitlines()}
    a, b = table(outs["gfa"][1]), table(outs["og"][1])
    assert a.keys() == b.keys() and len(a) == 5
    assert all(b[k] == pytest.approx(a[k], rel=1e-12) for k in a)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_tiled_kernel_fuzz_against_general_kernel(seed):
    """Random batches that satisfy the tile contract, built by hand over the C-ABI (no flatten step): ragged reads of
    1..1280 columns, quality strings shorter / longer than the read or empty, segments that stop before the read ends,
    node ids all over the graph (far outside any LDS window), zero-length segments and reads without any segment behind n_tileable, bases outside ACGT, qualities >= 90 and negative.  D_m from the LDS-tiled kernel must equal
    S_m - U_m from the general kernel (which the other tests hold against the oracle)."""
    rng = np.random.default_rng(seed)
    g = hc.synth_graph(seed=50 + seed, genome_len=1200, n_nodes=700, n_paths=70)
    n_nodes = g.max_id
    R = 3000 + 17 * seed
    R_tile = R - 250  # the last reads may hold empty segments and take the general kernel in both batches
    seg_off, col_off, qual_off = [0], [0], [0]
    algn_len, mapq, seg_node, seg_start, seg_len = [], [], [], [], []
    gseq, rseq, qual = [], [], []
    alphabet = np.frombuffer(b"ACGT" * 12 + b"NacgtS-RY", np.uint8)
    for r in range(R):
        if seed >= 4:  # short reads: the 24-reads-per-tile variant of the kernel (mean read below 110 columns)
            cols = int(rng.integers(1, 90)) if r % 50 else int(rng.choice([1, 300, 1280]))
        else:
            cols = int(rng.integers(1, 257)) if r % 7 else int(rng.choice([1, 2, 255, 256, 700, 1279, 1280]))
        A = cols
        ql = int(rng.choice([cols, cols, cols, max(0, cols - int(rng.integers(1, 9))), min(1280, cols + 5), 0]))
        if r >= R_tile and r % 9 == 5:  # degenerate reads (outside the tile contract): no columns at all, or columns that no mapping covers
            cols = 0 if r % 2 else cols
            A, ql = cols, min(ql, cols)
            seg_off.append(len(seg_node))
            gseq.append(rng.choice(alphabet, cols))
            rseq.append(rng.choice(alphabet, cols))
            qual.append(rng.integers(0, 42, ql).astype(np.uint8))
            col_off.append(col_off[-1] + cols)
            qual_off.append(qual_off[-1] + ql)
            algn_len.append(A)
            mapq.append(int(rng.integers(0, 61)))
            continue
        pos = 0
        nseg = 0
        max_run = int(rng.choice([3, 8, 40, 200]))
        stop_early = rng.random() < 0.2
        while pos < cols and nseg < 512:
            if r >= R_tile and rng.random() < 0.05:
                ln = 0  # a mapping without columns: such a read is outside the tile contract
            else:
                ln = int(min(cols - pos, rng.integers(1, max_run + 1)))
            seg_node.append(int(rng.integers(1, n_nodes + 1)))
            seg_start.append(pos)
            seg_len.append(ln)
            pos += ln
            nseg += 1
            if stop_early and pos > cols // 2:
                break
        seg_off.append(len(seg_node))
        gseq.append(rng.choice(alphabet, cols))
        rseq.append(rng.choice(alphabet, cols))
        q = rng.integers(0, 42, ql).astype(np.uint8)
        if ql and rng.random() < 0.15:
            q[rng.integers(0, ql)] = rng.choice([90, 93, 120, 127, 128, 200, 255])
        qual.append(q)
        col_off.append(col_off[-1] + cols)
        qual_off.append(qual_off[-1] + ql)
        algn_len.append(A)
        mapq.append(int(rng.integers(0, 61)))
    arrays = {"read_seg_off": seg_off, "read_col_off": col_off, "read_qual_off": qual_off, "read_algn_len": algn_len,
              "read_mapq": mapq, "seg_node": seg_node, "seg_start": seg_start, "seg_len": seg_len,
              "graph_seq": np.concatenate(gseq), "algnseq": np.concatenate(rseq),
              "qual": np.concatenate(qual) if sum(map(len, qual)) else np.zeros(0, np.uint8)}
    tiled = hc.ArrayBatch(arrays, n_tileable=R_tile)
    general = hc.ArrayBatch(arrays, n_tileable=0)
    for kw in (dict(), dict(background_error_prob=0.02, use_background_error_prob=True),
               dict(background_error_prob=0.01, use_background_error_prob=True, is_consensus_fasta=True)):
        ctx = hc.HcContext(g, **kw)
        S, U = ctx.segment_scalars(general)
        want = S - U
        got = ctx.segment_weights(tiled)
        also = ctx.segment_weights(general)
        fin = np.isfinite(want)
        assert np.array_equal(fin, np.isfinite(got)) and np.array_equal(got[~fin], want[~fin])
        # U_m is a difference of two tile-wide prefix sums (up to 1280 terms of a few units each: ulp ~ 2e-12)
        tol = 5e-12 + 1e-13 * np.maximum(np.abs(S), np.abs(U))
        assert np.all(np.abs(got[fin] - want[fin]) <= tol[fin])
        assert np.all(np.abs(also[fin] - want[fin]) <= tol[fin])
        # and through the accumulators: the same final vector from both routes
        outs = []
        for b in (tiled, general):
            ctx.reset()
            ctx.accumulate(b)
            outs.append(ctx.finalize())
        if np.all(fin):
            assert util.rel_err(outs[0], outs[1]) < 1e-12


def test_accumulate_in_batches_and_after_finalize():
    g = hc.synth_graph(seed=61, genome_len=1300, n_nodes=800, n_paths=90)
    a = hc.synth_reads(g, 5000, seed=62, read_len=110)
    whole = hc.HostBatch(g, a)
    n = a.n_reads
    parts = [hc.HostBatch(g, a, 0, n // 4), hc.HostBatch(g, a, n // 4, n // 2), hc.HostBatch(g, a, n // 2, n)]
    for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ):
        ctx = hc.HcContext(g)
        ctx.set_mode(mode)
        ctx.accumulate(whole)
        want = ctx.finalize()
        ctx.reset()
        ctx.accumulate(parts[0])
        mid = ctx.finalize()
        assert np.all(mid >= want - 1e-9 * np.abs(want))  # fewer reads: every log-likelihood sum is less negative
        ctx.accumulate(parts[1])
        ctx.accumulate(parts[2])
        got = ctx.finalize()
        assert util.rel_err(got, want) < 1e-12
        assert util.rel_err(ctx.finalize(), got) < 1e-13  # finalize does not consume the accumulators (atomic order may differ)


@pytest.mark.parametrize("P", [8128, 8192, 8200, 16384, 20000])
def test_path_counts_around_the_tile_split_boundaries(P):
    """The sweep deals the ceil(P/64) mask words to 8 x ceil(W/127) tiles of at most 16 words: path counts at and
    beyond the points where the number of tiles changes (W = 127, 128, 129, 256 ...), every mode, against the oracle."""
    g = hc.synth_graph(seed=P, genome_len=600, n_nodes=400, n_paths=P)
    a = hc.synth_reads(g, 300, seed=3, read_len=80)
    b = hc.HostBatch(g, a)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    _, ref, _ = orc.hc_run(og, oa, n_threads=8, faithful=False)
    ctx = hc.HcContext(g)
    for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ, hc.MODE_PER_READ_DENSE):
        ctx.reset()
        ctx.set_mode(mode)
        ctx.accumulate(b)
        assert util.rel_err(ctx.finalize(), ref) < RTOL, mode


@pytest.mark.parametrize("P,n_nodes,genome", [(1, 300, 500), (2, 300, 500), (63, 300, 500), (64, 300, 500), (65, 300, 500),
                                              (127, 300, 500), (128, 300, 500), (129, 300, 500), (1024, 300, 500),
                                              (40, 26000, 40000)])
def test_small_path_counts_and_large_node_counts(P, n_nodes, genome):
    """One path, word boundaries of the mask, and more nodes than the LDS-private node accumulator holds (> 19 200:
    the global-atomic accumulate kernel), every mode, against the oracle."""
    g = hc.synth_graph(seed=1000 + P, genome_len=genome, n_nodes=n_nodes, n_paths=P)
    a = hc.synth_reads(g, 400, seed=4, read_len=70)
    b = hc.HostBatch(g, a)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    _, ref, _ = orc.hc_run(og, oa, n_threads=8, faithful=False)
    ctx = hc.HcContext(g)
    for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ, hc.MODE_PER_READ_DENSE):
        ctx.reset()
        ctx.set_mode(mode)
        ctx.accumulate(b)
        got = ctx.finalize()
        assert util.rel_err(got, ref) < RTOL, mode
    # paths with the same support over every read tie exactly in exact arithmetic; which of them wins depends on the last bit
    # (and the device sums in a scheduling-dependent order), so: the chosen path is a maximum of the oracle's vector
    assert ref[ctx.argmax(got)] >= ref.max() - 1e-9 * abs(ref.max())


def test_batch_validation(tmp_path):
    """vgan_hc_batch_validate accepts what vgan_hc_flatten produces (tileable, long and irregular reads) and names the
    broken contract of hand-built batches."""
    from vgan_amd._native import NativeError
    g = hc.synth_graph(seed=71, genome_len=2000, n_nodes=1200, n_paths=40)
    a = util.concat_alnsets(tmp_path, hc.synth_reads(g, 300, seed=1, read_len=120, indel_rate=0.3, softclip_rate=0.3),
                            hc.synth_reads(g, 10, seed=2, read_len=1500))
    ctx = hc.HcContext(g)
    b = hc.HostBatch(g, a)
    ctx.validate(b)
    base = {k: np.array(v) for k, v in b.arrays().items() if k not in ("_owner", "read_src")}

    def broken(change, n_tileable=None):
        arr = {k: v.copy() for k, v in base.items()}
        change(arr)
        bad = hc.ArrayBatch(arr, n_tileable=b.n_tileable if n_tileable is None else n_tileable)
        with pytest.raises(NativeError):
            ctx.validate(bad)

    ctx.validate(hc.ArrayBatch(base, n_tileable=b.n_tileable))
    ctx.validate(hc.ArrayBatch(base, n_tileable=0))
    broken(lambda x: x["seg_node"].__setitem__(5, g.max_id + 1))                      # unknown node
    broken(lambda x: x["read_seg_off"].__setitem__(3, x["read_seg_off"][4] + 1))      # offsets descend
    broken(lambda x: x["seg_len"].__setitem__(0, 60000))                              # leaves the read
    broken(lambda x: x["read_mapq"].__setitem__(7, 120))
    broken(lambda x: x["seg_start"].__setitem__(1, 0) if x["seg_len"][0] else None, n_tileable=b.n_tileable)  # overlap / order
    broken(lambda x: x["seg_len"].__setitem__(2, 0), n_tileable=b.n_tileable)         # an empty segment below n_tileable
    broken(lambda x: x["read_seg_off"].__setitem__(1, x["read_seg_off"][0]), n_tileable=b.n_tileable)  # a read without segments below n_tileable
    broken(lambda x: None, n_tileable=b.n_reads)                                      # the 1500-column reads are not tileable
    broken(lambda x: x["read_col_off"].__setitem__(len(x["read_col_off"]) - 1, 5))    # final offset vs n_cols


def test_cli_reads_its_gam_from_a_pipe(tmp_path):
    """`vgan haplocart -g /dev/stdin`: the streamed reader works on a pipe (the reference hands its reader a FIFO)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = hc.synth_graph(seed=5, genome_len=800, n_nodes=560, n_paths=40)
    a = hc.synth_reads(g, 5000, seed=1, read_len=100)
    g.write(str(tmp_path))
    a.write_gam(str(tmp_path / "r.gam"))
    exe = os.path.join(root, "vgan_amd", "bin", "vgan")
    outs = []
    for src, kw in ((str(tmp_path / "r.gam"), {}), ("/dev/stdin", {"input": open(str(tmp_path / "r.gam"), "rb").read()})):
        out = str(tmp_path / ("o%d.tsv" % len(outs)))
        r = subprocess.run([exe, "haplocart", "-g", src, "--hc-files", str(tmp_path), "-q", "-np", "-o", out, "-s", "x", "-t", "-1"],
                           capture_output=True, **kw)
        assert r.returncode == 0, r.stderr[-500:]
        outs.append(open(out).read())
    assert outs[0] == outs[1] and outs[0].splitlines()[1].startswith("x\thg")


def test_cli_chunk_loop_lanes_and_small_batches_give_the_same_log_likelihoods(tmp_path):
    """The chunk loop of `vgan haplocart` (csrc/host/vgan_main.cpp): chunks taken and marked for duplicates in input order,
    flattened on several lanes side by side, queued for the device in order again.  Many small chunks on three lanes, with
    duplicate removal, against one lane and one chunk: the same reads kept, the same per-haplotype log-likelihoods (-d writes
    them) to rounding."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = hc.synth_graph(seed=15, genome_len=5000, n_nodes=3400, n_paths=60)
    a = hc.synth_reads(g, 30000, seed=3, read_len=100, indel_rate=0.05, softclip_rate=0.05, low_mapq_rate=0.2)
    g.write(str(tmp_path))
    a.write_gam(str(tmp_path / "r.gam"))
    res = {}
    # (the default route flattens on the device and leaves the reads with indels / soft clips to the host; VGAN_HC_HOST_FLATTEN=1
    # sends every read through the host flatten, on `lanes` threads)
    for tag, env in (("one", {"VGAN_HC_LANES": "1", "VGAN_HC_BATCH": "1000000", "VGAN_HC_HOST_FLATTEN": "1"}),
                     ("lanes", {"VGAN_HC_LANES": "3", "VGAN_HC_BATCH": "1024", "VGAN_HC_QUEUE": "3", "VGAN_GAM_SEG_BLOCKS": "4", "VGAN_HC_HOST_FLATTEN": "1"}),
                     ("device", {"VGAN_HC_BATCH": "1000000", "VGAN_HC_DEVICE_AFTER": "0"}),
                     ("device_small", {"VGAN_HC_BATCH": "1024", "VGAN_HC_QUEUE": "3", "VGAN_GAM_SEG_BLOCKS": "4", "VGAN_TIMING": "1", "VGAN_HC_DEVICE_AFTER": "2"}),
                     # the whole front end on the device (csrc/gam_kernels.hip: inflate, framing, protobuf walk, duplicate marks, flatten);
                     # the reads its flatten leaves come back as messages and take the host's parser and flatten
                     ("device_gam", {"VGAN_HC_DEVICE_GAM": "1", "VGAN_TIMING": "1"})):
        out = str(tmp_path / (tag + ".tsv"))
        r = subprocess.run([os.path.join(root, "vgan_amd", "bin", "vgan"), "haplocart", "-g", str(tmp_path / "r.gam"), "--hc-files",
                            str(tmp_path), "-q", "-np", "-d", "-o", out, "-s", "s", "-t", "6"],
                           capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        if tag == "device_small":  # some reads were the device's, some the host's
            line = [ln for ln in r.stderr.splitlines() if "device flatten" in ln][0]
            n_host = int(line.split("host flatten of the ")[1].split()[0])
            assert 0 < n_host < 30000, line
        if tag == "device_gam":
            line = [ln for ln in r.stderr.splitlines() if "device front end" in ln]
            assert line and "30000 messages" in line[0] and 0 < int(line[0].split(" left to the")[0].split()[-1]) < 30000, r.stderr[-1500:]
        ll = dict((ln.split("\t")[0], float(ln.split("\t")[1])) for ln in open(out + ".loglik.tsv").read().splitlines())
        res[tag] = (open(out).read().splitlines()[1], ll)
    for other in ("lanes", "device", "device_small", "device_gam"):
        assert res["one"][0] == res[other][0]  # sample, predicted haplogroup, reads kept
        assert res["one"][1].keys() == res[other][1].keys() and len(res["one"][1]) == 60
        for k, v in res["one"][1].items():
            assert res[other][1][k] == pytest.approx(v, rel=1e-9)
