"""GPU tests of hc_segment_col8_kernel (hc_col8_kernels.hip: eight columns to a lane, tables of column terms), through the C-ABI.
Every accumulate of a packed batch in the node-weights mode goes through it where the batch and the graph fit, so the oracle
holds it end to end in test_hc_gpu.py / test_pyref_gpu.py; here its three variants, its three kinds of tile (the workgroup's
table, the context's table, every column computed) and its routing are forced and held against the oracle and against the wave
kernel on the same batch."""
import os

import numpy as np
import pytest

import orc
import util
from vgan_amd import haplocart as hc

pytestmark = pytest.mark.gpu


@pytest.fixture
def kernel_switch():
    old = os.environ.get("VGAN_HC_KERNEL")

    def use(which):
        if which is None:
            os.environ.pop("VGAN_HC_KERNEL", None)
        else:
            os.environ["VGAN_HC_KERNEL"] = which
    yield use
    use(old)


def finals(ctx, batch, use):
    out = {}
    for which in (None, "wave"):
        use(which)
        ctx.reset()
        ctx.accumulate(batch)
        out[which] = ctx.finalize()
    use(None)
    return out[None], out["wave"]


@pytest.mark.parametrize("read_len,low_mapq", [(40, 0.3), (75, 0.1), (150, 0.1), (150, 1.0), (300, 0.3), (600, 0.2), (1100, 0.1)])
def test_the_three_variants_against_the_oracle_and_the_wave_kernel(read_len, low_mapq, kernel_switch):
    """40-150 bp reads take the variant of eight columns a lane, 300 bp sixteen, 600 bp and beyond twenty-four; low_mapq = 1: no
    read of the major mapping quality, every tile through the context's table."""
    kernel_switch(None)
    g = hc.synth_graph(seed=21, genome_len=6000, n_nodes=4000, n_paths=200)
    a = hc.synth_reads(g, max(300, 240000 // read_len), seed=22 + read_len, read_len=read_len, indel_rate=0.05, softclip_rate=0.1, low_mapq_rate=low_mapq)
    hb = hc.HostBatch(g, a, packed=True)
    assert hb.pk.n_reads > 0.7 * hb.n_reads
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    _, ref, _ = orc.hc_run(og, oa, n_threads=8, faithful=False)
    for kw in (dict(), dict(background_error_prob=0.02, use_background_error_prob=True)):
        ctx = hc.HcContext(g, **kw)
        got, wave = finals(ctx, hb, kernel_switch)
        assert util.rel_err(got, wave) < 1e-12
        if not kw:
            assert util.rel_err(got, ref) < 1e-9
        db = hc.DeviceBatch(hb)  # the resident batch gives the same sums
        ctx.reset()
        ctx.accumulate(db)
        assert util.rel_err(ctx.finalize(), got) < 1e-13


def _edit_qualities(hb, fn):
    """The packed arrays with the quality bytes of the column records (byte 2) and of qualp rewritten by fn(read index, bytes)."""
    pa = hb.packed_arrays()
    h = pa["rhdr"].reshape(-1, 4)
    for r in range(hb.pk.n_reads):
        q0, q1, c0 = int(h[r, 1]), int(h[r + 1, 1]), int(h[r, 2])
        q = fn(r, pa["qualp"][q0:q1].copy())
        pa["qualp"][q0:q1] = q
        rec = pa["crec"][c0:c0 + (q1 - q0)]
        pa["crec"][c0:c0 + (q1 - q0)] = (rec & np.uint32(0xFF00FFFF)) | (q.astype(np.uint32) << 16)


def test_tiles_outside_the_tables_compute_every_column(kernel_switch):
    """Quality bytes beyond the tables' range (48 and up), negative as signed chars, and the sticky Q >= 90 of
    update_likelihood.cpp:40-44: such tiles leave the tables; the sums must not notice."""
    kernel_switch(None)
    g = hc.synth_graph(seed=71, genome_len=5000, n_nodes=3300, n_paths=150)
    a = hc.synth_reads(g, 6000, seed=72, read_len=150, low_mapq_rate=0.2)
    hb = hc.HostBatch(g, a, packed=True)
    rng = np.random.default_rng(3)

    def edit(r, q):
        if r % 5 == 0 and len(q):
            k = rng.integers(0, len(q), 3)
            q[k] = rng.choice([48, 60, 89, 90, 93, 127, 128, 200, 255], 3)
        return q
    _edit_qualities(hb, edit)
    ctx = hc.HcContext(g)
    ctx.validate_packed(hb)
    got, wave = finals(ctx, hb, kernel_switch)
    assert np.all(np.isfinite(got)) and util.rel_err(got, wave) < 1e-12
    # per segment through the wave kernel against the general kernel on the SoA form of the same (edited) reads is held elsewhere;
    # here the per-read mode of the same packed batch (the wave kernel + the mask sweep) must land on the same vector
    ctx.set_mode(hc.MODE_PER_READ)
    ctx.reset()
    ctx.accumulate(hb)
    assert util.rel_err(ctx.finalize(), got) < 1e-11


def test_graphs_with_many_node_classes(kernel_switch):
    """More node classes than the workgroup's table covers (16: the tiles with the others read the context's wide table), more than
    the kernel keeps the scalars of in LDS (32), as many as it takes at all (256 values of mappability x the mutation rates: beyond 256
    classes the wave kernel takes the batch): a mappability track with many distinct values.  With quality bytes from 48 to 89 on a
    third of the reads as well (the wide table's other axis)."""
    kernel_switch(None)
    g0 = hc.synth_graph(seed=81, genome_len=5000, n_nodes=3300, n_paths=120)
    rng = np.random.default_rng(8)
    for n_values in (12, 60, 256):
        mp = np.array(g0.mappability).copy()
        vals = np.round(rng.uniform(0.3, 1.0, n_values), 3)
        for w in range(0, len(mp), 40):
            if rng.random() < 0.5:
                mp[w:w + 40] = rng.choice(vals)
        g = hc.Graph.from_arrays(g0.min_id, g0.max_id, np.array(g0.node_seq_off), np.array(g0.node_seq).tobytes(), g0.n_paths, np.array(g0.mask),
                                 np.array(g0.pangenome_base), mp, "\n".join(g0.path_names) + "\n", g0.parents_txt, g0.children_txt)
        a = hc.synth_reads(g, 5000, seed=82, read_len=150)
        hb = hc.HostBatch(g, a, packed=True)
        ctx = hc.HcContext(g)
        og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
        _, ref, _ = orc.hc_run(og, oa, n_threads=8, faithful=False)
        got, wave = finals(ctx, hb, kernel_switch)
        assert util.rel_err(got, ref) < 1e-9 and util.rel_err(got, wave) < 1e-12

        def edit(r, q):
            if r % 3 == 0 and len(q):
                q[:] = rng.integers(48, 90, len(q))
            return q
        _edit_qualities(hb, edit)
        got2, wave2 = finals(ctx, hb, kernel_switch)
        assert np.all(np.isfinite(got2)) and util.rel_err(got2, wave2) < 1e-12 and util.rel_err(got2, got) > 1e-6


def test_soa_batches_whose_quality_strings_outrun_their_columns_keep_to_the_other_kernels():
    """The SoA tile contract lets a quality string be longer than the read's columns; the column records cannot hold such a
    string, so a batch packed from such arrays must not take the kernel that reads the quality bytes there."""
    g = hc.synth_graph(seed=91, genome_len=3000, n_nodes=2000, n_paths=90)
    a = hc.synth_reads(g, 3000, seed=92, read_len=120)
    hb = hc.HostBatch(g, a)
    arr = {k: np.array(v) for k, v in hb.arrays().items() if not k.startswith("_") and k != "read_src"}
    # every fourth read gets five more quality bytes than it has columns
    qo = arr["read_qual_off"].astype(np.int64)
    quals, new_off = [], [0]
    for r in range(hb.n_reads):
        q = arr["qual"][qo[r]:qo[r + 1]]
        if r % 4 == 0:
            q = np.concatenate([q, np.full(5, 33, np.uint8)])
        quals.append(q)
        new_off.append(new_off[-1] + len(q))
    arr["qual"] = np.concatenate(quals)
    arr["read_qual_off"] = np.array(new_off, np.uint32)
    ctx = hc.HcContext(g)
    outs = []
    for nt in (hb.n_tileable, 0):  # the routed kernels, and the general kernel on every read
        ctx.reset()
        ctx.accumulate(hc.ArrayBatch(arr, n_tileable=nt))
        outs.append(ctx.finalize())
    assert util.rel_err(outs[0], outs[1]) < 1e-12


def test_profile_events_around_every_kernel_or_the_segment_kernel_alone():
    """vgan_hc_profile_enable: 1 times every kernel of a step, 2 the segment kernel alone (what bench.py's timed steps ask for); the
    sums come out the same with either."""
    g = hc.synth_graph(seed=41, genome_len=4000, n_nodes=2600, n_paths=200)
    a = hc.synth_reads(g, 20000, seed=42, read_len=150)
    hb = hc.HostBatch(g, a, packed=True)
    ctx = hc.HcContext(g)
    got = {}
    for mode, kw in ((1, {}), (2, {"segment_only": True})):
        ctx.reset()
        ctx.profile_enable(True, **kw)
        for _ in range(3):
            ctx.accumulate(hb)
        got[mode] = ctx.finalize()
        pr = ctx.profile_read()
        ctx.profile_enable(False)
        assert pr["segment"][1] == 3 and pr["segment"][0] > 0
        assert (pr["sweep_nodes"][1], pr["finish"][1]) == ((1, 1) if mode == 1 else (0, 0))
    assert np.max(np.abs(got[1] - got[2]) / np.abs(got[1])) < 1e-13
