"""GPU tests of the packed batch route (vgan_hc_flatten*_packed -> vgan_hc_accumulate_packed): the layout the host flatten
step writes is held word for word against what the device's layout pass (vgan_hc_pack) makes of the SoA batch of the same
reads, and the route's sums against the oracle in all three modes -- host arrays, arrays resident in HBM, and batches that
carry reads outside the tile contract beside the packed ones."""
import numpy as np
import pytest

import orc
import util
from vgan_amd import _native as N
from vgan_amd import haplocart as hc

pytestmark = pytest.mark.gpu


def long_read_mix(g, n_short, n_long, seed):
    """Reads of 150 columns and a few of 2000 (beyond the tile contract's 1280 columns: they stay in the SoA part)."""
    return [hc.synth_reads(g, n_short, seed=seed, read_len=150, indel_rate=0.03, softclip_rate=0.05, low_mapq_rate=0.3),
            hc.synth_reads(g, n_long, seed=seed + 1, read_len=2000)]


@pytest.mark.parametrize("read_len", [40, 75, 150, 300, 600])
def test_host_packed_layout_equals_the_device_layout_pass(read_len):
    g = hc.synth_graph(seed=61, genome_len=4000, n_nodes=2700, n_paths=120)
    a = hc.synth_reads(g, 6000, seed=62 + read_len, read_len=read_len, indel_rate=0.05, softclip_rate=0.1, low_mapq_rate=0.3)
    soa = hc.HostBatch(g, a)
    pk = hc.HostBatch(g, a, packed=True)
    assert pk.pk.n_reads == soa.n_tileable and pk.n_reads == soa.n_reads and pk.n_segments == soa.n_segments
    ctx = hc.HcContext(g)
    ctx.validate_packed(pk)
    dev = hc.DeviceBatch(soa, ctx=ctx).download_packed()  # hc_pack_kernel's output
    host = pk.packed_arrays()
    for name in ("rhdr", "srec", "crec", "qualp"):
        assert np.array_equal(dev[name], host[name]), name


def test_packed_route_against_the_oracle_in_all_modes(tmp_path):
    g = hc.synth_graph(seed=71, genome_len=5000, n_nodes=3400, n_paths=260)
    a = util.concat_alnsets(tmp_path, *long_read_mix(g, 6000, 12, 72))
    hb = hc.HostBatch(g, a, packed=True)
    assert hb.pk.n_reads > 0 and hb.c.n_reads >= 4 and hb.n_reads == a.n_reads
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    _, ref, _ = orc.hc_run(og, oa, n_threads=8, faithful=False)
    for kw in (dict(), dict(background_error_prob=0.02, use_background_error_prob=True)):
        ctx = hc.HcContext(g, **kw)
        want = orc.hc_run(og, oa, orc.hc_params(0.02, True), n_threads=8, faithful=False)[1] if kw else ref
        db = hc.DeviceBatch(hb)
        assert db.pk.on_device == 1
        for batch in (hb, db):
            for mode in (hc.MODE_NODE_WEIGHTS, hc.MODE_PER_READ, hc.MODE_PER_READ_DENSE):
                ctx.reset()
                ctx.set_mode(mode)
                ctx.accumulate(batch)
                assert util.rel_err(ctx.finalize(), want) < 1e-9, (kw, mode)


def test_packed_segment_weights_equal_the_soa_route_per_segment():
    g = hc.synth_graph(seed=81, genome_len=4000, n_nodes=2600, n_paths=90)
    a = hc.synth_reads(g, 8000, seed=82, read_len=150, low_mapq_rate=0.4)
    soa = hc.HostBatch(g, a)
    pk = hc.HostBatch(g, a, packed=True)
    ctx = hc.HcContext(g)
    S, U = ctx.segment_scalars(soa)  # the general kernel's S_m, U_m
    D = ctx.segment_weights_packed(pk)
    n = pk.pk.n_segments
    assert n == soa.n_segments  # (every read tileable, both batches in the same order)
    assert np.all(np.abs(D - (S - U)[:n]) <= 5e-12 + 1e-13 * np.maximum(np.abs(S), np.abs(U))[:n])


def test_a_broken_packed_batch_is_refused():
    g = hc.synth_graph(seed=91, genome_len=2000, n_nodes=1300, n_paths=40)
    a = hc.synth_reads(g, 400, seed=92, read_len=100)
    hb = hc.HostBatch(g, a, packed=True)
    ctx = hc.HcContext(g)
    ctx.validate_packed(hb)
    keep = hb.pk.max_read_segs
    hb.pk.max_read_segs = 0  # maxima missing: the kernel variant cannot be chosen
    with pytest.raises(N.NativeError):
        ctx.accumulate(hb)
    hb.pk.max_read_segs = 1  # understated
    with pytest.raises(N.NativeError):
        ctx.validate_packed(hb)
    hb.pk.max_read_segs = keep
    arr = hb.packed_arrays()
    node = int(arr["srec"][0])
    arr["srec"][0] = g.max_id + 5
    with pytest.raises(N.NativeError):
        ctx.validate_packed(hb)
    arr["srec"][0] = node
    ctx.validate_packed(hb)
    ctx.accumulate(hb)
    assert np.all(np.isfinite(ctx.finalize()))
    # long reads stay outside the packed part (the general kernel takes them); an empty batch is fine
    g2 = hc.synth_graph(seed=94, genome_len=9000, n_nodes=6000, n_paths=40)
    long_reads = hc.HostBatch(g2, hc.synth_reads(g2, 5, seed=93, read_len=3000, softclip_rate=0.0, indel_rate=0.0), packed=True)
    assert long_reads.c.n_reads >= 3 and long_reads.n_reads == 5
    ctx2 = hc.HcContext(g2)
    ctx2.accumulate(long_reads)
    ctx2.accumulate(hc.HostBatch(g2, hc.synth_reads(g2, 5, seed=95, read_len=100), 0, 0, packed=True))
    assert np.all(np.isfinite(ctx2.finalize()))
