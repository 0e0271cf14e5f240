"""GPU tests of a1 on the device (vgan_hc_devflat: reconstruct_graph_sequence + slicing + packed layout in one pass over the
parser's arrays).  Integer and byte work: what the device writes for the reads it takes must be the host flatten's packed
batch of those reads, every rhdr / srec / crec / qualp word -- on synthetic sets with reverse strands, substitutions,
indels and soft clips (the host's reads), on the reference's reconstruction KAT alignments and the bundled GAMs, and on a
1M-read set; and the two routes must add up to the same likelihood vector."""
import os

import numpy as np
import pytest

import util
from vgan_amd import haplocart as hc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def compare(g, parts, ctx, df, skip=None, want_host=None):
    res = df.run(parts, skip=skip)
    mask = res.host_mask.astype(bool)
    sk = mask if skip is None else (mask | np.asarray(skip, bool))
    host = hc.HostBatch(g, parts, packed=True, skip=sk.astype(np.uint8))  # the host's layout of exactly the reads the device took
    assert host.c.n_reads == 0, "the device took a read the host keeps out of the packed part"
    assert host.pk.n_reads == res.pk.n_reads and host.pk.n_segments == res.pk.n_segments
    assert (host.pk.n_cols, host.pk.n_qual) == (res.pk.n_cols, res.pk.n_qual)
    if res.pk.n_reads:
        assert (host.pk.max_read_segs, host.pk.max_read_qual, host.pk.max_read_cols, host.pk.max_read_node_span) == \
            (res.pk.max_read_segs, res.pk.max_read_qual, res.pk.max_read_cols, res.pk.max_read_node_span)
        got, want = res.download(), host.packed_arrays()
        for name in ("rhdr", "srec", "crec", "qualp", "read_src"):
            assert np.array_equal(got[name], want[name]), name
        ctx.validate_packed(host)
    # the reads left to the host are the ones its one-walk form or the tile contract does not cover -- nothing usable is lost
    everything = hc.HostBatch(g, parts, packed=True, skip=None if skip is None else np.asarray(skip, np.uint8))
    rest = hc.HostBatch(g, parts, packed=True, skip=(~mask if skip is None else (~mask | np.asarray(skip, bool))).astype(np.uint8))
    assert res.pk.n_reads + rest.n_reads == everything.n_reads
    assert res.stats.n_in == everything.stats.n_in and res.stats.n_unmapped == everything.stats.n_unmapped
    if want_host is not None:
        assert want_host(int(mask.sum()), parts.n_reads)
    # the two routes add up to the same vector
    ctx.reset()
    ctx.accumulate(everything)
    ref = ctx.finalize()
    ctx.reset()
    if res.pk.n_reads:
        ctx.accumulate(res)
    if rest.n_reads:
        ctx.accumulate(rest)
    assert util.rel_err(ctx.finalize(), ref) < 1e-12
    return res, mask


def parts_of(tmp_path, alns, name="x.gam"):
    f = str(tmp_path / name)
    alns.write_gam(f)
    return hc.AlnParts.read_gam(f)


@pytest.mark.parametrize("read_len, indel, clip", [(150, 0.0, 0.0), (100, 0.05, 0.1), (40, 0.3, 0.2), (600, 0.01, 0.0), (2000, 0.0, 0.0)])
def test_device_flatten_equals_the_host_flatten_word_for_word(tmp_path, read_len, indel, clip):
    g = hc.synth_graph(seed=61, genome_len=6000, n_nodes=4100, n_paths=90)
    a = hc.synth_reads(g, 20000 if read_len < 1000 else 300, seed=62 + read_len, read_len=read_len, indel_rate=indel, softclip_rate=clip, low_mapq_rate=0.3)
    parts = parts_of(tmp_path, a)
    ctx = hc.HcContext(g)
    df = hc.DeviceFlatten(ctx, g)
    pure = indel == 0.0 and clip == 0.0
    res, mask = compare(g, parts, ctx, df,
                        want_host=(lambda nh, n: nh == 0) if pure and read_len <= 1280 else (lambda nh, n: nh > 0))
    if read_len > 1280:
        assert mask.sum() >= parts.n_reads * 0.8  # beyond the tile contract: the general kernel's reads
    # with duplicate marks, and a second chunk through the same object
    dup = parts.mark_duplicates()
    compare(g, parts, ctx, df, skip=dup)


def test_device_flatten_on_the_reference_alignments(tmp_path):
    """The alignments of the reference's reconstruction KATs (src/test.cpp:855-994: test_reads.gam on target_graph) and its
    bundled GAMs: whatever the device takes equals the host's layout, the rest is flagged."""
    d = os.path.join(GOLD, "reconstruct")
    g = hc.Graph.load(os.path.join(d, "target_graph.gfa"))
    # (no hcfiles sidecars for this graph: every node gets a pangenome position so that the reads are usable)
    g2 = hc.Graph.from_arrays(g.min_id, g.max_id, g.node_seq_off, g.node_seq.tobytes(), max(g.n_paths, 1), g.mask,
                              np.arange(g.max_id + 1, dtype=np.int32), np.ones(g.max_id + 2), path_names="\n".join(g.path_names))
    ctx = hc.HcContext(g2)
    df = hc.DeviceFlatten(ctx, g2)
    taken = 0
    for f in ["reconstruct/test_reads.gam", "alignments/J2a1a1a1.gam", "alignments/all_the_same.gam", "alignments/all_the_same_reverse.gam",
              "alignments/two_unique.gam"]:
        parts = hc.AlnParts.read_gam(os.path.join(GOLD, f))
        res, mask = compare(g2, parts, ctx, df)
        taken += res.pk.n_reads
    # the pyref fixture's reads (reverse strands, indels, soft clips, short quality strings, mapq 0) on its own graph
    fx = os.path.join(GOLD, "hc_pyref")
    g3 = hc.Graph.load(os.path.join(fx, "graph.gfa"), fx)
    ctx3 = hc.HcContext(g3)
    res, mask = compare(g3, hc.AlnParts.read_gam(os.path.join(fx, "reads.gam")), ctx3, hc.DeviceFlatten(ctx3, g3))
    assert res.pk.n_reads > 50 and mask.sum() > 10
    assert taken >= 0


def test_device_flatten_of_a_million_reads(tmp_path):
    g = hc.synth_graph(seed=1)
    a = hc.synth_reads(g, 1000000, seed=2, read_len=150)
    parts = parts_of(tmp_path, a, "m.gam")
    ctx = hc.HcContext(g)
    df = hc.DeviceFlatten(ctx, g)
    res = df.run(parts)
    mask = res.host_mask.astype(bool)
    host = hc.HostBatch(g, parts, packed=True, skip=mask.astype(np.uint8))
    assert host.c.n_reads == 0 and host.pk.n_reads == res.pk.n_reads > 900000
    got, want = res.download(), host.packed_arrays()
    for name in ("rhdr", "srec", "crec", "qualp", "read_src"):
        assert np.array_equal(got[name], want[name]), name
