"""CPU tests of the packed flatten (vgan_hc_flatten_packed / _parts_packed): the layout the host writes for the segment kernel
is rebuilt here, in numpy, from the SoA batch of the same reads -- every rhdr / srec / crec / qualp word must agree -- and
the reads outside the tile contract must come out as a batch of their own."""
import numpy as np

import util
from vgan_amd import haplocart as hc


def pack_of_soa(arr, nt):
    """The packed arrays of reads [0, nt) of a SoA batch, as include/vgan_gpu.h defines them (vgan_hc_packed_view)."""
    so, co, qo = (arr[k].astype(np.int64) for k in ("read_seg_off", "read_col_off", "read_qual_off"))
    S, Cn, Q = int(so[nt]), int(co[nt]), int(qo[nt])
    rhdr = np.zeros((nt + 1, 4), np.uint32)
    rhdr[:, 0], rhdr[:, 1], rhdr[:, 2] = so[:nt + 1], qo[:nt + 1], co[:nt + 1]
    rhdr[:nt, 3] = arr["read_algn_len"][:nt].astype(np.uint32) | (arr["read_mapq"][:nt].astype(np.uint32) << 16)
    ridx = np.repeat(np.arange(nt, dtype=np.uint32), np.diff(so[:nt + 1]))
    # (ABI 5) one word a mapping: node id | seg_start << 18 | (read index & 7) << 29
    srec = arr["seg_node"][:S].astype(np.uint32) | (arr["seg_start"][:S].astype(np.uint32) << 18) | ((ridx & 7) << 29)
    # per column: the segment that scores it (tileable reads: ranges do not overlap)
    seg_first = co[ridx] + arr["seg_start"][:S].astype(np.int64)
    ln = arr["seg_len"][:S].astype(np.int64)
    col = np.repeat(seg_first, ln) + (np.arange(ln.sum()) - np.repeat(np.cumsum(ln) - ln, ln))
    j = np.arange(ln.sum()) - np.repeat(np.cumsum(ln) - ln, ln)       # position within the segment
    cr = np.repeat(ridx.astype(np.int64), ln)                           # the column's read
    c_in_read = col - co[cr]
    qlen = (qo[1:nt + 1] - qo[:nt])[cr]
    qb = np.where(c_in_read < qlen, arr["qual"][np.minimum(qo[cr] + c_in_read, max(len(arr["qual"]) - 1, 0))], 0).astype(np.uint32)
    # (ABI 5) the quality byte sits on EVERY column below the string's length, scored or not; the head flag is byte 3 = 4
    allc = np.arange(Cn, dtype=np.int64)
    rd_of = np.repeat(np.arange(nt, dtype=np.int64), np.diff(co[:nt + 1]))
    cin = allc - co[rd_of]
    ql_all = (qo[1:nt + 1] - qo[:nt])[rd_of]
    crec = (np.where(cin < ql_all, arr["qual"][np.minimum(qo[rd_of] + cin, max(len(arr["qual"]) - 1, 0))], 0).astype(np.uint32) << 16)
    crec[col] |= arr["graph_seq"][col].astype(np.uint32) | (arr["algnseq"][co[cr] + j].astype(np.uint32) << 8) | \
        np.where(j == 0, np.uint32(0x04000000), np.uint32(0))
    assert np.array_equal(crec[col] >> 16 & 0xFF, qb)
    qualp = np.zeros(Q + 32, np.uint8)
    qualp[:Q] = arr["qual"][:Q]
    return {"rhdr": rhdr.ravel(), "srec": srec.ravel(), "crec": crec, "qualp": qualp}


def check(g, a, n_threads=3, **kw):
    soa = hc.HostBatch(g, a, n_threads=n_threads, **kw)
    pk = hc.HostBatch(g, a, n_threads=n_threads, packed=True, **kw)
    nt = soa.n_tileable
    k = pk.pk
    assert k.n_reads == nt and pk.n_reads == soa.n_reads and pk.n_segments == soa.n_segments
    assert pk.stats.n_out == soa.stats.n_out and pk.stats.n_bad == soa.stats.n_bad and pk.stats.n_segments == soa.stats.n_segments
    arr = soa.arrays()
    want = pack_of_soa(arr, nt)
    got = pk.packed_arrays()
    for name in ("rhdr", "srec", "crec", "qualp"):
        assert np.array_equal(got[name], want[name]), name
    if nt:
        so, co, qo = arr["read_seg_off"], arr["read_col_off"], arr["read_qual_off"]
        assert k.max_read_segs == np.diff(so[:nt + 1]).max() and k.max_read_cols == np.diff(co[:nt + 1]).max()
        assert k.max_read_qual == np.diff(qo[:nt + 1]).max()
        sn = arr["seg_node"].astype(np.int64)
        assert k.max_read_node_span == max(int(sn[so[r]:so[r + 1]].max() - sn[so[r]:so[r + 1]].min()) for r in range(nt))
    assert np.array_equal(got["read_src"], arr["read_src"][:nt])
    # the other reads: a SoA batch of their own, same data, offsets from 0
    rest = pk.arrays()
    assert pk.c.n_tileable == 0 and pk.c.n_reads == soa.n_reads - nt
    s0, c0, q0 = int(arr["read_seg_off"][nt]), int(arr["read_col_off"][nt]), int(arr["read_qual_off"][nt])
    assert np.array_equal(rest["read_seg_off"], arr["read_seg_off"][nt:] - s0)
    assert np.array_equal(rest["read_col_off"], arr["read_col_off"][nt:] - c0)
    assert np.array_equal(rest["read_qual_off"], arr["read_qual_off"][nt:] - q0)
    for name, lo in (("read_algn_len", nt), ("read_mapq", nt), ("read_src", nt), ("seg_node", s0), ("seg_start", s0), ("seg_len", s0),
                     ("graph_seq", c0), ("algnseq", c0), ("qual", q0)):
        assert np.array_equal(rest[name], arr[name][lo:]), name
    assert np.array_equal(pk.read_src, soa.read_src)
    return soa, pk


def test_packed_flatten_equals_the_layout_of_the_soa_batch(tmp_path):
    g = hc.synth_graph(seed=3, genome_len=2000, n_nodes=900, n_paths=64)
    short = hc.synth_reads(g, 1500, seed=4, read_len=120, indel_rate=0.3, softclip_rate=0.2, low_mapq_rate=0.5)
    long_reads = hc.synth_reads(g, 40, seed=5, read_len=1500, indel_rate=0.0, softclip_rate=0.0)
    short2 = hc.synth_reads(g, 1500, seed=6, read_len=250, indel_rate=0.1, softclip_rate=0.1)
    mixed = util.concat_alnsets(tmp_path, short, long_reads, short2)
    for alns, kind in ((mixed, "mixed"), (long_reads, "general"), (short, "tiled")):
        soa, pk = check(g, alns)
        assert {"mixed": 0 < pk.pk.n_reads < pk.n_reads, "general": pk.pk.n_reads < pk.n_reads, "tiled": pk.c.n_reads == 0}[kind]
    # one thread, many threads: the same batch
    one = hc.HostBatch(g, mixed, n_threads=1, packed=True).packed_arrays()
    many = hc.HostBatch(g, mixed, n_threads=8, packed=True).packed_arrays()
    for name in ("rhdr", "srec", "crec", "qualp", "read_src"):
        assert np.array_equal(one[name], many[name]), name


def test_packed_flatten_with_duplicate_marks_and_short_quality_strings(tmp_path):
    g = hc.synth_graph(seed=8, genome_len=900, n_nodes=500, n_paths=32)
    a = hc.synth_reads(g, 6000, seed=9, read_len=100)
    dup = a.mark_duplicates()
    assert 0 < dup.sum() < a.n_reads
    check(g, a, skip=dup)
    # quality strings shorter than the read (and empty ones): the record's quality byte is 0 past the string
    import gamio
    f = str(tmp_path / "a.gam")
    hc.synth_reads(g, 1500, seed=10, read_len=100, indel_rate=0.05).write_gam(f)
    alns = gamio.read_gam(f)
    rng = np.random.default_rng(1)
    for al in alns:
        al["quality"] = al["quality"][:int(rng.integers(0, 130))]
    f2 = str(tmp_path / "b.gam")
    open(f2, "wb").write(gamio.write_gam(alns))
    b = hc.AlnSet.read_gam(f2)
    assert b.n_reads == len(alns)
    soa, pk = check(g, b)
    ql = np.diff(pk.packed_arrays()["rhdr"].reshape(-1, 4)[:, 1].astype(np.int64))
    assert ql.min() == 0 and ql.max() >= 100


def test_packed_flatten_of_a_sliced_alignment_set(tmp_path):
    g = hc.synth_graph(seed=12, genome_len=1500, n_nodes=900, n_paths=48)
    a = hc.synth_reads(g, 30000, seed=13, read_len=90)
    f = str(tmp_path / "x.gam")
    a.write_gam(f)
    parts = hc.AlnParts.read_gam(f)
    soa, pk = check(g, parts)
    whole = hc.HostBatch(g, hc.AlnSet.read_gam(f), packed=True).packed_arrays()
    got = pk.packed_arrays()
    for name in ("rhdr", "srec", "crec", "qualp", "read_src"):
        assert np.array_equal(got[name], whole[name]), name


def test_a_graph_beyond_the_segment_records_node_field_stays_in_the_soa_form():
    """(ABI 5) VGAN_HC_SREC keeps a node id in 18 bits: a read that touches a node beyond them is not packed -- it leaves with
    the other reads of the batch, for the general kernel."""
    g = hc.synth_graph(seed=5, genome_len=640000, n_nodes=280000, n_paths=8)
    a = hc.synth_reads(g, 3000, seed=6, read_len=100)
    soa = hc.HostBatch(g, a, n_threads=3)
    pk = hc.HostBatch(g, a, n_threads=3, packed=True)
    assert pk.n_reads == soa.n_reads and 0 < pk.pk.n_reads < soa.n_tileable
    nodes = pk.packed_arrays()["srec"] & 0x3FFFF
    arr = pk.arrays()
    assert nodes.max() <= 0x3FFFF and arr["seg_node"].max() > 0x3FFFF
    # every read whose nodes fit the field was packed
    so = arr["read_seg_off"]
    top = np.array([arr["seg_node"][so[r]:so[r + 1]].max() for r in range(pk.c.n_reads)])
    tileable_left = sum(1 for r in range(pk.c.n_reads) if top[r] <= 0x3FFFF)
    assert soa.n_tileable - pk.pk.n_reads >= pk.c.n_reads - (soa.n_reads - soa.n_tileable) - tileable_left
