"""CPU tests of the product's host front end (no GPU): library loads and exports every declared symbol,
GAM reader vs the independent Python decoder, reconstruction KATs through the product, flatten vs the oracle."""
import json
import os
import re

import numpy as np
import pytest

import gamio
import orc
import util
import vgan_amd
from vgan_amd import _native as N
from vgan_amd import haplocart as hc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = vgan_amd.load()
    hdr = open(os.path.join(ROOT, "include", "vgan_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vgan_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), "libvgan_gpu.so does not export %s" % name
    assert declared == set(N.SYMBOLS), declared ^ set(N.SYMBOLS)
    m = re.search(r"#define VGAN_ABI_VERSION (\d+)", hdr)
    assert m and L.vgan_abi_version() == int(m.group(1))


def test_no_device_is_an_error_not_a_fallback(golden_dir):
    if N.lib().vgan_device_count() > 0:
        pytest.skip("a GPU is visible")
    g = hc.Graph.load(os.path.join(golden_dir, "reconstruct", "target_graph.gfa"))
    with pytest.raises(N.NativeError) as e:
        hc.HcContext(g)
    assert e.value.code == N.VGAN_ENODEV


def test_gam_reader_matches_python_decoder(golden_dir):
    for f in ["reconstruct/test_reads.gam", "alignments/J2a1a1a1.gam", "alignments/all_the_same.gam",
              "alignments/all_the_same_reverse.gam", "alignments/two_unique.gam"]:
        p = os.path.join(golden_dir, f)
        ref = orc.AlnSet([a for a in gamio.read_gam(p) if a["identity"] != 0])
        got = hc.AlnSet.read_gam(p).arrays()
        assert got["n_reads"] == ref.n_reads
        for k in ("seq_off", "qual_off", "map_off", "m_node", "m_offset", "m_rev", "edit_off", "e_from", "e_to",
                  "e_seq_off", "mapq", "identity"):
            assert np.array_equal(got[k], getattr(ref, k)), (f, k)
        for k in ("seq", "qual", "e_seq"):
            assert got[k].tobytes() == getattr(ref, k)[:-1].tobytes(), (f, k)


def test_gam_unmapped_filter_and_roundtrip(tmp_path):
    alns = gamio.read_gam(os.path.join(ROOT, "tests/golden/reconstruct/test_reads.gam"))
    alns[3]["identity"] = 0.0  # readGAM.h:47 drops it
    data = gamio.write_gam(alns, group=4)
    a = hc.AlnSet.parse_gam(data)
    assert a.n_reads == 9
    b = hc.AlnSet.parse_gam(data, keep_unmapped=True)
    assert b.n_reads == 10
    out = str(tmp_path / "rt.gam")
    b.write_gam(out, group_size=3)
    again = gamio.read_gam(out)
    assert [x["sequence"] for x in again] == [x["sequence"] for x in alns]
    strip = lambda a: [(m["position"], m["edit"]) for m in a["path"]["mapping"]]  # rank is not kept
    assert [strip(x) for x in again] == [strip(x) for x in alns]
    assert [x["quality"] for x in again] == [x["quality"] for x in alns]
    # malformed stream -> error code, not a crash
    with pytest.raises(N.NativeError):
        hc.AlnSet.parse_gam(gamio.gunzip_all(data)[:-7])


@pytest.mark.parametrize("graph_file", ["target_graph.gfa", "target_graph.og"])
def test_reconstruction_kats_through_product(golden_dir, graph_file):
    """The reference's 10 reconstruction KATs (src/test.cpp:855-994); it runs them on target_graph.og, so does this."""
    d = os.path.join(golden_dir, "reconstruct")
    g = hc.Graph.load(os.path.join(d, graph_file))
    assert g.n_paths == 5 and sorted(g.path_names) == ["seq_1", "seq_2", "seq_3", "seq_4", "seq_5"]
    a = hc.AlnSet.read_gam(os.path.join(d, "test_reads.gam"))
    for case in json.load(open(os.path.join(d, "expected.json")))["cases"]:
        gs, rs, sizes = hc.reconstruct(g, a, case["read"])
        assert gs.decode() == case["graph_seq"], case["name"]
        assert rs.decode() == case["read_seq"], case["name"]
        assert sizes == case["mppg_sizes"], case["name"]


def test_odgi_graph_equals_its_gfa(golden_dir, tmp_path):
    """The ODGI reader on the reference's fixture: node ids, sequences, path names and which paths visit each
    node are those of the GFA of the same graph; damaged files are errors (or, when the damage is in a part the reader skips,
    the same graph), never a crash."""
    d = os.path.join(golden_dir, "reconstruct")
    a = hc.Graph.load(os.path.join(d, "target_graph.gfa"))
    b = hc.Graph.load(os.path.join(d, "target_graph.og"))
    assert (a.min_id, a.max_id, a.n_paths) == (b.min_id, b.max_id, b.n_paths) == (2, 29, 5)
    assert np.array_equal(a.node_seq, b.node_seq) and np.array_equal(a.node_seq_off, b.node_seq_off)
    assert b.path_names == ["seq_5", "seq_1", "seq_4", "seq_2", "seq_3"]  # odgi's path handle order
    col = {n: i for i, n in enumerate(a.path_names)}
    ga, gb = np.asarray(a.pathsgo()), np.asarray(b.pathsgo())
    for j, n in enumerate(b.path_names):
        assert np.array_equal(ga[:, col[n]], gb[:, j]), n
    # with a graph_paths sidecar (and no path_supports) the columns follow the sidecar's names, whatever order the .og has
    import shutil
    hcdir = tmp_path / "hc"
    hcdir.mkdir()
    shutil.copy(os.path.join(d, "target_graph.og"), str(hcdir / "graph.og"))
    (hcdir / "graph_paths").write_text("seq_3 extra tokens\nseq_1\nseq_2\nseq_5\nseq_4\n")
    c = hc.Graph.load(str(hcdir / "graph.og"), str(hcdir))
    assert c.path_names == ["seq_3", "seq_1", "seq_2", "seq_5", "seq_4"]
    gc = np.asarray(c.pathsgo())
    for j, n in enumerate(c.path_names):
        assert np.array_equal(ga[:, col[n]], gc[:, j]), n
    raw = open(os.path.join(d, "target_graph.og"), "rb").read()
    import random
    rng = random.Random(5)
    n_err = n_same = 0
    for trial in range(400):
        data = bytearray(raw)
        kind = trial % 4
        if kind == 0:
            data = data[:rng.randrange(4, len(data))]
        elif kind == 1:
            for _ in range(rng.randrange(1, 6)):
                data[rng.randrange(len(data))] = rng.randrange(256)
        elif kind == 2:
            i = rng.randrange(len(data))
            data[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
        else:
            i = rng.randrange(4, len(data) - 8)
            data[i:i + 8] = (rng.getrandbits(64) >> rng.randrange(64)).to_bytes(8, "little")
        f = tmp_path / "x.og"
        f.write_bytes(bytes(data))
        try:
            c = hc.Graph.load(str(f))
        except N.NativeError as e:
            assert e.code == N.VGAN_EIO
            n_err += 1
            continue
        assert c.max_id == 29 and len(c.node_seq) == len(b.node_seq)
        n_same += 1
    assert n_err > 200 and n_same > 0


def _check_flatten_against_oracle(g, a, batch):
    og = util.orc_graph_from_product(g)
    oa = util.orc_alnset_from_product(a)
    arr = batch.arrays()
    ident = a.arrays()["identity"]
    where = {int(r): k for k, r in enumerate(arr["read_src"])}  # the batch is ordered for the device, not as the input
    n_seen = 0
    for r in range(a.n_reads):
        if ident[r] < 1e-10:
            continue
        rc, gs, rs, sizes = orc.reconstruct(og, oa, r)
        rc2, S, U, node = orc.hc_read_segments(og, oa, r)
        if rc != 0 or rc2 != 0:
            assert r not in where
            continue  # the reference would terminate: product skips the read
        k = where[r]
        s0, s1 = arr["read_seg_off"][k], arr["read_seg_off"][k + 1]
        c0 = arr["read_col_off"][k]
        assert arr["read_algn_len"][k] == len(rs)
        assert arr["graph_seq"][c0:c0 + len(gs)].tobytes() == gs
        assert arr["algnseq"][c0:c0 + len(rs)].tobytes() == rs
        assert arr["seg_node"][s0:s1].tolist() == node.tolist()
        # segment starts/lengths follow update_likelihood.cpp:36-45
        pos = 0
        n_map = len(node)
        for i in range(n_map):
            assert arr["seg_start"][s0 + i] == pos
            assert arr["seg_len"][s0 + i] == min(sizes[i], len(gs) - pos)
            pos += min(sizes[i], len(rs) - pos)
        n_seen += 1
    assert n_seen == batch.n_reads
    # tileable reads come first: those of mapping quality VGAN_HC_MAPQ_MAJOR (60) in ascending order of their lowest node id,
    # then the others in theirs (include/vgan_gpu.h "Order", ABI 5)
    nt = batch.n_tileable
    low = [(int(arr["read_mapq"][k]) != 60, int(arr["seg_node"][arr["read_seg_off"][k]:arr["read_seg_off"][k + 1]].min()))
           for k in range(nt) if arr["read_seg_off"][k + 1] > arr["read_seg_off"][k]]
    assert low == sorted(low)


def test_flatten_matches_oracle_on_fixtures(golden_dir):
    d = os.path.join(golden_dir, "reconstruct")
    g = hc.Graph.load(os.path.join(d, "target_graph.gfa"))
    a = hc.AlnSet.read_gam(os.path.join(d, "test_reads.gam"))
    b = hc.HostBatch(g, a)
    assert b.stats.n_in == 10 and b.stats.n_out == 10 and b.stats.n_bad == 0
    _check_flatten_against_oracle(g, a, b)


def test_synth_graph_shape_and_flatten():
    g = hc.synth_graph(seed=7, genome_len=2000, n_nodes=1400, n_paths=130)
    assert g.min_id == 1 and g.max_id == 1400 and g.n_paths == 130
    lens = np.diff(g.node_seq_off)[1:]
    assert lens.max() <= 8 and lens.min() >= 1
    # every path visits exactly one node per site
    pg = g.pathsgo()
    site = g.pangenome_base[1:]
    for p in (0, 17, 129):
        sup = pg[1:, p].astype(bool)
        assert len(np.unique(site[sup])) == len(np.unique(site)) == sup.sum()
    assert len(g.path_names) == 130
    a = hc.synth_reads(g, 300, seed=3, read_len=150, indel_rate=0.2, softclip_rate=0.2)
    assert a.n_reads == 300
    b = hc.HostBatch(g, a, n_threads=3)
    assert b.stats.n_out + b.stats.n_bad + b.stats.n_unmapped == 300
    assert b.stats.n_out >= 295
    _check_flatten_against_oracle(g, a, b)
    # deterministic regardless of thread count
    b1 = hc.HostBatch(g, a, n_threads=1)
    for k, v in b.arrays().items():
        if k != "_owner":
            assert np.array_equal(v, b1.arrays()[k]), k


def test_flatten_counts_reads_the_reference_would_die_on():
    g = hc.synth_graph(seed=1, genome_len=600, n_nodes=420, n_paths=20)
    a = hc.synth_reads(g, 50, seed=9, read_len=60)
    arr = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in a.arrays().items()}
    arr["m_node"][arr["map_off"][5]] = 99999          # unknown node -> pangenome_map.at() throws
    arr["identity"][7] = 0.0                          # unmapped
    arr["mapq"][9] = 150                              # out-of-range table index (Q10): clamped and counted
    oa = orc.AlnSet.from_arrays(**arr)
    v = N.AlnSetView(oa.n_reads, *[getattr(oa, k).ctypes.data for k in
                                   ("seq_off", "seq", "qual_off", "qual", "mapq", "identity")],
                     None, None,
                     *[getattr(oa, k).ctypes.data for k in ("map_off", "m_node", "m_offset", "m_rev", "edit_off",
                                                            "e_from", "e_to", "e_seq_off", "e_seq")])
    h = N.vp()
    N.check(N.lib().vgan_aln_from_arrays(v, h))
    a2 = hc.AlnSet(h)
    b = hc.HostBatch(g, a2)
    assert b.stats.n_bad == 1 and b.stats.n_unmapped == 1 and b.stats.n_clamped == 1 and b.stats.n_out == 48


def test_graph_write_and_reload(tmp_path):
    g = hc.synth_graph(seed=5, genome_len=500, n_nodes=330, n_paths=70)
    g.write(str(tmp_path))
    g2 = hc.Graph.load(str(tmp_path / "graph.gfa"), str(tmp_path))
    assert g2.n_paths == 70 and g2.max_id == g.max_id
    assert np.array_equal(g2.mask, g.mask)
    assert np.array_equal(g2.pangenome_base, g.pangenome_base)
    assert np.array_equal(g2.mappability, g.mappability)
    assert g2.node_seq.tobytes() == g.node_seq.tobytes()
    assert g2.path_names == g.path_names and g2.parents_txt == g.parents_txt


def test_sidecar_loaders_match_oracle(tmp_path):
    import ctypes as C
    g = hc.synth_graph(seed=11, genome_len=400, n_nodes=260, n_paths=33)
    g.write(str(tmp_path))
    L = orc.lib()
    txt = gamio.gunzip_all(open(tmp_path / "path_supports.gz", "rb").read())
    rows = np.zeros((g.max_id + 1, 33), np.uint8)
    n = L.orc_load_path_supports(txt, C.c_int32(33), rows.ctypes.data_as(C.c_void_p), C.c_int64(g.max_id + 1))
    assert n == g.max_id + 1 and np.array_equal(rows, g.pathsgo())
    pb = np.full(g.max_id + 1, -1, np.int32)
    L.orc_load_pangenome_map(open(tmp_path / "parsed_pangenome_mapping", "rb").read(),
                             pb.ctypes.data_as(C.c_void_p), C.c_int64(g.max_id + 1))
    assert np.array_equal(pb, g.pangenome_base)
    mp = np.zeros(len(g.mappability) + 8)
    n = L.orc_load_mappabilities(open(tmp_path / "mappability.tsv", "rb").read(), mp.ctypes.data_as(C.c_void_p),
                                 C.c_int64(len(mp)))
    assert n == len(g.mappability) and np.array_equal(mp[:n], g.mappability)


def test_flatten_puts_tileable_reads_first(tmp_path):
    """vgan_hc_flatten orders a batch as [reads satisfying the tile contract, by lowest node id | the others] and reports
    the split point; read_src maps batch order back to the alignment set."""
    from vgan_amd import haplocart as hc
    import util
    g = hc.synth_graph(seed=3, genome_len=2000, n_nodes=900, n_paths=64)
    short = hc.synth_reads(g, 1500, seed=4, read_len=120, indel_rate=0.3, softclip_rate=0.2)
    long_reads = hc.synth_reads(g, 40, seed=5, read_len=1500, indel_rate=0.0, softclip_rate=0.0)
    short2 = hc.synth_reads(g, 1500, seed=6, read_len=250, indel_rate=0.1, softclip_rate=0.1)
    mixed = util.concat_alnsets(tmp_path, short, long_reads, short2)
    assert mixed.n_reads == 3040
    for alns, kind in ((mixed, "mixed"), (long_reads, "general"), (short, "tiled")):
        b = hc.HostBatch(g, alns, n_threads=3)
        arr = b.arrays()
        nt = b.n_tileable
        src = b.read_src
        assert len(set(src.tolist())) == b.n_reads and b.stats.n_out == b.n_reads
        assert {"mixed": 0 < nt < b.n_reads, "general": nt < b.n_reads, "tiled": nt == b.n_reads}[kind]
        so, co, qo = arr["read_seg_off"], arr["read_col_off"], arr["read_qual_off"]
        for r in range(b.n_reads):
            cols = int(co[r + 1] - co[r])
            st = arr["seg_start"][so[r]:so[r + 1]].astype(np.int64)
            ln = arr["seg_len"][so[r]:so[r + 1]].astype(np.int64)
            ok = (cols <= 1280 and qo[r + 1] - qo[r] <= 1280 and len(st) <= 512 and np.all(st[1:] >= st[:-1] + ln[:-1])
                  and np.all(ln > 0))
            assert ok == (r < nt), (r, nt, cols)
        # the tileable part ascends in the reads' lowest node id, input order kept among equals; the rest keeps the input order
        # ((ABI 5) the reads of mapping quality 60 first, the others behind them)
        low = np.array([arr["seg_node"][so[r]:so[r + 1]].min() for r in range(nt)], np.int64)
        low = low + (arr["read_mapq"][:nt] != 60).astype(np.int64) * (1 << 30)
        key = low * (1 << 32) + src[:nt].astype(np.int64)
        assert np.all(np.diff(key) > 0) and np.all(np.diff(src[nt:].astype(np.int64)) > 0)


def test_masked_flatten_equals_flatten_of_filtered_set():
    """vgan_hc_flatten_masked(skip = duplicate marks) gives the batch of the de-duplicated set without building it."""
    from vgan_amd import haplocart as hc
    g = hc.synth_graph(seed=8, genome_len=900, n_nodes=500, n_paths=32)
    a = hc.synth_reads(g, 6000, seed=9, read_len=100)
    dup = a.mark_duplicates()
    assert 0 < dup.sum() < a.n_reads
    masked = hc.HostBatch(g, a, n_threads=3, skip=dup)
    plain = hc.HostBatch(g, a.without(dup), n_threads=3)
    assert masked.n_reads == plain.n_reads == masked.stats.n_out and masked.n_tileable == plain.n_tileable
    ma, pa = masked.arrays(), plain.arrays()
    for k in ("read_seg_off", "read_col_off", "read_qual_off", "read_algn_len", "read_mapq", "seg_node", "seg_start",
              "seg_len", "graph_seq", "algnseq", "qual"):
        assert np.array_equal(ma[k], pa[k]), k
    kept = np.flatnonzero(~dup)
    assert np.array_equal(kept[pa["read_src"]], ma["read_src"])


def test_sliced_alignment_set_matches_the_merged_one(tmp_path):
    """vgan_alnparts_*: the GAM kept as its parser slices gives the same duplicate marks and the same flattened batch
    (arrays and read_src) as the merged alignment set, whole or in slice ranges."""
    from vgan_amd import haplocart as hc
    g = hc.synth_graph(seed=12, genome_len=1500, n_nodes=900, n_paths=48)
    a = hc.synth_reads(g, 30000, seed=13, read_len=90)
    f = str(tmp_path / "x.gam")
    a.write_gam(f)
    parts = hc.AlnParts.read_gam(f)
    merged = hc.AlnSet.read_gam(f)
    n = merged.n_reads  # unmapped reads (identity 0) are dropped by both readers
    assert parts.n_reads == n and 29990 <= n <= 30000 and parts.n_parts >= 3
    assert [parts.first_read(i) for i in range(parts.n_parts + 1)][-1] == n
    dup_p, dup_m = parts.mark_duplicates(), merged.mark_duplicates()
    assert np.array_equal(dup_p, dup_m) and dup_m.sum() > 0
    for skip in (None, dup_m):
        bp = hc.HostBatch(g, parts, n_threads=3, skip=skip)
        bm = hc.HostBatch(g, merged, n_threads=1, skip=skip)  # one chunk: tileable reads in input order
        ap, am = bp.arrays(), bm.arrays()
        assert bp.n_reads == bm.n_reads and bp.n_tileable == bm.n_tileable
        for k in ("read_seg_off", "read_col_off", "read_qual_off", "read_algn_len", "read_mapq", "seg_node", "seg_start",
                  "seg_len", "graph_seq", "algnseq", "qual", "read_src"):
            assert np.array_equal(ap[k], am[k]), k
    # a slice range covers exactly its reads
    b01 = hc.HostBatch(g, parts, 0, 2, n_threads=2)
    assert sorted(b01.read_src.tolist()) == list(range(parts.first_read(2)))
    # merging afterwards gives the merged set
    m2 = parts.merge()
    assert m2.n_reads == n and np.array_equal(m2.arrays()["seq"], merged.arrays()["seq"])


def test_messages_parsed_in_slices_on_threads_equal_the_file(tmp_path):
    """vgan_alnparts_from_messages (what the device front end's host-left reads come through): the Alignment messages of a GAM file,
    laid one after the other, parsed in slices on several threads -- against the file's own parse, on one thread and on many."""
    import ctypes as C
    g = hc.synth_graph(seed=12, genome_len=1500, n_nodes=900, n_paths=48)
    a = hc.synth_reads(g, 30000, seed=14, read_len=90, indel_rate=0.05)
    f = str(tmp_path / "x.gam")
    a.write_gam(f)
    raw = gamio.gunzip_all(open(f, "rb").read())
    msgs, i = [], 0
    while i < len(raw):  # groups {count, count x (length, bytes)}; a group's first item is its tag
        cnt, i = gamio._varint(raw, i)
        for j in range(cnt):
            ln, i = gamio._varint(raw, i)
            if j > 0:
                msgs.append(raw[i:i + ln])
            i += ln
    assert len(msgs) == 30000
    offs = np.zeros(len(msgs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(m) for m in msgs])
    byts = np.frombuffer(b"".join(msgs), np.uint8)
    want = hc.AlnSet.read_gam(f).arrays()
    for threads in (1, 7, 0):
        h = N.vp()
        N.check(N.lib().vgan_alnparts_from_messages(byts.ctypes.data, offs.ctypes.data, len(msgs), 0, threads, C.byref(h)))
        parts = hc.AlnParts(h)
        assert parts.n_parts == (1 if threads == 1 else 7) and parts.n_reads == want["n_reads"]
        got = parts.merge().arrays()
        for k in ("seq_off", "qual_off", "map_off", "seq", "qual", "mapq", "m_node", "m_offset", "edit_off", "e_from", "e_to"):
            assert np.array_equal(got[k], want[k]), (threads, k)
    # offsets that do not describe the bytes are an error, not an allocation of what they ask for
    bad = offs.copy()
    bad[0] = 1 << 60
    h = N.vp()
    assert N.lib().vgan_alnparts_from_messages(byts.ctypes.data, bad.ctypes.data, len(msgs), 0, 1, C.byref(h)) < 0


def test_argmax_names_the_first_of_the_paths_the_sums_do_not_tell_apart():
    """vgan_hc_argmax (HaploCart.cpp:423): the first maximum, ties taken as exact arithmetic means them -- sums equal up to the order
    of their additions (1e-12 relative) are one maximum; a sum a column's term apart is not."""
    def am(v):
        a = np.ascontiguousarray(v, np.float64)
        return N.lib().vgan_hc_argmax(a.ctypes.data, len(a))
    m = -1.0e7
    assert am([m - 5, m + 1e-9, m, m + 2e-9, m - 1e-3]) == 1   # 1, 2, 3 are one maximum: the first of them
    assert am([m, m + 1e-3]) == 1 and am([m + 1e-3, m]) == 0      # told apart: the larger
    assert am([m, m, m]) == 0 and am([3.0]) == 0
    assert am([-np.inf, -np.inf]) == 0 and am([-np.inf, m]) == 1
    assert am([0.0, 1e-300, -1e-300]) == 1                        # (around zero the tolerance is zero)


def test_gam_stream_chunks_equal_the_whole(tmp_path):
    """vgan_gam_stream_*: chunks arrive in input order with consecutive bases, their streamed duplicate marks equal the
    marks over the whole file, and their batches concatenate to the whole file's batch content."""
    from vgan_amd import haplocart as hc
    g = hc.synth_graph(seed=14, genome_len=1500, n_nodes=900, n_paths=48)
    a = hc.synth_reads(g, 50000, seed=15, read_len=80)
    f = str(tmp_path / "y.gam")
    a.write_gam(f)
    whole = hc.AlnSet.read_gam(f)
    dup_whole = whole.mark_duplicates()
    bw = hc.HostBatch(g, whole, n_threads=1, skip=dup_whole)
    want = {int(s): (int(n), int(c)) for s, n, c in zip(bw.read_src, np.diff(bw.arrays()["read_seg_off"]), np.diff(bw.arrays()["read_col_off"]))}
    st = hc.GamStream(f)
    dd = hc.Dedup()
    base, marks, got = 0, [], {}
    n_chunks = 0
    for chunk in st.chunks(12000):
        assert chunk.base == base and chunk.n_reads >= min(12000, whole.n_reads - base)
        m = dd.mark(chunk)
        marks.append(m)
        b = hc.HostBatch(g, chunk, n_threads=2, skip=m)
        arr = b.arrays()
        for s, n, c in zip(b.read_src, np.diff(arr["read_seg_off"]), np.diff(arr["read_col_off"])):
            got[int(s)] = (int(n), int(c))
        base += chunk.n_reads
        n_chunks += 1
    assert n_chunks >= 3 and base == whole.n_reads
    assert np.array_equal(np.concatenate(marks), dup_whole)
    assert got == want
    # a corrupt file fails in next(), not silently
    blob = bytearray(open(f, "rb").read())
    blob[len(blob) // 2] ^= 0xFF
    bad = str(tmp_path / "bad.gam")
    open(bad, "wb").write(bytes(blob))
    with pytest.raises(Exception):
        for _ in hc.GamStream(bad).chunks(12000):
            pass


def test_gbwt_reader_on_the_reference_fixture(golden_dir, tmp_path):
    """vgan_gbwt_*: the threads of test/reconstructInputSeq/target_graph.gbwt are the P lines of target_graph.gfa (forward
    at sequence 2k, reverse complement at 2k+1); the node x path matrix follows readOG_Euka.h:55-73 with its quirks; broken
    files give an error code, never a crash."""
    import ctypes as C
    import random
    L = N.load()
    d = os.path.join(golden_dir, "reconstruct")
    raw = open(os.path.join(d, "target_graph.gbwt"), "rb").read()
    h = N.vp()
    N.check(L.vgan_gbwt_load(os.path.join(d, "target_graph.gbwt").encode(), C.byref(h)))
    _, paths = orc.read_gfa(os.path.join(d, "target_graph.gfa"))
    assert L.vgan_gbwt_sequences(h) == 2 * len(paths) == 10 and L.vgan_gbwt_bidirectional(h) == 1

    def extract(handle, s, cap=64):
        buf = np.zeros(cap, np.uint64)
        n = L.vgan_gbwt_extract(handle, s, buf.ctypes.data, cap)
        assert 0 <= n <= cap
        return [int(v) for v in buf[:n]]

    for k, (_, steps) in enumerate(paths):
        assert extract(h, 2 * k) == [2 * nid + int(rev) for nid, rev in steps]
        assert extract(h, 2 * k + 1) == [2 * nid + 1 - int(rev) for nid, rev in reversed(steps)]
    assert extract(h, 10) == [] and extract(h, -1) == []       # gbwt::GBWT::extract of an id beyond the index
    assert L.vgan_gbwt_extract(h, 0, None, 0) == len(paths[0][1])  # length query
    # readOG_Euka.h:55-73: sequences 0..4 (not paths 0..4), encoded numbers as node ids, row = number - 1
    n_nodes, n_paths = 28, 5
    m = np.zeros((n_nodes, n_paths), np.uint8)
    N.check(L.vgan_gbwt_node_path_matrix(h, n_nodes, n_paths, m.ctypes.data))
    want = np.zeros_like(m)
    for p in range(n_paths):
        for v in extract(h, p):
            if 0 <= v - 1 < n_nodes:
                want[v - 1, p] = 1
    assert np.array_equal(m, want) and m.any() and not m[:, 1].all()
    assert m[2 * 2 - 1, 0] == 1 and m[2 * 4 - 1, 0] == 1 and m[2 - 1, 0] == 0  # seq_1: 2+,4+,... -> rows 3, 7 (node "4", "8")
    # the bare payload without vg's type tag reads the same
    bare = str(tmp_path / "bare.gbwt")
    open(bare, "wb").write(raw[8:])
    h2 = N.vp()
    N.check(L.vgan_gbwt_load(bare.encode(), C.byref(h2)))
    assert [extract(h2, s) for s in range(10)] == [extract(h, s) for s in range(10)]
    L.vgan_gbwt_free(h2)
    L.vgan_gbwt_free(h)
    # corrupted / truncated files
    rng = random.Random(5)
    n_err = 0
    for trial in range(400):
        data = bytearray(raw)
        if trial % 2:
            data = data[: rng.randrange(1, len(data))]
        else:
            for _ in range(rng.randrange(1, 6)):
                data[rng.randrange(len(data))] = rng.randrange(256)
        f = str(tmp_path / "bad.gbwt")
        open(f, "wb").write(bytes(data))
        hb = N.vp()
        if L.vgan_gbwt_load(f.encode(), C.byref(hb)) < 0:
            n_err += 1
            continue
        for s in range(10):  # whatever loads must walk to an end or report an error
            buf = np.zeros(4096, np.uint64)
            assert L.vgan_gbwt_extract(hb, s, buf.ctypes.data, 4096) <= 4096
        L.vgan_gbwt_free(hb)
    assert n_err > 150

