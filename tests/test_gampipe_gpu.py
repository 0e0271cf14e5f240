"""GPU tests of the GAM front end as a pipeline over the file's pieces (csrc/gam_pipe.hip + the per-piece functions of
csrc/gam_kernels.hip): byte / integer work, so every array a piece's parse leaves is bit for bit the host parser's for the same reads,
wherever the pieces' boundaries fall -- inside a group, inside a message, inside a group's tag."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

from vgan_amd import _native as N
from vgan_amd import haplocart as hc

from test_gamdev_gpu import GamDev, _bgzf, gunzip_members, host_slice

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_OFFS = {0: "M", 1: "Q", 2: "E", 3: "S"}  # the offset arrays and what they count


def pieces_arrays(data, piece_bytes, keep_unmapped=False, tail_bytes=0):
    """The file through ONE device object, piece by piece (each framed from what the piece before left), the pieces' arrays put
    together as one parse of the whole file would leave them."""
    plan = hc.gampipe_plan(data, piece_bytes)
    gd = hc.GamDevice()
    h = GamDev.__new__(GamDev)  # (its array() reads a vgan_gamdev through the test aid)
    h._h = gd._h
    carry = None
    parts = {w: [] for w in list(range(14)) + [17]}
    tot = {"R": 0, "M": 0, "E": 0, "S": 0, "Q": 0}
    n_msgs = 0
    for i in range(len(plan)):
        carry = gd.parse_piece(data, i, carry, piece_bytes, keep_unmapped, tail_bytes)
        sizes = np.zeros(8, np.uint64)
        N.check(N.lib().vgan_gamdev_sizes(gd._h, sizes.ctypes.data, None))
        h.sizes = dict(zip(("inflated", "messages", "R", "M", "E", "S", "Q", "reanchored"), (int(x) for x in sizes[:8])))
        n_msgs += h.sizes["messages"]
        if h.sizes["messages"] == 0 or h.sizes["R"] == 0:
            continue
        for w in parts:
            if w == 17:
                out = np.zeros(max(h.sizes["R"], 1), np.uint32)
                N.check(N.lib().vgan_gamdev_download(gd._h, 17, out.ctypes.data))
                parts[w].append(out[:h.sizes["R"]])
                continue
            a = h.array(w)
            if w in _OFFS:  # offsets: from the piece's own zero to the file's
                a = a.astype(np.uint64) + tot[_OFFS[w]]
                a = a[1:] if parts[w] else a
            parts[w].append(a)
        for k in tot:
            tot[k] += h.sizes[k]
    h._h = None
    gd.close()
    out = {}
    for w, v in parts.items():
        out[w] = np.concatenate(v) if v else np.zeros(0)
    return out, tot, len(plan), n_msgs


def host_arrays(data, keep_unmapped, tmp_path):
    p = str(tmp_path / "h.gam")
    open(p, "wb").write(data)
    a = hc.AlnSet.read_gam(p, keep_unmapped=keep_unmapped)
    want = host_slice(a)
    x = a.arrays()
    want[17] = np.diff(x["seq_off"]).astype(np.uint32)
    return a, want


def assert_same(got, want):
    for w, v in want.items():
        g = got[w]
        if w in _OFFS:
            assert np.array_equal(g.astype(np.uint64), v.astype(np.uint64)), w
        else:
            assert g.shape == v.shape and np.array_equal(g, v), w


@pytest.mark.parametrize("piece_bytes", [1, 70_000, 300_000, 1 << 20, 1 << 30])
def test_the_pieces_arrays_are_the_host_parsers_whatever_the_piece_size(tmp_path, piece_bytes):
    """A file of ~70 BGZF members, groups of 37 messages, reads with indels and soft clips: pieces of one member each (every boundary
    falls inside a message), of a few members, of a megabyte, and the whole file as one piece."""
    g = hc.synth_graph(seed=5, genome_len=3000, n_nodes=2000, n_paths=50)
    a = hc.synth_reads(g, 9000, seed=61, read_len=150, indel_rate=0.1, softclip_rate=0.1)
    p = str(tmp_path / "s.gam")
    a.write_gam(p, group_size=37)
    data = open(p, "rb").read()
    for keep in (False, True):
        host, want = host_arrays(data, keep, tmp_path)
        got, tot, n_pieces, n_msgs = pieces_arrays(data, piece_bytes, keep)
        assert tot["R"] == host.n_reads and n_msgs == 9000
        assert_same(got, want)
    plan = hc.gampipe_plan(data, piece_bytes)
    assert len(plan) == n_pieces and sum(c for _, c, _ in plan) <= len(data) and sum(o for _, _, o in plan) == len(gunzip_members(data))
    if piece_bytes == 1:
        assert n_pieces > 50  # (a piece per member)
    if piece_bytes == 1 << 30:
        assert n_pieces == 1


def test_piece_boundaries_at_every_place_of_a_group(tmp_path):
    """The stream re-cut into members of 1000 bytes, a piece per member: over a few hundred boundaries every place is hit -- before a
    group's count, between the count and the tag, inside the tag, inside a message's length, inside a message; groups of one message,
    of three, of many; and the reference's own GAMs the same way."""
    import gamio
    rng = np.random.default_rng(3)
    for trial, group in enumerate((1, 3, 64, 1000)):
        alns = []
        for r in range(1500):
            n = int(rng.integers(20, 90))
            alns.append({"sequence": bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)), "name": b"r%d" % r,
                         "quality": bytes(rng.integers(2, 41, n, dtype=np.uint8)), "mapping_quality": int(rng.integers(0, 61)), "identity": float(rng.random() < 0.97),
                         "path": {"mapping": [{"position": {"node_id": int(rng.integers(1, 900)), "offset": int(rng.integers(0, 4)), "is_reverse": bool(r & 1)},
                                               "edit": [{"from_length": n, "to_length": n}], "rank": 1}]}})
        raw = gamio.write_gam(alns, group=group, compress=False)
        data = _bgzf(raw, chunk=1000 + 7 * trial)
        for keep in (False, True):
            host, want = host_arrays(data, keep, tmp_path)
            got, tot, n_pieces, n_msgs = pieces_arrays(data, 1, keep)
            assert n_pieces > 100 and tot["R"] == host.n_reads and n_msgs == 1500
            assert_same(got, want)
    for name in ("alignments/J2a1a1a1.gam", "reconstruct/test_reads.gam", "alignments/two_unique.gam"):
        raw = gunzip_members(open(os.path.join(GOLD, name), "rb").read())
        data = _bgzf(raw, chunk=777)
        host, want = host_arrays(data, True, tmp_path)
        got, tot, n_pieces, _ = pieces_arrays(data, 1, True)
        assert n_pieces > 3 and tot["R"] == host.n_reads
        assert_same(got, want)


def test_tag_like_bytes_and_pieces(tmp_path):
    """Read names that hold the group tag's bytes with a plausible count in front, in a file of several 1 MiB segments per piece and
    several pieces: false tags are given up inside a piece as inside a whole file, and a piece's end never takes one for a group."""
    import gamio
    rng = np.random.default_rng(21)
    fake = bytes([5, 3]) + b"GAM" + bytes([40]) + b"\x0a\x10ACGTACGTACGTACGT"
    alns = []
    for r in range(24000):
        n = int(rng.integers(30, 120))
        name = b"read%d" % r + (fake if rng.random() < 0.04 else b"")
        alns.append({"sequence": bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)), "name": name, "quality": bytes(rng.integers(2, 41, n, dtype=np.uint8)),
                     "mapping_quality": 60, "identity": 1.0,
                     "path": {"mapping": [{"position": {"node_id": int(rng.integers(1, 500)), "offset": int(rng.integers(0, 5))},
                                           "edit": [{"from_length": n, "to_length": n}], "rank": 1}]}})
    raw = gamio.write_gam(alns, group=211, compress=False)
    data = _bgzf(raw)
    host, want = host_arrays(data, True, tmp_path)
    for piece_bytes in (500_000, 1_100_000):
        got, tot, n_pieces, _ = pieces_arrays(data, piece_bytes, True)
        assert n_pieces >= 2 and tot["R"] == 24000
        assert_same(got, want)


def test_an_item_longer_than_the_room_kept_for_it_is_refused_and_a_truncated_file_too(tmp_path):
    g = hc.synth_graph(seed=5, genome_len=3000, n_nodes=2000, n_paths=50)
    a = hc.synth_reads(g, 3000, seed=62, read_len=600)
    p = str(tmp_path / "s.gam")
    a.write_gam(p)
    data = open(p, "rb").read()
    with pytest.raises(N.NativeError, match="left over"):
        pieces_arrays(data, 1, tail_bytes=256)  # (a read of 600 bp is a message of ~2.5 KB)
    pieces_arrays(data, 1, tail_bytes=1 << 16)
    # the file cut after its 5th member: the last piece's walk ends inside an item
    plan = hc.gampipe_plan(data, 1)
    cut = data[:plan[5][0]]
    with pytest.raises(N.NativeError):
        pieces_arrays(cut, 1)
    with pytest.raises(N.NativeError):
        hc.gampipe_plan(data[:plan[5][0] + 100], 1)  # (ends inside a member)


def _host_final(g, p, dedup):
    ctx = hc.HcContext(g)
    parts = hc.AlnParts.read_gam(p)
    dup = parts.mark_duplicates() if dedup else None
    hb = hc.HostBatch(g, parts, skip=dup, packed=True)
    ctx.accumulate(hb)
    return ctx.finalize(), parts.n_reads, int(dup.sum()) if dedup else 0, hb.n_reads


@pytest.mark.parametrize("dedup", [False, True])
def test_the_pipeline_gives_the_host_pipelines_sums(tmp_path, dedup):
    """vgan_hc_accumulate_gam_bytes -- pieces of ~5 members on three slots, one lane and four (contexts on one GPU, piece i to context
    i mod 4, the duplicate keys handed from lane to lane) -- against host parse + duplicate marks + host flatten: the same reads kept,
    the same reads on either side of the device flatten, the same final vector."""
    g = hc.synth_graph(seed=25, genome_len=5000, n_nodes=3300, n_paths=120)
    a = hc.synth_reads(g, 60000, seed=26, read_len=150, indel_rate=0.04, softclip_rate=0.04, low_mapq_rate=0.1)
    p = str(tmp_path / "x.gam")
    a.write_gam(p)
    data = open(p, "rb").read()
    want, n_reads, n_dup, n_kept = _host_final(g, p, dedup)
    assert (n_dup > 0) == dedup
    for n_ctx, piece_bytes, slots in ((1, 120_000, 3), (4, 120_000, 2), (1, 1 << 30, 1), (3, 1, 3)):
        ctxs = [hc.HcContext(g) for _ in range(n_ctx)]
        st, ps = hc.accumulate_gam_bytes(ctxs, g, data, piece_bytes=piece_bytes, slots=slots, mark_duplicates=dedup, n_threads=4)
        assert ps["n_reads"] == n_reads and ps["n_duplicates"] == n_dup and ps["n_messages"] == 60000
        assert ps["n_pieces"] == len(hc.gampipe_plan(data, piece_bytes))
        assert st.n_out == n_kept and ps["n_device_reads"] + ps["n_host_reads"] >= n_kept and 0 < ps["n_host_reads"] < 0.2 * n_reads
        got = ctxs[0].finalize() if n_ctx == 1 else hc.reduce_contexts(ctxs)[0]
        assert np.max(np.abs(got - want) / np.abs(want)) < 1e-12, (n_ctx, piece_bytes)
        for c in ctxs:
            c.close()


def test_device_memory_follows_the_piece_not_the_file(tmp_path):
    g = hc.synth_graph(seed=25, genome_len=5000, n_nodes=3300, n_paths=120)
    held = []
    for n in (40000, 160000):
        a = hc.synth_reads(g, n, seed=27, read_len=150)
        p = str(tmp_path / ("m%d.gam" % n))
        a.write_gam(p)
        data = open(p, "rb").read()
        ctx = hc.HcContext(g)
        _, ps = hc.accumulate_gam_bytes([ctx], g, data, piece_bytes=2 << 20, slots=3)
        assert ps["n_reads"] == n and ps["n_pieces"] >= 4
        held.append(ps["device_bytes"])
        ctx.close()
    assert held[1] < 1.3 * held[0], held  # (four times the reads, the same buffers)
    assert held[1] < 3 * (8 << 20) + 3 * (2 << 20) * (1 + 3 * 2.4) + (64 << 20), held


def test_a_damaged_member_in_a_later_piece_fails_the_run_and_says_where(tmp_path):
    g = hc.synth_graph(seed=25, genome_len=5000, n_nodes=3300, n_paths=120)
    a = hc.synth_reads(g, 20000, seed=28, read_len=150)
    p = str(tmp_path / "d.gam")
    a.write_gam(p)
    data = bytearray(open(p, "rb").read())
    plan = hc.gampipe_plan(bytes(data), 200_000)
    assert len(plan) > 6
    n_err = 0
    for at in (plan[4][0] + 600, plan[5][0] + 3000, plan[6][0] + 9000):
        bad = bytearray(data)
        bad[at] ^= 0x55
        ctx = hc.HcContext(g)
        try:
            hc.accumulate_gam_bytes([ctx], g, bytes(bad), piece_bytes=200_000, slots=3)
        except N.NativeError:
            n_err += 1
        ctx.close()
    assert n_err >= 1
    with pytest.raises(N.NativeError):
        ctx = hc.HcContext(g)
        hc.accumulate_gam_bytes([ctx], g, b"not a BGZF stream at all, not even close")


def test_cli_takes_the_device_path_with_several_contexts_and_falls_back(tmp_path):
    """`vgan haplocart --gpus 0,0,0,0` on a BGZF GAM with the device front end forced on: the pieces are dealt to four contexts, the
    files written are the host pipeline's; a file the device refuses (a damaged member) goes through the host pipeline, which says what
    is wrong with it."""
    g = hc.synth_graph(seed=15, genome_len=5000, n_nodes=3400, n_paths=60)
    a = hc.synth_reads(g, 40000, seed=3, read_len=100, indel_rate=0.05, softclip_rate=0.05, low_mapq_rate=0.2)
    g.write(str(tmp_path))
    gam = str(tmp_path / "r.gam")
    a.write_gam(gam)
    exe = os.path.join(ROOT, "vgan_amd", "bin", "vgan")
    res = {}
    for tag, extra, env in (("host", [], {"VGAN_HC_DEVICE_GAM": "0"}),
                            ("dev1", [], {"VGAN_HC_DEVICE_GAM": "1", "VGAN_GAMPIPE_PIECE": "300000", "VGAN_TIMING": "1"}),
                            ("dev4", ["--gpus", "0,0,0,0"], {"VGAN_HC_DEVICE_GAM": "1", "VGAN_GAMPIPE_PIECE": "300000", "VGAN_TIMING": "1"}),
                            ("dev4keep", ["--gpus", "0,0,0,0", "--keep-duplicates"], {"VGAN_HC_DEVICE_GAM": "1", "VGAN_GAMPIPE_PIECE": "300000", "VGAN_TIMING": "1"}),
                            ("hostkeep", ["--keep-duplicates"], {"VGAN_HC_DEVICE_GAM": "0"})):
        out = str(tmp_path / (tag + ".tsv"))
        r = subprocess.run([exe, "haplocart", "-g", gam, "--hc-files", str(tmp_path), "-q", "-np", "-d", "-o", out, "-s", "s", "-t", "6"] + extra,
                           capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        if tag.startswith("dev"):
            line = [ln for ln in r.stderr.splitlines() if "device front end" in ln]
            assert line and "40000 messages" in line[0] and ("on %d lane(s)" % (4 if "4" in tag else 1)) in line[0], r.stderr[-1500:]
            assert int(line[0].split(" pieces")[0].split()[-1]) > 8
        ll = dict((ln.split("\t")[0], float(ln.split("\t")[1])) for ln in open(out + ".loglik.tsv").read().splitlines())
        res[tag] = (open(out).read().splitlines()[1], ll)
    for want, others in (("host", ("dev1", "dev4")), ("hostkeep", ("dev4keep",))):
        for other in others:
            assert res[want][0] == res[other][0]
            assert res[want][1].keys() == res[other][1].keys()
            for k, v in res[want][1].items():
                assert res[other][1][k] == pytest.approx(v, rel=1e-9)
    data = bytearray(open(gam, "rb").read())
    data[len(data) // 2] ^= 0x55
    open(gam, "wb").write(bytes(data))
    r = subprocess.run([exe, "haplocart", "-g", gam, "--hc-files", str(tmp_path), "-np", "-o", str(tmp_path / "bad.tsv"), "-s", "s"],
                       capture_output=True, text=True, env=dict(os.environ, VGAN_HC_DEVICE_GAM="1", VGAN_GAMPIPE_PIECE="300000"))
    assert "the host pipeline does" in r.stderr and r.returncode != 0  # (the host pipeline reads the same damaged member: an I/O error, said once)


def test_a_piece_whose_columns_would_wrap_the_offsets_is_refused_and_the_cli_takes_the_host_pipeline(tmp_path, monkeypatch):
    """The packed batch's offsets are 32 bits wide; the device flatten sums a chunk's alignment columns in 64 bits and refuses a chunk
    that would wrap them (VGAN_HC_DEVFLAT_MAX_COLS puts the limit where a test can reach it).  Through the library the pipeline fails
    with that reason; `vgan haplocart` clears its contexts and runs the host pipeline, which writes the host pipeline's files."""
    g = hc.synth_graph(seed=15, genome_len=5000, n_nodes=3400, n_paths=60)
    a = hc.synth_reads(g, 20000, seed=3, read_len=100)
    g.write(str(tmp_path))
    gam = str(tmp_path / "r.gam")
    a.write_gam(gam)
    exe = os.path.join(ROOT, "vgan_amd", "bin", "vgan")
    outs = {}
    for tag, env in (("host", {"VGAN_HC_DEVICE_GAM": "0"}), ("refused", {"VGAN_HC_DEVICE_GAM": "1", "VGAN_HC_DEVFLAT_MAX_COLS": "100000"})):
        out = str(tmp_path / (tag + ".tsv"))
        r = subprocess.run([exe, "haplocart", "-g", gam, "--hc-files", str(tmp_path), "-np", "-d", "-o", out, "-s", "s", "-t", "4"], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-1500:]
        assert ("the host pipeline does" in r.stderr and "32-bit offsets" in r.stderr) == (tag == "refused"), r.stderr[-1500:]
        outs[tag] = (open(out).read(), open(out + ".loglik.tsv").read())
    assert outs["host"][0] == outs["refused"][0]
    ll = [dict((ln.split("\t")[0], float(ln.split("\t")[1])) for ln in o[1].splitlines()) for o in (outs["host"], outs["refused"])]
    assert ll[0].keys() == ll[1].keys() and all(ll[1][k] == pytest.approx(v, rel=1e-9) for k, v in ll[0].items())


def test_the_code_objects_of_every_translation_unit_can_be_loaded_ahead():
    """vgan_device_preload: every VGAN_PRELOAD_* bit names kernels the runtime finds (a stale anchor would fail here, not at a run's start)."""
    L = N.lib()
    assert L.vgan_device_warmup(0) == 0
    for what in (1, 2, 4, 8, 15):
        assert L.vgan_device_preload(0, what) == 0
    assert L.vgan_device_preload(1 << 20, 15) < 0  # (no such device)



@pytest.mark.parametrize("seed", [1, 2, 3])
def test_corrupted_alignments_give_the_host_pipelines_sums(tmp_path, seed):
    """Alignments with node ids, offsets, edit lengths and strands changed at random through the HaploCart pipeline: the same reads kept and
    refused, the same final vector as host parse + host flatten."""
    from test_sb_pipe_gpu import _corrupted
    g = hc.synth_graph(seed=25, genome_len=5000, n_nodes=3300, n_paths=120)
    a0 = hc.synth_reads(g, 6000, seed=70 + seed, read_len=100, indel_rate=0.05, softclip_rate=0.05, low_mapq_rate=0.1)
    a = _corrupted(a0, g, seed, n_max=600)
    p = str(tmp_path / "c.gam")
    a.write_gam(p)
    data = open(p, "rb").read()
    want, n_reads, _, n_kept = _host_final(g, p, False)
    ctx = hc.HcContext(g)
    st, ps = hc.accumulate_gam_bytes([ctx], g, data, piece_bytes=150_000, slots=2, n_threads=4)
    assert ps["n_reads"] == n_reads and st.n_out == n_kept and ps["n_device_reads"] > 0 and ps["n_host_reads"] > 0
    got = ctx.finalize()
    assert np.max(np.abs(got - want) / np.abs(want)) < 1e-12
    ctx.close()
