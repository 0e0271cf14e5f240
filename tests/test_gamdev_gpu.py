"""GPU tests of the GAM front end on the device (csrc/gam_kernels.hip): byte / integer work, so everything is bit for bit what the
host pipeline (csrc/host/gam.cpp, zlib) makes of the same file."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

from vgan_amd import _native as N
from vgan_amd import haplocart as hc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gunzip_members(data):
    out, d = [], data
    while d:
        z = zlib.decompressobj(16 + zlib.MAX_WBITS)
        out.append(z.decompress(d))
        d = z.unused_data
    return b"".join(out)


def device_inflate(data):
    L = N.lib()
    buf = np.frombuffer(data, np.uint8)
    size = C.c_uint64(0)
    N.check(L.vgan_gamdev_inflate_bytes(buf.ctypes.data, len(data), None, 0, C.byref(size), None))
    out = np.zeros(max(int(size.value), 1), np.uint8)
    ms = C.c_double(0)
    N.check(L.vgan_gamdev_inflate_bytes(buf.ctypes.data, len(data), out.ctypes.data, len(out), C.byref(size), C.byref(ms)))
    return out[:int(size.value)].tobytes(), ms.value


@pytest.mark.parametrize("name", ["alignments/J2a1a1a1.gam", "alignments/two_unique.gam", "alignments/all_the_same.gam",
                                  "alignments/all_the_same_reverse.gam", "reconstruct/test_reads.gam"])
def test_device_inflate_of_the_reference_gams_equals_zlib(name):
    data = open(os.path.join(GOLD, name), "rb").read()
    got, _ = device_inflate(data)
    assert got == gunzip_members(data)


def test_device_inflate_of_a_synthetic_gam_and_of_odd_members(tmp_path):
    g = hc.synth_graph(seed=5, genome_len=3000, n_nodes=2000, n_paths=50)
    a = hc.synth_reads(g, 120000, seed=6, read_len=150)
    p = str(tmp_path / "big.gam")
    a.write_gam(p)
    data = open(p, "rb").read()
    got, ms = device_inflate(data)
    want = gunzip_members(data)
    assert len(want) > 40_000_000 and got == want
    print("device inflate: %.1f MB -> %.1f MB in %.2f ms" % (len(data) / 1e6, len(want) / 1e6, ms))

    # members written by other deflaters: stored blocks (level 0), fixed codes (tiny inputs), long runs (distance 1), random bytes
    def bgzf(chunks, level):
        out = b""
        for c in chunks:
            z = zlib.compressobj(level, zlib.DEFLATED, -15)
            payload = z.compress(c) + z.flush()
            bsize = len(payload) + 25
            out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + (bsize).to_bytes(2, "little") + payload +
                    zlib.crc32(c).to_bytes(4, "little") + len(c).to_bytes(4, "little"))
        return out
    rng = np.random.default_rng(1)
    chunks = [b"", b"a", b"abc" * 5, bytes(60000), bytes(rng.integers(0, 256, 50000, dtype=np.uint8)), b"ACGT" * 16000,
              bytes(rng.integers(65, 69, 65000, dtype=np.uint8)), want[:65280], want[1000:40000]]
    for level in (0, 1, 6, 9):
        data = bgzf(chunks, level)
        got, _ = device_inflate(data)
        assert got == b"".join(chunks), level


def test_device_inflate_of_every_period_strategy_and_many_random_members():
    """The decode loop's corners: matches of every period from 1 to 20 (those below 16 come out of registers), overlapping long matches,
    zlib's strategies (fixed codes only, run-length only, Huffman only, filtered), members that end a few bytes into a word, and a few
    hundred random members of random sizes -- a lane each, side by side in the same waves."""
    def member(c, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
        z = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        payload = z.compress(c) + z.flush()
        assert len(payload) + 25 < 65536
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + (len(payload) + 25).to_bytes(2, "little") + payload +
                zlib.crc32(c).to_bytes(4, "little") + len(c).to_bytes(4, "little"))
    rng = np.random.default_rng(7)
    chunks = []
    for period in range(1, 21):
        pat = bytes(rng.integers(65, 91, period, dtype=np.uint8))
        for total in (period + 3, 259, 777, 4001):
            chunks.append(bytes(rng.integers(0, 256, int(rng.integers(0, 9)), dtype=np.uint8)) + (pat * (total // period + 1))[:total])
    text = bytes(rng.integers(97, 101, 3000, dtype=np.uint8))
    chunks += [text * 15, (text[:300] + b"#") * 100, bytes(40000), b"x" * 65000]
    for _ in range(300):
        n = int(rng.integers(0, 3000))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            c = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif kind == 1:
            c = bytes(rng.integers(65, 69, n, dtype=np.uint8))
        elif kind == 2:
            c = bytes(np.repeat(rng.integers(0, 256, n // 7 + 1, dtype=np.uint8), rng.integers(1, 14, n // 7 + 1)))[:n]
        else:
            w = bytes(rng.integers(97, 123, 40, dtype=np.uint8))
            c = b"".join(w[int(a):int(a) + int(b)] for a, b in zip(rng.integers(0, 30, n // 5 + 1), rng.integers(1, 10, n // 5 + 1)))[:n]
        chunks.append(c)
    for level, strategy in ((6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_RLE),
                            (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_FILTERED), (0, zlib.Z_DEFAULT_STRATEGY)):
        order = rng.permutation(len(chunks))
        data = b"".join(member(chunks[i], level, strategy) for i in order)
        got, _ = device_inflate(data)
        assert got == b"".join(chunks[i] for i in order), (level, strategy)


def test_a_damaged_member_is_refused(tmp_path):
    g = hc.synth_graph(seed=5, genome_len=3000, n_nodes=2000, n_paths=50)
    a = hc.synth_reads(g, 3000, seed=7, read_len=150)
    p = str(tmp_path / "small.gam")
    a.write_gam(p)
    good = open(p, "rb").read()
    want = gunzip_members(good)
    n_bad = 0
    for at in (400, 2000, 9000, 20000):  # inside DEFLATE payloads of the first members
        data = bytearray(good)
        data[at] ^= 0x55
        buf = np.frombuffer(bytes(data), np.uint8)
        size = C.c_uint64(0)
        out = np.zeros(len(want) + 1024, np.uint8)
        rc = N.lib().vgan_gamdev_inflate_bytes(buf.ctypes.data, len(data), out.ctypes.data, len(out), C.byref(size), None)
        # (a flipped bit breaks the code stream, or yields other bytes -- which the member's CRC-32 does not fit: refused either way,
        # as zlib / libdeflate refuse it on the host)
        assert rc != 0, at
        n_bad += rc != 0
    assert n_bad == 4
    # the trailer's CRC-32 itself: the bytes inflate, the member is refused all the same
    import struct
    bsize = struct.unpack_from("<H", good, 16)[0] + 1
    data = bytearray(good)
    data[bsize - 8] ^= 0x01
    buf = np.frombuffer(bytes(data), np.uint8)
    size = C.c_uint64(0)
    out = np.zeros(len(want) + 1024, np.uint8)
    assert N.lib().vgan_gamdev_inflate_bytes(buf.ctypes.data, len(data), out.ctypes.data, len(out), C.byref(size), None) != 0
    assert "code 6" in (N.lib().vgan_last_error() or b"").decode()


# ---------------------------------------------------------------------------------------------- framing + parsing on the device
_DT = {0: np.uint32, 1: np.uint32, 2: np.uint32, 3: np.uint32, 4: np.uint32, 5: np.int32, 6: np.int32, 7: np.int32, 8: np.uint8, 9: np.uint8,
       10: np.uint8, 11: np.uint8, 12: np.int64, 13: np.int64, 14: np.uint8, 15: np.uint64, 16: np.uint32}


class GamDev:
    def __init__(self):
        self._h = N.vp()
        N.check(N.lib().vgan_gamdev_create(0, None, C.byref(self._h)))

    def parse(self, data, keep_unmapped=False):
        buf = np.frombuffer(data, np.uint8)
        N.check(N.lib().vgan_gamdev_parse(self._h, buf.ctypes.data, len(data), int(keep_unmapped)))
        sizes = np.zeros(8, np.uint64)
        ms = np.zeros(4)
        N.check(N.lib().vgan_gamdev_sizes(self._h, sizes.ctypes.data, ms.ctypes.data))
        self.sizes = dict(zip(("inflated", "messages", "R", "M", "E", "S", "Q", "reanchored"), (int(x) for x in sizes[:8])))
        self.ms = dict(zip(("upload", "inflate", "frame", "parse"), ms))
        return self

    def array(self, which):
        z = self.sizes
        n = {0: z["R"] + 1, 1: z["R"] + 1, 2: z["M"] + 1, 3: z["E"] + 1, 4: z["M"], 5: z["M"], 6: z["R"], 7: z["E"], 8: z["R"], 9: z["M"], 10: z["S"], 11: z["Q"],
             12: z["R"], 13: z["R"], 14: z["inflated"], 15: z["messages"], 16: z["messages"]}[which]
        out = np.zeros(max(n, 1), _DT[which])
        N.check(N.lib().vgan_gamdev_download(self._h, which, out.ctypes.data))
        return out[:n]

    def close(self):
        if self._h:
            N.lib().vgan_gamdev_free(self._h)
            self._h = None


def host_slice(a):
    """What vgan_hc_devflat_run's narrowing makes of a host-parsed alignment set (csrc/hc_flatten_kernels.hip: DfSlice)."""
    x = a.arrays()
    node = x["m_node"]
    off = x["m_offset"]
    return {0: x["map_off"].astype(np.uint32), 1: x["qual_off"].astype(np.uint32), 2: x["edit_off"].astype(np.uint32), 3: x["e_seq_off"].astype(np.uint32),
            4: np.where((node < 0) | (node > 0xFFFFFFFE), 0xFFFFFFFF, node).astype(np.uint32),
            5: np.where((off != off.astype(np.int32)) | (off.astype(np.int32) == -2**31), -2**31, off).astype(np.int32),
            6: x["mapq"].astype(np.int32), 7: np.where((x["e_from"] == x["e_to"]) & (x["e_from"] >= 0), x["e_from"], -1).astype(np.int32),
            8: (x["identity"] < 1e-10).astype(np.uint8), 9: x["m_rev"].astype(np.uint8), 10: np.array(x["e_seq"]), 11: np.array(x["qual"]),
            12: np.array([node[x["map_off"][r]] if x["map_off"][r + 1] > x["map_off"][r] else -1 for r in range(x["n_reads"])], np.int64),
            13: np.array([off[x["map_off"][r]] if x["map_off"][r + 1] > x["map_off"][r] else 0 for r in range(x["n_reads"])], np.int64)}


def check_against_host(gd, data, keep_unmapped):
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".gam") as f:
        f.write(data)
        f.flush()
        a = hc.AlnSet.read_gam(f.name, keep_unmapped=keep_unmapped)
    gd.parse(data, keep_unmapped)
    want = host_slice(a)
    assert gd.sizes["R"] == a.n_reads
    for which, w in want.items():
        got = gd.array(which)
        assert got.shape == w.shape and np.array_equal(got, w), which
    return a.n_reads


@pytest.mark.parametrize("name", ["alignments/J2a1a1a1.gam", "alignments/two_unique.gam", "alignments/all_the_same.gam",
                                  "alignments/all_the_same_reverse.gam", "reconstruct/test_reads.gam"])
def test_device_framing_and_parsing_of_the_reference_gams(name):
    data = open(os.path.join(GOLD, name), "rb").read()
    gd = GamDev()
    for keep in (False, True):
        assert check_against_host(gd, data, keep) > 0
    gd.close()


def test_device_front_end_on_synthetic_files(tmp_path):
    """Files of several segments (the walks must meet on the groups' tags), groups of odd sizes, reads with indels and soft clips
    (edits that are neither match nor substitution), unmapped reads."""
    g = hc.synth_graph(seed=5, genome_len=3000, n_nodes=2000, n_paths=50)
    gd = GamDev()
    for n, rl, group in ((40000, 150, 512), (3000, 600, 7), (9000, 75, 1000)):
        a = hc.synth_reads(g, n, seed=6 + n, read_len=rl, indel_rate=0.1, softclip_rate=0.1)
        p = str(tmp_path / ("s%d.gam" % n))
        a.write_gam(p, group_size=group)
        data = open(p, "rb").read()
        assert a.n_reads - 5 <= check_against_host(gd, data, False) <= a.n_reads  # (a read without a single matching base has identity 0: dropped)
        assert check_against_host(gd, data, True) == a.n_reads
        print("device front end %d reads: %s" % (n, {k: round(v, 2) for k, v in gd.ms.items()}))
    # the message list itself: offsets ascend, lengths add up with the framing bytes to the inflated size
    off, ln = gd.array(15), gd.array(16)
    assert np.all(off[1:] > off[:-1]) and int(off[-1] + ln[-1]) == gd.sizes["inflated"]
    gd.close()


def test_open_is_create_plus_parse_and_the_pieces_of_the_upload_do_not_show(tmp_path, monkeypatch):
    """vgan_gamdev_open (member index before the first HIP call) against vgan_gamdev_parse; the file sent up in pieces of 100 kB (each
    piece's members inflated while the next is copied) against one piece."""
    g = hc.synth_graph(seed=5, genome_len=3000, n_nodes=2000, n_paths=50)
    a = hc.synth_reads(g, 20000, seed=77, read_len=150, indel_rate=0.05)
    p = str(tmp_path / "s.gam")
    a.write_gam(p)
    data = open(p, "rb").read()
    gd = GamDev()
    n = check_against_host(gd, data, False)
    want = {w: gd.array(w) for w in range(17)}
    gd.close()
    monkeypatch.setenv("VGAN_GAMDEV_PIECE", "100000")
    o = hc.GamDevice.open(data)
    assert o.sizes["reads"] == n
    g2 = GamDev()
    g2.parse(data)
    for w, v in want.items():
        assert np.array_equal(g2.array(w), v), w
    g2.close()
    o.close()
    with pytest.raises(N.NativeError):
        hc.GamDevice.open(b"not a BGZF stream at all, not even close")


def test_an_empty_file_and_a_stream_without_tags(tmp_path):
    import gamio
    gd = GamDev()
    empty = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00"  # the BGZF end-of-file member
    gd.parse(empty)
    assert gd.sizes["R"] == 0 and gd.sizes["messages"] == 0
    gd.close()


def test_device_parse_feeds_the_device_flatten_like_the_host_parser(tmp_path):
    """file bytes -> vgan_gamdev_parse -> vgan_hc_devflat_run_gamdev must give, word for word, the packed batch (and the mask of
    the reads left to the host) that host parse -> vgan_hc_devflat_run gives; and the same final vector."""
    g = hc.synth_graph(seed=15, genome_len=4000, n_nodes=2600, n_paths=80)
    a = hc.synth_reads(g, 30000, seed=16, read_len=150, indel_rate=0.05, softclip_rate=0.05)
    p = str(tmp_path / "x.gam")
    a.write_gam(p)
    data = open(p, "rb").read()
    ctx = hc.HcContext(g)
    df = hc.DeviceFlatten(ctx, g)
    parts = hc.AlnParts.read_gam(p)
    want = df.run(parts)
    w = want.download()
    w_mask = np.array(want.host_mask)
    ctx.accumulate(want)
    f_host = ctx.finalize()
    gdev = hc.GamDevice().parse(data)
    assert gdev.sizes["reads"] == parts.n_reads
    got = df.run_gamdev(gdev)
    gg = got.download()
    for name in ("rhdr", "srec", "crec", "qualp", "read_src"):
        assert np.array_equal(gg[name], w[name]), name
    assert np.array_equal(np.array(got.host_mask), w_mask) and 0 < w_mask.sum() < parts.n_reads
    for f in ("n_in", "n_out", "n_unmapped", "n_segments", "n_cols"):
        assert getattr(got.stats, f) == getattr(want.stats, f), f
    ctx.reset()
    ctx.accumulate(got)
    f_dev = ctx.finalize()  # (the same words in, the same kernels: equal up to the order of the window flushes' fp64 atomics)
    assert np.max(np.abs(f_dev - f_host) / np.abs(f_host)) < 1e-13
    # with duplicate marks
    dup = parts.mark_duplicates()
    w2 = df.run(parts, skip=dup).download()
    g2 = df.run_gamdev(gdev, skip=dup).download()
    for name in ("rhdr", "srec", "crec", "read_src"):
        assert np.array_equal(g2[name], w2[name]), name
    df.close()
    gdev.close()


def _bgzf(raw, chunk=60000):
    out = b""
    for i in range(0, len(raw), chunk):
        c = raw[i:i + chunk]
        z = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = z.compress(c) + z.flush()
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + (len(payload) + 25).to_bytes(2, "little") + payload +
                zlib.crc32(c).to_bytes(4, "little") + len(c).to_bytes(4, "little"))
    return out + b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00"


def test_messages_with_a_damaged_byte_are_taken_or_refused_as_the_host_parser_does(tmp_path):
    """One byte of one Alignment message changed (the framing left whole), 150 times: the device's wire walk accepts exactly the files
    the host parser accepts, with the same arrays, and refuses the others."""
    import gamio
    g = hc.synth_graph(seed=5, genome_len=3000, n_nodes=2000, n_paths=50)
    a = hc.synth_reads(g, 60, seed=8, read_len=120, indel_rate=0.1, softclip_rate=0.1)
    p = str(tmp_path / "s.gam")
    a.write_gam(p, group_size=16)
    raw = gunzip_members(open(p, "rb").read())
    spans, i = [], 0
    while i < len(raw):  # {count, count x (length, bytes)}: the bodies of the alignment messages
        cnt, i = gamio._varint(raw, i)
        for j in range(cnt):
            ln, i = gamio._varint(raw, i)
            if j > 0:
                spans.append((i, ln))
            i += ln
    assert len(spans) == 60
    rng = np.random.default_rng(11)
    gd = GamDev()
    taken = refused = 0
    for trial in range(150):
        at, ln = spans[int(rng.integers(len(spans)))]
        k = at + int(rng.integers(ln))
        bad = bytearray(raw)
        bad[k] ^= int(rng.integers(1, 256))
        data = _bgzf(bytes(bad))
        f = str(tmp_path / "bad.gam")
        open(f, "wb").write(data)
        try:
            hc.AlnSet.read_gam(f, keep_unmapped=True)
            host_ok = True
        except N.NativeError:
            host_ok = False
        if host_ok:
            assert check_against_host(gd, data, True) <= 60, (trial, k)
            taken += 1
        else:
            with pytest.raises(N.NativeError):
                gd.parse(data, True)
            refused += 1
    assert taken > 20 and refused > 20, (taken, refused)
    gd.close()


def test_tag_like_bytes_inside_messages_never_give_another_framing(tmp_path):
    """Read names and sequences that hold the group tag's bytes (03 'G' 'A' 'M', with a plausible count in front): a segment whose first
    tag-like bytes are not a group's start is found out by the walk in front of it, which does not arrive there, and takes the next
    ones; the file is framed as the host frames it, array for array."""
    import gamio
    rng = np.random.default_rng(21)
    fake = bytes([5, 3]) + b"GAM" + bytes([40]) + b"\x0a\x10ACGTACGTACGTACGT"  # count, tag, a length, the start of an Alignment
    taken = refused = 0
    for trial in range(6):
        alns = []
        for r in range(16000):
            node = int(rng.integers(1, 500))
            n = int(rng.integers(30, 120))
            name = b"read%d" % r
            seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))
            if trial and rng.random() < 0.02 * trial:
                name = name + fake
            if trial > 2 and rng.random() < 0.01:
                seq = seq + fake
            alns.append({"sequence": seq, "name": name, "quality": bytes(rng.integers(2, 41, n, dtype=np.uint8)), "mapping_quality": 60, "identity": 1.0,
                         "path": {"mapping": [{"position": {"node_id": node, "offset": int(rng.integers(0, 5))},
                                               "edit": [{"from_length": n, "to_length": n}], "rank": 1}]}})
        raw = gamio.write_gam(alns, group=int(rng.integers(3, 700)), compress=False)
        assert len(raw) > 2 * (1 << 20)  # several segments
        data = _bgzf(raw)
        gd = GamDev()
        assert check_against_host(gd, data, True) == 16000
        refused += gd.sizes["reanchored"]
        taken += 1
        gd.close()
    assert taken == 6 and refused > 0, (taken, refused)  # (tag-like bytes were met first in some segment, and given up)
    print("tag-like bytes: %d files framed, %d false tags given up" % (taken, refused))


def test_the_mask_call_back_comes_once_with_the_final_mask(tmp_path):
    """vgan_hc_devflat_run_gamdev_cb: the call-back runs on the calling thread when host_mask is final (before the write pass), once;
    the batch is the one vgan_hc_devflat_run_gamdev gives."""
    g = hc.synth_graph(seed=25, genome_len=5000, n_nodes=3300, n_paths=120)
    a = hc.synth_reads(g, 20000, seed=27, read_len=150, indel_rate=0.04, softclip_rate=0.04)
    p = str(tmp_path / "x.gam")
    a.write_gam(p)
    ctx = hc.HcContext(g)
    df = hc.DeviceFlatten(ctx, g)
    gd = hc.GamDevice().parse(open(p, "rb").read())
    want = df.run_gamdev(gd)
    want_arrays, want_mask = want.download(), np.array(want.host_mask)
    n = gd.sizes["reads"]
    mask = np.zeros(n, np.uint8)
    seen = []
    CB = C.CFUNCTYPE(None, C.c_void_p)
    cb = CB(lambda user: seen.append(mask.copy()))
    pk, st = N.HcPackedView(), N.FlattenStats()
    N.check(N.lib().vgan_hc_devflat_run_gamdev_cb(df._h, gd._h, None, 0, 0, C.byref(pk), mask.ctypes.data, C.byref(st), C.cast(cb, C.c_void_p), None))
    assert len(seen) == 1 and np.array_equal(seen[0], want_mask) and np.array_equal(mask, want_mask) and 0 < want_mask.sum() < n
    got = hc.DeviceFlatten.Result(pk, mask, st).download()
    for name in ("rhdr", "srec", "crec", "read_src"):
        assert np.array_equal(got[name], want_arrays[name]), name
    df.close()
    gd.close()


def test_whole_chain_on_the_device_against_the_host_pipeline(tmp_path):
    """What `vgan haplocart` does with a BGZF GAM when the front end runs on the device: parse, duplicate marks, flatten, segment
    kernel -- and the reads the device flatten leaves (indels, soft clips) handed back as their messages, parsed and flattened on the
    host.  Against the host pipeline on the same file: the same marks, the same reads on either side, the same final vector."""
    g = hc.synth_graph(seed=25, genome_len=5000, n_nodes=3300, n_paths=120)
    a = hc.synth_reads(g, 50000, seed=26, read_len=150, indel_rate=0.04, softclip_rate=0.04)
    p = str(tmp_path / "x.gam")
    a.write_gam(p)
    data = open(p, "rb").read()
    ctx = hc.HcContext(g)
    df = hc.DeviceFlatten(ctx, g)
    # ---- the host pipeline
    parts = hc.AlnParts.read_gam(p)
    dup = parts.mark_duplicates()
    assert 0 < dup.sum() < parts.n_reads
    hb = hc.HostBatch(g, parts, skip=dup, packed=True)
    ctx.accumulate(hb)
    want = ctx.finalize()
    # ---- the device's
    gd = hc.GamDevice().parse(data)
    assert gd.mark_duplicates() == int(dup.sum())
    res = df.run_gamdev(gd, device_marks=True)
    want_dev = df.run(parts, skip=dup)
    assert np.array_equal(np.array(res.host_mask), np.array(want_dev.host_mask))
    ctx.reset()
    ctx.accumulate(res)
    n_left = int(np.array(res.host_mask).sum())
    assert 0 < n_left < 0.2 * parts.n_reads
    left = gd.picked_parts(res.host_mask)
    assert left.n_reads == n_left
    hb2 = hc.HostBatch(g, left, packed=True)
    assert hb2.n_reads + res.n_reads == hb.n_reads
    ctx.accumulate(hb2)
    got = ctx.finalize()
    assert np.max(np.abs(got - want) / np.abs(want)) < 1e-12
    # ---- device memory given back early (the CLI does, beside the kernels that follow): the file's bytes change nothing; without the
    # inflated bytes the flatten still runs on the parsed arrays, the hand-back of messages says why it cannot, a new parse brings all back
    gd.drop_bytes()
    assert gd.picked_parts(res.host_mask).n_reads == n_left
    gd.drop_bytes(inflated=True)
    again = df.run_gamdev(gd, device_marks=True)
    assert np.array_equal(np.array(again.host_mask), np.array(res.host_mask)) and again.n_reads == res.n_reads
    with pytest.raises(Exception, match="given back"):
        gd.picked_parts(res.host_mask)
    gd.parse(data)
    assert gd.mark_duplicates() == int(dup.sum())
    assert gd.picked_parts(res.host_mask).n_reads == n_left
    df.close()
    gd.close()
