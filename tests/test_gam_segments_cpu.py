"""The BGZF segment pipeline of the GAM reader (csrc/host/gam.cpp: speculative per-segment framing, stitched by one serial
walk) against the plain serial walk of the same bytes: every way a group header, a length or a message can lie across a
segment boundary, streams with and without type tags, bytes that look like a tag inside a message, messages larger than a
buffer's headroom, and damaged streams (the two paths must fail or succeed together, with the same reads)."""
import os
import random
import struct
import zlib

import numpy as np
import pytest

import gamio
from vgan_amd import _native as N
from vgan_amd import haplocart as hc


def bgzf(plain, block):
    """BGZF members of `block` payload bytes each + the end-of-file block."""
    out = bytearray()
    for off in range(0, len(plain), block):
        chunk = plain[off:off + block]
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        bsize = len(body) + 26
        assert bsize <= 65536
        out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", bsize - 1) + body
        out += struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return bytes(out)


def varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def stream(msgs, group, tagged=True, mixed=False, rng=None):
    out = bytearray()
    i = 0
    while i < len(msgs):
        k = group if not rng else rng.randrange(1, 2 * group)
        part = msgs[i:i + k]
        i += k
        tag = tagged if not mixed else rng.random() < 0.5
        out += varint(len(part) + (1 if tag else 0))
        if tag:
            out += b"\x03GAM"
        for m in part:
            out += varint(len(m)) + m
    return bytes(out)


def messages(n, seed, tricky=False, big=None):
    rng = random.Random(seed)
    g = hc.synth_graph(seed=3, genome_len=700, n_nodes=480, n_paths=20)
    a = hc.synth_reads(g, n, seed=seed, read_len=60, low_mapq_rate=0.2)
    alns = gamio.read_gam(_bytes_of(a))
    out = []
    for j, al in enumerate(alns):
        if tricky and j % 3 == 0:
            al["name"] = b"r\x03GAM" + bytes([rng.randrange(1, 120)]) + b"\x03GAM"  # looks like a group header inside a message
        if tricky and j % 7 == 0:
            al["quality"] = (b"\x02\x03GAM" * 40)[:len(al["sequence"])]
        if big and j == big[0]:
            al["sequence"] = b"ACGT" * (big[1] // 4)
            al["quality"] = b"\x1e" * len(al["sequence"])
        out.append(gamio.enc_alignment(al))
    return out


def _bytes_of(a):
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.gam")
        a.write_gam(p)
        return open(p, "rb").read()


def counts():
    import ctypes as C
    out = (C.c_int64 * 5)()
    N.lib().vgan_gam_decode_counts(out)
    return np.array(list(out))


def same(a, b):
    aa, bb = a.arrays(), b.arrays()
    for k, v in aa.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, bb[k]), k


@pytest.fixture
def seg_env():
    keep = {k: os.environ.get(k) for k in ("VGAN_GAM_SEG_BLOCKS", "VGAN_GAM_THREADS")}

    def use(blocks, threads=4):
        os.environ["VGAN_GAM_SEG_BLOCKS"] = str(blocks)
        os.environ["VGAN_GAM_THREADS"] = str(threads)
    yield use
    for k, v in keep.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("group,tagged,mixed", [(1, True, False), (3, True, False), (40, True, False), (512, True, False),
                                                 (7, False, False), (5, True, True)])
def test_segmented_framing_equals_the_serial_walk(group, tagged, mixed, seg_env):
    msgs = messages(900, seed=group, tricky=True)
    plain = stream(msgs, group, tagged=tagged, mixed=mixed, rng=random.Random(group) if mixed else None)
    want = hc.AlnSet.parse_gam(plain)  # plain bytes: the serial walk
    assert want.n_reads > 600
    c0 = counts()
    for block in (97, 700, 5000, 65280):
        blob = bgzf(plain, block)
        for seg_blocks in (1, 2, 5):
            seg_env(seg_blocks, threads=1 + seg_blocks)
            same(want, hc.AlnSet.parse_gam(blob))
    c = counts() - c0
    assert c[0] > 100 and c[3] > 0  # many segments, some too short to hold a group header
    if tagged and not mixed:
        assert c[1] > (20 if group < 100 else 5), c  # taken from the group the segment's own walk started in
    if group in (1, 3) and tagged:
        assert c[2] > 0, c  # a tag inside a message sent the own walk astray: taken from a later group
    if not tagged:
        assert c[1] == 0 and c[0] == c[2] + c[3]  # nothing to anchor on but the tags inside messages


def test_stream_chunks_through_small_segments(tmp_path, seg_env):
    msgs = messages(3000, seed=5)
    plain = stream(msgs, 100)
    f = str(tmp_path / "s.gam")
    open(f, "wb").write(bgzf(plain, 3000))
    seg_env(2, threads=3)
    want = hc.AlnSet.parse_gam(plain)
    base, names = 0, []
    for chunk in hc.GamStream(f).chunks(500):
        assert chunk.base == base
        base += chunk.n_reads
        merged = chunk.merge()
        names.append(merged.arrays()["seq"].copy())
    assert base == want.n_reads
    assert np.array_equal(np.concatenate(names), want.arrays()["seq"])


def test_a_message_larger_than_the_headroom(seg_env):
    # one segment of one block has 32 KB of headroom in front of its buffer: a 300 KB read crosses several segments
    msgs = messages(60, seed=9, big=(20, 300000))
    plain = stream(msgs, 16)
    want = hc.AlnSet.parse_gam(plain)
    assert want.arrays()["seq_off"][-1] > 300000
    c0 = counts()
    for seg_blocks in (1, 3):
        seg_env(seg_blocks)
        same(want, hc.AlnSet.parse_gam(bgzf(plain, 4000)))
        same(want, hc.AlnSet.parse_gam(bgzf(plain, 65280)))
    assert (counts() - c0)[4] >= 4


def test_damaged_streams_fail_or_pass_together(seg_env):
    msgs = messages(300, seed=11, tricky=True)
    plain = stream(msgs, 25)
    rng = random.Random(3)
    n_err = 0
    for trial in range(150):
        data = bytearray(plain)
        kind = trial % 3
        if kind == 0:
            data = data[: rng.randrange(1, len(data))]
        elif kind == 1:
            for _ in range(rng.randrange(1, 4)):
                data[rng.randrange(len(data))] = rng.randrange(256)
        else:
            i = rng.randrange(len(data))
            data[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 6)))
        data = bytes(data)
        try:
            want = hc.AlnSet.parse_gam(data)
        except N.NativeError:
            want = None
            n_err += 1
        seg_env(1 + trial % 3, threads=3)
        blob = bgzf(data, 500 + 37 * (trial % 11))
        if want is None:
            with pytest.raises(N.NativeError):
                hc.AlnSet.parse_gam(blob)
        else:
            same(want, hc.AlnSet.parse_gam(blob))
    assert n_err > 30
    # a damaged block is the gzip error, wherever it lies
    blob = bytearray(bgzf(plain, 800))
    blob[len(blob) // 2] ^= 0x55
    seg_env(2)
    with pytest.raises(N.NativeError):
        hc.AlnSet.parse_gam(bytes(blob))


def test_default_segments_on_a_file_of_many(tmp_path):
    g = hc.synth_graph(seed=4, genome_len=900, n_nodes=620, n_paths=30)
    a = hc.synth_reads(g, 60000, seed=2, read_len=100)
    f = str(tmp_path / "m.gam")
    a.write_gam(f)
    c0 = counts()
    b = hc.AlnSet.read_gam(f)
    c = counts() - c0
    assert c[0] >= 4 and c[1] >= c[0] - 1 and c[4] == 0, c  # every segment but the first taken from its own walk
    want = hc.AlnSet.parse_gam(gamio.gunzip_all(open(f, "rb").read()))  # plain bytes: the serial walk
    assert want.n_reads > 59000
    same(want, b)
