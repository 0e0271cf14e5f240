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
        # (a flipped bit breaks the code stream, or yields other bytes: it must not pass for the original)
        assert rc != 0 or out[:int(size.value)].tobytes() != want
        n_bad += rc != 0
    assert n_bad >= 1
