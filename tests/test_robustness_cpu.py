"""Robustness of the host front end: corrupted / truncated GAM input must produce an error code, never a crash or
an out-of-bounds read (this file is also what the AddressSanitizer build of the host sources runs)."""
import os
import random

import numpy as np
import pytest

import gamio
from vgan_amd import _native as N
from vgan_amd import haplocart as hc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_truncated_and_corrupted_gam_streams():
    alns = gamio.read_gam(os.path.join(ROOT, "tests/golden/alignments/J2a1a1a1.gam"))
    plain = gamio.write_gam(alns, compress=False)
    ok = hc.AlnSet.parse_gam(plain)
    assert ok.n_reads == len(alns)
    rng = random.Random(7)
    n_err = 0
    for trial in range(300):
        data = bytearray(plain)
        kind = trial % 3
        if kind == 0:
            data = data[: rng.randrange(1, len(data))]
        elif kind == 1:
            for _ in range(rng.randrange(1, 8)):
                data[rng.randrange(len(data))] = rng.randrange(256)
        else:
            i = rng.randrange(len(data))
            data[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 6)))
        try:
            a = hc.AlnSet.parse_gam(bytes(data))
            arr = a.arrays()  # whatever was parsed must be internally consistent
            assert arr["seq_off"][-1] == len(arr["seq"]) and arr["map_off"][-1] == len(arr["m_node"])
            assert arr["edit_off"][-1] == len(arr["e_from"]) and arr["e_seq_off"][-1] == len(arr["e_seq"])
        except N.NativeError as e:
            assert e.code == N.VGAN_EIO
            n_err += 1
    assert n_err > 50


def test_corrupted_gzip_and_bgzf_roundtrip(tmp_path):
    g = hc.synth_graph(seed=4, genome_len=900, n_nodes=620, n_paths=30)
    a = hc.synth_reads(g, 3000, seed=1, read_len=120)
    p = str(tmp_path / "x.gam")
    a.write_gam(p)
    raw = open(p, "rb").read()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC"  # BGZF, as vg writes GAM
    assert raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")  # BGZF EOF block
    b = hc.AlnSet.read_gam(p)
    for k, v in a.arrays().items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, b.arrays()[k]), k
    assert gamio.read_gam(raw)[5]["sequence"] == bytes(a.arrays()["seq"][a.arrays()["seq_off"][5]:a.arrays()["seq_off"][6]])
    bad = bytearray(raw)
    bad[len(bad) // 2] ^= 0xFF
    with pytest.raises(N.NativeError):
        hc.AlnSet.parse_gam(bytes(bad))
    with pytest.raises(N.NativeError):
        hc.AlnSet.parse_gam(raw[: len(raw) // 3])
    # plain (non-BGZF) gzip members are still accepted
    import gzip
    plain = gamio.gunzip_all(raw)
    c = hc.AlnSet.parse_gam(gzip.compress(plain[: len(plain)]))
    assert c.n_reads == a.n_reads


def test_flatten_rejects_nonsense_without_crashing():
    g = hc.synth_graph(seed=4, genome_len=900, n_nodes=620, n_paths=30)
    a = hc.synth_reads(g, 200, seed=1, read_len=80)
    arr = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in a.arrays().items()}
    rng = np.random.default_rng(5)
    for _ in range(60):  # corrupt mappings / edits at random
        i = rng.integers(len(arr["m_node"]))
        arr["m_node"][i] = rng.integers(0, 5000)
        j = rng.integers(len(arr["e_from"]))
        arr["e_from"][j] = rng.integers(0, 50)
        arr["e_to"][rng.integers(len(arr["e_to"]))] = rng.integers(0, 50)
        arr["m_offset"][rng.integers(len(arr["m_offset"]))] = rng.integers(0, 12)
    import orc
    import util
    oa = orc.AlnSet.from_arrays(**arr)
    v = N.AlnSetView(oa.n_reads, *[getattr(oa, k).ctypes.data for k in ("seq_off", "seq", "qual_off", "qual", "mapq", "identity")],
                     None, None, *[getattr(oa, k).ctypes.data for k in ("map_off", "m_node", "m_offset", "m_rev", "edit_off",
                                                                        "e_from", "e_to", "e_seq_off", "e_seq")])
    h = N.vp()
    N.check(N.lib().vgan_aln_from_arrays(v, h))
    a2 = hc.AlnSet(h)
    b = hc.HostBatch(g, a2)
    assert b.stats.n_out + b.stats.n_bad + b.stats.n_unmapped == 200 and b.stats.n_bad > 0
    # the oracle rejects exactly the same reads
    og = util.orc_graph_from_product(g)
    _, _, bad = orc.hc_run(og, oa, n_threads=2, faithful=False)
    assert bad == b.stats.n_bad
    from vgan_amd import euka as ek
    from vgan_amd import soibean as sb
    e = ek.EukaHostBatch(g, a2)
    s = sb.SbHostBatch(g, a2)
    assert e.stats.n_out + e.stats.n_bad == 200 and s.stats.n_out + s.stats.n_bad == 200


def _mutate(b, rng):
    b = bytearray(b)
    mode = rng.randrange(5)
    if mode == 0 and b:
        del b[rng.randrange(len(b)):]
    elif mode == 1 and b:
        for _ in range(rng.randrange(1, 30)):
            b[rng.randrange(len(b))] = rng.randrange(256)
    elif mode == 2:
        b = b[:len(b) // 3] + bytes(rng.randrange(32, 127) for _ in range(80)) + b[len(b) // 3:]
    elif mode == 3:
        b = bytearray(b.replace(b"\n", b"\r\n"))
    else:
        b = bytearray(b.replace(b"\t", b" ").replace(b"1", b"9999999999"))
    return bytes(b)


def test_table_and_sidecar_loaders_survive_corrupt_files(tmp_path, golden_dir):
    """Corrupt / truncated euka tables, damage profiles, GFA and HaploCart sidecars are either loaded (tolerant parsing,
    as the reference's whitespace tokenisers) or rejected with an error -- never a crash (a malformed *.bins line used to
    read one token past the end)."""
    import gzip
    import random
    import shutil
    from vgan_amd import euka as ek
    from vgan_amd import haplocart as hc
    rng = random.Random(7)
    srcs = {"clade": os.path.join(golden_dir, "euka_dir/euka_db.clade"), "bins": os.path.join(golden_dir, "euka_dir/euka_db.bins"),
            "p5": os.path.join(golden_dir, "damageProfiles/dhigh5p.prof"), "p3": os.path.join(golden_dir, "damageProfiles/dhigh3p.prof")}
    loaded = rejected = 0
    for it in range(150):
        d = tmp_path / ("e%d" % it)
        d.mkdir()
        paths = {k: str(d / k) for k in srcs}
        for k, p in srcs.items():
            shutil.copy(p, paths[k])
        k = rng.choice(list(srcs))
        open(paths[k], "wb").write(_mutate(open(paths[k], "rb").read(), rng))
        try:
            if k in ("clade", "bins"):
                ek.EukaDb.load(paths["clade"], paths["bins"])
            else:
                ek.Damage.load(paths["p5"], paths["p3"])
            loaded += 1
        except Exception:
            rejected += 1
        shutil.rmtree(d)
    assert loaded > 10 and rejected > 10
    # a truncated bins line (name + 2 of 3 fields)
    bad = tmp_path / "trunc.bins"
    lines = open(srcs["bins"]).read().splitlines()
    lines[0] = "\t".join(lines[0].split()[:3])
    bad.write_text("\n".join(lines) + "\n")
    ek.EukaDb.load(srcs["clade"], str(bad))
    # HaploCart graph directory
    g = hc.synth_graph(seed=5, genome_len=300, n_nodes=200, n_paths=70)
    base = tmp_path / "hc"
    base.mkdir()
    g.write(str(base))
    files = os.listdir(base)
    for it in range(120):
        d = tmp_path / ("g%d" % it)
        shutil.copytree(base, d)
        f = rng.choice(files)
        p = str(d / f)
        data = open(p, "rb").read()
        if f.endswith(".gz"):
            data = gzip.decompress(data)
            os.remove(p)
            p = p[:-3]
        open(p, "wb").write(_mutate(data, rng))
        try:
            hc.Graph.load(str(d / "graph.gfa"), str(d))
            loaded += 1
        except Exception:
            rejected += 1
        shutil.rmtree(d)
    # an absurd mappability interval is an error, not an allocation of gigabytes
    d = tmp_path / "gm"
    shutil.copytree(base, d)
    (d / "mappability.tsv").write_text("chrM\t0\t4000000000\t1.0\n")
    with pytest.raises(Exception):
        hc.Graph.load(str(d / "graph.gfa"), str(d))


def test_gam_without_reads(tmp_path):
    """A GAM without reads is the 28-byte BGZF end-of-file block alone (what vg writes for an empty result): every reader
    returns zero reads (the empty deflate member used to be reported as a corrupt stream)."""
    g = hc.synth_graph(seed=4, genome_len=900, n_nodes=620, n_paths=30)
    a = hc.synth_reads(g, 30, seed=1, read_len=80)
    p = str(tmp_path / "none.gam")
    a.without(np.ones(a.n_reads, np.uint8)).write_gam(p)
    raw = open(p, "rb").read()
    assert raw == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    assert hc.AlnSet.read_gam(p).n_reads == 0 and hc.AlnSet.parse_gam(raw).n_reads == 0 and hc.AlnSet.parse_gam(b"").n_reads == 0
    assert hc.AlnParts.read_gam(p).n_reads == 0
    st = hc.GamStream(p)
    chunks = list(st.chunks(1000))
    assert sum(c.n_reads for c in chunks) == 0
    b = hc.HostBatch(g, hc.AlnSet.read_gam(p))
    assert b.n_reads == 0
    # the end-of-file block after real data is still just the end
    a.write_gam(p)
    assert hc.AlnSet.read_gam(p).n_reads == a.n_reads


def test_oversized_quality_string_is_refused_not_indexed():
    """The GAM parser takes quality bytes independently of |sequence|; the general segment kernel keeps one quality
    prefix per 64 bytes for 65536 of them, so a read with a longer quality string must not reach it (it would write
    past its wave's LDS slice).  Flatten drops and counts it; the boundary value is still accepted."""
    from test_euka_cpu import _mk
    g = hc.Graph.from_arrays(1, 1, np.array([0, 0, 20], np.int64), b"ACGTACGTACGTACGTACGT", 1, np.zeros((2, 1), np.uint64),
                             np.array([-1, 5], np.int32), np.ones(30), "p0\n")
    ed = [(1, 0, False, [(20, 20, b"")])]
    alns = [_mk(b"ACGTACGTACGTACGTACGT", [30] * 20, ed), _mk(b"ACGTACGTACGTACGTACGT", [30] * 70000, ed),
            _mk(b"ACGTACGTACGTACGTACGT", [31] * 65535, ed)]
    a = hc.AlnSet.parse_gam(gamio.write_gam(alns))
    assert a.n_reads == 3
    b = hc.HostBatch(g, a)
    assert b.n_reads == 2 and b.stats.n_bad == 1 and sorted(b.read_src.tolist()) == [0, 2]
    ql = np.diff(b.arrays()["read_qual_off"])
    assert ql.max() == 65535 and b.n_tileable == 1


def test_the_worker_pool_survives_a_fork():
    """csrc/host/util.cpp parallel_run: a child of fork() (Python's multiprocessing default) inherits the pool's counters and
    none of its threads; it must start a pool of its own instead of waiting for workers that do not exist."""
    import os
    from vgan_amd import haplocart as hc
    g = hc.synth_graph(seed=3, genome_len=1200, n_nodes=800, n_paths=50)
    a = hc.synth_reads(g, 20000, seed=4, read_len=100)
    hb = hc.HostBatch(g, a, n_threads=4)  # the pool has threads now
    pid = os.fork()
    if pid == 0:
        try:
            signal_ok = hc.HostBatch(g, a, n_threads=4).n_reads == hb.n_reads
        except BaseException:  # noqa: BLE001 -- the child must not fall back into pytest
            signal_ok = False
        os._exit(0 if signal_ok else 3)
    import time
    t0 = time.time()
    while time.time() - t0 < 60:
        done, st = os.waitpid(pid, os.WNOHANG)
        if done:
            assert os.WIFEXITED(st) and os.WEXITSTATUS(st) == 0
            return
        time.sleep(0.05)
    os.kill(pid, 9)
    os.waitpid(pid, 0)
    raise AssertionError("the forked child hung in parallel_run")
