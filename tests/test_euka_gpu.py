"""GPU parity tests of the euka path (run with -m gpu): product (C-ABI, HIP) vs oracle."""
import os

import numpy as np
import pytest

import gamio
import orc
import util
from vgan_amd import euka as ek
from vgan_amd import haplocart as hc
from test_euka_cpu import _mk, GOLD

pytestmark = pytest.mark.gpu


def compare(g, db, a, dm_texts, min_mapq=29, ltp=5):
    dm = ek.Damage.from_text(*dm_texts)
    hb = ek.EukaHostBatch(g, a)
    ctx = ek.EukaContext(db, dm, min_mapq=min_mapq, length_to_prof=ltp)
    got = ctx.accumulate(hb)
    fin = ctx.finalize()
    og, oa = util.orc_graph_nodes_only(g), util.orc_alnset_from_product(a)
    ref = orc.euka_run(og, oa, util.orc_euka_db_from_product(db), orc.OrcDamage(*dm_texts), min_mapq, ltp)
    src = hb.arrays()["read_src"]
    kept = np.zeros(a.n_reads, bool)
    kept[src] = True
    assert np.all(ref["clade"][~kept] == -1)  # reads the flatten step drops are the ones the oracle skips
    assert np.array_equal(got["clade"], ref["clade"][src])
    ok = got["clade"] >= 0
    for k in ("in_lik", "out_lik", "like"):
        assert util.rel_err(got[k][ok], ref[k][src][ok]) < 1e-10, k
    assert np.allclose(got["not_like"][ok], ref["not_like"][src][ok], rtol=0, atol=1e-12)
    assert np.array_equal(got["pass"], ref["pass"][src])
    assert np.array_equal(fin["clade_count"], ref["clade_count"])
    assert np.array_equal(fin["baseshift"], ref["baseshift"])
    assert np.allclose(fin["bin_cov"], ref["bin_cov"], rtol=1e-12, atol=1e-12)
    assert fin["n_bad"] + hb.stats.n_bad == ref["n_bad"]
    return got, fin, ref


def test_synthetic_clades_with_damage():
    d = os.path.join(GOLD, "damageProfiles")
    texts = (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read())
    dm = ek.Damage.from_text(*texts)
    g, db, a = ek.synth_euka(3000, dm, seed=5, n_clades=12, nodes_per_clade=200)
    got, fin, ref = compare(g, db, a, texts)
    assert got["pass"].sum() > 1000 and fin["baseshift"].sum() > 10000
    # no damage model, other thresholds
    compare(g, db, a, ("", ""), min_mapq=0, ltp=3)


@pytest.mark.parametrize("blocks", ["3", "11"])
def test_full_blocks_of_reads_per_wave(monkeypatch, blocks):
    """The kernel takes a wave's reads in blocks of 64 (a lane per read around the column loop): few workgroups give every wave
    several full blocks and a ragged last one, with clade boundaries inside blocks (VGAN_EUKA_BLOCKS caps the grid; a launch of
    the default size gives a wave of a small batch four reads)."""
    monkeypatch.setenv("VGAN_EUKA_BLOCKS", blocks)
    d = os.path.join(GOLD, "damageProfiles")
    texts = (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read())
    dm = ek.Damage.from_text(*texts)
    g, db, a = ek.synth_euka(5003, dm, seed=17, n_clades=12, nodes_per_clade=200)
    got, fin, ref = compare(g, db, a, texts)
    assert got["pass"].sum() > 1500
    compare(g, db, a, ("", ""), min_mapq=0, ltp=12)  # base shifts beyond the LDS accumulators' length


def test_special_columns_and_filters():
    seqs = [b"ACGTNACGTRACGTACGTACGTAAAA", b"CCCCGGGGTTTTAAAACCCCGGGGTTTT", b"ACGTACGTACGTACGTACGT"]
    node_seq = b"".join(seqs)
    off = np.array([0, 0, 26, 54, 74], np.int64)
    g = hc.Graph.from_arrays(1, 3, off, node_seq, 1, np.zeros((4, 1), np.uint64), np.full(4, -1, np.int32), np.ones(1))
    import ctypes as C
    from vgan_amd import _native as N
    cd = np.array([0.2, 0.07])
    bo = np.array([0, 2, 3], np.uint32)
    lo, hi = np.array([1, 2, 3], np.int32), np.array([1, 2, 3], np.int32)
    en = np.zeros(3)
    v = N.EukaDbView(2, None, cd.ctypes.data, None, None, None, b"c0\nc1\n", bo.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                     en.ctypes.data)
    h = N.vp()
    N.check(N.lib().vgan_euka_db_from_arrays(C.byref(v), C.byref(h)))
    db = ek.EukaDb(h)
    q = list(range(20, 60))
    ed = [(0, 3, b"GGG"), (4, 4, b""), (1, 1, b""), (2, 2, b""), (0, 2, b"TT"), (2, 2, b""), (1, 1, b""), (3, 3, b""), (2, 0, b""), (6, 6, b"")]
    read = b"GGG" + b"ACGT" + b"N" + b"AC" + b"TT" + b"GT" + b"R" + b"ACG" + b"CGTACG"
    alns = [
        _mk(read, q[:len(read)], [(1, 0, False, ed)], mapq=40),
        _mk(b"AAAACGGGGAAAACCCC", [2, 0, 1, 93] + [30] * 13, [(2, 4, True, [(8, 8, b""), (1, 1, b"C"), (8, 8, b"")])], mapq=60),
        _mk(b"ACGTACGTACGTACGTACGT", [35] * 20, [(3, 0, False, [(20, 20, b"")]), (2, 0, False, [(0, 0, b"")])], mapq=29),  # mapq == MINIMUMMQ fails
        _mk(b"ACGTACGTACGTACGTACGT", [35] * 20, [(3, 0, False, [(20, 20, b"")])], mapq=30),
        _mk(b"ACGTACGT", [30] * 8, [(3, 0, False, [(8, 8, b"")])]),                        # Lseq < 15: skipped
        _mk(b"ACGTACGTACGTACGTACGT", [30] * 20, [(3, 0, False, [(20, 20, b"")])], identity=0.0),  # unmapped
        _mk(b"ACGTACGTACGTACG", [30] * 15, [(3, 0, False, [(20, 20, b"")])]),              # path longer than the read: n runs out
    ]
    a = hc.AlnSet.parse_gam(gamio.write_gam(alns), keep_unmapped=True)
    d = os.path.join(GOLD, "damageProfiles")
    got, fin, ref = compare(g, db, a, (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read()))
    assert ref["n_bad"] >= 1 and list(got["pass"][:4]) == [ref["pass"][i] for i in range(4)]


def test_shipped_bins_table_clade_lookup():
    """Real euka_db.{clade,bins} (335 clades, 3929 overlapping bins): the device's breakpoint search returns the
    clade the reference's double loop ends on, for reads starting anywhere in the id space."""
    db = ek.EukaDb.load(GOLD + "/euka_dir/euka_db.clade", GOLD + "/euka_dir/euka_db.bins")
    rng = np.random.default_rng(3)
    # a tiny graph whose node ids are spread over the whole table: ids listed explicitly
    ids = np.unique(np.concatenate([rng.integers(1, 6_930_000, 400), db.bin_lo[::37], db.bin_hi[::41], [1, 2, 28140, 28141]]))
    ids = ids[ids > 0]
    max_id = int(ids.max())
    off = np.zeros(max_id + 2, np.int64)
    seq = bytearray()
    present = np.zeros(max_id + 2, bool)
    present[ids] = True
    pos = 0
    lens = np.where(present[:max_id + 1], 20, 0)
    off[1:] = np.cumsum(lens)
    node_seq = b"ACGTACGTACGTACGTACGT" * len(ids)
    g = hc.Graph.from_arrays(int(ids.min()), max_id, off, node_seq, 1, np.zeros((max_id + 1, 1), np.uint64),
                             np.full(max_id + 1, -1, np.int32), np.ones(1))
    alns = [_mk(b"ACGTACGTACGTACGTACGT", [37] * 20, [(int(i), 0, False, [(20, 20, b"")])]) for i in ids]
    a = hc.AlnSet.parse_gam(gamio.write_gam(alns))
    got, fin, ref = compare(g, db, a, ("", ""))
    assert len(set(got["clade"].tolist())) > 50


def test_long_reads_and_ragged_read_counts():
    """Reads of 15..960 columns (many 16-column steps per row, carries across steps), both strands, a read count that
    does not fill the last wave, multi-node paths with more than 16 mappings and clades with more than 16 bins."""
    import ctypes as C
    from vgan_amd import _native as N
    rng = np.random.default_rng(11)
    n_nodes = 60
    lens = rng.integers(3, 40, n_nodes)
    seqs = [bytes(rng.choice(list(b"ACGT"), int(L)).astype(np.uint8)) for L in lens]
    off = np.zeros(n_nodes + 2, np.int64)
    off[2:] = np.cumsum(lens)
    g = hc.Graph.from_arrays(1, n_nodes, off, b"".join(seqs), 1, np.zeros((n_nodes + 1, 1), np.uint64),
                             np.full(n_nodes + 1, -1, np.int32), np.ones(1))
    # clade 0: 20 single-node bins over nodes 1..20 (> 16 bins), clade 1: 3 wide bins over the rest
    bo = np.array([0, 20, 23], np.uint32)
    lo = np.array(list(range(1, 21)) + [21, 35, 50], np.int32)
    hi = np.array(list(range(1, 21)) + [34, 49, 60], np.int32)
    cd = np.array([0.11, 0.2])
    en = np.zeros(23)
    v = N.EukaDbView(2, None, cd.ctypes.data, None, None, None, b"c0\nc1\n", bo.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                     en.ctypes.data)
    h = N.vp()
    N.check(N.lib().vgan_euka_db_from_arrays(C.byref(v), C.byref(h)))
    db = ek.EukaDb(h)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    alns = []
    for i in range(37):  # 37 reads: the last wave holds one
        first = int(rng.integers(1, 12)) if i % 3 else int(rng.integers(21, 30))
        n_map = int(rng.integers(1, 45))
        last = min(n_nodes, first + n_map - 1)
        rev = bool(i % 4 == 1)
        nodes = list(range(first, last + 1))
        if rev:
            nodes = nodes[::-1]
        read = b""
        maps = []
        for nd in nodes:
            s = seqs[nd - 1]
            s = s.translate(comp)[::-1] if rev else s
            sb = bytearray(s)
            edits = []
            run = 0
            for k in range(len(sb)):
                if rng.random() < 0.01:  # a substitution splits the match run
                    if run:
                        edits.append((run, run, b""))
                        run = 0
                    alt = b"ACGT"[(b"ACGT".index(sb[k]) + 1 + int(rng.integers(3))) % 4]
                    sb[k] = alt
                    edits.append((1, 1, bytes([alt])))
                else:
                    run += 1
            if run:
                edits.append((run, run, b""))
            maps.append((nd, 0, rev, edits))
            read += bytes(sb)
        if len(read) < 15 or len(read) > 1000:
            continue
        q = rng.integers(0 if i % 5 == 0 else 28, 42, len(read)).tolist()
        alns.append(_mk(read, q, maps, mapq=int(rng.integers(20, 61))))
    if len(alns) % 4 == 0:
        alns.pop()
    assert max(len(x["sequence"]) for x in alns) > 300
    a = hc.AlnSet.parse_gam(gamio.write_gam(alns))
    d = os.path.join(GOLD, "damageProfiles")
    got, fin, ref = compare(g, db, a, (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read()))
    assert got["pass"].sum() >= 5 and (fin["bin_cov"] > 0).sum() >= 8


def test_bin_coverage_of_scattered_and_repeated_nodes():
    """Bin coverage (readGAM_Euka.h:520-546) when a read's nodes are NOT a short run of ids: nodes 64 ids and more apart, a node
    met twice, exactly 63 / 64 ids of span, in rows beside ordinary reads and on their own -- the device counts a bin's nodes
    from a 64-bit word of node bits where it can and by comparison where it cannot, and both must give the oracle's sums."""
    import ctypes as C
    from vgan_amd import _native as N
    rng = np.random.default_rng(23)
    n_nodes = 260
    lens = rng.integers(4, 9, n_nodes)
    seqs = [bytes(rng.choice(list(b"ACGT"), int(L)).astype(np.uint8)) for L in lens]
    off = np.zeros(n_nodes + 2, np.int64)
    off[2:] = np.cumsum(lens)
    g = hc.Graph.from_arrays(1, n_nodes, off, b"".join(seqs), 1, np.zeros((n_nodes + 1, 1), np.uint64),
                             np.full(n_nodes + 1, -1, np.int32), np.ones(1))
    # clade 0: nodes 1..130 in overlapping bins (one of a single node, one the wrong way round); clade 1: the rest, 18 bins
    lo0, hi0 = [1, 30, 64, 64, 100, 90], [40, 70, 64, 130, 99, 130]
    lo1 = [131 + 7 * k for k in range(18)]
    hi1 = [min(260, x + 9) for x in lo1]
    bo = np.array([0, len(lo0), len(lo0) + len(lo1)], np.uint32)
    lo, hi = np.array(lo0 + lo1, np.int32), np.array(hi0 + hi1, np.int32)
    cd = np.array([0.09, 0.17])
    en = np.zeros(len(lo))
    v = N.EukaDbView(2, None, cd.ctypes.data, None, None, None, b"c0\nc1\n", bo.ctypes.data, lo.ctypes.data, hi.ctypes.data, en.ctypes.data)
    h = N.vp()
    N.check(N.lib().vgan_euka_db_from_arrays(C.byref(v), C.byref(h)))
    db = ek.EukaDb(h)
    shapes = [lambda a: list(range(a, a + 6)),                       # ordinary
              lambda a: [a, a + 1, a + 100, a + 101, a + 2],         # far apart
              lambda a: [a, a + 1, a, a + 1, a + 2],                 # met twice
              lambda a: [a, a + 63, a + 1],                          # the last bit of the word
              lambda a: [a, a + 64, a + 1],                          # one beyond it
              lambda a: [a + 70, a + 3, a],                          # the smallest node last
              lambda a: list(range(a, a + 30)),                      # both registers of the row
              lambda a: list(range(a, a + 34))]                      # beyond them
    alns = []
    order = [0] * 8 + list(range(8)) * 4 + [0, 0, 0, 1, 0, 2, 0, 0, 6, 6, 6, 6, 7, 0, 0, 0]
    for i, k in enumerate(order):
        a0 = int(rng.integers(1, 25)) if i % 2 else int(rng.integers(131, 150))
        nodes = shapes[k](a0)
        read = b"".join(seqs[nd - 1] for nd in nodes)
        maps = [(nd, 0, False, [(len(seqs[nd - 1]), len(seqs[nd - 1]), b"")]) for nd in nodes]
        alns.append(_mk(read, rng.integers(30, 42, len(read)).tolist(), maps, mapq=60))
    a = hc.AlnSet.parse_gam(gamio.write_gam(alns))
    d = os.path.join(GOLD, "damageProfiles")
    got, fin, ref = compare(g, db, a, (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read()))
    assert got["pass"].sum() >= 40 and (fin["bin_cov"] > 0).sum() >= 12


def test_accumulate_in_batches_and_after_finalize():
    """Per-clade accumulators are kept in replicas that finalize folds together: batches before and after a finalize
    must add up to the single-batch result."""
    d = os.path.join(GOLD, "damageProfiles")
    dm = ek.Damage.load(d + "/dhigh5p.prof", d + "/dhigh3p.prof")
    g, db, a = ek.synth_euka(6000, dm, seed=21, n_clades=9, nodes_per_clade=150)
    whole = ek.EukaHostBatch(g, a)
    ctx = ek.EukaContext(db, dm)
    ctx.accumulate(whole)
    want = ctx.finalize()
    n = a.n_reads
    parts = [ek.EukaHostBatch(g, a, 0, n // 3), ek.EukaHostBatch(g, a, n // 3, n // 2), ek.EukaHostBatch(g, a, n // 2, n)]
    ctx.reset()
    ctx.accumulate(parts[0])
    mid = ctx.finalize()
    assert mid["clade_count"].sum() < want["clade_count"].sum()
    ctx.accumulate(parts[1])
    ctx.accumulate(parts[2])
    got = ctx.finalize()
    assert np.array_equal(got["clade_count"], want["clade_count"]) and np.array_equal(got["baseshift"], want["baseshift"])
    assert np.allclose(got["bin_cov"], want["bin_cov"], rtol=1e-12, atol=1e-12)
    again = ctx.finalize()  # idempotent
    assert np.array_equal(again["clade_count"], got["clade_count"]) and np.allclose(again["bin_cov"], got["bin_cov"], rtol=0, atol=0)


def test_long_damage_profiles_take_the_global_table_path():
    """Profiles of 12 x 11 rows make 132 (5' row, 3' row) pairs: more than the 64 the kernel keeps in LDS, so the
    instantiation that reads the pair table from HBM runs; every substitution type non-zero, both ends different."""
    hdr = "A>C\tA>G\tA>T\tC>A\tC>G\tC>T\tG>A\tG>C\tG>T\tT>A\tT>C\tT>G\n"
    rng = np.random.default_rng(5)

    def prof(rows, main_col):
        out = hdr
        for i in range(rows):
            v = rng.uniform(0.0, 0.004, 12)
            v[main_col] = 0.33 * 0.75 ** i
            out += "\t".join("%.6g" % x for x in v) + "\n"
        return out

    texts = (prof(12, 5), prof(11, 6))  # C>T at the 5' end, G>A at the 3' end
    dm = ek.Damage.from_text(*texts)
    g, db, a = ek.synth_euka(2500, dm, seed=31, n_clades=7, nodes_per_clade=180)
    got, fin, ref = compare(g, db, a, texts)
    assert got["pass"].sum() > 500


def test_like_sums_feed_the_abundance_chain():
    """vgan_euka_like_sums: per clade the number of clade_like entries and the sum of their logs (what
    MCMC::get_proposal_likelihood reads, MCMC.cpp:1175-1215), accumulated by the read kernel."""
    d = os.path.join(GOLD, "damageProfiles")
    texts = (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read())
    dm = ek.Damage.from_text(*texts)
    g, db, a = ek.synth_euka(6000, dm, seed=8, n_clades=9, nodes_per_clade=150)
    hb = ek.EukaHostBatch(g, a)
    ctx = ek.EukaContext(db, dm)
    got = ctx.accumulate(hb)
    ctx.finalize()
    n, s = ctx.like_sums()
    ok = got["clade"] >= 0
    assert np.array_equal(n, np.bincount(got["clade"][ok], minlength=db.n_clades))
    with np.errstate(divide="ignore"):
        want = np.bincount(got["clade"][ok], weights=np.log(got["like"][ok]), minlength=db.n_clades)
    fin_ = np.isfinite(want)  # the synthetic reads include mapq 0 ones: like == 0, log -> -inf, for the oracle's sums as well
    assert fin_.sum() >= 3 and np.array_equal(np.isneginf(s), ~fin_) and util.rel_err(s[fin_], want[fin_]) < 1e-12
    # a second batch adds on; a read with mapq 0 (like == 0) makes its clade's sum -inf, as the reference's log(0)
    arr = a.arrays()
    arr = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in arr.items()}
    arr["mapq"][:] = 0
    oa = orc.AlnSet.from_arrays(**arr)
    import ctypes as C
    from vgan_amd import _native as N
    v = N.AlnSetView(oa.n_reads, *[getattr(oa, k).ctypes.data for k in ("seq_off", "seq", "qual_off", "qual", "mapq", "identity")],
                     None, None, *[getattr(oa, k).ctypes.data for k in ("map_off", "m_node", "m_offset", "m_rev", "edit_off",
                                                                        "e_from", "e_to", "e_seq_off", "e_seq")])
    h = N.vp()
    N.check(N.lib().vgan_aln_from_arrays(v, h))
    a0 = hc.AlnSet(h)
    got0 = ctx.accumulate(ek.EukaHostBatch(g, a0, 0, 50))
    ctx.finalize()
    n2, s2 = ctx.like_sums()
    hit = np.unique(got0["clade"][got0["clade"] >= 0])
    assert np.all(got0["like"][got0["clade"] >= 0] == 0) and np.all(np.isneginf(s2[hit]))
    assert np.array_equal(n2 - n, np.bincount(got0["clade"][got0["clade"] >= 0], minlength=db.n_clades))
    rest = np.setdiff1d(np.arange(db.n_clades)[fin_], hit)
    assert util.rel_err(s2[rest], want[rest]) < 1e-12
    ctx.reset()
    ctx.finalize()
    n3, s3 = ctx.like_sums()
    assert not n3.any() and not s3.any()


def _tree(prefix):
    import glob
    return {os.path.basename(p)[len(os.path.basename(prefix)):]: open(p, "rb").read() for p in sorted(glob.glob(prefix + "_*"))}


@pytest.mark.parametrize("mcmc", [True, False])
def test_vgan_euka_cli_end_to_end_matches_the_oracle(tmp_path, mcmc):
    """`vgan euka -g` (GPU per-read pass + host abundance chain) writes the files the oracle's restatement of Euka::run
    writes for the same GAM, tables and seed -- byte for byte."""
    import subprocess
    d = os.path.join(GOLD, "damageProfiles")
    p5, p3 = d + "/dhigh5p.prof", d + "/dhigh3p.prof"
    texts = (open(p5).read(), open(p3).read())
    dm = ek.Damage.from_text(*texts)
    g, db, a = ek.synth_euka(20000, dm, seed=21, n_clades=14, nodes_per_clade=180)
    util.write_euka_db(db, g, tmp_path)
    gam = str(tmp_path / "reads.gam")
    if mcmc:  # lift the generator's mapq 0 reads (clade_like == 0 -> log-likelihood -inf -> the chain would never accept)
        import ctypes as C
        from vgan_amd import _native as N
        arr = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in a.arrays().items()}
        arr["mapq"] = np.maximum(arr["mapq"], 1)
        oa_ = orc.AlnSet.from_arrays(**arr)
        no, nm = np.ascontiguousarray(arr["name_off"]), np.ascontiguousarray(arr["name"])
        v = N.AlnSetView(oa_.n_reads, *[getattr(oa_, k).ctypes.data for k in ("seq_off", "seq", "qual_off", "qual", "mapq", "identity")],
                         no.ctypes.data, nm.ctypes.data,
                         *[getattr(oa_, k).ctypes.data for k in ("map_off", "m_node", "m_offset", "m_rev", "edit_off", "e_from", "e_to",
                                                                 "e_seq_off", "e_seq")])
        h = N.vp()
        N.check(N.lib().vgan_aln_from_arrays(v, h))
        a = hc.AlnSet(h)
    a.write_gam(gam)
    og_name = db.clade_names[3]
    args = ["--entropy", "0", "--minBins", "2", "--minFrag", "40", "--outFrag", "--outGroup", og_name, "-l", "4", "--seed", "77",
            "--iter", "800", "--burnin", "60", "--minMQ", "20"] + ([] if mcmc else ["--no-mcmc"])
    exe = os.path.join(os.path.dirname(GOLD), "..", "vgan_amd", "bin", "vgan")
    r = subprocess.run([exe, "euka", "-g", gam, "--euka_dir", str(tmp_path), "--deam5p", p5, "--deam3p", p3, "-o", str(tmp_path / "prod"),
                        "-t", "-1"] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Number of fragments in input file: 20000" in r.stderr
    # the same through the oracle, reading the same files back
    db2 = ek.EukaDb.load(str(tmp_path / "euka_db.clade"), str(tmp_path / "euka_db.bins"))
    a2 = hc.AlnSet.read_gam(gam, keep_unmapped=True)
    og, oa = util.orc_graph_nodes_only(g), util.orc_alnset_from_product(a2)
    odb = util.orc_euka_db_from_product(db2)
    ref = orc.euka_run(og, oa, odb, orc.OrcDamage(*texts), 20, 4)
    arr = a2.arrays()
    names = [bytes(arr["name"][arr["name_off"][i]:arr["name_off"][i + 1]]) for i in range(a2.n_reads)]
    orc.euka_report(odb, db2.clade_id, db2.clade_names, ref, np.diff(arr["seq_off"]), str(tmp_path / "orc"), names=names, min_bins=2,
                    min_reads=40, entropy=0.0, length_to_prof=4, run_mcmc=mcmc, iters=800, burnin=60, seed=77, out_frag=True,
                    out_group=og_name)
    fo, fp = _tree(str(tmp_path / "orc")), _tree(str(tmp_path / "prod"))
    assert sorted(fo) == sorted(fp) and len(fo) >= 8
    for k in fo:
        assert fo[k] == fp[k], (k, fo[k][:400], fp[k][:400])
    assert b"yes" in fp["_abundance.tsv"] and ("_%s.prof" % og_name) in fp


def test_vgan_euka_reads_its_gam_from_a_pipe(tmp_path):
    """The reference feeds readGAM3 through a FIFO (Euka.cpp:490-523): `-g /dev/stdin` gives the files of `-g file`."""
    import subprocess
    g, db, a = ek.synth_euka(3000, None, seed=4, n_clades=8, nodes_per_clade=150)
    util.write_euka_db(db, g, tmp_path)
    gam = str(tmp_path / "r.gam")
    a.write_gam(gam)
    exe = os.path.join(os.path.dirname(GOLD), "..", "vgan_amd", "bin", "vgan")
    common = ["--euka_dir", str(tmp_path), "--entropy", "0", "--minBins", "1", "--seed", "3", "--iter", "300", "--burnin", "30"]
    r1 = subprocess.run([exe, "euka", "-g", gam, "-o", str(tmp_path / "f")] + common, capture_output=True)
    r2 = subprocess.run([exe, "euka", "-g", "/dev/stdin", "-o", str(tmp_path / "p")] + common, input=open(gam, "rb").read(), capture_output=True)
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-500:], r2.stderr[-500:])
    a_, b_ = _tree(str(tmp_path / "f")), _tree(str(tmp_path / "p"))
    assert sorted(a_) == sorted(b_) and len(a_) >= 6 and all(a_[k] == b_[k] for k in a_)


def test_vgan_euka_deals_the_fragments_to_several_device_contexts(tmp_path):
    """`vgan euka --gpus 0,0,0` (three contexts sharing the one GPU of the test rig, a host thread each): the integer tables
    add exactly and the per-read results go back in input order, so every output file equals the single-context run's (the
    coverage sums are doubles added in another order: compared as numbers)."""
    import subprocess
    from test_sb_gpu import _same_tables
    d = os.path.join(GOLD, "damageProfiles")
    p5, p3 = d + "/dhigh5p.prof", d + "/dhigh3p.prof"
    dm = ek.Damage.load(p5, p3)
    g, db, a = ek.synth_euka(350_000, dm, seed=31, n_clades=12, nodes_per_clade=180)
    util.write_euka_db(db, g, tmp_path)
    gam = str(tmp_path / "reads.gam")
    a.write_gam(gam)
    exe = os.path.join(os.path.dirname(GOLD), "..", "vgan_amd", "bin", "vgan")
    args = ["--entropy", "0", "--minBins", "2", "--minFrag", "40", "--outFrag", "-l", "4", "--no-mcmc", "--minMQ", "20", "-t", "-1"]
    outs = {}
    for tag, extra in (("one", []), ("three", ["--gpus", "0,0,0"])):
        r = subprocess.run([exe, "euka", "-g", gam, "--euka_dir", str(tmp_path), "--deam5p", p5, "--deam3p", p3, "-o", str(tmp_path / tag)] + args + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("Summed the per-clade tables of 3 device contexts" in r.stderr) == (tag == "three")
        outs[tag] = _tree(str(tmp_path / tag))
        counts = [ln for ln in r.stderr.splitlines() if ln.startswith("Number of")]
        outs[tag + "_counts"] = counts
    assert outs["one_counts"] == outs["three_counts"] and sorted(outs["one"]) == sorted(outs["three"]) and len(outs["one"]) >= 6
    for k in outs["one"]:
        if outs["one"][k] != outs["three"][k]:  # only rounding of summed doubles may differ
            _same_tables(outs["one"][k], outs["three"][k], 1e-9)
