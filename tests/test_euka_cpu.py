"""CPU tests of the euka oracle (closed forms with mpmath) and of the product's euka host side against it."""
import ctypes as C
import os

import mpmath as mp
import numpy as np
import pytest

import gamio
import orc
import util
from vgan_amd import euka as ek
from vgan_amd import haplocart as hc

mp.mp.dps = 40
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DHIGH5 = ("A>C\tA>G\tA>T\tC>A\tC>G\tC>T\tG>A\tG>C\tG>T\tT>A\tT>C\tT>G\n"
          "0\t0\t0\t0\t0\t0.329405\t0\t0\t0\t0\t0\t0\n0\t0\t0\t0\t0\t0.221745\t0\t0\t0\t0\t0\t0\n"
          "0\t0\t0\t0\t0\t0.187678\t0\t0\t0\t0\t0\t0\n0\t0\t0\t0\t0\t0.161196\t0\t0\t0\t0\t0\t0\n"
          "0\t0\t0\t0\t0\t0.144011\t0\t0\t0\t0\t0\t0\n")
DHIGH3 = DHIGH5.replace("0\t0\t0\t0\t0\t0.", "0\t0\t0\t0\t0\t0\t0.").replace("\t0\t0\t0\t0\t0\t0\n", "\t0\t0\t0\t0\t0\n")


def _mk(seq, qual, mappings, mapq=60, identity=1.0):
    return {"sequence": seq, "quality": bytes(qual), "mapping_quality": mapq, "identity": identity, "name": b"r",
            "path": {"name": b"", "mapping": [{"position": {"node_id": n, "offset": o, "is_reverse": rv},
                                                 "edit": [{"from_length": f, "to_length": t, "sequence": s} for f, t, s in ed],
                                                 "rank": i + 1} for i, (n, o, rv, ed) in enumerate(mappings)]}}


def _graph(seqs):
    node_seqs = {i + 1: s for i, s in enumerate(seqs)}
    mx = len(seqs)
    return orc.Graph(node_seqs, 1, np.zeros((mx + 1, 1), np.uint8), np.full(mx + 1, -1, np.int32), np.ones(1))


def test_shipped_profiles_product_vs_oracle():
    """share/vgan/damageProfiles/*.prof (copied under tests/golden/ as data): combined matrices agree."""
    d = os.path.join(GOLD, "damageProfiles")
    t5, t3 = open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read()
    od = orc.OrcDamage(t5, t3)
    pd = ek.Damage.load(d + "/dhigh5p.prof", d + "/dhigh3p.prof")
    s5, s3 = pd.sub5p, pd.sub3p
    assert s5[0][1, 3] == 0.329405 and s3[0][2, 0] == 0.32891
    for L_, l in [(15, 0), (15, 14), (75, 2), (75, 72), (150, 60)]:
        a5, a3 = s5[min(l, len(s5) - 1)], s3[min(L_ - l - 1, len(s3) - 1)]
        got = np.stack([a5[b] if a5[b, b] <= a3[b, b] else a3[b] for b in range(4)])
        assert np.array_equal(got, od.matrix(L_, l)), (L_, l)
    none = ek.Damage.load(d + "/none.prof", d + "/none.prof")
    assert np.array_equal(none.sub5p[0], np.eye(4))


def test_damage_matrices_closed_form_and_product_tables():
    od = orc.OrcDamage(DHIGH5, DHIGH3)
    # position 0 of a 40-mer: C row from the 5' profile (C>T .329405), G row from the 3' profile only at the 3' end
    m = od.matrix(40, 0)
    assert m[1, 3] == pytest.approx(0.329405) and m[1, 1] == pytest.approx(1 - 0.329405)
    assert m[2, 0] == pytest.approx(0.144011)  # 3' matrix at distance 39 -> padded last row (damage.cpp:134-136)
    m = od.matrix(40, 39)
    assert m[2, 0] == pytest.approx(0.329405) and m[1, 3] == pytest.approx(0.144011)
    m = od.matrix(40, 20)
    assert m[1, 3] == pytest.approx(0.144011) and m[2, 0] == pytest.approx(0.144011)
    # the product keeps only the K rows of each end; combine on the fly must equal subDeamDiNuc[L][l]
    pd = ek.Damage.from_text(DHIGH5, DHIGH3)
    s5, s3 = pd.sub5p, pd.sub3p
    assert s5.shape == (5, 4, 4) and s3.shape == (5, 4, 4)
    for L_, l in [(15, 0), (15, 14), (40, 3), (40, 36), (1000, 999), (77, 4), (77, 5), (30, 12)]:
        a5, a3 = s5[min(l, 4)], s3[min(L_ - l - 1, 4)]
        got = np.stack([a5[b] if a5[b, b] <= a3[b, b] else a3[b] for b in range(4)])
        assert np.array_equal(got, od.matrix(L_, l)), (L_, l)
    none = ek.Damage.from_text("", "")
    assert np.array_equal(none.sub5p[0], np.eye(4)) and np.array_equal(orc.OrcDamage("", "").matrix(50, 7), np.eye(4))
    with pytest.raises(Exception):
        ek.Damage.from_text("A>C\tA>G\n0\t0\n", "")


def test_db_loaders_match_oracle_on_the_shipped_tables():
    REF_SHARE = GOLD  # share/vgan/euka_dir/euka_db.{clade,bins}, copied as data fixtures
    db = ek.EukaDb.load(REF_SHARE + "/euka_dir/euka_db.clade", REF_SHARE + "/euka_dir/euka_db.bins")
    assert db.n_clades == 335
    L = orc.lib()
    bo, lo, hi, en = np.zeros(400, np.int32), np.zeros(8000, np.int32), np.zeros(8000, np.int32), np.zeros(8000)
    n = L.orc_load_clade_chunks(open(REF_SHARE + "/euka_dir/euka_db.bins", "rb").read(), bo.ctypes.data_as(C.c_void_p),
                                lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), en.ctypes.data_as(C.c_void_p),
                                C.c_int64(399), C.c_int64(8000))
    assert n == 335 and np.array_equal(bo[:336], db.bin_off.astype(np.int32))
    nb = db.n_bins
    assert np.array_equal(lo[:nb], db.bin_lo) and np.array_equal(hi[:nb], db.bin_hi) and np.array_equal(en[:nb], db.bin_entropy)
    ids, dist, npth, sn, enn = (np.zeros(400, np.int32), np.zeros(400), np.zeros(400, np.int32), np.zeros(400, np.int32),
                                np.zeros(400, np.int32))
    names = C.create_string_buffer(1 << 16)
    n = L.orc_load_clade_info(open(REF_SHARE + "/euka_dir/euka_db.clade", "rb").read(), ids.ctypes.data_as(C.c_void_p),
                              dist.ctypes.data_as(C.c_void_p), npth.ctypes.data_as(C.c_void_p), sn.ctypes.data_as(C.c_void_p),
                              enn.ctypes.data_as(C.c_void_p), names, C.c_int64(1 << 16), C.c_int64(400))
    assert n == 335 and np.array_equal(dist[:335], db.clade_dist) and names.value.decode().split() == db.clade_names


def test_oracle_closed_forms():
    """One clade, one node 'ACGTACGTACGTACGTACGT' (20 bp), read = node with one C->T at the 5' end."""
    g = _graph([b"ACGTACGTACGTACGTACGT"])
    db = orc.EukaDb([0.1], [0, 1], [1], [1])
    dmg = orc.OrcDamage(DHIGH5, DHIGH3)
    read = b"ATGTACGTACGTACGTACGT"
    q = [40] * 20
    a = orc.AlnSet([_mk(read, q, [(1, 0, False, [(1, 1, b""), (1, 1, b"T"), (18, 18, b"")])], mapq=60)])
    o = orc.euka_run(g, a, db, dmg)
    assert o["clade"][0] == 0 and o["n_bad"] == 0
    d, e = mp.mpf("0.1"), mp.mpf(10) ** -4
    tT = lambda x, y: mp.mpf(1) if x == y else (mp.mpf("0.95238") if {x, y} in ({"A", "G"}, {"C", "T"}) else mp.mpf("0.02381"))
    s5 = [mp.mpf(x) for x in ("0.329405", "0.221745", "0.187678", "0.161196", "0.144011")]
    tot1 = mp.mpf(0)
    tot2 = mp.mpf(0)
    graph = "ACGTACGTACGTACGTACGT"
    for n, (gb, rb) in enumerate(zip(graph, read.decode())):
        c2t5, g2a3 = s5[min(n, 4)], s5[min(20 - n - 1, 4)]
        # row C: 5' matrix if its diagonal (1-c2t5) <= the 3' matrix's diagonal (1 for C) -> always the 5' row
        # row G: 3' matrix (diag 1-g2a3) vs 5' matrix (diag 1) -> the 3' row
        M = {"A": {"A": 1}, "T": {"T": 1}, "C": {"C": 1 - c2t5, "T": c2t5}, "G": {"G": 1 - g2a3, "A": g2a3}}
        pre = {b: (1 - d if b == gb else d * tT(gb, b)) for b in "ACGT"}
        post = {b2: sum(pre[b1] * M[b1].get(b2, 0) for b1 in "ACGT") for b2 in "ACGT"}
        tot1 += mp.log(sum(post[b] * ((1 - e) if b == rb else e / 3) for b in "ACGT"))
        tot2 += mp.log(1 - mp.mpf("0.25536")) if gb == rb else mp.log(mp.mpf("0.25536"))
    assert o["in_lik"][0] == pytest.approx(float(tot1), rel=1e-13)
    assert o["out_lik"][0] == pytest.approx(float(tot2), rel=1e-13)
    like = (1 - mp.mpf(10) ** -6) * mp.e ** (tot1 - mp.log(mp.e ** tot1 + mp.e ** tot2))
    assert o["like"][0] == pytest.approx(float(like), rel=1e-12) and o["not_like"][0] == pytest.approx(1 - float(like), abs=1e-15)
    assert o["pass"][0] == 1 and o["clade_count"][0] == 1 and o["bin_cov"][0] == 1.0
    # base shifts: first 5 and last 5 columns, index 4*graph+read (baseshift.cpp:84): position 1 is C>T
    bs = o["baseshift"][0]
    assert bs[0, 0] == 1 and bs[1, 1 * 4 + 3] == 1 and bs[2, 2 * 4 + 2] == 1 and bs[9, 3 * 4 + 3] == 1 and bs.sum() == 10


def test_oracle_reverse_strand_walks_the_damage_position_backwards():
    """A fragment on the reverse strand: the graph sequence is the node's reverse complement and the damage position starts
    at |sequence| - 1 and counts down (readGAM_Euka.h:208-216, 457-461), so column m uses subDeamDiNuc[L][L - 1 - m]: the
    G>A end of the profile meets the FIRST columns, the C>T end the last ones."""
    node = b"ACGTACGTACGTACGTACGTAC"
    rc_node = node[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))  # GTACGTACGTACGTACGTACGT
    g = _graph([node])
    db = orc.EukaDb([0.07], [0, 1], [1], [1])
    dmg = orc.OrcDamage(DHIGH5, DHIGH3)
    read = bytearray(rc_node)
    read[0] = ord("A")   # G>A at column 0 (position L-1: the 3' profile's first row)
    read[21] = ord("T")  # the last column holds a T already; force a C>T two columns earlier instead
    read[19] = ord("T")  # C>T at column 19 (position 2 from the 5' end)
    read = bytes(read)
    q = [35] * 22
    edits = []
    for i, (gb, rb) in enumerate(zip(rc_node, read)):
        edits.append((1, 1, b"" if gb == rb else bytes([rb])))
    a = orc.AlnSet([_mk(read, q, [(1, 0, True, edits)], mapq=60)])
    o = orc.euka_run(g, a, db, dmg)
    assert o["clade"][0] == 0 and o["n_bad"] == 0
    d, e = mp.mpf("0.07"), mp.mpf(10) ** (-mp.mpf(35) / 10)
    tT = lambda x, y: mp.mpf(1) if x == y else (mp.mpf("0.95238") if {x, y} in ({"A", "G"}, {"C", "T"}) else mp.mpf("0.02381"))
    s5 = [mp.mpf(x) for x in ("0.329405", "0.221745", "0.187678", "0.161196", "0.144011")]
    L_ = 22
    tot1 = mp.mpf(0)
    tot2 = mp.mpf(0)
    for m, (gb, rb) in enumerate(zip(rc_node.decode(), read.decode())):
        n = L_ - 1 - m
        c2t5, g2a3 = s5[min(n, 4)], s5[min(L_ - n - 1, 4)]
        M = {"A": {"A": 1}, "T": {"T": 1}, "C": {"C": 1 - c2t5, "T": c2t5}, "G": {"G": 1 - g2a3, "A": g2a3}}
        pre = {b: (1 - d if b == gb else d * tT(gb, b)) for b in "ACGT"}
        post = {b2: sum(pre[b1] * M[b1].get(b2, 0) for b1 in "ACGT") for b2 in "ACGT"}
        tot1 += mp.log(sum(post[b] * ((1 - e) if b == rb else e / 3) for b in "ACGT"))
        tot2 += mp.log(1 - mp.mpf("0.25536")) if gb == rb else mp.log(mp.mpf("0.25536"))
    assert o["in_lik"][0] == pytest.approx(float(tot1), rel=1e-13)
    assert o["out_lik"][0] == pytest.approx(float(tot2), rel=1e-13)
    # the same read taken as a forward fragment over the reverse-complemented node differs: the positions run the other way
    g2 = _graph([rc_node])
    a2 = orc.AlnSet([_mk(read, q, [(1, 0, False, edits)], mapq=60)])
    o2 = orc.euka_run(g2, a2, db, dmg)
    assert o2["out_lik"][0] == pytest.approx(float(tot2), rel=1e-13) and abs(o2["in_lik"][0] - float(tot1)) > 0.1


def test_oracle_special_columns():
    """N, gap (insertion + deletion), rare base, softclips, reverse strand walk."""
    g = _graph([b"ACGTNACGTRACGTACGTACGTAAAA", b"CCCC"])
    db = orc.EukaDb([0.2], [0, 1], [1], [2])
    dmg = orc.OrcDamage("", "")
    q = list(range(20, 20 + 40))
    ed = [(0, 3, b"GGG"), (4, 4, b""), (1, 1, b""), (2, 2, b""), (0, 2, b"TT"), (2, 2, b""), (1, 1, b""), (3, 3, b""), (2, 0, b""), (6, 6, b"")]
    read = b"GGG" + b"ACGT" + b"N" + b"AC" + b"TT" + b"GT" + b"R" + b"ACG" + b"CGTACG"
    a = orc.AlnSet([_mk(read, q[:len(read)], [(1, 0, False, ed)], mapq=40)])
    o = orc.euka_run(g, a, db, dmg)
    d = mp.mpf("0.2")
    qs = lambda Q: mp.mpf(10) ** (-mp.mpf(Q) / 10) if Q >= 2 else mp.mpf("0.25")  # Euka.cpp:38-51
    rc, gs, rs, sizes = orc.reconstruct(g, a, 0)
    # Q8: the deletion gap lands at sum(from_length) = 13, ignoring the 5 inserted read bases before it
    assert gs == b"SSSACGTNAC--GTRACGTACGTACG" and rs == b"GGGACGTNACTTG--TRACGCGTACG"
    t1 = mp.mpf(0)
    t2 = mp.mpf(0)
    sc = 0
    for m, (G, R) in enumerate(zip(gs.decode(), rs.decode())):
        Q = q[m] if m < len(read) else 0
        if G == "N" or R == "N":
            t1 += mp.log(mp.mpf("0.25"))
            t2 += mp.log(mp.mpf("0.25"))
        elif G == "-" or R == "-":
            t1 += mp.log(mp.mpf("0.002"))
            t2 += mp.log(mp.mpf("0.2"))
        elif G == "R" or R == "R":
            t1 += mp.log((1 - d) * mp.mpf("0.001"))
            t2 += mp.log(mp.mpf("0.001"))
        elif G == "S":
            sc += 1
            t1 += mp.log(1 - qs(Q)) if sc % 3 == 0 else mp.log(qs(Q) / 3)
            t2 += mp.log(mp.mpf("0.25"))
        else:
            e = qs(Q)
            tT = lambda x, y: mp.mpf("0.95238") if {x, y} in ({"A", "G"}, {"C", "T"}) else mp.mpf("0.02381")
            p = sum(((1 - d) if b == G else d * tT(G, b)) * ((1 - e) if b == R else e / 3) for b in "ACGT")
            t1 += mp.log(p)
            t2 += mp.log(1 - mp.mpf("0.25536")) if G == R else mp.log(mp.mpf("0.25536"))
    assert o["in_lik"][0] == pytest.approx(float(t1), rel=1e-13) and o["out_lik"][0] == pytest.approx(float(t2), rel=1e-13)
    # reads the oracle defines as bad (the reference indexes out of range): too short for subDeamDiNuc
    b = orc.AlnSet([_mk(b"ACGTACGT", [30] * 8, [(1, 0, False, [(4, 4, b""), (1, 1, b"A"), (3, 3, b"")])]),
                    _mk(b"ACGTACGTACGTACGTACGT", [30] * 20, [(1, 0, False, [(4, 4, b"")])], identity=0.0)])
    o = orc.euka_run(g, b, db, dmg)
    assert o["n_bad"] == 1 and list(o["clade"]) == [-1, -1] and o["clade_count"][0] == 0


def test_flatten_matches_oracle_reconstruct_and_filters():
    dm = ek.Damage.from_text(DHIGH5, DHIGH3)
    g, db, a = ek.synth_euka(300, dm, seed=11, n_clades=6, nodes_per_clade=120)
    hb = ek.EukaHostBatch(g, a, n_threads=3)
    assert hb.stats.n_out + hb.stats.n_bad + hb.stats.n_unmapped == 300 and hb.stats.n_out >= 290
    og, oa = util.orc_graph_nodes_only(g), util.orc_alnset_from_product(a)
    arr, src = hb.arrays(), hb.arrays()["read_src"]
    al = a.arrays()
    for k in range(0, hb.n_reads, 7):
        r = int(src[k])
        rc, gs, rs, _ = orc.reconstruct(og, oa, r)
        assert rc == 0
        c0 = arr["read_col_off"][k]
        assert arr["graph_seq"][c0:c0 + len(gs)].tobytes() == gs and arr["read_seq"][c0:c0 + len(rs)].tobytes() == rs
        assert arr["read_gseq_len"][k] == len(gs) and arr["read_rseq_len"][k] == len(rs)
        assert arr["read_seq_len"][k] == al["seq_off"][r + 1] - al["seq_off"][r]
        m0, m1 = arr["read_map_off"][k], arr["read_map_off"][k + 1]
        assert arr["map_node"][m0:m1].tolist() == al["m_node"][al["map_off"][r]:al["map_off"][r + 1]].tolist()
        assert arr["read_rev"][k] == al["m_rev"][al["map_off"][r]]
