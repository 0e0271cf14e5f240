"""Closed-form known-answer tests for the HaploCart oracle, checked with mpmath.
The reference pins no numeric likelihood, so these anchor the restated arithmetic (Q1-Q5, Q10)."""
import mpmath as mp
import numpy as np
import pytest

import orc

mp.mp.dps = 40
BG = {"A": mp.mpf("0.27532"), "C": mp.mpf("0.30044"), "G": mp.mpf("0.16644"), "T": mp.mpf("0.25780")}


def one_node_graph(seq, coord, n_map=17000, mappability=1.0):
    pathsgo = np.zeros((2, 2), np.uint8)
    pathsgo[1, 0] = 1  # path 0 supports node 1, path 1 does not
    pb = np.array([-1, coord], np.int32)
    return orc.Graph({1: seq}, 2, pathsgo, pb, np.full(n_map, mappability))


def aln(seq, qual, mapq=60, node=1, edits=None, rev=False):
    edits = edits or [{"from_length": len(seq), "to_length": len(seq), "sequence": b""}]
    return {"sequence": seq, "quality": bytes(qual), "mapping_quality": mapq, "identity": 1.0, "name": b"r",
            "path": {"name": b"", "mapping": [{"position": {"node_id": node, "offset": 0, "is_reverse": rev},
                                                 "edit": edits, "rank": 1}]}}


def expected_supported(read, graph, quals, mu, mapq=60, mappability=1):
    pcm = (1 - mp.mpf(10) ** (-mp.mpf(mapq) / 10)) * mappability
    match = (1 - 30 * mp.mpf(mu)) ** 8
    tot = mp.mpf(0)
    for r, g, q in zip(read, graph, quals):
        e = mp.mpf("0.25") if q <= 2 else mp.mpf(10) ** (-mp.mpf(q) / 10)
        eps = e if r == g else 1 - e
        tot += mp.log((1 - pcm) * BG[r] + pcm * match * (1 - eps))
    return tot


def test_survey_kats():
    # protein-coding coordinate (mu = 0 by integer division, Q2)
    g = one_node_graph(b"ACGT", 4000)
    a = orc.AlnSet([aln(b"ACGT", [40] * 4)])
    rc, ll, flags = orc.hc_read(g, a, 0)
    assert rc == 0 and flags == 0
    assert float(ll[0]) == pytest.approx(-4.03019902453559e-4, rel=1e-12)
    assert float(ll[1]) == pytest.approx(-36.8413614879047, rel=1e-13)
    assert float(ll[0]) == pytest.approx(float(expected_supported("ACGT", "ACGT", [40] * 4, 0)), rel=1e-13)
    # HVS-I coordinate 57..372
    g = one_node_graph(b"ACGT", 100)
    rc, ll, _ = orc.hc_read(g, a, 0)
    assert float(ll[0]) == pytest.approx(-5.60722331616443e-4, rel=1e-12)
    assert float(ll[0]) == pytest.approx(float(expected_supported("ACGT", "ACGT", [40] * 4, "1.64273e-7")), rel=1e-13)
    # a mismatching A column at mu = 0 (graph C, read A via a substitution edit)
    g = one_node_graph(b"C", 4000)
    a = orc.AlnSet([aln(b"A", [40], edits=[{"from_length": 1, "to_length": 1, "sequence": b"A"}])])
    rc, ll, _ = orc.hc_read(g, a, 0)
    assert float(ll[0]) == pytest.approx(-9.20759195234389, rel=1e-13)


@pytest.mark.parametrize("coord,mu", [(10, "2.29640e-8"), (400, "2.29640e-8"), (16400, "1.54555e-8"), (600, "6.91285e-9"),
                                      (1000, "6.91285e-9"), (16100, "2.48537e-8"), (5000, 0), (3306, "2.48537e-8")])
def test_regions(coord, mu):
    g = one_node_graph(b"ACGTAC", coord, mappability=0.7)
    quals = [2, 3, 17, 40, 41, 0]
    a = orc.AlnSet([aln(b"ACGTAC", quals, mapq=37)])
    rc, ll, _ = orc.hc_read(g, a, 0)
    assert rc == 0
    exp = expected_supported("ACGTAC", "ACGTAC", quals, mu, mapq=37, mappability=mp.mpf("0.7"))
    assert float(ll[0]) == pytest.approx(float(exp), rel=1e-13)
    # unsupported: sum over the whole quality window (Q3), Q<=2 -> 0.25 (Q10)
    unsup = sum(mp.log(mp.mpf("0.25") if q <= 2 else mp.mpf(10) ** (-mp.mpf(q) / 10)) for q in quals)
    assert float(ll[1]) == pytest.approx(float(unsup), rel=1e-13)


def test_q4_q5_window_and_whole_read_compare():
    """Two mappings: the second segment compares its graph bases with the START of the read (Q4) and its
    unsupported penalty runs |algnseq| entries from its offset, zero-padded past the end (Q5)."""
    pathsgo = np.zeros((3, 2), np.uint8)
    pathsgo[1, 0] = pathsgo[2, 0] = 1
    g = orc.Graph({1: b"ACG", 2: b"TTA"}, 2, pathsgo, np.array([-1, 4000, 4003], np.int32), np.ones(17000))
    quals = [30, 31, 32, 33, 34, 35]
    al = {"sequence": b"ACGTTA", "quality": bytes(quals), "mapping_quality": 60, "identity": 1.0, "name": b"r",
          "path": {"name": b"", "mapping": [
              {"position": {"node_id": 1, "offset": 0, "is_reverse": False},
               "edit": [{"from_length": 3, "to_length": 3, "sequence": b""}], "rank": 1},
              {"position": {"node_id": 2, "offset": 0, "is_reverse": False},
               "edit": [{"from_length": 3, "to_length": 3, "sequence": b""}], "rank": 2}]}}
    a = orc.AlnSet([al])
    rc, S, U, node = orc.hc_read_segments(g, a, 0)
    assert rc == 0 and node.tolist() == [1, 2]
    s0 = expected_supported("ACG", "ACG", quals[0:3], 0)
    s1 = expected_supported("ACG", "TTA", quals[3:6], 0)  # read bases from the read start, qualities offset
    assert S[0] == pytest.approx(float(s0), rel=1e-13)
    assert S[1] == pytest.approx(float(s1), rel=1e-13)
    lq = lambda q: mp.log(mp.mpf("0.25") if q <= 2 else mp.mpf(10) ** (-mp.mpf(q) / 10))
    assert U[0] == pytest.approx(float(sum(lq(q) for q in quals)), rel=1e-13)
    assert U[1] == pytest.approx(float(sum(lq(q) for q in quals[3:]) + 3 * mp.log(mp.mpf("0.25"))), rel=1e-13)
    rc, ll, _ = orc.hc_read(g, a, 0)
    assert float(ll[0]) == pytest.approx(S[0] + S[1], rel=1e-13)
    assert float(ll[1]) == pytest.approx(U[0] + U[1], rel=1e-13)


def test_sticky_background_error_prob():
    g = one_node_graph(b"ACGT", 4000)
    a = orc.AlnSet([aln(b"ACGT", [40, 93, 40, 40])])  # Q >= 90 flips to the background error probability
    rc, ll, _ = orc.hc_read(g, a, 0)
    pcm = 1 - mp.mpf(10) ** -6
    bep = mp.mpf("0.0001")
    exp = sum(mp.log((1 - pcm) * BG[c] + pcm * (1 - bep)) for c in "ACGT")
    assert float(ll[0]) == pytest.approx(float(exp), rel=1e-12)


def test_consensus_fasta_mode():
    g = one_node_graph(b"ACGT", 100)
    a = orc.AlnSet([aln(b"ACTT", [40] * 4, edits=[{"from_length": 2, "to_length": 2, "sequence": b""},
                                                    {"from_length": 1, "to_length": 1, "sequence": b"T"},
                                                    {"from_length": 1, "to_length": 1, "sequence": b""}])])
    p = orc.hc_params(background_error_prob=0.001, use_background_error_prob=True, is_consensus_fasta=True)
    rc, S, U, node = orc.hc_read_segments(g, a, 0, p)
    # Q6: one mapping with three edits -> only mppg_sizes[0] = 2 columns are scored
    match = (1 - 30 * mp.mpf("1.64273e-7")) ** 8
    bep = mp.mpf("0.001")
    exp = 2 * mp.log((1 - bep) * match * (1 - bep))
    assert len(S) == 1 and S[0] == pytest.approx(float(exp), rel=1e-13)


def test_helpers():
    L = orc.lib()
    assert L.orc_p_seq_error(2) == 0.25 and L.orc_p_seq_error(3) == 10 ** (-3 * 0.1)
    assert L.orc_qscore(0) == 0.25 and L.orc_qscore(1) == 0.25 and L.orc_qscore(2) == 0.25
    assert L.orc_qscore(40) == pytest.approx(1e-4, rel=1e-15)
    assert L.orc_background_freq(b"N") == 0.25 and L.orc_background_freq(b"G") == 0.16644
    # oplusInitnatl: 0 means "no value" (Q11)
    assert float(L.orc_oplusInitnatl(0.0, -3.0)) == -3.0
    assert float(L.orc_oplusInitnatl(-3.0, 0.0)) == pytest.approx(float(mp.log(mp.e ** -3 + 1)), rel=1e-15)
    assert float(L.orc_oplusnatl(-1000.0, -1001.0)) == pytest.approx(float(-1000 + mp.log1p(mp.e ** -1)), rel=1e-15)


def test_posterior_counts_a_path_once_per_level_it_is_reached_at():
    """get_posterior.cpp:51-76 on a children.txt that is not a tree, closed form on flat likelihoods: the confidence
    of a clade is (entries of all_top) / P, and all_top gets one entry per recursion level a path's name is in."""
    n = ["p%d" % i for i in range(10)]
    children = "\n".join(["p0 p1 p2", "p1 p3", "p2 p3 p4", "p3 p5", "p4 p5", "p6 p1 p3", "p7 nopath [tok]", "p8"]) + "\n"
    parents = "p5 p3 p3 p6 p0 p7 p8 p9\n"
    fv = np.full(10, -4.0, np.longdouble)
    got = orc.hc_posterior(fv, n, parents, children, "p5")
    assert [x[0] for x in got] == ["p5", "p3", "p6", "p0", "p7", "p8", "p9"]  # Q9: the repeated p3 is emitted once
    conf = dict((x[0], x[1]) for x in got)
    assert conf["p5"] == pytest.approx(0.1, rel=1e-15)
    assert conf["p3"] == pytest.approx(0.1, rel=1e-15)   # p5
    assert conf["p6"] == pytest.approx(0.5, rel=1e-15)   # p1 p3 | p3 p5 | p5: p3 and p5 twice
    assert conf["p0"] == pytest.approx(0.5, rel=1e-15)   # p1 p2 | p3 p4 | p5: once per level although two parents list them
    for k in ("p7", "p8", "p9"):                         # nothing to sum: defined as exp(0 - total)
        assert conf[k] == pytest.approx(float(mp.e ** 4 / 10), rel=1e-15)
    # unequal values: a twice-listed path weighs double
    fv = np.array([-9, -1, -9, -2, -9, -3, -9, -9, -9, -9], np.longdouble)
    got = dict((x[0], x[1]) for x in orc.hc_posterior(fv, n, parents, children, "p5"))
    tot = sum(mp.e ** float(v) for v in fv)
    assert got["p6"] == pytest.approx(float((mp.e ** -1 + 2 * mp.e ** -2 + 2 * mp.e ** -3) / tot), rel=1e-14)


def _term(read_base, graph_base, q, pcm, match):
    e = mp.mpf("0.25") if q <= 2 else mp.mpf(10) ** (-mp.mpf(q) / 10)
    eps = e if read_base == graph_base else 1 - e
    return mp.log((1 - pcm) * BG[read_base] + pcm * match * (1 - eps))


def _lq(q):
    return mp.log(mp.mpf("0.25") if q <= 2 else mp.mpf(10) ** (-mp.mpf(q) / 10))


def test_mixed_supported_and_unsupported_mappings_per_path():
    """Two mappings on two nodes, three paths: path 0 goes through both nodes, path 1 only through the first, path 2 only
    through the second.  ll[p] = sum over mappings of (supported ? S_m : U_m) (process_mapping.cpp:54-88) with the Q4 / Q5
    quirks: the second mapping compares its graph bases with the START of the read and sums its unsupported penalty over
    a window of |algnseq| qualities starting at ITS offset, zero (Q = 0 -> 0.25) beyond the quality string."""
    pathsgo = np.zeros((3, 3), np.uint8)
    pathsgo[1, 0] = pathsgo[2, 0] = 1
    pathsgo[1, 1] = 1
    pathsgo[2, 2] = 1
    g = orc.Graph({1: b"ACG", 2: b"ACT"}, 3, pathsgo, np.array([-1, 4000, 100], np.int32), np.full(17000, 1.0))
    quals = [30, 31, 32, 33, 34, 35]
    mappings = [{"position": {"node_id": 1, "offset": 0, "is_reverse": False}, "edit": [{"from_length": 3, "to_length": 3, "sequence": b""}], "rank": 1},
                {"position": {"node_id": 2, "offset": 0, "is_reverse": False}, "edit": [{"from_length": 3, "to_length": 3, "sequence": b""}], "rank": 2}]
    a = orc.AlnSet([{"sequence": b"ACGACT", "quality": bytes(quals), "mapping_quality": 50, "identity": 1.0, "name": b"r",
                     "path": {"name": b"", "mapping": mappings}}])
    rc, ll, _ = orc.hc_read(g, a, 0)
    assert rc == 0
    pcm = 1 - mp.mpf(10) ** -5
    m1, m2 = mp.mpf(1), (1 - 30 * mp.mpf("1.64273e-7")) ** 8  # coordinate 4000: mu = 0 (Q2); 100: HVS-I
    S1 = sum(_term(r, gb, q, pcm, m1) for r, gb, q in zip("ACG", "ACG", quals[0:3]))
    S2 = sum(_term(r, gb, q, pcm, m2) for r, gb, q in zip("ACG", "ACT", quals[3:6]))  # Q4: read bases from the read start
    U1 = sum(_lq(q) for q in quals)                       # window [0, 6)
    U2 = sum(_lq(q) for q in quals[3:]) + 3 * _lq(0)      # window [3, 9): three qualities, three zero pads (Q5)
    assert float(ll[0]) == pytest.approx(float(S1 + S2), rel=1e-13)
    assert float(ll[1]) == pytest.approx(float(S1 + U2), rel=1e-13)
    assert float(ll[2]) == pytest.approx(float(U1 + S2), rel=1e-13)


def test_reverse_strand_mapping():
    """A mapping on the reverse strand: the graph sequence is the node's reverse complement (vgan_utils.h:24 through
    get_sequence of the reversed handle), compared base by base with the read."""
    g = one_node_graph(b"AACGT", 5000)  # reverse complement ACGTT
    a = orc.AlnSet([aln(b"ACGTT", [38] * 5, rev=True), aln(b"ACGAT", [38] * 5, rev=True,
                                                           edits=[{"from_length": 3, "to_length": 3, "sequence": b""},
                                                                  {"from_length": 1, "to_length": 1, "sequence": b"A"},
                                                                  {"from_length": 1, "to_length": 1, "sequence": b""}])])
    rc, ll, _ = orc.hc_read(g, a, 0)
    pcm = 1 - mp.mpf(10) ** -6
    assert float(ll[0]) == pytest.approx(float(sum(_term(b, b, 38, pcm, 1) for b in "ACGTT")), rel=1e-13)
    rc, S, U, node = orc.hc_read_segments(g, a, 1)
    # Q6: one mapping, three edits -> the mapping scores mppg_sizes[0] = 3 columns (all matches); the substitution is never seen
    assert len(S) == 1 and S[0] == pytest.approx(float(sum(_term(b, b, 38, pcm, 1) for b in "ACG")), rel=1e-13)
    assert U[0] == pytest.approx(float(5 * _lq(38)), rel=1e-13)


def test_deletion_and_insertion_columns():
    """Q7 / Q8: an insertion at running offset 0 is a softclip ('S' columns in the graph sequence), elsewhere a gap ('-');
    a deletion puts '-' into the read string at the running from_length sum.  Columns with a non-ACGT base on either side
    are skipped (process_mapping.cpp:62-63); the unsupported penalty still runs over the whole quality window."""
    g = one_node_graph(b"ACGTAC", 4000)
    # one mapping per edit so that every edit is scored (Q6 indexes the sizes per mapping)
    def m(off, ed, rank):
        return {"position": {"node_id": 1, "offset": off, "is_reverse": False}, "edit": [ed], "rank": rank}
    quals = [30, 30, 30, 30, 30]
    mappings = [m(0, {"from_length": 2, "to_length": 2, "sequence": b""}, 1),   # AC
                m(2, {"from_length": 1, "to_length": 0, "sequence": b""}, 2),   # deletion of G
                m(3, {"from_length": 0, "to_length": 1, "sequence": b"G"}, 3),  # insertion away from offset 0 -> '-'
                m(3, {"from_length": 2, "to_length": 2, "sequence": b""}, 4)]   # TA
    a = orc.AlnSet([{"sequence": b"ACGTA", "quality": bytes(quals), "mapping_quality": 60, "identity": 1.0, "name": b"r",
                     "path": {"name": b"", "mapping": mappings}}])
    rc, gs, rs, sizes = orc.reconstruct(g, a, 0)
    assert rc == 0 and gs == b"ACG-TA" and list(sizes) == [2, 1, 1, 2]
    assert rs == b"AC-GTA"  # Q8: the gap sits at the running sum of from_length (2)
    rc, S, U, node = orc.hc_read_segments(g, a, 0)
    pcm = 1 - mp.mpf(10) ** -6
    t = lambda b: _term(b, b, 30, pcm, 1)
    # segment 0: columns 0-1 vs read[0:2] = AC; 1: graph G vs read[0] = A (Q4), a mismatch at quality[2];
    # 2: graph '-' is skipped; 3: graph TA vs read[0:2] = AC (Q4): two mismatches with qualities[4], beyond the string -> Q 0
    assert S[0] == pytest.approx(float(t("A") + t("C")), rel=1e-13)
    assert S[1] == pytest.approx(float(_term("A", "G", 30, pcm, 1)), rel=1e-13)
    assert S[2] == 0.0
    assert S[3] == pytest.approx(float(_term("A", "T", 30, pcm, 1) + _term("C", "A", 0, pcm, 1)), rel=1e-13)
    # windows of |algnseq| = 6 qualities from each segment's start over a 5-byte string (Q5)
    starts = [0, 2, 3, 4]
    for k, s0 in enumerate(starts):
        assert U[k] == pytest.approx(float((5 - s0) * _lq(30) + (6 - (5 - s0)) * _lq(0)), rel=1e-13), k
