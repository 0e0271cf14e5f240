"""CPU tests of the soibean oracle (closed forms with mpmath) and of the product's soibean front half against it."""
import mpmath as mp
import numpy as np
import pytest

import gamio
import orc
import util
from vgan_amd import haplocart as hc
from vgan_amd import soibean as sb
from test_euka_cpu import _mk

mp.mp.dps = 40
FREQS = [0.31, 0.27, 0.13, 0.29, 0.44, 0.56, 0.0012]  # A C G T R Y M


def small_graph():
    # nodes: 1 ACGTA, 2 CC (path 0 only), 3 GG (path 1 only), 4 TTTTACGTACGTAC
    pathsgo = np.zeros((5, 2), np.uint8)
    pathsgo[1] = pathsgo[4] = 1
    pathsgo[2, 0] = 1
    pathsgo[3, 1] = 1
    return orc.Graph({1: b"ACGTA", 2: b"CC", 3: b"GG", 4: b"TTTTACGTACGTAC"}, 2, pathsgo, np.full(5, -1, np.int32), np.ones(1))


def hky_mp(ref, read, t, con):
    fA, fC, fG, fT, fR, fY, mu = [mp.mpf(x) for x in FREQS]
    F = {"A": fA, "C": fC, "G": fG, "T": fT}
    tot = mp.mpf(0)
    for b in "ACGT":
        grp = fR if b in "AG" else fY
        Aexp = 1 - grp  # kappa = 0
        if b == ref:
            v = F[b] + F[b] * (1 / grp - 1) * mp.e ** (-mu * t) + ((grp - F[b]) / grp) * mp.e ** (-mu * t * Aexp)
        elif {b, ref} in ({"A", "G"}, {"C", "T"}):
            v = abs(F[b] + F[b] * (1 / grp - 1) * mp.e ** (-mu * t) - (F[b] / grp) * mp.e ** (-mu * t * Aexp))
        else:
            v = F[b] * (1 - mp.e ** (-mu * t))
        v = max(v, mp.mpf("1e-8"))
        tot += v * ((1 - con) if b == read else con / 3)
    return mp.log(tot)


def test_oracle_closed_forms_forward_read():
    g = small_graph()
    dmg = orc.OrcDamage("", "")
    # read over nodes 1, 2, 4 with one mismatch in node 4 (sub edit -> 3 edits in that mapping: Q6-style extra segments)
    q = [30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39]
    a = orc.AlnSet([_mk(b"ACGTACCTTTTACGAACGTAC", q,
                        [(1, 0, False, [(5, 5, b"")]), (2, 0, False, [(2, 2, b"")]),
                         (4, 0, False, [(7, 7, b""), (1, 1, b"A"), (6, 6, b"")])])])
    o = orc.SbOracle(g, a, dmg, penalty=7)
    assert o.n_bad == 0 and o.ok(0)
    pm = o.pathmap(0)
    qs = lambda Q: mp.mpf(10) ** (-mp.mpf(Q) / 10) if Q >= 2 else mp.mpf("0.25")
    CL = mp.log(mp.mpf("0.9999999"))
    # segments are edit level: sizes [5, 2, 7, 1, 6]; nodes for the first 3 = mappings' nodes, the last two "No_support"
    sizes, nodes = [5, 2, 7, 1, 6], [1, 2, 4, None, None]
    sup = {0: {1, 2, 4}, 1: {1, 3, 4}}
    exp = [mp.mpf(0), mp.mpf(0)]
    base = 0
    for size, node in zip(sizes, nodes):
        for p in (0, 1):
            if node is not None and node in sup[p]:
                exp[p] += size * CL  # supported regular columns: log(sum post) clamped
            else:
                for s in range(size):
                    Q = q[s]  # quality by within-segment index (Q12)
                    exp[p] += mp.log(1 - qs(Q)) if (base + s) % 7 == 0 else mp.log(qs(Q) / 3)
        base += size
    assert pm[0] == pytest.approx(float(exp[0]), rel=1e-12) and pm[1] == pytest.approx(float(exp[1]), rel=1e-12)
    c0, n0 = o.counts(0, 0)
    assert n0 == 21 and c0.sum() == 14 and c0[0 * 5 + 0] == 3  # 14 supported bases on path 0; A->A: two in node 1, one in node 4
    c1, _ = o.counts(0, 1)
    assert c1.sum() == 12  # node 2 is not on path 1
    # one likelihood refresh, k = 1: child path 0, parent path 1
    rc, ll = o.loglike([(0, 1, 0.03, 0.4, 1.0)], 0.01, FREQS)
    assert rc == 0
    t1, t2 = mp.mpf("0.4") * mp.mpf("0.03"), mp.mpf("0.03") - mp.mpf("0.4") * mp.mpf("0.03")
    refs = "ACGTA" + "CC" + "TTTTACG"
    LL = mp.mpf(float(pm[0])) + sum(hky_mp(b, b, t2, mp.mpf("0.01")) for b in refs)
    refs_p = "ACGTA" + "TTTTACG"
    LLP = mp.mpf(float(pm[1])) + sum(hky_mp(b, b, t1, mp.mpf("0.01")) for b in refs_p)
    expect = mp.log(mp.mpf("0.4") * mp.e ** LL + mp.mpf("0.6") * mp.e ** LLP)
    assert ll == pytest.approx(float(expect), rel=1e-11)
    # k = 2 mixture
    rc, ll2 = o.loglike([(0, 1, 0.03, 0.4, 0.7), (1, 0, 0.0, 0.25, 0.3)], 0.01, FREQS)
    assert rc == 0
    t = mp.mpf("0.00001")
    t1b, t2b = mp.mpf("0.25") * t, t - mp.mpf("0.25") * t
    LLb = mp.mpf(float(pm[1])) + sum(hky_mp(b, b, t2b, mp.mpf("0.01")) for b in refs_p)
    LLPb = mp.mpf(float(pm[0])) + sum(hky_mp(b, b, t1b, mp.mpf("0.01")) for b in refs)
    e2 = mp.log(mp.mpf("0.7") * (mp.mpf("0.4") * mp.e ** LL + mp.mpf("0.6") * mp.e ** LLP) +
                mp.mpf("0.3") * (mp.mpf("0.25") * mp.e ** LLb + mp.mpf("0.75") * mp.e ** LLPb))
    assert ll2 == pytest.approx(float(e2), rel=1e-11)
    # k = 3 mixture (BASELINE configs[4]): a third source on the same branch as the first with another position and length;
    # the read's likelihood is sum_y theta_y * (pos_y e^LL_y + (1 - pos_y) e^LLP_y) (MCMC.cpp:868,967-974)
    rc, ll3 = o.loglike([(0, 1, 0.03, 0.4, 0.5), (1, 0, 0.0, 0.25, 0.3), (0, 1, 0.012, 0.9, 0.2)], 0.01, FREQS)
    assert rc == 0
    tc = mp.mpf("0.012")
    t1c, t2c = mp.mpf("0.9") * tc, tc - mp.mpf("0.9") * tc
    LLc = mp.mpf(float(pm[0])) + sum(hky_mp(b, b, t2c, mp.mpf("0.01")) for b in refs)
    LLPc = mp.mpf(float(pm[1])) + sum(hky_mp(b, b, t1c, mp.mpf("0.01")) for b in refs_p)
    e3 = mp.log(mp.mpf("0.5") * (mp.mpf("0.4") * mp.e ** LL + mp.mpf("0.6") * mp.e ** LLP) +
                mp.mpf("0.3") * (mp.mpf("0.25") * mp.e ** LLb + mp.mpf("0.75") * mp.e ** LLPb) +
                mp.mpf("0.2") * (mp.mpf("0.9") * mp.e ** LLc + mp.mpf("0.1") * mp.e ** LLPc))
    assert ll3 == pytest.approx(float(e3), rel=1e-11)


def test_flatten_matches_oracle_slicing():
    g = hc.synth_graph(seed=9, genome_len=1500, n_nodes=1000, n_paths=12)
    a = hc.synth_reads(g, 250, seed=4, read_len=70, indel_rate=0.2, softclip_rate=0.2)
    hb = sb.SbHostBatch(g, a, n_threads=2)
    assert hb.stats.n_out + hb.stats.n_bad + hb.stats.n_unmapped == 250 and hb.stats.n_out > 200
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    o = orc.SbOracle(og, oa, orc.OrcDamage("", ""))
    arr = hb.arrays()
    src = arr["read_src"]
    kept = np.zeros(a.n_reads, bool)
    kept[src] = True
    al = a.arrays()
    for r in range(a.n_reads):
        if al["identity"][r] == 0:
            continue
        assert o.ok(r) == bool(kept[r]), r  # the reads the front half drops are exactly the ones the oracle rejects
    # edit-level segment count and the "No_support" tail
    for k in range(0, hb.n_reads, 11):
        r = int(src[k])
        rc, gs, rs, sizes = orc.reconstruct(og, oa, r)
        s0, s1 = arr["read_seg_off"][k], arr["read_seg_off"][k + 1]
        nM = al["map_off"][r + 1] - al["map_off"][r]
        assert s1 - s0 == len(sizes)
        nodes = arr["seg_node"][s0:s1]
        assert nodes[:min(nM, len(sizes))].tolist() == al["m_node"][al["map_off"][r]:al["map_off"][r] + min(nM, len(sizes))].tolist()
        assert np.all(nodes[nM:] == 0)


def test_signature_paths_rule():
    """Initial sources from the signature counts (soibean.cpp:669-712): >= 1 % of the reads, by descending count."""
    from vgan_amd import soibean as sb
    sig = np.array([0, 50, 3, 50, 700, 9, 0, 10], np.int64)
    assert list(sb.signature_paths(sig, 1000)) == [4, 1, 3, 7]          # 10 reads is exactly 1 %
    assert list(sb.signature_paths(sig, 1000, cutk=2)) == [4, 1]
    assert list(sb.signature_paths(sig, 1000, cutk=9)) == [4, 1, 3, 7]
    assert list(sb.signature_paths(np.array([0, 2, 0, 5], np.int64), 100000)) == [3, 1]  # nothing reaches 1 %: every seen path
    assert list(sb.signature_paths(np.zeros(6, np.int64), 100)) == []
