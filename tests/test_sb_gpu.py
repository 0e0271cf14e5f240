"""GPU parity tests of the soibean path (run with -m gpu): factorised tables and likelihood refresh vs the oracle."""
import os

import numpy as np
import pytest

import gamio
import orc
import util
from vgan_amd import euka as ek
from vgan_amd import haplocart as hc
from vgan_amd import soibean as sb
from test_euka_cpu import _mk, GOLD
from test_sb_cpu import FREQS

pytestmark = pytest.mark.gpu


def tree_pairs(g):
    """(child, parent, branch length) for every non-root path of the synthetic tree."""
    names = g.path_names
    idx = {n: i for i, n in enumerate(names)}
    pairs = []
    for line in g.parents_txt.splitlines():
        t = line.split()
        if len(t) >= 2:
            pairs.append((idx[t[0]], idx[t[1]]))
    return pairs


def run_case(g, a, texts, penalty, states_k1, states_k3):
    dm = ek.Damage.from_text(*texts)
    hb = sb.SbHostBatch(g, a)
    ctx = sb.SbContext(g, dm, penalty=penalty)
    dev_bad = ctx.precompute(hb)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    findable = np.array([len(n) <= 101 for n in g.path_names], np.uint8)
    o = orc.SbOracle(og, oa, orc.OrcDamage(*texts), penalty=penalty, path_findable=findable)
    assert o.n_bad == hb.stats.n_bad + dev_bad
    pm, cnt, ok = ctx.read_tables()
    src = hb.arrays()["read_src"]
    assert ok.all()
    step = max(1, hb.n_reads // 150)
    for k in range(0, hb.n_reads, step):
        r = int(src[k])
        ref = o.pathmap(r)
        assert util.rel_err(pm[:, k], ref) < 1e-11, r
        for p in range(0, g.n_paths, max(1, g.n_paths // 5)):
            c, _ = o.counts(r, p)
            assert np.array_equal(cnt[p, :, k].astype(np.uint32), c), (r, p)
    # the oracle must hold exactly the reads the device holds for the refresh comparison
    assert sum(o.ok(r) for r in range(a.n_reads)) == hb.n_reads
    got, guard = ctx.loglike(states_k1 + [], 0.01, FREQS)
    for st, v, gd in zip(states_k1, got, guard):
        rc, ref = o.loglike(st, 0.01, FREQS)
        assert rc == 0 and gd == 0
        assert v == pytest.approx(ref, rel=1e-10)
    got, guard = ctx.loglike(states_k3, 0.02, FREQS)
    for st, v in zip(states_k3, got):
        rc, ref = o.loglike(st, 0.02, FREQS)
        assert rc == 0 and v == pytest.approx(ref, rel=1e-10)
    # determinism of the refresh: identical bits on a second evaluation
    again, _ = ctx.loglike(states_k3, 0.02, FREQS)
    assert np.array_equal(again, got)
    # analyse_GAM's mostProbPath and the initial estimate of soibean.cpp:655-756
    best, sig, n_ok = ctx.best_paths()
    obest, osig, on = o.best_paths(a.n_reads)
    assert n_ok == on == hb.n_reads and np.array_equal(best, obest[src]) and np.array_equal(sig, osig)
    assert (best >= 0).sum() == sig.sum()
    paths = sb.signature_paths(sig, n_ok)
    want = [p for p in sorted(range(g.n_paths), key=lambda p: (-sig[p], p)) if sig[p] > 0 and sig[p] >= 0.01 * n_ok]
    assert list(paths) == (want or [p for p in sorted(range(g.n_paths), key=lambda p: (-sig[p], p)) if sig[p] > 0])
    use = list(paths[:4]) if len(paths) else [0, 1]
    for n in range(1, len(use) + 1):
        lf = float(np.log(1.0 / len(use))) if len(use) > 1 else 0.0
        v = ctx.mixture_loglike(use[:n], lf)
        assert v == pytest.approx(o.mixture_loglike(use[:n], lf), rel=1e-12)
        assert v == ctx.mixture_loglike(use[:n], lf)  # fixed summation order
    return ctx, o


def test_tree_of_28_paths_with_damage():
    g = hc.synth_graph(seed=17, genome_len=6000, n_nodes=4000, n_paths=28)
    a = hc.synth_reads(g, 1500, seed=6, read_len=60, indel_rate=0.1, softclip_rate=0.1)
    pairs = tree_pairs(g)
    d = os.path.join(GOLD, "damageProfiles")
    texts = (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read())
    k1 = [[(c, p, 0.02 + 0.001 * i, 0.1 + 0.08 * i, 1.0)] for i, (c, p) in enumerate(pairs[:9])] + [[(pairs[3][0], pairs[3][1], 0.0, 0.5, 1.0)]]
    k3 = [[(pairs[1][0], pairs[1][1], 0.03, 0.35, 0.5), (pairs[7][0], pairs[7][1], 0.011, 0.8, 0.3), (pairs[12][0], pairs[12][1], 0.04, 0.02, 0.2)],
          [(pairs[20][0], pairs[20][1], 0.0, 0.5, 0.2), (pairs[5][0], pairs[5][1], 0.2, 0.99, 0.2), (pairs[9][0], pairs[9][1], 0.003, 0.5, 0.6)]]
    run_case(g, a, texts, 7, k1, k3)


def test_many_paths_long_names_and_other_penalty():
    """130 paths (3 path slots per lane), a path name longer than 101 characters (never supported), PENALTY 3."""
    g0 = hc.synth_graph(seed=23, genome_len=3000, n_nodes=2000, n_paths=130)
    names = g0.path_names
    names[5] = "x" * 120
    g = hc.Graph.from_arrays(g0.min_id, g0.max_id, g0.node_seq_off, g0.node_seq.tobytes(), 130, g0.mask, g0.pangenome_base,
                             g0.mappability, "\n".join(names) + "\n", g0.parents_txt, g0.children_txt)
    a = hc.synth_reads(g0, 600, seed=2, read_len=80)
    pairs = [(c, p) for c, p in tree_pairs(g0) if c != 5 and p != 5]
    k1 = [[(pairs[10][0], pairs[10][1], 0.05, 0.3, 1.0)], [(5, pairs[0][1], 0.05, 0.3, 1.0)]]
    k3 = [[(pairs[40][0], pairs[40][1], 0.03, 0.35, 0.5), (pairs[80][0], pairs[80][1], 0.011, 0.8, 0.3), (pairs[100][0], pairs[100][1], 0.04, 0.2, 0.2)]]
    ctx, o = run_case(g, a, ("", ""), 3, k1, k3)
    pm, cnt, ok = ctx.read_tables(0, 50)
    assert cnt[5].sum() == 0  # no base of the over-long path is ever "supported"


def test_special_reads():
    """N / softclip / gap columns, reverse strand slicing, reads the reference cannot process."""
    seqs = {1: b"ACGTNACGTAACGTACGTACGTAAAA", 2: b"CCCCGGGGTTTTAAAACCCCGGGGTTTT", 3: b"ACGTACGTACGTACGTACGT"}
    node_seq = b"".join(seqs[i] for i in (1, 2, 3))
    off = np.array([0, 0, 26, 54, 74], np.int64)
    mask = np.zeros((4, 1), np.uint64)
    mask[1, 0] = 0b011
    mask[2, 0] = 0b101
    mask[3, 0] = 0b111
    g = hc.Graph.from_arrays(1, 3, off, node_seq, 3, mask, np.full(4, -1, np.int32), np.ones(1), "p0\np1\np2\n",
                             "p1 p0\np2 p0\n", "p0 p1 p2\n")
    q = list(range(20, 60))
    ed = [(0, 3, b"GGG"), (4, 4, b""), (1, 1, b""), (2, 2, b""), (0, 2, b"TT"), (2, 2, b""), (1, 1, b""), (3, 3, b""), (2, 0, b""), (6, 6, b"")]
    read = b"GGG" + b"ACGT" + b"N" + b"AC" + b"TT" + b"GT" + b"A" + b"ACG" + b"CGTACG"
    alns = [
        _mk(read, q[:len(read)], [(1, 0, False, ed)], mapq=40),
        _mk(b"AAAACGGGGAAAACCCC", [2, 0, 1, 93] + [30] * 13, [(2, 4, True, [(8, 8, b""), (1, 1, b"C"), (8, 8, b"")])]),
        _mk(b"ACGTACGTACGTACGTACGT", [35] * 20, [(3, 0, False, [(20, 20, b"")]), (2, 0, True, [(5, 5, b"")])]),
        _mk(b"ACGTACGT", [30] * 8, [(3, 0, False, [(8, 8, b"")])]),                                  # |graph_seq| < 15
        _mk(b"ACGTACGTACGTACGTACGT", [30] * 20, [(3, 0, False, [(20, 20, b"")])], identity=0.0),   # unmapped
    ]
    a = hc.AlnSet.parse_gam(gamio.write_gam(alns), keep_unmapped=True)
    d = os.path.join(GOLD, "damageProfiles")
    texts = (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read())
    k1 = [[(1, 0, 0.02, 0.3, 1.0)], [(2, 0, 0.05, 0.9, 1.0)]]
    k3 = [[(1, 0, 0.02, 0.3, 0.2), (2, 0, 0.05, 0.9, 0.5), (0, 1, 0.01, 0.5, 0.3)]]
    run_case(g, a, texts, 7, k1, k3)


def test_long_match_edits_count_beyond_255():
    """One match edit of 300 identical (reference, read) pairs, and a 900-column one: the per-segment 5x5 counts are 16 bit
    (an 8-bit counter wrapped at 256 and lowered every HKY term of the read)."""
    rng = np.random.default_rng(11)
    long_seq = bytes(rng.choice(list(b"ACGT"), 900).tolist())
    seqs = {1: b"A" * 320, 2: long_seq, 3: b"ACGTACGTACGTACGTACGT"}
    node_seq = b"".join(seqs[i] for i in (1, 2, 3))
    off = np.array([0, 0, 320, 1220, 1240], np.int64)
    mask = np.zeros((4, 1), np.uint64)
    mask[1, 0] = 0b011
    mask[2, 0] = 0b110
    mask[3, 0] = 0b111
    g = hc.Graph.from_arrays(1, 3, off, node_seq, 3, mask, np.full(4, -1, np.int32), np.ones(1), "p0\np1\np2\n",
                             "p1 p0\np2 p0\n", "p0 p1 p2\n")
    alns = [
        _mk(b"A" * 300, [30 + (i % 11) for i in range(300)], [(1, 5, False, [(300, 300, b"")])]),
        _mk(long_seq, [25 + (i % 17) for i in range(900)], [(2, 0, False, [(900, 900, b"")])]),
        _mk(b"A" * 256 + b"ACGTACGTAC", [33] * 266, [(1, 64, False, [(256, 256, b"")]), (3, 0, False, [(10, 10, b"")])]),
    ]
    a = hc.AlnSet.parse_gam(gamio.write_gam(alns), keep_unmapped=True)
    ctx, o = run_case(g, a, ("", ""), 7, [[(1, 0, 0.02, 0.3, 1.0)]], [[(1, 0, 0.02, 0.3, 0.2), (2, 0, 0.05, 0.9, 0.5), (0, 1, 0.01, 0.5, 0.3)]])
    _, cnt, _ = ctx.read_tables()
    # A->A pairs on p0: none (read 1 is off p0), 256 + 3 of node 3 (read 2), 300 (read 0); all 900 pairs of read 1 on p1
    assert sorted(cnt[0, 0, :].tolist()) == [0, 259, 300] and int(cnt[1].sum(axis=0).max()) == 900


def _same_tables(a, b, rel):
    """Tab-separated text files field by field: words identical, numbers within rel."""
    la, lb = a.decode().splitlines(), b.decode().splitlines()
    assert len(la) == len(lb)
    for x, y in zip(la, lb):
        fx, fy = x.split("\t"), y.split("\t")
        assert len(fx) == len(fy), (x, y)
        for u, v in zip(fx, fy):
            try:
                fu, fv = float(u), float(v)
            except ValueError:
                assert u == v, (x, y)
                continue
            assert fu == pytest.approx(fv, rel=rel, abs=1e-12) or (np.isnan(fu) and np.isnan(fv)), (x, y)


def _chain_files(prefix):
    import glob
    import gzip
    out = {}
    for p in sorted(glob.glob(prefix + "*")):
        raw = open(p, "rb").read()
        out[os.path.basename(p)[len(os.path.basename(prefix)):]] = gzip.decompress(raw) if p.endswith(".mcmc") else raw
    return out


def _soibean_case(n_reads=600):
    from test_sb_chain_cpu import _newick_of
    g = hc.synth_graph(seed=17, genome_len=6000, n_nodes=4000, n_paths=28)
    a = hc.synth_reads(g, n_reads, seed=6, read_len=60, indel_rate=0.1, softclip_rate=0.1)
    d = os.path.join(GOLD, "damageProfiles")
    return g, a, (d + "/dhigh5p.prof", d + "/dhigh3p.prof"), _newick_of(g)


def test_chain_over_the_gpu_refresh_follows_the_oracle_chain(tmp_path):
    """vgan_sb_estimate with the device engine vs the oracle's restatement of run_tree_proportion / processMCMCiterations with
    its own likelihood: same seed, so the same proposals; the likelihoods agree to ~1e-11, so every accept / reject decision
    and every recorded state does too."""
    g, a, profs, newick = _soibean_case()
    texts = tuple(open(p).read() for p in profs)
    dm = ek.Damage.from_text(*texts)
    hb = sb.SbHostBatch(g, a)
    ctx = sb.SbContext(g, dm, penalty=7)
    ctx.precompute(hb)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    o = orc.SbOracle(og, oa, orc.OrcDamage(*texts), penalty=7, path_findable=np.ones(g.n_paths, np.uint8))
    tree = sb.Tree.parse(newick)
    node_path = tree.node_paths(g.path_names)
    _, sig, n_ok = ctx.best_paths()
    paths = sb.signature_paths(sig, n_ok, cutk=2)
    inv = {int(p): v for v, p in enumerate(node_path)}
    sig_nodes = [inv[int(p)] for p in paths]
    kw = dict(con=0.004, iters=120, burnin=30, chains=2, seed=9)
    sb.estimate(ctx, tree, node_path, sig_nodes, str(tmp_path / "gpu_"), g.n_paths, FREQS, **kw)
    o.estimate(newick, g.path_names, sig_nodes, str(tmp_path / "orc_"), FREQS, **kw)
    fg, fo = _chain_files(str(tmp_path / "gpu_")), _chain_files(str(tmp_path / "orc_"))
    assert sorted(fg) == sorted(fo) and len(fg) == len(sig_nodes) * 7
    for name in fg:
        _same_tables(fg[name], fo[name], 1e-6 if name.endswith(".txt") else 1e-9)
    moves = "".join(fg[n].decode() for n in fg if n.startswith("Trace"))
    assert "rejected" in moves and "accepted" in moves


def test_vgan_soibean_cli_end_to_end(tmp_path):
    import shutil
    import subprocess
    g, a, profs, newick = _soibean_case(n_reads=500)
    db = tmp_path / "db"
    (db / "tree_dir").mkdir(parents=True)
    g.write(str(db))
    shutil.move(str(db / "graph.gfa"), str(db / "Synth.gfa"))
    (db / "tree_dir" / "Synth.new.dnd").write_text(newick + "\n")
    (db / "soibean_db.baseFreq").write_text("Other .25 .25 .25 .25\nSynth .31 .25 .15 .29\n")
    gam = str(tmp_path / "reads.gam")
    a.write_gam(gam)
    exe = os.path.join(os.path.dirname(GOLD), "..", "vgan_amd", "bin", "vgan")
    out = str(tmp_path / "bean_")
    r = subprocess.run([exe, "soibean", "-g", gam, "--soibean_dir", str(db), "--dbprefix", "Synth", "--deam5p", profs[0], "--deam3p", profs[1],
                        "-k", "2", "--iter", "100", "--burnin", "20", "--chains", "2", "--seed", "5", "-o", out, "-t", "-1"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Number of paths: 28" in r.stderr and "Initial log-likelihood" in r.stderr
    start = [int(x) for x in r.stderr.split("Random starting nodes: ")[1].splitlines()[0].split()]
    assert len(start) == 2
    # the oracle on the same files, same starting nodes, same seed
    texts = tuple(open(p).read() for p in profs)
    g2 = hc.Graph.load(str(db / "Synth.gfa"), None)
    a2 = hc.AlnSet.read_gam(gam)
    og, oa = util.orc_graph_from_product(g2), util.orc_alnset_from_product(a2)
    o = orc.SbOracle(og, oa, orc.OrcDamage(*texts), penalty=7, path_findable=np.ones(g2.n_paths, np.uint8))
    fr = [.31, .25, .15, .29]
    f7 = fr + [fr[0] + fr[2], fr[1] + fr[3],
               1 / (2 * ((22 * (fr[0] * fr[2])) + (22 * (fr[1] * fr[3])) + (fr[0] * fr[1] + (fr[0] * fr[3]) + (fr[2] * fr[1] + (fr[2] * fr[3])))))]
    tree = sb.Tree.parse(newick)
    # soibean.cpp:576-602 starts the search for the shortest branch at nodes[0] -- the root, length 0 -- so nothing is ever
    # shorter and con falls back to 0.01 (the CLI does the same)
    shortest = tree.dist[0]
    for d in tree.dist:
        if d < shortest and d != 0.0:
            shortest = d
    con = shortest if (shortest != 0 and shortest < 1) else 0.01
    assert con == 0.01
    o.estimate(newick, g2.path_names, start, str(tmp_path / "orc_"), f7, con=con, iters=100, burnin=20, chains=2, seed=5)
    fg, fo = _chain_files(out), _chain_files(str(tmp_path / "orc_"))
    assert sorted(fg) == sorted(fo) and len(fg) == 14
    for name in fg:
        _same_tables(fg[name], fo[name], 1e-6 if name.endswith(".txt") else 1e-9)
    # without -k: the initial estimate from the signature counts; --no-mcmc stops after the initial log-likelihoods
    r = subprocess.run([exe, "soibean", "-g", gam, "--soibean_dir", str(db), "--dbprefix", "Synth", "--no-mcmc", "-o", str(tmp_path / "nm_")],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "Identified signature paths" in r.stderr and r.stderr.count("Initial log-likelihood") >= 1


def test_column_kernel_equals_segment_kernel_bit_for_bit():
    """sb_precompute_cols_kernel (a lane per column) against sb_precompute_kernel (a lane per segment): the same arithmetic
    in the same order per value, so the tables are identical; reads beyond the column kernel's capacities (here: 300-column
    reads among 60-column ones) are left to the segment kernel inside the same call."""
    old = os.environ.get("VGAN_SB_PRECOMPUTE")
    d = os.path.join(GOLD, "damageProfiles")
    texts = (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read())
    try:
        for n_paths, read_len, seed in ((28, 60, 3), (28, 300, 4), (130, 80, 5), (64, 120, 6)):
            g = hc.synth_graph(seed=seed, genome_len=5000, n_nodes=3300, n_paths=n_paths)
            a = hc.synth_reads(g, 3000, seed=seed + 1, read_len=read_len, indel_rate=0.05, softclip_rate=0.05)
            hb = sb.SbHostBatch(g, a)
            tabs = {}
            for which in ("segments", "columns"):
                os.environ["VGAN_SB_PRECOMPUTE"] = which
                ctx = sb.SbContext(g, ek.Damage.from_text(*texts), penalty=7)
                bad = ctx.precompute(hb)
                tabs[which] = ctx.read_tables() + (bad,)
            for x, y in zip(tabs["segments"][:3], tabs["columns"][:3]):
                assert x.dtype == y.dtype and np.array_equal(x.view(np.uint8), y.view(np.uint8)), (n_paths, read_len)
            assert tabs["segments"][3] == tabs["columns"][3]
            assert set(np.unique(tabs["columns"][2])) <= {0, 1}
    finally:
        if old is None:
            os.environ.pop("VGAN_SB_PRECOMPUTE", None)
        else:
            os.environ["VGAN_SB_PRECOMPUTE"] = old


def _group_case(tmp_path, n_reads=900, shares=(0.0, 0.37, 0.5, 1.0)):
    """One context holding every read, and a group of contexts holding contiguous shares of them."""
    g, a, profs, newick = _soibean_case(n_reads=n_reads)
    texts = tuple(open(p).read() for p in profs)
    dm = ek.Damage.from_text(*texts)
    one = sb.SbContext(g, dm, penalty=7)
    one.precompute(sb.SbHostBatch(g, a))
    parts = []
    for f0, f1 in zip(shares[:-1], shares[1:]):
        c = sb.SbContext(g, dm, penalty=7)
        c.precompute(sb.SbHostBatch(g, a, int(a.n_reads * f0), int(a.n_reads * f1)))
        parts.append(c)
    return g, a, profs, newick, one, parts


def test_sums_over_reads_do_not_depend_on_how_the_reads_are_dealt(tmp_path):
    """Every sum over reads is taken in fixed point (vgan_sb_sum): the contexts' sums added up are the one context's sum,
    integer for integer, and the log-likelihood is the same double -- refresh, batched refresh and the initial mixture."""
    g, a, profs, newick, one, parts = _group_case(tmp_path)
    grp = sb.SbGroup(parts)
    states = [[(1, 0, 0.02, 0.3, 1.0)], [(3, 2, 0.01, 0.7, 1.0)], [(5, 4, 0.0, 0.5, 1.0)]]
    k3 = [[(1, 0, 0.02, 0.3, 0.5), (3, 2, 0.01, 0.7, 0.3), (6, 5, 0.03, 0.2, 0.2)]]
    for sts in (states, k3):
        whole, gw = one.loglike_sums(sts, 0.01, FREQS)
        split = [c.loglike_sums(sts, 0.01, FREQS) for c in parts]
        for e in range(len(sts)):
            hi = sum(s[0][e][0] for s in split)
            lo = sum(s[0][e][1] for s in split)
            assert (hi, lo) == whole[e][:2] and whole[e][2] == 0.0
            assert sb.sum_value([s[0][e] for s in split]) == sb.sum_value([whole[e]])
        v, _ = one.loglike(sts, 0.01, FREQS)
        assert [sb.sum_value([w]) for w in whole] == list(v)
        for e, st in enumerate(sts):
            assert grp.refresh(st, 0.01, FREQS)[0] == v[e] == one.refresh(st, 0.01, FREQS)[0]
    paths = [0, 3, 5]
    lf = float(np.log(1 / 3))
    assert grp.mixture_loglike(paths, lf) == one.mixture_loglike(paths, lf)
    assert sb.sum_value([c.mixture_sums(paths, lf) for c in parts]) == one.mixture_loglike(paths, lf)
    _, sig1, n1 = one.best_paths()
    sigg, ng = grp.best_paths()
    assert np.array_equal(sig1, sigg) and n1 == ng
    # and against the oracle (long double, its own order): the fixed-point unit is 2^-44 per read
    texts = tuple(open(p).read() for p in profs)
    o = orc.SbOracle(util.orc_graph_from_product(g), util.orc_alnset_from_product(a), orc.OrcDamage(*texts), penalty=7,
                     path_findable=np.ones(g.n_paths, np.uint8))
    rc, ref = o.loglike(k3[0], 0.01, FREQS)
    assert rc == 0 and abs(grp.refresh(k3[0], 0.01, FREQS)[0] - ref) <= 1e-10 * abs(ref)


def test_resident_refresh_kernel_gives_the_launched_refresh_bit_for_bit(tmp_path):
    """vgan_sb_resident: the engine's refresh served by a kernel that stays on the device, against the launched refresh -- states of one
    and three sources, the chains' batched call, other calls on the context in between (the kernel leaves and comes back), a pause
    beyond the kernel's idle limit, a second context on the same device (its refreshes are launched), and whole chains."""
    import time
    g, a, profs, newick, one, parts = _group_case(tmp_path)
    rng = np.random.default_rng(5)
    pairs = tree_pairs(g)
    states = []
    for i in range(60):
        k = 1 if i % 3 == 0 else 3
        th = rng.dirichlet([1] * k)
        states.append([(*pairs[rng.integers(len(pairs))], 0.01 + 0.05 * rng.random(), rng.random() * 0.98 + 0.01, float(th[y])) for y in range(k)])
    assert one.resident() is False
    want = [one.refresh(st, 0.01, FREQS) for st in states]
    one.resident(True)
    assert one.resident() is True and one.resident_launches() == 0
    got = [one.refresh(st, 0.01, FREQS) for st in states]
    assert got == want and one.resident_launches() == 1
    # another call on the context: the kernel leaves, the next refresh starts it again
    v = one.mixture_loglike([0, 3, 5], float(np.log(1 / 3)))
    assert one.refresh(states[1], 0.01, FREQS) == want[1] and one.resident_launches() == 2
    assert one.mixture_loglike([0, 3, 5], float(np.log(1 / 3))) == v
    # 5 ms without a refresh: it leaves by itself
    assert one.refresh(states[2], 0.01, FREQS) == want[2]
    n = one.resident_launches()
    time.sleep(0.05)
    assert one.refresh(states[3], 0.01, FREQS) == want[3] and one.resident_launches() == n + 1
    # back to back with pauses around the limit (the kernel may be leaving while the refresh is posted)
    for i, ms in enumerate((4.0, 4.5, 5.0, 5.5, 6.0, 5.2, 4.8, 5.1)):
        time.sleep(ms * 1e-3)
        assert one.refresh(states[i], 0.01, FREQS) == want[i]
    # a second context on the device keeps launching; the group's refresh (its contexts' sums added) is unchanged
    parts[0].resident(True)
    before = parts[0].resident_launches()
    grp = sb.SbGroup(parts)
    for st in states[:6]:
        assert grp.refresh(st, 0.01, FREQS)[0] == one.refresh(st, 0.01, FREQS)[0]
    assert parts[0].resident_launches() == before  # (`one` holds the device)
    km = one.kernel_ms()["refresh"]
    assert km[1] > 0 and 0 < km[0] / km[1] < 5.0  # the kernel's own clock per refresh, ms
    # whole chains, file for file
    tree = sb.Tree.parse(newick)
    node_path = tree.node_paths(g.path_names)
    _, sig, n_ok = one.best_paths()
    paths = sb.signature_paths(sig, n_ok, cutk=2)
    inv = {int(p): v for v, p in enumerate(node_path)}
    sig_nodes = [inv[int(p)] for p in paths]
    kw = dict(con=0.004, iters=150, burnin=30, chains=3, seed=11)
    sb.estimate(one, tree, node_path, sig_nodes, str(tmp_path / "res_"), g.n_paths, FREQS, **kw)
    assert one.resident_launches() > n + 1
    one.resident(False)
    sb.estimate(one, tree, node_path, sig_nodes, str(tmp_path / "lau_"), g.n_paths, FREQS, **kw)
    f1, f2 = _chain_files(str(tmp_path / "res_")), _chain_files(str(tmp_path / "lau_"))
    assert sorted(f1) == sorted(f2) and len(f1) >= 7
    for name in f1:
        assert f1[name] == f2[name], name


def test_chains_over_a_group_of_contexts_write_the_files_of_one_context(tmp_path):
    """vgan_sb_estimate over vgan_sb_engine_group (three contexts on this GPU, uneven shares) against the same chains over one
    context: every output file byte for byte."""
    g, a, profs, newick, one, parts = _group_case(tmp_path)
    tree = sb.Tree.parse(newick)
    node_path = tree.node_paths(g.path_names)
    _, sig, n_ok = one.best_paths()
    paths = sb.signature_paths(sig, n_ok, cutk=2)
    inv = {int(p): v for v, p in enumerate(node_path)}
    sig_nodes = [inv[int(p)] for p in paths]
    kw = dict(con=0.004, iters=150, burnin=30, chains=3, seed=11)
    sb.estimate(one, tree, node_path, sig_nodes, str(tmp_path / "one_"), g.n_paths, FREQS, **kw)
    sb.estimate(sb.SbGroup(parts), tree, node_path, sig_nodes, str(tmp_path / "grp_"), g.n_paths, FREQS, **kw)
    f1, fg = _chain_files(str(tmp_path / "one_")), _chain_files(str(tmp_path / "grp_"))
    assert sorted(f1) == sorted(fg) and len(f1) >= 7
    for name in f1:
        assert f1[name] == fg[name], name


def test_vgan_soibean_deals_the_reads_to_several_device_contexts(tmp_path):
    """`vgan soibean --gpus 0,0,0` (three contexts on the one GPU of the test rig) writes the files of `--gpus 0`."""
    import shutil
    import subprocess
    g, a, profs, newick = _soibean_case(n_reads=700)
    db = tmp_path / "db"
    (db / "tree_dir").mkdir(parents=True)
    g.write(str(db))
    shutil.move(str(db / "graph.gfa"), str(db / "Synth.gfa"))
    (db / "tree_dir" / "Synth.new.dnd").write_text(newick + "\n")
    (db / "soibean_db.baseFreq").write_text("Other .25 .25 .25 .25\nSynth .31 .25 .15 .29\n")
    gam = str(tmp_path / "reads.gam")
    a.write_gam(gam)
    exe = os.path.join(os.path.dirname(GOLD), "..", "vgan_amd", "bin", "vgan")
    outs = {}
    for tag, gpus in (("one", "0"), ("three", "0,0,0")):
        out = str(tmp_path / (tag + "_"))
        r = subprocess.run([exe, "soibean", "-g", gam, "--soibean_dir", str(db), "--dbprefix", "Synth", "--deam5p", profs[0], "--deam3p", profs[1],
                            "--iter", "120", "--burnin", "20", "--chains", "2", "--seed", "7", "-o", out, "--gpus", gpus],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        assert ("3 device contexts" in r.stderr) == (tag == "three")
        outs[tag] = (_chain_files(out), [l for l in r.stderr.splitlines() if "log-likelihood" in l or "signature" in l])
    assert sorted(outs["one"][0]) == sorted(outs["three"][0]) and len(outs["one"][0]) >= 7
    for name in outs["one"][0]:
        assert outs["one"][0][name] == outs["three"][0][name], name
    assert outs["one"][1] == outs["three"][1]
    env = dict(os.environ, VGAN_GPUS="0,0")
    r = subprocess.run([exe, "soibean", "-g", gam, "--soibean_dir", str(db), "--dbprefix", "Synth", "--no-mcmc", "-o", str(tmp_path / "env_")],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "2 device contexts" in r.stderr
