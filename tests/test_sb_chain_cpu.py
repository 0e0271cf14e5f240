"""soibean downstream of analyse_GAM: the taxon tree, the tree-placement chain around the likelihood refresh, its
summaries and R-hat diagnostics (vgan_amd/csrc/host/sb_chain.cpp) against the oracle's restatement of the reference's loops
(oracle/sb_chain_oracle.cpp).  Host control flow, so it runs on the CPU: the product chain is driven through the engine
interface with the oracle's likelihood plugged in, hence identical doubles and files that must match byte for byte; on the GPU
(tests/test_sb_gpu.py) the same chain runs over vgan_sb_loglike."""
import glob
import gzip
import os
import re

import numpy as np
import pytest

import orc
import util
from vgan_amd import _native as N
from vgan_amd import euka as ek
from vgan_amd import haplocart as hc
from vgan_amd import soibean as sb
from test_euka_cpu import GOLD
from test_sb_cpu import FREQS

TREES = os.path.join(GOLD, "trees")


def _py_newick(text):
    """Independent mini-parser: [(name, dist, parent index)] in pre-order."""
    tok = re.findall(r"[(),;]|[^(),;:\s]+|:\s*[-+0-9.eE]+", text)
    nodes, stack, i, last = [], [], 0, None
    for t in tok:
        if t == "(":
            nodes.append(["", 0.0, stack[-1] if stack else -1])
            stack.append(len(nodes) - 1)
            last = None
        elif t == ",":
            last = None
        elif t == ")":
            last = stack.pop()
        elif t == ";":
            break
        elif t.startswith(":"):
            nodes[last][1] = float(t[1:])
        else:
            if last is None:
                nodes.append([t, 0.0, stack[-1] if stack else -1])
                last = len(nodes) - 1
            else:
                nodes[last][0] = t
    return nodes


@pytest.mark.parametrize("name", ["Paguroidea", "Ursidae", "Bovidae", "Muridae"])
def test_shipped_trees_parse(name):
    text = open(os.path.join(TREES, name + ".new.dnd")).read()
    t = sb.Tree.load(os.path.join(TREES, name + ".new.dnd"))
    want = _py_newick(text)
    assert t.n_nodes == len(want) and t.view.root == 0
    assert t.names == [w[0] for w in want]
    assert np.array_equal(t.dist, np.array([w[1] for w in want]))
    assert list(t.parent) == [w[2] for w in want]
    kids = [[] for _ in want]
    for i, w in enumerate(want):
        if w[2] >= 0:
            kids[w[2]].append(i)
    assert all(t.children(v) == kids[v] for v in range(t.n_nodes))
    assert t.n_leaves == sum(1 for k in kids if not k) == text.count(",") + 1
    # every inner label of the shipped trees is N<i><taxon>, every leaf an accession
    assert all((n.startswith("N") and n.endswith(name)) == bool(kids[i]) for i, n in enumerate(t.names) if i)


def test_newick_errors():
    for bad in ("", "(a:1,b:2)", "(a:1,b:2;", "(a:x,b:1);", "(a:1 b:2);"):
        with pytest.raises(N.NativeError):
            sb.Tree.parse(bad)
    with pytest.raises(N.NativeError):  # a parser that recurses per level must refuse absurd nesting, not overflow its stack
        sb.Tree.parse("(" * 200000 + "a" + ")" * 200000 + ";")
    deep = "(" * 5000 + "a:1" + ",b:1)" * 5000 + ";"
    assert sb.Tree.parse(deep).n_nodes == 10001
    t = sb.Tree.parse(" ( a:0.5 , (b:1e-3,c)d:2 )r ; ")
    assert t.names == ["r", "a", "d", "b", "c"] and list(t.dist) == [0.0, 0.5, 2.0, 1e-3, 0.0] and t.n_leaves == 3


def _newick_of(g):
    """The synthetic graph's path tree (parents.txt: `child parent`) as Newick, every node -- inner ones too -- a path."""
    kids, has_parent = {}, set()
    for line in g.parents_txt.splitlines():
        c = line.split()
        if len(c) >= 2:
            kids.setdefault(c[1], []).append(c[0])
            has_parent.add(c[0])
    roots = [n for n in g.path_names if n not in has_parent]
    assert len(roots) == 1
    names = {n: i for i, n in enumerate(g.path_names)}

    def rec(n):
        d = 0.004 + 0.003 * (names[n] % 7)
        inner = "(" + ",".join(rec(c) for c in kids[n]) + ")" if n in kids else ""
        return inner + n + ("" if n == roots[0] else ":%.6f" % d)
    return rec(roots[0]) + ";"


def _setup(n_reads=100, seed=6):
    g = hc.synth_graph(seed=17, genome_len=6000, n_nodes=4000, n_paths=28)
    a = hc.synth_reads(g, n_reads, seed=seed, read_len=60, indel_rate=0.1, softclip_rate=0.1)
    d = os.path.join(GOLD, "damageProfiles")
    texts = (open(d + "/dhigh5p.prof").read(), open(d + "/dhigh3p.prof").read())
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    o = orc.SbOracle(og, oa, orc.OrcDamage(*texts), penalty=7, path_findable=np.ones(g.n_paths, np.uint8))
    newick = _newick_of(g)
    return g, o, newick, a.n_reads


def _engine(o, batched=False):
    def refresh(st, con, f7):
        rc, v = o.loglike(st, con, f7, n_threads=1)
        return v, (1 if rc else 0)
    return sb.python_engine(refresh, lambda paths, lf: o.mixture_loglike(paths, lf), batched=batched)


def _files(prefix):
    out = {}
    for p in sorted(glob.glob(prefix + "*")):
        raw = open(p, "rb").read()
        out[os.path.basename(p)[len(os.path.basename(prefix)):]] = gzip.decompress(raw) if p.endswith(".mcmc") else raw
    return out


@pytest.mark.parametrize("seed", [1, 2])
def test_chain_files_match_the_oracle(tmp_path, seed):
    g, o, newick, n_reads = _setup()
    tree = sb.Tree.parse(newick)
    node_path = tree.node_paths(g.path_names)
    assert sorted(node_path) == list(range(g.n_paths))
    best, sig, n_ok = o.best_paths(n_reads)
    paths = sb.signature_paths(sig, n_ok, cutk=2)
    assert len(paths) == 2
    inv = {int(p): v for v, p in enumerate(node_path)}
    sig_nodes = [inv[int(p)] for p in paths]
    kw = dict(con=0.004, iters=90, burnin=30, chains=2, seed=seed)
    po, pp = str(tmp_path / "orc_"), str(tmp_path / "prod_")
    o.estimate(newick, g.path_names, sig_nodes, po, FREQS, **kw)
    # the product advances its chains together (seed 1: one call per chain and iteration, seed 2: one call for all chains);
    # the oracle runs them one after the other -- the chains own their generators, so the files are the same
    sb.estimate(_engine(o, batched=(seed == 2)), tree, node_path, sig_nodes, pp, g.n_paths, FREQS, **kw)
    fo, fp = _files(po), _files(pp)
    k = len(sig_nodes)
    assert sorted(fo) == sorted(fp) and len(fo) == k * (2 * 2 + 3)
    for name in fo:
        assert fo[name] == fp[name], (name, fo[name][:500], fp[name][:500])
    # shape of what was written: 60 recorded states per chain, every proposal traced, accepted and rejected moves both occur
    res = fp["Result%d0.mcmc" % k].decode().splitlines()
    assert len(res) == 1 + (90 - 30) and res[0].count("Source_") == k
    tr = fp["Trace%d1.detail.mcmc" % k].decode().splitlines()
    assert len(tr) == 1 + 91 and tr[1].endswith("accepted\t")
    moves = "".join(fp[n].decode() for n in fp if n.startswith("Trace"))
    assert "rejected" in moves and moves.count("accepted") > 2 * k
    est = fp["ProportionEstimates%d.txt" % k].decode().splitlines()
    assert len(est) == 2 * (1 + k) and est[0].startswith("Source\tChain\tMean Proportion Estimate")
    diag = fp["Diagnostics%d0.txt" % k].decode().splitlines()
    assert diag[0].startswith("Source\tHighest log-likelihood") and len(diag) >= 2
    # the sources of a state sum to one and stay on the tree
    last = res[-1].split("\t")
    assert abs(sum(float(last[4 * j + 2]) for j in range(k)) - 1) < 1e-9 and all(last[4 * j] in g.path_names for j in range(k))
    # a different seed walks differently
    sb.estimate(_engine(o), tree, node_path, sig_nodes, str(tmp_path / "other_"), g.n_paths, FREQS, **dict(kw, seed=seed + 50))
    assert _files(str(tmp_path / "other_"))["Trace%d0.detail.mcmc" % k] != fp["Trace%d0.detail.mcmc" % k]


def test_no_mcmc_and_argument_errors(tmp_path):
    g, o, newick, n_reads = _setup(n_reads=40)
    tree = sb.Tree.parse(newick)
    node_path = tree.node_paths(g.path_names)
    e = _engine(o)
    sb.estimate(e, tree, node_path, [1, 2], str(tmp_path / "n_"), g.n_paths, FREQS, run_mcmc=False)
    assert not glob.glob(str(tmp_path / "n_*"))
    with pytest.raises(N.NativeError):  # soibean.cpp:439-441
        sb.estimate(e, tree, node_path, [1], str(tmp_path / "x_"), g.n_paths, FREQS, iters=10, burnin=10)
    with pytest.raises(N.NativeError):
        sb.estimate(e, tree, node_path, [tree.n_nodes], str(tmp_path / "x_"), g.n_paths, FREQS, iters=20, burnin=5)
    with pytest.raises(N.NativeError):
        sb.estimate(e, tree, node_path, [], str(tmp_path / "x_"), g.n_paths, FREQS, iters=20, burnin=5)
    # a tree node without a graph path is an error when the walk reaches it, not a crash
    broken = node_path.copy()
    broken[:] = -1
    with pytest.raises(N.NativeError):
        sb.estimate(e, tree, broken, [1], str(tmp_path / "y_"), g.n_paths, FREQS, iters=20, burnin=5, chains=1)
    # a likelihood that reports one of the reference's guards stops the run with its message
    bad = sb.python_engine(lambda st, con, f7: (0.0, 1), lambda paths, lf: -1.0)
    with pytest.raises(N.NativeError) as ei:
        sb.estimate(bad, tree, node_path, [1], str(tmp_path / "z_"), g.n_paths, FREQS, iters=20, burnin=5, chains=1)
    assert "Problem in the likelihood compuation" in str(ei.value)
