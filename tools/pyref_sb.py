#!/usr/bin/env python3
"""A second, independent restatement of soibean's per-read path -- Python + mpmath (40 digits), written from the reference's
sources, NOT from oracle/ (see tools/pyref_hc.py for why; the alignment reconstruction and the readers are shared with it,
the damage matrices with tools/pyref_euka.py).

What it follows (paths under /root/reference/src/):
    getLCAfromGAM.h:92-560     analyse_GAM: the edit-level segments of a read (mppg_sizes; those beyond the mappings are
                               "No_support"), the slice of the reconstructed sequences a segment takes on either strand, and
                               per path the supported walk (N / soft clip / gap constants, else the damage-marginalised
                               probability of the graph base at the SEGMENT's base index, clamped at log(0.9999999)) or the
                               unsupported walk (log(1 - e) on every PENALTY-th read position, log(e / 3) elsewhere) -- the
                               read's pathMap and its per-base records (reference base, read base, pathSupport)
    getLCAfromGAM.h:80-88      path names cut at 101 characters: a longer name is never found among a node's paths
    MCMC.h:66,108-296          computeBaseLogLike: HKY with kappa = 1 / 22 = 0 (integer division), floors at 1e-8, the
                               marginal over the read base with the error rate `con`, the cap log(0.999999999)
    MCMC.h:298-312             calculateLogWeightedAverage
    MCMC.cpp:738-993           the likelihood of a state: per read child and parent sums over the per-base records, mixed over the
                               branch position (k = 1) or over branch position and sources (k > 1)
    Euka.cpp:38-52             qscore_vec
    damage.cpp                 tools/pyref_euka.py

The outputs are the factorised form the product keeps: pm[path] = pathMap, cnt[path][5 x 5] = the
(reference, read) pairs of the records with pathSupport -- and the state log-likelihoods computed from the per-base records
themselves, as the reference does.

Undefined in the reference, refused here ("undefined_reads"): a quality index past the quality string or a quality >= 100,
a fragment whose |graph_seq| is below 15 or above 1000, a base index at or past it on a supported regular column, substr
past the end, a node without path list.

Usage (build container only):
    python tools/pyref_sb.py --make tests/golden/sb_pyref
    python tools/pyref_sb.py --run DIR [--out FILE]
"""
import argparse
import json
import os
import sys

import mpmath as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gamio  # noqa: E402
from pyref_hc import Undefined, load_gfa, reconstruct_graph_sequence, signed_char, substr  # noqa: E402
from pyref_euka import DamageModel, qscore, MINLENGTHFRAGMENT, MAXLENGTHFRAGMENT  # noqa: E402

mp.mp.dps = 40


def D(x):
    """A literal of the C++ source: the double it is there, not the decimal it is written as (log(0.9999999) differs from the
    log of the decimal by 1e-9 relative)."""
    return mp.mpf(float(x))
ACGT = "ACGT"
NEG_INF = mp.mpf("-inf")


def oplus(x, y):  # libgab oplusnatl
    hi, lo = (x, y) if x > y else (y, x)
    if lo == NEG_INF:
        return hi
    return hi + mp.log1p(mp.exp(lo - hi))


def oplus_init(x, y):  # libgab oplusInitnatl
    return y if x == 0 else oplus(x, y)


def pair_class(c):
    return ACGT.index(c) if c in ACGT else 4


# ---------------------------------------------------------------------------------------------------------- analyse_GAM
def analyse_read(seqs, node_paths, path_names, aln, dmg, penalty):
    """pathMap[p] and records[p] = list of (reference base, read base, pathSupport, logLikelihood) in the reference's order."""
    maps = aln["path"]["mapping"]
    if not maps:
        raise Undefined("no mapping")
    qual = aln["quality"]
    rev = maps[0]["position"]["is_reverse"]
    graph_seq, read_seq, sizes = reconstruct_graph_sequence(seqs, aln["path"])
    base_ix = len(aln["sequence"]) - 1 if rev else 0  # :107
    Lseq = len(graph_seq)
    names = [n[:101] for n in path_names]  # :80-88 (pathNames is resized in place)
    P = len(names)
    path_map = [mp.mpf(0)] * P
    records = [[] for _ in range(P)]

    def quality(s):
        if s > len(qual):
            raise Undefined("quality index %d past a string of %d" % (s, len(qual)))
        q = signed_char(qual[s]) if s < len(qual) else 0
        if q < 0 or q >= 100:
            raise Undefined("qscore_vec[%d]" % q)
        return qscore(q)

    for i, size in enumerate(sizes):
        if len(sizes) != len(maps) and i >= len(maps):  # :156-160
            prob_paths = ["No_support"]
        else:
            nid = maps[i]["position"]["node_id"]
            if nid not in node_paths:
                raise Undefined("nodepaths.at(%d)" % nid)
            prob_paths = node_paths[nid]
        if rev:  # :179-186
            start = base_ix - size - 1 if base_ix - size - 1 >= 0 else 0
        else:
            start = base_ix
        node_seq, part = substr(graph_seq, start, size), substr(read_seq, start, size)
        if len(part) + 1 < len(node_seq):
            raise Undefined("partReadSeq[%d] past a string of %d" % (len(part) + 1, len(part)))
        for m in range(P):
            on_read = base_ix
            if names[m] in prob_paths:
                for s in range(len(node_seq)):
                    g = node_seq[s]
                    r = part[s] if s < len(part) else "\0"
                    e = quality(s)
                    if g == "N" or r == "N":
                        ll, sup = mp.log(D(0.25)), False
                    elif g == "S" or r == "S":
                        ll, sup = mp.log(e / 3), False
                    elif g == "-" or r == "-":
                        ll, sup = mp.log(D(0.02)), False
                    else:
                        pre = [(1 - e) if ACGT[o] == g else e / 3 for o in range(4)]
                        if Lseq < MINLENGTHFRAGMENT or Lseq > MAXLENGTHFRAGMENT or base_ix < 0 or base_ix >= Lseq:
                            raise Undefined("subDeamDiNuc[%d][%d]" % (Lseq, base_ix))
                        post = [mp.mpf(0)] * 4
                        for d in range(4):
                            for o in range(4):
                                post[d] += pre[o] * dmg.row(Lseq, base_ix, o)[d]
                        ll = NEG_INF
                        for d in range(4):
                            ll = oplus_init(ll, mp.log(post[d]))
                        if ll > mp.log(D(0.9999999)):
                            ll = mp.log(D(0.9999999))
                        sup = True
                    path_map[m] += ll
                    records[m].append((g, r, sup, ll))
            else:
                for s in range(len(node_seq)):
                    g = node_seq[s]
                    r = part[s] if s < len(part) else "\0"
                    e = quality(s)
                    if g == "N" or r == "N":
                        ll = mp.log(D(0.25))
                    elif g == "S" or r == "S":
                        ll = mp.log(e / 3)
                    elif g == "-" or r == "-":
                        ll = mp.log(D(0.02))
                    elif abs(on_read) % penalty == 0:
                        ll = mp.log(1 - e)
                    else:
                        ll = mp.log(e / 3)
                    path_map[m] += ll
                    records[m].append((g, "-", False, ll))
                    if r != "-":
                        on_read += -1 if rev else 1
        if rev:  # :537-544
            base_ix = start
        else:
            base_ix += size
    return path_map, records


# -------------------------------------------------------------------------------------------------------- the likelihood
def hky_log(ref, read, t, con, freqs):
    """computeBaseLogLike without the record's own logLikelihood (MCMC.h:108-290)."""
    fA, fC, fG, fT, fR, fY, mu = freqs
    F = {"A": fA, "C": fC, "G": fG, "T": fT}
    kappa = mp.mpf(0)  # MCMC.h:66: 1 / 22 in integers
    prob = []
    for rb in ACGT:
        grp = fR if rb in "AG" else fY
        A = 1 + grp * (kappa - 1)
        if rb == ref:
            p = F[rb] + F[rb] * ((1 / grp) - 1) * mp.exp(-(mu * t)) + ((grp - F[rb]) / grp) * mp.exp(-(mu * t * A))
        elif {rb, ref} in ({"A", "G"}, {"C", "T"}):
            j1 = F[rb] + F[rb] * ((1 / grp) - 1) * mp.exp(-(mu * t))
            j11 = (F[rb] / grp) * mp.exp(-(mu * t * A))
            p = j1 - j11 if j1 > j11 else j11 - j1
        else:
            p = F[rb] * (1 - mp.exp(-(mu * t)))
        if p < D(1e-8):
            p = D(1e-8)
        prob.append(p)
    ll = NEG_INF
    for d in range(4):
        ll = oplus_init(ll, mp.log(prob[d]) + (mp.log(1 - con) if ACGT[d] == read else mp.log(con / 3)))
    if ll > D(1e-8):
        ll = mp.log(D(0.999999999))
    return ll


def read_sum(recs, t, con, freqs):
    tot = mp.mpf(0)
    for g, r, sup, ll in recs:
        tot += (hky_log(g, r, t, con, freqs) + ll) if sup else ll
    return tot


def state_loglike(all_records, state, con, freqs):
    """MCMC.cpp:738-993; state: list of (child, parent, dist, pos_branch, theta)."""
    total = mp.mpf(0)
    for records in all_records:
        if len(state) == 1:
            c, p, dist, pos, _ = state[0]
            t = dist if dist != 0 else D(0.00001)
            t1 = pos * t
            t2 = t - t1
            ll, llp = read_sum(records[c], t2, con, freqs), read_sum(records[p], t1, con, freqs)
            a, b = ll + mp.log(pos), llp + mp.log(1 - pos)  # calculateLogWeightedAverage
            hi = max(a, b)
            total += hi + mp.log(mp.exp(a - hi) + mp.exp(b - hi)) - mp.log(pos + (1 - pos))
        else:
            inter = NEG_INF
            for c, p, dist, pos, theta in state:
                t = dist if dist != 0 else D(0.00001)
                t1 = pos * t
                t2 = t - t1
                ll, llp = read_sum(records[c], t2, con, freqs), read_sum(records[p], t1, con, freqs)
                inter2 = oplus(mp.log(pos) + ll, mp.log(1 - pos) + llp)
                inter = oplus_init(inter, inter2 + mp.log(theta))
            total += inter
    return total


# ------------------------------------------------------------------------------------------------------------------ runs
FREQS = ["0.31", "0.27", "0.13", "0.29", "0.44", "0.56", "0.0012"]  # A C G T R Y M


def load_inputs(d):
    seqs = load_gfa(os.path.join(d, "graph.gfa"))
    path_names = [ln.split()[0] for ln in open(os.path.join(d, "graph_paths")) if ln.split()]
    node_paths = {}
    for nid, ln in enumerate(open(os.path.join(d, "path_supports"))):  # row = node id (soibean.cpp:476-491 asks the graph itself)
        if nid in seqs:
            node_paths[nid] = [path_names[p] for p, c in enumerate(ln.rstrip("\n")) if c == "1" and p < len(path_names)]

    def text(name):
        p = os.path.join(d, name)
        return open(p).read() if os.path.exists(p) else ""
    return seqs, path_names, node_paths, DamageModel(text("damage5p.prof"), text("damage3p.prof"))


def run(d, penalty=7):
    seqs, path_names, node_paths, dmg = load_inputs(d)
    states = json.load(open(os.path.join(d, "states.json")))
    alns = gamio.read_gam(os.path.join(d, "reads.gam"))
    reads, undefined, all_records = [], [], []
    for r, a in enumerate(alns):
        if a["identity"] == 0:  # :101
            continue
        try:
            pm, recs = analyse_read(seqs, node_paths, path_names, a, dmg, penalty)
        except Undefined as e:
            undefined.append({"read": r, "why": str(e)})
            continue
        except KeyError as e:
            undefined.append({"read": r, "why": "node %s" % e})
            continue
        cnt = []
        for p in range(len(path_names)):
            c = [0] * 25
            for g, rb, sup, _ in recs[p]:
                if sup:
                    c[pair_class(g) * 5 + pair_class(rb)] += 1
            cnt.append(c)
        all_records.append(recs)
        reads.append({"read": r, "pm": [mp.nstr(x, 25) for x in pm], "cnt": cnt})
    freqs = [D(x) for x in FREQS]
    lls = []
    for st in states:
        s = [(c, p, mp.mpf(repr(dist)), mp.mpf(repr(pos)), mp.mpf(repr(theta))) for c, p, dist, pos, theta in st["sources"]]
        lls.append({"sources": st["sources"], "con": st["con"], "loglike": mp.nstr(state_loglike(all_records, s, mp.mpf(repr(st["con"])), freqs), 25)})
    return {"n_alignments": len(alns), "params": {"penalty": penalty, "freqs": [float(x) for x in FREQS]}, "undefined_reads": undefined,
            "reads": reads, "states": lls}


def make(d):
    """Inputs built by tools/pyref_inputs.py (plain seeded Python, no product code): a 12-path tree, reads of both strands with
    substitutions, indels and soft clips, one over-long path name; the reads the reference leaves undefined are taken out."""
    import pyref_inputs as pi
    os.makedirs(d, exist_ok=True)
    gold = os.path.join(ROOT, "tests", "golden", "damageProfiles")
    open(os.path.join(d, "damage5p.prof"), "w").write(open(gold + "/dhigh5p.prof").read())
    open(os.path.join(d, "damage3p.prof"), "w").write(open(gold + "/dhigh3p.prof").read())
    g = pi.variation_graph(seed=61, genome_len=900, n_paths=12)
    real_names = list(g["names"])
    g["names"][7] = "L" * 110  # never "found" among a node's paths (getLCAfromGAM.h:80-88)
    pi.write_hcfiles(d, g)
    alns = pi.simulate_reads(62, g, 120, read_len=50, sub_rate=0.03, indel_rate=0.1, softclip_rate=0.1, reverse_rate=0.5, low_mapq_rate=0.1)
    tmp = os.path.join(d, "reads.gam")
    open(tmp, "wb").write(gamio.write_gam(alns, group=40))
    pairs = [(q, par) for q, par in enumerate(g["parent"]) if par >= 0 and 7 not in (q, par)]
    assert len(pairs) >= 7, real_names
    states = [{"sources": [[pairs[0][0], pairs[0][1], 0.03, 0.4, 1.0]], "con": 0.01},
              {"sources": [[pairs[3][0], pairs[3][1], 0.0, 0.5, 1.0]], "con": 0.01},
              {"sources": [[pairs[1][0], pairs[1][1], 0.03, 0.35, 0.5], [pairs[4][0], pairs[4][1], 0.011, 0.8, 0.3],
                           [pairs[6][0], pairs[6][1], 0.04, 0.02, 0.2]], "con": 0.02}]
    json.dump(states, open(os.path.join(d, "states.json"), "w"))
    for _ in range(3):
        und = {u["read"] for u in run(d)["undefined_reads"]}
        if not und:
            break
        alns = [al for r, al in enumerate(alns) if r not in und]
        open(tmp, "wb").write(gamio.write_gam(alns, group=40))
    out = {"_what": "tools/pyref_sb.py: an independent Python + mpmath (40 digits) restatement of soibean's analyse_GAM tables and "
                    "state likelihood on the inputs beside this file (tools/pyref_inputs.py: plain seeded Python); NOT generated by "
                    "oracle/ or by the product",
           "default": run(d)}
    json.dump(out, open(os.path.join(d, "sb_pyref.json"), "w"), indent=0)
    print("wrote", d, "reads", len(out["default"]["reads"]), "undefined", len(out["default"]["undefined_reads"]),
          [s["loglike"] for s in out["default"]["states"]])


def newick_tree(text):
    """(names, parent_of) of a Newick string's nodes in pre-order (the root first: parent_of[q] < q), unnamed nodes as "N<k>"."""
    text = text.strip().rstrip(";")
    names, parent_of = [], []

    def node(i, par):
        me = len(names)
        names.append(None)
        parent_of.append(par)
        if text[i] == "(":
            i += 1
            while True:
                i = node(i, me)
                if text[i] == ",":
                    i += 1
                    continue
                assert text[i] == ")"
                i += 1
                break
        j = i
        while j < len(text) and text[j] not in ",():":
            j += 1
        names[me] = text[i:j] or "N%d" % me
        i = j
        if i < len(text) and text[i] == ":":
            j = i + 1
            while j < len(text) and text[j] not in ",()":
                j += 1
            i = j
        return i

    end = node(0, -1)
    assert end == len(text), (end, len(text))
    return names, parent_of


def make_full(d):
    """The shape of the reference's own soibean test (test.cpp:243-248: the Ursidae tree, 28 nodes / 15 leaves): a graph of 28 paths drawn
    down THAT tree (tests/golden/trees/Ursidae.new.dnd), 510 reads of 50-75 bp on both strands, states of one, two and three sources
    with branch positions 0 and 1 and a branch of length zero (MCMC.cpp:756,906)."""
    import pyref_inputs as pi
    os.makedirs(d, exist_ok=True)
    gold = os.path.join(ROOT, "tests", "golden", "damageProfiles")
    open(os.path.join(d, "damage5p.prof"), "w").write(open(gold + "/dhigh5p.prof").read())
    open(os.path.join(d, "damage3p.prof"), "w").write(open(gold + "/dhigh3p.prof").read())
    names, parent_of = newick_tree(open(os.path.join(ROOT, "tests", "golden", "trees", "Ursidae.new.dnd")).read())
    assert len(names) == 28 and sum(1 for q in range(28) if q not in parent_of) == 15, (len(names), names)
    g = pi.variation_graph(seed=71, genome_len=1400, n_paths=28, parent_of=parent_of, names=names)
    pi.write_hcfiles(d, g)
    alns = []
    for k, (n, rl) in enumerate(((170, 50), (170, 62), (170, 75))):
        alns += pi.simulate_reads(72 + k, g, n, read_len=rl, sub_rate=0.03, indel_rate=0.08, softclip_rate=0.08, reverse_rate=0.5, low_mapq_rate=0.1, name="r%d_" % k)
    tmp = os.path.join(d, "reads.gam")
    open(tmp, "wb").write(gamio.write_gam(alns, group=64))
    leaf = [q for q in range(28) if q not in parent_of]
    inner = [q for q in range(1, 28) if q in parent_of]
    pr = lambda q: [q, parent_of[q]]
    states = [{"sources": [pr(leaf[0]) + [0.042357, 0.4, 1.0]], "con": 0.01},
              {"sources": [pr(leaf[3]) + [0.0, 0.5, 1.0]], "con": 0.01},                       # a branch of length zero
              {"sources": [pr(inner[2]) + [0.0293, 0.0, 1.0]], "con": 0.02},                   # at the branch's start
              {"sources": [pr(leaf[5]) + [0.0036, 1.0, 1.0]], "con": 0.02},                    # ... and at its end
              {"sources": [pr(leaf[1]) + [0.0442, 0.35, 0.6], pr(leaf[9]) + [0.0303, 0.8, 0.4]], "con": 0.01},
              {"sources": [pr(leaf[2]) + [0.0078, 0.0, 0.5], pr(inner[5]) + [0.0011, 1.0, 0.3], pr(leaf[12]) + [0.0501, 0.25, 0.2]], "con": 0.02},
              {"sources": [pr(leaf[14]) + [0.1128, 0.6, 0.5], pr(leaf[7]) + [0.0, 0.5, 0.3], pr(leaf[10]) + [0.0113, 0.02, 0.2]], "con": 0.015}]
    json.dump(states, open(os.path.join(d, "states.json"), "w"))
    for _ in range(3):
        und = {u["read"] for u in run(d)["undefined_reads"]}
        if not und:
            break
        alns = [al for r, al in enumerate(alns) if r not in und]
        open(tmp, "wb").write(gamio.write_gam(alns, group=64))
    out = {"_what": "tools/pyref_sb.py --make-full: the independent Python + mpmath (40 digits) restatement of soibean's analyse_GAM tables and "
                    "state likelihood at the shape of the reference's own soibean test (the Ursidae tree: 28 paths, 15 leaves), on the inputs beside "
                    "this file (tools/pyref_inputs.py: plain seeded Python); NOT generated by oracle/ or by the product",
           "default": run(d)}
    json.dump(out, open(os.path.join(d, "sb_pyref.json"), "w"), indent=0)
    print("wrote", d, "reads", len(out["default"]["reads"]), "undefined", len(out["default"]["undefined_reads"]),
          [s["loglike"] for s in out["default"]["states"]])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--make")
    ap.add_argument("--make-full", help="the fixture at the shape of the reference's soibean test (Ursidae: 28 paths): tests/golden/sb_pyref_full")
    ap.add_argument("--run")
    ap.add_argument("--out")
    args = ap.parse_args()
    if args.make_full:
        make_full(args.make_full)
    elif args.make:
        make(args.make)
    elif args.run:
        json.dump(run(args.run), open(args.out, "w") if args.out else sys.stdout, indent=0)
