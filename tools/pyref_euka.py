#!/usr/bin/env python3
"""A second, independent restatement of euka's per-read path -- Python + mpmath (40 digits), written from the reference's
sources, NOT from oracle/ (see tools/pyref_hc.py for why; the alignment reconstruction and the GAM / GFA readers are shared
with it).

What it follows (paths under /root/reference/src/):
    readGAM_Euka.h:120-470     clade of a read (first mapping's node against every clade's bins: the LAST hit wins, clade 0 when
                               none), model 1 (pre-damage base by the clade's pairwise distance and the transition /
                               transversion table, post-damage base by the position's substitution matrix, sequencing error
                               marginalised) and model 2 per alignment column; n, the coordinate on the fragment
    readGAM_Euka.h:471-549     clade_like / clade_not_like, the detection rule (ratio > 1 and mapping quality > MINIMUMMQ),
                               count, bin coverage (1 / mappings per mapping whose node lies in a bin of the clade)
    baseshift.cpp:57-88        the (graph base, read base) counts of the first and last lengthToProf columns
    damage.cpp:18-36,42-260    substitution matrices per fragment length and position (the row of the 5' or of the 3' profile
                               whose diagonal element is smaller), identity where no profile is given
    miscfunc.h:84-136,216      the profile reader (12 rates per line, a 13th field dropped), get_p_incorrectly_mapped
    Euka.cpp:38-52,446-486     qscore_vec (0.25 below Q2), base_freq, t_T_ratio, rare bases
    load.cpp:70-160            euka_db.bins (name, then lo hi entropy triples), euka_db.clade (id name dist ...)
    vgan_utils.h:6-79          reconstruct_graph_sequence (tools/pyref_hc.py)

Third-party pieces not in the tree (libgab): dimer2indexInt (the profile's column order A>C A>G A>T C>A ... T>G),
oplusInitnatl (log-sum-exp whose first argument 0 means "nothing yet"), allTokens.

Where the reference is undefined this script does not guess; such reads are listed in "undefined_reads" and left out: a
fragment shorter than 15 or longer than 1000 (subDeamDiNuc[Lseq] has no rows), a coordinate n at or past the fragment's
length on a column that needs the matrix, a quality index past the quality string or a quality of 100 or more, a base
other than A C G T in the counted columns of baseshift (dna2int is uninitialised there), base_freq of anything but
A C G N T, fewer alignment columns than lengthToProf.

Usage (build container only):
    python tools/pyref_euka.py --make tests/golden/euka_pyref
    python tools/pyref_euka.py --run DIR [--out FILE]
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
from pyref_hc import Undefined, load_gfa, reconstruct_graph_sequence, signed_char  # noqa: E402

mp.mp.dps = 40


def D(x):
    """A literal of the C++ source: the double it is there, not the decimal it is written as (log(0.9999999) differs from the
    log of the decimal by 1e-9 relative)."""
    return mp.mpf(float(x))
MINLENGTHFRAGMENT, MAXLENGTHFRAGMENT = 15, 1000  # damage.h:42-43
ACGT = "ACGT"


# ----------------------------------------------------------------------------------------------------------------- inputs
def load_clades(path):  # load.cpp:108-158 (six objects per line; index c_n * 6 + 1 is clade c_n's)
    out = []
    for ln in open(path):
        t = ln.split()
        if not t:
            continue
        assert len(t) == 6
        out.append({"id": int(t[0]), "name": t[1], "dist": D(t[2])})  # stod
    return out


def load_bins(path):  # load.cpp:70-95
    chunks = []
    for ln in open(path):
        t = ln.split()
        if not ln.strip():
            chunks.append([])
            continue
        chunks.append([(stoi(t[j]), stoi(t[j + 1]), float(t[j + 2])) for j in range(1, len(t) - 2, 3)])
    return chunks


def stoi(s):  # std::stoi: the longest integer prefix ("1836.0" -> 1836: the shipped euka_db.bins writes its node ids so)
    import re
    m = re.match(r"\s*[+-]?\d+", s)
    if not m:
        raise ValueError("stoi: no conversion: %r" % s)
    return int(m.group(0))


def read_rates(text):  # miscfunc.h:84-136: header, then 12 rates per line
    rows = []
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    for ln in lines[1:]:
        f = ln.split("\t")
        if len(f) == 13:
            f.pop()
        assert len(f) == 12
        rows.append([mp.mpf(x) for x in f])
    return rows


def dimer2index(n1, n2):  # libgab: A>C A>G A>T C>A C>G C>T G>A G>C G>T T>A T>C T>G
    return n1 * 3 + (n2 if n2 < n1 else n2 - 1)


def end_matrices(text):  # damage.cpp:62-101 (5') and :104-142 (3'): [position][from][to], the last row repeated
    if text:
        rates = read_rates(text)
    else:
        rates = [[mp.mpf(0)] * 12 for _ in range(MAXLENGTHFRAGMENT)]
    sub = []
    for r in rates:
        m = [[mp.mpf(0)] * 4 for _ in range(4)]
        for n1 in range(4):
            ident = mp.mpf(1)
            for n2 in range(4):
                if n1 == n2:
                    continue
                ident -= r[dimer2index(n1, n2)]
                m[n1][n2] = r[dimer2index(n1, n2)]
            if ident < 0:
                raise ValueError("identity probability below 0")
            m[n1][n1] = ident
        sub.append(m)
    i = len(sub) - 1
    while i < MAXLENGTHFRAGMENT:  # :99-101
        sub.append(sub[-1])
        i += 1
    return sub


class DamageModel:
    def __init__(self, text5, text3):
        self.s5, self.s3 = end_matrices(text5), end_matrices(text3)

    def row(self, L, l, b1):  # damage.cpp:18-36 through :238-256: subDeamDiNuc[L][l].p[b1]
        f1, f2 = self.s5[l][b1], self.s3[L - l - 1][b1]
        return f1 if f1[b1] <= f2[b1] else f2  # "if (f1[b] == min(f1[b], f2[b])) use f1"


# ------------------------------------------------------------------------------------------------------------------ tables
def qscore(Q):  # Euka.cpp:38-52
    return mp.mpf(10) ** mp.mpf(float((-1 * Q) * 0.1)) if Q >= 2 else D(0.25)


BASE_FREQ = {"A": mp.log(D(0.362815)), "C": mp.log(D(0.207743)), "G": mp.log(D(0.116809)),
             "N": mp.log(D(0.25)), "T": mp.log(D(0.312435))}  # Euka.cpp:446-450
TS, TV = D(0.95238), D(0.02381)
T_T_RATIO = {a: {b: (mp.mpf(1) if a == b else (TS if {a, b} in ({"A", "G"}, {"C", "T"}) else TV)) for b in ACGT} for a in ACGT}
RARE = set("WMKRYBDHV")  # Euka.cpp:472-480
KNOWN = RARE | set("ACGTSN-")  # the letters rare_bases is given a value for (:472-486), and the gap tested before it


def oplus_init(x, y):  # libgab oplusInitnatl: the first term of a sum is taken as it is
    if x == 0:
        return y
    hi, lo = (x, y) if x > y else (y, x)
    return hi + mp.log1p(mp.exp(lo - hi))


def p_incorrectly_mapped(Q):  # miscfunc.h:216
    return mp.mpf(10) ** mp.mpf(float((-1 * Q) * 0.1))


# --------------------------------------------------------------------------------------------------------------- per read
def clade_of(chunks, n_index):  # readGAM_Euka.h:120-132
    c_n = 0
    for i, bins in enumerate(chunks):
        for lo, hi, _ in bins:
            if lo <= n_index <= hi:
                c_n = i
    return c_n


def baseshift_columns(graph_seq, read_seq, ltp):  # baseshift.cpp:57-88: (p, index) pairs to count
    out = []
    for p in range(2 * ltp):
        pos = p if p < ltp else len(graph_seq) - 2 * ltp + p
        pos_r = p if p < ltp else len(read_seq) - 2 * ltp + p
        if pos < 0 or pos >= len(graph_seq) or pos_r < 0 or pos_r >= len(read_seq):
            raise Undefined("baseshift: fewer columns than lengthToProf")
        gb, rb = graph_seq[pos].upper(), read_seq[pos_r].upper()
        if gb in "SI-N" or rb in "SI-N":
            continue
        if gb not in ACGT or rb not in ACGT:
            raise Undefined("baseshift: dna2int of %r / %r is uninitialised" % (gb, rb))
        out.append((p, ACGT.index(gb) * 4 + ACGT.index(rb)))
    return out


def read_models(aln, graph_seq, read_seq, pair_dist, dmg):
    """in_clade_lik, not_in_clade_lik (readGAM_Euka.h:175-470)."""
    seq, qual = aln["sequence"], aln["quality"]
    Lseq = len(seq)
    isrev = aln["path"]["mapping"][0]["position"]["is_reverse"]
    n = Lseq - 1 if isrev else 0
    in_lik = not_lik = mp.mpf(0)
    softclips = 0

    def quality(m):
        if m > len(qual):
            raise Undefined("quality index %d past a string of %d" % (m, len(qual)))
        q = signed_char(qual[m]) if m < len(qual) else 0  # std::string: s[size()] is '\0'
        if q < 0 or q >= 100:
            raise Undefined("qscore_vec[%d]" % q)
        return q

    for m in range(len(graph_seq)):
        g = graph_seq[m]
        if m >= len(read_seq):
            raise Undefined("read_seq shorter than graph_seq")
        r = read_seq[m]
        if g not in KNOWN or r not in KNOWN:
            raise Undefined("rare_bases[%r / %r] is uninitialised" % (g, r))
        if g == "N" or r == "N":
            if r not in BASE_FREQ:
                raise Undefined("base_freq[%r]" % r)
            l1 = l2 = BASE_FREQ[r]
        elif g == "-" or r == "-":
            l1, l2 = mp.log(D(0.002)), mp.log(D(0.2))
        elif g in RARE or r in RARE:
            l1, l2 = mp.log((1 - pair_dist) * D(0.001)), mp.log(D(0.001))
        elif g == "S" or r == "S":
            q = quality(m)
            softclips += 1
            l1 = mp.log(1 - qscore(q)) if softclips % 3 == 0 else mp.log(qscore(q) / 3)
            l2 = mp.log(D(0.25))
        else:
            q = quality(m)
            if g not in ACGT:
                raise Undefined("t_T_ratio[%r]" % g)
            pre = [(1 - pair_dist) if ACGT[o] == g else pair_dist * T_T_RATIO[g][ACGT[o]] for o in range(4)]
            if Lseq < MINLENGTHFRAGMENT or Lseq > MAXLENGTHFRAGMENT or n < 0 or n >= Lseq:
                raise Undefined("subDeamDiNuc[%d][%d]" % (Lseq, n))
            post = [mp.mpf(0)] * 4
            for d in range(4):
                for o in range(4):
                    post[d] += pre[o] * dmg.row(Lseq, n, o)[d]
            marg = mp.mpf(0)
            e = qscore(q)
            for d in range(4):
                marg = oplus_init(marg, mp.log(post[d] * (1 - e)) if ACGT[d] == r else mp.log(post[d] * (e / 3)))
            l1 = marg
            l2 = mp.log(1 - D(0.25536)) if g == r else mp.log(D(0.25536))
        in_lik += l1
        not_lik += l2
        if r != "-":
            n += -1 if isrev else 1
            if n < 0:
                n = 1 << 32  # unsigned wrap: any further use is out of range
    return in_lik, not_lik


def run(d, min_mapq=29, ltp=5):
    seqs = load_gfa(os.path.join(d, "graph.gfa"))
    clades = load_clades(os.path.join(d, "euka_db.clade"))
    chunks = load_bins(os.path.join(d, "euka_db.bins"))

    def text(name):
        p = os.path.join(d, name)
        return open(p).read() if os.path.exists(p) else ""
    dmg = DamageModel(text("damage5p.prof"), text("damage3p.prof"))
    alns = gamio.read_gam(os.path.join(d, "reads.gam"))
    n_cl = len(chunks)
    count = [0] * n_cl
    baseshift = [[[0] * 16 for _ in range(2 * ltp)] for _ in range(n_cl)]
    bin_cov = [[mp.mpf(0)] * len(b) for b in chunks]
    reads, undefined = [], []
    for r, a in enumerate(alns):
        if a["identity"] == 0:  # readGAM_Euka.h:84
            continue
        try:
            maps = a["path"]["mapping"]
            if not maps:
                raise Undefined("no mapping")
            c_n = clade_of(chunks, maps[0]["position"]["node_id"])
            graph_seq, read_seq, _ = reconstruct_graph_sequence(seqs, a["path"])
            shifts = baseshift_columns(graph_seq, read_seq, ltp)
            in_lik, not_lik = read_models(a, graph_seq, read_seq, clades[c_n]["dist"], dmg)
        except Undefined as e:
            undefined.append({"read": r, "why": str(e)})
            continue
        except KeyError as e:  # a node the graph does not have
            undefined.append({"read": r, "why": "node %s" % e})
            continue
        for p, ix in shifts:
            baseshift[c_n][p][ix] += 1
        mq = a["mapping_quality"]
        like = (1 - p_incorrectly_mapped(mq)) * mp.exp(in_lik - oplus_init(in_lik, not_lik))
        passed = (in_lik - not_lik > 1) and mq > min_mapq
        if passed:
            count[c_n] += 1
            for mp_ in maps:
                nid = mp_["position"]["node_id"]
                for j, (lo, hi, _) in enumerate(chunks[c_n]):
                    if lo <= nid <= hi:
                        bin_cov[c_n][j] += mp.mpf(1) / len(maps)
        reads.append({"read": r, "clade": c_n, "in_lik": mp.nstr(in_lik, 25), "out_lik": mp.nstr(not_lik, 25), "like": mp.nstr(like, 25),
                      "not_like": mp.nstr(1 - like, 25), "pass": bool(passed)})
    return {"n_alignments": len(alns), "params": {"min_mapq": min_mapq, "length_to_prof": ltp}, "undefined_reads": undefined,
            "reads": reads, "clade_count": count, "baseshift": baseshift, "bin_cov": [[mp.nstr(x, 25) for x in b] for b in bin_cov]}


# ---------------------------------------------------------------------------------------------------------- fixture writer
def make(d):
    """Inputs built by tools/pyref_inputs.py (plain seeded Python, no product code): six clades, each a small variation graph of
    its own over a contiguous range of node ids, with the clade and bin tables euka loads (load.cpp:70-158); reads of both strands
    with substitutions, indels and soft clips, rewritten with the cases a plain simulator does not draw: an N and a rare base
    in the graph, mapping qualities around the threshold and 0, qualities below Q2."""
    import random
    import pyref_inputs as pi
    os.makedirs(d, exist_ok=True)
    gold = os.path.join(ROOT, "tests", "golden", "damageProfiles")
    t5, t3 = open(gold + "/dhigh5p.prof").read(), open(gold + "/dhigh3p.prof").read()
    open(os.path.join(d, "damage5p.prof"), "w").write(t5)
    open(os.path.join(d, "damage3p.prof"), "w").write(t3)
    rng = random.Random(91)
    seqs, clades, bins, alns = {}, [], [], []
    first = 1
    for c in range(6):
        g = pi.variation_graph(seed=910 + c, genome_len=150, n_paths=4, first_id=first)
        ids = sorted(g["seqs"])
        lo, hi = ids[0], ids[-1]
        seqs.update(g["seqs"])
        clades.append((c, "clade%03d" % c, rng.uniform(0.05, 0.2), lo, hi))
        width = max(3, (hi - lo + 1) // 9)
        row = []
        for j in range(10):
            b0 = lo + j * (hi - lo + 1) // 10
            row.append((b0, min(hi, b0 + width), rng.uniform(1.1, 1.35)))
        bins.append(row)
        alns += pi.simulate_reads(920 + c, g, 44, read_len=60, sub_rate=0.03, indel_rate=0.12, softclip_rate=0.15, reverse_rate=0.5,
                                  low_mapq_rate=0.2, name="c%d_" % c)
        first = hi + 1
    some = sorted(seqs)
    seqs[some[7]] = "N" + seqs[some[7]][1:]       # an unresolved base
    seqs[some[31]] = seqs[some[31]][:-1] + "R"    # a base outside ACGTN
    rng.shuffle(alns)
    with open(os.path.join(d, "graph.gfa"), "w") as f:
        f.write("H\tVN:Z:1.0\n")
        for nid in sorted(seqs):
            f.write("S\t%d\t%s\n" % (nid, seqs[nid]))
    with open(os.path.join(d, "euka_db.clade"), "w") as f:
        for c, name, dist, lo, hi in clades:
            f.write("%d %s %.17g 1 %d %d\n" % (c, name, dist, lo, hi))
    with open(os.path.join(d, "euka_db.bins"), "w") as f:
        for (c, name, _, _, _), row in zip(clades, bins):
            t = [name]
            for lo, hi, e in row:
                t += [str(lo), str(hi), "%.17g" % e]
            f.write(" ".join(t) + "\n")
    for r, al in enumerate(alns):
        if r % 9 == 2:
            al["mapping_quality"] = rng.randint(27, 32)
        if r % 13 == 4:
            al["mapping_quality"] = 0
        if r % 11 == 6 and len(al["quality"]) > 20:
            q = bytearray(al["quality"])
            q[7], q[8] = 0, 1  # below Q2: 0.25
            al["quality"] = bytes(q)
    tmp = os.path.join(d, "reads.gam")
    open(tmp, "wb").write(gamio.write_gam(alns, group=50))
    for _ in range(3):  # the reads whose treatment the reference leaves undefined are taken out: the sums below must not depend on them
        und = {u["read"] for u in run(d)["undefined_reads"]} | {u["read"] for u in run(d, min_mapq=0, ltp=3)["undefined_reads"]}
        if not und:
            break
        alns = [al for r, al in enumerate(alns) if r not in und]
        open(tmp, "wb").write(gamio.write_gam(alns, group=50))
    out = {"_what": "tools/pyref_euka.py: an independent Python + mpmath (40 digits) restatement of euka's per-read path on the inputs "
                    "beside this file (tools/pyref_inputs.py: plain seeded Python); NOT generated by oracle/ or by the product",
           "default": run(d), "other_thresholds": run(d, min_mapq=0, ltp=3)}
    json.dump(out, open(os.path.join(d, "euka_pyref.json"), "w"), indent=0)
    print("wrote", d, "reads", len(out["default"]["reads"]), "undefined", len(out["default"]["undefined_reads"]),
          "passed", sum(x["pass"] for x in out["default"]["reads"]))


def make_full(d):
    """The fixture with the SHIPPED tables (share/vgan/euka_dir/euka_db.{clade,bins}, copied under tests/golden/euka_dir/ as data):
    335 clades over node ids up to 6.9 million.  The euka graph itself is not shipped, so every clade gets a small variation graph of
    its own (tools/pyref_inputs.py) placed at the first node id of one of ITS coverage bins, and two reads drawn from it."""
    import random
    import shutil
    import pyref_inputs as pi
    os.makedirs(d, exist_ok=True)
    gold = os.path.join(ROOT, "tests", "golden")
    for name in ("euka_db.clade", "euka_db.bins"):
        shutil.copyfile(os.path.join(gold, "euka_dir", name), os.path.join(d, name))
    t5, t3 = open(gold + "/damageProfiles/dhigh5p.prof").read(), open(gold + "/damageProfiles/dhigh3p.prof").read()
    open(os.path.join(d, "damage5p.prof"), "w").write(t5)
    open(os.path.join(d, "damage3p.prof"), "w").write(t3)
    clades = load_clades(os.path.join(d, "euka_db.clade"))
    chunks = load_bins(os.path.join(d, "euka_db.bins"))
    assert len(clades) == 335 and len(chunks) == 335
    rng = random.Random(191)
    seqs, alns = {}, []
    for c, bins in enumerate(chunks):
        j = rng.randrange(len(bins))
        first = bins[j][0]
        g = pi.variation_graph(seed=1910 + c, genome_len=110, n_paths=3, first_id=max(first, 1))
        seqs.update(g["seqs"])
        alns += pi.simulate_reads(2920 + c, g, 2, read_len=64, sub_rate=0.03, indel_rate=0.12, softclip_rate=0.15, reverse_rate=0.5,
                                  low_mapq_rate=0.2, name="c%d_" % c)
    rng.shuffle(alns)
    with open(os.path.join(d, "graph.gfa"), "w") as f:
        f.write("H\tVN:Z:1.0\n")
        for nid in sorted(seqs):
            f.write("S\t%d\t%s\n" % (nid, seqs[nid]))
    for r, al in enumerate(alns):
        if r % 9 == 2:
            al["mapping_quality"] = rng.randint(27, 32)
        if r % 13 == 4:
            al["mapping_quality"] = 0
    tmp = os.path.join(d, "reads.gam")
    open(tmp, "wb").write(gamio.write_gam(alns, group=100))
    for _ in range(3):
        und = {u["read"] for u in run(d)["undefined_reads"]}
        if not und:
            break
        alns = [al for r, al in enumerate(alns) if r not in und]
        open(tmp, "wb").write(gamio.write_gam(alns, group=100))
    out = {"_what": "tools/pyref_euka.py --make-full: the independent Python + mpmath restatement of euka's per-read path with the SHIPPED "
                    "clade and bin tables (335 clades); graph and reads by tools/pyref_inputs.py; NOT generated by oracle/ or by the product",
           "default": run(d)}
    json.dump(out, open(os.path.join(d, "euka_pyref.json"), "w"), indent=0)
    print("wrote", d, "reads", len(out["default"]["reads"]), "undefined", len(out["default"]["undefined_reads"]),
          "passed", sum(x["pass"] for x in out["default"]["reads"]), "clades hit", sum(1 for x in out["default"]["clade_count"] if x))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--make")
    ap.add_argument("--make-full", help="the fixture with the shipped 335-clade tables: tests/golden/euka_pyref_full")
    ap.add_argument("--run")
    ap.add_argument("--out")
    args = ap.parse_args()
    if args.make:
        make(args.make)
    elif args.make_full:
        make_full(args.make_full)
    elif args.run:
        res = run(args.run)
        json.dump(res, open(args.out, "w") if args.out else sys.stdout, indent=0)
