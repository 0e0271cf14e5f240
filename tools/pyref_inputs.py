#!/usr/bin/env python3
"""Inputs for the independent restatements (tools/pyref_hc.py, pyref_euka.py, pyref_sb.py), built here in plain seeded Python:
no product code, no oracle code -- neither the arithmetic nor the input distribution of a pyref fixture is the product's.

  variation_graph()   a small mtDNA-like variation graph: a backbone cut into nodes of 1..4 bases, SNP and short indel bubbles,
                      haplotype paths drawn down a tree (every path inherits its parent's alleles and flips a few)
  write_hcfiles()     graph.gfa (S / L / P lines) + the hcfiles sidecars HaploCart loads (load.cpp:6-58,283-345): graph_paths,
                      path_supports, parsed_pangenome_mapping, mappability.tsv, parents.txt, children.txt
  simulate_reads()    alignments in the GAM codec's form (tests/gamio.py): reads sampled from a path's walk on either strand,
                      with substitutions, insertions, deletions, soft clips and a range of qualities and mapping qualities
  covering_graph()    a graph whose nodes cover the mappings of given alignments (the reference's bundled J2a1a1a1.gam: 81 real
                      giraffe alignments): node bases taken from the reads where an edit matches, random elsewhere
"""
import random

BASES = "ACGT"
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP.get(c, "N") for c in reversed(s))


def variation_graph(seed, genome_len=700, n_paths=64, site_rate=0.16, indel_share=0.15, name_fmt="hg%05d", first_id=1, parent_of=None, names=None):
    """-> dict: seqs {node id: str}, paths [[node ids]], names, pos {node id: 0-based reference coordinate of its first base},
    parent [path index or -1], genome_len.  parent_of / names: a given tree (parent_of[q] < q, -1 for the root) and its nodes'
    names instead of a random one."""
    rng = random.Random(seed)
    genome = "".join(rng.choice(BASES) for _ in range(genome_len))
    elems = []  # ("node", id) shared by every path | ("site", [allele node ids]) -- an allele may be None (deletion allele)
    seqs, pos = {}, {}
    nid = first_id
    p = 0
    while p < genome_len:
        if rng.random() < site_rate and p + 3 < genome_len:
            if rng.random() < indel_share:  # a short indel bubble: the reference bases or nothing
                n = rng.randint(1, 3)
                seqs[nid], pos[nid] = genome[p:p + n], p
                elems.append(("site", [nid, None]))
                nid += 1
                p += n
            else:  # a SNP bubble
                ref = genome[p]
                alt = rng.choice([b for b in BASES if b != ref])
                seqs[nid], pos[nid] = ref, p
                seqs[nid + 1], pos[nid + 1] = alt, p
                elems.append(("site", [nid, nid + 1]))
                nid += 2
                p += 1
        else:
            n = min(rng.randint(1, 4), genome_len - p)
            seqs[nid], pos[nid] = genome[p:p + n], p
            elems.append(("node", nid))
            nid += 1
            p += n
    n_sites = sum(1 for e in elems if e[0] == "site")
    parent, alleles = [-1], [[0] * n_sites]
    for q in range(1, n_paths):
        par = parent_of[q] if parent_of is not None else rng.randrange(q)
        a = list(alleles[par])
        for _ in range(rng.randint(1, max(2, n_sites // 12))):
            a[rng.randrange(n_sites)] ^= 1
        parent.append(par)
        alleles.append(a)
    paths = []
    for q in range(n_paths):
        walk, s = [], 0
        for e in elems:
            if e[0] == "node":
                walk.append(e[1])
            else:
                v = e[1][alleles[q][s]]
                s += 1
                if v is not None:
                    walk.append(v)
        paths.append(walk)
    return {"seqs": seqs, "paths": paths, "names": list(names) if names is not None else [name_fmt % q for q in range(n_paths)], "pos": pos, "parent": parent,
            "genome_len": genome_len}


def write_gfa(path, g, with_paths=True):
    """with_paths False: no P lines (thousands of paths of thousands of steps: the hcfiles sidecars graph_paths / path_supports say
    which node lies on which path, as they do for the reference, load.cpp:43-58,283-300)."""
    with open(path, "w") as f:
        f.write("H\tVN:Z:1.0\n")
        for nid in sorted(g["seqs"]):
            f.write("S\t%d\t%s\n" % (nid, g["seqs"][nid]))
        links = set()
        for walk in g["paths"]:
            links.update(zip(walk[:-1], walk[1:]))
        for a, b in sorted(links):
            f.write("L\t%d\t+\t%d\t+\t0M\n" % (a, b))
        if with_paths:
            for name, walk in zip(g["names"], g["paths"]):
                f.write("P\t%s\t%s\t*\n" % (name, ",".join("%d+" % v for v in walk)))


def write_hcfiles(d, g, mappability=None, gfa_paths=True):
    """The files HaploCart's loaders read (load.cpp), beside graph.gfa."""
    import os
    os.makedirs(d, exist_ok=True)
    write_gfa(os.path.join(d, "graph.gfa"), g, with_paths=gfa_paths)
    P = len(g["paths"])
    with open(os.path.join(d, "graph_paths"), "w") as f:
        for n in g["names"]:
            f.write(n + "\n")
    member = {nid: ["0"] * P for nid in g["seqs"]}
    for q, walk in enumerate(g["paths"]):
        for v in walk:
            member[v][q] = "1"
    with open(os.path.join(d, "path_supports"), "w") as f:  # row = node id (load.cpp:283-300)
        for nid in range(0, max(g["seqs"]) + 1):
            f.write("".join(member.get(nid, ["0"] * P)) + "\n")
    with open(os.path.join(d, "parsed_pangenome_mapping"), "w") as f:
        for nid in sorted(g["seqs"]):
            f.write("%d\t%d\n" % (nid, g["pos"][nid]))
    with open(os.path.join(d, "mappability.tsv"), "w") as f:
        rows = mappability or [(0, g["genome_len"] + 2, 1.0)]
        for lo, hi, v in rows:
            f.write("chrM\t%d\t%d\t%s\n" % (lo, hi, repr(float(v)) if v != int(v) else "%d" % int(v)))
    kids = {q: [] for q in range(P)}
    for q, par in enumerate(g["parent"]):
        if par >= 0:
            kids[par].append(q)
    with open(os.path.join(d, "parents.txt"), "w") as f:  # a name, then its ancestors from the parent up
        for q in range(P):
            chain, x = [], g["parent"][q]
            while x >= 0:
                chain.append(g["names"][x])
                x = g["parent"][x]
            f.write(" ".join([g["names"][q]] + chain) + "\n")
    with open(os.path.join(d, "children.txt"), "w") as f:  # a name, then its children
        for q in range(P):
            f.write(" ".join([g["names"][q]] + [g["names"][c] for c in kids[q]]) + "\n")


def simulate_reads(seed, g, n_reads, read_len=120, sub_rate=0.02, indel_rate=0.15, softclip_rate=0.1, reverse_rate=0.5, low_mapq_rate=0.3,
                   name="r", paths=None):
    """Alignments (tests/gamio.py's dicts) of reads drawn from the graph's paths.  indel_rate / softclip_rate: share of reads that
    carry one insertion or deletion / a soft clip at one end."""
    rng = random.Random(seed)
    seqs = g["seqs"]
    out = []
    for r in range(n_reads):
        q = rng.choice(paths) if paths else rng.randrange(len(g["paths"]))
        walk = g["paths"][q]
        hap = [(v, k) for v in walk for k in range(len(seqs[v]))]  # (node, offset) per haplotype base
        L = min(read_len + rng.randint(-read_len // 4, read_len // 4), len(hap))
        a0 = rng.randrange(0, len(hap) - L + 1)
        seg = hap[a0:a0 + L]
        rev = rng.random() < reverse_rate
        # the read's columns in READ orientation: (node, offset on the traversed strand, graph base on that strand)
        cols = []
        for v, k in (reversed(seg) if rev else seg):
            n = len(seqs[v])
            cols.append((v, n - 1 - k, COMP[seqs[v][k]]) if rev else (v, k, seqs[v][k]))
        # edits per column: 'M' match, ('X', base) substitution, 'D' deletion; insertions ride in front of a column
        ops = [["M", None] for _ in cols]
        for i in range(len(cols)):
            if rng.random() < sub_rate:
                ops[i] = ["X", rng.choice([b for b in BASES if b != cols[i][2]])]
        ins_at, ins_seq = None, ""
        if rng.random() < indel_rate and len(cols) > 20:
            i = rng.randrange(5, len(cols) - 5)
            if rng.random() < 0.5:
                for j in range(i, min(i + rng.randint(1, 3), len(cols) - 2)):
                    ops[j] = ["D", None]
            else:
                ins_at, ins_seq = i, "".join(rng.choice(BASES) for _ in range(rng.randint(1, 3)))
        clip5 = clip3 = ""
        if rng.random() < softclip_rate:
            s = "".join(rng.choice(BASES) for _ in range(rng.randint(3, 9)))
            if rng.random() < 0.5:
                clip5 = s
            else:
                clip3 = s
        # mappings: one per node run, edits merged
        mappings, read = [], []
        i = 0
        while i < len(cols):
            v = cols[i][0]
            j = i
            while j < len(cols) and cols[j][0] == v and (j == i or cols[j][1] == cols[j - 1][1] + 1):
                j += 1
            edits = []

            def add(fl, tl, sq):
                if edits and edits[-1]["sequence"] == b"" and sq == b"" and edits[-1]["from_length"] == edits[-1]["to_length"] and fl == tl:
                    edits[-1]["from_length"] += fl
                    edits[-1]["to_length"] += tl
                elif edits and fl > 0 and tl == 0 and edits[-1]["to_length"] == 0 and edits[-1]["from_length"] > 0:
                    edits[-1]["from_length"] += fl
                else:
                    edits.append({"from_length": fl, "to_length": tl, "sequence": sq})
            if i == 0 and clip5:
                add(0, len(clip5), clip5.encode())
                read.append(clip5)
            for c in range(i, j):
                if ins_at == c:
                    add(0, len(ins_seq), ins_seq.encode())
                    read.append(ins_seq)
                op, b = ops[c]
                if op == "M":
                    add(1, 1, b"")
                    read.append(cols[c][2])
                elif op == "X":
                    edits.append({"from_length": 1, "to_length": 1, "sequence": b.encode()})
                    read.append(b)
                else:
                    add(1, 0, b"")
            if j == len(cols) and clip3:
                add(0, len(clip3), clip3.encode())
                read.append(clip3)
            mappings.append({"position": {"node_id": v, "offset": cols[i][1], "is_reverse": rev}, "edit": edits, "rank": len(mappings) + 1})
            i = j
        seq = "".join(read)
        n_match = sum(1 for o in ops if o[0] == "M")
        qual = bytes(rng.choice([2, 11, 20, 25, 30, 33, 37, 40, 41]) if rng.random() < 0.3 else rng.randint(20, 41) for _ in seq)
        mq = rng.randrange(0, 60) if rng.random() < low_mapq_rate else 60
        out.append({"sequence": seq.encode(), "path": {"name": b"", "mapping": mappings}, "name": ("%s%d" % (name, r)).encode(), "quality": qual,
                    "mapping_quality": mq, "score": n_match, "identity": n_match / max(1, len(seq))})
    return out


def covering_graph(seed, alns, n_paths=24, name_fmt="hg%05d"):
    """A graph for alignments that came without one: every node their mappings name, long enough for every edit; a base under a
    match edit is the read's base there (complemented for a reverse mapping), the rest random.  Paths: a backbone through all
    the nodes in id order plus random subsets of it (membership is what the likelihood path asks of a path)."""
    rng = random.Random(seed)
    need, known = {}, {}
    for a in alns:
        seq = a["sequence"].decode()
        rp = 0
        for m in a["path"]["mapping"]:
            v, off, rev = m["position"]["node_id"], m["position"].get("offset", 0), m["position"].get("is_reverse", False)
            for e in m["edit"]:
                fl, tl = e["from_length"], e["to_length"]
                if fl == tl and not e["sequence"]:
                    for k in range(fl):
                        if rp + k < len(seq):
                            known.setdefault(v, {})[(off + k, rev)] = seq[rp + k]
                off += fl
                rp += tl
            need[v] = max(need.get(v, 1), off)
    seqs = {}
    for v, n in need.items():
        s = [rng.choice(BASES) for _ in range(n)]
        for (k, rev), b in known.get(v, {}).items():  # (reverse-strand offsets count from the node's end)
            if rev and 0 <= n - 1 - k < n:
                s[n - 1 - k] = COMP.get(b, "N")
        for (k, rev), b in known.get(v, {}).items():
            if not rev and k < n:
                s[k] = b
        seqs[v] = "".join(s)
    order = sorted(seqs)
    paths, parent = [order], [-1]
    for q in range(1, n_paths):
        par = parent_of[q] if parent_of is not None else rng.randrange(q)
        keep = [v for v in paths[par] if rng.random() < 0.97]
        paths.append(keep if keep else order)
        parent.append(par)
    pos, p = {}, 0
    for v in order:
        pos[v] = p
        p += len(seqs[v])
    return {"seqs": seqs, "paths": paths, "names": [name_fmt % q for q in range(n_paths)], "pos": pos, "parent": parent, "genome_len": p}
