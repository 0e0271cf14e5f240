#!/usr/bin/env python3
"""A second, independent restatement of HaploCart's per-read likelihood path -- Python + mpmath (40 digits), written from the
reference's sources, NOT from oracle/ (the C++ long double restatement the test-suite otherwise leans on).  It exists so
that the numbers every parity test is held against do not rest on one author path alone: two restatements in two languages
with two arithmetics agreeing to 1e-15 is a far smaller common-mode risk than one.

What it follows, line by line (paths under /root/reference/src/):
    vgan_utils.h:6-79             reconstruct_graph_sequence (with vg's path_string / edit_is_* semantics, see below)
    update_likelihood.cpp:19-53   the loop over mappings, the quality window, the sticky Q >= 90 switch
    process_mapping.cpp:4-91      supported / unsupported sums per path
    get_p_obs_base.cpp:3-69       epsilon per base, the per-region mutation rate (its integer divisions kept)
    miscfunc.h:180-216            get_p_seq_error, get_qscore_vec
    haplocart_functions.cpp:81-107 background frequencies, incorrect_mapping_vec
    HaploCart.cpp:408-424         the accumulate over reads (identity < 1e-10 skipped), argmax
    get_posterior.cpp:36-127      clade posteriors (libgab's oplusInitnatl restated from its published behaviour)
    load.cpp:6-58,283-345         mappability.tsv, parsed_pangenome_mapping (+1), graph_paths, path_supports, parents / children

Third-party pieces that are not in the reference tree (vg: path_string, edit_is_match / _sub / _insertion / _deletion,
get_sequence of a reversed handle; libgab: oplusInitnatl, isValidDNA) are restated from their published semantics; the
reconstruction is pinned on the reference's own 10 known-answer cases (tests/golden/reconstruct/expected.json, src/test.cpp:855-994)
by `--check-kats`.

Where the reference is undefined the run stops being a restatement; this script does NOT guess: a read that indexes a string
past its end inside substr / insert, names an unknown node, has more mappings than edits or a mapping quality >= 100 is
reported in "undefined_reads" and left out (the product counts the same reads as n_bad / clamps them).  ONE definition is
shared with the build because every multi-mapping read needs it (include/vgan_gpu.h): a quality index at or past
the end of the quality string reads as 0.  The out-of-bounds parent_vec[j-1] at j = 0 (get_posterior.cpp:110,117) is read as
"different" (SURVEY Q9).

Usage (in the build container; the GPU box only loads the JSON):
    python tools/pyref_hc.py --make tests/golden/hc_pyref        # writes the inputs (GFA, sidecars, GAM) and hc_pyref.json
    python tools/pyref_hc.py --run DIR [--out FILE]              # recomputes the JSON from the files in DIR
    python tools/pyref_hc.py --check-kats                        # the reconstruction against the reference's KATs
"""
import argparse
import math
import gzip
import json
import os
import sys

import mpmath as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gamio  # noqa: E402  (the test-side GAM decoder / encoder: no product code)

mp.mp.dps = 40


class Undefined(Exception):
    """The reference's behaviour on this read is undefined (out-of-range access, uncaught throw in a noexcept function)."""


# ----------------------------------------------------------------------------------------------------------------- inputs
def _open(path):
    if os.path.exists(path):
        return open(path, "rt")
    if os.path.exists(path + ".gz"):
        return gzip.open(path + ".gz", "rt")
    raise FileNotFoundError(path)


def load_gfa(path):
    seqs = {}
    for ln in open(path):
        t = ln.rstrip("\n").split("\t")
        if t and t[0] == "S":
            seqs[int(t[1])] = t[2]
    return seqs


def load_hcfiles(d, supports_as_numpy=False, supports_lists=True):
    """load.cpp:6-58,283-345"""
    mappabilities = []
    for ln in _open(os.path.join(d, "mappability.tsv")):
        t = ln.split()
        if len(t) < 4:
            continue
        for _ in range(int(t[1]), int(t[2])):  # load.cpp:18-20
            mappabilities.append(float(t[3]))
    pangenome_map = {}
    for ln in _open(os.path.join(d, "parsed_pangenome_mapping")):
        t = ln.split()
        if len(t) >= 2 and t[0] not in pangenome_map:  # map::insert keeps the first
            pangenome_map[t[0]] = int(t[1]) + 1  # load.cpp:37
    path_names = [ln.split()[0] for ln in _open(os.path.join(d, "graph_paths")) if ln.split()]  # whole first token
    supports, rows_np = [], []
    for ln in _open(os.path.join(d, "path_supports")):
        supports.append([c == "1" for c in ln.rstrip("\n")] if supports_lists else None)  # row index = line number = node id (load.cpp:283-300)
        if supports_as_numpy:
            import numpy as np
            rows_np.append(np.frombuffer(ln.rstrip("\n").encode(), np.uint8) == ord("1"))

    def relatives(name):
        rel = {}
        for ln in _open(os.path.join(d, name)):
            t = ln.split()
            if not t:
                continue
            if t[0] not in rel:
                rel[t[0]] = [x for x in t[1:] if "[" not in x]
        return rel
    out = {"mappabilities": mappabilities, "pangenome_map": pangenome_map, "path_names": path_names, "supports": supports,
           "n_support_rows": len(supports), "parents": relatives("parents.txt"), "children": relatives("children.txt")}
    if supports_as_numpy:
        import numpy as np
        width = max(len(x) for x in rows_np)
        out["supports_np"] = np.stack([np.pad(x, (0, width - len(x))) for x in rows_np])
    return out


# -------------------------------------------------------------------------------------- vg / libgab semantics (published)
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "a": "t", "c": "g", "g": "c", "t": "a"}


def node_sequence(seqs, node_id, is_reverse):
    """bdsg::ODGI::get_sequence(get_handle(id, is_reverse)): the reverse complement on the reverse strand."""
    if node_id not in seqs:
        raise Undefined("unknown node %d" % node_id)
    s = seqs[node_id]
    # (libbdsg / vg reverse_complement(char): A<->T, C<->G in either case, and every other character -- N, IUPAC codes -- 'N';
    # restated from the published sources of the pinned dependency, which is not in the reference's tree)
    return "".join(_COMP.get(c, "N") for c in reversed(s)) if is_reverse else s


def edit_is_match(e):
    return e["from_length"] == e["to_length"] and len(e["sequence"]) == 0


def edit_is_sub(e):
    return e["from_length"] == e["to_length"] and len(e["sequence"]) > 0


def edit_is_insertion(e):
    return e["from_length"] == 0 and e["to_length"] > 0 and len(e["sequence"]) > 0


def edit_is_deletion(e):
    return e["from_length"] > 0 and e["to_length"] == 0


def path_string(seqs, path):
    """vg::algorithms::path_string: the sequence the path spells -- graph bases under matches, the edit's own sequence otherwise."""
    out = []
    for m in path["mapping"]:
        pos = m["position"]
        ns = node_sequence(seqs, pos["node_id"], pos["is_reverse"])
        off = pos["offset"]
        for e in m["edit"]:
            if edit_is_match(e):
                out.append(ns[off:off + e["from_length"]])
            else:
                out.append(e["sequence"].decode())
            off += e["from_length"]
    return "".join(out)


def is_valid_dna(c):  # libgab
    return c in "ACGT"


def oplus_init(x, y):
    """libgab oplusInitnatl: log(exp(x) + exp(y)), with a running value of exactly 0 standing for "nothing yet"."""
    if x == 0:
        return y
    big, small = (x, y) if x > y else (y, x)
    return big + mp.log1p(mp.exp(small - big))


# ------------------------------------------------------------------------------------------------- vgan_utils.h:6-79
def substr(s, pos, n):
    if pos > len(s):
        raise Undefined("substr: pos %d beyond a string of %d" % (pos, len(s)))  # std::out_of_range inside noexcept
    return s[pos:pos + n]


def reconstruct_graph_sequence(seqs, path):
    graph_seq = ""
    mppg_sizes = []
    mppg_counter = 0  # :12, never incremented
    ps = path_string(seqs, path)  # :18
    mppgs = path["mapping"]
    f = 0
    for mppg in mppgs:
        pos = mppg["position"]
        node_seq = node_sequence(seqs, pos["node_id"], pos["is_reverse"])  # :24
        aligned_length = 0  # :26
        ed = mppg["edit"]
        edit_counter = 0  # :28, never incremented
        offset = pos["offset"]
        for edit in ed:
            to_length, from_length = edit["to_length"], edit["from_length"]
            softclip = ((mppg_counter == 0 and offset == 0 and edit_counter == 0 and from_length == 0 and to_length > 0 and edit_is_insertion(edit)) or
                        (mppg_counter == len(mppgs) - 1 and offset == 0 and edit_counter == len(ed) and from_length == 0 and to_length > 0 and
                         edit_is_insertion(edit)))  # :38-39
            if edit_is_match(edit) or edit_is_sub(edit):  # :41-47
                piece = substr(node_seq, offset, from_length)
                graph_seq += piece
                aligned_length = len(piece)
            elif edit_is_insertion(edit):  # :49-64
                graph_seq += ("S" if softclip else "-") * to_length
                aligned_length = to_length
            elif edit_is_deletion(edit):  # :66-70
                piece = substr(node_seq, offset, from_length)
                graph_seq += piece
                aligned_length = len(piece)
                if f > len(ps):
                    raise Undefined("insert: index %d beyond a string of %d" % (f, len(ps)))
                ps = ps[:f] + "-" * from_length + ps[f:]
            offset += from_length  # :71
            f += from_length
            mppg_sizes.append(aligned_length)  # :74: one entry per EDIT
    return graph_seq, ps, mppg_sizes


# ------------------------------------------------------------------------------------------ tables (miscfunc.h, haplocart_functions.cpp)
# Types are the reference's: what it computes in `double` is computed here in Python floats through libm (math.pow / math.log are
# the C library's pow / log the reference calls), what it keeps in `long double` in mpmath at 40 digits.  The difference shows:
# `1 - qscore_vec[Q]` is a double, and get_p_obs_base's `1 - epsilon` then gives back 1 - (1 - e), not e -- 1e-10 relative at Q = 60
# (the reference's bundled J2a1a1a1.gam holds such qualities; a simulator drawing Q <= 41 never sees it).
def get_p_seq_error(Q):  # miscfunc.h:180-188: const double, pow(10, (-1 * Q) * 0.1) in double
    return math.pow(10, ((-1 * Q) * 0.1)) if Q > 2 else 0.25


# miscfunc.h:199-212 (vector<double>): Q >= 2 takes get_p_seq_error (which answers 0.25 up to 2), the others 0.25
QSCORE_VEC = [get_p_seq_error(Q) if Q >= 2 else 0.25 for Q in range(100)]
INCORRECT_MAPPING_VEC = [math.pow(10, ((-1 * Q) * 0.1)) for Q in range(100)]  # haplocart_functions.cpp:101-107 (vector<double>)
BACKGROUND = {"A": mp.mpf(0.27532), "C": mp.mpf(0.30044), "G": mp.mpf(0.16644), "T": mp.mpf(0.25780)}  # :81-98 (doubles)


def get_background_freq(c):
    return BACKGROUND.get(c, mp.mpf(0.25))


def _in(lo, hi, x):
    return lo <= x <= hi


def get_p_obs_base(pangenome_base, epsilon, generations=8):
    """get_p_obs_base.cpp:38-69.  (22/23), (1/46), (2/3) and (1/3) are integer divisions in the reference: 0."""
    b = pangenome_base & 0xFFFFFFFF  # inRange takes unsigned
    if _in(57, 372, b):
        mu = 1.64273e-7
    elif _in(1, 56, b) or _in(373, 576, b):
        mu = 2.29640e-8
    elif _in(16384, 16569, b):
        mu = 1.54555e-8
    elif (_in(3307, 4262, b) or _in(4470, 5511, b) or _in(5904, 7445, b) or _in(7586, 8269, b) or _in(8366, 9990, b) or _in(10059, 10403, b) or
          _in(10470, 12137, b) or _in(12337, 14673, b) or _in(14747, 15886, b)):
        mu = 8.87640e-9 * (2 // 3) * 1.92596e-8 * (1 // 3)
    elif (_in(577, 647, b) or _in(1602, 1670, b) or _in(3230, 3304, b) or _in(4263, 4400, b) or _in(4402, 4469, b) or _in(5512, 5579, b) or
          _in(5587, 5654, b) or _in(5657, 5728, b) or _in(5761, 5891, b) or _in(7446, 7514, b) or _in(7518, 7585, b) or _in(8295, 8364, b) or
          _in(15888, 15953, b) or _in(15956, 16023, b)):
        mu = 6.91285e-9
    elif _in(648, 1601, b) or _in(1671, 3229, b):
        mu = 6.91285e-9
    else:
        mu = 2.48537e-8
    mu *= 30  # a double
    match = math.pow((1 - mu), generations)  # const double
    tv = (1 - math.pow((1 - mu), generations)) * (22 // 23)
    ts = (1 - math.pow((1 - mu), generations)) * (1 // 46)
    epsilon = float(epsilon)  # the parameter is `const double epsilon`
    return mp.mpf(match * (1 - epsilon) + (epsilon * (2 * tv + ts)))  # evaluated in double, then stored in a long double (:67)


# ------------------------------------------------------------------------------------- update_likelihood.cpp / process_mapping.cpp
def signed_char(b):
    return b - 256 if b >= 128 else b


def read_loglik(seqs, hc, aln, background_error_prob, use_background_error_prob, is_consensus_fasta):
    """Haplocart::update for one read: the vector over paths (mpf).  Raises Undefined where the reference does not define one."""
    n_paths = len(hc["path_names"])
    path = aln["path"]
    graph_full, algnseq, mppg_sizes = reconstruct_graph_sequence(seqs, path)
    quality = aln["quality"]
    mapq = aln["mapping_quality"]
    if mapq >= 100 or mapq < 0:
        raise Undefined("mapping quality %d indexes incorrect_mapping_vec out of range" % mapq)
    ll = [mp.mpf(0)] * n_paths
    position_in_read = 0
    use_bep = use_background_error_prob  # a by-value parameter of update_likelihood, sticky across the read's mappings (:42)
    for i, mppg in enumerate(path["mapping"]):
        if i >= len(mppg_sizes):
            raise Undefined("mapping %d has no entry in mppg_sizes (%d edits)" % (i, len(mppg_sizes)))
        graph_seq = substr(graph_full, position_in_read, mppg_sizes[i])  # :36
        read_seq = substr(algnseq, position_in_read, mppg_sizes[i])      # :37
        quality_scores = []
        for j in range(position_in_read, position_in_read + len(algnseq)):  # :40-44
            q = signed_char(quality[j]) if j < len(quality) else 0  # past the end: 0 (the build's definition)
            if q >= 90:
                use_bep = True
            quality_scores.append(q)
        position_in_read += len(read_seq)  # :45
        # ---- process_mapping (:46): mapping_seq is the WHOLE algnseq
        mapping_seq = algnseq
        node_id = mppg["position"]["node_id"]
        key = str(node_id)
        if key not in hc["pangenome_map"]:
            raise Undefined("node %d is not in parsed_pangenome_mapping (map::at throws)" % node_id)
        pangenome_base = hc["pangenome_map"][key]
        if not (0 <= pangenome_base < len(hc["mappabilities"])):
            raise Undefined("mappabilities[%d] is out of range" % pangenome_base)
        mappability = float(hc["mappabilities"][pangenome_base])  # const double
        p_correctly_mapped = mp.mpf((1 - INCORRECT_MAPPING_VEC[mapq]) * mappability)  # :41: a product of doubles, kept in a long double
        # get_p_no_seq_error_mapping (get_p_obs_base.cpp:3-27)
        p_no_seq_error = []
        for k in range(len(graph_seq)):
            if k >= len(mapping_seq):
                raise Undefined("mapping_seq[%d] beyond its end" % k)
            same = graph_seq[k] == mapping_seq[k]
            if use_bep:  # (the vector is of long double, the values pushed into it are doubles: `1 - background_error_prob` too)
                p_no_seq_error.append(float(background_error_prob) if same else 1 - float(background_error_prob))
            else:
                q = quality_scores[k] if k < len(quality_scores) else None
                if q is None or not (0 <= q < 100):
                    raise Undefined("qscore_vec[%r] is out of range" % (q,))
                p_no_seq_error.append(QSCORE_VEC[q] if same else 1 - QSCORE_VEC[q])  # doubles
        # supported paths (:57-80)
        log_lik_if_mapped = mp.mpf(0)
        for j in range(len(graph_seq)):
            g, r = graph_seq[j], mapping_seq[j]
            if g == "N" or r == "N":
                continue
            if not is_valid_dna(g) or not is_valid_dna(r):
                continue
            p_obs_base = get_p_obs_base(pangenome_base, p_no_seq_error[j], 8)
            if not is_consensus_fasta:
                x = (1 - p_correctly_mapped) * get_background_freq(r) + p_correctly_mapped * p_obs_base
            else:
                x = mp.mpf(1 - float(background_error_prob)) * p_obs_base  # (1 - background_error_prob) is a double
            log_lik_if_mapped += mp.log(x) if x > 0 else mp.mpf("-inf")
        # unsupported paths: get_log_lik_if_unsupported (:4-24) -- `counter % 4 == 4` never holds, every entry is a "mismatch"
        log_lik_if_unsupported = 0.0  # `double ret`, log of a double: all of it in double
        for Q in quality_scores:
            log_lik_if_unsupported += math.log(get_p_seq_error(Q))
        log_lik_if_unsupported = mp.mpf(log_lik_if_unsupported)
        row = node_id  # nodevector.at(node_id - minid)->pathsgo: path_supports row = node id (load.cpp:283-300)
        if not (0 <= row < len(hc["supports"])):
            raise Undefined("node %d has no path_supports row" % node_id)
        sup = hc["supports"][row]
        ll = [ll[p] + (log_lik_if_mapped if (p < len(sup) and sup[p]) else log_lik_if_unsupported) for p in range(n_paths)]
    return ll


def get_posterior(final_vec, hc, predicted):
    """get_posterior.cpp:87-127 with get_posterior_of_clade :51-76 and get_children :36-49."""
    path_names, parents, children = hc["path_names"], hc["parents"], hc["children"]

    def sum_ll(v):
        ret = v[0]
        for x in v[1:]:
            ret = oplus_init(ret, x)
        return ret

    def get_children(preds):
        out = set()
        for p in preds:
            out.update(children.get(p, []))  # (find()->second on a missing key is undefined: read as "no children")
        return out

    def clade(all_top, preds, depth=0):
        if depth > 10000:
            raise Undefined("children.txt does not end (cycle)")
        child_set = get_children(preds)
        for idx, name in enumerate(path_names):
            if name in child_set:
                all_top.append(final_vec[idx])
        if child_set:
            clade(all_top, child_set, depth + 1)
        return all_top
    total = sum_ll(final_vec)
    parent_vec = parents.get(predicted, [])
    clades = [predicted]
    conf = [mp.exp(sum_ll([final_vec[path_names.index(predicted)]]) - total)]
    for j, par in enumerate(parent_vec):
        differs = j == 0 or parent_vec[j] != parent_vec[j - 1]  # (j = 0 reads parent_vec[-1]: taken as "different")
        all_top = clade([], {par})
        if differs:
            clades.append(par)
            conf.append(mp.exp(sum_ll(all_top) - total) if all_top else mp.exp(mp.mpf(0) - total))  # (an empty sum: the build's 0)
    return clades, conf


def run(d, background_error_prob=0.0001, use_background_error_prob=False, is_consensus_fasta=False):
    seqs = load_gfa(os.path.join(d, "graph.gfa"))
    hc = load_hcfiles(d)
    alns = gamio.read_gam(os.path.join(d, "reads.gam"))
    n_paths = len(hc["path_names"])
    final_vec = [mp.mpf(0)] * n_paths
    undefined, used, per_read = [], 0, []
    for r, a in enumerate(alns):
        if a["identity"] < 1e-10:  # HaploCart.cpp:410
            continue
        try:
            ll = read_loglik(seqs, hc, a, background_error_prob, use_background_error_prob, is_consensus_fasta)
        except Undefined as e:
            undefined.append({"read": r, "why": str(e)})
            continue
        used += 1
        final_vec = [x + y for x, y in zip(final_vec, ll)]
        if len(per_read) < 12:
            per_read.append({"read": r, "loglik": [mp.nstr(x, 25) for x in ll]})
    best = max(range(n_paths), key=lambda p: (final_vec[p], -p))  # std::max_element: the first maximum
    clades, conf = get_posterior(final_vec, hc, hc["path_names"][best])
    return {"n_alignments": len(alns), "n_used": used, "undefined_reads": undefined,
            "params": {"background_error_prob": background_error_prob, "use_background_error_prob": use_background_error_prob,
                       "is_consensus_fasta": is_consensus_fasta},
            "final_vec": [mp.nstr(x, 25) for x in final_vec], "predicted": hc["path_names"][best],
            "posterior": [{"clade": c, "confidence": mp.nstr(v, 25)} for c, v in zip(clades, conf)],
            "first_reads": per_read}


# ---------------------------------------------------------------------------------------------------------- fixture writer
def make(d):
    """Inputs built by tools/pyref_inputs.py (plain seeded Python: no product code on this side at all), then edited so that the
    cases a plain simulator does not draw are there too: qualities >= 90 and >= 128, mapping quality 0, a short quality string.
    A second fixture beside it (<d>_j2): the reference's own bundled alignments (test/input_files/J2a1a1a1.gam: 81 giraffe
    alignments, 43.9 mappings per read) on a graph built to cover their node ids."""
    import pyref_inputs as pi
    os.makedirs(d, exist_ok=True)
    g = pi.variation_graph(seed=77, genome_len=700, n_paths=64)
    pi.write_hcfiles(d, g, mappability=[(0, 200, 1.0), (200, 260, 0.5), (260, 480, 1.0), (480, 500, 0.25), (500, g["genome_len"] + 2, 1.0)])
    alns = pi.simulate_reads(78, g, 240, read_len=120, indel_rate=0.15, softclip_rate=0.1, low_mapq_rate=0.3)
    for r, al in enumerate(alns):
        q = bytearray(al["quality"])
        if r % 23 == 5 and len(q) > 40:
            q[37] = 93  # the sticky switch, mid read
        if r % 31 == 7 and len(q) > 10:
            q[3] = 200  # a negative quality (signed char)
        if r % 41 == 11:
            q = q[:max(0, len(q) - 9)]  # a quality string shorter than the read
        al["quality"] = bytes(q)
        if r % 17 == 3:
            al["mapping_quality"] = 0
    open(os.path.join(d, "reads.gam"), "wb").write(gamio.write_gam(alns, group=64))
    out = {"_what": "tools/pyref_hc.py: an independent Python + mpmath (40 digits) restatement of HaploCart's likelihood path on the "
                    "inputs beside this file (tools/pyref_inputs.py: plain seeded Python); NOT generated by oracle/ or by the product",
           "default": run(d), "background": run(d, background_error_prob=0.02, use_background_error_prob=True)}
    json.dump(out, open(os.path.join(d, "hc_pyref.json"), "w"), indent=0)
    print("wrote", d, "used", out["default"]["n_used"], "undefined", len(out["default"]["undefined_reads"]))
    # ---- the reference's bundled alignments on a covering graph
    d2 = d.rstrip("/") + "_j2"
    os.makedirs(d2, exist_ok=True)
    src = os.path.join(ROOT, "tests", "golden", "alignments", "J2a1a1a1.gam")
    real = gamio.read_gam(src)
    g2 = pi.covering_graph(79, real, n_paths=24)
    pi.write_hcfiles(d2, g2, mappability=[(0, g2["genome_len"] // 3, 1.0), (g2["genome_len"] // 3, g2["genome_len"] // 2, 0.5),
                                          (g2["genome_len"] // 2, g2["genome_len"] + 2, 1.0)])
    open(os.path.join(d2, "reads.gam"), "wb").write(open(src, "rb").read())
    out2 = {"_what": "tools/pyref_hc.py on the reference's test/input_files/J2a1a1a1.gam (copied beside this file) and a graph covering its "
                     "node ids (tools/pyref_inputs.py covering_graph)",
            "default": run(d2), "background": run(d2, background_error_prob=0.02, use_background_error_prob=True)}
    json.dump(out2, open(os.path.join(d2, "hc_pyref.json"), "w"), indent=0)
    print("wrote", d2, "used", out2["default"]["n_used"], "undefined", len(out2["default"]["undefined_reads"]))


def segment_sums(seqs, hc, aln, background_error_prob, use_background_error_prob, is_consensus_fasta):
    """read_loglik's two numbers per mapping -- (node id, log_lik_if_mapped, log_lik_if_unsupported) -- without the loop over the
    paths: the SAME statements as read_loglik above, in the same order (kept side by side on purpose; run_full() holds the two
    against each other on a subset of the reads, path by path)."""
    path = aln["path"]
    graph_full, algnseq, mppg_sizes = reconstruct_graph_sequence(seqs, path)
    quality = aln["quality"]
    mapq = aln["mapping_quality"]
    if mapq >= 100 or mapq < 0:
        raise Undefined("mapping quality %d indexes incorrect_mapping_vec out of range" % mapq)
    out = []
    position_in_read = 0
    use_bep = use_background_error_prob
    for i, mppg in enumerate(path["mapping"]):
        if i >= len(mppg_sizes):
            raise Undefined("mapping %d has no entry in mppg_sizes (%d edits)" % (i, len(mppg_sizes)))
        graph_seq = substr(graph_full, position_in_read, mppg_sizes[i])
        read_seq = substr(algnseq, position_in_read, mppg_sizes[i])
        quality_scores = []
        for j in range(position_in_read, position_in_read + len(algnseq)):
            q = signed_char(quality[j]) if j < len(quality) else 0
            if q >= 90:
                use_bep = True
            quality_scores.append(q)
        position_in_read += len(read_seq)
        mapping_seq = algnseq
        node_id = mppg["position"]["node_id"]
        key = str(node_id)
        if key not in hc["pangenome_map"]:
            raise Undefined("node %d is not in parsed_pangenome_mapping (map::at throws)" % node_id)
        pangenome_base = hc["pangenome_map"][key]
        if not (0 <= pangenome_base < len(hc["mappabilities"])):
            raise Undefined("mappabilities[%d] is out of range" % pangenome_base)
        mappability = float(hc["mappabilities"][pangenome_base])
        p_correctly_mapped = mp.mpf((1 - INCORRECT_MAPPING_VEC[mapq]) * mappability)
        p_no_seq_error = []
        for k in range(len(graph_seq)):
            if k >= len(mapping_seq):
                raise Undefined("mapping_seq[%d] beyond its end" % k)
            same = graph_seq[k] == mapping_seq[k]
            if use_bep:
                p_no_seq_error.append(float(background_error_prob) if same else 1 - float(background_error_prob))
            else:
                q = quality_scores[k] if k < len(quality_scores) else None
                if q is None or not (0 <= q < 100):
                    raise Undefined("qscore_vec[%r] is out of range" % (q,))
                p_no_seq_error.append(QSCORE_VEC[q] if same else 1 - QSCORE_VEC[q])
        log_lik_if_mapped = mp.mpf(0)
        for j in range(len(graph_seq)):
            g, r = graph_seq[j], mapping_seq[j]
            if g == "N" or r == "N":
                continue
            if not is_valid_dna(g) or not is_valid_dna(r):
                continue
            p_obs_base = get_p_obs_base(pangenome_base, p_no_seq_error[j], 8)
            if not is_consensus_fasta:
                x = (1 - p_correctly_mapped) * get_background_freq(r) + p_correctly_mapped * p_obs_base
            else:
                x = mp.mpf(1 - float(background_error_prob)) * p_obs_base
            log_lik_if_mapped += mp.log(x) if x > 0 else mp.mpf("-inf")
        log_lik_if_unsupported = 0.0
        for Q in quality_scores:
            log_lik_if_unsupported += math.log(get_p_seq_error(Q))
        if not (0 <= node_id < hc["n_support_rows"]):
            raise Undefined("node %d has no path_supports row" % node_id)
        out.append((node_id, log_lik_if_mapped, mp.mpf(log_lik_if_unsupported)))
    return out


def run_full(d, background_error_prob=0.0001, use_background_error_prob=False, is_consensus_fasta=False, literal_reads=200, sample_paths=64, vector_reads=50):
    """run() at the reference's real shape (thousands of paths): the per-mapping sums in mpmath as everywhere in this file, the sum
    over a read's mappings THROUGH the path_supports rows in numpy long double (x87: 64-bit mantissa) over all paths at once -- the
    literal statement `ll[p] += supported ? mapped : unsupported` for every p -- and, for the first `literal_reads` usable reads,
    read_loglik's own loop over the paths in mpmath beside it, which must give the same vector."""
    import numpy as np
    seqs = load_gfa(os.path.join(d, "graph.gfa"))
    hc = load_hcfiles(d, supports_as_numpy=True)
    alns = gamio.read_gam(os.path.join(d, "reads.gam"))
    sup = hc["supports_np"]
    n_paths = len(hc["path_names"])
    final = np.zeros(n_paths, np.longdouble)
    undefined, used, per_read = [], 0, []
    pick = sorted(random_sample(n_paths, sample_paths))
    for r, a in enumerate(alns):
        if a["identity"] < 1e-10:
            continue
        try:
            segs = segment_sums(seqs, hc, a, background_error_prob, use_background_error_prob, is_consensus_fasta)
        except Undefined as e:
            undefined.append({"read": r, "why": str(e)})
            continue
        ll = np.zeros(n_paths, np.longdouble)
        for node, m_, u_ in segs:
            row = sup[node, :n_paths]
            ll += np.where(row, np.longdouble(mp.nstr(m_, 30)), np.longdouble(mp.nstr(u_, 30)))
        if used < literal_reads:  # the loop over the paths as the reference has it, in mpmath, on the plain lists
            lit = read_loglik(seqs, hc, a, background_error_prob, use_background_error_prob, is_consensus_fasta)
            worst = max(abs(mp.mpf(str(ll[p_])) - lit[p_]) / abs(lit[p_]) for p_ in range(n_paths) if lit[p_] != 0)
            assert worst < mp.mpf("1e-17"), (r, worst)
            if len(per_read) < vector_reads:
                per_read.append({"read": r, "paths": pick, "loglik": [mp.nstr(lit[p_], 25) for p_ in pick]})
        used += 1
        final += ll
    fv = [mp.mpf(np.format_float_positional(x, unique=True, trim="-")) for x in final]  # (long double -> mpmath, every digit)
    best = max(range(n_paths), key=lambda p_: (fv[p_], -p_))
    clades, conf = get_posterior(fv, hc, hc["path_names"][best])
    return {"n_alignments": len(alns), "n_used": used, "undefined_reads": undefined,
            "params": {"background_error_prob": background_error_prob, "use_background_error_prob": use_background_error_prob,
                       "is_consensus_fasta": is_consensus_fasta},
            "final_vec": [mp.nstr(x, 22) for x in fv], "predicted": hc["path_names"][best],
            "posterior": [{"clade": c, "confidence": mp.nstr(v, 22)} for c, v in zip(clades, conf)],
            "first_reads": per_read, "literal_reads_checked": min(used, literal_reads)}


def make_bundled(base):
    """The reference's other bundled alignments (test/input_files/{two_unique, all_the_same, all_the_same_reverse}.gam, copied under
    tests/golden/alignments/ as data) on graphs built to cover their node ids, like <base>_j2 for J2a1a1a1.gam."""
    import pyref_inputs as pi
    for k, name in enumerate(("two_unique", "all_the_same", "all_the_same_reverse")):
        d = "%s_%s" % (base.rstrip("/"), name)
        os.makedirs(d, exist_ok=True)
        src = os.path.join(ROOT, "tests", "golden", "alignments", name + ".gam")
        real = gamio.read_gam(src)
        g = pi.covering_graph(83 + k, real, n_paths=24)
        L = g["genome_len"]
        pi.write_hcfiles(d, g, mappability=[(0, L // 4, 1.0), (L // 4, L // 3, 0.75), (L // 3, L + 2, 1.0)], gfa_paths=False)
        raw = open(os.path.join(d, "path_supports"), "rb").read()
        with gzip.GzipFile(os.path.join(d, "path_supports.gz"), "wb", mtime=0) as f:
            f.write(raw)
        os.remove(os.path.join(d, "path_supports"))
        open(os.path.join(d, "reads.gam"), "wb").write(open(src, "rb").read())
        out = {"_what": "tools/pyref_hc.py --make-bundled on the reference's test/input_files/%s.gam (copied beside this file) and a graph "
                        "covering its node ids (tools/pyref_inputs.py covering_graph)" % name,
               "default": run(d), "background": run(d, background_error_prob=0.02, use_background_error_prob=True)}
        json.dump(out, open(os.path.join(d, "hc_pyref.json"), "w"), indent=0)
        print("wrote", d, "alignments", out["default"]["n_alignments"], "used", out["default"]["n_used"], "undefined", len(out["default"]["undefined_reads"]))


def random_sample(n, k):
    import random
    return random.Random(4242).sample(range(n), min(k, n))


def make_full(d):
    """The fixture at the reference's shape: 11 820 nodes, 5 179 paths (81 mask words), 16 569 reference bases; 1 200 reads of
    ~150 bases (tools/pyref_inputs.py: plain seeded Python).  The path sidecars are gzipped (the loaders read them so)."""
    import pyref_inputs as pi
    os.makedirs(d, exist_ok=True)
    g = pi.variation_graph(seed=179, genome_len=16569, n_paths=5179, site_rate=0.425)
    L = g["genome_len"]
    pi.write_hcfiles(d, g, mappability=[(0, 3000, 1.0), (3000, 3050, 0.5), (3050, 9000, 1.0), (9000, 9100, 0.8125), (9100, 14000, 1.0),
                                        (14000, 14040, 0.25), (14040, L + 2, 1.0)], gfa_paths=False)
    for name in ("path_supports", "parents.txt", "children.txt"):
        raw = open(os.path.join(d, name), "rb").read()
        with gzip.GzipFile(os.path.join(d, name + ".gz"), "wb", mtime=0) as f:
            f.write(raw)
        os.remove(os.path.join(d, name))
    alns = pi.simulate_reads(180, g, 1200, read_len=150, indel_rate=0.04, softclip_rate=0.04, low_mapq_rate=0.1)
    for r, al in enumerate(alns):
        q = bytearray(al["quality"])
        if r % 97 == 5 and len(q) > 60:
            q[57] = 93
        if r % 131 == 7 and len(q) > 10:
            q[3] = 200
        if r % 151 == 11:
            q = q[:max(0, len(q) - 9)]
        al["quality"] = bytes(q)
        if r % 89 == 3:
            al["mapping_quality"] = 0
    open(os.path.join(d, "reads.gam"), "wb").write(gamio.write_gam(alns, group=256))
    out = {"_what": "tools/pyref_hc.py --make-full: the independent Python + mpmath restatement at the reference's shape (5 179 paths, "
                    "11 820 nodes, ~150-base reads); inputs by tools/pyref_inputs.py; NOT generated by oracle/ or by the product",
           "default": run_full(d), "background": run_full(d, background_error_prob=0.02, use_background_error_prob=True, literal_reads=10, vector_reads=8)}
    json.dump(out, open(os.path.join(d, "hc_pyref.json"), "w"), indent=0)
    print("wrote", d, "used", out["default"]["n_used"], "undefined", len(out["default"]["undefined_reads"]))


def check_kats():
    d = os.path.join(ROOT, "tests", "golden", "reconstruct")
    seqs = load_gfa(os.path.join(d, "target_graph.gfa"))
    alns = gamio.read_gam(os.path.join(d, "test_reads.gam"))
    exp = json.load(open(os.path.join(d, "expected.json")))["cases"]
    for c in exp:
        gs, ps, sizes = reconstruct_graph_sequence(seqs, alns[c["read"]]["path"])
        assert gs == c["graph_seq"] and ps == c["read_seq"], (c["name"], gs, ps)
        if "mppg_sizes" in c:
            assert sizes == c["mppg_sizes"], (c["name"], sizes)
    print("reconstruction: %d of the reference's known-answer cases reproduced" % len(exp))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--make")
    ap.add_argument("--make-full", help="the fixture at the reference's shape (5 179 paths): tests/golden/hc_pyref_full")
    ap.add_argument("--make-bundled", help="fixtures of the reference's other three bundled GAMs: <arg>_two_unique, _all_the_same, _all_the_same_reverse")
    ap.add_argument("--run")
    ap.add_argument("--out")
    ap.add_argument("--check-kats", action="store_true")
    args = ap.parse_args()
    if args.check_kats:
        check_kats()
    if args.make:
        make(args.make)
    if args.make_full:
        make_full(args.make_full)
    if args.make_bundled:
        make_bundled(args.make_bundled)
    if args.run:
        res = run(args.run)
        txt = json.dumps(res, indent=0)
        open(args.out, "w").write(txt) if args.out else print(txt[:2000])
