"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class OrcGraph(C.Structure):
    _fields_ = [("min_id", C.c_int64), ("max_id", C.c_int64),
                ("node_seq_off", C.c_void_p), ("node_seq", C.c_void_p),
                ("n_paths", C.c_int32),
                ("pathsgo", C.c_void_p), ("pangenome_base", C.c_void_p),
                ("mappability", C.c_void_p), ("n_mappability", C.c_int64)]


class OrcAlnSet(C.Structure):
    _fields_ = [("n_reads", C.c_int64),
                ("seq_off", C.c_void_p), ("seq", C.c_void_p),
                ("qual_off", C.c_void_p), ("qual", C.c_void_p),
                ("mapq", C.c_void_p), ("identity", C.c_void_p),
                ("map_off", C.c_void_p), ("m_node", C.c_void_p), ("m_offset", C.c_void_p), ("m_rev", C.c_void_p),
                ("edit_off", C.c_void_p), ("e_from", C.c_void_p), ("e_to", C.c_void_p),
                ("e_seq_off", C.c_void_p), ("e_seq", C.c_void_p)]


class OrcHcParams(C.Structure):
    _fields_ = [("background_error_prob", C.c_double),
                ("use_background_error_prob", C.c_int32),
                ("is_consensus_fasta", C.c_int32)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".h"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            build()
        _lib = C.CDLL(so)
        _lib.orc_p_seq_error.restype = C.c_double
        _lib.orc_qscore.restype = C.c_double
        _lib.orc_p_incorrect_mapping.restype = C.c_double
        _lib.orc_background_freq.restype = C.c_double
        _lib.orc_background_freq.argtypes = [C.c_char]
        _lib.orc_oplusnatl.restype = C.c_longdouble
        _lib.orc_oplusnatl.argtypes = [C.c_longdouble, C.c_longdouble]
        _lib.orc_oplusInitnatl.restype = C.c_longdouble
        _lib.orc_oplusInitnatl.argtypes = [C.c_longdouble, C.c_longdouble]
        _lib.orc_p_obs_base.restype = C.c_longdouble
        _lib.orc_p_obs_base.argtypes = [C.c_int, C.c_double, C.c_int]
        for f in ("orc_load_mappabilities", "orc_load_pangenome_map", "orc_load_path_supports"):
            getattr(_lib, f).restype = C.c_int64
    return _lib


def _p(arr):
    return arr.ctypes.data_as(C.c_void_p)


class Graph:
    """Graph view for the oracle. node_seqs: dict id -> forward sequence (bytes)."""

    def __init__(self, node_seqs, n_paths, pathsgo, pangenome_base, mappability):
        ids = sorted(node_seqs)
        self.min_id, self.max_id = ids[0], ids[-1]
        offs = [0]
        blob = bytearray()
        for i in range(self.min_id, self.max_id + 1):
            blob += node_seqs.get(i, b"")
            offs.append(len(blob))
        self.node_seq_off = np.asarray(offs, dtype=np.int64)
        self.node_seq = np.frombuffer(bytes(blob) + b"\0", dtype=np.uint8).copy()
        self.n_paths = int(n_paths)
        self.pathsgo = np.ascontiguousarray(pathsgo, dtype=np.uint8)
        assert self.pathsgo.shape == (self.max_id + 1, self.n_paths)
        self.pangenome_base = np.ascontiguousarray(pangenome_base, dtype=np.int32)
        assert self.pangenome_base.shape == (self.max_id + 1,)
        self.mappability = np.ascontiguousarray(mappability, dtype=np.float64)
        self.node_seqs = dict(node_seqs)
        self.c = OrcGraph(self.min_id, self.max_id, _p(self.node_seq_off), _p(self.node_seq), self.n_paths,
                          _p(self.pathsgo), _p(self.pangenome_base), _p(self.mappability), len(self.mappability))


class AlnSet:
    """Flat arrays from a list of gamio-style alignment dicts."""

    def __init__(self, alns):
        seq_off, qual_off, map_off, edit_off, e_seq_off = [0], [0], [0], [0], [0]
        seq, qual, e_seq = bytearray(), bytearray(), bytearray()
        mapq, ident, m_node, m_offset, m_rev, e_from, e_to = [], [], [], [], [], [], []
        for a in alns:
            seq += a["sequence"]
            seq_off.append(len(seq))
            qual += a["quality"]
            qual_off.append(len(qual))
            mapq.append(a["mapping_quality"])
            ident.append(a["identity"])
            for m in a["path"]["mapping"]:
                m_node.append(m["position"]["node_id"])
                m_offset.append(m["position"]["offset"])
                m_rev.append(1 if m["position"]["is_reverse"] else 0)
                for e in m["edit"]:
                    e_from.append(e["from_length"])
                    e_to.append(e["to_length"])
                    e_seq += e["sequence"]
                    e_seq_off.append(len(e_seq))
                edit_off.append(len(e_from))
            map_off.append(len(m_node))
        self.n_reads = len(alns)
        self.seq_off = np.asarray(seq_off, np.int64)
        self.seq = np.frombuffer(bytes(seq) + b"\0", np.uint8).copy()
        self.qual_off = np.asarray(qual_off, np.int64)
        self.qual = np.frombuffer(bytes(qual) + b"\0", np.uint8).copy()
        self.mapq = np.asarray(mapq, np.int32)
        self.identity = np.asarray(ident, np.float64)
        self.map_off = np.asarray(map_off, np.int64)
        self.m_node = np.asarray(m_node, np.int64)
        self.m_offset = np.asarray(m_offset, np.int64)
        self.m_rev = np.asarray(m_rev, np.uint8)
        self.edit_off = np.asarray(edit_off, np.int64)
        self.e_from = np.asarray(e_from, np.int32)
        self.e_to = np.asarray(e_to, np.int32)
        self.e_seq_off = np.asarray(e_seq_off, np.int64)
        self.e_seq = np.frombuffer(bytes(e_seq) + b"\0", np.uint8).copy()
        self.c = OrcAlnSet(self.n_reads, _p(self.seq_off), _p(self.seq), _p(self.qual_off), _p(self.qual),
                           _p(self.mapq), _p(self.identity), _p(self.map_off), _p(self.m_node), _p(self.m_offset),
                           _p(self.m_rev), _p(self.edit_off), _p(self.e_from), _p(self.e_to), _p(self.e_seq_off),
                           _p(self.e_seq))

    @classmethod
    def from_arrays(cls, **kw):
        """Build from the product's flat alignment-set arrays (same field names)."""
        self = cls.__new__(cls)
        self.n_reads = int(kw["n_reads"])
        for k in ("seq_off", "qual_off", "map_off", "m_node", "m_offset", "edit_off", "e_seq_off"):
            setattr(self, k, np.ascontiguousarray(kw[k], np.int64))
        for k in ("seq", "qual", "m_rev", "e_seq"):
            v = np.ascontiguousarray(kw[k], np.uint8)
            setattr(self, k, np.concatenate([v, np.zeros(1, np.uint8)]))
        for k in ("mapq", "e_from", "e_to"):
            setattr(self, k, np.ascontiguousarray(kw[k], np.int32))
        self.identity = np.ascontiguousarray(kw["identity"], np.float64)
        self.c = OrcAlnSet(self.n_reads, _p(self.seq_off), _p(self.seq), _p(self.qual_off), _p(self.qual),
                           _p(self.mapq), _p(self.identity), _p(self.map_off), _p(self.m_node), _p(self.m_offset),
                           _p(self.m_rev), _p(self.edit_off), _p(self.e_from), _p(self.e_to), _p(self.e_seq_off),
                           _p(self.e_seq))
        return self


def hc_params(background_error_prob=0.0001, use_background_error_prob=False, is_consensus_fasta=False):
    return OrcHcParams(background_error_prob, int(use_background_error_prob), int(is_consensus_fasta))


def reconstruct(g, a, r, cap=4096):
    gs = C.create_string_buffer(cap)
    rs = C.create_string_buffer(cap)
    sizes = np.zeros(cap, np.int32)
    lens = np.zeros(3, np.int64)
    rc = lib().orc_reconstruct(C.byref(g.c), C.byref(a.c), C.c_int64(r), gs, rs, _p(sizes), C.c_int64(cap), _p(lens))
    if rc != 0:
        return rc, None, None, None
    return 0, gs.raw[:lens[0]], rs.raw[:lens[1]], sizes[:lens[2]].tolist()


def hc_read(g, a, r, params=None):
    params = params or hc_params()
    out = np.zeros(g.n_paths, np.longdouble)
    flags = C.c_int32(0)
    rc = lib().orc_hc_read(C.byref(g.c), C.byref(a.c), C.c_int64(r), C.byref(params), _p(out), C.byref(flags))
    return rc, out, flags.value


def hc_read_segments(g, a, r, params=None, cap=4096):
    params = params or hc_params()
    S = np.zeros(cap)
    U = np.zeros(cap)
    node = np.zeros(cap, np.int64)
    n = C.c_int64(0)
    rc = lib().orc_hc_read_segments(C.byref(g.c), C.byref(a.c), C.c_int64(r), C.byref(params), _p(S), _p(U), _p(node),
                                    C.c_int64(cap), C.byref(n))
    return rc, S[:n.value].copy(), U[:n.value].copy(), node[:n.value].copy()


def hc_run(g, a, params=None, r0=0, r1=None, n_threads=1, faithful=True):
    params = params or hc_params()
    r1 = a.n_reads if r1 is None else r1
    fld = np.zeros(g.n_paths, np.longdouble)
    fd = np.zeros(g.n_paths, np.float64)
    bad = C.c_int64(0)
    rc = lib().orc_hc_run(C.byref(g.c), C.byref(a.c), C.c_int64(r0), C.c_int64(r1), C.byref(params),
                          C.c_int(n_threads), C.c_int(int(faithful)), _p(fld), _p(fd), C.byref(bad))
    assert rc == 0
    return fld, fd, bad.value


def hc_posterior(final_vec, path_names, parents_txt, children_txt, predicted):
    fv = np.ascontiguousarray(final_vec, np.longdouble)
    out = C.create_string_buffer(1 << 20)
    conf = np.zeros(4096)
    n = lib().orc_hc_posterior(_p(fv), C.c_int32(len(fv)), "\n".join(path_names).encode() + b"\n",
                               parents_txt.encode(), children_txt.encode(), predicted.encode(), out,
                               C.c_int64(1 << 20), _p(conf), C.c_int32(4096))
    if n < 0:
        raise RuntimeError("orc_hc_posterior rc=%d" % n)
    recs = [ln.split("\t") for ln in out.value.decode().splitlines()]
    return [(r[0], float(r[1]), int(r[2])) for r in recs]


# ---------------------------------------------------------------- GFA (test-side reader)
def read_gfa(path):
    node_seqs, paths = {}, []
    with open(path) as f:
        for line in f:
            t = line.rstrip("\n").split("\t")
            if t[0] == "S":
                node_seqs[int(t[1])] = t[2].encode()
            elif t[0] == "P":
                steps = [(int(s[:-1]), s[-1] == "-") for s in t[2].split(",")]
                paths.append((t[1], steps))
    return node_seqs, paths


def graph_from_gfa(path, pangenome_base=None, mappability=None):
    node_seqs, paths = read_gfa(path)
    max_id = max(node_seqs)
    P = len(paths)
    pathsgo = np.zeros((max_id + 1, P), np.uint8)
    for j, (_, steps) in enumerate(paths):
        for nid, _rev in steps:
            pathsgo[nid, j] = 1
    if pangenome_base is None:
        # coordinate = 1 + running offset of the node in id order (a stand-in for parsed_pangenome_mapping)
        pangenome_base = np.full(max_id + 1, -1, np.int32)
        pos = 0
        for i in sorted(node_seqs):
            pangenome_base[i] = pos + 1
            pos += len(node_seqs[i])
    if mappability is None:
        mappability = np.ones(int(max(pangenome_base)) + 2)
    return Graph(node_seqs, P, pathsgo, pangenome_base, mappability), [p[0] for p in paths]


# ---------------------------------------------------------------- euka
class OrcEukaDb(C.Structure):
    _fields_ = [("n_clades", C.c_int32), ("clade_dist", C.c_void_p), ("bin_off", C.c_void_p), ("bin_lo", C.c_void_p),
                ("bin_hi", C.c_void_p), ("bin_entropy", C.c_void_p)]


class OrcEukaParams(C.Structure):
    _fields_ = [("MINIMUMMQ", C.c_uint32), ("lengthToProf", C.c_int32)]


class OrcEukaOut(C.Structure):
    _fields_ = [("read_clade", C.c_void_p), ("read_in", C.c_void_p), ("read_out", C.c_void_p), ("read_like", C.c_void_p),
                ("read_not_like", C.c_void_p), ("read_pass", C.c_void_p), ("clade_count", C.c_void_p),
                ("baseshift", C.c_void_p), ("bin_cov", C.c_void_p), ("n_bad", C.c_int64)]


class EukaDb:
    def __init__(self, clade_dist, bin_off, bin_lo, bin_hi, bin_entropy=None):
        self.clade_dist = np.ascontiguousarray(clade_dist, np.float64)
        self.bin_off = np.ascontiguousarray(bin_off, np.int32)
        self.bin_lo = np.ascontiguousarray(bin_lo, np.int32)
        self.bin_hi = np.ascontiguousarray(bin_hi, np.int32)
        self.bin_entropy = np.ascontiguousarray(bin_entropy if bin_entropy is not None else np.zeros(len(self.bin_lo)), np.float64)
        self.n_clades = len(self.clade_dist)
        self.c = OrcEukaDb(self.n_clades, _p(self.clade_dist), _p(self.bin_off), _p(self.bin_lo), _p(self.bin_hi),
                           _p(self.bin_entropy))


class OrcDamage:
    def __init__(self, prof5="", prof3=""):
        L = lib()
        L.orc_damage_create.restype = C.c_void_p
        self.h = L.orc_damage_create(prof5.encode(), prof3.encode())
        if not self.h:
            raise ValueError("malformed damage profile")

    def matrix(self, L_, l):
        out = np.zeros(16)
        rc = lib().orc_damage_matrix(C.c_void_p(self.h), C.c_uint32(L_), C.c_uint32(l), _p(out))
        assert rc == 0
        return out.reshape(4, 4)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_damage_free(C.c_void_p(self.h))
            self.h = None


def euka_run(g, a, db, dmg, min_mapq=29, length_to_prof=5):
    R, Cn, nb = a.n_reads, db.n_clades, len(db.bin_lo)
    o = {"clade": np.zeros(R, np.int32), "in_lik": np.zeros(R), "out_lik": np.zeros(R), "like": np.zeros(R),
         "not_like": np.zeros(R), "pass": np.zeros(R, np.uint8), "clade_count": np.zeros(Cn, np.int32),
         "baseshift": np.zeros((Cn, 2 * length_to_prof, 16), np.uint32), "bin_cov": np.zeros(max(nb, 1))}
    oc = OrcEukaOut(_p(o["clade"]), _p(o["in_lik"]), _p(o["out_lik"]), _p(o["like"]), _p(o["not_like"]), _p(o["pass"]),
                    _p(o["clade_count"]), _p(o["baseshift"]), _p(o["bin_cov"]), 0)
    prm = OrcEukaParams(min_mapq, length_to_prof)
    rc = lib().orc_euka_run(C.byref(g.c), C.byref(a.c), C.byref(db.c), C.c_void_p(dmg.h), C.byref(prm), C.byref(oc))
    assert rc == 0
    o["n_bad"] = oc.n_bad
    o["bin_cov"] = o["bin_cov"][:nb]
    return o


class OrcEukaReportCfg(C.Structure):
    _fields_ = [("MINNUMOFBINS", C.c_uint32), ("MINNUMOFREADS", C.c_uint32), ("MAXIMUMOFBINS", C.c_int32),
                ("ENTROPY_SCORE_THRESHOLD", C.c_double), ("lengthToProf", C.c_int32), ("run_mcmc", C.c_int32),
                ("iter", C.c_int32), ("burnin", C.c_int32), ("seed", C.c_uint64), ("outFrag", C.c_int32),
                ("outGroup", C.c_char_p), ("out_dir", C.c_char_p)]


def euka_report(db, clade_id, clade_names, o, read_seq_len, prefix, names=None, min_bins=6, min_reads=10, max_zero_bins=0,
                entropy=1.17, length_to_prof=5, run_mcmc=True, iters=10000, burnin=100, seed=1, out_frag=False,
                out_group=None, out_dir=None):
    """Euka::run after readGAM3 (oracle/euka_abundance_oracle.cpp) on an euka_run() result `o`; writes <prefix>_* files and
    returns (detected clade ids, estimates[n, 5])."""
    R = len(o["clade"])
    oc = OrcEukaOut(_p(o["clade"]), _p(o["in_lik"]), _p(o["out_lik"]), _p(o["like"]), _p(o["not_like"]), _p(o["pass"]),
                    _p(o["clade_count"]), _p(np.ascontiguousarray(o["baseshift"])), _p(np.ascontiguousarray(o["bin_cov"])), 0)
    cid = np.ascontiguousarray(clade_id, np.int32)
    sl = np.ascontiguousarray(read_seq_len, np.int32)
    blob, off = None, None
    if names is not None:
        off = np.zeros(R + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in names])
        blob = b"".join(names)
    cfg = OrcEukaReportCfg(min_bins, min_reads, max_zero_bins, entropy, length_to_prof, int(run_mcmc), iters, burnin, seed,
                           int(out_frag), out_group.encode() if out_group else None, out_dir.encode() if out_dir else None)
    det = np.zeros(db.n_clades + 1, np.int32)
    est = np.zeros((db.n_clades + 1, 5))
    nd = C.c_int32(0)
    rc = lib().orc_euka_report(C.byref(db.c), _p(cid), "\n".join(clade_names).encode(), C.byref(oc), C.c_int64(R), _p(sl), blob,
                               _p(off) if off is not None else None, C.byref(cfg), prefix.encode(), _p(det), C.byref(nd), _p(est))
    if rc != 0:
        raise ValueError("orc_euka_report failed")
    return det[:nd.value].copy(), est[:nd.value].copy()


# ---------------------------------------------------------------- soibean
class SbOracle:
    """analyse_GAM result (pathMap / detailMap per read) held by the oracle."""

    def __init__(self, g, a, dmg, penalty=7, path_findable=None):
        L = lib()
        L.orc_sb_analyse.restype = C.c_void_p
        self.P = g.n_paths
        pf = np.ones(self.P, np.uint8) if path_findable is None else np.ascontiguousarray(path_findable, np.uint8)
        bad = C.c_int64(0)
        self.h = L.orc_sb_analyse(C.byref(g.c), C.byref(a.c), _p(pf), C.c_void_p(dmg.h), C.c_int(penalty), C.byref(bad))
        self.n_bad = bad.value
        self._keep = (g, a, dmg, pf)

    def ok(self, r):
        return bool(lib().orc_sb_read_ok(C.c_void_p(self.h), C.c_int64(r)))

    def pathmap(self, r):
        out = np.zeros(self.P)
        rc = lib().orc_sb_pathmap(C.c_void_p(self.h), C.c_int64(r), _p(out))
        return out if rc == 0 else None

    def counts(self, r, p):
        out = np.zeros(25, np.uint32)
        n = C.c_uint32(0)
        rc = lib().orc_sb_counts(C.c_void_p(self.h), C.c_int64(r), C.c_int32(p), _p(out), C.byref(n))
        return (out, n.value) if rc == 0 else (None, 0)

    def best_paths(self, n_reads):
        best = np.zeros(n_reads, np.int32)
        sig = np.zeros(self.P, np.int64)
        L = lib()
        L.orc_sb_best_paths.restype = C.c_int64
        n = L.orc_sb_best_paths(C.c_void_p(self.h), _p(best), _p(sig))
        return best, sig, int(n)

    def mixture_loglike(self, paths, log_freq):
        p = np.ascontiguousarray(paths, np.int32)
        L = lib()
        L.orc_sb_mixture_loglike.restype = C.c_double
        return L.orc_sb_mixture_loglike(C.c_void_p(self.h), C.c_int32(len(p)), _p(p), C.c_double(log_freq))

    def estimate(self, newick, path_names, sig_nodes, prefix, freqs7, con=0.01, iters=2000, burnin=200, chains=2, seed=1, run_mcmc=True):
        """oracle/sb_chain_oracle.cpp: the chain loop of soibean.cpp:738-944 on this handle's reads."""
        class Cfg(C.Structure):
            _fields_ = [("max_iter", C.c_uint32), ("burn", C.c_uint32), ("chains", C.c_uint32), ("seed", C.c_uint64), ("con", C.c_double),
                        ("freqs7", C.c_double * 7), ("run_mcmc", C.c_int32)]
        cfg = Cfg(iters, burnin, chains, seed, con, (C.c_double * 7)(*freqs7), int(run_mcmc))
        sg = np.ascontiguousarray(sig_nodes, np.int32)
        rc = lib().orc_sb_estimate(C.c_void_p(self.h), newick.encode(), ("\n".join(path_names) + "\n").encode(), _p(sg), C.c_int32(len(sg)),
                                   C.byref(cfg), prefix.encode())
        if rc != 0:
            raise ValueError("orc_sb_estimate failed")

    def loglike(self, sources, con, freqs7, n_threads=8):
        k = len(sources)
        child = np.array([s[0] for s in sources], np.int32)
        parent = np.array([s[1] for s in sources], np.int32)
        dist = np.array([s[2] for s in sources], np.float64)
        pos = np.array([s[3] for s in sources], np.float64)
        theta = np.array([s[4] for s in sources], np.float64)
        f = np.ascontiguousarray(freqs7, np.float64)
        out = C.c_double(0)
        rc = lib().orc_sb_loglike(C.c_void_p(self.h), C.c_int32(k), _p(child), _p(parent), _p(dist), _p(pos), _p(theta),
                                  C.c_double(con), _p(f), C.c_int(n_threads), C.byref(out))
        return rc, out.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_sb_free(C.c_void_p(self.h))
            self.h = None
