"""Shared helpers for the test-suite: bridge product objects (vgan_amd) to the oracle's views (orc)."""
import numpy as np

import orc


def orc_graph_from_product(g):
    """Oracle graph view of a product Graph (same arrays, reference-style bool-per-byte pathsgo)."""
    off = g.node_seq_off
    seq = g.node_seq.tobytes()
    node_seqs = {}
    for i in range(g.min_id, g.max_id + 1):
        node_seqs[i] = seq[off[i]:off[i + 1]]
    return orc.Graph(node_seqs, g.n_paths, g.pathsgo(), g.pangenome_base.copy(), g.mappability.copy())


def orc_alnset_from_product(a):
    oa = orc.AlnSet.from_arrays(**a.arrays())
    oa._product_owner = a  # the arrays may be views into the product's memory: keep it alive with the view
    return oa


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.maximum(np.abs(b), 1e-300)
    return np.max(np.abs(a - b) / den) if a.size else 0.0


def orc_graph_nodes_only(g):
    """Oracle graph view for euka / soibean: node sequences only (no path matrix)."""
    off = g.node_seq_off
    seq = g.node_seq.tobytes()
    node_seqs = {i: seq[off[i]:off[i + 1]] for i in range(g.min_id, g.max_id + 1)}
    return orc.Graph(node_seqs, 1, np.zeros((g.max_id + 1, 1), np.uint8), np.full(g.max_id + 1, -1, np.int32), np.ones(1))


def orc_euka_db_from_product(db):
    return orc.EukaDb(db.clade_dist.copy(), db.bin_off.astype(np.int32), db.bin_lo.copy(), db.bin_hi.copy(),
                      db.bin_entropy.copy())


def unsupported_mask(g):
    """[rows][P] bool: path p is NOT supported by node row (the complement of pathsgo over the real paths)."""
    return np.asarray(g.pathsgo())[:, :g.n_paths] == 0


def concat_alnsets(tmp_path, *sets):
    """One alignment set holding the reads of all `sets` in order (through GAM files: gzip members concatenate)."""
    from vgan_amd import haplocart as hc
    blob = b""
    for i, a in enumerate(sets):
        f = str(tmp_path / ("part%d.gam" % i))
        a.write_gam(f)
        blob += open(f, "rb").read()
    out = str(tmp_path / "all.gam")
    open(out, "wb").write(blob)
    return hc.AlnSet.read_gam(out)


def write_euka_db(db, g, directory, prefix="euka_db"):
    """<prefix>.clade / .bins in the reference's formats (src/load.cpp:71-157) and the graph as <prefix>.gfa."""
    import os
    import shutil
    import tempfile
    base = os.path.join(str(directory), prefix)
    v = db.view
    names = db.clade_names
    from vgan_amd.haplocart import _np_view
    npaths, sn, en = (_np_view(p, db.n_clades, np.int32) if p else np.zeros(db.n_clades, np.int32)
                      for p in (v.clade_npaths, v.clade_snode, v.clade_enode))
    with open(base + ".clade", "w") as f:
        for c in range(db.n_clades):
            f.write("%d\t%s\t%.17g\t%d\t%d\t%d\n" % (db.clade_id[c], names[c], db.clade_dist[c], npaths[c], sn[c], en[c]))
    with open(base + ".bins", "w") as f:
        for c in range(db.n_clades):
            cols = [names[c]]
            for j in range(int(db.bin_off[c]), int(db.bin_off[c + 1])):
                cols += ["%d.0" % db.bin_lo[j], "%d.0" % db.bin_hi[j], "%.17g" % db.bin_entropy[j]]
            f.write("\t".join(cols) + "\n")
    tmp = tempfile.mkdtemp(dir=str(directory))
    g.write(tmp)
    shutil.move(os.path.join(tmp, "graph.gfa"), base + ".gfa")
    shutil.rmtree(tmp)
    return base


def orc_graph_from_hcfiles(directory):
    """Oracle graph read from an hcfiles directory WITHOUT any product code: node sequences by the test-side GFA reader,
    path_supports / parsed_pangenome_mapping / mappability.tsv / graph_paths by the oracle's restated loaders
    (src/load.cpp:6-58,283-300).  Returns (orc.Graph, path names, parents text, children text)."""
    import ctypes as C
    import os

    import gamio
    L = orc.lib()

    def raw(name):
        for cand in (name, name + ".gz"):
            p = os.path.join(directory, cand)
            if os.path.exists(p):
                b = open(p, "rb").read()
                return gamio.gunzip_all(b) if b[:2] == b"\x1f\x8b" else b
        raise FileNotFoundError(name)

    node_seqs, _ = orc.read_gfa(os.path.join(directory, "graph.gfa"))
    max_id = max(node_seqs)
    names = [ln.split()[0] for ln in raw("graph_paths").decode().splitlines() if ln.strip()]
    P = len(names)
    rows = np.zeros((max_id + 1, P), np.uint8)
    n = L.orc_load_path_supports(raw("path_supports"), C.c_int32(P), rows.ctypes.data_as(C.c_void_p), C.c_int64(max_id + 1))
    assert n == max_id + 1
    pb = np.full(max_id + 1, -1, np.int32)
    L.orc_load_pangenome_map(raw("parsed_pangenome_mapping"), pb.ctypes.data_as(C.c_void_p), C.c_int64(max_id + 1))
    mtxt = raw("mappability.tsv")
    mp = np.zeros(1 << 16)
    nm = L.orc_load_mappabilities(mtxt, mp.ctypes.data_as(C.c_void_p), C.c_int64(len(mp)))
    if nm > len(mp):  # the loader reports the full count: size the table by it
        mp = np.zeros(nm)
        nm = L.orc_load_mappabilities(mtxt, mp.ctypes.data_as(C.c_void_p), C.c_int64(len(mp)))
    assert 0 < nm <= len(mp)
    return orc.Graph(node_seqs, P, rows, pb, mp[:nm].copy()), names, raw("parents.txt").decode(), raw("children.txt").decode()


def gamio_dicts_from_product(a, r0=0, r1=None):
    """The reads [r0, r1) of a product alignment set as gamio-style dicts (to be re-encoded by the test-side GAM writer)."""
    arr = a.arrays()
    r1 = a.n_reads if r1 is None else r1
    out = []
    for r in range(r0, r1):
        maps = []
        for m in range(arr["map_off"][r], arr["map_off"][r + 1]):
            edits = [{"from_length": int(arr["e_from"][e]), "to_length": int(arr["e_to"][e]),
                      "sequence": bytes(arr["e_seq"][arr["e_seq_off"][e]:arr["e_seq_off"][e + 1]])}
                     for e in range(arr["edit_off"][m], arr["edit_off"][m + 1])]
            maps.append({"position": {"node_id": int(arr["m_node"][m]), "offset": int(arr["m_offset"][m]), "is_reverse": bool(arr["m_rev"][m])},
                         "edit": edits, "rank": len(maps) + 1})
        out.append({"sequence": bytes(arr["seq"][arr["seq_off"][r]:arr["seq_off"][r + 1]]),
                    "quality": bytes(arr["qual"][arr["qual_off"][r]:arr["qual_off"][r + 1]]),
                    "mapping_quality": int(arr["mapq"][r]), "identity": float(arr["identity"][r]),
                    "name": bytes(arr["name"][arr["name_off"][r]:arr["name_off"][r + 1]]),
                    "path": {"name": b"", "mapping": maps}})
    return out
