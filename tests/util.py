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
