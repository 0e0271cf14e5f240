"""Pin the oracle's a1 (reconstruct_graph_sequence) on the reference's own known-answer tests
(src/test.cpp:855-994): 10 reads x {graph_seq, read_seq} = 20 strings, exact."""
import json
import os

import gamio
import orc


def test_reconstruction_kats(golden_dir):
    d = os.path.join(golden_dir, "reconstruct")
    g, names = orc.graph_from_gfa(os.path.join(d, "target_graph.gfa"))
    assert names == ["seq_1", "seq_2", "seq_3", "seq_4", "seq_5"]
    alns = gamio.read_gam(os.path.join(d, "test_reads.gam"))
    assert len(alns) == 10
    a = orc.AlnSet(alns)
    exp = json.load(open(os.path.join(d, "expected.json")))["cases"]
    for case in exp:
        rc, gs, rs, sizes = orc.reconstruct(g, a, case["read"])
        assert rc == 0, case["name"]
        assert gs.decode() == case["graph_seq"], case["name"]
        assert rs.decode() == case["read_seq"], case["name"]
        assert sizes == case["mppg_sizes"], case["name"]
