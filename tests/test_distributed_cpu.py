"""world_size-2 gloo test of the N>1 path on CPU: reads shard by contiguous ranges, each rank's final_vec is
summed onto rank 0 with the product's reduce helper.  The per-rank vectors come from the oracle here (no GPU in
this container); tests/test_distributed_gpu.py runs the same two-rank reduce over the product's device vectors, and
bench.py's own rank launcher, on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import orc
import util
from vgan_amd import distributed as vd
from vgan_amd import haplocart as hc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = vd.init(backend="gloo")
    assert (r, w) == (rank, world)
    g = hc.synth_graph(seed=12, genome_len=900, n_nodes=620, n_paths=90)
    a = hc.synth_reads(g, 101, seed=2, read_len=80)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    r0, r1 = vd.shard_bounds(a.n_reads, rank, world)
    _, mine, _ = orc.hc_run(og, oa, r0=r0, r1=r1, n_threads=2, faithful=False)
    t = torch.from_numpy(mine.copy())
    vd.reduce_loglik(t, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 100, 1000003):
        for w in (1, 2, 3, 8):
            b = [vd.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(x[1] - x[0] for x in b) - min(x[1] - x[0] for x in b) <= 1


def test_two_rank_reduce_matches_single_process(tmp_path):
    out = str(tmp_path / "final.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    g = hc.synth_graph(seed=12, genome_len=900, n_nodes=620, n_paths=90)
    a = hc.synth_reads(g, 101, seed=2, read_len=80)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(a)
    _, ref, _ = orc.hc_run(og, oa, n_threads=2, faithful=False)
    assert util.rel_err(got, ref) < 1e-12


def _preflight_worker(rank, world, port, out_dir, hang_rank):
    import json
    import struct
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    vd.init(backend="gloo")
    if rank == hang_rank:  # a rank that never reaches the collectives: the others' watchdogs must end the job, with a reason
        import time
        time.sleep(30)
        os._exit(0)
    rec = vd.preflight(None, timeout_s=3.0 if hang_rank >= 0 else 30.0)
    # the one-collective gather of a fixed-point sum's three words (bench.py --path soibean): every rank sees every rank's words
    f = 0.25 * (rank + 1)
    parts = vd.all_gather_words([rank + 1, (1 << 64) - 1 - rank, struct.unpack("<q", struct.pack("<d", f))[0]])
    with open(os.path.join(out_dir, "r%d.json" % rank), "w") as fh:
        json.dump({"rec": rec, "parts": parts}, fh)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_preflight_and_the_one_collective_gather_over_two_ranks(tmp_path):
    import json
    import struct
    mp.spawn(_preflight_worker, args=(2, _free_port(), str(tmp_path), -1), nprocs=2, join=True)
    for r in range(2):
        d = json.load(open(tmp_path / ("r%d.json" % r)))
        assert d["rec"]["ok"] and d["rec"]["world_size"] == 2 and d["rec"]["backend"] == "gloo" and d["rec"]["int64_sum"] == [3, 3 << 40]
        assert d["parts"] == [[1, (1 << 64) - 1, struct.unpack("<Q", struct.pack("<d", 0.25))[0]],
                              [2, (1 << 64) - 2, struct.unpack("<Q", struct.pack("<d", 0.5))[0]]]


def test_preflight_ends_a_job_whose_rank_never_arrives(tmp_path):
    """The failure mode of a first multi-GPU run: one rank stuck (here: asleep).  The rank that waits leaves with exit code 3 inside
    the timeout and says where it was, instead of hanging in the collective."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_distributed_cpu as t, torch.multiprocessing as mp\n"
            "mp.spawn(t._preflight_worker, args=(2, %d, %r, 1), nprocs=2, join=True)\n") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), _free_port(), str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "[vgan preflight] rank 0 of 2" in r.stderr and "still in 'all_reduce(int64)'" in r.stderr
