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
