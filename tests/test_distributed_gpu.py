"""N > 1 path on the GPU box (one physical GPU): ranks are separate processes sharing cuda:0, rendezvous over gloo,
per-rank vectors come from the PRODUCT (device tensors out of vgan_hc_finalize), and the reduced result is held against
the one-rank run of the same reads.  bench.py --gpus N must start its ranks itself."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, n_reads):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from vgan_amd import distributed as vd
    from vgan_amd import haplocart as hc
    vd.init(backend="gloo")
    dev = torch.device("cuda", 0)
    g = hc.synth_graph(seed=12, genome_len=2000, n_nodes=1400, n_paths=333)
    r0, r1 = vd.shard_bounds(n_reads, rank, world)
    a = hc.synth_reads(g, r1 - r0, seed=5, read_len=120, first_read=r0)
    db = hc.DeviceBatch(hc.HostBatch(g, a), dev)
    ctx = hc.HcContext(g, device=0)
    ctx.use_torch_stream()
    fin = torch.zeros(g.n_paths, dtype=torch.float64, device=dev)
    ctx.accumulate(db)
    ctx.finalize_device(fin)
    vd.reduce_loglik(fin, dst=0)
    if rank == 0:
        np.save(out_path, fin.cpu().numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_on_one_gpu_reduce_product_vectors(tmp_path):
    import torch.multiprocessing as mp
    from vgan_amd import haplocart as hc
    n = 6001
    out = str(tmp_path / "final.npy")
    mp.spawn(_worker, args=(2, _free_port(), out, n), nprocs=2, join=True)
    got = np.load(out)
    g = hc.synth_graph(seed=12, genome_len=2000, n_nodes=1400, n_paths=333)
    a = hc.synth_reads(g, n, seed=5, read_len=120)
    ctx = hc.HcContext(g)
    ctx.accumulate(hc.HostBatch(g, a))
    one = ctx.finalize()
    assert np.max(np.abs(got - one) / np.maximum(np.abs(one), 1e-300)) < 1e-12
    # and the shards really are the one stream: rank 1's first read is read 3000 of it
    b = hc.synth_reads(g, 5, seed=5, read_len=120, first_read=3000)
    sa, sb = a.arrays(), b.arrays()
    assert bytes(sa["seq"][sa["seq_off"][3000]:sa["seq_off"][3005]]) == bytes(sb["seq"][:sb["seq_off"][5]])


def _bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-extra",
           "--no-pmc", "--no-frontend", "--no-ingest"] + list(extra)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_weak_and_strong():
    one = _bench("--reads", "60000")
    assert one["n_gpus"] == 1 and one["scaling"] == "weak" and one["config"]["reads_total"] == one["config"]["reads_per_gpu"]
    two = _bench("--gpus", "2", "--dist-backend", "gloo", "--scaling", "weak", "--reads", "30000")
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["physical_gpus"] == 1
    assert two["config"]["reads_total"] == one["config"]["reads_total"]  # ranks hold reads [0, 30000) and [30000, 60000)
    assert len(two["per_rank"]) == 2 and two["reduce_ms"] > 0
    # (N > 1 defaults to strong scaling over --reads-total, BASELINE configs[2]; the weak figure is measured beside it)
    strong = _bench("--gpus", "3", "--dist-backend", "gloo", "--reads-total", "60000")
    assert strong["n_gpus"] == 3 and strong["scaling"] == "strong" and strong["config"]["reads_total"] == one["config"]["reads_total"]
    assert strong["dist"]["world_size"] == 3 and strong["dist"]["backend"] == "gloo" and len(strong["dist"]["ranks"]) == 3
    assert strong["dist"]["distinct_devices"] == 1 and strong["weak_beside"]["scaling"] == "weak" and strong["weak_beside"]["value"] > 0
    for other in (two, strong):  # the same read set whatever the sharding: same result up to the summation order
        # (paths with identical node sets tie exactly in exact arithmetic, so the argmax itself may be any of them)
        assert other["result_check"]["max_final_vec"] == pytest.approx(one["result_check"]["max_final_vec"], rel=1e-12)
        assert other["result_check"]["sum_final_vec"] == pytest.approx(one["result_check"]["sum_final_vec"], rel=1e-12)
        assert other["posterior"]["confidence"] == pytest.approx(one["posterior"]["confidence"], rel=1e-6)
    # (bound / achieved / peak / frac are one coherent HBM record; what limits the kernel is named beside it)
    assert one["posterior_ms"] > 0 and one["roofline"]["bound"] == "hbm" and 0 < one["roofline"]["frac"] < 1
    # (the batch arrives in the kernel's layout from the host flatten: no layout pass on the device, inside the step or beside it)
    assert one["kernel_ms_per_step"]["pack"] == 0 and "limiter" in one["roofline"]
    # (both byte bases of the roofline fraction are printed; the stricter one is `frac`)
    bb = one["roofline"]["byte_bases"]
    assert bb["frac_on_input_bytes"] == pytest.approx(one["roofline"]["frac"]) and bb["frac_on_survey_8d"] > bb["frac_on_input_bytes"]


def test_bench_preflight_over_two_ranks():
    """bench.py --gpus N --preflight: the process group comes up, the collectives' self-test runs on the tensors the job would use,
    rank 0 prints its record, everyone leaves with 0 (and the same self-test opens every N > 1 run: the `dist` record carries it)."""
    pre = _bench("--gpus", "2", "--dist-backend", "gloo", "--preflight")
    assert pre["n_gpus"] == 2 and pre["preflight"]["ok"] and pre["preflight"]["world_size"] == 2 and pre["preflight"]["backend"] == "gloo"
    two = _bench("--gpus", "2", "--dist-backend", "gloo", "--scaling", "weak", "--reads", "20000")
    assert two["dist"]["preflight"]["ok"] and two["dist"]["preflight"]["int64_sum"] == [3, 3 << 40]


def test_soibean_bench_over_two_ranks_gives_the_one_rank_chain():
    """bench.py --path soibean: two ranks holding reads [0, R) and [R, 2R) of the job's stream all-reduce the state's
    fixed-point sum every iteration; the chain -- every log-likelihood, every accept -- is the one rank's over [0, 2R)."""
    one = _bench("--path", "soibean", "--reads", "40000", "--steps", "12", "--warmup", "3")
    two = _bench("--path", "soibean", "--gpus", "2", "--dist-backend", "gloo", "--reads", "20000", "--steps", "12", "--warmup", "3")
    assert two["n_gpus"] == 2 and "all-reduce" in two["config"]["sharding"]
    assert two["result_check"]["accepted"] == one["result_check"]["accepted"]
    # (a read needs its neighbours for nothing, so the halves' reads are the whole's reads; unusable reads are dropped per share)
    assert two["result_check"]["last_loglike"] == one["result_check"]["last_loglike"]


def test_several_contexts_in_one_process_reduce_to_the_single_context_result():
    """vgan_hc_reduce: the C++ multi-GPU path (one context per GPU, chunks dealt round-robin, one reduce of P doubles).  On a
    one-GPU box the contexts share the device, so the reduce goes through the host; the collective needs distinct GPUs."""
    from vgan_amd import haplocart as hc
    g = hc.synth_graph(seed=21, genome_len=3000, n_nodes=2100, n_paths=200)
    chunks = [hc.synth_reads(g, 2500, seed=3, read_len=140, first_read=2500 * i) for i in range(6)]
    one = hc.HcContext(g)
    for a in chunks:
        one.accumulate(hc.HostBatch(g, a))
    want = one.finalize()
    ctxs = [hc.HcContext(g, device=0) for _ in range(3)]
    for i, a in enumerate(chunks):
        ctxs[i % 3].accumulate(hc.HostBatch(g, a))
    got, used_rccl = hc.reduce_contexts(ctxs)
    assert not used_rccl  # three contexts on one device
    assert np.max(np.abs(got - want) / np.abs(want)) < 1e-12
    # a single context reduces to itself (a finalize sums with atomics: equal to rounding, not to the bit)
    alone, _ = hc.reduce_contexts([one])
    assert np.max(np.abs(alone - want) / np.abs(want)) < 1e-13


def test_cli_deals_the_reads_to_several_device_contexts(tmp_path):
    """`vgan haplocart --gpus LIST`: the same result line, log-likelihood table and posteriors from two contexts (here both
    on GPU 0) as from one; `-t N` sizes the host side only."""
    from vgan_amd import haplocart as hc
    g = hc.synth_graph(seed=5, genome_len=1500, n_nodes=1050, n_paths=60)
    a = hc.synth_reads(g, 180_000, seed=1, read_len=100)
    g.write(str(tmp_path))
    gam = str(tmp_path / "in.gam")
    a.write_gam(gam)
    exe = os.path.join(ROOT, "vgan_amd", "bin", "vgan")
    outs = {}
    for tag, extra in (("one", []), ("two", ["--gpus", "0,0"]), ("t4", ["-t", "4"])):
        out = str(tmp_path / (tag + ".tsv"))
        r = subprocess.run([exe, "haplocart", "-g", gam, "--hc-files", str(tmp_path), "-o", out, "-pf", out + ".post", "-d", "-s", "x"] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        assert ("Reduced the log-likelihoods of 2 device contexts (host)" in r.stderr) == (tag == "two")
        ll = {ln.split("\t")[0]: float(ln.split("\t")[1]) for ln in open(out + ".loglik.tsv").read().splitlines()}
        outs[tag] = (open(out).read(), ll, open(out + ".post").read())
    assert outs["one"][0] == outs["two"][0] == outs["t4"][0] and outs["one"][0].splitlines()[1].startswith("x\thg")
    for other in ("two", "t4"):  # -t 4 on a one-GPU box is one context again
        assert outs[other][1].keys() == outs["one"][1].keys()
        assert all(outs[other][1][k] == pytest.approx(v, rel=1e-10) for k, v in outs["one"][1].items())
        f1, f2 = outs["one"][2].split("\t"), outs[other][2].split("\t")
        assert len(f1) == len(f2) and f1[0] == f2[0]
        for x, y in zip(f1[1:], f2[1:]):
            try:
                assert float(y) == pytest.approx(float(x), rel=1e-5)
            except ValueError:
                assert x == y


def _n_gpus():
    from vgan_amd import _native as N
    return N.lib().vgan_device_count()


@pytest.mark.skipif("_n_gpus() < 2")
def test_rccl_reduce_between_two_distinct_gpus(tmp_path, monkeypatch):
    """Needs two GPUs (skipped on the one-GPU rig): the RCCL branch of vgan_hc_reduce -- one communicator per device set,
    created on the first reduce and reused by the second -- and the CLI with --gpus 0,1, against the one-context result."""
    from vgan_amd import haplocart as hc
    g = hc.synth_graph(seed=23, genome_len=3000, n_nodes=2100, n_paths=200)
    chunks = [hc.synth_reads(g, 4000, seed=7, read_len=150, first_read=4000 * i) for i in range(4)]
    one = hc.HcContext(g, device=0)
    for a in chunks:
        one.accumulate(hc.HostBatch(g, a))
    want = one.finalize()
    ctxs = [hc.HcContext(g, device=d) for d in (0, 1)]
    for i, a in enumerate(chunks):
        ctxs[i % 2].accumulate(hc.HostBatch(g, a))
    monkeypatch.setenv("VGAN_HC_REDUCE", "host")
    got, used = hc.reduce_contexts(ctxs)  # asked to stay on the host
    assert not used and np.max(np.abs(got - want) / np.abs(want)) < 1e-12
    monkeypatch.delenv("VGAN_HC_REDUCE")
    n0 = hc.reduce_info()[1]
    got, used = hc.reduce_contexts(ctxs)  # distinct devices: the collective, its communicator made here
    assert used and np.max(np.abs(got - want) / np.abs(want)) < 1e-12
    ms, n1 = hc.reduce_info()
    assert n1 == n0 + 1 and ms > 0
    got, used = hc.reduce_contexts(ctxs)  # the cached communicator again
    assert used and hc.reduce_info()[1] == n1 and np.max(np.abs(got - want) / np.abs(want)) < 1e-12
    # the CLI on the two GPUs
    a = hc.synth_reads(g, 120_000, seed=9, read_len=100)
    g.write(str(tmp_path))
    gam = str(tmp_path / "in.gam")
    a.write_gam(gam)
    exe = os.path.join(ROOT, "vgan_amd", "bin", "vgan")
    lls = {}
    for tag, extra, env in (("one", [], {}), ("host", ["--gpus", "0,1"], {"VGAN_HC_REDUCE": "host"}), ("rccl", ["--gpus", "0,1"], {})):
        out = str(tmp_path / (tag + ".tsv"))
        r = subprocess.run([exe, "haplocart", "-g", gam, "--hc-files", str(tmp_path), "-o", out, "-np", "-d", "-s", "x"] + extra,
                           capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-1500:]
        if tag != "one":
            assert ("(RCCL)" in r.stderr) == (tag == "rccl") and ("communicator over 2 devices set up in" in r.stderr) == (tag == "rccl")
        lls[tag] = {ln.split("\t")[0]: float(ln.split("\t")[1]) for ln in open(out + ".loglik.tsv").read().splitlines()}
    for tag in ("host", "rccl"):
        assert all(lls[tag][k] == pytest.approx(v, rel=1e-12) for k, v in lls["one"].items())
