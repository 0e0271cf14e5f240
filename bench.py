#!/usr/bin/env python3
"""bench.py -- reads/sec through the HaploCart posterior path on MI355X (BASELINE.json metric).

A "step" is one complete pass of the hot path over one batch of synthetic reads already resident in HBM:
reset accumulators -> per-read likelihood kernels over the batch -> final_vec[P] on the device
(-> RCCL sum-reduce of the P doubles to rank 0 when N > 1).  Workload at N=1: BASELINE configs[1]
(1M synthetic 150 bp reads against the hcfiles-shaped graph: 11821 nodes, 5179 paths); with N ranks each rank
holds its own 1M-read shard (weak scaling; reads shard with no data-path collective).

One JSON line on rank 0.  `roofline` is the dominant kernel's ALGORITHMIC bytes per launch / its mean launch
duration (HIP events on the launch stream, measured live by the library) against the 8 TB/s HBM peak.
`cpu_baseline` is the CPU oracle (a port: the reference cannot be built here) timed on a bounded sample.
"""
import argparse
import json
import os
import sys
import tempfile
import time

# (as the vgan CLI does: the device front end's pipeline -- the `front_end.device_flatten.device_gam` record -- keeps a dozen streams busy,
# which the runtime's default four hardware queues serialise; the timed step is one stream and does not care.  Before the runtime starts.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PREFLIGHT = None  # N > 1: what vgan_amd.distributed.preflight() found (part of the line's `dist` record)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--mode", choices=["node_weights", "per_read", "per_read_dense"], default="node_weights")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x76676131)
    ap.add_argument("--no-parity", action="store_true", help="accepted for old command lines; the parity diff is part of the cpu_baseline leg (--cpu-seconds 0 skips both)")
    ap.add_argument("--no-extra", action="store_true", help="skip the per-read-mode kernel measurement")
    ap.add_argument("--sb-paths", type=int, default=28, help="soibean path: paths of the synthetic tree (28: BASELINE config 5's shape; 210: the size of soibean_db.clade)")
    ap.add_argument("--clades", type=int, default=335, help="euka path: number of clades the synthetic reads spread over")
    ap.add_argument("--dist-backend", default="auto", help="auto: nccl (= RCCL) when every rank has a GPU of its own, else gloo with ranks sharing GPUs | nccl | gloo")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="weak: --reads per GPU (the default at N = 1: BASELINE configs[1]); strong: --reads-total split over the N "
                         "GPUs (the default at N > 1: BASELINE configs[2], 10M reads; the weak figure is measured beside it)")
    ap.add_argument("--reads-total", type=int, default=10_000_000, help="strong scaling: reads of the whole job")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--no-frontend", action="store_true", help="skip the GAM decode / flatten rates")
    ap.add_argument("--preflight", action="store_true", help="N > 1: initialise the process group, run the collectives' self-test, print its record and leave")
    ap.add_argument("--preflight-timeout", type=float, default=60.0, help="seconds a rank may spend inside the self-test's collectives")
    ap.add_argument("--no-ingest", action="store_true", help="skip the host-to-device / streamed single-pass figures")
    ap.add_argument("--e2e-reads", type=int, default=10_000_000, help="reads of the file the end_to_end record runs `vgan haplocart` on (0: skip; --no-frontend skips it too)")
    ap.add_argument("--path", choices=["haplocart", "euka", "soibean"], default="haplocart",
                    help="haplocart = the BASELINE metric; euka / soibean = configs 4 / 5 as extra lines")
    return ap.parse_args()


def pick_device(args):
    """(rank, world, local device index, torch device, backend) of this process.  --dist-backend auto: RCCL when every
    rank has a GPU of its own, else gloo with the ranks sharing the GPUs there are (test rigs)."""
    import torch
    from vgan_amd import distributed as vd
    rank, world, local_rank = vd.env_rank()
    n_dev = torch.cuda.device_count()  # counting does not initialise the GPU
    if n_dev == 0 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    backend = args.dist_backend
    if backend == "auto":
        backend = "nccl" if n_dev >= world else "gloo"
    if backend == "gloo":
        local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    vd.init(backend=backend, device=dev)
    # N > 1: the group is exercised before anything is built on it (one int64 and one float64 all-reduce and a reduce to rank 0 on
    # the tensors the job uses, under a watchdog: a rank stuck in a collective leaves with exit code 3 and a reason within
    # --preflight-timeout seconds).  --preflight: that, the record on rank 0's stdout, and out.
    global PREFLIGHT
    PREFLIGHT = vd.preflight(dev, timeout_s=args.preflight_timeout) if world > 1 else None
    if args.preflight:
        if rank == 0:
            import torch.distributed as dist
            print(json.dumps({"preflight": PREFLIGHT or {"world_size": 1, "ok": True}, "backend": backend if world > 1 else None, "n_gpus": world,
                              "visible_devices": n_dev}), flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(0)
    return rank, world, local_rank, dev, backend, n_dev


def cpu_baseline(graph, alns, budget_s, ctx=None, hc=None):
    """The oracle's literal restatement of the reference loop (OpenMP over reads, per-read vector, critical
    accumulate: src/HaploCart.cpp:408-421) on the host cores, on as many reads as fit the budget.  This is the only
    place the benchmark touches oracle/: the vector it produces for its sample is also what the device result of the
    same reads is diffed against (outside any timed region)."""
    import numpy as np
    import orc
    import util
    og, oa = util.orc_graph_from_product(graph), util.orc_alnset_from_product(alns)
    # as many OpenMP threads as the container may keep busy (affinity mask and cgroup CPU quota): threads beyond the quota are
    # time-sliced on the same processors and only slow the loop down
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(affinity, host_cpu_quota()))
    done, t_used = 0, 0.0
    chunk = cores
    ref = None
    t0 = time.perf_counter()
    while t_used < budget_s and done < alns.n_reads:
        n = min(chunk, alns.n_reads - done)
        _, part, _ = orc.hc_run(og, oa, r0=done, r1=done + n, n_threads=cores, faithful=True)
        ref = np.asarray(part, np.float64) if ref is None else ref + np.asarray(part, np.float64)
        done += n
        t_used = time.perf_counter() - t0
        if t_used < budget_s / 4:
            chunk *= 2
    faithful = done / t_used
    # hoisted variant (S_m/U_m once per mapping; NOT the reference's cost profile) on a larger sample
    n_h = min(alns.n_reads, 20000)
    t0 = time.perf_counter()
    orc.hc_run(og, oa, r0=0, r1=n_h, n_threads=cores, faithful=False)
    hoisted = n_h / (time.perf_counter() - t0)
    out = {"value": faithful, "unit": "reads/s", "cores": cores, "kind": "port",
           "sample": "first %d of the workload's reads, literal reference loops (oracle, long double, OpenMP x%d)" % (done, cores),
           "hoisted_variant_reads_per_s": hoisted, "cpu_quota": host_cpu_quota(), "affinity_cpus": affinity}
    parity = None
    if ctx is not None and ref is not None:  # the device on the same sample
        sub = hc.HostBatch(graph, alns, 0, done, packed=True)
        ctx.reset()
        ctx.accumulate(sub)
        got = ctx.finalize()
        parity = {"reads": done, "max_rel_err_vs_oracle": float(util.rel_err(got, ref)), "tolerance": 1e-6}
    return out, parity


def cpu_baseline_euka(g, db, alns, dm_texts, ctx, ek, budget_s):
    """The oracle's restatement of readGAM3's per-alignment lambda, serial as in the reference
    (src/readGAM_Euka.h:581), on a bounded prefix of the workload; the device result of the same reads is diffed
    against it (the only use of oracle/ in this leg)."""
    import numpy as np
    import orc
    import util
    og, odb, dmg = util.orc_graph_nodes_only(g), util.orc_euka_db_from_product(db), orc.OrcDamage(*dm_texts)
    n, rate, ref, sub = 2000, None, None, None
    while True:
        n = min(n, alns.n_reads)
        drop = np.ones(alns.n_reads, np.uint8)
        drop[:n] = 0
        sub = alns.without(drop)
        t0 = time.perf_counter()
        ref = orc.euka_run(og, util.orc_alnset_from_product(sub), odb, dmg)
        dt = time.perf_counter() - t0
        rate = n / dt
        if dt >= budget_s / 3 or n >= alns.n_reads:
            break
        n = int(min(alns.n_reads, max(2 * n, n * budget_s / max(dt, 1e-3) * 0.6)))
    hb = ek.EukaHostBatch(g, sub)
    ctx.reset()
    got = ctx.accumulate(hb)
    src = hb.arrays()["read_src"]
    ok = got["clade"] >= 0
    err = max(float(util.rel_err(got[k][ok], ref[k][src][ok])) for k in ("in_lik", "out_lik", "like")) if ok.any() else 0.0
    same = bool(np.array_equal(got["clade"], ref["clade"][src]) and np.array_equal(got["pass"], ref["pass"][src]))
    return ({"value": rate, "unit": "reads/s", "cores": 1, "kind": "port",
             "sample": "first %d of the workload's reads, per-alignment lambda restated serially (oracle, long double)" % n},
            {"reads": int(n), "max_rel_err_vs_oracle": err, "clade_and_pass_identical": same, "tolerance": 1e-6})


def bench_euka(args):
    """BASELINE config 4 shape: synthetic 75 bp aDNA reads with the dhigh damage profiles against a 335-clade graph."""
    import numpy as np
    import torch
    from vgan_amd import distributed as vd
    from vgan_amd import euka as ek
    rank, world, local_rank, dev, backend, n_dev = pick_device(args)
    gold = os.path.join(ROOT, "tests", "golden", "damageProfiles")
    dm = ek.Damage.load(os.path.join(gold, "dhigh5p.prof"), os.path.join(gold, "dhigh3p.prof"))
    dm_texts = (open(os.path.join(gold, "dhigh5p.prof")).read(), open(os.path.join(gold, "dhigh3p.prof")).read())
    g, db, alns = ek.synth_euka(args.reads, dm, seed=args.seed, n_clades=args.clades, nodes_per_clade=400,
                                read_len_mean=75, read_seed=args.seed + 1000003 * rank)
    hb = ek.EukaHostBatch(g, alns)
    dbt = ek.EukaDeviceBatch(hb, dev)
    ctx = ek.EukaContext(db, dm, device=local_rank)
    ctx.use_torch_stream()

    def step():
        ctx.reset()
        ctx.accumulate(dbt)

    import torch.distributed as dist
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.kernel_ms()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = vd.all_reduce_max(time.perf_counter() - t0, dev)
    ms, n = ctx.kernel_ms()
    fin = ctx.finalize()
    n_total = int(vd.all_reduce_sum(float(hb.n_reads), dev))
    if world > 1:  # per-clade accumulators are summed over ranks once per job (per-read outputs stay sharded)
        for k in ("clade_count", "baseshift", "bin_cov"):
            t = torch.from_numpy(np.ascontiguousarray(fin[k])).to(dev if dist.get_backend() != "gloo" else "cpu")
            t = t.to(torch.float64) if t.dtype not in (torch.float64, torch.int32, torch.int64) else t
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
            fin[k] = t.cpu().numpy()
    if rank == 0:
        euka_traffic = collect_traffic(args, "euka_read_kernel") if world == 1 and not args.no_pmc else None
        kb = hb.algorithmic_bytes()
        avg = ms / max(n, 1)
        euka_issue = collect_issue(args, "euka_read_kernel", avg) if world == 1 and not args.no_pmc else None
        gbs = kb / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        out = {
            "metric": "reads/sec through euka per-read two-model likelihood (readGAM3), 75bp aDNA", "value": n_total * args.steps / elapsed,
            "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "euka %d synthetic 75bp aDNA reads per GPU, dhigh damage profiles, 335-clade graph" % args.reads,
                       "reads_per_gpu": hb.n_reads, "passing_reads": int(fin["clade_count"].sum())},
            # (HBM figures as one coherent record; the kernel's limiter is fp64 VALU issue: one table log + the damage-matrix
            # products per base, SURVEY 8d.  traffic: in-run rocprofv3 --pmc passes, as on the HaploCart line)
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": euka_traffic.get("bytes") if euka_traffic else None, "traffic_detail": euka_traffic,
                         "kernel": "euka_read_kernel", "algorithmic_bytes_per_launch": kb, "avg_launch_ms": avg,
                         "launches": n,
                         # this kernel is fp64-VALU bound -- the ALU-side fraction (vector instruction issue) beside the HBM one
                         "alu": euka_issue, "limiter": dict(euka_issue or {}, kind="valu (fp64) issue")}}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"], out["parity"] = cpu_baseline_euka(g, db, alns, dm_texts, ctx, ek, args.cpu_seconds)
        print(json.dumps(out), flush=True)


SB_PROFILED_READS = 1998288  # reads of the committed soibean PMC passes (bench.py --path soibean --reads 2000000)


def committed_traffic(profile, kernel, applies):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in
    separate runs of the same bench command, profiles/<profile>_pmc.json); None when the workload differs or no pass is
    committed."""
    if not applies:
        return None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", profile + "_pmc.json")))
        for kn, v in pmc.items():
            if kernel in kn:
                return {"bytes": v["fetch_bytes"] + v["write_bytes"], "profile": "profiles/%s_pmc.json" % profile,
                        "note": "uncorrected FETCH_SIZE + WRITE_SIZE of a committed rocprofv3 --pmc pass, not measured in this run"}
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline_soibean(g, alns, dm, sb, state_fn, freqs, budget_s):
    """The oracle's restatement of analyse_GAM + one MCMC likelihood refresh (OpenMP reduction over reads as
    src/MCMC.cpp:739) on a bounded prefix of the workload; the device refresh of the same reads and the same state
    is diffed against it (the only use of oracle/ in this leg)."""
    import numpy as np
    import orc
    import util
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(affinity, host_cpu_quota()))  # (threads beyond the container's CPU quota only time-slice)
    n = min(alns.n_reads, 20000)
    drop = np.ones(alns.n_reads, np.uint8)
    drop[:n] = 0
    sub = alns.without(drop)
    og, oa = util.orc_graph_from_product(g), util.orc_alnset_from_product(sub)
    so = orc.SbOracle(og, oa, orc.OrcDamage("", ""))
    st = state_fn()[0]
    reps, t_used, ref = 0, 0.0, None
    t0 = time.perf_counter()
    while t_used < budget_s / 2 and reps < 200:
        rc, ref = so.loglike(st, 0.01, freqs, n_threads=cores)
        assert rc == 0
        reps += 1
        t_used = time.perf_counter() - t0
    rate = n * reps / t_used
    hb = sb.SbHostBatch(g, sub)
    ctx = sb.SbContext(g, dm)
    ctx.precompute(hb)
    got = float(ctx.loglike([st], 0.01, freqs)[0][0])
    err = abs(got - ref) / max(abs(ref), 1e-300)
    return ({"value": rate, "unit": "reads*iterations/s", "cores": cores, "kind": "port", "cpu_quota": host_cpu_quota(), "affinity_cpus": affinity,
             "sample": "refresh over the first %d reads, %d repetitions (oracle, long double, OpenMP x%d)" % (n, reps, cores)},
            {"reads": int(n), "max_rel_err_vs_oracle": float(err), "tolerance": 1e-6})


def bench_soibean(args):
    """BASELINE config 5 shape: k = 3 sources, synthetic reads against a 28-path tree; a host Metropolis loop proposes
    branch positions / proportions and the GPU refreshes the likelihood every iteration (one step = one iteration)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from vgan_amd import distributed as vd
    from vgan_amd import euka as ek
    from vgan_amd import haplocart as hc
    from vgan_amd import soibean as sb
    rank, world, local_rank, dev, backend, n_dev = pick_device(args)
    g = hc.synth_graph(seed=args.seed, genome_len=16569, n_nodes=11000, n_paths=args.sb_paths)
    # the job's reads are ONE seeded stream (read i depends on (seed, i) only); rank r holds reads [r * R, (r + 1) * R): the split
    # MCMC.cpp:739 makes over OpenMP threads (`reduction(+:logLike)`), made over GPUs
    alns = hc.synth_reads(g, args.reads, seed=args.seed, read_len=65, indel_rate=0.005, softclip_rate=0.01, first_read=rank * args.reads)
    dm = ek.Damage.from_text("", "")
    hb = sb.SbHostBatch(g, alns)
    ctx = sb.SbContext(g, dm, device=local_rank)
    ctx.use_torch_stream()
    t0 = time.perf_counter()
    ctx.precompute(hb)
    t_pre = time.perf_counter() - t0
    names = g.path_names
    idx = {n: i for i, n in enumerate(names)}
    pairs = [(idx[t[0]], idx[t[1]]) for t in (ln.split() for ln in g.parents_txt.splitlines()) if len(t) >= 2]
    freqs = [0.31, 0.27, 0.13, 0.29, 0.44, 0.56, 0.0012]
    rng = np.random.default_rng(args.seed)  # same proposal sequence on every rank
    d_out = torch.zeros(1, dtype=torch.float64, device=dev)

    def state():
        src = []
        th = rng.dirichlet([1, 1, 1])
        for y in range(3):
            c, p = pairs[rng.integers(len(pairs))]
            src.append((c, p, 0.01 + 0.05 * rng.random(), rng.random() * 0.98 + 0.01, float(th[y])))
        return [src]

    cur = None
    accepted = 0

    fused = world == 1  # the chain driver's call: one fused kernel + a fold into pinned host memory, nothing to all-reduce
    if fused:
        ctx.time_engine(True)

    # the proposals are drawn before the clock starts (numpy's scalar generators cost more per iteration than the refresh);
    # the timed loop is what the chain driver does per iteration: hand a state to the GPU, get its log-likelihood, accept or not
    proposals = [state() for _ in range(args.warmup + args.steps)]
    uniforms = np.log(rng.random(args.warmup + args.steps))
    it_no = 0

    def iteration():
        nonlocal cur, accepted, it_no
        st, logu = proposals[it_no], uniforms[it_no]
        it_no += 1
        if fused:
            ll, _ = ctx.refresh(st[0], 0.01, freqs)
            if cur is None or logu < ll - cur:
                cur = ll
                accepted += 1
            return
        # N > 1: every rank refreshes its share, then ONE all-reduce of the state's sum per iteration.  The sums are fixed point
        # (vgan_sb_sum: two 64-bit integers and a double for what does not fit), so the total -- and with it every accept /
        # reject -- is the same bits whatever the number of ranks
        sums, _ = ctx.loglike_sums(st, 0.01, freqs)
        hi, lo, nf = sums[0]
        # ONE collective per iteration: the three words of every rank's sum (the double as its bit pattern) gathered on every rank
        # (RCCL over xGMI on device tensors, gloo on host ones), then added in rank order -- sb.sum_value adds the fixed-point
        # parts exactly, so the total does not depend on the number of ranks
        import struct
        parts = vd.all_gather_words([hi, lo, struct.unpack("<q", struct.pack("<d", nf))[0]], dev)
        ll = sb.sum_value([(h - (1 << 64) if h >= (1 << 63) else h, l, struct.unpack("<d", struct.pack("<Q", f))[0]) for h, l, f in parts])
        if cur is None or logu < ll - cur:
            cur = ll
            accepted += 1

    for _ in range(args.warmup):
        iteration()
    torch.cuda.synchronize()
    ctx.kernel_ms()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        iteration()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    km = ctx.kernel_ms()
    if rank == 0:
        R = hb.n_reads
        sb_traffic = collect_traffic(args, "sb_refresh_fused_kernel" if fused else "sb_loglike_kernel") if world == 1 and not args.no_pmc else None
        avg = km["refresh"][0] / max(km["refresh"][1], 1)
        sb_issue = collect_issue(args, "sb_refresh_fused_kernel" if fused else "sb_loglike_kernel", avg) if world == 1 and not args.no_pmc else None
        kb = R * 3 * 2 * (8 + 25 * 2) + R  # 2k path rows of pm (8 B) + cnt (25 x 2 B) + ok flags
        gbs = kb / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        out = {
            "metric": "read-iterations/sec through the soibean MCMC likelihood refresh (k=3)", "value": R * world * args.steps / elapsed,
            "unit": "reads*iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "soibean k=3, %d synthetic reads per GPU, %d-path tree, host Metropolis loop + GPU refresh per iteration" % (args.reads, args.sb_paths),
                       "reads_per_gpu": R, "precompute_s": t_pre, "accepted": accepted,
                       "sharding": "contiguous read ranges x%d, one all-reduce (%s) of the state's fixed-point sum per iteration" % (world, backend) if world > 1 else "single GPU"},
            "result_check": {"last_loglike": cur, "accepted": accepted},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": sb_traffic.get("bytes") if sb_traffic else None, "traffic_detail": sb_traffic,
                         "kernel": "sb_refresh_fused_kernel" if fused else "sb_loglike_kernel + sb_finish_kernel", "algorithmic_bytes_per_launch": kb, "avg_launch_ms": avg,
                         "launches": km["refresh"][1], "alu": sb_issue}}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"], out["parity"] = cpu_baseline_soibean(g, alns, dm, sb, state, freqs, args.cpu_seconds)
        print(json.dumps(out), flush=True)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of this one, which has not
    imported torch or touched HIP (a process that has initialised the GPU must not exec), wait for them and leave with
    the worst exit code.  Rank 0 prints the JSON line on the inherited stdout."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, live, deadline = 0, list(procs), None
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:  # one rank failed: the others would wait in a collective for ever
                rc = code
                deadline = time.time() + 15
        if deadline is not None and live and time.time() > deadline:
            for q in live:
                q.kill()  # exactly the processes started above
            deadline = time.time() + 1e9
    sys.exit(rc if rc >= 0 else 1)


CHUNK_READS = 1_000_000  # reads per device batch (a rank's shard is a list of such batches, all resident in HBM)


def pmc_pass(args, kernel_substr, counters):
    """Mean per-dispatch values of `counters` (one rocprofv3 --pmc pass: they must fit the block's slots) for the kernels whose
    name holds kernel_substr: a child run of this very file, 3 steps, outside every timed region.  {counter: value} or
    {"failed": why}; None where no profiler exists or this run is itself being profiled."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    # a run that is itself being profiled (someone else's rocprofv3 around bench.py) does not start a profiler of its own
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    tmp = tempfile.mkdtemp(prefix="vgan_pmc_")
    try:
        out = os.path.join(tmp, "pass")
        cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
               "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--no-extra", "--no-pmc", "--no-frontend", "--no-ingest",
               "--reads", str(args.reads), "--read-len", str(args.read_len), "--mode", args.mode, "--seed", str(args.seed),
               "--path", args.path, "--clades", str(args.clades)]
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        # its own session: on a timeout the whole group goes (the profiled python is a grandchild of ours and would
        # otherwise keep running on the GPU beside the legs that follow)
        pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = pr.wait(timeout=180)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(pr.pid, 9)
            except OSError:
                pass
            pr.wait()
            return {"failed": "%s pass timed out" % "+".join(counters)}
        if rc != 0:
            return {"failed": "%s pass exited with %d" % ("+".join(counters), rc)}
        vals = {c: [] for c in counters}
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] in vals:
                    vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
        if not all(vals.values()):
            return {"failed": "no %s row for a kernel named *%s*" % ("+".join(c for c in counters if not vals[c]), kernel_substr)}
        return {c: sum(v) / len(v) for c, v in vals.items()}
    except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
        return {"failed": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def collect_traffic(args, kernel_substr):
    """HBM bytes per launch of the dominant kernel, from the PMC counters of THIS tree on THIS box: two child runs under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE cannot share a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units are
    KB per dispatch.  gfx950 corrections per the guide's HBM section: FETCH_SIZE tallies the 128-byte requests of a coalesced
    stream at 64 B, so it is doubled; WRITE_SIZE is exact for 8/16-byte stores and float atomics; Infinity-Cache hits are
    counted.  Raw values are kept beside the sum."""
    raw = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        got = pmc_pass(args, kernel_substr, [counter])
        if got is None:
            return None
        if "failed" in got:
            return {"bytes": None, "failed": got["failed"]}
        raw[counter] = got[counter] * 1024.0
    return {"fetch_size_bytes_raw": raw["FETCH_SIZE"], "write_size_bytes_raw": raw["WRITE_SIZE"],
            "bytes": 2.0 * raw["FETCH_SIZE"] + raw["WRITE_SIZE"],
            "correction": "2 x FETCH_SIZE + WRITE_SIZE (gfx950: coalesced 128-B read requests are tallied at 64 B)"}


VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4  # wave-instructions/s: 1024 SIMDs, one vector instruction of a wave per 4 cycles, 2.4 GHz


def collect_issue(args, kernel_substr, avg_launch_ms):
    """What the dominant kernel does to the chip's instruction issue, from one SQ pass of THIS tree on THIS box (a child run
    under rocprofv3 --pmc, beside the timed region): the ALU-side roofline SURVEY 8d asks for beside the HBM one.  alu_frac =
    VALU wave-instructions per launch / the launch's time / the chip's VALU issue peak (1024 SIMDs x clock / 4: every vector
    instruction of a wave occupies its SIMD for four cycles, measured: SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU = 4.0)."""
    got = pmc_pass(args, kernel_substr, ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE",
                                         "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"])
    if got is None:
        return None
    if "failed" in got:
        return {"alu_frac": None, "failed": got["failed"]}
    cyc = got["GRBM_GUI_ACTIVE"] / 8.0  # (summed over the 8 XCDs)
    rate = got["SQ_INSTS_VALU"] / (avg_launch_ms * 1e-3) if avg_launch_ms > 0 else 0.0
    return {"bound": "valu issue", "achieved": rate / 1e9, "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instructions/s",
            "alu_frac": rate / VALU_ISSUE_PEAK,
            "valu_insts": got["SQ_INSTS_VALU"], "salu_insts": got["SQ_INSTS_SALU"], "lds_insts": got["SQ_INSTS_LDS"],
            # the same launch under the profiler: busy fractions of the SIMDs' vector issue and of the CUs' LDS
            "valu_issue_frac_profiled": got["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024.0) if cyc > 0 else None,
            "lds_busy_frac_profiled": got["SQ_LDS_IDX_ACTIVE"] / (cyc * 256.0) if cyc > 0 else None,
            "lds_bank_conflict_share": got["SQ_LDS_BANK_CONFLICT"] / max(got["SQ_LDS_IDX_ACTIVE"], 1.0),
            "source": "in-run rocprofv3 --pmc pass of this tree (3 steps, beside the timed region)"}


def host_cpu_quota():
    """Processors the container may keep busy (affinity mask and cgroup CPU quota, as the product's host code reads them): the
    OpenMP threads of a cpu_baseline leg share this many, however many were started."""
    from vgan_amd import _native as N
    return int(N.lib().vgan_host_cpus())


def front_end_rates(graph, hc, seed, n=200_000, ctx=None):
    """SURVEY 8f-1 beside the metric (the timed step starts from a flattened batch in HBM): GAM decode and flatten
    rates of the host front end on a bounded sample of the same workload."""
    import tempfile
    a = hc.synth_reads(graph, n, seed=seed, read_len=150)
    with tempfile.TemporaryDirectory(prefix="vgan_fe_") as d:
        p = os.path.join(d, "sample.gam")
        a.write_gam(p)
        size = os.path.getsize(p)
        t0 = time.perf_counter()
        b = hc.AlnSet.read_gam(p)
        t_dec = time.perf_counter() - t0
    t0, c0 = time.perf_counter(), time.process_time()
    hb = hc.HostBatch(graph, b, packed=True)
    t_fl, cpu_fl = time.perf_counter() - t0, time.process_time() - c0
    # a1 on the device (vgan_hc_devflat): the parser's arrays go up as they are (PCIe inside the figure), reconstruct + slicing +
    # layout run as kernels; what `vgan haplocart` does for every chunk of a long input once the device contexts are up
    dev = None
    if ctx is not None:
        with tempfile.TemporaryDirectory(prefix="vgan_fe_") as d:
            p = os.path.join(d, "sample.gam")
            a.write_gam(p)
            parts = hc.AlnParts.read_gam(p)
        df = hc.DeviceFlatten(ctx, graph)
        df.run(parts)  # (buffers, pinned staging)
        t0, c0 = time.perf_counter(), time.process_time()
        res = df.run(parts)
        t_df, cpu_df = time.perf_counter() - t0, time.process_time() - c0
        # (wall: PCIe transfer of the parser's arrays included; host_cpu_us_per_read: what the stage costs the host's processors --
        # on a CPU quota that, not the wall time of one stage, is what a long input's throughput follows)
        dev = {"reads_per_s": parts.n_reads / t_df, "host_cpu_us_per_read": cpu_df / parts.n_reads * 1e6, "reads_taken": int(res.pk.n_reads),
               "reads_left_to_the_host": int(res.host_mask.sum())}
        # f1 on the device as `vgan haplocart` runs it (vgan_hc_accumulate_gam_bytes: csrc/gam_pipe.hip): the FILE's bytes go up in
        # pieces, and BGZF inflate + framing + protobuf wire walk + flatten + the segment kernel run as a pipeline over them; only the
        # messages of the reads the device flatten leaves to the host come back.  (The whole-file figure, with HIP start-up and process
        # exit: the `end_to_end` record.)
        try:
            with tempfile.TemporaryDirectory(prefix="vgan_fe_") as d:
                p = os.path.join(d, "sample.gam")
                a.write_gam(p)
                data = open(p, "rb").read()
            # (ten copies of the sample's file one after the other -- BGZF members concatenate, the end-of-file member of all but the last
            # dropped: a file of the size the device front end is for)
            data = data[:-28] * 9 + data
            ctx.reset()
            hc.accumulate_gam_bytes([ctx], graph, data)  # (buffers, code objects)
            ctx.reset()
            t0, c0 = time.perf_counter(), time.process_time()
            st, ps = hc.accumulate_gam_bytes([ctx], graph, data)
            ctx.synchronize()
            t_all, cpu_all = time.perf_counter() - t0, time.process_time() - c0
            ctx.reset()
            dev["device_gam"] = {"reads_per_s": ps["n_reads"] / t_all, "host_cpu_us_per_read": cpu_all / max(ps["n_reads"], 1) * 1e6,
                                 "gam_bytes": len(data), "inflated_bytes": ps["inflated_bytes"], "pieces": ps["n_pieces"], "wall_ms": t_all * 1e3,
                                 "device_bytes": ps["device_bytes"], "reads_taken": ps["n_device_reads"], "reads_left_to_the_host": ps["n_host_reads"],
                                 "summed_over_pieces_ms": {k[3:]: round(ps[k], 2) for k in ("ms_upload", "ms_inflate", "ms_frame", "ms_parse", "ms_consume")},
                                 "what": "ten copies of the sample GAM's bytes, one after the other -> the device front end's pipeline over the file's pieces "
                                         "(inflate, framing, protobuf walk, flatten, segment kernel); the file's bytes start in pageable host memory (PCIe inside "
                                         "the figure)"}
        except Exception as e:  # (a figure beside the metric: its failure is reported, not fatal)
            dev["device_gam"] = {"failed": repr(e)[:300]}
        df.close()
    from vgan_amd import _native as N
    return {"sample_reads": n, "gam_bytes": size, "decode_reads_per_s": n / t_dec, "flatten_reads_per_s": hb.n_reads / t_fl,
            "flatten_host_cpu_us_per_read": cpu_fl / max(hb.n_reads, 1) * 1e6, "device_flatten": dev,
            "threads": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1),
            # what the container may keep busy (affinity mask and cgroup CPU quota): a sample this size is a burst on the
            # quota's slack; a long input runs at ~7 us of CPU per read on this many processors
            "cpu_quota": int(N.lib().vgan_host_cpus())}


def end_to_end(graph, hc, args):
    """`vgan haplocart -g FILE` measured in-run: a synthetic GAM of args.e2e_reads 150 bp reads is written (not timed), then the C++ binary
    runs on it twice -- process start, HIP start-up, the device front end's pipeline over the file, posterior, output files and process
    exit inside the wall time -- and once through the host pipeline (VGAN_HC_DEVICE_GAM=0) beside it."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vgan_amd", "bin", "vgan")
    n = args.e2e_reads
    if n <= 0 or not os.path.exists(exe):
        return None
    rec = {"reads": n, "read_len": args.read_len}
    with tempfile.TemporaryDirectory(prefix="vgan_e2e_", dir="/tmp") as d:
        graph.write(d)
        t0 = time.perf_counter()
        CH = 1_000_000
        with open(d + "/r.gam", "wb") as f:  # (chunks of 1 M reads: BGZF files concatenate)
            for c0 in range(0, n, CH):
                a = hc.synth_reads(graph, min(CH, n - c0), seed=args.seed, read_len=args.read_len, first_read=c0)
                a.write_gam(d + "/part.gam")
                blob = open(d + "/part.gam", "rb").read()
                f.write(blob[:-28] if c0 + CH < n else blob)
                del a
        rec["gam_bytes"] = os.path.getsize(d + "/r.gam")
        rec["file_written_in_s"] = round(time.perf_counter() - t0, 1)
        cmd = [exe, "haplocart", "-g", d + "/r.gam", "--hc-files", d, "-q", "-t", "-1", "--keep-duplicates", "-o", d + "/o.tsv", "-pf", d + "/p.txt"]
        runs = []
        for tag, env in (("device", {"VGAN_HC_DEVICE_GAM": "1"}), ("device", {"VGAN_HC_DEVICE_GAM": "1"}), ("host", {"VGAN_HC_DEVICE_GAM": "0"})):
            t0, c0 = time.perf_counter(), os.times()
            r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, VGAN_TIMING="1", **env))
            dt, c1 = time.perf_counter() - t0, os.times()
            cpu = (c1.children_user - c0.children_user) + (c1.children_system - c0.children_system)
            line = [ln for ln in r.stderr.splitlines() if "device front end" in ln]
            runs.append({"front_end": tag, "rc": r.returncode, "wall_s": round(dt, 3), "reads_per_s": n / dt, "host_cpu_s": round(cpu, 2),
                         "result": open(d + "/o.tsv").read().splitlines()[-1].split("\t")[1:] if os.path.exists(d + "/o.tsv") else None,
                         "pipeline": line[0].split("front end: ")[1][:600] if line else None})
    dev_runs = [x for x in runs if x["front_end"] == "device" and x["rc"] == 0]
    rec["runs"] = runs
    if dev_runs:
        best = min(dev_runs, key=lambda x: x["wall_s"])
        rec["wall_s"], rec["reads_per_s"] = best["wall_s"], best["reads_per_s"]
    rec["what"] = ("`vgan haplocart -g FILE --keep-duplicates` on a synthetic BGZF GAM, the whole process timed from outside (start, HIP runtime, graph load, "
                   "device front end in pieces, segment kernel, posterior, output, exit); the better of two runs with the device front end, the host pipeline beside them")
    return rec


def ingest_rates(graph, hc, ctx, dev, args, n_batches=3, passes=8):
    """What it costs to get a batch to the kernel (the timed step replays a batch that is already in HBM; HaploCart touches each
    read once, HaploCart.cpp:408-421): (a) the packed batch over the link from pinned host memory, alone; (b) a streamed SINGLE
    PASS -- fresh batches, every read uploaded and accumulated exactly once: a copy stream fills one of two device buffers while
    the kernels of the batch before run on the compute stream (events order buffer reuse).  What crosses the link is what
    vgan_hc_accumulate_packed sends: rhdr + srec + crec (the kernel reads the quality bytes from the column records)."""
    import numpy as np
    import torch
    from vgan_amd import _native as N
    import ctypes as C
    host = []
    for b in range(n_batches):
        alns = hc.synth_reads(graph, args.reads, seed=args.seed, read_len=args.read_len, first_read=(b + 1) * args.reads)
        hb = hc.HostBatch(graph, alns, packed=True)
        if hb.c.n_reads:  # (reads outside the tile contract would take the general kernel: not this figure's business)
            return {"skipped": "the sample holds %d reads outside the packed layout" % hb.c.n_reads}
        pa, k = hb.packed_arrays(), hb.pk
        pinned = {n: torch.from_numpy(np.ascontiguousarray(pa[n].view(np.int32))).pin_memory() for n in ("rhdr", "srec", "crec")}
        meta = {f: getattr(k, f) for f in ("n_reads", "n_segments", "n_cols", "n_qual", "max_read_segs", "max_read_qual", "max_read_cols", "max_read_node_span")}
        host.append((pinned, meta))
        del hb, pa, alns
    cap = {n: max(h[0][n].numel() for h in host) for n in ("rhdr", "srec", "crec")}
    bufs = [{n: torch.empty(cap[n], dtype=torch.int32, device=dev) for n in cap} for _ in range(2)]
    n_bytes = sum(h[0][n].numel() * 4 for h in host for n in cap) / len(host)
    compute, copy = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)

    def view(buf, meta):
        v = N.HcPackedView()
        for f, x in meta.items():
            setattr(v, f, x)
        v.rhdr, v.srec, v.crec, v.qualp = buf["rhdr"].data_ptr(), buf["srec"].data_ptr(), buf["crec"].data_ptr(), None
        v.on_device, v.read_src = 1, None
        return v

    # (a) the link alone
    with torch.cuda.stream(copy):
        for n in cap:
            bufs[0][n][:host[0][0][n].numel()].copy_(host[0][0][n], non_blocking=True)
        copy.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            for n in cap:
                bufs[0][n][:host[0][0][n].numel()].copy_(host[0][0][n], non_blocking=True)
        copy.synchronize()
        h2d_s = (time.perf_counter() - t0) / 3
    # (b) the streamed single pass
    def run(n_pass):
        copied = [torch.cuda.Event() for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]
        reads = 0
        for i in range(n_pass):
            b = i & 1
            pinned, meta = host[i % len(host)]
            with torch.cuda.stream(copy):
                if i >= 2:
                    copy.wait_event(done[b])  # (the kernels that read this buffer two batches ago)
                for n in cap:
                    bufs[b][n][:pinned[n].numel()].copy_(pinned[n], non_blocking=True)
                copied[b].record(copy)
            compute.wait_event(copied[b])
            N.check(N.lib().vgan_hc_accumulate_packed(ctx._h, C.byref(view(bufs[b], meta))))
            done[b].record(compute)
            reads += meta["n_reads"]
        return reads
    ctx.reset()
    run(2)
    torch.cuda.synchronize()
    ctx.reset()
    t0 = time.perf_counter()
    reads = run(passes)
    ctx.finalize()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"link_bytes_per_read": n_bytes / args.reads, "h2d_GBps_pinned": n_bytes / h2d_s / 1e9, "h2d_ms_per_1M_reads": h2d_s * 1e3 * 1e6 / args.reads,
            "single_pass_reads_per_s": reads / el, "single_pass_batches": passes, "single_pass_reads": reads,
            "what": "fresh %d-read packed batches from pinned host memory, double-buffered: hipMemcpyAsync on a copy stream beside the "
                    "kernels of the batch before; every read crosses the link and is accumulated exactly once; final_vec at the end" % args.reads}


def main():
    args = parse()
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)  # does not return
    if args.path == "euka":
        return bench_euka(args)
    if args.path == "soibean":
        return bench_soibean(args)
    import numpy as np
    import torch
    import torch.distributed as dist
    from vgan_amd import distributed as vd
    from vgan_amd import haplocart as hc

    rank, world, local_rank, dev, backend, n_dev = pick_device(args)

    # ---- synthetic workload: one graph on every rank; the reads are ONE stream defined by the seed (read i depends on
    # (seed, i) only) and every rank takes a contiguous range of it, so the job's read set does not depend on N.
    #   weak   : rank r holds reads [r*R, (r+1)*R), R = --reads (N = 1 is BASELINE configs[1]: 1M x 150 bp)
    #   strong : the --reads-total reads of configs[2] (10M) are split into N contiguous shards
    graph = hc.synth_graph(seed=args.seed)
    if args.scaling == "strong":
        r0, r1 = vd.shard_bounds(args.reads_total, rank, world)
    else:
        r0, r1 = rank * args.reads, (rank + 1) * args.reads
    ctx = hc.HcContext(graph, device=local_rank)
    ctx.use_torch_stream()
    mode = {"node_weights": hc.MODE_NODE_WEIGHTS, "per_read": hc.MODE_PER_READ, "per_read_dense": hc.MODE_PER_READ_DENSE}[args.mode]
    ctx.set_mode(mode)
    batches, n_reads, n_seg, n_cols, algo_nw = [], 0, 0, 0, 0
    alns0 = None
    for c0 in range(r0, r1, CHUNK_READS):
        c1 = min(r1, c0 + CHUNK_READS)
        alns = hc.synth_reads(graph, c1 - c0, seed=args.seed, read_len=args.read_len, first_read=c0)
        # the batch as the front half hands it over: vgan_hc_flatten*_packed writes the layout the segment kernel streams
        # (byte moves, no arithmetic), the upload copies it as it is -- one copy in HBM, no layout pass on the device
        hb = hc.HostBatch(graph, alns, packed=True)
        db = hc.DeviceBatch(hb, dev)
        batches.append(db)
        n_reads += hb.n_reads
        n_seg += hb.n_segments
        n_cols += hb.c.n_cols + (hb.pk.n_cols if hb.pk is not None else 0)
        algo_nw += hb.algorithmic_bytes(graph.n_paths)["node_weights"]
        if alns0 is None:
            alns0 = alns  # the cpu_baseline leg samples the first reads of rank 0
        del hb
    final_dev = torch.zeros(graph.n_paths, dtype=torch.float64, device=dev)

    def step():
        ctx.reset()
        for db in batches:
            ctx.accumulate(db)
        ctx.finalize_device(final_dev)
        vd.reduce_loglik(final_dev, dst=0)  # RCCL over xGMI when N > 1: 41 KB of per-path sums

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # HIP events in the timed region around the segment kernel alone (the roofline's kernel: its average launch duration is measured
    # live, over these very steps); the other kernels' times come from a few more steps behind the clock -- a pair of events is ~8 us of
    # the stream's time, and three pairs a step were 4 % of it
    ctx.profile_enable(True, segment_only=(mode == hc.MODE_NODE_WEIGHTS))  # (the per-read modes' dominant kernel is the mask sweep: every kernel timed)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile_enable(True)
    n_extra = max(1, min(args.steps, 10))
    for _ in range(n_extra):
        step()
    fence()
    prof_all = ctx.profile_read()
    for k, v in prof_all.items():
        if k != "segment" and mode == hc.MODE_NODE_WEIGHTS:
            prof[k] = (v[0] * args.steps / n_extra, v[1] * args.steps // n_extra)  # (scaled to the timed steps: per-step figures below)
    ctx.profile_enable(False)
    elapsed = vd.all_reduce_max(elapsed, dev)            # MAX over ranks
    total_reads = vd.all_reduce_sum(float(n_reads), dev)  # whole-job reads per step
    final_host = final_dev.cpu().numpy().copy()           # complete on rank 0

    # ---- per-rank kernel times and the cost of the reduce alone (outside the timed region)
    my_kernels = {k: v[0] / max(args.steps, 1) for k, v in prof.items()}
    per_rank, reduce_ms = [my_kernels], None
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, {"rank": rank, "reads": n_reads, "kernel_ms_per_step": my_kernels})
        per_rank = gathered
        fence()
        t0 = time.perf_counter()
        for _ in range(20):
            vd.reduce_loglik(final_dev, dst=0)
        fence()
        reduce_ms = vd.all_reduce_max((time.perf_counter() - t0) / 20 * 1e3, dev)

    # ---- N > 1: who took part (is RCCL really running over N ranks on N devices?), and, beside the strong figure, the weak one
    # (the first 1M-read device batch of every rank: configs[1] per GPU)
    dist_info, weak_beside = None, None
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": torch.cuda.current_device(), "device_name": props.name,
                "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        rccl_version = None
        if backend == "nccl":
            try:
                rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception as e:  # noqa: BLE001 -- a version string must not end a benchmark
                rccl_version = "unknown (%r)" % (e,)
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl_version, "preflight": PREFLIGHT,
                     "distinct_devices": len({(d["device_index"], d["pci_bus_id"], d["uuid"]) for d in everyone}), "ranks": everyone}
        if args.scaling == "strong":
            def weak_step():
                ctx.reset()
                ctx.accumulate(batches[0])
                ctx.finalize_device(final_dev)
                vd.reduce_loglik(final_dev, dst=0)
            for _ in range(args.warmup):
                weak_step()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                weak_step()
            fence()
            w_el = vd.all_reduce_max(time.perf_counter() - t0, dev)
            w_reads = vd.all_reduce_sum(float(batches[0].n_reads), dev)
            weak_beside = {"scaling": "weak", "reads_per_gpu": batches[0].n_reads, "value": w_reads * args.steps / w_el,
                           "unit": "reads/s", "ms_per_step": w_el / args.steps * 1e3}

    # ---- a9 (get_posterior.cpp:87-127) on rank 0, timed on its own: host tree walk + upload + one kernel
    posterior_ms, posterior_top = None, None
    if rank == 0:
        pred = graph.path_names[ctx.argmax(final_host)]
        ctx.posterior(final_host, pred)
        t0 = time.perf_counter()
        for _ in range(10):
            post = ctx.posterior(final_host, pred)
        posterior_ms = (time.perf_counter() - t0) / 10 * 1e3
        posterior_top = {"predicted": pred, "confidence": post[0][1], "records": len(post)}

    # ---- the reference's loop order (one mask row per segment) measured beside the default mode: the kernel
    # BASELINE.json's north_star puts the >= 30 % HBM-roofline target on.  Outside the timed region.
    per_read = None
    if rank == 0 and mode == hc.MODE_NODE_WEIGHTS and not args.no_extra:
        per_read = {}
        db0 = batches[0]
        row_bytes = 8 * ((graph.n_paths + 63) // 64) + 4 + 8
        for label, m in (("dense", hc.MODE_PER_READ_DENSE), ("skip_unset_tiles", hc.MODE_PER_READ)):
            ctx.set_mode(m)
            ctx.reset()
            ctx.accumulate(db0)
            ctx.synchronize()
            ctx.profile_enable(True)
            for _ in range(3):
                ctx.reset()
                ctx.accumulate(db0)
            pr = ctx.profile_read()
            ctx.profile_enable(False)
            ms = pr["sweep_segments"][0] / max(pr["sweep_segments"][1], 1)
            gbs = db0.n_segments * row_bytes / (ms * 1e-3) / 1e9
            per_read[label] = {"kernel": "hc_sweep_kernel", "debug": label == "dense",  # (dense: every tile swept, a debug variant -- DESIGN.md 4.3)
                               "avg_launch_ms": ms, "achieved": gbs, "unit": "GB/s",
                               "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": db0.n_segments * row_bytes,
                               "reads_per_s_kernel_only": db0.n_reads / (ms * 1e-3)}
        ctx.set_mode(mode)

    if rank == 0:
        # dominant kernel and its algorithmic bytes per launch (DESIGN.md "Roofline accounting"); a rank launches it once
        # per device batch, so bytes per launch = the rank's bytes / its batches
        if mode == hc.MODE_NODE_WEIGHTS:
            kname, kbytes = "segment", algo_nw / len(batches)
        else:
            kname, kbytes = "sweep_segments", n_seg * (8 * ((graph.n_paths + 63) // 64) + 4 + 8) / len(batches)
        k_ms, k_n = prof[kname]
        avg_ms = k_ms / max(k_n, 1)
        kernel_name = {"segment": "hc_segment", "sweep_segments": "hc_sweep_kernel"}[kname]
        traffic, issue = None, None
        if world == 1 and not args.no_pmc and args.scaling == "weak":
            ksub = kernel_name if kname == "segment" else "hc_sweep_kernel<10, false>"
            traffic = collect_traffic(args, ksub)
            issue = collect_issue(args, ksub, avg_ms)
        achieved = kbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # the same launch on SURVEY 8d's literal formula for the collapsed form, B_read' = 2 L + 16 M (L columns, M mappings per read)
        survey_bytes = (2.0 * n_cols + 16.0 * n_seg) / len(batches)
        survey_gbs = survey_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        what = ("%d of the %d reads of BASELINE configs[2] per GPU (contiguous shards)" % (n_reads, args.reads_total)
                if args.scaling == "strong" else "%d synthetic %dbp reads per GPU" % (args.reads, args.read_len))
        out = {
            "metric": "reads/sec (whole node) through HaploCart posterior path, 150bp",
            "value": total_reads * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "HaploCart %s vs hcfiles-shaped mtDNA graph (11821 nodes, 5179 paths); step = reset + per-read "
                                   "likelihood kernels over the flattened batch resident in HBM in the layout the host flatten step "
                                   "writes (vgan_hc_flatten_packed: one copy, no device-side layout pass) + final_vec%s; GAM decode / "
                                   "flatten (front_end) and get_posterior (posterior_ms) are timed beside it, not inside" % (
                                       what, " + RCCL reduce" if world > 1 else ""),
                       "reads_per_gpu": n_reads, "reads_total": int(total_reads), "segments_per_read": n_seg / max(n_reads, 1),
                       "mode": args.mode, "device_batches_per_gpu": len(batches),
                       "sharding": "contiguous read ranges x%d, %s reduce of final_vec[%d] to rank 0" % (world, "RCCL" if backend == "nccl" else "gloo (host staged)", graph.n_paths)
                                   if world > 1 else "single GPU",
                       "physical_gpus": min(n_dev, world), "dist_backend": backend if world > 1 else None},
            # bound / achieved / peak / frac are the HBM figures BASELINE.json asks for (algorithmic bytes over the kernel's
            # time against the 8 TB/s peak); what actually limits the launch is named beside them, from a committed profile
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic.get("bytes") if traffic else None, "traffic_detail": traffic,
                         "kernel": kernel_name, "algorithmic_bytes_per_launch": kbytes, "avg_launch_ms": avg_ms, "launches": k_n,
                         # both byte bases, so that no reader has to recompute: `frac` is on the stricter one (the input bytes of the
                         # SoA form: 2 per column + the quality string + 8 per mapping + 15 per read)
                         "byte_bases": {"input_bytes_per_read": kbytes * len(batches) / max(n_reads, 1), "frac_on_input_bytes": achieved / HBM_PEAK_GBS,
                                        "survey_8d_2L_16M_bytes_per_read": survey_bytes * len(batches) / max(n_reads, 1),
                                        "frac_on_survey_8d": survey_gbs / HBM_PEAK_GBS} if mode == hc.MODE_NODE_WEIGHTS else None,
                         # what limits the launch: vector-instruction issue and LDS, from an in-run SQ pass
                         "alu": issue, "limiter": issue},
            "kernel_ms_per_step": my_kernels,
            "per_rank": per_rank if world > 1 else None,
            "dist": dist_info,
            "weak_beside": weak_beside,
            "reduce_ms": reduce_ms,
            "posterior_ms": posterior_ms,
            "posterior": posterior_top,
            "result_check": {"argmax": int(np.argmax(final_host)), "max_final_vec": float(final_host.max()),
                             "sum_final_vec": float(final_host.sum())},
            "parity": None,
            "per_read_kernel": per_read,
        }
        if world == 1 and not args.no_frontend:
            out["front_end"] = front_end_rates(graph, hc, args.seed, ctx=ctx)
        if world == 1 and not args.no_ingest and mode == hc.MODE_NODE_WEIGHTS:
            batches.clear()  # (the resident batch's HBM is not needed any more)
            out["ingest"] = ingest_rates(graph, hc, ctx, dev, args)
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"], out["parity"] = cpu_baseline(graph, alns0, args.cpu_seconds, ctx, hc)
        if world == 1 and not args.no_frontend and args.e2e_reads > 0 and mode == hc.MODE_NODE_WEIGHTS and args.path == "haplocart":
            try:
                del ctx  # (the binary brings its own contexts up: this process's device memory goes first)
                out["end_to_end"] = end_to_end(graph, hc, args)
            except Exception as e:  # (a figure beside the metric: its failure is reported, not fatal)
                out["end_to_end"] = {"failed": repr(e)[:300]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
