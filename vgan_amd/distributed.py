"""Read sharding across GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests).  The hot path has no exchange step: reads shard embarrassingly, every
rank accumulates its own final_vec[P]; the only collective is one sum-reduce of those P doubles to rank 0
(41 KB for the hcfiles shape: latency bound, one call per job)."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, device=None):
    """Initialise the default process group from the torchrun environment; no-op for world size 1."""
    import torch
    import torch.distributed as dist
    rank, world, local_rank = env_rank()
    if world <= 1 or dist.is_initialized():
        return rank, world, local_rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = device if device is not None else torch.device("cuda", local_rank)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shard_bounds(n_reads, rank, world):
    """Contiguous, near-equal read range of this rank (the reference's OpenMP loop is order independent up to
    floating-point summation order, src/HaploCart.cpp:408-421)."""
    return n_reads * rank // world, n_reads * (rank + 1) // world


def reduce_loglik(final_vec, dst=0):
    """Sum the per-rank final_vec tensors (float64[P]) onto rank dst.  Returns the tensor (complete on dst only)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if final_vec.is_cuda and dist.get_backend() == "gloo":  # test rigs without RCCL: stage through the host
            host = final_vec.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            final_vec.copy_(host)
        else:
            dist.reduce(final_vec, dst=dst, op=dist.ReduceOp.SUM)
    return final_vec


def all_reduce_max(value, device=None):
    """max over ranks of a python float (bench timing)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return value
    on = device if (device is not None and dist.get_backend() != "gloo") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return value
    on = device if (device is not None and dist.get_backend() != "gloo") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def preflight(device=None, timeout_s=60.0):
    """First thing after init() on N > 1 ranks: one int64 and one float64 all-reduce on the tensors the job will use (device
    tensors under RCCL, host tensors under gloo), checked against the closed forms, under a watchdog -- a rank that is still inside
    a collective after timeout_s says so on stderr (rank, backend, device, what it was waiting in) and leaves with exit code 3, so
    that a first run on a new machine fails in a minute with a reason instead of hanging.  Returns the record for the bench line."""
    import sys
    import threading
    import time
    import torch
    import torch.distributed as dist
    rank, world, local_rank = env_rank()
    if not (dist.is_available() and dist.is_initialized()) or world <= 1:
        return {"world_size": 1, "ok": True, "skipped": "one rank"}
    backend = dist.get_backend()
    on = device if (device is not None and backend != "gloo") else "cpu"
    stage = {"at": "start"}
    done = threading.Event()

    def watchdog():
        if not done.wait(timeout_s):
            sys.stderr.write("[vgan preflight] rank %d of %d (backend %s, tensors on %s, local rank %d): still in '%s' after %.0f s -- giving up\n"
                             % (rank, world, backend, on, local_rank, stage["at"], timeout_s))
            sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    t0 = time.perf_counter()
    try:
        stage["at"] = "all_reduce(int64)"
        ti = torch.tensor([rank + 1, (rank + 1) * (1 << 40)], dtype=torch.int64, device=on)
        dist.all_reduce(ti, op=dist.ReduceOp.SUM)
        stage["at"] = "all_reduce(float64)"
        tf = torch.tensor([0.5 * (rank + 1)], dtype=torch.float64, device=on)
        dist.all_reduce(tf, op=dist.ReduceOp.SUM)
        stage["at"] = "reduce(float64 -> rank 0)"
        tr = torch.full((8,), float(rank + 1), dtype=torch.float64, device=on)
        dist.reduce(tr, dst=0, op=dist.ReduceOp.SUM)
        if on != "cpu":
            stage["at"] = "device synchronize"
            torch.cuda.synchronize(on)
        tri = world * (world + 1) // 2
        got_i, got_f, got_r = [int(x) for x in ti.cpu()], float(tf.cpu()[0]), float(tr.cpu()[0])
        ok = got_i == [tri, tri * (1 << 40)] and got_f == 0.5 * tri and (rank != 0 or got_r == float(tri))
    except Exception as e:  # noqa: BLE001 -- whatever the backend raises is the message
        done.set()
        sys.stderr.write("[vgan preflight] rank %d of %d (backend %s, tensors on %s): %s failed: %r\n" % (rank, world, backend, on, stage["at"], e))
        sys.stderr.flush()
        raise SystemExit(3)
    done.set()
    rec = {"world_size": world, "backend": backend, "tensors_on": str(on), "ok": bool(ok), "seconds": time.perf_counter() - t0,
           "int64_sum": got_i, "float64_sum": got_f, "timeout_s": timeout_s}
    if not ok:
        sys.stderr.write("[vgan preflight] rank %d: collectives returned %r / %r, expected %r / %r\n" % (rank, got_i, got_f, [tri, tri * (1 << 40)], 0.5 * tri))
        raise SystemExit(3)
    return rec


def all_gather_words(words, device=None):
    """ONE collective for a small record of 64-bit words (ints; a float goes in through its bit pattern): every rank gets every
    rank's words, in rank order -- what it makes of them (an exact integer sum, a float sum in a fixed order) is then the same
    on every rank.  Returns a list of per-rank lists of python ints (unsigned 64-bit)."""
    import torch
    import torch.distributed as dist
    vals = [int(w) & ((1 << 64) - 1) for w in words]
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [vals]
    world = dist.get_world_size()
    on = device if (device is not None and dist.get_backend() != "gloo") else "cpu"
    signed = [v - (1 << 64) if v >= (1 << 63) else v for v in vals]
    mine = torch.tensor(signed, dtype=torch.int64, device=on)
    out = torch.empty(world * len(vals), dtype=torch.int64, device=on)
    dist.all_gather_into_tensor(out, mine)
    flat = [int(x) & ((1 << 64) - 1) for x in out.cpu().tolist()]
    return [flat[r * len(vals):(r + 1) * len(vals)] for r in range(world)]
