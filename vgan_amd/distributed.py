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
