"""Image-batch sharding across the GPUs of one node.

Every image is filtered independently, so multi-GPU operation is pure data
parallelism: one process per GPU, each takes a contiguous slice of the batch, and there is no
data-path collective at all (no RCCL traffic over xGMI).  torch.distributed is used only for
the host-side barrier and for gathering per-rank (pixels, seconds) when measuring - two scalars -
and does that over gloo: no RCCL communicator is brought up unless one is asked for
(``init_distributed(backend="nccl")`` or ``RF_DIST_BACKEND=nccl``; SURVEY.md 5: "do not add RCCL").
"""


def shard_range(n_items, world_size, rank):
    """Contiguous [begin, end) slice of ``n_items`` owned by ``rank``; remainders go to the
    low ranks, so slice sizes differ by at most one."""
    if world_size < 1 or not 0 <= rank < world_size or n_items < 0:
        raise ValueError("bad shard request n=%r world=%r rank=%r" % (n_items, world_size, rank))
    base, extra = divmod(n_items, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_sizes(n_items, world_size):
    return [shard_range(n_items, world_size, r)[1] - shard_range(n_items, world_size, r)[0]
            for r in range(world_size)]


def init_distributed(backend=None):
    """Join the torchrun rendezvous if one is described by the environment.
    Returns (rank, world_size, local_rank)."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend is None:
                backend = os.environ.get("RF_DIST_BACKEND")
            if backend is None:
                backend = "gloo"  # host-side barrier + two scalar reductions: nothing for xGMI to carry
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def barrier(world_size):
    if world_size > 1:
        import torch
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def reduce_job(local_units, local_seconds, world_size, device=None):
    """Whole-job view of a timed region: (sum of units over ranks, max of seconds over ranks)."""
    if world_size == 1:
        return float(local_units), float(local_seconds)
    import torch
    import torch.distributed as dist
    dev = device if device is not None and dist.get_backend() != "gloo" else "cpu"
    units = torch.tensor([float(local_units)], dtype=torch.float64, device=dev)
    secs = torch.tensor([float(local_seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(units, op=dist.ReduceOp.SUM)
    dist.all_reduce(secs, op=dist.ReduceOp.MAX)
    return float(units.item()), float(secs.item())


def gather_scalars(local_value, world_size, device=None):
    """Every rank's value of one per-rank measurement, in rank order, on every rank (a slow GPU must
    be visible behind the job's max: bench.py's `per_rank_ms`)."""
    if world_size == 1:
        return [float(local_value)]
    import torch
    import torch.distributed as dist
    dev = device if device is not None and dist.get_backend() != "gloo" else "cpu"
    mine = torch.tensor([float(local_value)], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(world_size)]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]
