"""Multi-GPU layout of the path: independent engines / streams are dealt out to
ranks (one process per GPU) and never exchange data -- channels do not mix
(brutefir/brutefir.cpp:252-334), so there is no data-path collective.  The only
communication is the timing reduction bench.py needs (MAX of the elapsed time)."""


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) share of n_items for `rank` of `world`
    (the first n_items % world ranks get one more)."""
    if world < 1 or not 0 <= rank < world or n_items < 0:
        raise ValueError("bad shard request")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_sizes(n_items, world):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def max_over_ranks(value, device=None):
    """MAX of a python float over all ranks (identity without a process group)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
