"""Multi-GPU plumbing of the batched balance controller.

The path shards trivially: every robot is independent, so rank r of G owns a
contiguous slice of the batch axis and there is NO data-path collective.
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests) is used only for the barrier around the timed region and to
reduce a handful of counters.
"""
from __future__ import annotations


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of a batch of n robots owned by `rank`."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_counters(dist, wall_s, solved, robots, device=None):
    """(max wall time over ranks, total solved, total robots).  `dist` is
    torch.distributed (initialised) or None for a single process."""
    if dist is None:
        return float(wall_s), int(solved), int(robots)
    import torch

    kw = {} if device is None else {"device": device}
    t = torch.tensor([float(wall_s)], dtype=torch.float64, **kw)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([int(solved), int(robots)], dtype=torch.int64, **kw)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c[0].item()), int(c[1].item())


def gather_results(dist, grf_shard):
    """Optional result collection: all-gather of the per-rank [n_r, 12] GRF blocks, outside the timed
    hot path.  Returns (gathered tensor [sum n_r, 12] in rank order, seconds).  Shards may be uneven (shard_bounds
    gives the first n % world ranks one robot more): the sizes are exchanged first and the blocks are gathered into
    a padded buffer, then trimmed.  RCCL over xGMI when the tensor lives on a GPU and the backend is "nccl"; the CPU
    tests run it over gloo."""
    import time

    import torch

    world = dist.get_world_size()
    dev = grf_shard.device
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    sizes[dist.get_rank()] = grf_shard.shape[0]
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM)
    sizes = [int(v) for v in sizes.tolist()]
    pad = max(sizes)
    cols = grf_shard.shape[1]
    mine = grf_shard.contiguous()
    if mine.shape[0] != pad:
        mine = torch.cat([mine, torch.zeros((pad - mine.shape[0], cols), dtype=mine.dtype, device=dev)])
    out = torch.empty((world * pad, cols), dtype=grf_shard.dtype, device=dev)
    if grf_shard.is_cuda:
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    dist.all_gather_into_tensor(out, mine)
    if grf_shard.is_cuda:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if any(sz != pad for sz in sizes):
        out = torch.cat([out[r * pad:r * pad + sizes[r]] for r in range(world)])
    return out, dt
