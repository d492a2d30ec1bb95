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


def reduce_rank_stats(dist, value, device=None, reps=20):
    """(min, max) of a per-rank scalar (e.g. each rank's average kernel time) over the group, plus the measured time
    of one such tiny all-reduce (seconds, mean of `reps` after a warm-up) - the only collective the timed region's
    bracket uses, reported so that its cost can be seen next to the kernel time."""
    if dist is None:
        return float(value), float(value), 0.0
    import time

    import torch

    kw = {} if device is None else {"device": device}
    lo = torch.tensor([float(value)], dtype=torch.float64, **kw)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    t = torch.zeros(1, dtype=torch.float64, **kw)
    on_gpu = device is not None and str(device).startswith("cuda")
    if on_gpu:
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if on_gpu:
        torch.cuda.synchronize()
    return float(lo.item()), float(hi.item()), (time.perf_counter() - t0) / reps


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
