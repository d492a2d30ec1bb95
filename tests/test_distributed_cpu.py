"""CPU, world_size 2, gloo: the N>1 plumbing of bench.py (shard ownership,
max-over-ranks timing, counter reduction).  No GPU, no data-path collective."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import c_oracle, numpy_restatement as R  # checker standing in for the GPU on CPU
    from quadruped_control_amd import workloads as W
    from quadruped_control_amd.sharding import reduce_counters, shard_bounds

    lo, hi = shard_bounds(n, rank, world)
    batch = W.config5(n=hi - lo, start=lo)
    grf, status, _ = c_oracle.control_batch(R.cheetah_params(0.6), batch)
    wall, solved, robots = reduce_counters(dist, 1.0 + rank, int((status == 0).sum()), hi - lo)
    from quadruped_control_amd.sharding import reduce_rank_stats

    k_min, k_max, allreduce_s = reduce_rank_stats(dist, 10.0 + rank, reps=5)  # per-rank kernel time -> min / max over the group
    assert (k_min, k_max) == (10.0, 9.0 + world) and allreduce_s > 0.0
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, grf))
    dist.barrier()
    if rank == 0:
        q.put((wall, solved, robots, gathered))
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    n, world = 301, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    wall, solved, robots, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert wall == 2.0 and robots == n and solved == n
    gathered.sort(key=lambda t: t[0])
    assert gathered[0][0] == 0 and gathered[0][1] == gathered[1][0] and gathered[1][1] == n
    from oracle import c_oracle, numpy_restatement as R
    from quadruped_control_amd import workloads as W

    full, st, _ = c_oracle.control_batch(R.cheetah_params(0.6), W.config5(n=n, start=0))
    np.testing.assert_array_equal(np.concatenate([g[2] for g in gathered]), full)


def _gather_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quadruped_control_amd.sharding import gather_results

    from quadruped_control_amd.sharding import shard_bounds

    lo, hi = shard_bounds(n * world + (1 if n % 2 else 0), rank, world)  # odd n: uneven shards (rank 0 holds one robot more)
    m = hi - lo
    shard = torch.arange(m * 12, dtype=torch.float64).reshape(m, 12) + 1000.0 * rank
    out, secs = gather_results(dist, shard)
    if rank == 0:
        q.put((out.numpy(), secs))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("n", [64, 33])
def test_result_gather_gloo(n):
    """SURVEY 8e optional result collection: per-rank GRF blocks all-gathered in rank order; n = 33 gives the
    uneven shards shard_bounds() produces when the batch does not divide (34 + 33 robots)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, secs = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n0 = n + (1 if n % 2 else 0)
    b0 = np.arange(n0 * 12, dtype=np.float64).reshape(n0, 12)
    b1 = np.arange(n * 12, dtype=np.float64).reshape(n, 12) + 1000.0
    np.testing.assert_array_equal(out, np.concatenate([b0, b1]))
    assert secs >= 0.0
