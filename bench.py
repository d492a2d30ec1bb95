#!/usr/bin/env python
"""Headline benchmark: friction-cone QPs/sec of the batched balance controller.

  python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5] [--n ROBOTS]

A "step" is one pass of the hot path (one qc_control_batch launch) over one
batch of synthetic robots that is already resident in HBM.  Default workload =
BASELINE.json configs[1]: 4096 randomised COM poses/velocities, all four feet
in contact, mu = 0.6 (SURVEY.md 8d "config 2").  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank owns its own shard of the
batch axis (weak scaling: `n` robots per GPU); the path has no data exchange,
RCCL is used only for the barrier and to reduce the timing / solved counters.

Prints ONE JSON line on rank 0 (see the task contract): value = robots solved
per second over all ranks; roofline = algorithmic bytes (488 B per robot:
388 B read + 100 B written) / average kernel time measured with
HIP events on the launch stream; cpu_baseline = the C oracle (a port, not
qpOASES) timed on the host cores on the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_ROBOT_COLD = 488   # 48 f64 + 4 B stance read; 12 f64 + 4 B status written
BYTES_PER_ROBOT_WARM = 496   # + 4 B warm word read + 4 B active-set word written
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)

CONFIG_N = {2: 4096, 3: 65536, 4: 262144, 5: 262144}  # robots per GPU
CONFIG_DESC = {
    2: "config2: batch of {n} randomised COM poses/velocities per GPU, all 4 feet in contact, mu=0.6 pyramid cone, cold start",
    3: "config3: batch of {n} per GPU, mixed 2/3/4-foot contact states from trot/walk gait schedules, cold start",
    4: "config4: batch of {n} per GPU, tick 1 warm-started from tick 0's active set (dt=1/300 s)",
    5: "config5: 2,097,152-robot batch (config-3 distribution) sharded contiguously, {n} per GPU",
}


def make_batch(cfg, n, start):
    from quadruped_control_amd import workloads as W

    if cfg == 2:
        return W.config2(n, start=start), None
    if cfg == 3:
        return W.config3(n, start=start), None
    if cfg == 4:
        t0, t1 = W.config4(n, start=start)
        return t1, t0
    return W.config5(n, start=start), None


def time_steps(ctl, dev_batch, warm, out, steps, warmup, dist=None):
    """W untimed steps, then K timed steps bracketed by barrier + synchronize.
    Returns (wall seconds for K steps, HIP-event seconds for K steps)."""
    import torch

    launch, _ = ctl.plan_batch(dev_batch, warm=warm, out=out)  # arguments marshalled once; launch() = one C call
    for _ in range(warmup):
        launch()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # torch's current stream == the stream control_batch launches on
    for _ in range(steps):
        launch()
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0  # this rank's K steps; the caller takes the MAX over ranks
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    return wall, ev0.elapsed_time(ev1) * 1e-3


def run_config(ctl, q, cfg, n, start, steps, warmup, dist=None, device=0, fused=False):
    import torch

    batch, prev = make_batch(cfg, n, start)
    if fused:  # SURVEY 8f rows 1+2: joint angles in (device FK), joint torques out (J^T f, clamped)
        from quadruped_control_amd import workloads as W

        batch = W.with_joint_angles(batch, start=start)
        if fused == "full":  # rows 3+4 too: contact state from gait phases, swing planner/trajectories/IK/PD
            import numpy as np

            batch = W.with_swing_references(batch, start=start)
            idx = np.arange(start, start + n, dtype=np.uint64)
            phase = np.fmod(np.array([0.0, 0.5, 0.5, 0.0])[None] + W.uniform(0x5EED0009, idx, 3)[:, None], 1.0)
            batch = {k: v for k, v in batch.items() if k not in ("stance", "swing_pos", "swing_vel")}
            batch["gait_phase"] = np.ascontiguousarray(phase)
    dev_batch = q.to_device(batch, device)
    if fused == "full":
        dev_batch["swing_state"] = torch.from_numpy(q.new_swing_states(n).view("uint8").reshape(-1).copy()).to(f"cuda:{device}")
    warm = None
    if prev is not None:  # config 4: tick 0 (cold) produces the warm-start words for tick 1
        o0 = ctl.control_batch(q.to_device(prev, device), want_active_set=True)
        torch.cuda.synchronize()
        warm = o0["active_set"]
    out = {"grf_body": torch.empty((n, 12), dtype=torch.float64, device=f"cuda:{device}"),
           "status": torch.empty((n,), dtype=torch.int32, device=f"cuda:{device}")}
    if warm is not None:
        out["active_set"] = torch.empty((n,), dtype=torch.int32, device=f"cuda:{device}")
    if fused:
        out["joint_tau"] = torch.empty((n, 12), dtype=torch.float64, device=f"cuda:{device}")
    wall, evs = time_steps(ctl, dev_batch, warm, out, steps, warmup, dist)
    solved = int((out["status"] == 0).sum().item())
    return dict(wall=wall, event_s=evs, solved=solved, n=n, batch=batch, warm=warm is not None, out=out)


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(P, batch, budget_s=4.0):
    """C oracle (oracle/balance_oracle.c: a port of the reference path, NOT
    qpOASES) on the host cores over the same robots; bounded to a few seconds
    of wall time (~10-30 s of CPU work on >= 4 threads)."""
    from oracle import c_oracle

    import numpy as np

    n_all = batch["x"].shape[0]
    threads = host_cores()
    base = min(n_all, 4096)
    tile = max(1, (256 * threads + base - 1) // base)  # >= 256 robots per thread per call
    sample = {k: np.ascontiguousarray(np.tile(v[:base], (tile, 1))) for k, v in batch.items()}
    n = base * tile
    c_oracle.control_batch(P, {k: v[:64] for k, v in sample.items()}, threads=threads)  # warm-up / thread pool
    t0 = time.perf_counter()
    c_oracle.control_batch(P, {k: v[:512] for k, v in sample.items()}, threads=1)
    one = 512 / (time.perf_counter() - t0)
    reps, t0 = 0, time.perf_counter()
    while True:
        c_oracle.control_batch(P, sample, threads=threads)
        reps += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": reps * n / dt, "unit": "QPs/s", "cores": threads, "kind": "port",
            "sample": f"first {base} robots of the benchmark batch tiled x{tile} = {n} robots per call x {reps} calls, {threads} OpenMP threads, "
                      f"{dt:.1f} s wall; C restatement (textbook primal active set), not qpOASES",
            "single_thread_value": one}


def batch_load_probe(q, P, device, nb=2097152, steps=10):
    """north_star: "achieved HBM GB/s on the batch load".  Same kernel with the solver iterations skipped
    (QC_PROBE_BATCH_LOAD: load -> PD law / rotation log / Newton-Euler rhs -> output transform -> store;
    status = max_iter for every robot) on a batch four times the 256 MiB Infinity Cache.  The rows are
    262,144 config-5 robots tiled 8x on the device (their values do not matter without iterations)."""
    import torch

    from quadruped_control_amd import workloads as W

    os.environ["QC_PROBE_BATCH_LOAD"] = "1"
    try:
        probe = q.BalanceController.from_params(P, device=device)
    finally:
        del os.environ["QC_PROBE_BATCH_LOAD"]
    base = q.to_device(W.config5(nb // 8), device)
    batch = {k: v.repeat(8, 1).contiguous() for k, v in base.items()}
    out = {"grf_body": torch.empty((nb, 12), dtype=torch.float64, device=f"cuda:{device}"),
           "status": torch.empty((nb,), dtype=torch.int32, device=f"cuda:{device}")}
    _, evs = time_steps(probe, batch, None, out, steps, 2)
    gbs = BYTES_PER_ROBOT_COLD * nb * steps / evs / 1e9
    return {"robots": nb, "bytes": BYTES_PER_ROBOT_COLD * nb, "us": evs / steps * 1e6, "achieved": gbs, "unit": "GB/s",
            "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
            "what": "load -> assemble (PD law, rotation log, Newton-Euler rhs) -> output transform -> store, no QP iterations"}


def host_boundary(ctl, q):
    """The host-pointer entry points (PCIe staging included; never what `value` reports):
    config 1 = one robot through the reference-signature control() (qc_control), and configs 2/4 through
    qc_control_batch_host (pageable numpy arrays in, numpy arrays out)."""
    from quadruped_control_amd import workloads

    b1 = workloads.config1()
    names = ctl.leg_names
    foot_map = {nm: b1["feet"].reshape(-1, 4, 3)[0, i] for i, nm in enumerate(names)}
    a = (b1["Rwb"][0], b1["Rwb_d"][0], b1["x"][0], b1["xdot"][0], b1["w"][0], b1["x_d"][0], b1["xdot_d"][0], b1["w_d"][0])
    for _ in range(20):
        f = ctl.control(*a, foot_map)
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        f = ctl.control(*a, foot_map)
    lat = (time.perf_counter() - t0) / reps
    _, stance, _, ptrs = ctl._one  # the same record through the bare C entry point
    t0 = time.perf_counter()
    for _ in range(reps):
        ctl._lib.qc_control(ctl._h, *ptrs)
    lat_c = (time.perf_counter() - t0) / reps
    res = {"config1_control_latency_us": lat * 1e6, "config1_qc_control_latency_us": lat_c * 1e6,
           "config1_grf_RL": [float(v) for v in f[names[0]]]}
    exe = os.path.join(ROOT, "tests", "cpp", "adapter_test")  # built by __graft_entry__.build()
    if os.path.exists(exe):
        import subprocess

        try:
            out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
            res["config1_cpp_adapter_latency_us"] = float(out.split("latency_us")[1].split()[0])
        except Exception:
            pass
    for cfg, n in ((2, CONFIG_N[2]), (4, CONFIG_N[4])):
        hb, _ = make_batch(cfg, n, 0)
        ctl.control_batch_host(hb)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ctl.control_batch_host(hb)
        res[f"config{cfg}_host_QPs_per_s"] = n * reps / (time.perf_counter() - t0)
    return res


def pmc_traffic(cfg, n):
    """HBM bytes per launch from the newest committed PMC pass of this workload
    (profiles/rNN_cfg<cfg>.json, produced by tools/profile_r.sh +
    tools/summarize_profile.py: separate --pmc FETCH_SIZE / WRITE_SIZE passes,
    FETCH_SIZE doubled per MI355X_MICROARCH.md).  None if no matching pass."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_cfg{cfg}.json"))):
        try:
            d = json.load(open(f))
            if d.get("bench_line", {}).get("config", {}).get("robots_per_gpu") == n and "traffic" in d:
                best = (d["traffic"]["hbm_bytes_per_launch"], os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--n", type=int, default=0, help="robots per GPU (default: the config's size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the informational runs of the other configs")
    ap.add_argument("--gather-results", action="store_true",
                    help="N > 1: after the timed region, all-gather the per-rank GRF blocks (result collection over xGMI, "
                         "SURVEY 8e) and report the collective's time separately")
    ap.add_argument("--probe-batch-load", action="store_true",
                    help="time the load -> assemble -> store phase alone (no solver iterations) on 2,097,152 robots "
                         "(done by default together with the sweep)")
    args = ap.parse_args()

    import torch

    import quadruped_control_amd as q

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # QC_BENCH_ONE_DEVICE=1 (test hook): all ranks share cuda:0 and talk over gloo, so the
        # multi-process path can be exercised on a 1-GPU box; the real run is one rank per GPU over RCCL.
        one_dev = os.environ.get("QC_BENCH_ONE_DEVICE") == "1"
        if one_dev:
            local_rank = 0
            dist_mod.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod
    device = local_rank if world > 1 else 0
    torch.cuda.set_device(device)

    P = q.cheetah_params(mu=0.6)
    ctl = q.BalanceController.from_params(P, device=device)
    cfg = args.config
    n = args.n or CONFIG_N[cfg]
    res = run_config(ctl, q, cfg, n, rank * n, args.steps, args.warmup, dist, device)

    from quadruped_control_amd.sharding import reduce_counters

    on_gpu = dist is not None and dist.get_backend() == "nccl"
    wall, solved_total, total_robots = reduce_counters(dist, res["wall"], res["solved"], n,
                                                       device=f"cuda:{device}" if on_gpu else None)

    gather_s = None
    if dist is not None and args.gather_results:
        from quadruped_control_amd.sharding import gather_results

        shard = res["out"]["grf_body"] if on_gpu else res["out"]["grf_body"].cpu()
        gathered, gather_s = gather_results(dist, shard)
        assert gathered.shape[0] == total_robots

    if rank == 0:
        bytes_per = BYTES_PER_ROBOT_WARM if res["warm"] else BYTES_PER_ROBOT_COLD
        kernel_s = res["event_s"] / args.steps
        achieved = bytes_per * n / kernel_s / 1e9
        line = {
            "metric": "friction-cone QPs/sec (12 vars, 4-foot stance) at 1/2/4/8 MI355X",
            "value": total_robots * args.steps / wall,
            "unit": "QPs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": CONFIG_DESC[cfg].format(n=n), "robots_per_gpu": n, "global_batch": total_robots,
                       "kernel": ctl.kernel_name, "parallelism": f"batch-shard x{world} (no data-path collective)"},
            "solved_fraction": solved_total / total_robots,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "qc::balance_kernel", "bytes_per_launch": bytes_per * n,
                         "avg_kernel_us": kernel_s * 1e6,
                         "note": "latency/FP64-VALU bound active-set solve; traffic from the PMC pass is in profiles/"},
        }
        if gather_s is not None:
            line["result_gather"] = {"bytes_per_rank": n * 96, "seconds": gather_s, "GBs_per_rank": n * 96 * (world - 1) / gather_s / 1e9,
                                     "what": "all_gather_into_tensor of the [n, 12] GRF blocks after the timed region (not part of value)"}
        tr = pmc_traffic(cfg, n)
        if tr is not None:
            line["roofline"]["traffic"] = tr[0]
            line["roofline"]["traffic_source"] = tr[1] + " (rocprofv3 --pmc pass of this command, not re-measured in this run)"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(P, res["batch"])
        if world == 1 and not args.no_sweep:
            other = {}
            for c in (2, 3, 4):
                if c == cfg:
                    continue
                r = run_config(ctl, q, c, CONFIG_N[c], 0, max(5, min(args.steps, 50)), 3, None, device)
                k = max(5, min(args.steps, 50))
                other[f"config{c}"] = {"robots": CONFIG_N[c], "QPs_per_s": CONFIG_N[c] * k / r["wall"],
                                       "solved_fraction": r["solved"] / CONFIG_N[c],
                                       "hbm_GBs": (BYTES_PER_ROBOT_WARM if r["warm"] else BYTES_PER_ROBOT_COLD) * CONFIG_N[c] * k / r["event_s"] / 1e9}
            r = run_config(ctl, q, 2, CONFIG_N[2], 0, max(5, min(args.steps, 50)), 3, None, device, fused=True)
            other["config2_fused_tick"] = {"robots": CONFIG_N[2], "QPs_per_s": CONFIG_N[2] * max(5, min(args.steps, 50)) / r["wall"],
                                           "solved_fraction": r["solved"] / CONFIG_N[2],
                                           "what": "joint_q -> forward kinematics -> control() -> clamp(J^T f) -> joint_tau in one launch (584 B/robot)"}
            r = run_config(ctl, q, 3, CONFIG_N[3], 0, max(5, min(args.steps, 50)), 3, None, device, fused="full")
            other["config3_full_tick"] = {"robots": CONFIG_N[3], "ticks_per_s": CONFIG_N[3] * max(5, min(args.steps, 50)) / r["wall"],
                                          "solved_fraction": r["solved"] / CONFIG_N[3],
                                          "what": "joint states + COM state + gait phases -> complete joint torque command "
                                                  "(FK, contact rule, foothold planner, swing trajectories, IK, joint PD, QP, J^T) in one launch"}
            line["other_configs"] = other
        if world == 1 and not args.no_sweep:
            line["host_boundary"] = host_boundary(ctl, q)
        if world == 1 and (args.probe_batch_load or not args.no_sweep):
            line["batch_load_probe"] = batch_load_probe(q, P, device)
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
