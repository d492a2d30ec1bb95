#!/usr/bin/env python
"""Headline benchmark: friction-cone QPs/sec of the batched balance controller.

  python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5] [--robots R] [--scaling strong|weak]

A "step" is one pass of the hot path (one qc_control_batch launch) over one batch of synthetic robots that is
resident in HBM.  N = 1 (default): BASELINE.json configs[1] - 4096 randomised COM poses/velocities, all four feet
in contact, mu = 0.6 (SURVEY.md 8d "config 2").  N > 1: BASELINE.json configs[4] ("config 5") - a FIXED total of
2,097,152 robots of the config-3 distribution, rank k of N owning the contiguous shard shard_bounds(2097152, k, N)
("scaling": "strong"; `--scaling weak` keeps `--robots` robots per GPU instead).  The path has no data exchange: RCCL
(torch.distributed backend "nccl") carries only the barrier and the reduction of the timing / solved counters.

`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment) starts its own N ranks,
one per GPU, through torch.distributed.run; started by a launcher it checks that --gpus equals WORLD_SIZE.

Cold-cache protocol: the per-GPU working sets of configs 2-5 (2 ... 130 MB) fit the 256 MiB
Infinity Cache, so the timed loop rotates through K distinct input/output sets with K x bytes > 512 MiB; `value`
and `roofline` are measured that way (inputs come from HBM), and the replay of ONE resident set is reported next
to it as `warm_cache`.

Prints ONE JSON line on rank 0 (see the task contract): value = robots solved per second over all ranks;
roofline = algorithmic bytes (488 B per robot: 388 B read + 100 B written) / average kernel time
measured with HIP events on the launch stream; cpu_baseline = the C oracle (a port, not qpOASES) timed on the host
cores on the same workload.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_ROBOT_COLD = 488   # 48 f64 + 4 B stance read; 12 f64 + 4 B status written
BYTES_PER_ROBOT_WARM = 496   # + 4 B warm word read + 4 B active-set word written
# SURVEY 8(f) ticks.  fused: joint_q replaces feet (same 384 B), joint_tau adds 96 B written.  full: no stance bytes; + gait phases
# (32 B), joint_qdot (96 B) and the 224-byte swing-planning record read, + the record's 32 B of state words (leg_state, has_traj)
# every tick rewrites - its 192 B of trajectory end points are written only on a stance -> swing edge, which the replayed
# bench tick does not have after its first launch.
SWEEP_STEPS = 50  # timed steps of every other_configs entry (independent of --steps)
SWEEP_WARMUP = 20     # warm-up launches of the tick entries ...
# ... and the least time EVERY timed region of this file spends launching the same workload before its clock starts (on top of
# the W warm-up steps asked for).  tools/tick_protocol_scan.py (profiles/r05_tick_protocol_*.log): after the seconds of host-side
# input generation that precede a timed region the device needs ~10-20 ms of work to come back to its running clocks - the first
# ~60 launches of the 262,144-robot tick take 157-167 us, then 151, then 141-145 in steady state; a 50-step region after 3 warm-up
# launches read 158-160 us, the same launches after 25 ms of warm-up 140.5-141.7 (that was round 4's "profile says 143-148, bench
# line says 157-160").  A controller stepping robots every tick runs on a device that is up; that is what is reported.
SWEEP_WARM_MS = 25.0
BYTES_PER_ROBOT_FUSED = 584  # 384 + 4 read; 96 + 4 + 96 written
BYTES_PER_ROBOT_FULL = 964   # 384 + 32 + 96 + 224 read; 96 + 4 + 96 + 32 written (frozen gait phase: "full-frozen")
# the complete tick with the gait clock running: + 8 B gait_dt read + 32 B advanced phases written back, and 48 B
# (p_start + p_final of one leg) for every stance -> swing edge, at the realised rate of the timed region
BYTES_PER_ROBOT_CLOCK = 8 + 32
BYTES_PER_LEG_EDGE = 48
GAIT_DT = 1.0 / 300.0        # the controller's tick, mit_cheetah_config.yaml:3 (commander_node.cpp:343 loop rate)
GAIT_T_SWING, GAIT_T_STANCE = 0.18, 0.8  # mit_cheetah_config.yaml:17-18 = qc_set_gait's defaults
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
ROTATE_BYTES = 512 << 20     # the rotating sets of the cold-cache protocol cover more than this
CONFIG5_TOTAL = 2097152

CONFIG_N = {2: 4096, 3: 65536, 4: 262144, 5: 262144}  # robots per GPU (weak scaling / N = 1)
TICK_DESC = {"fused": "; TICK: joint_q -> forward kinematics -> control() -> clamp(J^T f) -> joint_tau in the same launch (584 B/robot)",
             "full": "; TICK: joint states + COM state + gait phases + dt -> complete joint torque command (gait clock GaitScheduler::update, FK, contact "
                     "rule, foothold planner on every stance->swing edge, swing trajectories, IK, joint PD, QP, J^T) in the same launch; the clock "
                     "advances by dt = 1/300 s per launch, so edges and replanning occur inside the timed region at their natural rate "
                     "(1004 B/robot + 48 B per edge)",
             "full-frozen": "; TICK: as 'full' with the gait phase FROZEN (no gait_dt): contact rule, swing trajectories, IK, joint PD, QP, J^T are "
                            "timed, GaitScheduler::update and FootPlanner::singleFoot are not (964 B/robot; rounds 3-4's protocol, kept for continuity)"}
CONFIG_DESC = {
    2: "config2: batch of {n} randomised COM poses/velocities per GPU, all 4 feet in contact, mu=0.6 pyramid cone, cold start",
    3: "config3: batch of {n} per GPU, mixed 2/3/4-foot contact states from trot/walk gait schedules, cold start",
    4: "config4: batch of {n} per GPU, tick 1 warm-started from tick 0's active set (dt=1/300 s)",
    5: "config5: 2,097,152-robot batch (config-3 distribution, seed 0x5EED0005) sharded contiguously, {n} robots on this GPU's shard",
}


def kernel_src_sha16():
    """Fingerprint of the kernel sources this run was built from (the .git directory does not travel to the GPU
    box): profiles carry it, so a committed PMC pass can be matched to the kernels that produced it."""
    h = hashlib.sha256()
    for f in ("quadruped_control_amd/csrc/qc_balance.hip", "quadruped_control_amd/csrc/qc_device.hpp", "include/qc_balance.h"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def make_batch(cfg, n, start, seed_shift=0):
    from quadruped_control_amd import workloads as W

    seed = W.SEEDS[cfg] + seed_shift
    if cfg == 2:
        return W.config2(n, start=start, seed=seed), None
    if cfg == 3:
        return W.config3(n, start=start, seed=seed), None
    if cfg == 4:
        t0, t1 = W.config4(n, start=start, seed=seed)
        return t1, t0
    return W.config5(n, start=start, seed=seed), None


def bytes_per_robot(warm, fused=False, edge_legs=0.0):
    """Algorithmic bytes per robot; `edge_legs` = stance -> swing edges per robot per tick realised in the timed region."""
    if fused == "full":
        return BYTES_PER_ROBOT_FULL + BYTES_PER_ROBOT_CLOCK + BYTES_PER_LEG_EDGE * edge_legs
    if fused:
        return BYTES_PER_ROBOT_FULL if fused == "full-frozen" else BYTES_PER_ROBOT_FUSED
    return BYTES_PER_ROBOT_WARM if warm else BYTES_PER_ROBOT_COLD


def rotation_sets(n, warm, fused=False):
    per = int(bytes_per_robot(warm, fused)) * n
    return 1 if per > ROTATE_BYTES else ROTATE_BYTES // per + 1


def make_tick_batch(cfg, n, start, fused, j=0):
    """Inputs of the SURVEY 8(f) ticks for rotation set j.  fused = True: rows 1 + 2 (joint angles in - forward kinematics on
    the device -, joint torques out: J^T f, clamped); "full": rows 3 + 4 too (contact state from gait phases, foothold
    planner, swing trajectories, IK, joint PD)."""
    import numpy as np

    from quadruped_control_amd import workloads as W

    batch, _ = make_batch(cfg, n, start, seed_shift=0x100 * j)
    batch = W.with_joint_angles(batch, seed=0x5EED0006 + 0x100 * j, start=start)
    if fused in ("full", "full-frozen"):
        batch = W.with_swing_references(batch, seed=0x5EED0007 + 0x100 * j, start=start)
        idx = np.arange(start, start + n, dtype=np.uint64)
        phase = np.fmod(np.array([0.0, 0.5, 0.5, 0.0])[None] + W.uniform(0x5EED0009 + 0x100 * j, idx, 3)[:, None], 1.0)
        batch = {k: v for k, v in batch.items() if k not in ("stance", "swing_pos", "swing_vel")}
        batch["gait_phase"] = np.ascontiguousarray(phase)
        if fused == "full":  # the on-device gait clock (gait.cpp:113-123): every launch is one controller tick later
            batch["gait_dt"] = np.full(n, GAIT_DT)
    return batch


def stance_from_phase(ph, duty):
    """GaitScheduler::phase(), gait.cpp:125-134, on a torch tensor of phases (the 1e-12 slack of math::almost_equal)."""
    return ((ph > 0.0) | (ph.abs() < 1.0e-12)) & ((ph < duty) | ((ph - duty).abs() < 1.0e-12))


def count_gait_edges(phase0, launches_of_set, n, phase_after):
    """What the timed region of a clocked tick did to the gait: replays GaitScheduler::update (gait.cpp:113-123, the same
    fmod arithmetic as the device) from the phases snapshotted when the clock started, `launches_of_set[j]` times for set j,
    counts the stance -> swing edges (each one a FootPlanner::singleFoot call + a trajectory reset on the device) and checks
    the replayed phases against the ones the device left behind.  Returns (edges, leg-ticks, robots with an edge, robot-ticks,
    max |replayed - device phase|)."""
    import torch

    step = 1.0 / (GAIT_T_SWING + GAIT_T_STANCE) * GAIT_DT
    duty = GAIT_T_STANCE / (GAIT_T_SWING + GAIT_T_STANCE)
    edges = legs = robots_edge = robots = 0
    worst = 0.0
    for j, c in enumerate(launches_of_set):
        ph = phase0[j * n:(j + 1) * n].clone()
        st = stance_from_phase(ph, duty)
        for _ in range(c):
            ph = torch.fmod(ph + step, 1.0)
            st_new = stance_from_phase(ph, duty)
            e = st & ~st_new
            edges += int(e.sum().item())
            robots_edge += int(e.any(dim=1).sum().item())
            st = st_new
        legs += 4 * n * c
        robots += n * c
        worst = max(worst, float((ph - phase_after[j * n:(j + 1) * n]).abs().max().item()))
    return edges, legs, robots_edge, robots, worst


def time_launches(launches, steps, warmup, dist=None, warm_all=False, warm_ms=0.0, before_timed=None):
    """W untimed steps, then K timed steps bracketed by barrier + synchronize; step i runs launches[i % len].
    `warm_ms` > 0: after the W steps keep launching whole rotations (untimed) until that much wall time has passed, so that a
    short timed region does not start on a device whose clocks are still coming up (SWEEP_WARM_MS above).
    `before_timed(w)`: called once after the warm-up has drained (w = launches done so far), outside the timed region.
    Returns (wall seconds for K steps, HIP-event seconds for K steps)."""
    import torch

    m = len(launches)
    w = max(warmup, min(m, 8))  # at least a few sets so that every code path is paged in
    if warm_all:  # stateful ticks: every set has seen its first call (which plans all swinging legs) before the clock starts
        w = max(w, m)
    # (the events exist before the bracket: creating them between the synchronize and the first timed launch would only
    # lengthen the idle gap the GPU comes out of - at the driver's 20 steps every microsecond of it is 0.25 % of the region)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    for i in range(w):
        launches[i % m]()
    # (the instruments get their warm-up too: the first record / elapsed_time of a process - and of these two event objects - sets up
    # the runtime's event machinery; left to the timed region that is 20-30 us of host time between its two clock reads, tools/r06_sync_latency.py)
    ev0.record()
    ev1.record()
    torch.cuda.synchronize()
    ev0.elapsed_time(ev1)
    if warm_ms > 0.0:
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < warm_ms:
            for _ in range(m):
                launches[w % m]()
                w += 1
            torch.cuda.synchronize()
    if before_timed is not None:
        before_timed(w)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()  # torch's current stream == the stream control_batch launches on
    for i in range(steps):
        launches[(w + i) % m]()  # continues where the warm-up stopped: with K distinct sets the first timed ones were never touched
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0  # this rank's K steps; the caller takes the MAX over ranks
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    return wall, ev0.elapsed_time(ev1) * 1e-3


def run_config(ctl, q, cfg, n, start, steps, warmup, dist=None, device=0, fused=False, protocols=("cold", "warm"), warm_ms=0.0, from_idle_s=0.0):
    """Runs the workload under the cold-cache protocol (rotating sets) and/or as a replay of one resident set.
    Returns dict(cold=(wall, event_s), warm_cache=(wall, event_s), solved, n, sets, batch (host, set 0), warm) and, for the
    clocked complete tick (fused == "full"), gait = what the gait clock did inside the cold timed region."""
    import numpy as np
    import torch

    dev = f"cuda:{device}"
    on_device = cfg in (3, 5) and not fused
    # the host copy of set 0 only feeds cpu_baseline; configs 3 / 5 make theirs on the device (below) and keep a small host sample
    if fused:
        batch, prev = make_tick_batch(cfg, n, start, fused), None
    else:
        batch, prev = make_batch(cfg, min(n, 4096) if on_device else n, start)
    is_warm = prev is not None
    sets = rotation_sets(n, is_warm, fused) if "cold" in protocols else 1
    # set 0 is the canonical workload; sets 1.. hold other robots of the same distribution (shifted seed)
    if cfg in (3, 5) and not fused:
        # config 3 / 5 inputs are generated on the device, per shard (same counter-based PRNG; the
        # contact states are bit-identical to the host generator's, the rotations agree to the last ulp)
        from quadruped_control_amd import workloads as W
        from quadruped_control_amd import workloads_device as WD

        parts = [WD.config3(n, start=start, seed=W.SEEDS[cfg] + 0x100 * j, device=device) for j in range(sets)]
        big = {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
        del parts
        prevs = [None]
    else:
        hosts, prevs = [batch], [prev]
        for j in range(1, sets):
            if fused:
                b, p = make_tick_batch(cfg, n, start, fused, j), None
            else:
                b, p = make_batch(cfg, n, start, seed_shift=0x100 * j)
            hosts.append(b)
            prevs.append(p)
        big = {k: torch.from_numpy(np.concatenate([h[k] for h in hosts])).to(dev) for k in batch}
        del hosts
    stateful = fused in ("full", "full-frozen")
    if stateful:  # one planning record per robot of every set (in/out)
        big["swing_state"] = torch.from_numpy(q.new_swing_states(sets * n).view("uint8").reshape(-1).copy()).to(dev)
    out = {"grf_body": torch.empty((sets * n, 12), dtype=torch.float64, device=dev),
           "status": torch.empty((sets * n,), dtype=torch.int32, device=dev)}
    warm = None
    if is_warm:  # config 4: tick 0 (cold) produces the warm-start words for tick 1
        big_prev = {k: torch.from_numpy(np.concatenate([p[k] for p in prevs])).to(dev) for k in prevs[0]}
        o0 = ctl.control_batch(big_prev, want_active_set=True)
        torch.cuda.synchronize()
        warm = o0["active_set"]
        del big_prev, o0["grf_body"]
        out["active_set"] = torch.empty((sets * n,), dtype=torch.int32, device=dev)
    del prevs
    if fused:
        out["joint_tau"] = torch.empty((sets * n, 12), dtype=torch.float64, device=dev)
    launches = []
    for j in range(sets):
        sl = slice(j * n, (j + 1) * n)
        bj = {k: (v[j * n * 224:(j + 1) * n * 224] if k == "swing_state" else v[sl]) for k, v in big.items()}
        oj = {k: v[sl] for k, v in out.items()}
        launches.append(ctl.plan_batch(bj, warm=None if warm is None else warm[sl], out=oj)[0])
    res = dict(n=n, sets=sets, batch=batch, warm=is_warm, out={k: v[:n] for k, v in out.items()})
    if "cold" in protocols:
        snap = {}

        def before_timed(w):  # the clock is about to start: remember where every robot's gait clock stands
            snap["w"] = res["warmup_done"] = w
            if fused == "full":
                snap["phase"] = big["gait_phase"].clone()

        res["cold"] = time_launches(launches, steps, warmup, dist, warm_all=stateful, warm_ms=warm_ms, before_timed=before_timed)
        if fused == "full":
            per_set = [sum(1 for i in range(steps) if (snap["w"] + i) % sets == j) for j in range(sets)]
            edges, legs, r_edge, r_ticks, worst = count_gait_edges(snap["phase"], per_set, n, big["gait_phase"])
            res["gait"] = {"dt_s": GAIT_DT, "period_s": GAIT_T_SWING + GAIT_T_STANCE, "timed_ticks_per_set": per_set,
                           "stance_to_swing_edges": edges, "edge_legs_per_robot_tick": edges / r_ticks,
                           "robots_replanning_per_tick": r_edge / r_ticks,
                           "expected_edge_legs_per_robot_tick": 4.0 * GAIT_DT / (GAIT_T_SWING + GAIT_T_STANCE),
                           "device_phase_vs_replayed_clock_max_abs": worst,
                           "what": "GaitScheduler::update (gait.cpp:113-123) runs on the device at the start of every robot's tick: each launch "
                                   "is one controller tick (dt = 1/300 s) later; a stance -> swing edge makes that leg replan its foothold "
                                   "(FootPlanner::singleFoot, foot_planner.cpp:76-157) and resets the trajectories (trajectory.cpp:308-344) - "
                                   "counted by replaying the clock on the snapshot taken when the timed region started and checked against the "
                                   "phases the device left behind (0 = bit-equal)"}
            del snap["phase"]
    if "warm" in protocols:
        res["warm_cache"] = time_launches(launches[:1], steps, warmup, dist, warm_ms=warm_ms)
    if from_idle_s > 0.0 and "cold" in protocols:
        # rounds 1-4's protocol, for continuity: exactly W warm-up steps on a device that has been idle (what a timed region that
        # follows seconds of host-side input generation sees) - reported next to `value`, never as `value`; with ranks, under the
        # same barrier bracket as `value` (every line carries the figure that compares with earlier rounds)
        time.sleep(from_idle_s)
        res["cold_from_idle"] = time_launches(launches, steps, warmup, dist, warm_all=stateful, warm_ms=0.0)
    res["solved"] = int((out["status"][:n] == 0).sum().item())
    res["solved_all_sets"] = int((out["status"] == 0).sum().item()) if "cold" in protocols else res["solved"]
    return res


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


def cpu_model():
    """The host CPU as /proc/cpuinfo names it ("T and the CPU model are printed")."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform

    return platform.processor() or platform.machine() or "unknown"


def oracle_build_flags():
    """Compiler and flags the C oracle was built with, read from oracle/Makefile (what `make -C oracle liboracle.so` ran)."""
    try:
        txt = open(os.path.join(ROOT, "oracle", "Makefile")).read()
        import re

        cc = re.search(r"^CC \?= *(.*)$", txt, re.M).group(1).strip()
        fl = re.search(r"^CFLAGS \?= *(.*)$", txt, re.M).group(1).strip()
        cc = os.environ.get("CC", cc)
        try:
            ver = subprocess.run([cc, "-dumpfullversion"], capture_output=True, text=True, timeout=10).stdout.strip()
        except Exception:
            ver = ""
        return f"{cc} {ver} {os.environ.get('CFLAGS', fl)}".replace("  ", " ")
    except Exception:
        return "unknown"


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
    c_oracle.set_refine(False)  # time the algorithm, not the checker's long-double recomputation of the accepted point
    try:
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
    finally:
        c_oracle.set_refine(True)  # the checker's mode, whatever happened above
    literal = None
    try:  # the literal reference sequence (qpOASES init -> hotstart), only where qpOASES itself is installed
        from oracle.qpoases_ref import run as qref

        if qref.find_qpoases() is not None and qref.build() is not None:
            literal = json.loads(subprocess.run([sys.executable, qref.__file__, "time", "4096"], capture_output=True, text=True,
                                                timeout=300).stdout.strip().splitlines()[-1])
    except Exception:
        literal = None
    if literal is not None and literal.get("kind") == "reference":
        literal["port_value"] = reps * n / dt
        return literal
    return {"value": reps * n / dt, "unit": "QPs/s", "cores": threads, "kind": "port",
            "sample": f"first {base} robots of the benchmark batch tiled x{tile} = {n} robots per call x {reps} calls, {threads} OpenMP threads, "
                      f"{dt:.1f} s wall; C restatement (textbook primal active set), not qpOASES",
            "single_thread_value": one,
            # what the figure depends on besides the algorithm (it moved 5.4e5 -> 4.4e5 between rounds 2 and 3 on different host CPUs)
            "cpu_model": cpu_model(), "threads": threads, "host_logical_cpus": os.cpu_count(),
            "compiler": oracle_build_flags(),
            "refinement": "off (the checker's long-double recomputation of the accepted point is not timed; decisions are the same)"}


def cpu_baseline_tick(P, batch, budget_s=2.5):
    """The C oracle's complete tick (oracle_gait_update + oracle_tick_planned_batch: gait clock, FK, contact rule, foothold
    planner, swing trajectories, IK, joint PD, QP, J^T - the restatement of commander_node.cpp:383-531) on the host cores over
    robots of the same workload, the clock advancing by 1/300 s per call like the timed launches."""
    from oracle import c_oracle

    import numpy as np

    threads = host_cores()
    base = min(batch["x"].shape[0], 4096)
    tile = max(1, (256 * threads + base - 1) // base)
    keys = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "joint_q", "joint_qdot", "gait_phase")
    sample = {k: np.ascontiguousarray(np.tile(batch[k][:base], (tile, 1))) for k in keys}
    n = base * tile
    dt = np.full(n, GAIT_DT)
    kin = c_oracle.default_kinematics()
    kin.t_swing, kin.t_stance = GAIT_T_SWING, GAIT_T_STANCE
    states = c_oracle.new_swing_states(n)
    c_oracle.set_refine(False)
    try:
        c_oracle.tick_planned_batch(P, sample, states, kin=kin, threads=threads)  # the first call plans every swinging leg: not timed, as on the GPU
        one_b = {k: v[:256].copy() for k, v in sample.items()}
        one_s = states[:256].copy()
        t0 = time.perf_counter()
        c_oracle.gait_update(one_b["gait_phase"], dt[:256], kin=kin)
        c_oracle.tick_planned_batch(P, one_b, one_s, kin=kin, threads=1)
        one = 256 / (time.perf_counter() - t0)
        reps, t0 = 0, time.perf_counter()
        while True:
            c_oracle.gait_update(sample["gait_phase"], dt, kin=kin)
            c_oracle.tick_planned_batch(P, sample, states, kin=kin, threads=threads)
            reps += 1
            if time.perf_counter() - t0 >= budget_s:
                break
        wall = time.perf_counter() - t0
    finally:
        c_oracle.set_refine(True)
    return {"value": reps * n / wall, "unit": "robot ticks/s", "cores": threads, "kind": "port",
            "sample": f"first {base} robots of the entry's set 0 tiled x{tile} = {n} robots per call x {reps} consecutive ticks (dt = 1/300 s), {threads} "
                      f"OpenMP threads, {wall:.1f} s wall; C restatement of the reference's tick, not the reference binary",
            "single_thread_value": one, "cpu_model": cpu_model(), "threads": threads, "compiler": oracle_build_flags()}


def probe_pmc(sha):
    """The committed rocprofv3 passes of the batch-load probe on THESE kernel sources (profiles/rNN_batchload.json, written by
    tools/profile_probe.sh + tools/summarize_probe.py: one --kernel-trace --stats pass, separate --pmc FETCH_SIZE / WRITE_SIZE
    passes, FETCH_SIZE doubled per MI355X_MICROARCH.md), or None."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_batchload.json"))):
        try:
            d = json.load(open(f))
            if d.get("kernel_src_sha16") == sha and "traffic" in d:
                best = (d, os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def batch_load_probe(q, P, device, nb=2097152, steps=10):
    """north_star: "achieved HBM GB/s on the batch load".  Same kernel with the solver iterations skipped
    (qc_set_tuning "probe_batch_load": load -> PD law / rotation log / Newton-Euler rhs -> output transform -> store;
    status = max_iter for every robot) on a batch four times the 256 MiB Infinity Cache.  The rows are
    262,144 config-5 robots tiled 8x on the device (their values do not matter without iterations)."""
    import torch

    from quadruped_control_amd import workloads as W

    probe = q.BalanceController.from_params(P, device=device).set_tuning(probe_batch_load=1)
    base = q.to_device(W.config5(nb // 8), device)
    batch = {k: v.repeat(8, 1).contiguous() for k, v in base.items()}
    out = {"grf_body": torch.empty((nb, 12), dtype=torch.float64, device=f"cuda:{device}"),
           "status": torch.empty((nb,), dtype=torch.int32, device=f"cuda:{device}")}
    _, evs = time_launches([probe.plan_batch(batch, out=out)[0]], steps, 2, warm_ms=SWEEP_WARM_MS)
    gbs = BYTES_PER_ROBOT_COLD * nb * steps / evs / 1e9
    out = {"robots": nb, "bytes": BYTES_PER_ROBOT_COLD * nb, "us": evs / steps * 1e6, "achieved": gbs, "unit": "GB/s",
           "peak": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS, "kernel": probe.kernel_name, "kernel_mode": probe.query_launch(nb)["mode"],
           "traffic": None}
    pm = probe_pmc(kernel_src_sha16())
    if pm is not None:  # counters cannot be read from inside the benchmarked process: the hash-matched committed passes
        d, src = pm
        out["traffic"] = d["traffic"]["hbm_bytes_per_launch"]
        out["traffic_ratio_to_algorithmic"] = d["traffic"]["hbm_bytes_per_launch"] / (BYTES_PER_ROBOT_COLD * nb)
        out["avg_kernel_us_rocprof"] = d["kernel"]["avg_ns"] * 1e-3
        out["achieved_rocprof"] = BYTES_PER_ROBOT_COLD * nb / d["kernel"]["avg_ns"]  # GB/s, algorithmic bytes / rocprof's kernel average
        out["traffic_source"] = src + " (rocprofv3 --kernel-trace --stats + separate --pmc FETCH_SIZE / WRITE_SIZE passes of this probe on these kernel sources)"
    out["what"] = "load -> assemble (PD law, rotation log, Newton-Euler rhs) -> output transform -> store, no QP iterations"
    return out


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


def pmc_traffic(cfg, n, sha, kernel, tick=None):
    """HBM bytes per launch from a committed PMC pass of this workload AND these kernel sources
    (profiles/rNN_cfg<cfg>.json, produced by tools/profile_r.sh + tools/summarize_profile.py: separate --pmc
    FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md).  None if no matching pass:
    counters cannot be read from inside the benchmarked process."""
    import glob

    best = None
    pattern = f"r*_tick_{tick}*.json" if tick else f"r*_cfg{cfg}*.json"  # (r03_cfg5.json = the shard, r03_cfg5_n1.json = the whole batch)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            d = json.load(open(f))
            bl = d.get("bench_line", {})
            if (bl.get("config", {}).get("robots_per_gpu") == n and bl.get("config", {}).get("kernel") == kernel
                    and bl.get("config", {}).get("tick") == tick and bl.get("config", {}).get("config_id", cfg) == cfg
                    and bl.get("kernel_src_sha16") == sha and "traffic" in d):
                best = (d["traffic"]["hbm_bytes_per_launch"], os.path.relpath(f, ROOT), d.get("valu"))
        except Exception:
            pass
    return best


def attach_pmc(target, cfg, n, sha, kernel, tick=None, kernel_us=None):
    """roofline.traffic / roofline_valu of `target` (a bench line or an other_configs entry) from the committed,
    hash-matched PMC pass of that workload.  The HBM fraction is what the contract asks for; the bound that binds the
    solve is FP64 VALU issue, so that one travels next to it.  `kernel_us` = the average kernel time THIS run measured for
    the entry: the issue fraction is the profile's instruction count over this run's time (the profile's own
    kernel time described another run)."""
    tr = pmc_traffic(cfg, n, sha, kernel, tick)
    if tr is None:
        return
    src = tr[1] + " (rocprofv3 --pmc passes of this workload on these kernel sources)"
    if "roofline" in target:
        target["roofline"]["traffic"] = tr[0]
        target["roofline"]["traffic_source"] = src
    else:
        target["hbm_traffic_bytes_per_launch"] = tr[0]
    if tr[2]:
        v = tr[2]
        own = kernel_us is not None and kernel_us > 0.0
        frac = v["insts_valu_per_launch"] * 4.0 / (1024.0 * kernel_us * 1e3 * 2.4) if own else v["issue_frac"]
        target["roofline_valu"] = {"bound": "fp64-valu-issue", "insts_valu_per_launch": v["insts_valu_per_launch"],
                                   "insts_valu_per_robot": v["insts_valu_per_robot"], "issue_frac": frac,
                                   "kernel_us": kernel_us if own else v.get("kernel_ns", 0.0) * 1e-3,
                                   "issue_frac_of_profiled_run": v["issue_frac"],
                                   "how": "SQ_INSTS_VALU (committed PMC pass, an instruction count: the same in every run of these kernels on this "
                                          "workload) x 4 cycles / (1024 SIMDs x kernel time x 2.4 GHz), kernel time = "
                                          + ("this entry's own HIP-event average" if own else "the profiled run's"),
                                   "source": src}


def rates(r, key, n, steps, bytes_per):
    wall, evs = r[key]
    return {"QPs_per_s": n * steps / wall, "avg_kernel_us": evs / steps * 1e6, "hbm_GBs": bytes_per * n * steps / evs / 1e9}


def parse_rccl_log(text, world):
    """What RCCL said about the communicator it built (NCCL_DEBUG=INFO, subsystems INIT,GRAPH - set by main() for the group formation
    and its first all-reduce only): ranks, channels, and over WHAT each rank reaches its peers.  RCCL prints one line per channel and
    peer, "Channel 03/0 : 2[2] -> 3[3] via P2P/IPC" (xGMI / PCIe peer access), "... via SHM/direct/direct" (host memory) or
    "... via NET/..." - counted here as they come; the first matching lines travel verbatim in `lines`, so the record explains itself
    even where this parser has never seen the format (it has only ever run on one-GPU boxes)."""
    import re

    via = {}
    chans = set()
    nranks = None
    graph = {}
    keep, seen_kinds = [], {}
    for ln in text.splitlines():
        body = re.sub(r"^.*NCCL INFO ", "", ln)
        m = re.search(r"Channel (\d+)(?:/\d+)?\s*:\s*(\d+)\[[^\]]*\]\s*->\s*(\d+)\[[^\]]*\]\s*(?:\[\w+\]\s*)?via\s+(\S+)", ln)
        if m:
            chans.add(int(m.group(1)))
            via[m.group(4)] = via.get(m.group(4), 0) + 1
        m2 = re.search(r"nranks\s+(\d+)", ln)
        if m2:
            nranks = int(m2.group(1))
        m3 = re.search(r"Pattern (\d+).*nChannels (\d+), bw ([\d.]+)/([\d.]+), type (\S+?),", ln)  # the topology search's result: link type per pattern (XGMI / PIX / ...)
        if m3:
            graph["pattern%s" % m3.group(1)] = {"nChannels": int(m3.group(2)), "bw": float(m3.group(3)), "type": m3.group(5)}
        kind = ("via" if " via " in body else "nranks" if "nranks" in body else "pattern" if body.startswith("Pattern") else "xgmi" if re.search(r"xgmi", body, re.I)
                else "connected" if "Connected all" in body else "ring" if re.match(r"Ring \d", body) else "tree" if re.match(r"Tree \d", body) else None)
        cap = {"via": 8, "nranks": 2, "pattern": 4, "xgmi": 6, "connected": 2, "ring": 2, "tree": 2}
        if kind and seen_kinds.get(kind, 0) < cap[kind]:
            seen_kinds[kind] = seen_kinds.get(kind, 0) + 1
            keep.append(body[:170])
    kinds = sorted(via)
    transport = ("none (one rank)" if world == 1 else
                 "P2P (xGMI / peer access)" if kinds and all(k.startswith("P2P") for k in kinds) else
                 "+".join(kinds) if kinds else "unknown (no channel lines in the log)")
    return {"nranks": nranks, "channels": len(chans) or None, "via": via, "graph": graph, "transport": transport, "lines": keep}


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5], help="default: 2 at N = 1, 5 at N > 1")
    ap.add_argument("--robots", type=int, default=0, dest="n",
                    help="robots per GPU (default: the config's size; with --scaling strong the total is robots x N).  "
                         "(Not '--n': torch.distributed.run's own parser would take that for an abbreviation of --nnodes.)")
    ap.add_argument("--scaling", choices=["auto", "strong", "weak"], default="auto",
                    help="N > 1: strong = config 5's fixed 2,097,152 robots sharded over the ranks (default); weak = --robots robots per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the informational runs of the other configs")
    ap.add_argument("--gather-results", action="store_true",
                    help="N > 1: after the timed region, all-gather the per-rank GRF blocks (result collection over xGMI, "
                         "SURVEY 8e) and report the collective's time separately")
    ap.add_argument("--probe-batch-load", action="store_true",
                    help="time the load -> assemble -> store phase alone (no solver iterations) on 2,097,152 robots "
                         "(done by default together with the sweep)")
    ap.add_argument("--probe-only", action="store_true",
                    help="profiling (tools/profile_probe.sh): run ONLY the batch-load probe (--steps launches of it) and print its entry with the kernel-source hash")
    ap.add_argument("--tick", choices=["fused", "full", "full-frozen"], default=None,
                    help="development / profiling: time the SURVEY 8(f) tick built around the QP instead of the QP alone - fused = joint_q -> "
                         "FK -> control() -> J^T -> joint_tau; full = + gait clock (dt = 1/300 s per launch), contact rule, foothold planner on every "
                         "stance->swing edge, swing trajectories, IK, joint PD; full-frozen = full without gait_dt (the phases never move)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="development: qc_set_tuning override(s)")
    ap.add_argument("--device-warm-ms", type=float, default=SWEEP_WARM_MS,
                    help="besides the W warm-up steps, keep launching the workload (untimed) for this many milliseconds before the timed region, "
                         "so that it starts on a device at its running clocks (0: exactly W warm-up steps, the device possibly still ramping up)")
    args = ap.parse_args()

    one_dev = os.environ.get("QC_BENCH_ONE_DEVICE") == "1"  # test hook: all ranks on cuda:0, gloo instead of RCCL
    # the N > 1 line says HOW RCCL moved its bytes.  RCCL fixes its debug level the first time anything in the
    # library logs - possibly while torch is being imported - so the variables are set HERE, before `import torch`, for every process of
    # a run that will form a process group (the self-launched ranks inherit them; %p = each rank's own pid).  Subsystems INIT and
    # GRAPH only: the communicator's construction is logged, a collective in the timed region is not.
    # (a quiet level in the environment - the GPU boxes export NCCL_DEBUG=VERSION - is raised; a caller already debugging at INFO / TRACE keeps its own setup)
    if (args.gpus > 1 or os.environ.get("QC_BENCH_FORCE_DIST") == "1") and os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):
        import tempfile

        os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH", NCCL_DEBUG_FILE=os.path.join(tempfile.gettempdir(), "qc_bench_rccl_%p.log"),
                          QC_BENCH_RCCL_LOG="1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the N ranks ourselves, one per GPU
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus and not one_dev:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch

    import quadruped_control_amd as q

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}")
    dist = None
    # QC_BENCH_FORCE_DIST=1 (test hook): bring the process group up even for one rank, so that the RCCL code path
    # (init, barrier, device-tensor all-reduces, the optional gather) executes on a 1-GPU box too
    if world > 1 or os.environ.get("QC_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist_mod

        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # Fail fast and readably: a rank that never joins (or a collective that never completes) ends every rank with a
        # one-line diagnostic after QC_BENCH_TIMEOUT_S (default 60 s) instead of RCCL's silent ten-minute wait.
        limit = float(os.environ.get("QC_BENCH_TIMEOUT_S", "60"))
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        backend = "gloo" if (one_dev and os.environ.get("QC_BENCH_ONE_DEVICE_BACKEND", "gloo") == "gloo") else "nccl"
        # (each rank's RCCL writes its own log file - see the top of main(); it is read right after the first all-reduce)
        rccl_log = None
        if backend == "nccl" and os.environ.get("QC_BENCH_RCCL_LOG") == "1":
            rccl_log = os.environ["NCCL_DEBUG_FILE"].replace("%p", str(os.getpid()))
        try:
            if one_dev:
                local_rank = 0  # test hook: every rank on cuda:0 (gloo; "nccl" where RCCL accepts two ranks per device)
            torch.cuda.set_device(local_rank)
            kw = {} if backend == "gloo" else {"device_id": torch.device("cuda", local_rank)}
            dist_mod.init_process_group(backend, timeout=datetime.timedelta(seconds=limit), **kw)
            probe = torch.ones(1, device=f"cuda:{local_rank}") if backend == "nccl" else torch.ones(1)
            dist_mod.all_reduce(probe)  # the first collective builds the communicator: do it here, under the same watch
            if backend == "nccl":
                torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError(f"first all-reduce saw {int(probe.item())} of {world} ranks")
            rccl_info = {"backend": backend}
            if backend == "nccl":
                try:
                    txt = open(rccl_log).read() if rccl_log and os.path.exists(rccl_log) else ""
                    rccl_info.update(parse_rccl_log(txt, world))
                    rccl_info["log_bytes"] = len(txt)
                    if rccl_log and os.path.exists(rccl_log):
                        os.remove(rccl_log)
                except Exception as e:  # noqa: BLE001 - diagnostics must not end the run
                    rccl_info["error"] = f"{type(e).__name__}: {e}"
                nd = torch.cuda.device_count()
                rccl_info["visible_devices"] = nd
                rccl_info["peer_access"] = [[int(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(nd)] for i in range(nd)] if nd <= 16 else None
                rccl_info["log_source"] = "NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH, this rank's file" if rccl_log else "NCCL_DEBUG was set by the caller: not captured"
        except Exception as e:  # noqa: BLE001 - whatever went wrong, say which rank and where it was waiting
            print(f"bench.py: rank {rank}/{world} (local GPU {local_rank}, backend {backend}, master "
                  f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}): the process group did not form within "
                  f"{limit:.0f} s - {type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}", file=sys.stderr, flush=True)
            os._exit(3)
        dist = dist_mod
    else:
        rccl_info = None
    device = local_rank if dist is not None else 0
    torch.cuda.set_device(device)

    P = q.cheetah_params(mu=0.6)
    if args.probe_only:
        if world != 1:
            sys.exit("bench.py: --probe-only is a one-GPU profiling mode")
        print(json.dumps({"batch_load_probe": batch_load_probe(q, P, device, steps=args.steps), "kernel_src_sha16": kernel_src_sha16(),
                          "steps": args.steps}), flush=True)
        return
    ctl = q.BalanceController.from_params(P, device=device)
    tune = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in args.tune}
    ctl.set_tuning(**tune)
    cfg = args.config or (2 if world == 1 else 5)
    scaling = args.scaling if args.scaling != "auto" else ("strong" if (world > 1 and cfg == 5 and not args.n) else "weak")
    if scaling == "strong":
        from quadruped_control_amd.sharding import shard_bounds

        total = args.n * world if args.n else CONFIG5_TOTAL
        lo, hi = shard_bounds(total, rank, world)
        n, start = hi - lo, lo
    else:
        n = args.n or CONFIG_N[cfg]
        start = rank * n

    def _bounds(r):  # robots [lo, hi) of rank r, as every rank computes them for itself
        if scaling == "strong":
            return shard_bounds(total, r, world)
        return r * n, (r + 1) * n

    fused = {None: False, "fused": True, "full": "full", "full-frozen": "full-frozen"}[args.tick]
    res = run_config(ctl, q, cfg, n, start, args.steps, args.warmup, dist, device, fused=fused, warm_ms=args.device_warm_ms,
                     from_idle_s=2.0 if args.device_warm_ms > 0 else 0.0)

    from quadruped_control_amd.sharding import reduce_counters

    on_gpu = dist is not None and dist.get_backend() == "nccl"
    rdev = f"cuda:{device}" if on_gpu else None
    wall, solved_total, total_robots = reduce_counters(dist, res["cold"][0], res["solved"], n, device=rdev)
    wall_warm, _, _ = reduce_counters(dist, res["warm_cache"][0], 0, 0, device=rdev)
    wall_idle = reduce_counters(dist, res["cold_from_idle"][0], 0, 0, device=rdev)[0] if "cold_from_idle" in res else None  # (max over ranks, like `wall`)
    from quadruped_control_amd.sharding import reduce_rank_stats

    k_min, k_max, allreduce_s = reduce_rank_stats(dist, res["cold"][1] / args.steps * 1e6, device=rdev)

    gather_s = None
    if dist is not None and args.gather_results:
        from quadruped_control_amd.sharding import gather_results

        shard = res["out"]["grf_body"] if on_gpu else res["out"]["grf_body"].cpu()
        gathered, gather_s = gather_results(dist, shard)
        assert gathered.shape[0] == total_robots

    if rank == 0:
        sha = kernel_src_sha16()
        bytes_per = bytes_per_robot(res["warm"], fused, res.get("gait", {}).get("edge_legs_per_robot_tick", 0.0))
        kernel_s = res["cold"][1] / args.steps
        achieved = bytes_per * n / kernel_s / 1e9
        info = ctl.query_launch(n, kin=bool(fused), warm=res["warm"])
        line = {
            "metric": "friction-cone QPs/sec (12 vars, 4-foot stance) at 1/2/4/8 MI355X",
            "value": total_robots * args.steps / wall,
            "unit": "QPs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling if world > 1 else None,  # no scaling claim on a one-GPU line (the key stays for the contract's parsers)
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" + (" (generated on the device, per shard)" if cfg in (3, 5) else ""),
            "config": {"workload": CONFIG_DESC[cfg].format(n=n) + (TICK_DESC[args.tick] if args.tick else ""), "config_id": cfg, "tick": args.tick,
                       "robots_per_gpu": n, "global_batch": total_robots,
                       "kernel": ctl.kernel_name, "lanes_per_robot": info["lanes_per_robot"], "kernel_mode": info["mode"],
                       "resident_workgroups": info["resident_workgroups"],
                       "parallelism": f"batch-shard x{world} (no data-path collective)",
                       "shards": [list(_bounds(r)) for r in range(world)],
                       "cache_protocol": f"cold: the timed loop rotates through {res['sets']} distinct input/output sets "
                                         f"({res['sets'] * bytes_per * n / 2**20:.0f} MiB in total) so every launch reads its inputs from HBM"},
            "solved_fraction": solved_total / total_robots,
            "device_warmup": {"ms": args.device_warm_ms, "warmup_launches_done": res.get("warmup_done"),
                              "what": "untimed launches of the same workload before the K timed steps: the W asked for, at least one per rotating set "
                                      "(capped at 8), then until `ms` of wall time have passed - after the host-side input generation the device "
                                      "needs 10-20 ms of work to be back at its running clocks (profiles/r05_tick_protocol_*.log); "
                                      "`--device-warm-ms 0` times the ramp instead"},
            **({"from_idle": {"value": total_robots * args.steps / wall_idle, "ms_per_step": wall_idle / args.steps * 1e3,
                              "avg_kernel_us": res["cold_from_idle"][1] / args.steps * 1e6, "idle_s": 2.0,
                              "what": "the same K steps after exactly W warm-up steps on a device that sat idle for 2 s first - how rounds 1-4 "
                                      "measured (their timed regions followed the host-side input generation); `value` is the device at its running clocks"}}
               if "cold_from_idle" in res else {}),
            "kernel_src_sha16": sha,
            "warm_cache": {"value": total_robots * args.steps / wall_warm, "ms_per_step": wall_warm / args.steps * 1e3,
                           "avg_kernel_us": res["warm_cache"][1] / args.steps * 1e6,
                           "hbm_GBs": bytes_per * n * args.steps / res["warm_cache"][1] / 1e9,
                           "what": "the same launch replayed on ONE resident set (inputs served by L2 / Infinity Cache)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "qc::balance_kernel", "bytes_per_launch": bytes_per * n,
                         "avg_kernel_us": kernel_s * 1e6,
                         "note": "cold-cache protocol; latency / FP64-VALU bound active-set solve, see DESIGN.md 4"},
        }
        if res["warm"]:
            line["roofline"]["bytes_per_robot"] = ("496 = 388 read + 4 warm-start word read + 100 written + 4 active-set word written; SURVEY 8d's 592 "
                                                   "also counts a 96-byte warm primal, which this solver does not read (it restarts from the active set alone)")
        if res["solved_all_sets"] != res["sets"] * n:
            line["solved_fraction_all_sets"] = res["solved_all_sets"] / (res["sets"] * n)
        if gather_s is not None:
            line["result_gather"] = {"bytes_per_rank": n * 96, "seconds": gather_s, "GBs_per_rank": n * 96 * (world - 1) / gather_s / 1e9,
                                     "what": "all-gather of the [n, 12] GRF blocks after the timed region (not part of value)"}
        attach_pmc(line, cfg, n, sha, ctl.kernel_name, tick=args.tick, kernel_us=kernel_s * 1e6)
        if "gait" in res:
            line["gait_clock"] = res["gait"]
            line["roofline"]["bytes_per_robot"] = (f"{bytes_per:.2f} = 964 (frozen-phase tick) + 8 gait_dt read + 32 advanced phases written + 48 x "
                                                   f"{res['gait']['edge_legs_per_robot_tick']:.5f} stance->swing edges per robot per tick")
        if dist is not None:
            line["ranks"] = {"avg_kernel_us_min": k_min, "avg_kernel_us_max": k_max,
                             "allreduce_us": allreduce_s * 1e6, "backend": dist.get_backend(), "rccl": rccl_info,
                             "what": "slowest / fastest rank's average kernel time (HIP events); one 8-byte all-reduce of the kind that "
                                     "brackets the timed region (barrier + counter reduction are the only collectives: no data-path exchange)"}
        if world > 1 and scaling == "strong":
            # the N = 1 point of this strong-scaling curve, ALWAYS measured in the same run on rank 0's GPU while the other
            # ranks wait at the final barrier: the WHOLE batch on one device, so that the line explains itself
            del res
            torch.cuda.empty_cache()
            r1 = run_config(ctl, q, cfg, total_robots, 0, 10, 10, None, device, protocols=("cold",), warm_ms=args.device_warm_ms)
            line["n1_reference"] = {"robots": total_robots, "value": total_robots * 10 / r1["cold"][0], "avg_kernel_us": r1["cold"][1] / 10 * 1e6,
                                    "solved_fraction": r1["solved"] / total_robots,
                                    "what": "the whole batch on rank 0's GPU alone, after the timed region (not part of value)"}
            line["scaling_efficiency"] = line["value"] / (world * line["n1_reference"]["value"])
            line["scaling_note"] = ("strong scaling of BASELINE.json configs[4]: the same 2,097,152 robots on N GPUs; efficiency = value / "
                                    "(N x n1_reference.value).  A one-GPU line of `bench.py` without --config measures configs[1] (4096 robots) "
                                    "and is NOT the N = 1 point of this curve; n1_reference (or other_configs.config5_n1 of that line) is.")
            res = {"batch": None}
            del r1
        if world == 1 and not args.no_cpu_baseline and not args.tick:
            line["cpu_baseline"] = cpu_baseline(P, res["batch"])  # (configs 3 / 5: the first 4096 robots of the batch)
        del res
        torch.cuda.empty_cache()
        if world == 1 and not args.no_sweep:
            other = {}
            # the sweep's entries do not follow --steps: fifty steps each, whatever the headline was asked for, so that an entry reads the
            # same from a default run and from the driver's 20-step run (the barrier + synchronize bracket costs ~30 us per timed region:
            # 1.5 us per step over 20 steps, 0.6 over 50 - DESIGN.md section 6 "Steps")
            k = SWEEP_STEPS
            for c in (2, 3, 4):
                if c == cfg:
                    continue
                r = run_config(ctl, q, c, CONFIG_N[c], 0, k, 10, None, device, warm_ms=args.device_warm_ms)
                bp = BYTES_PER_ROBOT_WARM if r["warm"] else BYTES_PER_ROBOT_COLD
                other[f"config{c}"] = {"robots": CONFIG_N[c], "solved_fraction": r["solved_all_sets"] / (r["sets"] * CONFIG_N[c]), "sets": r["sets"], "steps": k,
                                       "cold_cache": rates(r, "cold", CONFIG_N[c], k, bp), "warm_cache": rates(r, "warm_cache", CONFIG_N[c], k, bp)}
                attach_pmc(other[f"config{c}"], c, CONFIG_N[c], sha, ctl.kernel_name, kernel_us=other[f"config{c}"]["cold_cache"]["avg_kernel_us"])
                del r
                torch.cuda.empty_cache()
            # the N = 1 point of the config-5 scaling curve: the full 2,097,152-robot batch on this GPU (1 GB: cold by size)
            r = run_config(ctl, q, 5, CONFIG5_TOTAL, 0, 10, 10, None, device, protocols=("cold",), warm_ms=args.device_warm_ms)
            other["config5_n1"] = {"robots": CONFIG5_TOTAL, "solved_fraction": r["solved"] / CONFIG5_TOTAL,
                                   "cold_cache": rates(r, "cold", CONFIG5_TOTAL, 10, BYTES_PER_ROBOT_COLD),
                                   "what": "N = 1 point of the strong-scaling curve that `bench.py --gpus N` (N > 1) continues"}
            attach_pmc(other["config5_n1"], 5, CONFIG5_TOTAL, sha, ctl.kernel_name, kernel_us=other["config5_n1"]["cold_cache"]["avg_kernel_us"])
            # WHICH number of this line is the N = 1 point of the 1/2/4/8 curve (`value` above is configs[1], 4096 robots)
            line["scaling_n1"] = {"workload": "config5", "robots": CONFIG5_TOTAL, "value": other["config5_n1"]["cold_cache"]["QPs_per_s"], "unit": "QPs/s",
                                  "avg_kernel_us": other["config5_n1"]["cold_cache"]["avg_kernel_us"],
                                  "what": "BASELINE.json configs[4] on ONE GPU: the whole 2,097,152-robot batch.  `bench.py --gpus N` (N > 1) shards exactly "
                                          "this batch (strong scaling), so a 1/2/4/8 curve starts HERE, not at this line's `value`"}
            del r
            torch.cuda.empty_cache()
            # ... and the kernel every rank runs at N = 8: rank 0's shard of that batch (262,144 robots, cold by rotation)
            r = run_config(ctl, q, 5, CONFIG5_TOTAL // 8, 0, k, 10, None, device, protocols=("cold",), warm_ms=args.device_warm_ms)
            other["config5_shard8"] = {"robots": CONFIG5_TOTAL // 8, "solved_fraction": r["solved_all_sets"] / (r["sets"] * (CONFIG5_TOTAL // 8)), "sets": r["sets"], "steps": k,
                                       "cold_cache": rates(r, "cold", CONFIG5_TOTAL // 8, k, BYTES_PER_ROBOT_COLD),
                                       "what": "robots [0, 262144) of config 5: what one of eight ranks solves per step (no collective on the data path)"}
            attach_pmc(other["config5_shard8"], 5, CONFIG5_TOTAL // 8, sha, ctl.kernel_name, kernel_us=other["config5_shard8"]["cold_cache"]["avg_kernel_us"])
            del r
            torch.cuda.empty_cache()
            # the dense 12x12 form (a general SPD W is part of the reference's constructor contract, balance_controller.hpp:76-77, 85-88),
            # forced on the reference's diagonal W so that the workloads are the headline's: four lanes per robot with racing
            # strategies at config 2's size, one lane per robot with the Hessian staged in LDS at config 3's
            dense = q.BalanceController.from_params(P, device=device).set_tuning(force_dense=1, **tune)
            for c in (2, 3):
                r = run_config(dense, q, c, CONFIG_N[c], 0, k, 10, None, device, warm_ms=args.device_warm_ms)
                info_d = dense.query_launch(CONFIG_N[c])
                e = {"robots": CONFIG_N[c], "solved_fraction": r["solved_all_sets"] / (r["sets"] * CONFIG_N[c]), "sets": r["sets"], "steps": k,
                     "kernel": dense.kernel_name, "lanes_per_robot": info_d["lanes_per_robot"], "kernel_mode": info_d["mode"],
                     "resident_workgroups": info_d["resident_workgroups"], "lds_bytes": info_d["lds_bytes"],
                     "cold_cache": rates(r, "cold", CONFIG_N[c], k, BYTES_PER_ROBOT_COLD), "warm_cache": rates(r, "warm_cache", CONFIG_N[c], k, BYTES_PER_ROBOT_COLD),
                     "what": "qc_set_tuning force_dense = 1: the formulation a non-diagonal W selects, on this config's robots"}
                attach_pmc(e, c, CONFIG_N[c], sha, dense.kernel_name, kernel_us=e["cold_cache"]["avg_kernel_us"])
                other[f"dense_config{c}"] = e
                del r
                torch.cuda.empty_cache()
            # the ticks built around the QP, under the same cold-cache protocol as the hot path
            for key, c, nn, fz, what, tctl in (
                    ("config2_fused_tick", 2, CONFIG_N[2], True, "joint_q -> forward kinematics -> control() -> clamp(J^T f) -> joint_tau in one launch", ctl),
                    ("config3_full_tick", 3, CONFIG_N[3], "full", "joint states + COM state + gait phases + dt -> complete joint torque command "
                     "(gait clock, FK, contact rule, foothold planner on stance->swing edges, swing trajectories, IK, joint PD, QP, J^T) in one launch; "
                     "every launch is one controller tick (1/300 s) later", ctl),
                    ("full_tick_262144", 3, 262144, "full", "the same complete tick on 262,144 robots (two rounds of workgroups)", ctl),
                    # the tick with a general W (balance_controller.hpp:76-77) - the dense 12x12 form's joint_q kernels
                    ("dense_full_tick_65536", 3, CONFIG_N[3], "full", "the complete tick with the dense 12x12 form (qc_set_tuning force_dense = 1: the "
                     "formulation a non-diagonal W selects), one lane per robot, Hessian planes in LDS", dense)):
                # one protocol for the sweep entry and for the profiled run of the same workload (tools/profile_r.sh -> `bench.py --tick
                # full --steps 50 --warmup 20`): >= 20 warm-up launches AND >= SWEEP_WARM_MS of GPU work before the clock starts - the
                # entry used to start 3 launches (0.5 ms) after seconds of host-side input generation, on a device still ramping up
                r = run_config(tctl, q, c, nn, 0, k, SWEEP_WARMUP, None, device, fused=fz, warm_ms=args.device_warm_ms)
                gait = r.get("gait")
                bp = bytes_per_robot(False, fz, gait["edge_legs_per_robot_tick"] if gait else 0.0)
                tk = "full" if fz == "full" else "fused"
                e = {"robots": nn, "solved_fraction": r["solved_all_sets"] / (r["sets"] * nn), "sets": r["sets"], "steps": k, "bytes_per_robot": bp,
                     "ticks_per_s": nn * k / r["cold"][0],
                     "cold_cache": rates(r, "cold", nn, k, bp), "warm_cache": rates(r, "warm_cache", nn, k, bp), "what": what}
                e["roofline"] = {"bound": "hbm", "achieved": e["cold_cache"]["hbm_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": e["cold_cache"]["hbm_GBs"] / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": bp * nn,
                                 "avg_kernel_us": e["cold_cache"]["avg_kernel_us"]}
                attach_pmc(e, c, nn, sha, tctl.kernel_name, tick=tk, kernel_us=e["cold_cache"]["avg_kernel_us"])
                if tctl is dense:
                    e["kernel"] = dense.kernel_name
                if gait:
                    e["gait_clock"] = gait
                    if not args.no_cpu_baseline and tctl is ctl:
                        e["cpu_baseline"] = cpu_baseline_tick(P, r["batch"])
                    del r
                    torch.cuda.empty_cache()
                    # rounds 3-4's protocol next to it, for continuity: the same tick with the gait phase frozen (no gait_dt)
                    r = run_config(tctl, q, c, nn, 0, k, SWEEP_WARMUP, None, device, fused="full-frozen", protocols=("cold",), warm_ms=args.device_warm_ms)
                    e["frozen_phase"] = {"bytes_per_robot": BYTES_PER_ROBOT_FULL, "cold_cache": rates(r, "cold", nn, k, BYTES_PER_ROBOT_FULL),
                                         "what": "no gait_dt: the phases never move, so GaitScheduler::update and FootPlanner::singleFoot are outside "
                                                 "the timed region (what rounds 3 and 4 reported as the complete tick)"}
                other[key] = e
                del r
                torch.cuda.empty_cache()
            del dense
            line["other_configs"] = other
        if world == 1 and not args.no_sweep:
            line["host_boundary"] = host_boundary(ctl, q)
        if world == 1 and (args.probe_batch_load or not args.no_sweep):
            line["batch_load_probe"] = batch_load_probe(q, P, device)
        # the driver keeps the contract's keys and the last 2 000 characters of this line, so the numbers a reader needs
        # from the long entries above come once more, compactly, as its LAST key (us = average kernel time by HIP events, cold-cache
        # protocol; QPs = whole-batch throughput; hbm = fraction of 8 TB/s on algorithmic bytes; x = counter traffic / algorithmic)
        def _short(e):
            c = e["cold_cache"]
            o = {"n": e["robots"], "us": round(c["avg_kernel_us"], 2), "QPs": float("%.4g" % c["QPs_per_s"]), "hbm": round(c["hbm_GBs"] / HBM_PEAK_GBS, 4)}
            t = e.get("hbm_traffic_bytes_per_launch") or e.get("roofline", {}).get("traffic")
            if t:
                o["x"] = round(t / (c["hbm_GBs"] * 1e9 * c["avg_kernel_us"] * 1e-6), 3)
            if "roofline_valu" in e:
                o["issue"] = round(e["roofline_valu"]["issue_frac"], 3)
            return o
        summary = {"sha": sha, "cfg2": {"n": n, "us": round(kernel_s * 1e6, 2), "QPs": float("%.4g" % line["value"]), "hbm": round(line["roofline"]["frac"], 4),
                                        **({"x": round(line["roofline"]["traffic"] / line["roofline"]["bytes_per_launch"], 3)} if line["roofline"].get("traffic") else {}),
                                        **({"issue": round(line["roofline_valu"]["issue_frac"], 3)} if "roofline_valu" in line else {})}
                   if cfg == 2 and not args.tick else None}
        if "from_idle" in line:
            summary["cfg2_from_idle_QPs"] = float("%.4g" % line["from_idle"]["value"])
        for key, name in (("config3", "cfg3"), ("config4", "cfg4_warm"), ("config5_shard8", "cfg5_shard8"), ("config5_n1", "cfg5_n1"), ("dense_config2", "dense2"),
                          ("dense_config3", "dense3"), ("config2_fused_tick", "fused_tick4096"), ("config3_full_tick", "tick65536"), ("full_tick_262144", "tick262144"),
                          ("dense_full_tick_65536", "dense_tick65536")):
            if key in line.get("other_configs", {}):
                summary[name] = _short(line["other_configs"][key])
        if "batch_load_probe" in line:
            bp_ = line["batch_load_probe"]
            summary["batch_load"] = {"n": bp_["robots"], "us": round(bp_["us"], 1), "hbm": round(bp_["frac"], 4),
                                     **({"x": round(bp_["traffic_ratio_to_algorithmic"], 3), "rocprof_us": round(bp_["avg_kernel_us_rocprof"], 1)} if bp_.get("traffic") else {})}
        if "cpu_baseline" in line:
            summary["cpu_QPs"] = float("%.4g" % line["cpu_baseline"]["value"])
        line["summary"] = summary
        print(json.dumps(line), flush=True)

    if dist is not None:
        # Rank 0 may still be measuring n1_reference (generate the whole batch + 20 launches of it, possibly with a cold
        # first kernel load) while the others are done: they wait for it on the rendezvous store under their OWN limit
        # (QC_BENCH_FINAL_TIMEOUT_S, default 600 s) - not inside a collective, where the 60 s fail-fast timeout that guards
        # group formation would end the run - and only then meet in the last barrier.
        import datetime

        final_limit = float(os.environ.get("QC_BENCH_FINAL_TIMEOUT_S", "600"))
        try:
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("qc_bench_rank0_done", "1")
            else:
                store.wait(["qc_bench_rank0_done"], datetime.timedelta(seconds=final_limit))
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: rank {rank}/{world}: rank 0 did not finish its result line within {final_limit:.0f} s "
                  f"(QC_BENCH_FINAL_TIMEOUT_S) - {type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}", file=sys.stderr, flush=True)
            os._exit(4)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
