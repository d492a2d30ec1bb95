"""Seeded synthetic inputs for the BASELINE.json configs.

Host-side numpy only.  Everything is derived from a counter-based PRNG
(splitmix64 of (seed, instance, stream)) so any sub-range of a batch can be
generated independently - a rank generates exactly its shard.

A batch is a dict of C-contiguous float64 arrays shaped like the arguments of
the reference's control() (BC.hpp:104-107), one row per robot:
  Rwb, Rwb_d [n,9] row-major; x, xdot, w, x_d, xdot_d, w_d [n,3];
  feet [n,12] (RL,FL,RR,FR body frame); stance [n,4] uint8 (LegState).
"""
from __future__ import annotations

import numpy as np

from .gait import LEG_NAMES, leg_state_from_phase  # noqa: F401

SEEDS = {2: 0x5EED0002, 3: 0x5EED0003, 4: 0x5EED0004, 5: 0x5EED0005}
NOMINAL_FEET_XY = np.array([[-0.196, 0.127], [0.196, 0.127],
                            [-0.196, -0.127], [0.196, -0.127]])  # RL FL RR FR
STAND_HEIGHT = 0.26   # commander_node.cpp:354
CONTROL_DT = 1.0 / 300.0  # mit_cheetah_config.yaml:3

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform(seed, idx, stream, lo=0.0, hi=1.0):
    """U[lo,hi) for instance indices `idx` (uint64 array) and integer stream."""
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) * np.uint64(0xD1342543DE82EF95)
               + idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
               + np.uint64(stream) * np.uint64(0xC2B2AE3D27D4EB4F))
    u = (_splitmix64(_splitmix64(key)) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def _uvec(seed, idx, stream0, k, lo, hi):
    return np.stack([uniform(seed, idx, stream0 + j, lo, hi) for j in range(k)], axis=1)


def rotvec_to_matrix(rv):
    """Rodrigues, batched: rv [n,3] -> R [n,3,3]."""
    th = np.linalg.norm(rv, axis=1)
    small = th < 1e-12
    ths = np.where(small, 1.0, th)
    k = rv / ths[:, None]
    K = np.zeros((rv.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    s = np.sin(th)[:, None, None]
    c = (1.0 - np.cos(th))[:, None, None]
    R = np.eye(3)[None] + s * K + c * (K @ K)
    R[small] = np.eye(3)
    return R


def _state_batch(seed, idx):
    """Common state distribution of configs 2-5."""
    n = idx.shape[0]
    ft_world = np.zeros((n, 4, 3))
    ft_world[:, :, :2] = NOMINAL_FEET_XY[None] + _uvec(seed, idx, 0, 8, -0.03, 0.03).reshape(n, 4, 2)
    x = np.array([0.0, 0.0, STAND_HEIGHT])[None] + _uvec(seed, idx, 8, 3, -0.03, 0.03)
    R = rotvec_to_matrix(_uvec(seed, idx, 11, 3, -0.1, 0.1))
    xdot = _uvec(seed, idx, 14, 3, -0.3, 0.3)
    w = _uvec(seed, idx, 17, 3, -0.5, 0.5)
    return ft_world, x, R, xdot, w


def _pack(R, R_d, x, xdot, w, x_d, xdot_d, w_d, ft_world, stance):
    n = x.shape[0]
    # feet_body = Rwb^T (p_world - x)
    feet = np.einsum("nji,nkj->nki", R, ft_world - x[:, None, :])
    c = np.ascontiguousarray
    return dict(Rwb=c(R.reshape(n, 9)), Rwb_d=c(R_d.reshape(n, 9)), x=c(x), xdot=c(xdot), w=c(w),
                x_d=c(x_d), xdot_d=c(xdot_d), w_d=c(w_d), feet=c(feet.reshape(n, 12)),
                stance=c(stance.astype(np.uint8)))


def config1():
    """Single robot, 4-foot stance, identity orientation, flat ground (KAT1)."""
    n = 1
    ft_world = np.zeros((n, 4, 3)); ft_world[:, :, :2] = NOMINAL_FEET_XY
    x = np.array([[0.0, 0.0, STAND_HEIGHT]])
    I = np.eye(3)[None].repeat(n, 0)
    z = np.zeros((n, 3))
    return _pack(I, I.copy(), x, z, z.copy(), x.copy(), z.copy(), z.copy(), ft_world, np.ones((n, 4)))


def config2(n=4096, start=0, seed=SEEDS[2]):
    """Randomised COM poses/velocities, all 4 feet in contact."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    ft_world, x, R, xdot, w = _state_batch(seed, idx)
    x_d = np.tile(np.array([0.0, 0.0, STAND_HEIGHT]), (n, 1))
    R_d = np.tile(np.eye(3), (n, 1, 1))
    z = np.zeros((n, 3))
    return _pack(R, R_d, x, xdot, w, x_d, z, z.copy(), ft_world, np.ones((n, 4)))


def config3(n=65536, start=0, seed=SEEDS[3]):
    """Mixed contact states (2/3/4 feet) drawn from gait schedules."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    ft_world, x, R, xdot, w = _state_batch(seed, idx)
    phi = uniform(seed, idx, 20)
    kind = np.minimum((uniform(seed, idx, 21) * 3.0).astype(np.int64), 2)
    # kind 0: trot, commander defaults t_stance=t_swing=0.3 (commander_node.cpp:245-246)
    # kind 1: trot, yaml 0.8/0.18 (mit_cheetah_config.yaml:17-20)
    # kind 2: walk offsets [0,.25,.5,.75], 0.8/0.18
    offsets = np.array([[0.0, 0.5, 0.5, 0.0], [0.0, 0.5, 0.5, 0.0], [0.0, 0.25, 0.5, 0.75]])[kind]
    stance_phase = np.array([0.3 / 0.6, 0.8 / 0.98, 0.8 / 0.98])[kind]
    phases = np.fmod(offsets + phi[:, None], 1.0)            # gait.cpp:113-123
    stance = leg_state_from_phase(phases, stance_phase[:, None])  # gait.cpp:125-134
    lift = _uvec(seed, idx, 22, 4, 0.0, 0.08)                # swing height, yaml:19
    ft_world[:, :, 2] = np.where(stance == 1, 0.0, lift)
    sx = np.where(uniform(seed, idx, 26) < 0.5, -1.0, 1.0)
    sy = np.where(uniform(seed, idx, 27) < 0.5, -1.0, 1.0)
    sz = np.where(uniform(seed, idx, 28) < 0.5, -1.0, 1.0)
    xdot_d = np.stack([0.2 * sx, 0.1 * sy, np.zeros(n)], axis=1)  # teleop_ps4_walking.yaml:6-12
    w_d = np.stack([np.zeros(n), np.zeros(n), 0.05 * sz], axis=1)
    x_d = np.tile(np.array([0.0, 0.0, STAND_HEIGHT]), (n, 1))
    R_d = np.tile(np.eye(3), (n, 1, 1))
    return _pack(R, R_d, x, xdot, w, x_d, xdot_d, w_d, ft_world, stance)


def config4(n=262144, start=0, seed=SEEDS[4]):
    """Two consecutive ticks (dt = 1/300 s) of the config-2 distribution.
    Returns (tick0, tick1); tick1 is meant to be warm-started from tick0."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    ft_world, x, R, xdot, w = _state_batch(seed, idx)
    x_d = np.tile(np.array([0.0, 0.0, STAND_HEIGHT]), (n, 1))
    R_d = np.tile(np.eye(3), (n, 1, 1))
    z = np.zeros((n, 3))
    st = np.ones((n, 4))
    t0 = _pack(R, R_d, x, xdot, w, x_d, z, z.copy(), ft_world, st)
    x1 = x + xdot * CONTROL_DT
    R1 = R @ rotvec_to_matrix(w * CONTROL_DT)
    t1 = _pack(R1, R_d, x1, xdot, w, x_d, z.copy(), z.copy(), ft_world, st)
    return t0, t1


def config5(n=2097152, start=0, seed=SEEDS[5]):
    """Config-3 distribution; rank r of G generates [r*n/G, (r+1)*n/G)."""
    return config3(n=n, start=start, seed=seed)


def slice_batch(batch, lo, hi):
    return {k: np.ascontiguousarray(v[lo:hi]) for k, v in batch.items()}


def shard_bounds(n, rank, world):
    """Contiguous shard [lo,hi) of the batch axis owned by `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---- inputs for the fused tick (forward kinematics -> control -> J^T torque) ----
NOMINAL_JOINTS = np.array([0.0, 0.8, -1.6])  # hip, thigh, calf: feet ~0.3 m below the hips


def with_joint_angles(batch, seed=0x5EED0006, start=0):
    """Replace the `feet` of a batch by joint angles: joint_q [n,12] (RL,FL,RR,FR x hip,thigh,calf),
    nominal stance +- 0.25 rad per joint.  The matching foot positions are whatever the
    reference's forward kinematics (kinematics.cpp:81-103) makes of them."""
    n = batch["x"].shape[0]
    idx = np.arange(start, start + n, dtype=np.uint64)
    q = np.tile(NOMINAL_JOINTS, 4)[None] + _uvec(seed, idx, 40, 12, -0.25, 0.25)
    out = {k: v for k, v in batch.items() if k != "feet"}
    out["joint_q"] = np.ascontiguousarray(q)
    return out


def with_swing_references(batch, seed=0x5EED0007, start=0):
    """Add swing-leg references to a batch that already holds joint_q: world-frame reference foot
    positions / velocities (what FootTrajectoryManager::referenceState returns) placed near the current
    feet, and measured joint velocities.  Positions are built so that the frame change of
    commander_node.cpp:492 (Rwb^T pos - x) lands on a reachable body-frame target."""
    from . import gait  # noqa: F401

    n = batch["x"].shape[0]
    idx = np.arange(start, start + n, dtype=np.uint64)
    q_t = batch["joint_q"] + _uvec(seed, idx, 60, 12, -0.15, 0.15)      # target joint angles
    hip = np.array([[-0.196, 0.05, 0.0], [0.196, 0.05, 0.0], [-0.196, -0.05, 0.0], [0.196, -0.05, 0.0]])
    links = np.array([[0.077, -0.211, -0.230]] * 2 + [[-0.077, -0.211, -0.230]] * 2)
    qq = q_t.reshape(n, 4, 3)
    t1, t2, t3 = qq[..., 0], qq[..., 1], qq[..., 2]
    l1, l2, l3 = links[None, :, 0], links[None, :, 1], links[None, :, 2]
    pb = np.stack([l2 * np.sin(t2) + l3 * np.sin(t2 + t3) + hip[None, :, 0],
                   l1 * np.cos(t1) - l2 * np.sin(t1) * np.cos(t2) - l3 * np.sin(t1) * np.cos(t2 + t3) + hip[None, :, 1],
                   l1 * np.sin(t1) + l2 * np.cos(t1) * np.cos(t2) + l3 * np.cos(t1) * np.cos(t2 + t3) + hip[None, :, 2]], axis=-1)
    R = batch["Rwb"].reshape(n, 3, 3)
    pos = np.einsum("nij,nkj->nki", R, pb + batch["x"][:, None, :])      # pos = Rwb (p_b + x)
    vel = _uvec(seed, idx, 80, 12, -0.5, 0.5).reshape(n, 4, 3)
    out = dict(batch)
    out["swing_pos"] = np.ascontiguousarray(pos.reshape(n, 12))
    out["swing_vel"] = np.ascontiguousarray(vel.reshape(n, 12))
    out["joint_qdot"] = np.ascontiguousarray(_uvec(seed, idx, 100, 12, -2.0, 2.0))
    return out
