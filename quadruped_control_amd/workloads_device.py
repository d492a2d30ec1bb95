"""Config-3 / config-5 inputs generated ON THE DEVICE, per shard ("config 5 ... inputs
generated on-device per shard").

The same counter-based PRNG as workloads.py - splitmix64 of (seed, instance, stream) - evaluated with torch int64
arithmetic on the GPU: two's-complement multiplication and addition wrap exactly like uint64, and the logical right
shifts are arithmetic shifts with the sign-extension bits masked off.  The uniforms, and with them the gait kinds,
phases and contact states, are bit-identical to workloads.config3(); the floating-point part (Rodrigues formula, foot
positions) goes through the device's sin / cos / matmul and agrees with the host generator to the last few ulps
(tests/test_gpu_properties.py::test_on_device_generation_matches_host).  Only bench.py and the tests use it: a rank
of the 8-GPU run builds its 262 144-robot shard (and the rotation sets of the cold-cache protocol) in milliseconds
instead of generating it with numpy and pushing it through PCIe.
"""
from __future__ import annotations

from . import workloads as W

_MASK64 = (1 << 64) - 1


def _i64(v):
    """Python int (mod 2^64) -> the int64 with the same bit pattern."""
    v &= _MASK64
    return v - (1 << 64) if v >= (1 << 63) else v


_C_GOLD = _i64(0x9E3779B97F4A7C15)
_C_M1 = _i64(0xBF58476D1CE4E5B9)
_C_M2 = _i64(0x94D049BB133111EB)
_C_SEED = 0xD1342543DE82EF95
_C_STREAM = 0xC2B2AE3D27D4EB4F


def _lshr(z, k):
    """logical right shift of an int64 tensor"""
    return (z >> k) & ((1 << (64 - k)) - 1)


def _splitmix64(z):
    z = z + _C_GOLD
    z = (z ^ _lshr(z, 30)) * _C_M1
    z = (z ^ _lshr(z, 27)) * _C_M2
    return z ^ _lshr(z, 31)


def uniform(seed, idx, stream, lo=0.0, hi=1.0):
    """U[lo, hi) for the int64 index tensor `idx` - bit-identical to workloads.uniform."""
    import torch

    key = idx * _C_GOLD + _i64(seed * _C_SEED + stream * _C_STREAM)
    u = _lshr(_splitmix64(_splitmix64(key)), 11).to(torch.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


def _uvec(seed, idx, stream0, k, lo, hi):
    import torch

    return torch.stack([uniform(seed, idx, stream0 + j, lo, hi) for j in range(k)], dim=1)


def _rotvec_to_matrix(rv):
    import torch

    n = rv.shape[0]
    th = torch.linalg.norm(rv, dim=1)
    small = th < 1e-12
    ths = torch.where(small, torch.ones_like(th), th)
    k = rv / ths[:, None]
    K = torch.zeros((n, 3, 3), dtype=torch.float64, device=rv.device)
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    s = torch.sin(th)[:, None, None]
    c = (1.0 - torch.cos(th))[:, None, None]
    eye = torch.eye(3, dtype=torch.float64, device=rv.device)[None]
    R = eye + s * K + c * (K @ K)
    return torch.where(small[:, None, None], eye.expand(n, 3, 3), R)


def config3(n, start=0, seed=W.SEEDS[3], device=0):
    """workloads.config3 (mixed 2/3/4-foot contact states from gait schedules) as a dict of device tensors."""
    import torch

    dev = torch.device(device) if isinstance(device, str) else torch.device("cuda", device)  # "cpu" serves the CPU tests
    f64 = dict(dtype=torch.float64, device=dev)
    idx = torch.arange(start, start + n, dtype=torch.int64, device=dev)
    nominal = torch.tensor(W.NOMINAL_FEET_XY, **f64)
    ft_world = torch.zeros((n, 4, 3), **f64)
    ft_world[:, :, :2] = nominal[None] + _uvec(seed, idx, 0, 8, -0.03, 0.03).reshape(n, 4, 2)
    x = torch.tensor([0.0, 0.0, W.STAND_HEIGHT], **f64)[None] + _uvec(seed, idx, 8, 3, -0.03, 0.03)
    R = _rotvec_to_matrix(_uvec(seed, idx, 11, 3, -0.1, 0.1))
    xdot = _uvec(seed, idx, 14, 3, -0.3, 0.3)
    w = _uvec(seed, idx, 17, 3, -0.5, 0.5)
    phi = uniform(seed, idx, 20)
    kind = torch.clamp((uniform(seed, idx, 21) * 3.0).to(torch.int64), max=2)
    offsets = torch.tensor([[0.0, 0.5, 0.5, 0.0], [0.0, 0.5, 0.5, 0.0], [0.0, 0.25, 0.5, 0.75]], **f64)[kind]
    duty = torch.tensor([0.3 / 0.6, 0.8 / 0.98, 0.8 / 0.98], **f64)[kind][:, None]
    phases = torch.fmod(offsets + phi[:, None], 1.0)                 # gait.cpp:113-123
    eps = 1.0e-12                                                     # gait.cpp:125-134 with almost_equal's slack
    stance = ((phases > 0.0) | (phases.abs() < eps)) & ((phases < duty) | ((phases - duty).abs() < eps))
    lift = _uvec(seed, idx, 22, 4, 0.0, 0.08)
    ft_world[:, :, 2] = torch.where(stance, torch.zeros_like(lift), lift)
    one = torch.ones(n, **f64)
    sx = torch.where(uniform(seed, idx, 26) < 0.5, -one, one)
    sy = torch.where(uniform(seed, idx, 27) < 0.5, -one, one)
    sz = torch.where(uniform(seed, idx, 28) < 0.5, -one, one)
    zero = torch.zeros(n, **f64)
    xdot_d = torch.stack([0.2 * sx, 0.1 * sy, zero], dim=1)
    w_d = torch.stack([zero, zero, 0.05 * sz], dim=1)
    x_d = torch.tensor([0.0, 0.0, W.STAND_HEIGHT], **f64)[None].repeat(n, 1)
    R_d = torch.eye(3, **f64)[None].repeat(n, 1, 1)
    feet = torch.einsum("nji,nkj->nki", R, ft_world - x[:, None, :])  # feet_body = Rwb^T (p_world - x)
    c = lambda t: t.contiguous()
    return dict(Rwb=c(R.reshape(n, 9)), Rwb_d=c(R_d.reshape(n, 9)), x=c(x), xdot=c(xdot), w=c(w), x_d=c(x_d), xdot_d=c(xdot_d),
                w_d=c(w_d), feet=c(feet.reshape(n, 12)), stance=c(stance.to(torch.uint8)))


def config5(n=2097152, start=0, seed=W.SEEDS[5], device=0):
    """Config-3 distribution; rank r of G generates [r n / G, (r + 1) n / G) on its own device."""
    return config3(n=n, start=start, seed=seed, device=device)
