"""TEST INFRASTRUCTURE - vectorised, solver-independent KKT certificate for whole batches.

A third, independent restatement of the reference's problem data (numpy + scipy's rotation log, no code shared with
oracle/balance_oracle.c or the device): for every robot it rebuilds the wrench map A and target b
(balance_controller.cpp:126-139, 237-272), evaluates the gradient of the reference's objective
  grad = Q f + c = 2 (A^T S (A f - b) + W f)            (balance_controller.cpp:152-153)
at the world-frame forces f the device returned, and checks the KKT conditions of the reference's constraint rows
(balance_controller.cpp:274-330) foot by foot.  The rows of one foot only involve that foot's force, so the
stationarity condition separates per foot: with n1 = (1,0,-mu), n2 = (-1,0,-mu), n3 = (0,1,-mu), n4 = (0,-1,-mu),
n5 = (0,0,1), n6 = (0,0,-1) the outward normals of |fx| <= mu fz, |fy| <= mu fz, fz <= fzmax, fz >= fzmin,
  grad_i + sum_k lam_k n_k = 0,  lam_k >= 0,  lam_k = 0 on inactive rows
has the closed-form solution lam_x = -s_x grad_x, lam_y = -s_y grad_y, lam_z = s_z (mu (lam_x + lam_y) - grad_z),
where s in {-1, 0, +1} says which side of the axis is active at f (decided here from f itself, by tolerance), and
on an axis without an active row the corresponding gradient expression must vanish.  Swing feet are pinned to 0
by equality rows (balance_controller.cpp:312-316): no stationarity condition, f_i must be exactly 0.
The QP is strictly convex, so a point that passes is THE minimiser the reference's qpOASES call returns.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


def wrench_data(P, b):
    """A [n,6,12], bvec [n,6] of balance_controller.cpp:237-272 for a batch (dict of numpy arrays)."""
    n = b["x"].shape[0]
    R = b["Rwb"].reshape(n, 3, 3)
    Rd = b["Rwb_d"].reshape(n, 3, 3)
    kff = np.asarray(P["kff"], float)
    a = np.asarray(P["kp_p"]) * (b["x_d"] - b["x"]) + np.asarray(P["kd_p"]) * (b["xdot_d"] - b["xdot"])
    a[:, 0] += kff[0] * b["xdot_d"][:, 0]
    a[:, 1] += kff[1] * b["xdot_d"][:, 1]
    a[:, 2] += kff[2] * P["mass"] * 9.81
    e = Rotation.from_matrix(Rd @ np.swapaxes(R, 1, 2)).as_rotvec()  # axis * angle, angle in [0, pi]
    al = np.asarray(P["kp_w"]) * e + np.asarray(P["kd_w"]) * (b["w_d"] - b["w"])
    al[:, 0] += kff[3] * b["w_d"][:, 0]
    al[:, 1] += kff[4] * b["w_d"][:, 1] + kff[5] * b["w_d"][:, 2]  # sic: index 1 (balance_controller.cpp:139)
    r = np.einsum("nij,nkj->nki", R, b["feet"].reshape(n, 4, 3))   # r_i = Rwb p_i
    Iw = R @ np.asarray(P["Ib"], float).reshape(3, 3) @ np.swapaxes(R, 1, 2)
    bv = np.zeros((n, 6))
    bv[:, :3] = P["mass"] * (a + np.array([0.0, 0.0, -9.81]))
    Iwd = np.einsum("nij,nj->ni", Iw, b["w_d"])
    bv[:, 3:] = np.einsum("nij,nj->ni", Iw, al) + np.cross(b["w_d"], Iwd)
    A = np.zeros((n, 6, 12))
    for i in range(4):
        A[:, 0, 3 * i] = A[:, 1, 3 * i + 1] = A[:, 2, 3 * i + 2] = 1.0
        x, y, z = r[:, i, 0], r[:, i, 1], r[:, i, 2]
        A[:, 3, 3 * i + 1], A[:, 3, 3 * i + 2] = -z, y
        A[:, 4, 3 * i], A[:, 4, 3 * i + 2] = z, -x
        A[:, 5, 3 * i], A[:, 5, 3 * i + 1] = -y, x
    return A, bv


def kkt_batch(P, b, grf_body, act_tol=1e-7, chunk=65536):
    """Per-robot certificate of the device result `grf_body` [n,12] (body frame, as returned).
    Returns dict(primal [n] = largest constraint violation in N, stationarity [n] = largest stationarity /
    negative-multiplier residual relative to 1 + |grad|, swing_nonzero [n] bool)."""
    n = b["x"].shape[0]
    primal = np.zeros(n)
    stat = np.zeros(n)
    swing_bad = np.zeros(n, bool)
    S = np.asarray(P["S"], float).reshape(6, 6)
    Wm = np.asarray(P["W"], float).reshape(12, 12)
    mu, fzmin, fzmax = P["mu"], P["fzmin"], P["fzmax"]
    for lo in range(0, n, chunk):
        sl = slice(lo, min(n, lo + chunk))
        bb = {k: v[sl] for k, v in b.items()}
        m = bb["x"].shape[0]
        A, bv = wrench_data(P, bb)
        R = bb["Rwb"].reshape(m, 3, 3)
        fw = -np.einsum("nij,nkj->nki", R, grf_body[sl].reshape(m, 4, 3))  # f_w = -Rwb f_b (balance_controller.cpp:225)
        f = fw.reshape(m, 12)
        u = np.einsum("nij,nj->ni", A, f) - bv
        grad = 2.0 * (np.einsum("nji,nj->ni", A, u @ S.T) + f @ Wm.T)
        gn = 1.0 + np.linalg.norm(grad, axis=1)
        g = grad.reshape(m, 4, 3)
        st = bb["stance"].astype(bool) if bb.get("stance") is not None else np.ones((m, 4), bool)
        fx, fy, fz = fw[..., 0], fw[..., 1], fw[..., 2]
        viol = np.stack([np.abs(fx) - mu * fz, np.abs(fy) - mu * fz, fzmin - fz, fz - fzmax], axis=-1).max(axis=-1)
        primal[sl] = np.where(st, viol, 0.0).max(axis=1)
        swing_bad[sl] = (np.where(st[..., None], 0.0, np.abs(fw)) != 0.0).any(axis=(1, 2))
        tol = act_tol * (1.0 + np.abs(fz) * mu)
        sx = np.where(mu * fz - fx <= tol, 1, 0) - np.where(mu * fz + fx <= tol, 1, 0)
        sy = np.where(mu * fz - fy <= tol, 1, 0) - np.where(mu * fz + fy <= tol, 1, 0)
        sz = np.where(fzmax - fz <= act_tol * (1.0 + fzmax), 1, 0) - np.where(fz - fzmin <= act_tol * (1.0 + fzmin), 1, 0)
        lx = -sx * g[..., 0]
        ly = -sy * g[..., 1]
        # an axis without an active row: its gradient expression must vanish; with one: the multiplier must be >= 0
        rx = np.where(sx == 0, np.abs(g[..., 0]), np.maximum(0.0, -lx))
        ry = np.where(sy == 0, np.abs(g[..., 1]), np.maximum(0.0, -ly))
        lxa = np.where(sx == 0, 0.0, lx)
        lya = np.where(sy == 0, 0.0, ly)
        ez = mu * (lxa + lya) - g[..., 2]
        rz = np.where(sz == 0, np.abs(ez), np.maximum(0.0, -sz * ez))
        res = np.where(st, np.maximum(np.maximum(rx, ry), rz), 0.0).max(axis=1)
        stat[sl] = res / gn
    return dict(primal=primal, stationarity=stat, swing_nonzero=swing_bad)


def assert_kkt(P, b, grf_body, primal_tol=1e-7, stat_tol=1e-8):
    """KKT on EVERY robot of the batch; returns the worst stationarity residual."""
    c = kkt_batch(P, b, grf_body)
    assert not c["swing_nonzero"].any(), "swing feet must carry exactly zero force"
    worst_p = float(c["primal"].max())
    assert worst_p < primal_tol, (worst_p, int(c["primal"].argmax()))
    worst = float(c["stationarity"].max())
    assert worst < stat_tol, (worst, int(c["stationarity"].argmax()))
    return worst
