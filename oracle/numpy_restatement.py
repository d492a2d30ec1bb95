"""TEST INFRASTRUCTURE - CPU restatement (numpy/scipy) of the reference hot path.

This file is part of the *oracle*: it is imported only by tests/, by
tests/golden/make_golden.py and by bench.py's cpu_baseline leg.  It is never
on the product path (quadruped_control_amd/ does not import it).

PARITY UNPINNED BY REFERENCE FIXTURES.  The reference holds no test, golden
vector or fixture for BalanceController::control(), and its QP solve lives in
qpOASES (README.md:102 "qpOASES master (SHAID: 326a651)"), which is absent
from /root/reference and from this image, as are Armadillo, Drake/Eigen and
ROS.  What pins this restatement instead:
  * the assembly (everything except the QP solve) is first-party reference
    code and is restated line by line below with file:line citations;
  * the QP is strictly convex (W > 0) so it has ONE minimiser: any point that
    passes the KKT certificate `kkt_certificate()` is the point qpOASES
    returns up to its termination tolerance (~1e-9);
  * the solve is done by a THIRD-PARTY exact active-set code that is present
    in the image (scipy.optimize.nnls, Lawson-Hanson) through the classical
    least-distance-programming transformation, then polished on the
    identified active set, then certified;
  * closed-form known answers (KAT1..KAT4).

All citations are relative to /root/reference/quadruped_controller/.
BC.cpp = src/quadruped_controller/balance_controller.cpp
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import nnls

LEG_NAMES = ("RL", "FL", "RR", "FR")  # src/commander_node.cpp:61 (leg order)
NUM_EQ = 6     # BC.hpp:156
NUM_VAR = 12   # BC.hpp:157
NUM_CON = 20   # BC.hpp:158


# --------------------------------------------------------------------------
# parameters (constructor arguments, BC.hpp:85-88 / BC.cpp:70-96)
# --------------------------------------------------------------------------
def cheetah_params(mu=0.8):
    """Constants of quadruped_simulation/config/mit_cheetah_config.yaml:66-99
    as loaded by src/commander_node.cpp:289-334."""
    return dict(
        mu=float(mu), mass=11.0, fzmin=10.0, fzmax=120.0,
        Ib=np.diag([0.011253, 0.036203, 0.042673]),
        S=np.diag([1.0, 1.0, 1.0, 10.0, 10.0, 5.0]),
        W=np.eye(12) * 1e-5,
        kff=np.array([0.0, 0.0, 0.15, 0.0, 0.0, 0.0]),
        kp_p=np.full(3, 100.0), kd_p=np.full(3, 50.0),
        kp_w=np.full(3, 5000.0), kd_w=np.full(3, 500.0),
    )


# --------------------------------------------------------------------------
# math helpers on the path
# --------------------------------------------------------------------------
def skew_symmetric(v):
    """src/quadruped_controller/math/rigid3d.cpp:61-74."""
    x, y, z = v
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def angle_axis_total(R):
    """Rotation3d(mat).angleAxisTotal(): rigid3d.cpp:177-179,198-203.

    Drake RotationMatrix::ToAngleAxis() builds an Eigen::AngleAxisd from the
    matrix, i.e. matrix -> quaternion -> angle-axis (Drake v0.26.0 / bundled
    Eigen 3.3, README.md:101; neither is vendored - published algorithm
    restated): trace>0 branch or largest-diagonal pivot for the quaternion;
    angle = 2*atan2(|q_v|, |q_w|) in [0, pi]; axis = q_v / (+-|q_v|);
    zero rotation -> angle 0, axis (1,0,0).
    """
    m = np.asarray(R, dtype=np.float64)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)  # x y z w
    if t > 0.0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    n = np.linalg.norm(q[:3])
    if n != 0.0:
        angle = 2.0 * np.arctan2(n, abs(q[3]))
        if q[3] < 0.0:
            n = -n
        axis = q[:3] / n
    else:
        angle = 0.0
        axis = np.array([1.0, 0.0, 0.0])
    return axis * angle


# --------------------------------------------------------------------------
# assembly: BC.cpp:98-161 + 237-330
# --------------------------------------------------------------------------
def friction_cone_constraint(mu):
    """BC.cpp:274-292 -> C (20x12)."""
    Cf = np.array([[1.0, 0.0, -mu], [0.0, 1.0, -mu], [0.0, 1.0, mu],
                   [1.0, 0.0, mu], [0.0, 0.0, 1.0]])
    C = np.zeros((NUM_CON, NUM_VAR))
    for i in range(4):
        C[5 * i:5 * i + 5, 3 * i:3 * i + 3] = Cf
    return C


def friction_cone_bounds(stance, fzmin, fzmax):
    """BC.cpp:294-330 -> lbC, ubC (20 each). stance[i] = LegState (1 stance)."""
    upper, lower = 1000000.0, -1000000.0
    lbf = np.array([lower, lower, 0.0, 0.0, fzmin])
    ubf = np.array([0.0, 0.0, upper, upper, fzmax])
    lb = np.zeros(NUM_CON)
    ub = np.zeros(NUM_CON)
    for i in range(4):
        if stance[i]:
            lb[5 * i:5 * i + 5] = lbf
            ub[5 * i:5 * i + 5] = ubf
    return lb, ub


def dynamics(P, ft_p, Rwb, xddot_d, w_d, wdot_d):
    """BC.cpp:237-272 (`x` argument unused there, so dropped here).
    ft_p is 3x4 (columns RL,FL,RR,FR, body frame)."""
    com_ft_p = Rwb @ ft_p                                  # :244-248
    Iw = Rwb @ P["Ib"] @ Rwb.T                             # :251
    A = np.zeros((NUM_EQ, NUM_VAR))
    for i in range(4):
        A[0:3, 3 * i:3 * i + 3] = np.eye(3)                # :254-257
        A[3:6, 3 * i:3 * i + 3] = skew_symmetric(com_ft_p[:, i])  # :259-262
    g = np.array([0.0, 0.0, -9.81])                        # BC.cpp:76
    b = np.zeros(NUM_EQ)
    b[0:3] = P["mass"] * (xddot_d + g)                     # :265
    b[3:6] = Iw @ wdot_d + np.cross(w_d, Iw @ w_d)         # :269
    return A, b


def assemble(P, Rwb, Rwb_d, x, xdot, w, x_d, xdot_d, w_d, feet, stance):
    """Everything control() hands to qpOASES: H(12x12), g(12), C(20x12),
    lbC, ubC - BC.cpp:107-161.  feet is 4x3 (rows RL,FL,RR,FR, body frame)."""
    Rwb = np.asarray(Rwb, float); Rwb_d = np.asarray(Rwb_d, float)
    ft_p = np.asarray(feet, float).reshape(4, 3).T          # :107-116
    lb, ub = friction_cone_bounds(stance, P["fzmin"], P["fzmax"])  # :119
    kff = P["kff"]
    xddot_d = P["kp_p"] * (x_d - x) + P["kd_p"] * (xdot_d - xdot)  # :126
    xddot_d[0] += kff[0] * xdot_d[0]                        # :127
    xddot_d[1] += kff[1] * xdot_d[1]                        # :128
    xddot_d[2] += kff[2] * P["mass"] * 9.81                 # :129
    R_error = Rwb_d @ Rwb.T                                 # :133
    wdot_d = P["kp_w"] * angle_axis_total(R_error) + P["kd_w"] * (w_d - w)  # :136
    wdot_d[0] += kff[3] * w_d[0]                            # :137
    wdot_d[1] += kff[4] * w_d[1]                            # :138
    wdot_d[1] += kff[5] * w_d[2]                            # :139 (sic: index 1)
    A, b = dynamics(P, ft_p, Rwb, xddot_d, w_d, wdot_d)     # :144
    Q = 2.0 * (A.T @ P["S"] @ A + P["W"])                   # :152
    c = -2.0 * A.T @ P["S"] @ b                             # :153
    C = friction_cone_constraint(P["mu"])                   # BC.cpp:88, :327
    return dict(H=Q, g=c, C=C, lb=lb, ub=ub, A=A, b=b)


def output_transform(Rwb, fw, stance):
    """BC.cpp:218-232: body-frame, negated, stance legs only.  Returned as a
    4x3 array with swing-leg rows = 0 plus the dict the reference returns."""
    out = np.zeros((4, 3))
    fmap = {}
    for i, name in enumerate(LEG_NAMES):
        if stance[i]:
            fb = -1.0 * Rwb.T @ fw[3 * i:3 * i + 3]
            out[i] = fb
            fmap[name] = fb
    return out, fmap


# --------------------------------------------------------------------------
# the QP:  min 1/2 f'Hf + g'f   s.t.  lb <= C f <= ub      (BC.cpp:177-210)
# solved independently of any code in this repo's product path
# --------------------------------------------------------------------------
def _one_sided(C, lb, ub):
    """Rows G f <= h equivalent to lb <= C f <= ub (both sides kept)."""
    G = np.vstack([C, -C])
    h = np.concatenate([ub, -lb])
    return G, h


def solve_qp_ldp(H, g, C, lb, ub):
    """Lawson & Hanson least-distance programming through scipy NNLS, then an
    exact polish on the active set NNLS identified.  Returns f (12,)."""
    G, h = _one_sided(C, lb, ub)
    L = np.linalg.cholesky(H)
    Linv_g = np.linalg.solve(L, g)
    Gt = np.linalg.solve(L, G.T).T          # G L^-T
    ht = h + Gt @ Linv_g                    # h + G H^-1 g
    # LDP: min |z| s.t. (-Gt) z >= -ht
    E = np.vstack([-Gt.T, -ht[None, :]])
    e = np.zeros(E.shape[0]); e[-1] = 1.0
    scale = np.maximum(np.linalg.norm(E, axis=0), 1e-300)
    u, _ = nnls(E / scale, e, maxiter=2000)
    u = u / scale
    r = E @ u - e
    if abs(r[-1]) < 1e-14:
        raise RuntimeError("LDP: constraints infeasible")
    z = -r[:-1] / r[-1]
    f = np.linalg.solve(L.T, z - Linv_g)
    # polish: equality-constrained solve on the identified active rows
    act = np.flatnonzero(u > 0)
    f2 = _eqp(H, g, G[act], h[act])
    if f2 is not None and np.all(G @ f2 - h <= 1e-9 * (1 + np.abs(h))):
        f = f2
    return f


def _eqp(H, g, N, hN):
    """min 1/2 f'Hf+g'f s.t. N f = hN, by null-space of an SVD-reduced N."""
    if N.shape[0] == 0:
        return -np.linalg.solve(H, g)
    U, s, Vt = np.linalg.svd(N, full_matrices=True)
    rk = int(np.sum(s > 1e-10 * max(s[0], 1.0)))
    Y = Vt[:rk].T
    Z = Vt[rk:].T
    fy = Y @ ((U[:, :rk].T @ hN) / s[:rk])
    if Z.shape[1] == 0:
        return fy
    y = np.linalg.solve(Z.T @ H @ Z, -Z.T @ (H @ fy + g))
    return fy + Z @ y


def kkt_certificate(H, g, C, lb, ub, f, act_tol=1e-7):
    """Solver-independent optimality certificate.  Returns dict with
    `primal` (max bound violation), `stationarity` (|Hf+g+N'lam| minimised
    over lam>=0 on the rows that are active at f, relative to |Hf+g|+1)."""
    G, h = _one_sided(C, lb, ub)
    res = G @ f - h
    primal = float(np.max(res))
    act = np.flatnonzero(res >= -act_tol * (1 + np.abs(h)))
    grad = H @ f + g
    if act.size == 0:
        stat = float(np.linalg.norm(grad))
    else:
        N = G[act]
        nrm = np.maximum(np.linalg.norm(N, axis=1), 1e-300)
        _, stat = nnls((N / nrm[:, None]).T, -grad, maxiter=2000)
        stat = float(stat)
    return dict(primal=primal, stationarity=stat / (1.0 + float(np.linalg.norm(grad))),
                n_active=int(act.size))


def control(P, Rwb, Rwb_d, x, xdot, w, x_d, xdot_d, w_d, feet, stance=(1, 1, 1, 1)):
    """BalanceController::control() end to end (BC.cpp:98-235).  Returns
    (grf_body 4x3 with zeros on swing legs, ForceMap dict, world forces 12)."""
    qp = assemble(P, Rwb, Rwb_d, np.asarray(x, float), np.asarray(xdot, float),
                  np.asarray(w, float), np.asarray(x_d, float),
                  np.asarray(xdot_d, float), np.asarray(w_d, float), feet, stance)
    fw = solve_qp_ldp(qp["H"], qp["g"], qp["C"], qp["lb"], qp["ub"])
    out, fmap = output_transform(np.asarray(Rwb, float), fw, stance)
    return out, fmap, fw, qp
