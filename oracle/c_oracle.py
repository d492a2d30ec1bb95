"""TEST INFRASTRUCTURE - ctypes view of oracle/liboracle.so (balance_oracle.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("QC_ORACLE_SO") or os.path.join(_HERE, "liboracle.so")  # QC_ORACLE_SO: sanitizer build (make -C oracle sanitize)


class OracleParams(C.Structure):
    _fields_ = [("mu", C.c_double), ("mass", C.c_double), ("fzmin", C.c_double), ("fzmax", C.c_double),
                ("Ib", C.c_double * 9), ("S", C.c_double * 36), ("W", C.c_double * 144),
                ("kff", C.c_double * 6), ("kp_p", C.c_double * 3), ("kd_p", C.c_double * 3),
                ("kp_w", C.c_double * 3), ("kd_w", C.c_double * 3), ("max_iter", C.c_int)]


class OracleKinematics(C.Structure):
    _fields_ = [("hip", C.c_double * 12), ("links", C.c_double * 12), ("tau_min", C.c_double), ("tau_max", C.c_double),
                ("jc_kff", C.c_double * 3), ("jc_kp", C.c_double * 3), ("jc_kd", C.c_double * 3),
                ("planner_hip", C.c_double * 12), ("planner_k", C.c_double), ("swing_height", C.c_double),
                ("t_swing", C.c_double), ("t_stance", C.c_double)]


class OracleSwingState(C.Structure):
    _fields_ = [("leg_state", C.c_int * 4), ("has_traj", C.c_int * 4), ("p_start", C.c_double * 12), ("p_final", C.c_double * 12)]


def build(force=False):
    src = os.path.join(_HERE, "balance_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_qp_solve.restype = C.c_int
        _lib.oracle_control.restype = C.c_int
    return _lib


def set_refine(on):
    """1 (default): the C oracle recomputes every accepted point in long double; 0: plain double (timing runs)."""
    lib().oracle_set_refine(C.c_int(1 if on else 0))


def make_params(P, max_iter=200):
    p = OracleParams()
    p.mu, p.mass, p.fzmin, p.fzmax = P["mu"], P["mass"], P["fzmin"], P["fzmax"]
    for name, k in (("Ib", 9), ("S", 36), ("W", 144), ("kff", 6), ("kp_p", 3), ("kd_p", 3), ("kp_w", 3), ("kd_w", 3)):
        arr = np.ascontiguousarray(np.asarray(P[name], dtype=np.float64).reshape(-1))
        assert arr.size == k, name
        getattr(p, name)[:] = arr.tolist()
    p.max_iter = max_iter
    return p


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def assemble(P, Rwb, Rwb_d, x, xdot, w, x_d, xdot_d, w_d, feet, stance):
    p = make_params(P)
    a = [np.ascontiguousarray(np.asarray(v, np.float64).reshape(-1)) for v in (Rwb, Rwb_d, x, xdot, w, x_d, xdot_d, w_d, feet)]
    st = np.ascontiguousarray(np.asarray(stance, np.uint8))
    H = np.zeros((12, 12)); g = np.zeros(12); Cm = np.zeros((20, 12)); lb = np.zeros(20); ub = np.zeros(20)
    A = np.zeros((6, 12)); b = np.zeros(6)
    lib().oracle_assemble(C.byref(p), *[_dp(v) for v in a], st.ctypes.data_as(C.POINTER(C.c_ubyte)),
                          _dp(H), _dp(g), _dp(Cm), _dp(lb), _dp(ub), _dp(A), _dp(b))
    return dict(H=H, g=g, C=Cm, lb=lb, ub=ub, A=A, b=b)


def qp_solve(H, g, Cm, lb, ub, max_iter=200):
    f = np.zeros(12); lam = np.zeros(20); it = C.c_int(0)
    H, g, Cm, lb, ub = [np.ascontiguousarray(v, np.float64) for v in (H, g, Cm, lb, ub)]
    st = lib().oracle_qp_solve(_dp(H), _dp(g), _dp(Cm), _dp(lb), _dp(ub), C.c_int(max_iter), _dp(f), _dp(lam), C.byref(it))
    return st, f, lam, it.value


def kkt(H, g, Cm, lb, ub, f, lam):
    s, p, d = C.c_double(), C.c_double(), C.c_double()
    H, g, Cm, lb, ub, f, lam = [np.ascontiguousarray(v, np.float64) for v in (H, g, Cm, lb, ub, f, lam)]
    lib().oracle_kkt(_dp(H), _dp(g), _dp(Cm), _dp(lb), _dp(ub), _dp(f), _dp(lam), C.byref(s), C.byref(p), C.byref(d))
    return s.value, p.value, d.value


def angle_axis_total(R):
    R = np.ascontiguousarray(np.asarray(R, np.float64).reshape(-1)); o = np.zeros(3)
    lib().oracle_angle_axis_total(_dp(R), _dp(o))
    return o


def control_batch(P, batch, threads=1, max_iter=200):
    """batch: dict of arrays as produced by quadruped_control_amd.workloads."""
    p = make_params(P, max_iter)
    n = batch["x"].shape[0]
    names = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "feet")
    a = [np.ascontiguousarray(batch[k], np.float64) for k in names]
    st = np.ascontiguousarray(batch["stance"], np.uint8)
    grf = np.zeros((n, 12)); status = np.zeros(n, np.int32); iters = np.zeros(n, np.int32)
    lib().oracle_control_batch(C.byref(p), C.c_long(n), *[_dp(v) for v in a], st.ctypes.data_as(C.POINTER(C.c_ubyte)),
                               _dp(grf), status.ctypes.data_as(C.POINTER(C.c_int)),
                               iters.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(threads))
    return grf, status, iters


def default_kinematics():
    k = OracleKinematics()
    lib().oracle_default_kinematics(C.byref(k))
    return k


def leg_fk(leg, q, kin=None):
    kin = kin or default_kinematics()
    q = np.ascontiguousarray(q, np.float64); p = np.zeros(3)
    lib().oracle_leg_fk(C.byref(kin), C.c_int(leg), _dp(q), _dp(p))
    return p


def leg_jacobian(leg, q, kin=None):
    kin = kin or default_kinematics()
    q = np.ascontiguousarray(q, np.float64); J = np.zeros((3, 3))
    lib().oracle_leg_jacobian(C.byref(kin), C.c_int(leg), _dp(q), _dp(J))
    return J


def tick_batch(P, batch, kin=None, threads=1, max_iter=200):
    """FK -> control() -> J^T f -> clamp for n robots; batch holds 'joint_q' [n,12]."""
    kin = kin or default_kinematics()
    p = make_params(P, max_iter)
    n = batch["x"].shape[0]
    names = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "joint_q")
    a = [np.ascontiguousarray(batch[k], np.float64) for k in names]
    st = np.ascontiguousarray(batch["stance"], np.uint8)
    feet = np.zeros((n, 12)); grf = np.zeros((n, 12)); tau = np.zeros((n, 12)); status = np.zeros(n, np.int32)
    lib().oracle_tick_batch(C.byref(p), C.byref(kin), C.c_long(n), *[_dp(v) for v in a], st.ctypes.data_as(C.POINTER(C.c_ubyte)),
                            _dp(feet), _dp(grf), _dp(tau), status.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(threads))
    return dict(feet=feet, grf_body=grf, joint_tau=tau, status=status)


def leg_ik(leg, p, kin=None):
    kin = kin or default_kinematics()
    p = np.ascontiguousarray(p, np.float64); q = np.zeros(3)
    lib().oracle_leg_ik(C.byref(kin), C.c_int(leg), _dp(p), _dp(q))
    return q


def pinv3(J):
    """oracle_pinv3: the pseudo-inverse legJacobianInverse falls back to (kinematics.cpp:196)."""
    J = np.ascontiguousarray(J, np.float64).reshape(9); out = np.zeros(9)
    lib().oracle_pinv3.restype = C.c_int
    ok = lib().oracle_pinv3(_dp(J), _dp(out))
    return out.reshape(3, 3), bool(ok)


def pinv3_band(J):
    """oracle_pinv3_band: the DEVICE's pseudo-inverse restated - the rank by its complete-pivoting rule (at most 2), the values from
    the SVD truncated to it.  Not what oracle_swing_torque uses (that is arma::pinv's own rule, pinv3); kept to put numbers on where
    the two differ (tests/test_oracle_cpu.py).  Returns (pinv, rank)."""
    J = np.ascontiguousarray(J, np.float64).reshape(9); out = np.zeros(9)
    lib().oracle_pinv3_band.restype = C.c_int
    lib().oracle_cp_rank3.restype = C.c_int
    ok = lib().oracle_pinv3_band(_dp(J), _dp(out))
    assert ok
    return out.reshape(3, 3), int(lib().oracle_cp_rank3(_dp(J)))


def pinv_rule_disagreements(reset=False):
    """Swing legs so far on which arma::pinv's own tolerance - what oracle_swing_torque applies, like the reference - and the
    device's rank rule (complete pivoting, oracle_cp_rank3) would keep a different number of singular values: the
    parity tests assert that this stays 0 instead of letting the checker take its rank from the device."""
    lib().oracle_pinv_rule_disagreements.restype = C.c_long
    return int(lib().oracle_pinv_rule_disagreements(C.c_int(1 if reset else 0)))


def swing_torque(leg, Rwb, x, pos, vel, q, qdot, kin=None):
    """oracle_swing_torque of one leg (unclamped)."""
    kin = kin or default_kinematics()
    a = [np.ascontiguousarray(v, np.float64).reshape(-1) for v in (Rwb, x, pos, vel, q, qdot)]
    tau = np.zeros(3)
    lib().oracle_swing_torque(C.byref(kin), C.c_int(leg), *[_dp(v) for v in a], _dp(tau))
    return tau


def tick_swing_batch(P, batch, kin=None, threads=1, max_iter=200):
    """tick_batch + swing-leg torques; batch also holds joint_qdot, swing_pos, swing_vel [n,12]."""
    kin = kin or default_kinematics()
    p = make_params(P, max_iter)
    n = batch["x"].shape[0]
    names = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "joint_q", "joint_qdot", "swing_pos", "swing_vel")
    a = [np.ascontiguousarray(batch[k], np.float64) for k in names]
    st = np.ascontiguousarray(batch["stance"], np.uint8)
    grf = np.zeros((n, 12)); tau = np.zeros((n, 12)); status = np.zeros(n, np.int32)
    lib().oracle_tick_swing_batch(C.byref(p), C.byref(kin), C.c_long(n), *[_dp(v) for v in a],
                                  st.ctypes.data_as(C.POINTER(C.c_ubyte)), _dp(grf), _dp(tau),
                                  status.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(threads))
    return dict(grf_body=grf, joint_tau=tau, status=status)


SWING_STATE_DTYPE = np.dtype([("leg_state", np.int32, 4), ("has_traj", np.int32, 4), ("p_start", np.float64, 12), ("p_final", np.float64, 12)])


def new_swing_states(n):
    s = np.zeros(n, dtype=SWING_STATE_DTYPE)
    s["leg_state"] = -1
    return s


def tick_planned_batch(P, batch, states, kin=None, threads=1, max_iter=200):
    """Full tick with on-device-style swing planning; `states` (SWING_STATE_DTYPE array) is updated in place."""
    kin = kin or default_kinematics()
    p = make_params(P, max_iter)
    n = batch["x"].shape[0]
    names = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "joint_q", "joint_qdot", "gait_phase")
    a = [np.ascontiguousarray(batch[k], np.float64) for k in names]
    grf = np.zeros((n, 12)); tau = np.zeros((n, 12)); status = np.zeros(n, np.int32)
    assert states.dtype == SWING_STATE_DTYPE and states.flags["C_CONTIGUOUS"] and C.sizeof(OracleSwingState) == SWING_STATE_DTYPE.itemsize
    lib().oracle_tick_planned_batch(C.byref(p), C.byref(kin), C.c_long(n), states.ctypes.data_as(C.c_void_p), *[_dp(v) for v in a],
                                    _dp(grf), _dp(tau), status.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(threads))
    return dict(grf_body=grf, joint_tau=tau, status=status)


def gait_update(phases, dt, kin=None):
    """GaitScheduler::update(dt) on [n, 4] phases, in place (gait.cpp:113-123)."""
    kin = kin or default_kinematics()
    assert phases.dtype == np.float64 and phases.flags["C_CONTIGUOUS"]
    dt = np.ascontiguousarray(dt, np.float64)
    lib().oracle_gait_update(C.byref(kin), C.c_long(phases.shape[0]), _dp(phases), _dp(dt))
    return phases
