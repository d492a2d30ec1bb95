"""Host-side mirror of the reference's BalanceController for this path.

Reference interface being mirrored (names, argument order and meaning, error
behaviour):
  quadruped_controller/include/quadruped_controller/balance_controller.hpp:85-107
  quadruped_controller/src/quadruped_controller/balance_controller.cpp:70-235
  * constructor(mu, mass, fzmin, fzmax, Ib, S, W, kff, kp_p, kd_p, kp_w, kd_w, leg_names)
  * control(Rwb, Rwb_d, x, xdot, w, x_d, xdot_d, w_d, foot_map, gait_map=make_stance_gait())
      -> ForceMap {leg: vec3} holding STANCE legs only; EMPTY map on solver
      failure (balance_controller.cpp:182-216); KeyError (std::out_of_range in
      the reference) if a leg is missing from foot_map / gait_map.
plus the batched entry point the GPU exists for, control_batch().

Everything numeric runs in the HIP library behind include/qc_balance.h;
torch is used only to own device memory and streams.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .gait import LEG_NAMES, LegState, make_stance_gait

_IN_FIELDS = (("Rwb", 9), ("Rwb_d", 9), ("x", 3), ("xdot", 3), ("w", 3), ("x_d", 3), ("xdot_d", 3),
              ("w_d", 3), ("feet", 12))


def _fill(carr, values, n, name):
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
    if a.size != n:
        raise ValueError(f"{name}: expected {n} values, got {a.size}")
    carr[:] = a.tolist()


class BalanceController:
    """Reactive optimal force-balance controller (drop-in for the reference class)."""

    def __init__(self, mu, mass, fzmin, fzmax, Ib, S, W, kff, kp_p, kd_p, kp_w, kd_w,
                 leg_names=LEG_NAMES, *, device=0, max_iter=200):
        lib = _lib.load()
        p = _lib.QcParams()
        p.mu, p.mass, p.fzmin, p.fzmax = float(mu), float(mass), float(fzmin), float(fzmax)
        _fill(p.Ib, Ib, 9, "Ib"); _fill(p.S, S, 36, "S"); _fill(p.W, W, 144, "W")
        _fill(p.kff, kff, 6, "kff"); _fill(p.kp_p, kp_p, 3, "kp_p"); _fill(p.kd_p, kd_p, 3, "kd_p")
        _fill(p.kp_w, kp_w, 3, "kp_w"); _fill(p.kd_w, kd_w, 3, "kd_w")
        p.max_iter = int(max_iter)
        self.leg_names = tuple(leg_names)
        if len(self.leg_names) != 4:
            raise ValueError("leg_names must hold 4 names (order RL, FL, RR, FR in the reference)")
        self.device = int(device)
        self._one = None  # control(): single-robot record buffers
        self._h = C.c_void_p()
        self._lib = lib
        rc = _lib.create(lib, p, self.device, self._h)
        if rc != _lib.QC_OK:
            raise RuntimeError(f"qc_create failed ({rc}): {_lib.last_error()}")

    @classmethod
    def from_params(cls, P, **kw):
        return cls(P["mu"], P["mass"], P["fzmin"], P["fzmax"], P["Ib"], P["S"], P["W"], P["kff"],
                   P["kp_p"], P["kd_p"], P["kp_w"], P["kd_w"], **kw)

    def set_kinematics(self, hip=None, links=None, tau_min=None, tau_max=None, jc_kff=None, jc_kp=None, jc_kd=None,
                       planner_hip=None, planner_k=None, swing_height=None):
        """Kinematic model of the joint_q / joint_tau extension; unspecified parts keep
        the reference's constants (kinematics.cpp:20-47, commander_node.cpp:324-325)."""
        k = _lib.QcKinematics()
        self._lib.qc_default_kinematics(C.byref(k))
        if hip is not None:
            _fill(k.hip, hip, 12, "hip")
        if links is not None:
            _fill(k.links, links, 12, "links")
        if tau_min is not None:
            k.tau_min = float(tau_min)
        if tau_max is not None:
            k.tau_max = float(tau_max)
        for name, val in (("jc_kff", jc_kff), ("jc_kp", jc_kp), ("jc_kd", jc_kd)):  # swing-leg joint PD gains
            if val is not None:
                _fill(getattr(k, name), val, 3, name)
        if planner_hip is not None:
            _fill(k.planner_hip, planner_hip, 12, "planner_hip")
        if planner_k is not None:
            k.planner_k = float(planner_k)
        if swing_height is not None:
            k.swing_height = float(swing_height)
        rc = self._lib.qc_set_kinematics(self._h, C.byref(k))
        if rc != _lib.QC_OK:
            raise RuntimeError(f"qc_set_kinematics failed ({rc}): {_lib.last_error()}")

    def set_gait(self, t_swing, t_stance):
        """Default stance_phase = t_stance / (t_swing + t_stance) (gait.cpp:36-46) of the on-device contact rule."""
        rc = self._lib.qc_set_gait(self._h, float(t_swing), float(t_stance))
        if rc != _lib.QC_OK:
            raise RuntimeError(f"qc_set_gait failed ({rc}): {_lib.last_error()}")

    def set_tuning(self, **kw):
        """Development / test interface (qc_set_tuning): explicit overrides of the launch heuristics, e.g.
        set_tuning(group=2, one_fill=1, force_general=1, clamp_steps=1).  A request the build cannot honour - persistent waves
        (one_fill=0, chunk beyond one fill) for a 6x6 form without -DQC_PERSISTENT_6X6=1 - makes the next launch or
        query_launch() fail instead of silently running one-fill workgroups.  The library reads no environment variables."""
        for key, value in kw.items():
            rc = self._lib.qc_set_tuning(self._h, key.encode(), float(value))
            if rc != _lib.QC_OK:
                raise ValueError(f"qc_set_tuning({key}) failed ({rc}): {_lib.last_error()}")
        return self

    def query_launch(self, n, kin=False, warm=False):
        """Which kernel instantiation a batch of n robots runs on: dict(lanes_per_robot, mode, form, strategies,
        chunk, blocks, resident_workgroups, lds_bytes) - qc_query_launch."""
        info = _lib.QcLaunchInfo()
        rc = self._lib.qc_query_launch(self._h, int(n), int(bool(kin)), int(bool(warm)), C.byref(info))
        if rc != _lib.QC_OK:
            raise RuntimeError(f"qc_query_launch failed ({rc}): {_lib.last_error()}")
        return {k: int(getattr(info, k)) for k, _ in info._fields_}

    @property
    def kernel_name(self):
        return self._lib.qc_kernel_name(self._h).decode()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.qc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------ single robot
    def control(self, Rwb, Rwb_d, x, xdot, w, x_d, xdot_d, w_d, foot_map, gait_map=None):
        """Same contract as the reference's control(); returns {leg_name: np.ndarray(3)}."""
        one = self._one
        if one is None:  # record buffers and their pointers, marshalled once
            buf = np.zeros(48 + 12)
            stance = np.zeros(4, dtype=np.uint8)
            status = np.zeros(1, dtype=np.int32)
            offs = (0, 9, 18, 21, 24, 27, 30, 33, 36, 48)
            ptrs = [C.c_void_p(buf.ctypes.data + 8 * o) for o in offs[:9]]
            ptrs += [C.c_void_p(stance.ctypes.data), C.c_void_p(buf.ctypes.data + 8 * 48), C.c_void_p(status.ctypes.data)]
            one = self._one = (buf, stance, status, ptrs)
        buf, stance, status, ptrs = one
        try:
            buf[0:9] = np.asarray(Rwb, dtype=np.float64).reshape(9)
            buf[9:18] = np.asarray(Rwb_d, dtype=np.float64).reshape(9)
            for o, v in ((18, x), (21, xdot), (24, w), (27, x_d), (30, xdot_d), (33, w_d)):
                buf[o:o + 3] = np.asarray(v, dtype=np.float64).reshape(3)
        except ValueError:
            raise ValueError("control(): argument has the wrong size") from None
        names = self.leg_names
        for i in range(4):
            name = names[i]
            buf[36 + 3 * i:39 + 3 * i] = np.asarray(foot_map[name], dtype=np.float64).reshape(3)  # KeyError == out_of_range
            stance[i] = 1 if gait_map is None or int(gait_map[name][0]) == 1 else 0  # LegState.stance == 1
        rc = self._lib.qc_control(self._h, *ptrs)
        if rc != _lib.QC_OK:
            raise RuntimeError(f"qc_control failed ({rc}): {_lib.last_error()}")
        force_map = {}
        if status[0] != 0:
            return force_map  # reference: ROS_ERROR + empty ForceMap
        grf = buf[48:60]
        for i in range(4):
            if stance[i]:
                force_map[names[i]] = grf[3 * i:3 * i + 3].copy()
        return force_map

    # ------------------------------------------------------------------ batches
    def _marshal(self, batch, warm, out, want_active_set, want_iterations, want_torques):
        """Validate the arguments of control_batch() and build the C structs; allocates `out` when it is None.
        Launches nothing.  Returns (n, bi, bo, warm_ptr, out)."""
        import torch

        n = batch["x"].shape[0]
        dev = torch.device("cuda", self.device)
        bi = _lib.QcBatchIn()
        fields = _IN_FIELDS + ((("joint_q", 12),) if batch.get("joint_q") is not None else ())
        for name, k in fields:
            t = batch.get(name)
            if t is None and name == "feet" and batch.get("joint_q") is not None:
                continue  # feet come from forward kinematics on the device
            if t is None or t.dtype != torch.float64 or not t.is_contiguous() or t.device != dev or t.numel() != n * k:
                raise ValueError(f"{name}: need contiguous float64 [{n},{k}] on {dev}")
            setattr(bi, name, t.data_ptr())
        st = batch.get("stance")
        if st is not None:
            if st.dtype != torch.uint8 or not st.is_contiguous() or st.numel() != n * 4 or st.device != dev:
                raise ValueError("stance: need contiguous uint8 [n,4]")
            bi.stance = st.data_ptr()
        for name, k in (("gait_phase", 4), ("gait_duty", 1),  # on-device contact rule (gait.cpp:125-134)
                        ("gait_dt", 1),  # on-device gait clock (gait.cpp:113-123): gait_phase is advanced in place
                        ("swing_pos", 12), ("swing_vel", 12), ("joint_qdot", 12)):  # swing-leg torques
            t = batch.get(name)
            if t is not None:
                if t.dtype != torch.float64 or not t.is_contiguous() or t.numel() != n * k or t.device != dev:
                    raise ValueError(f"{name}: need contiguous float64 [{n},{k}] on {dev}")
                setattr(bi, name, t.data_ptr())
        ss = batch.get("swing_state")
        if ss is not None:  # torch.uint8 tensor of n * sizeof(qc_swing_state) bytes, updated in place
            if ss.dtype != torch.uint8 or not ss.is_contiguous() or ss.numel() != n * SWING_STATE_DTYPE.itemsize or ss.device != dev:
                raise ValueError("swing_state: need a contiguous uint8 tensor of n * 224 bytes on the device")
            bi.swing_state = ss.data_ptr()
        if out is None:
            # Allocated here: defined contents until the first launch has run (plan_batch launches nothing) - zero forces and
            # status -1 ("not computed yet", no QC_* code), never uninitialised memory.
            out = {"grf_body": torch.zeros((n, 12), dtype=torch.float64, device=dev),
                   "status": torch.full((n,), -1, dtype=torch.int32, device=dev)}
            if want_active_set:
                out["active_set"] = torch.zeros((n,), dtype=torch.int32, device=dev)
            if want_iterations:
                out["iterations"] = torch.zeros((n,), dtype=torch.int32, device=dev)
            if want_torques:
                out["joint_tau"] = torch.zeros((n, 12), dtype=torch.float64, device=dev)
        else:
            for flag, name in ((want_active_set, "active_set"), (want_iterations, "iterations"), (want_torques, "joint_tau")):
                if flag and out.get(name) is None:
                    raise ValueError(f"out: '{name}' was asked for (want_{'torques' if name == 'joint_tau' else name}=True) but the supplied `out` has no such tensor")
            for name, shape, dt in (("grf_body", n * 12, torch.float64), ("status", n, torch.int32), ("active_set", n, torch.int32),
                                    ("iterations", n, torch.int32), ("joint_tau", n * 12, torch.float64)):
                t = out.get(name)
                if t is None:
                    if name in ("grf_body", "status"):
                        raise ValueError(f"out: '{name}' is required")
                    continue
                if t.dtype != dt or not t.is_contiguous() or t.numel() != shape or t.device != dev:
                    raise ValueError(f"out['{name}']: need contiguous {dt} with {shape} elements on {dev}")
        bo = _lib.QcBatchOut()
        bo.grf_body = out["grf_body"].data_ptr()
        bo.status = out["status"].data_ptr()
        bo.active_set = out["active_set"].data_ptr() if "active_set" in out else None
        bo.iterations = out["iterations"].data_ptr() if "iterations" in out else None
        bo.joint_tau = out["joint_tau"].data_ptr() if "joint_tau" in out else None
        warm_ptr = None
        if warm is not None:
            if warm.dtype != torch.int32 or warm.numel() != n or not warm.is_contiguous() or warm.device != dev:
                raise ValueError("warm: need contiguous int32 [n] (an active_set output) on the controller's device")
            warm_ptr = warm.data_ptr()
        return n, bi, bo, warm_ptr, out

    def control_batch(self, batch, warm=None, out=None, want_active_set=False, want_iterations=False, stream=None,
                      want_torques=False):
        """n robots, device-resident.  `batch`: dict of CUDA/HIP torch tensors
        (float64, contiguous; 'stance' uint8 [n,4] or None) on this controller's
        device.  Asynchronous on `stream` (default: torch's current stream).
        Returns dict(grf_body [n,12], status [n] int32, active_set?, iterations?).
        With batch['joint_q'] [n,12] the foot positions come from the reference's
        forward kinematics (kinematics.cpp:81-103) and want_torques=True adds
        joint_tau [n,12] = clamp(J^T f_body) for stance legs (kinematics.cpp:219-231)."""
        import torch

        n, bi, bo, warm_ptr, out = self._marshal(batch, warm, out, want_active_set, want_iterations, want_torques)
        s = stream if stream is not None else torch.cuda.current_stream(torch.device("cuda", self.device))
        rc = self._lib.qc_control_batch(self._h, n, C.byref(bi), warm_ptr, C.byref(bo), C.c_void_p(s.cuda_stream))
        if rc != _lib.QC_OK:
            raise RuntimeError(f"qc_control_batch failed ({rc}): {_lib.last_error()}")
        return out

    def plan_batch(self, batch, warm=None, out=None, want_active_set=False, want_iterations=False, stream=None,
                   want_torques=False):
        """Validate and marshal the arguments of control_batch() once and return
        (launch, out): `launch()` is a single C call (qc_control_batch) that can be
        issued every tick without Python-side marshalling, e.g. in a simulation or
        benchmark loop where the tensors are updated in place.  Planning launches
        NOTHING: the first launch() is the first tick (the gait clock and the swing
        planner of a stateful batch are not advanced by plan_batch itself)."""
        import torch

        n, bi, bo, warm_ptr, out = self._marshal(batch, warm, out, want_active_set, want_iterations, want_torques)
        self.query_launch(n, kin=batch.get("joint_q") is not None, warm=warm is not None)  # occupancy query done now, not inside a graph capture
        s = stream if stream is not None else torch.cuda.current_stream(torch.device("cuda", self.device))
        fn, h, sp = self._lib.qc_control_batch, self._h, C.c_void_p(s.cuda_stream)
        bi_ref, bo_ref = C.byref(bi), C.byref(bo)
        keep = (batch, warm, out, bi, bo)

        def launch(_keep=keep):
            rc = fn(h, n, bi_ref, warm_ptr, bo_ref, sp)
            if rc != _lib.QC_OK:
                raise RuntimeError(f"qc_control_batch failed ({rc}): {_lib.last_error()}")

        return launch, out

    def control_batch_host(self, batch, warm=None, want_active_set=False, want_iterations=False, want_torques=False):
        """n robots, numpy (host) arrays in and out; PCIe-inclusive convenience path."""
        n = batch["x"].shape[0]
        keep = []
        bi = _lib.QcBatchIn()
        fields = _IN_FIELDS + ((("joint_q", 12),) if batch.get("joint_q") is not None else ())
        for name, k in fields:
            if batch.get(name) is None and name == "feet" and batch.get("joint_q") is not None:
                continue
            a = np.ascontiguousarray(batch[name], dtype=np.float64)
            if a.size != n * k:
                raise ValueError(f"{name}: expected [{n},{k}]")
            keep.append(a)
            setattr(bi, name, a.ctypes.data)
        st = batch.get("stance")
        if st is not None:
            st = np.ascontiguousarray(st, dtype=np.uint8)
            keep.append(st)
            bi.stance = st.ctypes.data
        ss = batch.get("swing_state")
        if ss is not None:  # in/out: updated in place
            if ss.dtype != SWING_STATE_DTYPE or not ss.flags["C_CONTIGUOUS"] or ss.shape != (n,):
                raise ValueError("swing_state: need a C-contiguous new_swing_states(n) array")
            bi.swing_state = ss.ctypes.data
        for name in ("gait_phase", "gait_duty", "gait_dt", "swing_pos", "swing_vel", "joint_qdot"):
            if batch.get(name) is not None:
                a = np.ascontiguousarray(batch[name], dtype=np.float64)
                if name == "gait_phase" and batch.get("gait_dt") is not None and a is not batch[name]:
                    raise ValueError("gait_phase: with gait_dt the array is advanced in place, pass a C-contiguous float64 array")
                keep.append(a)
                setattr(bi, name, a.ctypes.data)
        out = {"grf_body": np.zeros((n, 12)), "status": np.zeros(n, dtype=np.int32)}
        if want_active_set:
            out["active_set"] = np.zeros(n, dtype=np.uint32)
        if want_iterations:
            out["iterations"] = np.zeros(n, dtype=np.int32)
        if want_torques:
            out["joint_tau"] = np.zeros((n, 12))
        bo = _lib.QcBatchOut()
        bo.joint_tau = out["joint_tau"].ctypes.data if want_torques else None
        bo.grf_body = out["grf_body"].ctypes.data
        bo.status = out["status"].ctypes.data
        bo.active_set = out["active_set"].ctypes.data if want_active_set else None
        bo.iterations = out["iterations"].ctypes.data if want_iterations else None
        wp = None
        if warm is not None:
            warm = np.ascontiguousarray(warm, dtype=np.uint32)
            wp = warm.ctypes.data
        rc = self._lib.qc_control_batch_host(self._h, n, C.byref(bi), wp, C.byref(bo))
        if rc != _lib.QC_OK:
            raise RuntimeError(f"qc_control_batch_host failed ({rc}): {_lib.last_error()}")
        return out


SWING_STATE_DTYPE = np.dtype([("leg_state", np.int32, 4), ("has_traj", np.int32, 4),
                              ("p_start", np.float64, 12), ("p_final", np.float64, 12)])  # == qc_swing_state


def new_swing_states(n):
    """Host array of n `qc_swing_state` records in the "nothing planned yet" state (qc_swing_state_init)."""
    s = np.zeros(n, dtype=SWING_STATE_DTYPE)
    _lib.load().qc_swing_state_init(s.ctypes.data_as(C.c_void_p), n)
    return s


def to_device(batch, device=0):
    """numpy batch (workloads.py) -> dict of torch tensors on cuda:<device>."""
    import torch

    dev = torch.device("cuda", device)
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in batch.items()}
