"""ctypes view of libqc_balance.so (C ABI: include/qc_balance.h).

There is NO fallback: if the HIP library is missing or cannot be loaded the
import of the controller fails loudly.  The .so is built in-tree by
__graft_entry__.build() (hipcc --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QC_LIB_PATH") or os.path.join(_HERE, "libqc_balance.so")  # QC_LIB_PATH: development builds (tools/)

QC_OK = 0
QC_ERR_ABI = -4
ABI_VERSION = 6  # the revision of include/qc_balance.h these ctypes structures were written against
STATUS_NAMES = {0: "solved", 1: "max_iter", 2: "infeasible", 3: "not_pd"}


class QcParams(C.Structure):
    _fields_ = [("mu", C.c_double), ("mass", C.c_double), ("fzmin", C.c_double), ("fzmax", C.c_double),
                ("Ib", C.c_double * 9), ("S", C.c_double * 36), ("W", C.c_double * 144),
                ("kff", C.c_double * 6), ("kp_p", C.c_double * 3), ("kd_p", C.c_double * 3),
                ("kp_w", C.c_double * 3), ("kd_w", C.c_double * 3),
                ("max_iter", C.c_int32), ("reserved", C.c_int32)]


class QcBatchIn(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in
                ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "feet", "stance", "joint_q", "gait_phase", "gait_duty",
                 "swing_pos", "swing_vel", "joint_qdot", "swing_state", "gait_dt")]


class QcBatchOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("grf_body", "status", "active_set", "iterations", "joint_tau")]


class QcKinematics(C.Structure):
    _fields_ = [("hip", C.c_double * 12), ("links", C.c_double * 12), ("tau_min", C.c_double), ("tau_max", C.c_double),
                ("jc_kff", C.c_double * 3), ("jc_kp", C.c_double * 3), ("jc_kd", C.c_double * 3),
                ("planner_hip", C.c_double * 12), ("planner_k", C.c_double), ("swing_height", C.c_double)]


class QcSwingState(C.Structure):
    _fields_ = [("leg_state", C.c_int32 * 4), ("has_traj", C.c_int32 * 4), ("p_start", C.c_double * 12), ("p_final", C.c_double * 12)]


class QcLaunchInfo(C.Structure):
    _fields_ = [("lanes_per_robot", C.c_int32), ("mode", C.c_int32), ("form", C.c_int32), ("strategies", C.c_int32),
                ("chunk", C.c_int64), ("blocks", C.c_int64), ("resident_workgroups", C.c_int64), ("lds_bytes", C.c_int64)]


EXPORTS = ("qc_create_abi", "qc_destroy", "qc_control_batch", "qc_control_batch_host", "qc_control",
           "qc_last_error", "qc_kernel_name", "qc_abi_version", "qc_default_kinematics", "qc_set_kinematics", "qc_set_gait", "qc_swing_state_init",
           "qc_set_tuning", "qc_query_launch", "qc_check_abi")

_lib = None


def load():
    """Load the HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 /
    # libhsa-runtime64 (same SONAME as /opt/rocm's).  If our library were loaded
    # first it would pull in /opt/rocm's copy and a later `import torch` would
    # bring up a second HSA runtime that sees no GPU.  Importing torch first
    # makes the dynamic loader resolve our NEEDED libamdhip64.so.7 to the copy
    # torch already mapped.  (A C++ host that never loads torch simply uses
    # /opt/rocm's runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - torch is optional plumbing
        pass
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise ImportError(f"{LIB_PATH} does not export {name}")
    # (ABI v6: the exported constructor takes the caller's ABI revision and struct sizes; `qc_create` is an inline wrapper in the header)
    lib.qc_create_abi.argtypes = [C.POINTER(QcParams), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
    lib.qc_create_abi.restype = C.c_int
    lib.qc_destroy.argtypes = [C.c_void_p]
    lib.qc_destroy.restype = None
    lib.qc_control_batch.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(QcBatchIn), C.c_void_p,
                                     C.POINTER(QcBatchOut), C.c_void_p]
    lib.qc_control_batch.restype = C.c_int
    lib.qc_control_batch_host.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(QcBatchIn), C.c_void_p,
                                          C.POINTER(QcBatchOut)]
    lib.qc_control_batch_host.restype = C.c_int
    lib.qc_control.argtypes = [C.c_void_p] + [C.c_void_p] * 10 + [C.c_void_p, C.c_void_p]
    lib.qc_control.restype = C.c_int
    lib.qc_last_error.restype = C.c_char_p
    lib.qc_kernel_name.argtypes = [C.c_void_p]
    lib.qc_kernel_name.restype = C.c_char_p
    lib.qc_abi_version.restype = C.c_int
    lib.qc_default_kinematics.argtypes = [C.POINTER(QcKinematics)]
    lib.qc_default_kinematics.restype = None
    lib.qc_set_kinematics.argtypes = [C.c_void_p, C.POINTER(QcKinematics)]
    lib.qc_set_kinematics.restype = C.c_int
    lib.qc_set_gait.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.qc_set_gait.restype = C.c_int
    lib.qc_swing_state_init.argtypes = [C.c_void_p, C.c_size_t]
    lib.qc_swing_state_init.restype = None
    lib.qc_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    lib.qc_set_tuning.restype = C.c_int
    lib.qc_query_launch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(QcLaunchInfo)]
    lib.qc_query_launch.restype = C.c_int
    lib.qc_check_abi.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
    lib.qc_check_abi.restype = C.c_int
    # the structures above are hand-written mirrors of the header: a library built from another revision is refused here,
    # before any of them crosses the boundary
    rc = lib.qc_check_abi(ABI_VERSION, C.sizeof(QcParams), C.sizeof(QcBatchIn), C.sizeof(QcBatchOut))
    if rc != QC_OK:
        raise ImportError(f"{LIB_PATH}: {lib.qc_last_error().decode()}")
    _lib = lib
    return lib


def create(lib, params, device, handle):
    """qc_create for this binding: qc_create_abi with the revision and the sizes of the ctypes mirrors above."""
    return lib.qc_create_abi(C.byref(params), int(device), C.byref(handle), ABI_VERSION, C.sizeof(QcParams), C.sizeof(QcBatchIn), C.sizeof(QcBatchOut))


def last_error():
    return load().qc_last_error().decode()
