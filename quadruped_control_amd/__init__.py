"""quadruped_control_amd - MI355X-native batched balance controller.

Scope: the BalanceController::control() hot path of bostoncleek/quadruped_control
(see DESIGN.md).  The numeric path is hand-written HIP for gfx950 behind the
C ABI in include/qc_balance.h; this package is the thin host-side mirror of
the reference interface.
"""
from .gait import LEG_NAMES, LegState, leg_state_from_phase, make_stance_gait, stance_phase  # noqa: F401


def cheetah_params(mu=0.8):
    """Constructor arguments used by the reference's commander node
    (commander_node.cpp:289-338 + mit_cheetah_config.yaml:66-99)."""
    import numpy as np

    return dict(mu=float(mu), mass=11.0, fzmin=10.0, fzmax=120.0,
                Ib=np.diag([0.011253, 0.036203, 0.042673]),
                S=np.diag([1.0, 1.0, 1.0, 10.0, 10.0, 5.0]), W=np.eye(12) * 1e-5,
                kff=np.array([0.0, 0.0, 0.15, 0.0, 0.0, 0.0]),
                kp_p=np.full(3, 100.0), kd_p=np.full(3, 50.0),
                kp_w=np.full(3, 5000.0), kd_w=np.full(3, 500.0))


def __getattr__(name):
    if name in ("BalanceController", "to_device", "new_swing_states", "SWING_STATE_DTYPE"):
        from . import balance_controller as _bc

        return getattr(_bc, name)
    raise AttributeError(name)
