"""Boundary types of the balance-controller path.

Mirrors the few pieces of the reference's types.hpp / gait.cpp that
BalanceController::control() touches:
  * LegState            include/quadruped_controller/types.hpp:91-95
  * GaitMap             types.hpp:100  (leg name -> (LegState, phase))
  * make_stance_gait()  src/quadruped_controller/gait.cpp:24-34
  * phase -> LegState   gait.cpp:36-46 (stance_phase), gait.cpp:125-134 (rule)
The GaitScheduler thread itself is out of scope.
"""
from __future__ import annotations

import enum

import numpy as np

LEG_NAMES = ("RL", "FL", "RR", "FR")  # src/commander_node.cpp:61


class LegState(enum.IntEnum):
    swing = 0
    stance = 1


def make_stance_gait():
    """Default gait_map argument of control(): all legs stance, phase 0."""
    return {name: (LegState.stance, 0.0) for name in LEG_NAMES}


def stance_phase(t_swing, t_stance):
    """gait.cpp:45: stance occupies phase domain [0, stance_phase]."""
    return t_stance / (t_swing + t_stance)


def leg_state_from_phase(phase, stance_phase_):
    """gait.cpp:125-134, vectorised.  almost_equal() there is |a-b| < 1e-12
    (math/numerics.hpp:29).  Returns uint8 array of LegState values."""
    phase = np.asarray(phase, dtype=np.float64)
    eps = 1.0e-12
    ge0 = (phase > 0.0) | (np.abs(phase) < eps)
    le = (phase < stance_phase_) | (np.abs(phase - stance_phase_) < eps)
    return (ge0 & le).astype(np.uint8)
