"""-m gpu: the fuzz campaigns of tests/stress_fuzz*.py under pytest with a fixed time budget, so that
the driver runs them: random constructor arguments (mu 0.05-2, w 1e-7-1e-2, fzmin = fzmax, full S, dense W, per-axis W)
with cold and warm-started ticks, and wild robot states (rotations up to pi, arbitrary contact patterns, large
velocities) - GPU against the C oracle.  Fixed seeds: the trials are a deterministic sequence, the budget only decides
how far down the sequence a run gets (the long versions are `python tests/stress_fuzz.py [trials] [robots]`).
Bars: no status mismatch; forces within north_star's 1e-4 relative (the campaigns' known worst is 2.9e-6, reached by
parameter sets with w <= 2e-7 where the 6x6 form's conditioning shows - DESIGN.md section 5)."""
import pytest

pytestmark = pytest.mark.gpu


def test_parameter_fuzz_30s(built):
    from tests import stress_fuzz

    worst, mismatches, done = stress_fuzz.run(trials=900, n=2048, budget_s=30.0)
    assert done >= 6 and mismatches == 0 and worst < 1e-4, (worst, mismatches, done)


def test_state_fuzz_30s(built):
    from tests import stress_fuzz_states

    worst, mismatches, done = stress_fuzz_states.run(batches=200, n=4096, budget_s=30.0)
    assert done >= 4 and mismatches == 0 and worst < 1e-6, (worst, mismatches, done)


def test_tick_fuzz_20s(built):
    """The widened tick (FK -> QP -> J^T, swing legs IK + J^-1 / pinv + PD) with random kinematic models, wild joint angles
    (stretched and folded legs) and swing references far from the feet."""
    from tests import stress_fuzz_tick

    worst_f, worst_tau, flips, mismatches, done = stress_fuzz_tick.run(batches=120, n=4096, budget_s=20.0)
    assert done >= 3 and mismatches == 0 and worst_f < 1e-6 and flips == 0 and worst_tau < 1e-6, (worst_f, worst_tau, flips, mismatches, done)


def test_planner_fuzz_20s(built):
    """The stateful complete tick over 60 ticks with random gait timing, planner gains, swing height and velocity commands:
    the carried swing state and the torques track the oracle tick by tick."""
    from tests import stress_fuzz_planner

    worst_tau, worst_p, state_mismatch_ticks, done = stress_fuzz_planner.run_campaign(runs=40, n=2048, ticks=60, budget_s=20.0)
    assert done >= 2 and state_mismatch_ticks == 0 and worst_tau < 1e-6 and worst_p < 1e-9, (worst_tau, worst_p, state_mismatch_ticks, done)
