"""-m gpu: the fuzz campaigns of tests/stress_fuzz*.py under pytest with a fixed time budget, so that
the driver runs them: random constructor arguments (mu 0.05-2, w 1e-7-1e-2, fzmin = fzmax, full S, dense W, per-axis W)
with cold and warm-started ticks, and wild robot states (rotations up to pi, arbitrary contact patterns, large
velocities) - GPU against the C oracle.  Fixed seeds: the trials are a deterministic sequence, the budget only decides
how far down the sequence a run gets (the long versions are `python tests/stress_fuzz.py [trials] [robots]`).
Bars: no status mismatch; forces within 1e-5 relative - north_star's bar is 1e-4; this deterministic sequence's worst is 2.9e-6, and
20 000-trial campaigns on other seeds end at 2.1e-6 ... 7.0e-6 (profiles/r06_fuzz_campaigns.log).  Two things used to sit above that and are
gone in round 6: the acceptance threshold's reach tol |g| / (2w) (seed 555, trial 138: 6.9e-5 -> 2.9e-8 with the polish at acceptance,
test_polish_closes_the_small_w_gap), and the 6x6 dual form's conditioning eps (S/w) |b| at weights 300x further below S than the reference's
(1.4e-5 ... 2.4e-5: identical working sets, GPU stationarity 1e-8) - handles with max diag(S) / min diag(W) > 3e8 run the dense 12x12 form now,
whose primal reduced Hessian is well-conditioned exactly there (tests/test_gpu_parity.py::test_small_w_golden)."""
import pytest

pytestmark = pytest.mark.gpu


def test_parameter_fuzz_30s(built):
    from tests import stress_fuzz

    worst, mismatches, done = stress_fuzz.run(trials=900, n=2048, budget_s=30.0)
    assert done >= 6 and mismatches == 0 and worst < 1e-5, (worst, mismatches, done)


def test_polish_closes_the_small_w_gap(built):
    """VERDICT r5 item 1.  Trial 138 of the campaign QC_FUZZ_SEED=555 (w = 1.07e-7): round 5 stopped 3.0e-3 N = 6.9e-5 relative from
    the minimiser on every formulation, because a multiplier inside the acceptance threshold -tol |g| is worth tol |g| / (2w) of
    force.  With the polish at acceptance (Lane::iterate) every form and width is within 1e-6 of the oracle; with it switched off
    (`polish` = 0: round 5's rule) the gap is still there - the test knows it is testing the thing."""
    import numpy as np

    import quadruped_control_amd as q
    from oracle import c_oracle as O
    from tests import stress_fuzz

    P, b = stress_fuzz.trial_at(555, 138)
    assert abs(P["W"][0, 0] - 1.07e-7) < 1e-9
    ref, st, _ = O.control_batch(P, b, threads=16)
    scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))

    def err(**tune):
        o = q.BalanceController.from_params(P).set_tuning(**tune).control_batch_host(b)
        assert (o["status"] == 0).all() and (st == 0).all()
        return float(np.max(np.abs(o["grf_body"] - ref) / scale))

    # (auto_dense = 0: at S / w = 5e8 a default handle runs the dense form since round 6 - these are about the 6x6 forms' walk)
    for tune in ({"auto_dense": 0}, {"auto_dense": 0, "force_general": 1}, {"force_dense": 1}, {"auto_dense": 0, "group": 4}, {"auto_dense": 0, "group": 2},
                 {"auto_dense": 0, "group": 1}, {"auto_dense": 0, "group": 4, "race": 1}, {"auto_dense": 0, "pair": 1}, {}):
        assert err(**tune) < 1e-6, tune
    assert err(auto_dense=0, polish=0) > 2e-5 and err(polish=0) > 2e-5  # (the threshold's reach is the same on every form)
    assert q.BalanceController.from_params(P).kernel_name == "dense-12x12" and q.BalanceController.from_params(P).set_tuning(auto_dense=0).kernel_name == "diagW-6x6-uniform"


def test_state_fuzz_30s(built):
    from tests import stress_fuzz_states

    worst, mismatches, done = stress_fuzz_states.run(batches=200, n=4096, budget_s=30.0)
    assert done >= 4 and mismatches == 0 and worst < 1e-6, (worst, mismatches, done)


def test_tick_fuzz_20s(built):
    """The widened tick (FK -> QP -> J^T, swing legs IK + J^-1 / pinv + PD) with random kinematic models, wild joint angles
    (stretched and folded legs) and swing references far from the feet."""
    from oracle import c_oracle
    from tests import stress_fuzz_tick

    c_oracle.pinv_rule_disagreements(reset=True)
    worst_f, worst_tau, flips, mismatches, done, nan_one_side = stress_fuzz_tick.run(batches=120, n=4096, budget_s=20.0)
    # the oracle applies arma::pinv's own rank rule, the device its complete-pivoting one; no swing leg of the campaign
    # (stretched and folded legs, random kinematic models, references far out of reach) sat where the two differ
    assert c_oracle.pinv_rule_disagreements(reset=True) == 0
    assert done >= 3 and mismatches == 0 and worst_f < 1e-6 and flips == 0 and worst_tau < 1e-6, (worst_f, worst_tau, flips, mismatches, done)
    # a torque that is NaN on one side only compares False against any tolerance - it is its own failure (round 4's
    # finding: the device's IK returned numbers where the reference leg is all-NaN).  26 000 batches since the fix: none.
    assert nan_one_side == 0, nan_one_side


def test_planner_fuzz_20s(built):
    """The stateful complete tick over 60 ticks with random gait timing, planner gains, swing height and velocity commands:
    the carried swing state and the torques track the oracle tick by tick."""
    from oracle import c_oracle
    from tests import stress_fuzz_planner

    c_oracle.pinv_rule_disagreements(reset=True)
    worst_tau, worst_p, state_mismatch_ticks, done = stress_fuzz_planner.run_campaign(runs=40, n=2048, ticks=60, budget_s=20.0)
    assert c_oracle.pinv_rule_disagreements(reset=True) == 0  # (see test_tick_fuzz_20s)
    assert done >= 2 and state_mismatch_ticks == 0 and worst_tau < 1e-6 and worst_p < 1e-9, (worst_tau, worst_p, state_mismatch_ticks, done)


def test_gait_clock_fuzz_20s(built):
    """random gait periods (qc_set_gait) and jittered gait_dt over hundreds of ticks, with a quarter of the
    steps aimed at the duty edge's 1e-12 slack (gait.cpp:125-134), the phase wrap and dt = 0 / whole periods: the clock the device
    carries stays bit-equal to the oracle's GaitScheduler::update, and contact states, swing state and torques follow."""
    from oracle import c_oracle
    from tests import stress_fuzz_gait

    c_oracle.pinv_rule_disagreements(reset=True)
    worst_tau, mismatch_ticks, edges, aimed_hits, done = stress_fuzz_gait.run_campaign(runs=30, n=1024, ticks=500, budget_s=20.0)
    assert c_oracle.pinv_rule_disagreements(reset=True) == 0  # (see test_tick_fuzz_20s)
    assert done >= 1 and mismatch_ticks == 0 and worst_tau < 1e-6, (worst_tau, mismatch_ticks, done)
    assert edges > 1000 and aimed_hits > 1000, (edges, aimed_hits)  # the campaign really exercised edges and the slack


def test_cross_form_parity_600k(built):
    """tests/stress_parity.py under pytest: every formulation and lane-group width on the same 600 000 robots - the default
    plan (paired waves at this size), one-fill workgroups (pair = 0), four lanes per robot, the general 6x6 and the dense
    12x12 form - agree with each other to 1e-7 of max|GRF| and with the oracle (65 536-robot sample) to 1e-6."""
    import numpy as np
    import torch

    import quadruped_control_amd as q
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    n = 600000
    P = q.cheetah_params(0.6)
    b = W.config3(n, seed=0x5EED00AA)
    d = q.to_device(b)
    res = {}
    for name, tune in (("default", {}), ("one-fill", {"pair": 0}), ("four lanes", {"group": 4}), ("general 6x6", {"force_general": 1}),
                       ("dense 12x12", {"force_dense": 1})):
        ctl = q.BalanceController.from_params(P).set_tuning(**tune)
        o = ctl.control_batch(d)
        torch.cuda.synchronize()
        assert int((o["status"] != 0).sum()) == 0, name
        res[name] = o["grf_body"].cpu().numpy()
    assert q.BalanceController.from_params(P).query_launch(n)["mode"] == 3
    base = res["default"]
    scale = np.maximum(1.0, np.abs(base).max(axis=1, keepdims=True))
    for name, g in res.items():
        assert np.max(np.abs(g - base) / scale) < 1e-7, name
    idx = np.random.default_rng(0).choice(n, 65536, replace=False)
    ref, st, _ = O.control_batch(P, {k: np.ascontiguousarray(v[idx]) for k, v in b.items()}, threads=16)
    assert (st == 0).all() and np.max(np.abs(base[idx] - ref) / scale[idx]) < 1e-6
