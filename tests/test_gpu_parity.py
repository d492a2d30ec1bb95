"""-m gpu: HIP path (through the C ABI) vs the oracle and the golden fixtures."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "balance_golden.json")
FIELDS = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "feet")
RTOL = 1e-6  # north_star bar is 1e-4 relative; we hold the kernel to 1e-6 of max|GRF|


def _relerr(a, b):
    scale = np.maximum(1.0, np.abs(b).max(axis=-1, keepdims=True))
    return float(np.max(np.abs(a - b) / scale))


@pytest.fixture(scope="module")
def q(built):
    import torch

    assert torch.cuda.is_available()
    import quadruped_control_amd as q

    return q


def _gold_batches():
    with open(GOLD) as f:
        gold = json.load(f)
    groups = {}
    for c in gold["cases"]:
        groups.setdefault((c["mu"], c["fzmin"], c["fzmax"]), []).append(c)
    return groups


def test_golden_fixtures(q):
    for (mu, fzmin, fzmax), cases in _gold_batches().items():
        P = q.cheetah_params(mu)
        P["fzmin"], P["fzmax"] = fzmin, fzmax
        ctl = q.BalanceController.from_params(P)
        batch = {k: np.array([c[k] for c in cases], dtype=np.float64) for k in FIELDS}
        batch["stance"] = np.array([c["stance"] for c in cases], dtype=np.uint8)
        out = ctl.control_batch_host(batch, want_iterations=True)
        assert (out["status"] == 0).all()
        exp = np.array([c["grf_body"] for c in cases])
        assert _relerr(out["grf_body"], exp) < RTOL
        # swing legs exactly zero
        sw = np.repeat(batch["stance"] == 0, 3, axis=1)
        assert np.all(out["grf_body"][sw] == 0.0)


def _small_w_groups(q):
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "small_w_golden.json")))
    for grp in gold["groups"]:
        P = q.cheetah_params(0.6)
        for k, v in grp["params"].items():
            P[k] = np.array(v, dtype=np.float64) if isinstance(v, list) else v
        cases = grp["cases"]
        batch = {k: np.array([c[k] for c in cases], dtype=np.float64) for k in FIELDS}
        batch["stance"] = np.array([c["stance"] for c in cases], dtype=np.uint8)
        yield grp, P, batch, np.array([c["grf_body"] for c in cases])


def test_small_w_golden(q):
    """Weights a hundredth of the reference's (w ~ 1e-7), against forces from the numpy / NNLS restatement (tests/golden/make_small_w_golden.py:
    least-distance programming, KKT-certified - not the C oracle's active-set solver).  Campaign 555's trial 138 is the acceptance threshold's
    reach: 6.9e-5 on robot 168 without the polish at acceptance, <= 2e-6 with it on every form (the C oracle itself is 4.8e-7 from NNLS on that
    trial).  Campaign 20260929's trials are the 6x6 dual form's conditioning eps (S / w) |b|: 2.4e-5 at worst, the dense form well below it - which is
    why a handle with max diag(S) / min diag(W) > 3e8 runs the dense form by default (qc_create, QC_DENSE_RATIO)."""
    for grp, P, batch, exp in _small_w_groups(q):
        scale = np.maximum(1.0, np.abs(exp).max(axis=1, keepdims=True))

        def err(**tune):
            o = q.BalanceController.from_params(P).set_tuning(**tune).control_batch_host(batch)
            assert (o["status"] == 0).all()
            return np.max(np.abs(o["grf_body"] - exp) / scale, axis=1)

        threshold_trial = grp["campaign_seed"] == 555
        # the 6x6 dual forms (auto_dense = 0: a default handle does not run them at these S / w ratios any more)
        for tune in ({"auto_dense": 0}, {"auto_dense": 0, "force_general": 1}, {"auto_dense": 0, "group": 1}, {"auto_dense": 0, "group": 2}):
            e = err(**tune)
            assert e.max() < (2e-6 if threshold_trial else 5e-5), (grp["trial"], tune, e)
        if threshold_trial:
            assert err(auto_dense=0, polish=0)[0] > 2e-5  # robot 168 comes first in its group: round 5's rule still shows the gap
        # what a default handle runs here: the dense 12x12 form - the PRIMAL reduced Hessian does not have the dual form's conditioning
        assert q.BalanceController.from_params(P).kernel_name == "dense-12x12"
        assert err().max() < 5e-6 and err(force_dense=1).max() < 5e-6, grp["trial"]


@pytest.mark.parametrize("cfg,n", [(2, 4096), (3, 8192)])
def test_against_c_oracle(q, cfg, n):
    import torch

    from oracle import c_oracle
    from quadruped_control_amd import workloads

    P = q.cheetah_params(0.6)
    batch = workloads.config2(n) if cfg == 2 else workloads.config3(n)
    ctl = q.BalanceController.from_params(P)
    out = ctl.control_batch(q.to_device(batch), want_iterations=True, want_active_set=True)
    torch.cuda.synchronize()
    grf = out["grf_body"].cpu().numpy()
    assert (out["status"].cpu().numpy() == 0).all()
    ref, st, _ = c_oracle.control_batch(P, batch, threads=8)
    assert (st == 0).all()
    assert _relerr(grf, ref) < RTOL
    it = out["iterations"].cpu().numpy()
    assert it.max() <= 200 and it.min() >= 1


def test_single_robot_control_kat1(q):
    """config 1 / KAT1 through the reference-shaped control()."""
    P = q.cheetah_params(0.8)
    ctl = q.BalanceController.from_params(P)
    feet = {"RL": [-0.196, 0.127, -0.26], "FL": [0.196, 0.127, -0.26], "RR": [-0.196, -0.127, -0.26], "FR": [0.196, -0.127, -0.26]}
    I = np.eye(3)
    z = np.zeros(3)
    x = np.array([0, 0, 0.26])
    fm = ctl.control(I, I, x, z, z, x, z, z, feet)
    assert sorted(fm) == ["FL", "FR", "RL", "RR"]
    for v in fm.values():
        assert abs(v[2] + 17.5353311617) < 1e-8 and abs(v[0]) < 1e-7 and abs(v[1]) < 1e-7
    # trot: swing legs are omitted from the map (balance_controller.cpp:222)
    gait = {"RL": (q.LegState.stance, 0.0), "FL": (q.LegState.swing, 0.6), "RR": (q.LegState.swing, 0.6), "FR": (q.LegState.stance, 0.0)}
    fm = ctl.control(I, I, x, z, z, x, z, z, feet, gait)
    assert sorted(fm) == ["FR", "RL"]
    assert abs(fm["RL"][2] + 35.0705746471) < 1e-8
    with pytest.raises(KeyError):
        ctl.control(I, I, x, z, z, x, z, z, {"RL": [0, 0, 0]})


def test_complete_tick_golden(q):
    """tests/golden/tick_golden.json: the complete controller tick over a trot and a walk, 30 jittered ticks each, written by the
    numpy restatement of commander_node.cpp:383-531 (oracle/tick_restatement.py) and cross-checked against the C oracle when it was
    made.  The device - gait clock, contact rule, foothold planner, sextic trajectories, IK / pinv, joint PD, QP, J^T in one launch per
    tick, state carried in qc_swing_state - reproduces every tick: clock bit for bit, contact states and trajectory flags, forces and
    torques to 1e-6.  (Device pointers and host pointers, the two entry points a caller has.)"""
    import torch

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tick_golden.json")))
    for case in gold["cases"]:
        P = q.cheetah_params(case["mu"])
        base = {k: np.ascontiguousarray(np.array(v, dtype=np.float64)) for k, v in case["inputs"].items()}
        n = base["Rwb"].shape[0]
        for entry in ("host", "device"):
            ctl = q.BalanceController.from_params(P)
            ctl.set_gait(case["t_swing"], case["t_stance"])
            phase = np.ascontiguousarray(np.array(case["phase0"]))
            state = q.new_swing_states(n)
            if entry == "device":
                d_phase = torch.from_numpy(phase).cuda()
                d_state = torch.from_numpy(state.view("uint8").reshape(-1).copy()).cuda()
                d_base = q.to_device(base)
            for t in case["ticks"]:
                x, dt = np.ascontiguousarray(np.array(t["x"])), np.ascontiguousarray(np.array(t["dt"]))
                if entry == "host":
                    o = ctl.control_batch_host(dict(base, x=x, gait_phase=phase, gait_dt=dt, swing_state=state), want_torques=True)
                    ph, st, grf, tau, status = phase, state, o["grf_body"], o["joint_tau"], o["status"]
                else:
                    o = ctl.control_batch(dict(d_base, x=torch.from_numpy(x).cuda(), gait_phase=d_phase, gait_dt=torch.from_numpy(dt).cuda(), swing_state=d_state),
                                          want_torques=True)
                    torch.cuda.synchronize()
                    ph = d_phase.cpu().numpy()
                    st = d_state.cpu().numpy().view(state.dtype)
                    grf, tau, status = o["grf_body"].cpu().numpy(), o["joint_tau"].cpu().numpy(), o["status"].cpu().numpy()
                assert np.array_equal(ph, np.array(t["phase"]))  # GaitScheduler::update on the device, bit for bit
                assert (status == 0).all()
                assert np.array_equal(st["leg_state"], np.array(t["leg_state"])) and np.array_equal(st["has_traj"], np.array(t["has_traj"]))
                assert _relerr(grf, np.array(t["grf_body"])) < RTOL
                assert np.abs(tau - np.array(t["joint_tau"])).max() < 1e-6 * 20.0
