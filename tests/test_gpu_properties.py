"""-m gpu: edge cases and size-independent properties at BASELINE.json's full
batch sizes (the oracle only sees bounded samples there)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "feet")


@pytest.fixture(scope="module")
def q(built):
    import quadruped_control_amd as q

    return q


def _world_forces(batch, grf):
    R = batch["Rwb"].reshape(-1, 3, 3)
    fb = grf.reshape(-1, 4, 3)
    return -np.einsum("nij,nkj->nki", R, fb)  # f_w = -R f_b


def _check_feasible(P, batch, grf, tol=1e-7):
    fw = _world_forces(batch, grf)
    st = batch["stance"].astype(bool)
    fx, fy, fz = fw[..., 0], fw[..., 1], fw[..., 2]
    assert np.all(np.abs(fw[~st]) == 0.0)
    assert np.all(fz[st] >= P["fzmin"] - tol) and np.all(fz[st] <= P["fzmax"] + tol)
    assert np.all(np.abs(fx[st]) <= P["mu"] * fz[st] + tol) and np.all(np.abs(fy[st]) <= P["mu"] * fz[st] + tol)


def _kkt_subset(P, batch, grf, idx):
    from oracle import c_oracle as O
    from oracle import numpy_restatement as R

    fw = _world_forces(batch, grf).reshape(-1, 12)
    worst = 0.0
    for i in idx:
        qp = O.assemble(P, *[batch[k][i] for k in FIELDS], batch["stance"][i])
        cert = R.kkt_certificate(qp["H"], qp["g"], qp["C"], qp["lb"], qp["ub"], fw[i])
        assert cert["primal"] < 1e-7, cert
        worst = max(worst, cert["stationarity"])
    return worst


def test_full_size_config4_warm_start(q):
    """262 144 robots, two ticks: cold == warm-started result, warm start
    needs fewer working-set recalculations, outputs feasible + KKT on a sample."""
    import torch

    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    ctl = q.BalanceController.from_params(P)
    t0, t1 = W.config4(262144)
    o0 = ctl.control_batch(q.to_device(t0), want_active_set=True, want_iterations=True)
    d1 = q.to_device(t1)
    cold = ctl.control_batch(d1, want_iterations=True, want_active_set=True)
    warm = ctl.control_batch(d1, warm=o0["active_set"], want_iterations=True, want_active_set=True)
    torch.cuda.synchronize()
    assert int((cold["status"] != 0).sum()) == 0 and int((warm["status"] != 0).sum()) == 0
    gc, gw = cold["grf_body"].cpu().numpy(), warm["grf_body"].cpu().numpy()
    scale = np.maximum(1.0, np.abs(gc).max(axis=1, keepdims=True))
    assert np.max(np.abs(gc - gw) / scale) < 1e-7  # unique minimiser: start must not matter
    ic, iw = cold["iterations"].float().mean().item(), warm["iterations"].float().mean().item()
    assert iw < 0.6 * ic, (ic, iw)
    _check_feasible(P, t1, gw)
    from tests.kkt_batch import assert_kkt

    assert_kkt(P, t1, gw)  # vectorised certificate on all 262 144 robots
    rng = np.random.default_rng(0)
    assert _kkt_subset(P, t1, gw, rng.choice(262144, 64, replace=False)) < 1e-8  # NNLS certificate on the oracle's literal rows, a sample
    # idempotence: restarting from the optimal working set takes exactly one recalculation
    again = ctl.control_batch(d1, warm=warm["active_set"], want_iterations=True)
    torch.cuda.synchronize()
    assert int(again["iterations"].max()) == 1
    # (stragglers of the first pass finished on the 4-lane layout, whose sums run in a different order: 1e-9 is the
    # rounding level of such a pair of solves, not a tolerance of the method)
    assert np.max(np.abs(again["grf_body"].cpu().numpy() - gw) / scale) < 1e-8


def test_full_size_config3_feasible_and_kkt(q):
    import torch

    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    ctl = q.BalanceController.from_params(P)
    b = W.config3(65536)
    o = ctl.control_batch(q.to_device(b), want_iterations=True)
    torch.cuda.synchronize()
    assert int((o["status"] != 0).sum()) == 0
    grf = o["grf_body"].cpu().numpy()
    _check_feasible(P, b, grf)
    from tests.kkt_batch import assert_kkt

    assert_kkt(P, b, grf)  # vectorised certificate on all 65 536 robots
    rng = np.random.default_rng(1)
    assert _kkt_subset(P, b, grf, rng.choice(65536, 64, replace=False)) < 1e-8


def test_mirror_symmetry(q):
    """Reflecting the robot state in the x-z plane mirrors the forces."""
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    ctl = q.BalanceController.from_params(P)
    b = W.config3(2048)
    M = np.diag([1.0, -1.0, 1.0])
    m = {k: v.copy() for k, v in b.items()}
    for k in ("Rwb", "Rwb_d"):
        m[k] = (M @ b[k].reshape(-1, 3, 3) @ M).reshape(-1, 9)
    for k in ("x", "xdot", "x_d", "xdot_d"):
        m[k] = b[k] * np.array([1.0, -1.0, 1.0])
    for k in ("w", "w_d"):  # pseudo-vectors
        m[k] = b[k] * np.array([-1.0, 1.0, -1.0])
    # mirrored feet swap left/right: RL<->RR, FL<->FR
    perm = [2, 3, 0, 1]
    m["feet"] = (b["feet"].reshape(-1, 4, 3)[:, perm] * np.array([1.0, -1.0, 1.0])).reshape(-1, 12)
    m["stance"] = np.ascontiguousarray(b["stance"][:, perm])
    g0 = ctl.control_batch_host(b)["grf_body"].reshape(-1, 4, 3)
    g1 = ctl.control_batch_host(m)["grf_body"].reshape(-1, 4, 3)
    exp = g0[:, perm] * np.array([1.0, -1.0, 1.0])
    assert np.max(np.abs(g1 - exp)) / max(1.0, np.abs(exp).max()) < 1e-7


def test_edge_cases(q):
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    ctl = q.BalanceController.from_params(P)
    b = W.config2(200)  # ragged: not a multiple of the 64-robot wavefront
    # all legs swing -> zero forces, solved
    b0 = dict(b); b0["stance"] = np.zeros((200, 4), np.uint8)
    o = ctl.control_batch_host(b0)
    assert (o["status"] == 0).all() and np.all(o["grf_body"] == 0.0)
    # stance=None means make_stance_gait()
    b1 = {k: v for k, v in b.items() if k != "stance"}
    o1 = ctl.control_batch_host(b1)
    o2 = ctl.control_batch_host(b)
    assert np.array_equal(o1["grf_body"], o2["grf_body"])
    # n = 0 and n = 1
    e = ctl.control_batch_host({k: v[:0] for k, v in b.items()})
    assert e["grf_body"].shape == (0, 12)
    one = ctl.control_batch_host({k: v[:1] for k, v in b.items()})
    assert np.array_equal(one["grf_body"][0], o2["grf_body"][0])
    # iteration cap -> status 1 (RET_MAX_NWSR_REACHED analogue) and zero forces
    capped = q.BalanceController.from_params(P, max_iter=2).control_batch_host(b, want_iterations=True)
    assert (capped["status"] == 1).any() and np.all(capped["grf_body"][capped["status"] == 1] == 0.0)
    assert capped["iterations"].max() <= 2
    # non-finite input -> status 3, zero forces, neighbours unaffected
    bn = {k: v.copy() for k, v in b.items()}
    bn["x"][7, 0] = np.nan
    on = ctl.control_batch_host(bn)
    assert on["status"][7] != 0 and np.all(on["grf_body"][7] == 0.0)
    assert np.array_equal(np.delete(on["grf_body"], 7, 0), np.delete(o2["grf_body"], 7, 0))
    # fzmin = 0 re-admits the cone apex (degenerate vertex)
    P0 = dict(P); P0["fzmin"] = 0.0
    b3 = W.config3(4096)
    o3 = q.BalanceController.from_params(P0).control_batch_host(b3)
    ref, st, _ = O.control_batch(P0, b3, threads=8)
    assert (o3["status"] == 0).all() and (st == 0).all()
    scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
    assert np.max(np.abs(o3["grf_body"] - ref) / scale) < 1e-6
    # bad constructor arguments are rejected, with a message
    bad = dict(P); bad["fzmin"] = 200.0
    with pytest.raises(RuntimeError, match="fzmin"):
        q.BalanceController.from_params(bad)
    # the +-1e6 sides of the reference's cone rows (balance_controller.cpp:296-301) are not carried: parameter sets for
    # which they could bind (2 mu fzmax >= 1e6) are refused, and just below the limit the oracle - which keeps the
    # literal two-sided rows - still agrees
    for mu, fzmax in ((0.6, 1.0e6), (2.0, 2.5e5), (50.0, 1.0e4)):
        bad = dict(P); bad["mu"] = mu; bad["fzmax"] = fzmax
        with pytest.raises(RuntimeError, match="1e6"):
            q.BalanceController.from_params(bad)
    edge = dict(P); edge["mu"] = 2.0; edge["fzmax"] = 2.4e5
    b4 = W.config3(2048)
    o4 = q.BalanceController.from_params(edge).control_batch_host(b4)
    ref4, st4, _ = O.control_batch(edge, b4, threads=8)
    assert (o4["status"] == 0).all() and (st4 == 0).all()
    assert np.max(np.abs(o4["grf_body"] - ref4) / np.maximum(1.0, np.abs(ref4).max(axis=1, keepdims=True))) < 1e-6


def test_general_S_and_per_axis_W(q):
    """non-diagonal SPD S and a non-uniform diagonal W against the oracle."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    rng = np.random.default_rng(5)
    P = q.cheetah_params(0.6)
    A = rng.normal(size=(6, 6))
    P["S"] = np.diag([1, 1, 1, 10, 10, 5.0]) + 0.3 * (A @ A.T)
    P["W"] = np.diag(rng.uniform(0.5e-5, 5e-5, 12))
    b = W.config3(2048)
    ctl = q.BalanceController.from_params(P)
    assert ctl.kernel_name == "diagW-6x6"  # general-S / per-axis-W form of the 6x6 path
    o = ctl.control_batch_host(b)
    ref, st, _ = O.control_batch(P, b, threads=8)
    assert (o["status"] == 0).all() and (st == 0).all()
    scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
    assert np.max(np.abs(o["grf_body"] - ref) / scale) < 1e-6


def test_dense_W_path(q, monkeypatch):
    """general (non-diagonal) SPD W -> dense 12x12 formulation; also force the
    dense kernel on the reference's diagonal W and compare both ways."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    rng = np.random.default_rng(11)
    b = W.config3(4096)
    # (a) diagonal W through the dense kernel == oracle == diagW kernel
    P = q.cheetah_params(0.6)
    ref, st, _ = O.control_batch(P, b, threads=8)
    dense = q.BalanceController.from_params(P).set_tuning(force_dense=1)
    assert dense.kernel_name == "dense-12x12"
    od = dense.control_batch_host(b, want_iterations=True)
    diag = q.BalanceController.from_params(P)
    assert diag.kernel_name == "diagW-6x6-uniform"  # S diagonal, W = w*I: scalar-constant specialisation
    og = diag.control_batch_host(b)
    scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
    assert (od["status"] == 0).all()
    assert np.max(np.abs(od["grf_body"] - ref) / scale) < 1e-6
    assert np.max(np.abs(od["grf_body"] - og["grf_body"]) / scale) < 1e-6
    # (b) coupled W
    A = rng.normal(size=(12, 12))
    P2 = dict(P)
    P2["W"] = 1e-5 * (np.eye(12) + 0.2 * (A @ A.T) / 12.0)
    P2["W"] = 0.5 * (P2["W"] + P2["W"].T)
    ctl = q.BalanceController.from_params(P2)
    assert ctl.kernel_name == "dense-12x12"
    o = ctl.control_batch_host(b)
    ref2, st2, _ = O.control_batch(P2, b, threads=8)
    assert (o["status"] == 0).all() and (st2 == 0).all()
    scale2 = np.maximum(1.0, np.abs(ref2).max(axis=1, keepdims=True))
    assert np.max(np.abs(o["grf_body"] - ref2) / scale2) < 1e-6


@pytest.mark.parametrize("n", [1000, 40000, 150000])  # four lanes per robot with racing strategies; one lane per robot (one and three rounds of one-fill waves)
def test_fused_tick_fk_and_torques(q, n):
    """joint_q in, joint_tau out (SURVEY 8f rows 1+2): device FK -> control -> clamp(J^T f)
    against the oracle's composition, and against the unfused path fed with the oracle's feet."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    b = W.with_joint_angles(W.config3(n))
    ctl = q.BalanceController.from_params(P)
    o = ctl.control_batch_host(b, want_torques=True)
    ref = O.tick_batch(P, b, threads=8)
    assert (o["status"] == 0).all() and (ref["status"] == 0).all()
    scale = np.maximum(1.0, np.abs(ref["grf_body"]).max(axis=1, keepdims=True))
    assert np.max(np.abs(o["grf_body"] - ref["grf_body"]) / scale) < 1e-6
    assert np.max(np.abs(o["joint_tau"] - ref["joint_tau"])) < 1e-6 * 20.0
    assert np.all(np.abs(o["joint_tau"]) <= 20.0) and np.all(o["joint_tau"][np.repeat(b["stance"] == 0, 3, axis=1)] == 0.0)
    assert (np.abs(o["joint_tau"]) == 20.0).any()  # the +-20 N m clamp is exercised
    # unfused path with the oracle's foot positions gives the same forces
    b2 = {k: v for k, v in b.items() if k != "joint_q"}
    b2["feet"] = ref["feet"]
    o2 = ctl.control_batch_host(b2)
    assert np.max(np.abs(o2["grf_body"] - o["grf_body"]) / scale) < 1e-8
    # custom kinematic model / torque limits
    ctl.set_kinematics(tau_min=-5.0, tau_max=7.0)
    o3 = ctl.control_batch_host(b, want_torques=True)
    assert o3["joint_tau"].min() >= -5.0 and o3["joint_tau"].max() <= 7.0
    with pytest.raises(RuntimeError, match="joint_tau needs joint_q"):
        ctl.control_batch_host(b2, want_torques=True)


def test_on_device_contact_rule(q):
    """gait phases in, contact state derived on the device (gait.cpp:125-134) ==
    the host-side rule used to build config 3 (quadruped_control_amd.gait)."""
    from quadruped_control_amd import leg_state_from_phase
    from quadruped_control_amd import workloads as W

    rng = np.random.default_rng(21)
    n = 5000
    P = q.cheetah_params(0.6)
    b = W.config2(n)
    phases = rng.uniform(0.0, 1.0, (n, 4))
    duty = np.where(rng.uniform(size=n) < 0.5, 0.5, 0.8 / 0.98)
    phases[:50, 0] = 0.0                      # boundary cases of the 1e-12 slack
    phases[50:100, 1] = duty[50:100]
    phases[100:150, 2] = duty[100:150] + 5e-13
    phases[150:200, 3] = duty[150:200] + 1e-9
    phases[200:220, 0] = -5e-13                # inside the slack below 0: stance (gait.cpp:125-134 via almost_equal)
    phases[220:240, 1] = -0.3                  # out of range either way: swing
    phases[240:260, 2] = 1.2
    phases[260:280, 3] = np.nan                # every comparison false: swing
    stance = leg_state_from_phase(phases, duty[:, None])
    ctl = q.BalanceController.from_params(P)
    ref = ctl.control_batch_host(dict(b, stance=stance))
    bg = {k: v for k, v in b.items() if k != "stance"}
    out = ctl.control_batch_host(dict(bg, gait_phase=phases, gait_duty=duty))
    assert np.array_equal(out["grf_body"], ref["grf_body"]) and np.array_equal(out["status"], ref["status"])
    # handle-wide default duty (qc_set_gait)
    ctl.set_gait(0.3, 0.3)
    out2 = ctl.control_batch_host(dict(bg, gait_phase=phases))
    ref2 = ctl.control_batch_host(dict(b, stance=leg_state_from_phase(phases, 0.5)))
    assert np.array_equal(out2["grf_body"], ref2["grf_body"])


def test_random_controller_parameters(q):
    """Parity vs the oracle over randomly drawn constructor arguments (friction,
    force limits, mass/inertia, gains, weights), incl. fzmin = fzmax and fzmin = 0."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    rng = np.random.default_rng(2024)
    b = W.config3(768)
    worst = 0.0
    for trial in range(12):
        P = q.cheetah_params(float(rng.uniform(0.2, 1.2)))
        P["fzmin"] = float(rng.choice([0.0, 5.0, 10.0, 25.0]))
        P["fzmax"] = P["fzmin"] if trial == 3 else float(P["fzmin"] + rng.uniform(20.0, 200.0))
        P["mass"] = float(rng.uniform(5.0, 30.0))
        P["Ib"] = np.diag(rng.uniform(0.01, 0.2, 3))
        P["S"] = np.diag(rng.uniform(0.5, 20.0, 6))
        P["W"] = np.eye(12) * float(10.0 ** rng.uniform(-6, -3))
        P["kp_w"] = np.full(3, float(rng.uniform(100.0, 5000.0)))
        P["kd_w"] = np.full(3, float(rng.uniform(10.0, 500.0)))
        P["kff"] = rng.uniform(0.0, 0.3, 6)
        if trial % 4 == 1:   # per-axis W -> general 6x6 form
            P["W"] = np.diag(10.0 ** rng.uniform(-6, -3, 12))
        if trial % 4 == 2:   # full SPD S -> general 6x6 form
            A = rng.normal(size=(6, 6)); P["S"] = P["S"] + 0.2 * A @ A.T
        ctl = q.BalanceController.from_params(P)
        o = ctl.control_batch_host(b)
        ref, st, _ = O.control_batch(P, b, threads=8)
        assert (o["status"] == 0).all() and (st == 0).all(), (trial, ctl.kernel_name)
        scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
        err = float(np.max(np.abs(o["grf_body"] - ref) / scale))
        worst = max(worst, err)
        assert err < 1e-6, (trial, ctl.kernel_name, err)


def test_graph_capture_replay(q):
    """qc_control_batch is a pure kernel launch on the given stream: it can be
    captured into a HIP graph and replayed (launch-bound small batches)."""
    import torch

    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    ctl = q.BalanceController.from_params(P)
    d = q.to_device(W.config2(4096))
    launch, out = ctl.plan_batch(d)  # plans only: nothing has been launched yet
    torch.cuda.synchronize()
    assert int((out["status"] != -1).sum()) == 0 and float(out["grf_body"].abs().max()) == 0.0  # defined, "not computed yet"
    with pytest.raises(ValueError, match="iterations"):  # asked for, but the supplied `out` cannot hold it
        ctl.plan_batch(d, out={"grf_body": out["grf_body"], "status": out["status"]}, want_iterations=True)
    launch()
    torch.cuda.synchronize()
    ref = out["grf_body"].clone()
    assert int((out["status"] != 0).sum()) == 0
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        launch_s, out_s = ctl.plan_batch(d, stream=s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            launch_s()
    out_s["grf_body"].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_s["grf_body"], ref)


@pytest.mark.parametrize("n", [900, 40000])
def test_swing_leg_torques(q, n):
    """SURVEY 8f rank 4 (stateless part): swing legs get IK + J^-1 + joint-PD torques
    (commander_node.cpp:482-504), stance legs keep J^T f; merged and clamped like :514-526."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    b = W.with_swing_references(W.with_joint_angles(W.config3(n)))
    ctl = q.BalanceController.from_params(P)
    o = ctl.control_batch_host(b, want_torques=True)
    ref = O.tick_swing_batch(P, b, threads=8)
    assert (o["status"] == 0).all()
    assert np.max(np.abs(o["joint_tau"] - ref["joint_tau"])) < 1e-6 * 20.0
    sw = np.repeat(b["stance"] == 0, 3, axis=1)
    assert np.any(o["joint_tau"][sw] != 0.0)
    # custom joint PD gains
    ctl.set_kinematics(jc_kp=[10.0, 20.0, 30.0], jc_kd=[0.5, 0.5, 0.5], jc_kff=[0.1, 0.2, 0.3])
    kin = O.default_kinematics()
    kin.jc_kp[:] = [10.0, 20.0, 30.0]; kin.jc_kd[:] = [0.5, 0.5, 0.5]; kin.jc_kff[:] = [0.1, 0.2, 0.3]
    o2 = ctl.control_batch_host(b, want_torques=True)
    ref2 = O.tick_swing_batch(P, b, kin=kin, threads=8)
    assert np.max(np.abs(o2["joint_tau"] - ref2["joint_tau"])) < 1e-6 * 20.0
    # the three swing inputs go together
    bad = {k: v for k, v in b.items() if k != "swing_vel"}
    with pytest.raises(RuntimeError, match="go together"):
        ctl.control_batch_host(bad, want_torques=True)


@pytest.mark.parametrize("n", [700, 20000, 70000])  # four / two / one lane(s) per robot
def test_swing_tasks_fill_every_pass_of_the_torque_pass(q, n):
    """The torque pass (qc_balance.hip) lists a wave's swinging (robot, leg) pairs and runs them in consecutive lanes, 64 per
    pass, requesting the next pass's inputs while the current one computes.  Config-3 contact states give one or two passes;
    here every contact pattern occurs - all four legs swinging included (256 tasks in a 64-robot wave: four passes; such a
    robot's QP is empty and solves with zero forces) - so every pass count and every position of the list is exercised, with
    the stateless references (swing_pos / swing_vel) against the oracle's composition."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    b = W.with_swing_references(W.with_joint_angles(W.config3(n, seed=0x5EED00B7)))
    rng = np.random.default_rng(41)
    st = (rng.random((n, 4)) < 0.45).astype(np.uint8)
    st[: n // 8] = 0                      # whole waves of robots with no stance leg
    st[n // 8: n // 4] = rng.integers(0, 2, (n // 4 - n // 8, 1), dtype=np.uint8)  # ... and of all-or-nothing robots
    b["stance"] = np.ascontiguousarray(st)
    ctl = q.BalanceController.from_params(P)
    if n == 20000:
        ctl.set_tuning(group=2)
    o = ctl.control_batch_host(b, want_torques=True)
    ref = O.tick_swing_batch(P, b, threads=8)
    assert np.array_equal(o["status"], ref["status"]) and (o["status"] == 0).mean() > 0.9
    scale = np.maximum(1.0, np.abs(ref["grf_body"]).max(axis=1, keepdims=True))
    assert np.max(np.abs(o["grf_body"] - ref["grf_body"]) / scale) < 1e-6
    assert np.max(np.abs(o["joint_tau"] - ref["joint_tau"])) < 1e-6 * 20.0
    assert np.all(o["grf_body"][st.sum(1) == 0] == 0.0) and np.any(o["joint_tau"][st.sum(1) == 0] != 0.0)


def test_swing_reference_inside_the_inner_reach_limit(q):
    """Round-4 tick fuzz finding: a swing reference closer to the hip than | |l2| - |l3| | (knee cosine d < -1, which
    legInverseKinematics does not clamp, kinematics.cpp:131-134) makes q3, q2 and with them the whole leg's torques NaN in the
    reference; the device's trig-free IK kept cos q3 = d finite and returned numbers for two of the three joints.  All three
    must be NaN, the leg's neighbours and the robot's forces untouched, on every lane-group width."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    n = 3000
    b = W.with_swing_references(W.with_joint_angles(W.config3(n)))
    hip = np.array([[-0.196, 0.05, 0.0], [0.196, 0.05, 0.0], [-0.196, -0.05, 0.0], [0.196, -0.05, 0.0]])
    rng = np.random.default_rng(8)
    inside = rng.random((n, 4)) < 0.3
    off = np.stack([rng.uniform(-0.012, 0.012, (n, 4)), np.where(np.arange(4) < 2, 0.077, -0.077)[None] + rng.uniform(-0.004, 0.004, (n, 4)),
                    rng.uniform(-0.012, 0.012, (n, 4))], axis=-1)  # within ~1.7 cm of the hip-roll offset: d < -1 (inner limit 1.9 cm) or just outside
    R = b["Rwb"].reshape(n, 3, 3)
    pos_in = np.einsum("nij,nkj->nki", R, hip[None] + off + b["x"][:, None, :])
    b["swing_pos"] = np.ascontiguousarray(np.where(inside[..., None], pos_in, b["swing_pos"].reshape(n, 4, 3)).reshape(n, 12))
    ref = O.tick_swing_batch(P, b, threads=8)
    sw = (b["stance"] == 0) & inside
    nan_ref = np.isnan(ref["joint_tau"].reshape(n, 4, 3))
    assert nan_ref[sw].all(axis=-1).mean() > 0.5  # most of the planted references are inside the limit
    for group in (1, 2, 4):
        o = q.BalanceController.from_params(P).set_tuning(group=group).control_batch_host(b, want_torques=True)
        tau = o["joint_tau"].reshape(n, 4, 3)
        assert np.array_equal(np.isnan(tau), nan_ref), group
        m = ~nan_ref
        assert np.abs(tau - ref["joint_tau"].reshape(n, 4, 3))[m].max() < 1e-6 * 20.0
        assert np.array_equal(o["status"], ref["status"]) and np.isfinite(o["grf_body"]).all()


@pytest.mark.parametrize("n", [900, 70000])  # 4 lanes per robot (one foot per lane) and one lane per robot
def test_swing_reference_out_of_reach_takes_pinv(q, n):
    """legJacobianInverse's second branch (kinematics.cpp:194-196): a swing reference the leg cannot reach makes
    legInverseKinematics clamp d to 1 (:131-134) - knee straight, J of rank 2, or rank 1 when the lateral clamp
    (:137-140) acts too - and the reference's joint-velocity target is arma::pinv(J) v.  Device (complete-pivoting
    full-rank factorisation) against the oracle (Jacobi SVD, Armadillo's tolerance); unclamped torques so that the
    comparison sees the values, not the saturation."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    b = W.with_swing_references(W.with_joint_angles(W.config3(n)))
    rng = np.random.default_rng(77)
    hip = np.array([[-0.196, 0.05, 0.0], [0.196, 0.05, 0.0], [-0.196, -0.05, 0.0], [0.196, -0.05, 0.0]])
    direction = rng.normal(size=(n, 4, 3))
    direction[::5, :, 1:] *= 0.05                       # nearly along x: the lateral clamp as well -> rank 1
    direction /= np.linalg.norm(direction, axis=-1, keepdims=True)
    far = rng.random((n, 4)) < 0.5                      # half of the legs get an unreachable target
    pb = hip[None] + direction * rng.uniform(0.55, 1.5, (n, 4, 1))   # reach: 0.077 + 0.211 + 0.230
    R = b["Rwb"].reshape(n, 3, 3)
    pos_far = np.einsum("nij,nkj->nki", R, pb + b["x"][:, None, :])  # commander_node.cpp:492 undone: pos = Rwb (p_b + x)
    pos = np.where(far[..., None], pos_far, b["swing_pos"].reshape(n, 4, 3))
    b["swing_pos"] = np.ascontiguousarray(pos.reshape(n, 12))
    kin = O.default_kinematics()
    kin.tau_min, kin.tau_max = -1.0e9, 1.0e9
    ctl = q.BalanceController.from_params(P)
    ctl.set_kinematics(tau_min=-1.0e9, tau_max=1.0e9)
    o = ctl.control_batch_host(b, want_torques=True)
    ref = O.tick_swing_batch(P, b, kin=kin, threads=8)
    sw = (b["stance"] == 0) & far
    assert sw.sum() > n // 4
    tau, rt = o["joint_tau"].reshape(n, 4, 3), ref["joint_tau"].reshape(n, 4, 3)
    assert np.isfinite(tau[sw]).all()
    scale = np.maximum(20.0, np.abs(rt).max(axis=-1, keepdims=True))
    assert np.max(np.abs(tau - rt) / scale) < 1e-6
    # the pseudo-inverse is what is in there: with the measured state at the IK solution only kd * pinv(J) v remains
    i, leg = np.argwhere(sw)[0]
    pbl = R[i].T @ pos[i, leg] - b["x"][i]
    qr = O.leg_ik(leg, pbl)
    J = O.leg_jacobian(leg, qr)
    want = np.array(kin.jc_kp) * 0.0 + np.array(kin.jc_kd) * (np.linalg.pinv(J, rcond=1e-15) @ (R[i].T @ b["swing_vel"].reshape(n, 4, 3)[i, leg])
                                                             - b["joint_qdot"].reshape(n, 4, 3)[i, leg])
    wrap = lambda a: (a + np.pi) % (2 * np.pi) - np.pi
    want = want + np.array(kin.jc_kp) * wrap(np.mod(qr, 2 * np.pi) - np.mod(b["joint_q"].reshape(n, 4, 3)[i, leg], 2 * np.pi))
    np.testing.assert_allclose(tau[i, leg], want, atol=1e-6 * max(20.0, np.abs(want).max()))


def test_nearly_straight_knee_inverts_like_the_reference(q):
    """ADVICE r3 / VERDICT r3 item 7: swing references that leave the knee some tens of ulps of its cosine (and more) from full
    stretch (sin q3 = 1e-7 ... 1e-3; closer than ~10 ulps the rounding of the knee cosine d itself - contracted arithmetic on the
    device, plain C in the oracle, whatever the reference's compiler did - decides between d < 1 and the clamp d = 1, i.e.
    between a saturated torque and the pseudo-inverse, on any pair of implementations: tools/knee_ulps.py shows it).
    |det J| is then 2e-9 or more - far above max(epsilon, 64 epsilon (sum |l|)^3), the bound
    below which device and oracle answer with pinv - so both take arma::inv's closed-form branch, as the reference does:
    joint-velocity targets up to 1e8 rad/s whose torques commander_node.cpp:526 clamps.  (Rounds 2-3 switched to the
    rank-2 pseudo-inverse at |det| <= 1e-9 (sum |l|)^3 and the device alone dropped sigma_3 there.)  Clamped torques must
    agree everywhere; unclamped ones wherever one ulp of the knee cosine - which the device's contracted arithmetic and
    the oracle's plain C round differently - is below the tolerance, i.e. from sin q3 = 1e-4 up."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    n = 8192
    b = W.with_swing_references(W.with_joint_angles(W.config3(n)))
    rng = np.random.default_rng(5)
    theta = np.array([1e-7, 3e-7, 1e-6, 1e-5, 1e-4, 1e-3])[rng.integers(0, 6, (n, 4))]
    qt = np.stack([rng.uniform(-0.3, 0.3, (n, 4)), rng.uniform(0.2, 1.0, (n, 4)), -theta], axis=-1)
    kin = O.default_kinematics()
    pb = np.array([[O.leg_fk(leg, qt[i, leg], kin) for leg in range(4)] for i in range(n)])
    R = b["Rwb"].reshape(n, 3, 3)
    b["swing_pos"] = np.ascontiguousarray(np.einsum("nij,nkj->nki", R, pb + b["x"][:, None, :]).reshape(n, 12))  # pos = Rwb (p_b + x)
    sw = b["stance"] == 0
    for limit in (20.0, 1.0e12):
        kin.tau_min, kin.tau_max = -limit, limit
        ctl = q.BalanceController.from_params(P)
        ctl.set_kinematics(tau_min=-limit, tau_max=limit)
        o = ctl.control_batch_host(b, want_torques=True)
        ref = O.tick_swing_batch(P, b, kin=kin, threads=8)
        tau, rt = o["joint_tau"].reshape(n, 4, 3), ref["joint_tau"].reshape(n, 4, 3)
        assert np.isfinite(tau[sw]).all() and np.array_equal(o["status"], ref["status"])
        if limit == 20.0:
            assert np.max(np.abs(tau - rt)[sw]) < 1e-6 * 20.0
            assert (np.abs(rt[sw]).max(axis=-1) == 20.0).mean() > 0.9  # saturated: that is what a nearly singular J does in the reference
        else:
            m = sw & (theta >= 1e-4)
            scale = np.maximum(20.0, np.abs(rt).max(axis=-1, keepdims=True))
            assert np.max((np.abs(tau - rt) / scale)[m]) < 1e-6
            assert np.abs(rt[sw & (theta < 1e-6)]).max() > 1e6 and np.abs(tau[sw & (theta < 1e-6)]).max() > 1e6  # the inverse, not a rank-2 pseudo-inverse


@pytest.mark.parametrize("n", [600, 36000, 140000])  # G = 4; G = 2 single fill; G = 2 persistent waves (dense restock)
def test_on_device_swing_planning_multi_tick(q, n):
    """SURVEY 8f rank 4, stateful half: foothold planner + sextic swing trajectories kept in a
    per-robot state buffer across ticks; torques and the carried state track the oracle tick by tick."""
    from oracle import c_oracle as O
    from tests.test_oracle_cpu import _planned_batch

    P = q.cheetah_params(0.6)
    ctl = q.BalanceController.from_params(P)
    dev_state = q.new_swing_states(n)
    ref_state = O.new_swing_states(n)
    for tick in range(0, 200, 8 if n < 100000 else 40):
        b = _planned_batch(n, tick)
        o = ctl.control_batch_host(dict(b, swing_state=dev_state), want_torques=True)
        ref = O.tick_planned_batch(P, b, ref_state, threads=8)
        assert (o["status"] == 0).all()
        assert np.array_equal(dev_state["leg_state"], ref_state["leg_state"]) and np.array_equal(dev_state["has_traj"], ref_state["has_traj"])
        m = ref_state["has_traj"].repeat(3, axis=1) == 1
        assert np.max(np.abs(dev_state["p_start"][m] - ref_state["p_start"][m])) < 1e-9
        assert np.max(np.abs(dev_state["p_final"][m] - ref_state["p_final"][m])) < 1e-9
        assert np.max(np.abs(o["joint_tau"] - ref["joint_tau"])) < 2e-5, tick
    assert (ref_state["has_traj"] == 1).any()


@pytest.mark.parametrize("force", ["force_general", "force_dense"])
def test_fused_tick_on_general_and_dense_forms(q, monkeypatch, force):
    """the joint_q / joint_tau / swing extensions are compiled into every kernel form"""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    b = W.with_swing_references(W.with_joint_angles(W.config3(3000)))
    ctl = q.BalanceController.from_params(P).set_tuning(**{force: 1})
    assert ctl.kernel_name in ("diagW-6x6", "dense-12x12")
    o = ctl.control_batch_host(b, want_torques=True)
    ref = O.tick_swing_batch(P, b, threads=8)
    assert (o["status"] == 0).all()
    scale = np.maximum(1.0, np.abs(ref["grf_body"]).max(axis=1, keepdims=True))
    assert np.max(np.abs(o["grf_body"] - ref["grf_body"]) / scale) < 1e-6
    assert np.max(np.abs(o["joint_tau"] - ref["joint_tau"])) < 2e-5


@pytest.mark.parametrize("form", ["force_general", "force_dense", "dense_w"])
@pytest.mark.parametrize("n", [700, 40000])
def test_stateful_tick_on_general_and_dense_forms(q, form, n):
    """the COMPLETE tick (planner state carried across ticks, swing trajectories, IK, joint PD, QP, J^T) on the
    general 6x6 form and on the dense 12x12 form - forced on the reference's weights, and selected by a W that really is not
    diagonal (balance_controller.hpp:76-77: W is the caller's) - tracks the oracle tick by tick like the uniform form does
    (test_on_device_swing_planning_multi_tick).  700 robots: four lanes per robot; 40 000: one lane per robot, Hessian planes in LDS."""
    from oracle import c_oracle as O
    from tests.test_oracle_cpu import _planned_batch

    P = q.cheetah_params(0.6)
    tune = {}
    if form == "dense_w":
        A = np.random.default_rng(3).normal(size=(12, 12))
        P["W"] = P["W"] + 2e-6 * A @ A.T
    else:
        tune[form] = 1
    ctl = q.BalanceController.from_params(P).set_tuning(**tune)
    assert ctl.kernel_name == ("diagW-6x6" if form == "force_general" else "dense-12x12")
    if form != "force_general":
        assert ctl.query_launch(n, kin=True)["lanes_per_robot"] == (4 if n == 700 else 1)
    dev_state = q.new_swing_states(n)
    ref_state = O.new_swing_states(n)
    for tick in range(0, 120, 12):
        b = _planned_batch(n, tick)
        o = ctl.control_batch_host(dict(b, swing_state=dev_state), want_torques=True)
        ref = O.tick_planned_batch(P, b, ref_state, threads=8)
        assert (o["status"] == 0).all()
        assert np.array_equal(dev_state["leg_state"], ref_state["leg_state"]) and np.array_equal(dev_state["has_traj"], ref_state["has_traj"])
        m = ref_state["has_traj"].repeat(3, axis=1) == 1
        assert np.max(np.abs(dev_state["p_start"][m] - ref_state["p_start"][m])) < 1e-9
        assert np.max(np.abs(dev_state["p_final"][m] - ref_state["p_final"][m])) < 1e-9
        scale = np.maximum(1.0, np.abs(ref["grf_body"]).max(axis=1, keepdims=True))
        assert np.max(np.abs(o["grf_body"] - ref["grf_body"]) / scale) < 1e-6, tick
        assert np.max(np.abs(o["joint_tau"] - ref["joint_tau"])) < 2e-5, tick
    assert (ref_state["has_traj"] == 1).any()


def test_every_contact_pattern(q):
    """all 16 stance patterns (incl. a single foot and no foot on the ground) against the oracle"""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    b = W.config2(16 * 64)
    pat = np.array([[(k >> i) & 1 for i in range(4)] for k in range(16)], dtype=np.uint8)
    b["stance"] = np.ascontiguousarray(np.repeat(pat, 64, axis=0))
    o = q.BalanceController.from_params(P).control_batch_host(b)
    ref, st, _ = O.control_batch(P, b, threads=8)
    assert (o["status"] == 0).all() and (st == 0).all()
    scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
    assert np.max(np.abs(o["grf_body"] - ref) / scale) < 1e-6
    assert np.all(o["grf_body"][np.repeat(b["stance"] == 0, 3, axis=1)] == 0.0)


def test_independent_handles_on_concurrent_streams(q):
    """distinct handles are independent (include/qc_balance.h): two controllers with different
    parameters launched back to back on two streams give the same results as run alone."""
    import torch

    from quadruped_control_amd import workloads as W

    Pa, Pb = q.cheetah_params(0.6), q.cheetah_params(0.9)
    Pb["fzmax"] = 80.0
    a, b = q.BalanceController.from_params(Pa), q.BalanceController.from_params(Pb)
    da, db = q.to_device(W.config3(50000)), q.to_device(W.config2(30000))
    ra = a.control_batch(da)["grf_body"].clone()
    rb = b.control_batch(db)["grf_body"].clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(5):
        oa = a.control_batch(da, stream=sa)
        ob = b.control_batch(db, stream=sb)
        outs.append((oa, ob))
    torch.cuda.synchronize()
    for oa, ob in outs:
        assert torch.equal(oa["grf_body"], ra) and torch.equal(ob["grf_body"], rb)


def test_small_batch_in_place_host_path_matches_staged(q):
    """qc_control_batch_host serves n <= 8192 from a pinned buffer the kernel reads/writes in place and larger
    batches through staged copies: the same robots must come back bit-identical either way, for the plain
    call and for the complete tick (swing state updated in place on the host)."""
    from tests.test_oracle_cpu import _planned_batch
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    # race=0: one pivoting strategy at every size, so that the sub-batches run the same arithmetic as the big batch
    # (with racing strategies a robot's working-set path, hence its iteration count and last bits, depends on the
    # batch size the planner saw - this test is about the two host paths)
    ctl = q.BalanceController.from_params(P).set_tuning(race=0)
    big = W.config3(9000)  # staged
    ref = ctl.control_batch_host(big, want_active_set=True, want_iterations=True)
    for lo, k in ((0, 1), (1, 7), (64, 64), (300, 33), (700, 8192)):  # in place
        part = ctl.control_batch_host({f: v[lo:lo + k] for f, v in big.items()}, want_active_set=True, want_iterations=True)
        for name in ("grf_body", "status", "active_set", "iterations"):
            assert np.array_equal(part[name], ref[name][lo:lo + k]), (name, lo, k)
    warm = ctl.control_batch_host({f: v[:40] for f, v in big.items()}, warm=ref["active_set"][:40], want_iterations=True)
    assert np.abs(warm["grf_body"] - ref["grf_body"][:40]).max() < 1e-7 and warm["iterations"].max() <= 1
    n = 8500
    full_state, part_state = q.new_swing_states(n), q.new_swing_states(n)
    for tick in (0, 8, 16, 24):
        b = _planned_batch(n, tick)
        o = ctl.control_batch_host(dict(b, swing_state=full_state), want_torques=True)  # staged
        for lo, k in ((0, 64), (64, 1), (65, 50), (115, 8192), (8307, 193)):  # in place
            sub = {f: v[lo:lo + k] for f, v in b.items()}
            st = part_state[lo:lo + k].copy()
            p = ctl.control_batch_host(dict(sub, swing_state=st), want_torques=True)
            part_state[lo:lo + k] = st
            assert np.array_equal(p["joint_tau"], o["joint_tau"][lo:lo + k]) and np.array_equal(p["grf_body"], o["grf_body"][lo:lo + k])
            assert st.tobytes() == full_state[lo:lo + k].tobytes()


def test_rotation_log_branches(q):
    """Orientation errors that walk every branch of the matrix -> quaternion -> angle-axis conversion
    (rigid3d.cpp:198-203 via Eigen): angle 0, tiny, near pi and exactly pi about each axis (trace <= 0, each
    diagonal pivot), both signs.  The device follows the same convention as the oracle."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    P["kp_w"] = np.full(3, 20.0)  # keep the commanded moments inside what the cone can deliver
    axes = [np.array(a, dtype=float) for a in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1], [1, 0, 1], [1, 1, 1], [1, -2, 3], [-3, 1, 2])]
    angles = [0.0, 1e-12, 1e-9, 1e-5, 0.5, np.pi / 2, 2.0, 3.0, np.pi - 1e-3, np.pi - 1e-7, np.pi - 1e-12, np.pi]
    rvs = np.array([s * a / np.linalg.norm(a) * t for a in axes for t in angles for s in (1.0, -1.0)])
    n = len(rvs)
    b = W.config2(n)
    Rd = W.rotvec_to_matrix(rvs)
    # exact pi rotations as exact matrices (2 a a^T - I), so trace = -1 without rounding
    for i, rv in enumerate(rvs):
        if abs(np.linalg.norm(rv) - np.pi) < 1e-15:
            a = rv / np.linalg.norm(rv)
            Rd[i] = 2.0 * np.outer(a, a) - np.eye(3)
    R = b["Rwb"].reshape(n, 3, 3)
    b["Rwb_d"] = np.ascontiguousarray((Rd @ R).reshape(n, 9))  # R_err = Rwb_d Rwb^T = Rd
    o = q.BalanceController.from_params(P).control_batch_host(b)
    ref, st, _ = O.control_batch(P, b, threads=4)
    assert np.array_equal(o["status"], st)
    ok = st == 0
    assert ok.sum() > n // 2
    # At exactly pi about a non-coordinate axis the sign of the axis is decided by the last-bit rounding of
    # Rwb_d Rwb^T (w ~ 1e-17): there the log map is discontinuous for any implementation, Eigen's included
    # (SURVEY 8a').  Those robots must still solve; the comparison holds everywhere else, pi - 1e-12 included.
    coord = np.array([np.count_nonzero(rv) == 1 for rv in rvs])
    ambiguous = (np.abs(np.linalg.norm(rvs, axis=1) - np.pi) < 1e-15) & ~coord
    assert ambiguous.sum() == 12 and np.isfinite(o["grf_body"]).all()
    scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
    assert np.max((np.abs(o["grf_body"] - ref) / scale)[ok & ~ambiguous]) < 1e-6


def test_single_robot_calls_hot_start_from_the_previous_tick(q):
    """qc_control keeps the working set of its previous call in the handle (the reference's per-object hot start,
    balance_controller.cpp:191-202): a sequence of ticks through one controller gives the same forces as a fresh
    controller per tick, also across contact changes and after a failed call."""
    from quadruped_control_amd import workloads as W
    from quadruped_control_amd.gait import LEG_NAMES, LegState

    P = q.cheetah_params(0.6)
    t0, t1 = W.config4(48)
    seq = q.BalanceController.from_params(P)
    worst = 0.0
    for tick, b in enumerate((t0, t1, t0, t1)):
        for i in range(48):
            a = [b[k][i] for k in ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d")]
            feet = {nm: b["feet"][i].reshape(4, 3)[j] for j, nm in enumerate(LEG_NAMES)}
            gait = {nm: ((LegState.stance if (i + tick + j) % 3 else LegState.swing), 0.0) for j, nm in enumerate(LEG_NAMES)}
            got = seq.control(*a, feet, gait)
            ref = q.BalanceController.from_params(P).control(*a, feet, gait) if i % 6 == 0 else None
            if ref is not None:
                assert set(got) == set(ref)
                for nm in ref:
                    worst = max(worst, float(np.abs(got[nm] - ref[nm]).max()))
            if i == 20:  # a failed call (NaN state) must not poison the next one
                bad = list(a); bad[2] = np.array([np.nan, 0.0, 0.26])
                assert seq.control(*bad, feet, gait) == {}
    assert worst < 1e-6


def test_unphysical_inputs_match_the_oracle(q):
    """The reference (Release build) checks nothing about its inputs; whatever arithmetic it would do on garbage,
    device and oracle do the same: non-orthonormal / random 'rotations', feet at the COM, four coincident feet,
    kilometre legs, denormals, and non-finite states (a failed instance on both sides, never NaN forces)."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    ctl = q.BalanceController.from_params(P)
    rng = np.random.default_rng(5)
    n = 2048
    b = W.config3(n)
    cases = {}
    cases["scaled"] = dict(b, Rwb=np.ascontiguousarray(b["Rwb"] * rng.uniform(0.5, 1.5, (n, 1))))
    cases["random"] = dict(b, Rwb=np.ascontiguousarray(rng.normal(size=(n, 9))), Rwb_d=np.ascontiguousarray(rng.normal(size=(n, 9))))
    cases["at_com"] = dict(b, feet=np.zeros_like(b["feet"]))
    f = b["feet"].copy().reshape(n, 4, 3); f[:, 1:] = f[:, :1]
    cases["coincident"] = dict(b, feet=np.ascontiguousarray(f.reshape(n, 12)))
    cases["km_legs"] = dict(b, feet=np.ascontiguousarray(b["feet"] * 1e3))
    cases["denormal"] = dict(b, x=np.ascontiguousarray(b["x"] + 1e-310))
    x = b["x"].copy(); x[::7, 0] = np.inf; x[3::7, 2] = np.nan
    cases["nonfinite"] = dict(b, x=x)
    for name, c in cases.items():
        o = ctl.control_batch_host(c)
        ref, st, _ = O.control_batch(P, c, threads=8)
        assert np.array_equal(o["status"], st), name
        assert np.isfinite(o["grf_body"]).all(), name
        ok = st == 0
        scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
        tol = 1e-4 if name == "km_legs" else 1e-6  # km legs: cond(A) ~ 1e6, still inside the north-star bar
        assert np.max((np.abs(o["grf_body"] - ref) / scale)[ok]) < tol, name
        assert np.all(o["grf_body"][~ok] == 0.0), name
    assert (ctl.control_batch_host(cases["nonfinite"])["status"][::7] == 3).all()


def test_non_finite_joint_states_in_the_widened_tick(q):
    """NaN / inf joint angles or swing references: the QP instance fails on both sides (status 3, zero GRFs) and
    the swing-leg torques of such legs come out NaN exactly where the reference's arithmetic (two-compare
    arma::clamp, commander_node.cpp:526) leaves them NaN - never as a clamped full-scale command."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    n = 1024
    b = W.with_swing_references(W.with_joint_angles(W.config3(n)))
    jq = b["joint_q"].copy(); jq[::5, 4] = np.nan; jq[2::5, 0] = np.inf
    sp = b["swing_pos"].copy(); sp[1::9, 7] = np.nan
    b = dict(b, joint_q=jq, swing_pos=sp)
    o = q.BalanceController.from_params(P).control_batch_host(b, want_torques=True)
    ref = O.tick_swing_batch(P, b, threads=8)
    assert np.array_equal(o["status"], ref["status"]) and (o["status"][::5] == 3).all()
    assert np.isfinite(o["grf_body"]).all() and np.all(o["grf_body"][o["status"] != 0] == 0.0)
    assert np.array_equal(np.isnan(o["joint_tau"]), np.isnan(ref["joint_tau"])) and np.isnan(o["joint_tau"]).any()
    m = ~np.isnan(ref["joint_tau"])
    assert np.abs(o["joint_tau"] - ref["joint_tau"])[m].max() < 2e-5


def test_huge_finite_swing_references_saturate_like_the_reference(q):
    """a FINITE swing reference beyond ~1e154 m overflows its own square.  The reference clamps the knee cosine
    d = inf to 1 (kinematics.cpp:131-134) and commands finite angles - a saturated, finite torque; the device's trig-free IK
    must hand such targets to its reference-shaped evaluation instead of returning NaN from rsqrt(inf)."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    n = 2048
    b = W.with_swing_references(W.with_joint_angles(W.config3(n)))
    sp = b["swing_pos"].copy()
    for k, (col, val) in enumerate(((0, 1e160), (1, -1e200), (2, 3e155), (4, 1e300), (8, -2e180), (9, 1e154), (11, 1e153))):
        sp[k::7, col] = val
    b = dict(b, swing_pos=sp)
    o = q.BalanceController.from_params(P).control_batch_host(b, want_torques=True)
    ref = O.tick_swing_batch(P, b, threads=8)
    assert np.array_equal(o["status"], ref["status"])
    assert np.array_equal(np.isnan(o["joint_tau"]), np.isnan(ref["joint_tau"]))
    swing = np.repeat(b["stance"] == 0, 3, axis=1)
    assert np.isfinite(ref["joint_tau"][swing]).mean() > 0.9  # the reference saturates, it does not give up
    m = ~np.isnan(ref["joint_tau"])
    assert np.abs(o["joint_tau"] - ref["joint_tau"])[m].max() < 2e-5


def test_non_finite_states_in_the_stateful_tick(q):
    """NaN velocities / inf positions / NaN gait phases over a sequence of planned ticks: the carried swing state,
    the statuses and the NaN pattern of the torques follow the oracle (std::clamp of the trajectory time,
    trajectory.cpp:369, keeps a NaN phase NaN)."""
    from oracle import c_oracle as O
    from tests.test_oracle_cpu import _planned_batch

    P = q.cheetah_params(0.6)
    ctl = q.BalanceController.from_params(P)
    n = 512
    dev, ref = q.new_swing_states(n), O.new_swing_states(n)
    saw_nan = False
    for tick in range(0, 120, 12):
        b = _planned_batch(n, tick)
        xd = b["xdot"].copy(); xd[::11, 0] = np.nan
        xx = b["x"].copy(); xx[5::13, 1] = np.inf
        gp = b["gait_phase"].copy(); gp[7::17, 2] = np.nan
        b = dict(b, xdot=xd, x=xx, gait_phase=gp)
        o = ctl.control_batch_host(dict(b, swing_state=dev), want_torques=True)
        r = O.tick_planned_batch(P, b, ref, threads=8)
        assert np.array_equal(o["status"], r["status"])
        assert np.array_equal(dev["leg_state"], ref["leg_state"]) and np.array_equal(dev["has_traj"], ref["has_traj"])
        m = ref["has_traj"].repeat(3, axis=1) == 1
        for k in ("p_start", "p_final"):
            a, c = dev[k][m], ref[k][m]
            assert np.array_equal(np.isfinite(a), np.isfinite(c)) and np.array_equal(np.isnan(a), np.isnan(c))
            ok = np.isfinite(c)  # inf - inf would only produce a warning
            assert np.max(np.abs(a[ok] - c[ok])) < 1e-9 and np.array_equal(a[~ok & ~np.isnan(c)], c[~ok & ~np.isnan(c)])
        assert np.array_equal(np.isnan(o["joint_tau"]), np.isnan(r["joint_tau"]))
        fin = ~np.isnan(r["joint_tau"])
        assert np.abs(o["joint_tau"] - r["joint_tau"])[fin].max() < 2e-5
        saw_nan |= bool(np.isnan(o["joint_tau"]).any())
    assert saw_nan


def test_on_device_gait_clock(q):
    """ABI v3: `gait_dt` advances the per-leg phases on the device the way GaitScheduler::update does
    (gait.cpp:113-123) before the contact rule, the planner and the trajectories of the tick use them; the
    phases carried by the device over many ticks stay bit-equal to the oracle's clock and the complete tick
    tracks the oracle fed with those phases."""
    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    n = 3000
    rng = np.random.default_rng(8)
    base = W.with_swing_references(W.with_joint_angles(W.config3(n)))
    base = {k: v for k, v in base.items() if k not in ("stance", "swing_pos", "swing_vel")}
    for t_sw, t_st in ((0.18, 0.8), (0.3, 0.3)):
        kin = O.default_kinematics(); kin.t_swing = t_sw; kin.t_stance = t_st
        ctl = q.BalanceController.from_params(P)
        ctl.set_gait(t_sw, t_st)
        dev_phase = np.ascontiguousarray(np.fmod(np.array([0.0, 0.5, 0.5, 0.0])[None] + rng.uniform(0, 1, (n, 1)), 1.0))
        ref_phase = dev_phase.copy()
        dev_state, ref_state = q.new_swing_states(n), O.new_swing_states(n)
        for tick in range(40):
            dt = rng.uniform(0.0, 0.02, n)
            b = dict(base, gait_phase=dev_phase, gait_dt=dt, swing_state=dev_state)
            o = ctl.control_batch_host(b, want_torques=True)
            O.gait_update(ref_phase, dt, kin=kin)
            assert np.array_equal(dev_phase, ref_phase), tick  # the device wrote the advanced clock back
            r = O.tick_planned_batch(P, dict(base, gait_phase=ref_phase), ref_state, kin=kin, threads=8)
            assert np.array_equal(o["status"], r["status"])
            assert np.array_equal(dev_state["leg_state"], ref_state["leg_state"]) and np.array_equal(dev_state["has_traj"], ref_state["has_traj"])
            assert np.max(np.abs(o["joint_tau"] - r["joint_tau"])) < 2e-5, tick
        assert (ref_state["has_traj"] == 1).any()
    # plain path: clock + contact rule only
    ph = np.ascontiguousarray(rng.uniform(0, 1, (n, 4))); ph0 = ph.copy()
    dt = rng.uniform(0.0, 0.5, n)
    b2 = {k: v for k, v in W.config2(n).items() if k != "stance"}
    out = ctl.control_batch_host(dict(b2, gait_phase=ph, gait_dt=dt))
    want = O.gait_update(ph0, dt, kin=kin)
    assert np.array_equal(ph, want)
    from quadruped_control_amd import leg_state_from_phase
    ref = ctl.control_batch_host(dict(b2, stance=leg_state_from_phase(want, 0.5)))
    assert np.array_equal(out["grf_body"], ref["grf_body"])
    with pytest.raises(RuntimeError, match="gait_dt"):
        ctl.control_batch_host(dict(W.config2(8), gait_dt=np.zeros(8)))


def test_plan_batch_does_not_advance_the_stateful_tick(q):
    """plan_batch() only validates and marshals.  With the on-device gait clock and the swing planner
    (both in/out state) plan_batch followed by ONE launch() must equal ONE control_batch(): same advanced phases,
    same swing state, same torques - the first launch() is tick 1, not tick 2."""
    import torch

    from quadruped_control_amd import workloads as W

    n = 3000
    P = q.cheetah_params(0.6)
    rng = np.random.default_rng(5)
    b = W.with_swing_references(W.with_joint_angles(W.config3(n)))
    b = {k: v for k, v in b.items() if k not in ("stance", "swing_pos", "swing_vel")}
    b["gait_phase"] = np.ascontiguousarray(np.fmod(np.array([0.0, 0.5, 0.5, 0.0])[None] + rng.uniform(size=(n, 1)), 1.0))
    b["gait_dt"] = np.full(n, 1.0 / 300.0)

    def fresh():
        d = q.to_device(b)
        d["gait_phase"] = d["gait_phase"].clone()
        d["swing_state"] = torch.from_numpy(q.new_swing_states(n).view("uint8").reshape(-1).copy()).to("cuda:0")
        return d

    ctl = q.BalanceController.from_params(P)
    d1 = fresh()
    o1 = ctl.control_batch(d1, want_torques=True)
    torch.cuda.synchronize()
    d2 = fresh()
    launch, o2 = ctl.plan_batch(d2, want_torques=True)
    torch.cuda.synchronize()
    assert torch.equal(d2["gait_phase"].cpu(), torch.from_numpy(b["gait_phase"]))  # planning launched nothing
    launch()
    torch.cuda.synchronize()
    assert torch.equal(d1["gait_phase"], d2["gait_phase"]) and not torch.equal(d2["gait_phase"].cpu(), torch.from_numpy(b["gait_phase"]))
    assert torch.equal(d1["swing_state"], d2["swing_state"])
    assert torch.equal(o1["joint_tau"], o2["joint_tau"]) and torch.equal(o1["grf_body"], o2["grf_body"])


def test_on_device_generation_matches_host(q):
    """config 5's shard generated on the GPU == the host generator (contact states identical, floats to a
    few ulps), and the controller's forces on it match the oracle's on the host-generated robots."""
    import torch

    from oracle import c_oracle as O
    from quadruped_control_amd import workloads as W
    from quadruped_control_amd import workloads_device as D

    n, start = 20000, 1048576 - 10000  # straddles the boundary between two of the eight shards
    h = W.config5(n, start=start)
    d = D.config5(n, start=start, device=0)
    for k in h:
        dd = d[k].cpu().numpy()
        if k == "stance":
            assert np.array_equal(dd, h[k])
        else:
            assert np.max(np.abs(dd - h[k])) < 1e-14, k
    P = q.cheetah_params(0.6)
    o = q.BalanceController.from_params(P).control_batch(d)
    torch.cuda.synchronize()
    ref, st, _ = O.control_batch(P, h, threads=8)
    assert int((o["status"] != 0).sum()) == 0 and (st == 0).all()
    scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
    assert np.max(np.abs(o["grf_body"].cpu().numpy() - ref) / scale) < 1e-6


def test_tuning_overrides_restore_what_the_handle_was_created_with(q):
    """qc_set_tuning's overrides restore the creation-time values - the handle's own iteration cap after the
    batch-load probe or max_iter <= 0 (not a hard-coded 200), and the weights' own formulation after any order of
    force_general / force_dense calls (not a form re-derived from the current flags)."""
    from quadruped_control_amd import workloads as W

    P = q.cheetah_params(0.6)
    b = W.config3(4096, seed=0x5EED00B7)
    fresh = q.BalanceController.from_params(P, max_iter=3).control_batch_host(b, want_iterations=True)
    assert (fresh["status"] == 1).any() and int(fresh["iterations"].max()) == 3
    ctl = q.BalanceController.from_params(P, max_iter=3)
    ctl.set_tuning(probe_batch_load=1)
    probe = ctl.control_batch_host(b)
    assert (probe["status"] == 1).all() and np.all(probe["grf_body"] == 0.0)  # load -> assemble -> store only
    ctl.set_tuning(probe_batch_load=0)
    back = ctl.control_batch_host(b, want_iterations=True)
    assert np.array_equal(back["status"], fresh["status"]) and np.array_equal(back["iterations"], fresh["iterations"])
    ctl.set_tuning(max_iter=100).set_tuning(max_iter=0)
    again = ctl.control_batch_host(b, want_iterations=True)
    assert np.array_equal(again["status"], fresh["status"]) and np.array_equal(again["iterations"], fresh["iterations"])
    # formulation: uniform weights stay uniform after a detour through the other forms, in any order
    ctl = q.BalanceController.from_params(P)
    assert ctl.query_launch(4096)["form"] == 0
    ctl.set_tuning(force_dense=1).set_tuning(force_general=0)
    assert ctl.query_launch(4096)["form"] == 2
    ctl.set_tuning(force_dense=0)
    assert ctl.query_launch(4096)["form"] == 0 and ctl.kernel_name == "diagW-6x6-uniform"
    ctl.set_tuning(force_general=1).set_tuning(force_dense=1).set_tuning(force_general=0)
    assert ctl.query_launch(4096)["form"] == 2
    ctl.set_tuning(force_dense=0)
    assert ctl.query_launch(4096)["form"] == 0


def test_paired_waves_kernel_full_size(q):
    """MODE 3 as the planner picks it (>= 524 288 robots on the 6x6 forms: two one-lane waves per workgroup, the last to
    arrive finishes both waves' stragglers from an LDS list, the last round of workgroups keeps them per wave): config 5's
    distribution cold - with a ragged last workgroup - and a warm-started config-4 tick, against the one-fill kernel
    (same minimiser: 1e-7 of max|GRF|; identical status) and the whole-batch KKT certificate."""
    import torch

    from quadruped_control_amd import workloads as W
    from quadruped_control_amd import workloads_device as WD
    from tests.kkt_batch import assert_kkt

    P = q.cheetah_params(0.6)
    n = 4 * 131072 + 64 + 9
    ctl = q.BalanceController.from_params(P)
    info = ctl.query_launch(n)
    assert (info["mode"], info["lanes_per_robot"], info["chunk"]) == (3, 1, 128), info
    assert q.BalanceController.from_params(P).query_launch(n - 200)["mode"] == 1  # below four rounds: one-fill workgroups
    b = WD.config3(n, seed=W.SEEDS[5], device=0)
    o = ctl.control_batch(b, want_iterations=True, want_active_set=True)
    ref = q.BalanceController.from_params(P).set_tuning(pair=0).control_batch(b, want_iterations=True)
    torch.cuda.synchronize()
    assert int((o["status"] != 0).sum()) == 0 and int((ref["status"] != 0).sum()) == 0
    scale = ref["grf_body"].abs().amax(dim=1, keepdim=True).clamp(min=1.0)
    assert float(((o["grf_body"] - ref["grf_body"]).abs() / scale).max()) < 1e-7
    assert int(o["iterations"].min()) >= 1 and int(o["iterations"].max()) <= 40
    host = {k: v.cpu().numpy() for k, v in b.items()}
    assert_kkt(P, host, o["grf_body"].cpu().numpy())
    # restarting from the reported working sets: one recalculation each, on the same kernel
    again = ctl.control_batch(b, warm=o["active_set"], want_iterations=True)
    torch.cuda.synchronize()
    assert int(again["iterations"].max()) == 1 and float(((again["grf_body"] - ref["grf_body"]).abs() / scale).max()) < 1e-7
    # warm-started tick (config 4's two ticks)
    t0, t1 = W.config4(n, seed=W.SEEDS[4])
    w = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
    d1 = q.to_device(t1)
    ow = ctl.control_batch(d1, warm=w, want_iterations=True)
    rw = q.BalanceController.from_params(P).set_tuning(pair=0).control_batch(d1, warm=w, want_iterations=True)
    torch.cuda.synchronize()
    assert int((ow["status"] != 0).sum()) == 0
    assert torch.equal(ow["iterations"], rw["iterations"])  # no race in a warm tail: the same walk robot by robot
    assert float((ow["grf_body"] - rw["grf_body"]).abs().max()) < 1e-9
