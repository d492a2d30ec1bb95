"""CPU: the oracle against its pins - closed-form known answers, the committed
golden vectors (scipy/NNLS-derived), cross-agreement of the two restatements,
and the KKT certificate.  The reference itself holds no fixtures for this path
(SURVEY.md 8c), hence 'parity unpinned by reference fixtures' in oracle/."""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle as O
from oracle import numpy_restatement as R
from quadruped_control_amd import workloads as W

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

GOLD = os.path.join(os.path.dirname(__file__), "golden", "balance_golden.json")
FIELDS = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "feet")


@pytest.fixture(scope="module", autouse=True)
def _build():
    O.build()


def _gold():
    with open(GOLD) as f:
        return json.load(f)["cases"]


def _params(c):
    P = R.cheetah_params(c["mu"])
    P["fzmin"], P["fzmax"] = c["fzmin"], c["fzmax"]
    return P


def _batch(cases):
    b = {k: np.array([c[k] for c in cases], dtype=np.float64) for k in FIELDS}
    b["stance"] = np.array([c["stance"] for c in cases], dtype=np.uint8)
    return b


def test_closed_form_kats():
    """KAT1-4 of SURVEY.md 8c have closed forms independent of any solver."""
    by = {c["name"]: c for c in _gold()}
    phi1 = 70.1415 / 4.00001  # b_z / (4 + w/S_zz)
    np.testing.assert_allclose(np.array(by["KAT1_nominal"]["grf_body"])[2::3], -phi1, rtol=1e-9)
    np.testing.assert_allclose(np.array(by["KAT2_fzmin_clamp"]["grf_body"])[2::3], -10.0, rtol=1e-12)
    phi3 = 70.1415 / 2.00001
    g3 = np.array(by["KAT3_trot_diag"]["grf_body"]).reshape(4, 3)
    np.testing.assert_allclose(g3[[0, 3], 2], -phi3, rtol=1e-9)
    assert np.all(g3[[1, 2]] == 0.0)
    np.testing.assert_allclose(np.array(by["KAT4_fzmax_clamp"]["grf_body"])[2::3], -15.0, rtol=1e-12)


def test_c_oracle_matches_golden():
    cases = _gold()
    groups = {}
    for c in cases:
        groups.setdefault((c["mu"], c["fzmin"], c["fzmax"]), []).append(c)
    for cs in groups.values():
        grf, st, it = O.control_batch(_params(cs[0]), _batch(cs))
        assert (st == 0).all()
        exp = np.array([c["grf_body"] for c in cs])
        scale = np.maximum(1.0, np.abs(exp).max(axis=1, keepdims=True))
        assert np.max(np.abs(grf - exp) / scale) < 1e-8


def test_numpy_restatement_matches_golden_subset():
    for c in _gold()[::9]:
        P = _params(c)
        out, fmap, fw, qp = R.control(P, np.array(c["Rwb"]).reshape(3, 3), np.array(c["Rwb_d"]).reshape(3, 3),
                                      c["x"], c["xdot"], c["w"], c["x_d"], c["xdot_d"], c["w_d"], c["feet"], c["stance"])
        np.testing.assert_allclose(out.reshape(-1), c["grf_body"], rtol=0, atol=1e-8 * max(1, np.abs(c["grf_body"]).max()))
        assert sorted(fmap) == sorted(n for n, s in zip(R.LEG_NAMES, c["stance"]) if s)
        cert = R.kkt_certificate(qp["H"], qp["g"], qp["C"], qp["lb"], qp["ub"], fw)
        assert cert["primal"] < 1e-8 and cert["stationarity"] < 1e-9


def test_assembly_c_equals_numpy_and_kkt():
    P = R.cheetah_params(0.6)
    b = W.config3(64)
    for i in range(64):
        args = [b[k][i] for k in FIELDS]
        qc = O.assemble(P, *args, b["stance"][i])
        qn = R.assemble(P, args[0].reshape(3, 3), args[1].reshape(3, 3), *args[2:], b["stance"][i])
        for k in ("H", "g", "C", "lb", "ub", "A", "b"):
            np.testing.assert_allclose(qc[k], qn[k], rtol=1e-12, atol=1e-11)
        st, f, lam, it = O.qp_solve(qc["H"], qc["g"], qc["C"], qc["lb"], qc["ub"])
        assert st == 0
        s, p, d = O.kkt(qc["H"], qc["g"], qc["C"], qc["lb"], qc["ub"], f, lam)
        assert s < 1e-7 and p < 1e-8 and d < 1e-6


def test_angle_axis_against_scipy():
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(3)
    for _ in range(200):
        ax = rng.normal(size=3)
        rv = ax / np.linalg.norm(ax) * rng.uniform(0.0, 3.1)  # log map is principal: angle in [0, pi]
        Rm = Rotation.from_rotvec(rv).as_matrix()
        np.testing.assert_allclose(O.angle_axis_total(Rm), rv, atol=1e-9)
        np.testing.assert_allclose(R.angle_axis_total(Rm), rv, atol=1e-9)
    assert np.all(O.angle_axis_total(np.eye(3)) == 0.0)
    # trace <= 0 branches (angles near pi about each axis)
    for ax in np.eye(3):
        Rm = Rotation.from_rotvec(ax * 3.1).as_matrix()
        np.testing.assert_allclose(O.angle_axis_total(Rm), ax * 3.1, atol=1e-9)
        np.testing.assert_allclose(R.angle_axis_total(Rm), ax * 3.1, atol=1e-9)


def test_failure_and_edge_cases():
    P = R.cheetah_params(0.6)
    b = W.config2(8)
    # all legs swing -> zero forces, solved
    b0 = dict(b); b0["stance"] = np.zeros((8, 4), np.uint8)
    grf, st, _ = O.control_batch(P, b0)
    assert (st == 0).all() and np.all(grf == 0.0)
    # iteration cap -> status 1 and all-zero forces (reference: empty ForceMap)
    grf, st, _ = O.control_batch(P, b, max_iter=1)
    assert (st == 1).any() and np.all(grf[st == 1] == 0.0)
    # fzmin = 0 re-admits the cone apex
    P0 = dict(P); P0["fzmin"] = 0.0
    grf, st, _ = O.control_batch(P0, W.config3(64))
    assert (st == 0).all()


def test_kinematics_against_reference_notebook():
    """The ONLY known answers the reference holds: the printed
    FK positions and Jacobians of scripts/kinematics/quadruped_kinematics.ipynb."""
    with open(os.path.join(os.path.dirname(__file__), "golden", "kinematics_notebook.json")) as f:
        g = json.load(f)
    q = np.array(g["q"])
    for leg, name in enumerate(R.LEG_NAMES):
        if name in g["fk"]:
            np.testing.assert_allclose(O.leg_fk(leg, q), g["fk"][name], atol=6e-9)  # 8 printed digits
        np.testing.assert_allclose(O.leg_jacobian(leg, q), np.array(g["jacobian"][name]), atol=6e-9)
    # Jacobian == derivative of FK (central differences)
    rng = np.random.default_rng(7)
    for _ in range(20):
        qq = rng.uniform(-1.5, 1.5, 3)
        for leg in range(4):
            J = O.leg_jacobian(leg, qq)
            h = 1e-6
            num = np.stack([(O.leg_fk(leg, qq + h * e) - O.leg_fk(leg, qq - h * e)) / (2 * h) for e in np.eye(3)], axis=1)
            np.testing.assert_allclose(J, num, atol=1e-8)


def test_tick_composition():
    """oracle_tick_batch == FK -> control -> clamp(J^T f) composed by hand."""
    P = R.cheetah_params(0.6)
    b = W.with_joint_angles(W.config3(64))
    t = O.tick_batch(P, b)
    feet = np.stack([np.concatenate([O.leg_fk(leg, b["joint_q"][i, 3 * leg:3 * leg + 3]) for leg in range(4)]) for i in range(64)])
    np.testing.assert_allclose(t["feet"], feet, atol=1e-15)
    b2 = dict(b); b2["feet"] = feet
    grf, st, _ = O.control_batch(P, b2)
    np.testing.assert_array_equal(t["grf_body"], grf)
    for i in range(64):
        for leg in range(4):
            tau = O.leg_jacobian(leg, b["joint_q"][i, 3 * leg:3 * leg + 3]).T @ grf[i, 3 * leg:3 * leg + 3]
            exp = np.clip(tau, -20.0, 20.0) if b["stance"][i, leg] else np.zeros(3)
            np.testing.assert_allclose(t["joint_tau"][i, 3 * leg:3 * leg + 3], exp, atol=1e-12)


def test_inverse_kinematics_round_trip():
    """legInverseKinematics (kinematics.cpp:117-160) inverts forwardKinematics (:81-103)
    on the knee-bent branch the reference assumes (calf angle in (-pi, 0))."""
    rng = np.random.default_rng(9)
    for _ in range(200):
        q = np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.2, 1.4), rng.uniform(-2.4, -0.3)])
        for leg in range(4):
            p = O.leg_fk(leg, q)
            np.testing.assert_allclose(O.leg_ik(leg, p), q, atol=1e-9)
            np.testing.assert_allclose(O.leg_fk(leg, O.leg_ik(leg, p)), p, atol=1e-12)


def test_swing_torque_composition():
    P = R.cheetah_params(0.6)
    b = W.with_swing_references(W.with_joint_angles(W.config3(256)))
    t = O.tick_swing_batch(P, b)
    base = O.tick_batch(P, b)
    st = np.repeat(b["stance"] == 1, 3, axis=1)
    np.testing.assert_array_equal(t["joint_tau"][st], base["joint_tau"][st])       # stance legs untouched
    assert np.all(np.abs(t["joint_tau"]) <= 20.0) and np.any(t["joint_tau"][~st] != 0.0)
    # zero velocity references and measured state at the target -> only the kff term (0) remains
    b0 = dict(b)
    b0["swing_vel"] = np.zeros_like(b["swing_vel"]); b0["joint_qdot"] = np.zeros_like(b["joint_qdot"])
    n = 256
    Rm = b["Rwb"].reshape(n, 3, 3)
    pb = np.einsum("nji,nkj->nki", Rm, b["swing_pos"].reshape(n, 4, 3)) - b["x"][:, None, :]
    b0["joint_q"] = np.stack([np.concatenate([O.leg_ik(leg, pb[i, leg]) for leg in range(4)]) for i in range(n)])
    t0 = O.tick_swing_batch(P, b0)
    assert np.max(np.abs(t0["joint_tau"][~st])) < 1e-9


def test_singular_leg_jacobian_takes_the_pinv_branch():
    """legJacobianInverse (kinematics.cpp:190-204): arma::inv, else arma::pinv, else J^T.  A swing reference out of
    reach makes legInverseKinematics clamp d to 1 (kinematics.cpp:131-134): the knee is straight, J has rank 2 (rank 1
    when y^2 + z^2 < l1^2 is clamped too, :137-140) and the reference's answer is pinv(J) v.  The oracle's pinv
    (one-sided Jacobi SVD, Armadillo's tolerance) against numpy.linalg.pinv (LAPACK gesdd), and the swing torque
    built on it against the formula of joint_controller.cpp:28-36."""
    rng = np.random.default_rng(12)
    kin = O.default_kinematics()
    hip = np.array(kin.hip).reshape(4, 3)
    seen_rank = set()
    for trial in range(400):
        leg = trial % 4
        direction = rng.normal(size=3)
        if trial % 5 == 0:
            direction[1:] *= 0.05  # nearly along x: y^2 + z^2 < l1^2, the second clamp -> rank 1
        target = hip[leg] + direction / np.linalg.norm(direction) * rng.uniform(0.55, 1.5)  # reach is 0.077 + 0.211 + 0.230
        q = O.leg_ik(leg, target)
        assert q[2] == 0.0  # knee straight: atan2(-0, 1)
        J = O.leg_jacobian(leg, q)
        sv = np.linalg.svd(J, compute_uv=False)
        rank = int((sv > 3 * sv[0] * np.finfo(float).eps).sum())
        assert rank in (1, 2)
        seen_rank.add(rank)
        Jp, ok = O.pinv3(J)
        assert ok
        ref = np.linalg.pinv(J, rcond=3 * np.finfo(float).eps)
        np.testing.assert_allclose(Jp, ref, atol=1e-12 * max(1.0, np.abs(ref).max()))
        vel = rng.normal(size=3)
        qm, qdm = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        tau = O.swing_torque(leg, np.eye(3), np.zeros(3), target, vel, qm, qdm, kin)
        wrap = lambda a: (a + np.pi) % (2 * np.pi) - np.pi
        want = np.array(kin.jc_kp) * wrap(np.mod(q, 2 * np.pi) - np.mod(qm, 2 * np.pi)) + np.array(kin.jc_kd) * (ref @ vel - qdm) + np.array(kin.jc_kff)
        np.testing.assert_allclose(tau, want, atol=1e-9)
    assert seen_rank == {1, 2}
    # a regular configuration still takes arma::inv
    q = np.array([0.2, 0.7, -1.4])
    J = O.leg_jacobian(1, q)
    tau = O.swing_torque(1, np.eye(3), np.zeros(3), O.leg_fk(1, q), np.array([0.3, -0.2, 0.1]), q, np.zeros(3), kin)
    np.testing.assert_allclose(tau, np.array(kin.jc_kd) * np.linalg.solve(J, [0.3, -0.2, 0.1]), atol=1e-9)


def _planned_batch(n, tick, dt=1.0 / 300.0):
    """config-3-like states with a trot gait clock (offsets 0,.5,.5,0; 0.8/0.18) advanced to `tick`."""
    b = W.with_joint_angles(W.config3(n))
    b = W.with_swing_references(b)  # for joint_qdot
    phi0 = W.uniform(0x5EED0008, np.arange(n, dtype=np.uint64), 7)
    ph = np.fmod(np.array([0.0, 0.5, 0.5, 0.0])[None] + phi0[:, None] + tick * dt / 0.98, 1.0)
    out = {k: v for k, v in b.items() if k not in ("stance", "swing_pos", "swing_vel")}
    out["gait_phase"] = np.ascontiguousarray(ph)
    return out


def test_swing_planner_and_trajectories():
    P = R.cheetah_params(0.6)
    n = 64
    st = O.new_swing_states(n)
    kin = O.default_kinematics()
    duty = 0.8 / 0.98
    saw_replan = False
    for tick in range(0, 240, 3):
        b = _planned_batch(n, tick)
        prev = st.copy()
        t = O.tick_planned_batch(P, b, st)
        stance = ((b["gait_phase"] >= 0) & (b["gait_phase"] <= duty + 1e-12)).astype(np.int32)
        assert np.array_equal(st["leg_state"], stance)
        # a stance -> swing edge (or a swinging leg on the first call) plans a foothold on the ground plane
        edge = (stance == 0) & ((prev["leg_state"] == 1) | (prev["leg_state"][:, :1] < 0))
        if tick > 0 and edge.any():
            saw_replan = True
        assert np.all(st["has_traj"][edge] == 1)
        assert np.all(st["p_final"].reshape(n, 4, 3)[edge][:, 2] == 0.0)
        # robots without an edge keep their trajectories untouched
        quiet = ~edge.any(axis=1)
        assert np.array_equal(st["has_traj"][quiet], prev["has_traj"][quiet])
        assert np.all(np.abs(t["joint_tau"]) <= 20.0)
    assert saw_replan
    # sextic trajectory end conditions through the public reference function (trajectory.cpp:256-277)
    s1 = O.new_swing_states(1)
    s1["leg_state"][0] = [1, 1, 1, 1]
    I, z = np.eye(3).reshape(-1), np.zeros(3)
    feet = np.array([-0.2, 0.13, -0.26, 0.2, 0.13, -0.26, -0.2, -0.13, -0.26, 0.2, -0.13, -0.26])
    x = np.array([0.0, 0.0, 0.26])
    stance = np.array([1, 0, 1, 1], np.uint8)
    pos = np.zeros(12); vel = np.zeros(12)
    import ctypes as C
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    def refs(phase):
        ph = np.full(4, phase)
        O.lib().oracle_swing_references(C.byref(kin), s1.ctypes.data_as(C.c_void_p), dp(I), dp(x), dp(z), dp(z), dp(z), dp(feet),
                                        stance.ctypes.data_as(C.POINTER(C.c_ubyte)), dp(ph), dp(pos), dp(vel))
        return pos[3:6].copy(), vel[3:6].copy()
    p0, v0 = refs(duty)            # swing starts: t = 0
    np.testing.assert_allclose(p0, s1["p_start"][0, 3:6], atol=1e-12); np.testing.assert_allclose(v0, 0.0, atol=1e-10)
    pm, _ = refs(duty + 0.5 * (1 - duty))
    assert abs(pm[2] - 0.08) < 1e-12  # apex = gait/height at the middle of the swing
    pf, vf = refs(1.0)             # t = 1
    np.testing.assert_allclose(pf, s1["p_final"][0, 3:6], atol=1e-12); np.testing.assert_allclose(vf, 0.0, atol=1e-9)


def test_batch_kkt_certificate_accepts_the_oracle_and_rejects_perturbations():
    """tests/kkt_batch.py (the vectorised certificate the -m gpu tests run over whole batches) agrees with the NNLS
    certificate on the oracle's literal rows, accepts the oracle's solutions and rejects perturbed ones."""
    from oracle import c_oracle as O, numpy_restatement as R
    from quadruped_control_amd import workloads as W
    from tests.kkt_batch import assert_kkt, kkt_batch

    P = R.cheetah_params(0.6)
    b = W.config3(2048)
    grf, st, _ = O.control_batch(P, b, threads=4)
    assert (st == 0).all()
    assert assert_kkt(P, b, grf) < 1e-10
    bad = grf.copy()
    stance_rows = np.flatnonzero(b["stance"][:, 0] == 1)[:50]
    bad[stance_rows, 2] += 1e-3   # 1 mN off on RL's z force
    c = kkt_batch(P, b, bad)
    assert (np.maximum(c["stationarity"], c["primal"])[stance_rows] > 1e-7).all()
    swing_rows = np.flatnonzero(b["stance"][:, 1] == 0)[:5]
    bad = grf.copy(); bad[swing_rows, 3] = 1e-12
    assert kkt_batch(P, b, bad)["swing_nonzero"][swing_rows].all()


def test_inv_pinv_switch_has_numbers():
    """INTEGRATION.md, "choices of this build", row 2: where legJacobianInverse (kinematics.cpp:190-204)
    stops inverting.  arma::inv of a 3x3 is closed-form while epsilon <= |det| <= 1 / epsilon and goes to LAPACK below
    that, which "succeeds" with a 1 / sigma_3-sized inverse unless a pivot is exactly zero; only then arma::pinv answers.
    The oracle (and the device) invert down to max(epsilon, 64 epsilon (sum |l|)^3) and answer everything below with
    pinv.  This test walks a leg towards full stretch and puts numbers on both ends of the band in between:
      * nearly straight but reachable (|det| above the bound): the oracle inverts, like the reference - joint-velocity
        targets of 1e4 ... 1e8 rad/s whose torque commander_node.cpp:526 clamps to +-tau_max on every implementation;
      * out of reach (IK clamps d to 1: exact rank loss, |det| = rounding noise ~1e-18): the oracle returns pinv(J) v, a
        bounded torque; LAPACK on the same J (numpy.linalg.inv = getrf + getri, what arma::inv falls back to) either
        reports singularity or returns entries > 1e12, i.e. the reference's answer there is pinv or a full-scale clamped
        torque depending on rounding - the one place where this build's torque can differ from the reference's by 2 tau_max."""
    kin = O.default_kinematics()
    eps = np.finfo(float).eps
    hip = np.array(kin.hip).reshape(4, 3)
    links = np.array(kin.links).reshape(4, 3)
    vel = np.array([0.3, -0.2, 0.25])
    rows = []
    for leg in range(4):
        det_lo = max(eps, 64 * eps * np.abs(links[leg]).sum() ** 3)
        assert 1e-15 < det_lo < 3e-15  # (sum |l|)^3 = 0.139: the noise floor, not Armadillo's bare epsilon, is the bound for this robot
        for theta in (1e-3, 1e-5, 1e-7, 3e-8):  # knee angle off straight
            q = np.array([0.15 if leg < 2 else -0.15, 0.6, -theta])
            target = O.leg_fk(leg, q, kin)
            qr = O.leg_ik(leg, target, kin)
            J = O.leg_jacobian(leg, qr, kin)
            det = np.linalg.det(J)
            assert abs(det) > 100 * det_lo and qr[2] != 0.0  # reachable: far above the bound even one ulp of d from straight
            tau = O.swing_torque(leg, np.eye(3), np.zeros(3), target, vel, qr, np.zeros(3), kin)  # (measured state = IK solution: only kd qd remains)
            want = np.array(kin.jc_kd) * np.linalg.solve(J, vel)
            np.testing.assert_allclose(tau, want, rtol=1e-5, atol=1e-9)
            rows.append((leg, theta, abs(det), np.abs(tau).max()))
            assert np.abs(tau).max() > 20.0 * (1e-4 / theta)  # far beyond tau_max: clamped the same way everywhere
        # out of reach: exact rank loss
        q = np.array([0.15 if leg < 2 else -0.15, 0.6, 0.0])
        target = hip[leg] + (O.leg_fk(leg, q, kin) - hip[leg]) * 1.3
        qr = O.leg_ik(leg, target, kin)
        J = O.leg_jacobian(leg, qr, kin)
        assert qr[2] == 0.0 and abs(np.linalg.det(J)) < det_lo / 50
        tau = O.swing_torque(leg, np.eye(3), np.zeros(3), target, vel, qr, np.zeros(3), kin)
        want = np.array(kin.jc_kd) * (np.linalg.pinv(J, rcond=3 * eps) @ vel)
        np.testing.assert_allclose(tau, want, atol=1e-9)
        assert np.abs(tau).max() < 20.0  # bounded: inside the torque limits for this velocity
        try:
            lapack = np.abs(np.linalg.inv(J)).max()  # what arma::inv's LAPACK fallback would hand back
        except np.linalg.LinAlgError:
            lapack = np.inf  # exact zero pivot: the reference reaches pinv as well
        assert lapack > 1e12
        rows.append((leg, 0.0, abs(np.linalg.det(J)), np.abs(tau).max()))
    # the band between the bound and exact rank loss cannot be entered through IK at all: the knee cosine d takes no value
    # between 1 - 2^-53 (|det| ~ 3e-10) and 1
    assert min(r[2] for r in rows if r[1] > 0) > 1e-11


def test_pinv3_converges_on_an_exactly_rank_deficient_jacobian():
    """Found by the round-4 tick fuzz (batch 519 of tests/stress_fuzz_tick.py): on this stretched-leg Jacobian of a random
    kinematic model (singular values 0.54, 0.46, 1e-17) the one-sided Jacobi sweeps of oracle_pinv3 never settled - the
    orthogonality test against a column of norm 1e-17 is rounding noise - so the oracle fell through to J^T
    (kinematics.cpp:198) where Armadillo's pinv (and the device) return the rank-2 pseudo-inverse; on two more (batches 3075,
    3888) two LARGE columns at the noise floor kept swapping the sign of their inner product.  Columns below the truncation
    tolerance now count as converged and the orthogonality test allows the inner product its own rounding error (4 epsilon)."""
    found = (  # batches 519 (stretched leg), 3075 (lateral clamp, kinematics.cpp:137-140: two parallel columns), 3888 (stretched leg)
        ('0x0.0p+0', '-0x1.d85eb9505ce1bp-2', '-0x1.bff7626ccb7bep-3', '0x1.f1e42b8af94a2p-3', '0x1.c9db5dd657bb5p-4',
         '0x1.b233f87bcccb7p-5', '-0x1.97de8d5c16c7dp-2', '0x1.813938b676175p-4', '0x1.6d52709bd6727p-5'),
        ('0x0.0p+0', '0x1.0000000000000p-55', '0x1.435907d076f24p-3', '-0x1.7ca32d7dde780p-8', '-0x1.9baaf35b4afccp-8',
         '-0x1.7ebf12851ea29p-8', '-0x1.7a64289092931p-4', '-0x1.993d0df968237p-4', '-0x1.7c7cddfcebd82p-4'),
        ('0x0.0p+0', '-0x1.abbd81fa189aep-3', '-0x1.4935e4b33f64fp-4', '0x1.83f9d0f42d538p-4', '-0x1.249903dc998e7p-2',
         '-0x1.c2655e762894ep-4', '-0x1.8d3b08bec9c99p-3', '-0x1.f68e836a5d925p-3', '-0x1.82cb03a72ccc1p-4'))
    for hexes in found:
        J = np.array([float.fromhex(h) for h in hexes]).reshape(3, 3)
        sv = np.linalg.svd(J, compute_uv=False)
        assert sv[2] < 1e-16 < 0.1 < sv[1]
        Jp, ok = O.pinv3(J)
        assert ok
        ref = np.linalg.pinv(J, rcond=3 * np.finfo(float).eps)
        np.testing.assert_allclose(Jp, ref, atol=1e-12 * np.abs(ref).max())
    # ... and over random exactly singular matrices (two parallel columns, a zero column, rank 1)
    rng = np.random.default_rng(3)
    for trial in range(20000):
        A = rng.normal(size=(3, 3)) * 10.0 ** rng.uniform(-3, 1)
        kind = trial % 4
        if kind == 0: A[:, 2] = A[:, 1] * rng.normal()
        elif kind == 1: A[:, rng.integers(3)] = 0.0
        elif kind == 2: A = np.outer(A[:, 0], A[0])
        else: A[2] = A[0] * rng.normal() + A[1] * rng.normal()
        Ap, ok = O.pinv3(A)
        assert ok, (trial, A)
        ref = np.linalg.pinv(A, rcond=3 * np.finfo(float).eps)
        np.testing.assert_allclose(Ap, ref, atol=1e-10 * max(1.0, np.abs(ref).max()))


def test_c_oracle_against_the_small_w_golden_vectors():
    """tests/golden/small_w_golden.json (numpy / NNLS restatement, w ~ 1e-7 - a hundredth of the reference's weight): the C oracle's primal
    active-set solver is within 1e-6 of it on every vector (its own multiplier threshold leaves it 4.8e-7 away on campaign 555's trial 138)."""
    import json

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "small_w_golden.json")))
    assert len(gold["groups"]) == 4 and sum(len(g["cases"]) for g in gold["groups"]) == 20
    fields = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "feet")
    for grp in gold["groups"]:
        import quadruped_control_amd as q

        P = q.cheetah_params(0.6)
        for k, v in grp["params"].items():
            P[k] = np.array(v, dtype=np.float64) if isinstance(v, list) else v
        assert P["W"][0][0] < 2.1e-7
        b = {k: np.array([c[k] for c in grp["cases"]], dtype=np.float64) for k in fields}
        b["stance"] = np.array([c["stance"] for c in grp["cases"]], dtype=np.uint8)
        ref, st, _ = O.control_batch(P, b, threads=2)
        exp = np.array([c["grf_body"] for c in grp["cases"]])
        assert (st == 0).all()
        assert np.max(np.abs(ref - exp) / np.maximum(1.0, np.abs(exp).max(axis=1, keepdims=True))) < 1e-6, grp["trial"]


def test_pinv_band_takes_its_rank_from_the_device_rule():
    """inside the band where this build answers legJacobianInverse with the pseudo-inverse (|det| below
    max(epsilon, 64 epsilon (sum |l|)^3)) the DEVICE's pinv3_apply never takes a third pivot and drops a second one below 1e-9
    of the first, while Armadillo's tolerance (3 sigma_max epsilon) would keep sigma_3 down to ~1e-16 sigma_1: for |det|
    between ~1e-17 and the band's upper end the two are a rank-2 and a rank-3 pseudo-inverse.  oracle_pinv3_band restates the
    device's rule (rank by complete pivoting, values from the SVD) so that the difference has numbers; oracle_swing_torque itself
    applies arma::pinv's rule (oracle_pinv3) like the reference and COUNTS the legs on which the two rules disagree
    (test_swing_torque_counts_pinv_rule_disagreements).  Planted matrices:"""
    eps = np.finfo(float).eps
    rng = np.random.default_rng(11)
    lo = max(eps, 64 * eps * 0.518 ** 3)

    def with_sv(s):
        U, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        V, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        return U @ np.diag(s) @ V.T

    for s3 in (6e-14, 3e-14, 1e-15, 1e-17, 0.0):  # |det| = 0.03 s3: just below the band's upper end ... exact rank loss
        J = with_sv([0.3, 0.1, s3])
        assert abs(np.linalg.det(J)) < lo
        Jb, rank = O.pinv3_band(J)
        assert rank == 2
        np.testing.assert_allclose(Jb, np.linalg.pinv(J, rcond=0.1), atol=1e-10)  # numpy told to keep two singular values
        assert np.abs(Jb).max() < 20.0
        Ja, ok = O.pinv3(J)  # Armadillo's tolerance alone: a 1 / s3-sized gain as long as s3 > 3 eps sigma_1
        assert ok and ((np.abs(Ja).max() > 0.1 / s3) if s3 >= 1e-15 else np.allclose(Ja, Jb, atol=1e-9))
    # rank 1 (the lateral clamp on top of a stretched knee): second pivot below 1e-9 of the first
    for s2 in (1e-11, 1e-14, 0.0):
        J = with_sv([0.4, s2, 0.0])
        Jb, rank = O.pinv3_band(J)
        assert rank == 1
        np.testing.assert_allclose(Jb, np.linalg.pinv(J, rcond=0.1), atol=1e-10)
    # on the exactly rank-deficient Jacobians IK produces the two rules agree: nothing changes for reachable inputs
    for trial in range(3000):
        A = rng.normal(size=(3, 3)) * 10.0 ** rng.uniform(-2, 0)
        if trial % 2: A[:, 2] = A[:, 1] * rng.normal()
        else: A = np.outer(A[:, 0], A[0])
        Jb, rank = O.pinv3_band(A)
        Ja, ok = O.pinv3(A)
        assert ok and rank == (2 if trial % 2 else 1)
        np.testing.assert_allclose(Jb, Ja, atol=1e-9 * max(1.0, np.abs(Ja).max()))


def test_swing_torque_counts_pinv_rule_disagreements():
    """the checker restates the reference (arma::pinv's tolerance), the device keeps its own rank rule, and every swing
    leg on which the two would keep a different number of singular values is counted - so "the window is unreachable" is an
    assertion the GPU tests make, not an assumption.  Here: out-of-reach and reachable swing references over random postures
    leave the counter at 0 (IK produces exact rank loss or a regular J); a kinematic model engineered to put J inside the
    window makes it fire."""
    kin = O.default_kinematics()
    rng = np.random.default_rng(5)
    O.pinv_rule_disagreements(reset=True)
    R = np.eye(3)
    for trial in range(4000):
        leg = trial % 4
        far = trial % 3 == 0
        pos = rng.normal(size=3) * (3.0 if far else 0.2) + np.array([0.0, 0.0, -0.25])
        O.swing_torque(leg, R, np.zeros(3), pos, rng.normal(size=3), rng.uniform(-1, 1, 3), rng.normal(size=3), kin)
    assert O.pinv_rule_disagreements() == 0
    # the counter is alive: a hip offset of 1e-13 m and a far target 1e-10 m off the hip's roll axis - the stretched leg lies along that
    # axis, the roll column of J is ~1e-10 of the pitch column: Armadillo's 3 sigma_max epsilon keeps it (rank 2), the device's
    # "second pivot above 1e-9 of the first" does not (rank 1)
    import copy

    k2 = copy.deepcopy(kin)
    k2.links[0] = 1e-13
    for trial in range(200):
        pos = np.array(k2.hip[0:3]) + np.array([5.0 if trial % 2 else -5.0, rng.normal() * 1e-10, rng.normal() * 1e-10])
        O.swing_torque(0, R, np.zeros(3), pos, rng.normal(size=3), rng.uniform(-1, 1, 3), rng.normal(size=3), k2)
    hits = O.pinv_rule_disagreements(reset=True)
    assert hits > 150 and O.pinv_rule_disagreements() == 0


def test_reference_inside_the_inner_reach_limit_gives_nan_torques():
    """legInverseKinematics clamps the knee cosine from above only (d > 1 -> 1, kinematics.cpp:131-134): a swing reference closer
    to the hip than | |l2| - |l3| | has d < -1, q3 = atan2(-sqrt(1 - d^2), d) = NaN, q2 = NaN, the Jacobian at q_ref is NaN, and
    Armadillo's closed-form inverse (cofactors over a NaN determinant) makes all three joint-velocity targets - hence all three
    torques of the leg - NaN, which commander_node.cpp:526's two-compare clamp leaves NaN."""
    kin = O.default_kinematics()
    hip = np.array(kin.hip).reshape(4, 3)
    links = np.abs(np.array(kin.links).reshape(4, 3))
    for leg in range(4):
        inner = abs(links[leg, 1] - links[leg, 2])  # 0.019 m
        # a point at distance sqrt(l1^2 + (0.5 inner)^2) from the hip: in the leg's plane only half the inner limit away
        target = hip[leg] + np.array([0.5 * inner, links[leg, 0] * (1 if leg < 2 else -1), 0.0])
        qr = O.leg_ik(leg, target, kin)
        assert np.isfinite(qr[0]) and np.isnan(qr[1]) and np.isnan(qr[2])
        tau = O.swing_torque(leg, np.eye(3), np.zeros(3), target, np.array([0.1, 0.2, -0.1]), np.zeros(3), np.zeros(3), kin)
        assert np.isnan(tau).all(), tau


def test_gait_clock_wrap_is_fmod_bit_for_bit():
    """The device wraps its gait phases with copysign(v - trunc(v), v) instead of calling fmod(v, 1.0) (qc_balance.hip,
    assemble_from_state; GaitScheduler::update, gait.cpp:113-123 uses fmod): the two are the same function bit for bit - the
    integer part of a double subtracts exactly and fmod's result carries its argument's sign - on phases, negative steps,
    exact integers (fmod(-1.0, 1) = -0.0), neighbours of integers, subnormals, huge values, infinities and NaN."""
    rng = np.random.default_rng(12)
    parts = [rng.uniform(0, 2, 400000), rng.uniform(-3, 4, 400000), rng.uniform(0, 1000, 100000),
             np.arange(-5, 6, dtype=np.float64), np.nextafter(np.arange(-5, 6, dtype=np.float64), 10.0), np.nextafter(np.arange(-5, 6, dtype=np.float64), -10.0),
             rng.integers(0, 2**63, 400000, dtype=np.uint64).view(np.float64), (rng.integers(0, 2**63, 400000, dtype=np.uint64) | np.uint64(1 << 63)).view(np.float64),
             np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1 - 2.0**-53, 2.0**52 + 0.5, 2.0**53, 1e308])]
    v = np.concatenate(parts)
    with np.errstate(all="ignore"):
        a = np.fmod(v, 1.0)
        b = np.copysign(v - np.trunc(v), v)
    same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
    assert same.all(), v[~same][:10]


def test_tick_restatement_matches_the_c_oracle_over_a_gait():
    """The tick AROUND the QP - gait clock, contact rule, foothold planner, sextic trajectories, IK, Jacobian inverse, joint PD,
    J^T, merge and clamp (commander_node.cpp:383-531) - restated twice: oracle/balance_oracle.c (flat arrays, per-trajectory Gaussian
    solve, Jacobi SVD) and oracle/tick_restatement.py (the reference's own maps and classes, numpy's solve / SVD, NNLS for the QP).
    Only FK and the Jacobian are pinned by reference-held numbers (the notebook); for everything else the two restatements hold
    each other: same contact states and trajectories tick by tick, footholds to 1e-12, torques to 1e-6 of tau_max, over a trot and a
    walk with jittered time steps, stretched legs (out-of-reach references: the pinv branch) included."""
    from oracle import tick_restatement as T

    kin = O.default_kinematics()
    # the constants each side took from the reference agree
    assert np.allclose(np.array(kin.hip).reshape(4, 3), [T.LINK_MAP[l][0] for l in T.LEGS]) and np.allclose(np.array(kin.links).reshape(4, 3), [T.LINK_MAP[l][1] for l in T.LEGS])
    assert np.allclose(np.array(kin.planner_hip).reshape(4, 3), [T.HIP_MAP[l] for l in T.LEGS]) and kin.planner_k == T.PLANNER_K
    assert (kin.t_swing, kin.t_stance, kin.swing_height, kin.tau_min, kin.tau_max) == (0.18, 0.8, 0.08, -20.0, 20.0)
    P = R.cheetah_params(0.6)
    rng = np.random.default_rng(21)
    n, ticks = 24, 90
    saw = dict(edges=0, pinv=0, swing_torque=0)
    for offs, (t_sw, t_st) in (((0.0, 0.5, 0.5, 0.0), (0.18, 0.8)), ((0.0, 0.25, 0.5, 0.75), (0.3, 0.3))):
        kin.t_swing, kin.t_stance = t_sw, t_st
        base = W.with_swing_references(W.with_joint_angles(W.config3(n, seed=int(rng.integers(1, 2**31)))))
        base = {k: v for k, v in base.items() if k not in ("stance", "swing_pos", "swing_vel")}
        base["joint_q"][::5] += rng.uniform(-0.6, 0.6, (len(base["joint_q"][::5]), 12))  # some far-from-nominal postures
        base["xdot"][1::4, :2] *= 8.0  # fast robots: the Raibert + LIP foothold lands out of reach, IK clamps d = 1, the Jacobian is singular
        phases = np.ascontiguousarray(np.fmod(np.array(offs)[None] + rng.uniform(0, 1, (n, 1)), 1.0))
        states = O.new_swing_states(n)
        robots = [T.Commander(P, phases[i], t_swing=t_sw, t_stance=t_st) for i in range(n)]
        for tick in range(ticks):
            dt = np.ascontiguousarray(rng.uniform(0.0, 0.02, n))
            b = dict(base)
            b["x"] = np.ascontiguousarray(base["x"] + 0.003 * tick * base["xdot"])  # drift: footholds and out-of-reach references change
            O.gait_update(phases, dt, kin=kin)
            prev = states["leg_state"].copy()
            ref = O.tick_planned_batch(P, dict(b, gait_phase=phases), states, kin=kin)
            saw["edges"] += int(((prev == 1) & (states["leg_state"] == 0)).sum())
            for i in range(n):
                tau, grf, status, st = robots[i].tick(b["Rwb"][i], b["Rwb_d"][i], b["x"][i], b["xdot"][i], b["w"][i], b["x_d"][i], b["xdot_d"][i], b["w_d"][i],
                                                      b["joint_q"][i], b["joint_qdot"][i], dt=dt[i])
                assert np.array_equal(robots[i].gait.phases, phases[i])  # the same clock, bit for bit
                assert status == ref["status"][i] == 0
                assert st["leg_state"] == list(states["leg_state"][i]) and st["has_traj"] == list(states["has_traj"][i]), (tick, i)
                for leg, (p0, pf) in st["bounds"].items():
                    k = T.LEGS.index(leg)
                    for mine, theirs in ((p0, states["p_start"][i, 3 * k:3 * k + 3]), (pf, states["p_final"][i, 3 * k:3 * k + 3])):
                        assert np.array_equal(np.isnan(mine), np.isnan(theirs)) and np.nanmax(np.abs(mine - theirs), initial=0.0) < 1e-12
                assert np.abs(grf - ref["grf_body"][i]).max() < 1e-6 * max(1.0, np.abs(grf).max())
                assert np.array_equal(np.isnan(tau), np.isnan(ref["joint_tau"][i]))
                assert np.nanmax(np.abs(tau - ref["joint_tau"][i]), initial=0.0) < 1e-6 * 20.0, (tick, i, tau, ref["joint_tau"][i])
                sw = [k for k in range(4) if st["leg_state"][k] == 0]
                saw["swing_torque"] += len(sw)
                for k in sw:  # how often the stretched-leg branch answered
                    pos, vel = robots[i].trajectories.reference_state(T.LEGS[k], phases[i, k])
                    qr = T.leg_inverse_kinematics(T.LEGS[k], b["Rwb"][i].reshape(3, 3).T @ pos - b["x"][i])
                    saw["pinv"] += int(qr[2] == 0.0)
    assert saw["edges"] > 100 and saw["swing_torque"] > 1000 and saw["pinv"] > 10, saw


def _tick_golden():
    import json

    return json.load(open(os.path.join(ROOT, "tests", "golden", "tick_golden.json")))


def test_c_oracle_matches_tick_golden():
    """tests/golden/tick_golden.json (written by the numpy restatement of the complete tick, tests/golden/make_tick_golden.py): the C
    oracle reproduces every tick - clock bit for bit, contact states and trajectories, forces and torques to 1e-6."""
    gold = _tick_golden()
    assert gold["leg_order"] == ["RL", "FL", "RR", "FR"] and len(gold["cases"]) == 2
    for case in gold["cases"]:
        P = R.cheetah_params(case["mu"])
        kin = O.default_kinematics()
        kin.t_swing, kin.t_stance = case["t_swing"], case["t_stance"]
        base = {k: np.ascontiguousarray(np.array(v, dtype=np.float64)) for k, v in case["inputs"].items()}
        n = base["Rwb"].shape[0]
        phases = np.ascontiguousarray(np.array(case["phase0"]))
        states = O.new_swing_states(n)
        for t in case["ticks"]:
            O.gait_update(phases, np.ascontiguousarray(np.array(t["dt"])), kin=kin)
            assert np.array_equal(phases, np.array(t["phase"]))
            r = O.tick_planned_batch(P, dict(base, x=np.ascontiguousarray(np.array(t["x"])), gait_phase=phases), states, kin=kin)
            assert (r["status"] == 0).all()
            assert np.array_equal(states["leg_state"], np.array(t["leg_state"])) and np.array_equal(states["has_traj"], np.array(t["has_traj"]))
            g = np.array(t["grf_body"])
            assert np.abs(r["grf_body"] - g).max() < 1e-6 * max(1.0, np.abs(g).max())
            assert np.abs(r["joint_tau"] - np.array(t["joint_tau"])).max() < 1e-6 * 20.0
