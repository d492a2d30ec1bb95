"""Fuzz campaign (run() is what tests/test_gpu_fuzz.py calls with a time budget; as a script it runs the long version): the stateful complete tick (gait phases -> contact rule, foothold planner,
sextic swing trajectories, IK/PD, QP, J^T) over many ticks with random gait timing, planner gains, swing height and
velocity commands; the carried swing state and the torques must track the C oracle tick by tick.
usage: python tests/stress_fuzz_planner.py [runs=12] [robots=2048] [ticks=60]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
from oracle import c_oracle as O



def run_campaign(runs=12, n=2048, ticks=60, budget_s=None, min_runs=2):
    """Returns (worst torque error / tau_max, worst foothold error [m], ticks with a state mismatch, runs done)."""
    rng = np.random.default_rng(int(os.environ.get("QC_FUZZ_SEED", 31337)))  # QC_FUZZ_SEED: another campaign (the default is the one pytest runs)
    worst_tau = 0.0; worst_p = 0.0; state_mism = 0; t0 = time.time()
    for run in range(runs):
        if budget_s is not None and run >= min_runs and time.time() - t0 > budget_s: run -= 1; break
        P = q.cheetah_params(0.6)
        t_sw, t_st = float(rng.uniform(0.1, 0.4)), float(rng.uniform(0.2, 0.9))
        pk, sh = float(rng.uniform(0.0, 0.1)), float(rng.uniform(0.02, 0.15))
        offs = np.array(rng.choice([[0, .5, .5, 0], [0, .25, .5, .75], [0, 0, .5, .5], [0, .5, 0, .5]]), dtype=float)
        kin = O.default_kinematics(); kin.t_swing = t_sw; kin.t_stance = t_st; kin.planner_k = pk; kin.swing_height = sh
        ctl = q.BalanceController.from_params(P)
        ctl.set_kinematics(planner_k=pk, swing_height=sh); ctl.set_gait(t_sw, t_st)
        seed = int(rng.integers(1, 2**31))
        base = W.with_swing_references(W.with_joint_angles(W.config3(n, seed=seed)))
        base = {k: v for k, v in base.items() if k not in ("stance", "swing_pos", "swing_vel")}
        phi0 = rng.uniform(0, 1, n)
        dev_state, ref_state = q.new_swing_states(n), O.new_swing_states(n)
        dt = float(rng.choice([1 / 300.0, 1 / 100.0, 1 / 30.0]))
        for tick in range(ticks):
            b = dict(base)
            b["gait_phase"] = np.ascontiguousarray(np.fmod(offs[None] + phi0[:, None] + tick * dt / (t_sw + t_st), 1.0))
            b["x"] = np.ascontiguousarray(base["x"] + tick * dt * base["xdot"])     # drift so that footholds move
            o = ctl.control_batch_host(dict(b, swing_state=dev_state), want_torques=True)
            ref = O.tick_planned_batch(P, b, ref_state, kin=kin, threads=16)
            if not (np.array_equal(dev_state["leg_state"], ref_state["leg_state"]) and np.array_equal(dev_state["has_traj"], ref_state["has_traj"])):
                state_mism += 1
            m = ref_state["has_traj"].repeat(3, axis=1) == 1
            if m.any():
                worst_p = max(worst_p, float(np.abs(dev_state["p_start"][m] - ref_state["p_start"][m]).max()), float(np.abs(dev_state["p_final"][m] - ref_state["p_final"][m]).max()))
            ok = (o["status"] == 0) & (ref["status"] == 0)
            assert np.array_equal(o["status"], ref["status"])
            # (a torque that is NaN on one side only must not hide behind |a - nan| > tol == False: it counts as a mismatching tick)
            if not np.array_equal(np.isnan(o["joint_tau"]), np.isnan(ref["joint_tau"])):
                state_mism += 1
                i, j = np.argwhere(np.isnan(o["joint_tau"]) != np.isnan(ref["joint_tau"]))[0]
                print("run %d tick %d: NaN on one side only (robot %d joint %d gpu %r oracle %r)" % (run, tick, i, j, o["joint_tau"][i, j], ref["joint_tau"][i, j]))
            dd = np.abs(o["joint_tau"] - ref["joint_tau"])[ok]
            d = float(np.nanmax(dd)) / 20.0 if dd.size and not np.isnan(dd).all() else 0.0
            if d > 1e-6:
                i, j = np.unravel_index(np.nanargmax(np.abs(o["joint_tau"] - ref["joint_tau"])), o["joint_tau"].shape)
                print("run %d tick %d: torque err %.2e of tau_max (robot %d joint %d gpu %.6f oracle %.6f)" % (run, tick, d, i, j, o["joint_tau"][i, j], ref["joint_tau"][i, j]))
            worst_tau = max(worst_tau, d)
    print("%d runs x %d robots x %d ticks in %.0f s: worst torque err %.2e of tau_max, worst foothold err %.2e m, ticks with a state mismatch %d" %
          (run + 1, n, ticks, time.time() - t0, worst_tau, worst_p, state_mism))
    return worst_tau, worst_p, state_mism, run + 1


if __name__ == "__main__":
    run_campaign(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 2048, int(sys.argv[3]) if len(sys.argv) > 3 else 60)
    from oracle import c_oracle as _O

    print("swing legs on which arma::pinv's rank rule (oracle) and the device's would differ: %d" % _O.pinv_rule_disagreements())
