"""Fuzz campaign: the on-device gait clock under random gait periods and jittered time steps over hundreds of
ticks - GaitScheduler::update (gait.cpp:113-123: phase += dt / (t_swing + t_stance), fmod(., 1)) feeding the contact rule with its
1e-12 slack (gait.cpp:125-134), the foothold planner and the swing trajectories of the complete tick.  Every tick a share of the
robots gets a dt AIMED at the edges: leg 0's advanced phase lands on the duty edge +- {0, 5e-13, 2e-12}, on the wrap 1 -> 0 +- a few
ulps, or dt = 0 / a whole number of periods.  The phases the device carries must stay BIT-EQUAL to the oracle's clock, the
carried swing state equal, the torques within 1e-6 of tau_max, tick by tick.
run_campaign() is what tests/test_gpu_fuzz.py calls with a time budget; as a script it runs the long version.
usage: python tests/stress_fuzz_gait.py [runs=6] [robots=1024] [ticks=500]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
from oracle import c_oracle as O


def run_campaign(runs=6, n=1024, ticks=500, budget_s=None, min_runs=1):
    """Returns (worst torque error / tau_max, ticks with a clock / state / NaN-pattern mismatch, stance->swing edges seen, aimed
    edge hits [legs whose phase sits within 3e-12 of the duty edge or of the wrap], runs done)."""
    rng = np.random.default_rng(int(os.environ.get("QC_FUZZ_SEED", 424242)))
    worst_tau = 0.0; mism = 0; edges = 0; aimed_hits = 0; t0 = time.time(); done = 0
    for run in range(runs):
        if budget_s is not None and run >= min_runs and time.time() - t0 > budget_s: break
        P = q.cheetah_params(float(rng.choice([0.4, 0.6, 0.8])))
        t_sw, t_st = float(rng.uniform(0.05, 0.5)), float(rng.uniform(0.05, 1.0))
        if run % 3 == 2: t_sw, t_st = 0.3, 0.3  # commander_node.cpp:245-246 defaults: duty exactly 0.5
        T = t_sw + t_st
        duty = t_st / T
        kin = O.default_kinematics(); kin.t_swing = t_sw; kin.t_stance = t_st
        ctl = q.BalanceController.from_params(P)
        ctl.set_gait(t_sw, t_st)
        offs = np.array(rng.choice([[0, .5, .5, 0], [0, .25, .5, .75], [0, 0, .5, .5], [0, .5, 0, .5]]), dtype=float)
        base = W.with_swing_references(W.with_joint_angles(W.config3(n, seed=int(rng.integers(1, 2**31)))))
        base = {k: v for k, v in base.items() if k not in ("stance", "swing_pos", "swing_vel")}
        dev_phase = np.ascontiguousarray(np.fmod(offs[None] + rng.uniform(0, 1, (n, 1)), 1.0))
        ref_phase = dev_phase.copy()
        dev_state, ref_state = q.new_swing_states(n), O.new_swing_states(n)
        dt_nom = float(rng.choice([1 / 300.0, 1 / 100.0, 1 / 1000.0]))
        for tick in range(ticks):
            if budget_s is not None and run >= min_runs and time.time() - t0 > budget_s: break
            dt = dt_nom * rng.uniform(0.0, 2.0, n)  # jitter: 0 ... 2 nominal steps
            # aimed steps for a quarter of the robots: leg 0 lands on the duty edge / the wrap +- a few 1e-13, or does not move
            aim = rng.random(n) < 0.25
            kind = rng.integers(0, 5, n)
            delta = rng.choice([0.0, 5e-13, -5e-13, 2e-12, -2e-12, 1.2e-12, -0.8e-12], n)
            target = np.where(kind < 2, duty + delta, np.where(kind == 2, 1.0 + delta * 1e-4, np.where(kind == 3, 0.0, -1.0)))
            gap = np.where(target >= 0, np.mod(target - ref_phase[:, 0], 1.0), 0.0)
            gap = np.where(kind == 4, float(rng.integers(1, 4)), gap)  # whole periods: the phase must come back to itself (up to rounding)
            dt = np.where(aim, gap * T, dt)
            b = dict(base, gait_phase=dev_phase, gait_dt=np.ascontiguousarray(dt), swing_state=dev_state)
            b["x"] = np.ascontiguousarray(base["x"] + 0.002 * tick * base["xdot"])
            prev_state = ref_state["leg_state"].copy()
            o = ctl.control_batch_host(b, want_torques=True)
            O.gait_update(ref_phase, np.ascontiguousarray(dt), kin=kin)
            bad = not np.array_equal(dev_phase, ref_phase)
            rb = {k: v for k, v in b.items() if k not in ("gait_dt", "swing_state")}
            rb["gait_phase"] = ref_phase
            r = O.tick_planned_batch(P, rb, ref_state, kin=kin, threads=16)
            bad |= not (np.array_equal(dev_state["leg_state"], ref_state["leg_state"]) and np.array_equal(dev_state["has_traj"], ref_state["has_traj"]))
            bad |= not np.array_equal(o["status"], r["status"])
            bad |= not np.array_equal(np.isnan(o["joint_tau"]), np.isnan(r["joint_tau"]))
            if bad:
                mism += 1
                print("run %d tick %d (t_swing %.4f t_stance %.4f): clock / state / status / NaN mismatch; phase diffs %d, leg_state diffs %d" %
                      (run, tick, t_sw, t_st, int((dev_phase != ref_phase).sum()), int((dev_state["leg_state"] != ref_state["leg_state"]).sum())))
                dev_phase[:] = ref_phase  # keep going from a common state
                dev_state[:] = ref_state
            edges += int(((prev_state == 1) & (ref_state["leg_state"] == 0)).sum())
            aimed_hits += int((np.abs(ref_phase - duty) < 3e-12).sum() + (ref_phase < 3e-12).sum() + (ref_phase > 1 - 3e-12).sum())
            ok = (o["status"] == 0) & (r["status"] == 0)
            dd = np.abs(o["joint_tau"] - r["joint_tau"])[ok]
            d = float(np.nanmax(dd)) / 20.0 if dd.size and not np.isnan(dd).all() else 0.0
            if d > 1e-6:
                i, j = np.unravel_index(np.nanargmax(np.abs(o["joint_tau"] - r["joint_tau"])), o["joint_tau"].shape)
                print("run %d tick %d: torque err %.2e of tau_max (robot %d joint %d gpu %.6f oracle %.6f phase %r)" %
                      (run, tick, d, i, j, o["joint_tau"][i, j], r["joint_tau"][i, j], ref_phase[i]))
            worst_tau = max(worst_tau, d)
        done = run + 1
    print("%d runs x %d robots x <= %d ticks in %.0f s: worst torque err %.2e of tau_max, ticks with a clock / state mismatch %d, stance->swing edges %d, "
          "phases within 3e-12 of the duty edge or the wrap %d" % (done, n, ticks, time.time() - t0, worst_tau, mism, edges, aimed_hits))
    return worst_tau, mism, edges, aimed_hits, done


if __name__ == "__main__":
    run_campaign(int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 1024, int(sys.argv[3]) if len(sys.argv) > 3 else 500)
    from oracle import c_oracle as _O

    print("swing legs on which arma::pinv's rank rule (oracle) and the device's would differ: %d" % _O.pinv_rule_disagreements())
