"""Generates tests/golden/tick_golden.json: the complete controller tick (commander_node.cpp:383-531) over two gaits.

The vectors come from oracle/tick_restatement.py (numpy; the reference's classes restated one robot at a time, the QP by NNLS) and
every tick of every robot is cross-checked against the independent C oracle (oracle/balance_oracle.c) before it is written - the
reference itself has no Python and no fixtures for this path, so nothing of /root/reference is imported or copied.

Layout: cases[g] = {t_swing, t_stance, inputs per robot (static: Rwb ... joint_qdot, x0 and the drift xdot), phase0 [n][4],
ticks[t] = {dt [n], x [n][3] (= x0 + drift), phase [n][4] (after the clock), leg_state [n][4], has_traj [n][4], grf_body [n][12],
joint_tau [n][12]}}.

Run:  python tests/golden/make_tick_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import c_oracle as O  # noqa: E402
from oracle import numpy_restatement as R  # noqa: E402
from oracle import tick_restatement as T  # noqa: E402
from quadruped_control_amd import workloads as W  # noqa: E402

STATIC = ("Rwb", "Rwb_d", "xdot", "w", "x_d", "xdot_d", "w_d", "joint_q", "joint_qdot")


def main(n=10, ticks=30):
    P = R.cheetah_params(0.6)
    rng = np.random.default_rng(20260929)
    cases = []
    for offs, (t_sw, t_st), dt_nom in (((0.0, 0.5, 0.5, 0.0), (0.18, 0.8), 1.0 / 60.0), ((0.0, 0.25, 0.5, 0.75), (0.3, 0.3), 1.0 / 50.0)):
        kin = O.default_kinematics()
        kin.t_swing, kin.t_stance = t_sw, t_st
        base = W.with_swing_references(W.with_joint_angles(W.config3(n, seed=int(rng.integers(1, 2**31)))))
        base = {k: v for k, v in base.items() if k not in ("stance", "swing_pos", "swing_vel")}
        base["joint_q"][::4] += rng.uniform(-0.5, 0.5, (len(base["joint_q"][::4]), 12))
        base["xdot"][1::4, :2] *= 8.0  # fast robots: out-of-reach footholds, the pinv branch of legJacobianInverse
        phase0 = np.ascontiguousarray(np.fmod(np.array(offs)[None] + rng.uniform(0, 1, (n, 1)), 1.0))
        phases = phase0.copy()
        states = O.new_swing_states(n)
        robots = [T.Commander(P, phase0[i], t_swing=t_sw, t_stance=t_st) for i in range(n)]
        out_ticks = []
        for tick in range(ticks):
            dt = np.ascontiguousarray(dt_nom * rng.uniform(0.5, 1.5, n))
            x = np.ascontiguousarray(base["x"] + 0.004 * tick * base["xdot"])
            b = dict(base, x=x)
            O.gait_update(phases, dt, kin=kin)
            ref = O.tick_planned_batch(P, dict(b, gait_phase=phases), states, kin=kin)
            rec = dict(dt=dt.tolist(), x=x.tolist(), phase=[], leg_state=[], has_traj=[], grf_body=[], joint_tau=[])
            for i in range(n):
                tau, grf, status, st = robots[i].tick(b["Rwb"][i], b["Rwb_d"][i], x[i], b["xdot"][i], b["w"][i], b["x_d"][i], b["xdot_d"][i], b["w_d"][i],
                                                      b["joint_q"][i], b["joint_qdot"][i], dt=dt[i])
                # the independent C oracle agrees on this robot's tick, or nothing is written
                assert status == 0 == ref["status"][i] and np.array_equal(robots[i].gait.phases, phases[i])
                assert st["leg_state"] == list(states["leg_state"][i]) and st["has_traj"] == list(states["has_traj"][i])
                assert np.isfinite(tau).all() and np.abs(tau - ref["joint_tau"][i]).max() < 1e-6 * 20.0
                assert np.abs(grf - ref["grf_body"][i]).max() < 1e-6 * max(1.0, np.abs(grf).max())
                rec["phase"].append(robots[i].gait.phases.tolist()); rec["leg_state"].append(st["leg_state"]); rec["has_traj"].append(st["has_traj"])
                rec["grf_body"].append(grf.tolist()); rec["joint_tau"].append(tau.tolist())
            out_ticks.append(rec)
        cases.append(dict(t_swing=t_sw, t_stance=t_st, mu=0.6, phase0=phase0.tolist(), x0=base["x"].tolist(),
                          inputs={k: base[k].tolist() for k in STATIC}, ticks=out_ticks))
        edges = sum(int(((np.array(out_ticks[t - 1]["leg_state"]) == 1) & (np.array(out_ticks[t]["leg_state"]) == 0)).sum()) for t in range(1, ticks))
        print(f"gait {t_sw}/{t_st}: {n} robots x {ticks} ticks, {edges} stance->swing edges, max |tau| {max(np.abs(t['joint_tau']).max() for t in out_ticks):.1f}")
    path = os.path.join(ROOT, "tests", "golden", "tick_golden.json")
    json.dump({"generator": "tests/golden/make_tick_golden.py (oracle/tick_restatement.py, cross-checked against oracle/balance_oracle.c)",
               "leg_order": list(T.LEGS), "cases": cases}, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
