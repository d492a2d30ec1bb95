"""Generates tests/golden/balance_golden.json.

The reference has no Python implementation and no fixtures for this path, so
nothing of /root/reference is imported or copied: the vectors come from
oracle/numpy_restatement.py (assembly restated from the reference's C++,
QP solved by scipy's Lawson-Hanson NNLS through least-distance programming,
polished and KKT-certified), and every vector is cross-checked against the
independent C oracle (oracle/balance_oracle.c, primal active set on the
literal qpOASES data) before it is written.

Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import c_oracle as O  # noqa: E402
from oracle import numpy_restatement as R  # noqa: E402
from quadruped_control_amd import workloads as W  # noqa: E402

FIELDS = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "feet", "stance")


def kat_inputs():
    """Known-answer cases of SURVEY.md section 8c (closed forms in comments)."""
    base = W.config1()
    cases = []
    # KAT1: nominal stance, mu=0.8 -> each foot (0,0,-17.5353311617) body frame
    cases.append(("KAT1_nominal", dict(mu=0.8), {k: v.copy() for k, v in base.items()}))
    # KAT2: x_z = 0.36 -> b_z < 0 -> every foot clamps at fzmin = 10
    b = {k: v.copy() for k, v in base.items()}; b["x"][0, 2] = 0.36
    b["feet"] = W._pack(np.eye(3)[None], np.eye(3)[None], b["x"], b["xdot"], b["w"], b["x_d"], b["xdot_d"], b["w_d"],
                        np.concatenate([W.NOMINAL_FEET_XY, np.zeros((4, 1))], 1)[None], np.ones((1, 4)))["feet"]
    cases.append(("KAT2_fzmin_clamp", dict(mu=0.8), b))
    # KAT3: trot RL+FR stance -> 35.0705746471 N on RL and FR
    b = {k: v.copy() for k, v in base.items()}; b["stance"][0] = [1, 0, 0, 1]
    cases.append(("KAT3_trot_diag", dict(mu=0.8), b))
    # KAT4: fzmax = 15 clamp
    cases.append(("KAT4_fzmax_clamp", dict(mu=0.8, fzmax=15.0), {k: v.copy() for k, v in base.items()}))
    # KAT5: cone saturation, mu=0.6, x=(-0.1,0,0.26)
    b = {k: v.copy() for k, v in base.items()}; b["x"][0, 0] = -0.1
    b["feet"] = W._pack(np.eye(3)[None], np.eye(3)[None], b["x"], b["xdot"], b["w"], b["x_d"], b["xdot_d"], b["w_d"],
                        np.concatenate([W.NOMINAL_FEET_XY, np.zeros((4, 1))], 1)[None], np.ones((1, 4)))["feet"]
    cases.append(("KAT5_cone_saturation", dict(mu=0.6), b))
    return cases


def solve_case(P, b, i):
    args = [b[k][i] for k in FIELDS[:-1]]
    out, fmap, fw, qp = R.control(P, args[0].reshape(3, 3), args[1].reshape(3, 3), *args[2:], b["stance"][i])
    cert = R.kkt_certificate(qp["H"], qp["g"], qp["C"], qp["lb"], qp["ub"], fw)
    assert cert["primal"] < 1e-8 and cert["stationarity"] < 1e-9, cert
    sub = {k: b[k][i:i + 1] for k in FIELDS}
    grf_c, st_c, _ = O.control_batch(P, sub)
    assert st_c[0] == 0
    scale = max(1.0, np.max(np.abs(out)))
    assert np.max(np.abs(grf_c[0] - out.reshape(-1))) / scale < 1e-8, "C oracle disagrees with numpy/NNLS"
    rec = {k: np.asarray(b[k][i]).reshape(-1).tolist() for k in FIELDS}
    rec["grf_body"] = out.reshape(-1).tolist()
    rec["f_world"] = fw.tolist()
    rec["n_active"] = cert["n_active"]
    return rec


def main():
    gold = {"leg_order": list(R.LEG_NAMES), "cases": []}
    for name, over, b in kat_inputs():
        P = R.cheetah_params(over.get("mu", 0.8))
        if "fzmax" in over:
            P["fzmax"] = over["fzmax"]
        rec = solve_case(P, b, 0)
        rec.update(name=name, mu=P["mu"], fzmin=P["fzmin"], fzmax=P["fzmax"])
        gold["cases"].append(rec)
    P = R.cheetah_params(0.6)
    for tag, batch in (("cfg2", W.config2(48)), ("cfg3", W.config3(80))):
        for i in range(batch["x"].shape[0]):
            rec = solve_case(P, batch, i)
            rec.update(name=f"{tag}_{i}", mu=P["mu"], fzmin=P["fzmin"], fzmax=P["fzmax"])
            gold["cases"].append(rec)
    path = os.path.join(os.path.dirname(__file__), "balance_golden.json")
    with open(path, "w") as f:
        json.dump(gold, f)
    print("wrote", path, len(gold["cases"]), "cases")
    for c in gold["cases"][:5]:
        print(c["name"], np.round(c["grf_body"], 6).tolist(), c["n_active"])


if __name__ == "__main__":
    main()
