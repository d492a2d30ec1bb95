"""Generates tests/golden/small_w_golden.json: robots of the parameter campaigns' hardest trials (weights W = w I with w ~ 1e-7, a
hundredth of the reference's 1e-5), with the forces of the numpy / NNLS restatement (oracle/numpy_restatement.py: least-distance
programming, polished and KKT-certified) - a pin of the small-w behaviour that does not go through the C oracle's active-set solver.

  * campaign QC_FUZZ_SEED=555, trial 138 (w = 1.07e-7): round 5 stopped 6.9e-5 relative from the minimiser on robot 168 - the
    acceptance threshold's reach tol |g| / (2w), closed by round 6's polish at acceptance;
  * campaign 20260929, trials 9010 / 19709 / 14087 (w = 1.7e-7 ... 2e-7): the 6x6 dual form's conditioning, 0.9e-5 ... 2.4e-5.

Nothing of /root/reference is read.  Run:  python tests/golden/make_small_w_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import c_oracle as O  # noqa: E402
from oracle import numpy_restatement as R  # noqa: E402
from tests import stress_fuzz  # noqa: E402

FIELDS = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "feet", "stance")
PICK = [(555, 138, [168, 0, 1, 2, 3, 500, 1000, 1990]), (20260929, 9010, [1772, 0, 1, 2]), (20260929, 19709, [208, 1719, 0, 1]), (20260929, 14087, [340, 885, 0, 1])]


def main():
    gold = {"leg_order": list(R.LEG_NAMES), "groups": []}
    for seed, trial, robots in PICK:
        P, b = stress_fuzz.trial_at(seed, trial)
        grp = {"campaign_seed": seed, "trial": trial,
               "params": {k: (np.asarray(v).tolist() if not np.isscalar(v) else float(v)) for k, v in P.items()}, "cases": []}
        ref, st, _ = O.control_batch(P, b, threads=8)
        for i in robots:
            args = [b[k][i] for k in FIELDS[:-1]]
            out, fmap, fw, qp = R.control(P, args[0].reshape(3, 3), args[1].reshape(3, 3), *args[2:], b["stance"][i])
            cert = R.kkt_certificate(qp["H"], qp["g"], qp["C"], qp["lb"], qp["ub"], fw)
            assert cert["primal"] < 1e-8 and cert["stationarity"] < 1e-9, cert
            scale = max(1.0, float(np.max(np.abs(out))))
            dc = float(np.max(np.abs(ref[i] - out.reshape(-1))) / scale)
            assert st[i] == 0 and dc < 2e-6, ("C oracle disagrees with numpy/NNLS", seed, trial, i, dc)
            rec = {k: np.asarray(b[k][i]).reshape(-1).tolist() for k in FIELDS}
            rec.update(robot=i, grf_body=out.reshape(-1).tolist(), n_active=cert["n_active"], c_oracle_vs_nnls=dc)
            grp["cases"].append(rec)
        gold["groups"].append(grp)
        print("campaign %d trial %d: w = %.3g, %d robots, C oracle vs NNLS <= %.1e" % (seed, trial, P["W"][0][0], len(robots), max(c["c_oracle_vs_nnls"] for c in grp["cases"])))
    path = os.path.join(os.path.dirname(__file__), "small_w_golden.json")
    with open(path, "w") as f:
        json.dump(gold, f)
    print("wrote", path)


if __name__ == "__main__":
    main()
