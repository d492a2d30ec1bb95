"""Re-run chosen trials of tests/stress_fuzz.py and arbitrate GPU vs C oracle with the NNLS/KKT restatement (development tool)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
from oracle import c_oracle as O
from oracle import numpy_restatement as R
want = set(int(a) for a in sys.argv[1:]) or {48, 95, 96}
n = int(os.environ.get("FUZZ_N", "2048"))
rng = np.random.default_rng(int(os.environ.get("QC_FUZZ_SEED", 77)))  # the campaign whose trial is being dug out
FIELDS = ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d", "feet")
from tests import stress_fuzz
for trial in range(max(want) + 1):
    P, seed = stress_fuzz.draw_trial(rng, trial)
    if trial not in want:
        continue
    b0 = W.config4(n, seed=seed)[0] if trial % 2 else W.config3(n, seed=seed)
    ctl = q.BalanceController.from_params(P)
    if os.environ.get("FUZZ_TUNE"):  # e.g. FUZZ_TUNE=force_dense=1,group=4: how another formulation / width does on the same trial
        ctl.set_tuning(**{kv.split("=")[0]: float(kv.split("=")[1]) for kv in os.environ["FUZZ_TUNE"].split(",")})
    o = ctl.control_batch_host(b0, want_active_set=True, want_iterations=True)
    ref, st, it = O.control_batch(P, b0, threads=16)
    scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
    e = (np.abs(o["grf_body"] - ref) / scale).max(axis=1)
    i = int(np.argmax(e))
    print("trial %d: worst robot %d err %.3e  w=%.2e mu=%.3f fz=[%g,%g] stance %s gpu iters %d oracle iters %d" %
          (trial, i, e[i], P["W"][0, 0], P["mu"], P["fzmin"], P["fzmax"], b0["stance"][i], o["iterations"][i], it[i]))
    qp = O.assemble(P, *[b0[k][i] for k in FIELDS], b0["stance"][i])
    Rm = b0["Rwb"][i].reshape(3, 3)
    fw_gpu = -(o["grf_body"][i].reshape(4, 3) @ Rm.T).reshape(-1)
    fw_ora = -(ref[i].reshape(4, 3) @ Rm.T).reshape(-1)
    lb, ub = qp["lb"], qp["ub"]
    f_ldp = R.solve_qp_ldp(qp["H"], qp["g"], qp["C"], lb, ub)
    phi = lambda f: 0.5 * f @ qp["H"] @ f + qp["g"] @ f
    for nm, f in (("gpu", fw_gpu), ("oracle", fw_ora), ("ldp", f_ldp)):
        c = R.kkt_certificate(qp["H"], qp["g"], qp["C"], lb, ub, f)
        print("   %-6s phi %.12e  kkt stationarity %.2e primal %.2e  |f-ldp| %.2e" % (nm, phi(f), c["stationarity"], c["primal"], np.abs(f - f_ldp).max()))
    print("   gpu   ", np.round(fw_gpu, 6)); print("   oracle", np.round(fw_ora, 6)); print("   active word %08x" % o["active_set"][i])
