"""Development tool: contact-class order (pre-pass kernel + indirection) on / off, cold config-3 distribution vs batch size,
seeds, and an all-stance batch (the pre-pass is pure cost there).  usage: python tools/order_scan.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W, workloads_device as WD
P = q.cheetah_params(0.6)
def timeit(ctl, b, reps=20):
    launch, out = ctl.plan_batch(b, want_iterations=True)
    for _ in range(3): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    return e0.elapsed_time(e1) / reps * 1e3, out["grf_body"].clone()
SEEDS = [0x5EED0005 + 0x1000 * k for k in range(4)]
for n in (65536, 98304, 131072, 196608, 262144, 524288, 1048576, 2097152):
    rows = {"off": [], "on": []}
    err = 0.0
    for seed in SEEDS:
        b = WD.config3(n, seed=seed)
        t0, g0 = timeit(q.BalanceController.from_params(P).set_tuning(order=0), b, 20 if n <= 262144 else 6)
        ctl = q.BalanceController.from_params(P).set_tuning(order=1)
        assert ctl.query_launch(n, stance=True)["contact_order"] == 1
        t1, g1 = timeit(ctl, b, 20 if n <= 262144 else 6)
        rows["off"].append(t0); rows["on"].append(t1)
        err = max(err, float((g1 - g0).abs().max()))
    print("cold n=%8d: off %7.1f us | on %7.1f us (%+.1f %%)  max |dGRF| %.1e" % (n, np.mean(rows["off"]), np.mean(rows["on"]), 100 * (np.mean(rows["on"]) / np.mean(rows["off"]) - 1), err), flush=True)
b = q.to_device(W.config4(262144)[0])
print("config-4 tick 0 (all four feet in stance, contact bytes given): off %.1f us | on %.1f us" %
      (timeit(q.BalanceController.from_params(P).set_tuning(order=0), b)[0], timeit(q.BalanceController.from_params(P).set_tuning(order=1), b)[0]))
