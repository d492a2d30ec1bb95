"""Development tool: the dense 12x12 form (general SPD W) on configs 2 / 3, four lanes per robot vs one.
usage: python tools/dense_scan.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
def timeit(ctl, b, reps=30):
    launch, out = ctl.plan_batch(b, want_iterations=True)
    for _ in range(3): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    return e0.elapsed_time(e1) / reps * 1e3, out
for name, b, n in (("config2 4096", q.to_device(W.config2(4096)), 4096), ("config3 8192", q.to_device(W.config3(8192)), 8192),
                   ("config3 65536", q.to_device(W.config3(65536)), 65536)):
    ref = None
    for tune in (dict(), dict(force_dense=1), dict(force_dense=1, group=4), dict(force_dense=1, group=1), dict(force_general=1)):
        ctl = q.BalanceController.from_params(P).set_tuning(**tune)
        info = ctl.query_launch(n)
        us, out = timeit(ctl, b)
        g = out["grf_body"]
        ref = g.clone() if ref is None else ref
        print("%-14s %-34s form=%d G=%d mode=%d : %7.1f us  %.3e QP/s  max iters %d  maxdiff vs uniform %.1e" %
              (name, tune, info["form"], info["lanes_per_robot"], info["mode"], us, n / us * 1e6, int(out["iterations"].max()), float((g - ref).abs().max())), flush=True)
