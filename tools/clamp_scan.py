"""Development tool: clamp steps a fresh robot takes before its first ratio test (qc_set_tuning "clamp_steps") - kernel time,
mean and maximum recalculations on configs 2-5.  usage: python tools/clamp_scan.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
def timeit(ctl, b, warm, reps=30):
    launch, out = ctl.plan_batch(b, warm=warm, want_iterations=True)
    for _ in range(5): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    it = out["iterations"].cpu().numpy()
    return e0.elapsed_time(e1) / reps * 1e3, it.mean(), int(it.max())
t0, t1 = W.config4(262144)
w4 = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
CASES = [("config2 4096 solo", q.to_device(W.config2(4096)), None, dict(race=0)),
         ("config3 65536", q.to_device(W.config3(65536)), None, dict()),
         ("config3 16384 (4 lanes)", q.to_device(W.config3(16384)), None, dict()),
         ("config3 32768 (2 lanes)", q.to_device(W.config3(32768)), None, dict()),
         ("config4 tick 1 warm 262144", q.to_device(t1), w4, dict()),
         ("config4 tick 0 cold 262144", q.to_device(t0), None, dict()),
         ("config5 shard 262144", q.to_device(W.config5(262144)), None, dict()),
         ("config3 65536 general form", q.to_device(W.config3(65536)), None, dict(force_general=1))]
for name, b, w, tune in CASES:
    row = []
    for k in (tuple(int(a) for a in sys.argv[1:]) or (1, 2, 3)):
        us, mean, mx = timeit(q.BalanceController.from_params(P).set_tuning(clamp_steps=k, **tune), b, w)
        row.append("%d: %6.1f us (mean %.2f max %2d)" % (k, us, mean, mx))
    print("%-30s " % name + " | ".join(row), flush=True)
