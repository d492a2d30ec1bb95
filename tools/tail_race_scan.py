"""Development tool: the 2-way race in the 4-lane tail (one / two lanes per robot) on / off, config-3 distribution over
several seeds and config 4's warm tick.  usage: python tools/tail_race_scan.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W, workloads_device as WD
P = q.cheetah_params(0.6)
def timeit(ctl, b, warm=None, reps=20):
    launch, out = ctl.plan_batch(b, warm=warm, want_iterations=True)
    for _ in range(3): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    return e0.elapsed_time(e1) / reps * 1e3, int(out["iterations"].max())
SEEDS = [0x5EED0003 + 0x1000 * k for k in range(8)]
for n in (6144, 8192, 12288, 16384, 32768, 65536, 131072, 262144, 1048576):
    rows = {"off": [], "on": []}
    for seed in SEEDS:
        b = WD.config3(n, seed=seed)
        rows["off"].append(timeit(q.BalanceController.from_params(P).set_tuning(race=0), b))
        rows["on"].append(timeit(q.BalanceController.from_params(P), b))
    print("cold n=%8d: " % n + " | ".join("%s: %6.1f us (max it %s)" % (k, np.mean([r[0] for r in v]), ",".join(str(r[1]) for r in v)) for k, v in rows.items()), flush=True)
for n in (65536, 262144):
    t0, t1 = W.config4(n)
    w = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
    b = q.to_device(t1)
    print("warm n=%8d: off %6.1f us (max it %d) | on %6.1f us (max it %d)" % ((n,) + timeit(q.BalanceController.from_params(P).set_tuning(race=0), b, w) + timeit(q.BalanceController.from_params(P), b, w)), flush=True)
