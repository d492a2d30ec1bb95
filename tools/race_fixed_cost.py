"""Development tool: fixed cost (fill + flush, no recalculation) and cost with the recalculation cap at 1 / 2 / 4 / 8 of the
4-lane kernels on 4096 config-2 robots, one strategy per robot vs four racing.  usage: python tools/race_fixed_cost.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
b = q.to_device(W.config2(4096))
def timeit(ctl, reps=200):
    launch, out = ctl.plan_batch(b)
    for _ in range(10): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for race in (0, 2, 4):
    row = []
    ctl = q.BalanceController.from_params(P).set_tuning(race=race, probe_batch_load=1)
    row.append("probe %.2f" % timeit(ctl))
    for cap in (1, 2, 4, 8, 200):
        ctl = q.BalanceController.from_params(P, max_iter=cap).set_tuning(race=race)
        row.append("cap %d: %.2f" % (cap, timeit(ctl)))
    print("race %d (%s): " % (race, ctl.query_launch(4096)) + " | ".join(row) + "  us", flush=True)
