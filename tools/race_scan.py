"""Development tool: racing pivoting strategies (0 / 2 / 4 per robot) vs batch size on config-2 and config-3 inputs.
usage: python tools/race_scan.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
def timeit(ctl, b, reps=50):
    launch, out = ctl.plan_batch(b, want_iterations=True)
    for _ in range(5): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    return e0.elapsed_time(e1) / reps * 1e3, int(out["iterations"].max())
for name, gen in (("config2", W.config2), ("config3", W.config3)):
    for seed in (None, 0x5EED0777):
        for n in (1024, 2048, 4096, 6144, 8192):
            b = q.to_device(gen(n) if seed is None else gen(n, seed=seed))
            row = []
            for race in (0, 2, 4):
                ctl = q.BalanceController.from_params(P).set_tuning(race=race)
                info = ctl.query_launch(n)
                us, mx = timeit(ctl, b)
                row.append("race %d(%d): %5.1f us max it %2d" % (race, info["strategies"], us, mx))
            print("%s seed %s n=%5d  " % (name, "std" if seed is None else "%x" % seed, n) + " | ".join(row), flush=True)
