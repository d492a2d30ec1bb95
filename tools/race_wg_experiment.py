import os, sys
sys.path.insert(0, "/root/repo")
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
    return e0.elapsed_time(e1) / reps * 1e3, int(out["iterations"].max())
for n in (1024, 4096):
    b = q.to_device(W.config2(n))
    for name, kw in (("solo (race 0)", dict(race=0)), ("wg race", dict()), ("wg, only wave 0 works", dict(refill_t=4)), ("wg, all four waves strategy 0", dict(refill_t=8))):
        ctl = q.BalanceController.from_params(P).set_tuning(**kw)
        print(n, name, "%.1f us, max it %d" % timeit(ctl, b), flush=True)
