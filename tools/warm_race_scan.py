import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
def timeit(ctl, b, warm, reps=100):
    launch, out = ctl.plan_batch(b, warm=warm, want_iterations=True)
    for _ in range(10): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, int(out["iterations"].max())
for n in (1024, 2048, 4096):
    for seed in (0x5EED0004, 0x5EED0F04, 0x5EED1F04):
        t0, t1 = W.config4(n, seed=seed)
        w = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
        b = q.to_device(t1)
        r = []
        for race in (0, 2, 4):
            r.append("race %d: %.2f us (max it %d)" % ((race,) + timeit(q.BalanceController.from_params(P).set_tuning(race=race), b, w)))
        print("warm tick n=%d seed %x: " % (n, seed) + " | ".join(r), flush=True)
