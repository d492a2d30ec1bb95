"""Dev tool: how much of a big warm batch is load/assemble/store? Compare max_iter=1 with the full solve."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
P = q.cheetah_params(0.6)
t0, t1 = W.config4(n)
d0, d1 = q.to_device(t0), q.to_device(t1)
def timeit(ctl, b, warm, reps=10):
    launch, out = ctl.plan_batch(b, warm=warm)
    for _ in range(2): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
full = q.BalanceController.from_params(P)
w = full.control_batch(d0, want_active_set=True)["active_set"]
one = q.BalanceController.from_params(P, max_iter=1)
two = q.BalanceController.from_params(P, max_iter=2)
print("n", n)
print("warm full   : %.1f us" % timeit(full, d1, w))
print("warm iter<=1: %.1f us" % timeit(one, d1, w))
print("warm iter<=2: %.1f us" % timeit(two, d1, w))
print("cold full   : %.1f us" % timeit(full, d1, None))
print("cold iter<=1: %.1f us" % timeit(one, d1, None))
print("cold iter<=4: %.1f us" % timeit(q.BalanceController.from_params(P, max_iter=4), d1, None))
