"""Per-iteration cost scan: time the kernel with max_iter = 1..K (development tool)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
P = q.cheetah_params(0.6)
b = q.to_device({2: W.config2, 3: W.config3}[cfg](n))
def t(ctl, reps=100):
    out = ctl.control_batch(b)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ctl.control_batch(b, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for k in (1, 2, 3, 4, 6, 8, 12, 16, 200):
    ctl = q.BalanceController.from_params(P, max_iter=k)
    print("max_iter %3d: %.1f us" % (k, t(ctl)))
ctl = q.BalanceController.from_params(P)
o = ctl.control_batch(b, want_iterations=True); torch.cuda.synchronize()
it = o["iterations"].cpu().numpy()
print("iters mean %.2f max %d; per-wave max mean %.2f; hist %s" % (it.mean(), it.max(), it.reshape(-1, 64).max(1).mean(), np.bincount(it)))
