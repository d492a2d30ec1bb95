"""Config-4 statistics (development tool): recalculations per robot on the warm-started tick and iterate calls per wave."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ctl = q.BalanceController.from_params(q.cheetah_params(0.6))
t0, t1 = W.config4(n)
o0 = ctl.control_batch(q.to_device(t0), want_active_set=True)
o1 = ctl.control_batch(q.to_device(t1), warm=o0["active_set"], want_iterations=True)
torch.cuda.synchronize()
it = o1["iterations"].cpu().numpy()
print("warm tick: mean %.3f max %d hist %s" % (it.mean(), it.max(), np.bincount(it)))
for chunk, rpw in ((128, 32),):
    c = it[: n // chunk * chunk].reshape(-1, chunk)
    print("per-chunk(%d) sum/%d = %.2f fills-worth of recalculations (perfect packing); per-chunk max mean %.2f, max %d" %
          (chunk, rpw, c.sum(1).mean() / rpw, c.max(1).mean(), c.max()))
    # tail model: the last fill's stragglers iterate alone: mean over chunks of max over the last 32 robots
    print("mean of max over last %d robots of each chunk: %.2f" % (rpw, c[:, -rpw:].max(1).mean()))
