"""Round 6 development tool: the polish at acceptance on config 4's warm-started tick - iteration counts with polish = 0 against the default."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
for j in range(3):
    t0, t1 = W.config4(262144, seed=W.SEEDS[4] + 0x100 * j)
    w = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
    d1 = q.to_device(t1)
    res = {}
    for pol in (0, 1):
        o = q.BalanceController.from_params(P).set_tuning(polish=pol).control_batch(d1, warm=w, want_iterations=True, want_active_set=True)
        res[pol] = (o["iterations"].cpu().numpy(), o["grf_body"].cpu().numpy(), o["active_set"].cpu().numpy())
    i0, i1 = res[0][0], res[1][0]
    print("cfg4 set %d: iters mean %.4f -> %.4f  max %d -> %d  robots with more iterations %d (fewer %d)  other working set %d  max |df| %.2e" %
          (j, i0.mean(), i1.mean(), i0.max(), i1.max(), int((i1 > i0).sum()), int((i1 < i0).sum()), int((res[0][2] != res[1][2]).sum()), np.abs(res[0][1] - res[1][1]).max()), flush=True)
