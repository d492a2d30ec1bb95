"""Round 6 development tool: what the polish at acceptance changes on the bench workloads - how many robots take a polish
release (iteration counts with polish = 0 against the default), and what it does to the forces."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
for name, n, mk in (("cfg2", 4096, lambda s: W.config2(4096, seed=s)), ("cfg3", 65536, lambda s: W.config3(65536, seed=s)),
                    ("cfg5s", 262144, lambda s: W.config3(262144, seed=s))):
    for j in range(3):
        b = q.to_device(mk(W.SEEDS[2] + 0x100 * j))
        res = {}
        for pol in (0, 1):
            ctl = q.BalanceController.from_params(P).set_tuning(polish=pol)
            o = ctl.control_batch(b, want_iterations=True, want_active_set=True)
            res[pol] = (o["iterations"].cpu().numpy(), o["grf_body"].cpu().numpy(), o["active_set"].cpu().numpy())
        i0, i2 = res[0][0], res[1][0]
        d = np.abs(res[0][1] - res[1][1]).max(axis=1)
        print("%s set %d: iters mean %.3f -> %.3f  max %d -> %d  robots with more iterations %d (fewer %d)  other working set %d  max |df| %.2e" %
              (name, j, i0.mean(), i2.mean(), i0.max(), i2.max(), int((i2 > i0).sum()), int((i2 < i0).sum()), int((res[0][2] != res[1][2]).sum()), d.max()), flush=True)
