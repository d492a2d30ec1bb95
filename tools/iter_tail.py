"""Recalculation-count tail statistics on a large sample (development tool): python tools/iter_tail.py [n] [cfg]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctl = q.BalanceController.from_params(q.cheetah_params(0.6))
b = q.to_device({2: W.config2, 3: W.config3}[cfg](n, seed=0x5EED00AB))
o = ctl.control_batch(b, want_iterations=True); torch.cuda.synchronize()
it = o["iterations"].cpu().numpy()
h = np.bincount(it)
print("cfg%d n=%d: mean %.3f p99 %d p99.9 %d p99.99 %d max %d; >=14: %d, >=17: %d, >=20: %d" %
      (cfg, n, it.mean(), np.percentile(it, 99), np.percentile(it, 99.9), np.percentile(it, 99.99), it.max(), (it >= 14).sum(), (it >= 17).sum(), (it >= 20).sum()))
for size in (4096, 65536):
    m = it[: n // size * size].reshape(-1, size).max(1)
    print("   max over batches of %d: mean %.2f (min %d max %d)" % (size, m.mean(), m.min(), m.max()))
