import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
b = W.config3(65536)
for cs in (1, 0):
    ctl = q.BalanceController.from_params(P).set_tuning(clamp_steps=cs)
    it = ctl.control_batch(q.to_device(b), want_iterations=True)["iterations"].cpu().numpy()
    ns = b["stance"].sum(1)
    print("clamp_steps", cs)
    for k in (2, 3, 4):
        m = ns == k
        print(" ", k, "feet: share %.3f mean %.2f max %d" % (m.mean(), it[m].mean(), it[m].max()), np.bincount(it[m]))
    print("  unsorted: mean %.2f, per-64 max mean %.2f; 17th largest of 64 (tail starts) mean %.2f" % (it.mean(), it.reshape(-1, 64).max(1).mean(), np.sort(it.reshape(-1, 64), 1)[:, -17].mean()))
    o = np.argsort(ns, kind="stable")
    s = it[o].reshape(-1, 64)
    print("  sorted by stance count: per-64 max mean %.2f; 17th largest mean %.2f" % (s.max(1).mean(), np.sort(s, 1)[:, -17].mean()))
    s = np.sort(it).reshape(-1, 64)
    print("  perfectly sorted: per-64 max mean %.2f" % s.max(1).mean())
