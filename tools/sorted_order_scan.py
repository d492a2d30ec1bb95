"""Development tool: what ordering the batch by stance-foot count would buy (inputs permuted on the host before the launch - the
potential of an on-device pre-pass, not a feature).  usage: python tools/sorted_order_scan.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
def timeit(ctl, b, reps=30):
    launch, out = ctl.plan_batch(b, want_iterations=True)
    for _ in range(5): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    it = out["iterations"].cpu().numpy()
    return e0.elapsed_time(e1) / reps * 1e3, it.reshape(-1, 64).max(1).mean()
for n in (65536, 262144, 1048576):
    b = W.config3(n, seed=0x5EED0005)
    ns = b["stance"].sum(1)
    orders = {"as generated": np.arange(n), "4-foot robots first": np.argsort(-ns, kind="stable"), "2-foot robots first": np.argsort(ns, kind="stable"),
              "random shuffle": np.random.default_rng(1).permutation(n)}
    row = []
    for name, o in orders.items():
        bo = {k: np.ascontiguousarray(v[o]) for k, v in b.items()}
        us, pm = timeit(q.BalanceController.from_params(P), q.to_device(bo), 30 if n <= 262144 else 10)
        row.append("%s: %.1f us (per-wave max %.2f)" % (name, us, pm))
    print("n=%8d  " % n + " | ".join(row), flush=True)
