"""Development tool: persistent-wave chunk size x refill threshold on config 5 (one shard and the full batch).
usage: python tools/chunk_scan.py"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
def timeit(ctl, b, reps=10):
    launch, out = ctl.plan_batch(b)
    for _ in range(2): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    return e0.elapsed_time(e1) / reps * 1e3
for n in (262144, 2097152):
    b = q.to_device(W.config5(n))
    print("n", n, "default %.1f us" % timeit(q.BalanceController.from_params(P), b), flush=True)
    for chunk in (32, 64, 96, 128, 192, 256, 512, 1024):
        row = []
        for rt in (8, 16, 32):
            ctl = q.BalanceController.from_params(P).set_tuning(group=2, chunk=chunk, refill_t=rt)
            row.append(timeit(ctl, b))
        print("  chunk %4d: refill_t 8/16/32 -> %s us   (blocks %d)" % (chunk, " / ".join("%.1f" % v for v in row), (n + chunk - 1) // chunk), flush=True)
