"""Development tool: the streaming kernel (mode 3: persistent waves, LDS-DMA prefetch of the next fill) against the one-fill
kernels, warm-started ticks (config 4) and cold batches (config 3 distribution) vs batch size.  usage: python tools/stream_scan.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
def timeit(ctl, b, warm, reps=30):
    launch, out = ctl.plan_batch(b, warm=warm)
    for _ in range(5): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    return e0.elapsed_time(e1) / reps * 1e3, out["grf_body"].clone()
for kind in ("warm", "cold"):
    for n in (65536, 131072, 196608, 262144, 524288, 1048576, 2097152):
        if kind == "warm":
            t0, t1 = W.config4(n)
            w = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
            b = q.to_device(t1)
        else:
            b, w = q.to_device(W.config3(n)), None
        row = []
        ref = None
        for name, tune in (("one-fill", dict(stream=0)), ("stream", dict(stream=1)), ("one-fill again", dict(stream=0)), ("stream again", dict(stream=1))):
            ctl = q.BalanceController.from_params(P).set_tuning(**tune)
            info = ctl.query_launch(n, warm=w is not None)
            us, grf = timeit(ctl, b, w, 30 if n <= 262144 else 8)
            if ref is None: ref = grf
            err = float((grf - ref).abs().max())
            row.append("%s (mode %d, %d blocks): %7.1f us, max diff %.1e" % (name, info["mode"], info["blocks"], us, err))
        print("%s n=%8d  " % (kind, n) + " | ".join(row), flush=True)
