"""Development tool: where a big warm batch's time goes - the same launch with the solver capped at 0 (load ->
assemble -> store only), 1, 2 recalculations and uncapped, per kernel mode.  usage: python tools/phase_split.py [n]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
P = q.cheetah_params(0.6)
t0, t1 = W.config4(n)
d0, d1 = q.to_device(t0), q.to_device(t1)
w = q.BalanceController.from_params(P).control_batch(d0, want_active_set=True)["active_set"]
def timeit(ctl, b, warm, reps=20):
    launch, out = ctl.plan_batch(b, warm=warm)
    for _ in range(3): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, tune in (("mode1 G1", dict(group=1, one_fill=1)), ("mode1 G2", dict(group=2, one_fill=1)), ("mode0 G2", dict(group=2, one_fill=0)), ("mode1 G4", dict(group=4, one_fill=1)), ("mode0 G1", dict(group=1, one_fill=0))):
    row = []
    try:  # persistent (mode 0) kernels of the 6x6 forms exist only in -DQC_PERSISTENT_6X6=1 builds: asking for one is an error since round 4
        q.BalanceController.from_params(P).set_tuning(**tune).query_launch(n, warm=True)
    except RuntimeError as e:
        print("%-9s n/a in this build (%s)" % (name, str(e).split(":")[-1].strip()[:80]), flush=True)
        continue
    for cap in (0, 1, 2, 200):
        ctl = q.BalanceController.from_params(P).set_tuning(**tune)
        if cap == 0: ctl.set_tuning(probe_batch_load=1)
        else: ctl.set_tuning(max_iter=cap)
        row.append(timeit(ctl, d1, w))
    cold = timeit(q.BalanceController.from_params(P).set_tuning(**tune), d1, None)
    print("%-9s n=%d warm: cap0 %.1f  cap1 %.1f  cap2 %.1f  full %.1f us | cold full %.1f us" % (name, n, *row, cold), flush=True)
