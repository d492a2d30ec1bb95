"""Development tool: kernel time vs batch size for the candidate (lanes per robot, mode) pairs - where the planner's
thresholds come from.  usage: python tools/size_scan.py [cold|warm]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
warm_mode = len(sys.argv) > 1 and sys.argv[1] == "warm"
def timeit(ctl, b, warm, reps):
    launch, out = ctl.plan_batch(b, warm=warm)
    for _ in range(5): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    return e0.elapsed_time(e1) / reps * 1e3
VAR = [("auto", dict()), ("G4 1fill", dict(group=4, one_fill=1)), ("G2 1fill", dict(group=2, one_fill=1)), ("G1 1fill", dict(group=1, one_fill=1)),
       ("G2 pers", dict(group=2, one_fill=0)), ("G1 pers", dict(group=1, one_fill=0))]
print("%9s " % "n" + " ".join("%9s" % v[0] for v in VAR))
for n in (4096, 8192, 16384, 24576, 32768, 49152, 65536, 98304, 131072, 196608, 262144, 524288, 1048576, 2097152):
    if warm_mode:
        t0, t1 = W.config4(n)
        w = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
        b = q.to_device(t1)
    else:
        b, w = q.to_device(W.config3(n)), None
    row = []
    for name, tune in VAR:
        if name == "G4 1fill" and n > 262144: row.append(float("nan")); continue
        try:
            row.append(timeit(q.BalanceController.from_params(P).set_tuning(**tune), b, w, 20 if n <= 262144 else 6))
        except RuntimeError:  # "pers" columns need a -DQC_PERSISTENT_6X6=1 build (an error, not a silent one-fill run, since round 4)
            row.append(float("nan"))
    best = int(np.nanargmin(row[1:])) + 1
    print("%9d " % n + " ".join("%9.1f" % v for v in row) + "   best: " + VAR[best][0], flush=True)
