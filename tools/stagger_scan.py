"""Development tool (needs a -DQC_EXPERIMENT_STAGGER build, QC_LIB_PATH=tools/_build/libqc_stagger.so): the second wave of every
SIMD waits `stagger` x 1024 cycles before its first load in the launch's first round.  usage: python tools/stagger_scan.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
def timeit(ctl, sets, warm, reps=40):
    plans = [ctl.plan_batch(b, warm=w) for b, w in zip(sets, warm)]
    for l, _ in plans: l()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps): plans[k % len(plans)][0]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def rot(gen, k):  # k distinct input sets: > 512 MiB in total = the cold-cache protocol
    return [gen(i) for i in range(k)]
t0, t1 = W.config4(262144)
w4 = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
CASES = [("config4 warm tick, one resident set", [q.to_device(t1)], [w4]),
         ("config4 warm tick, 5 rotating sets (cold)", [q.to_device(t1) for _ in range(5)], [w4.clone() for _ in range(5)]),
         ("config5 shard, one resident set", [q.to_device(W.config5(262144))], [None]),
         ("config5 shard, 5 rotating sets (cold)", [q.to_device(W.config5(262144)) for _ in range(5)], [None] * 5),
         ("config3 131072, 9 rotating sets (cold)", [q.to_device(W.config3(131072)) for _ in range(9)], [None] * 9)]
for name, sets, warm in CASES:
    row = []
    for st in (0, 1, 2, 4, 6, 8, 12):
        ctl = q.BalanceController.from_params(P).set_tuning(stagger=st)
        row.append("%d: %.1f" % (st, timeit(ctl, sets, warm)))
    print("%-45s " % name + " | ".join(row) + "  us", flush=True)
