"""Round 5: would the one-lane dense kernel gain from fewer robots per wave where the batch leaves SIMDs idle (16 384 < n <= 65 536: more waves, every
wave's idle lanes free for the twin race from the first steady recalculation on)?  No: chunk 32 / 16 are equal or slower at every size - the launch is its
slowest robot's chain either way (profiles/r05_dense_chunk.log).  The planner's rule (four lanes up to 16 384 robots, one lane with 64-robot fills above) stands.
usage: python tools/dense_chunk_scan.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
from quadruped_control_amd import workloads_device as WD
P = q.cheetah_params(0.6)
def timeit(ctl, b, reps=20):
    launch, out = ctl.plan_batch(b, want_iterations=True)
    for _ in range(5): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    return e0.elapsed_time(e1) / reps * 1e3, out
for n in (8192, 16384, 20000, 24576, 32768, 49152, 65536):
    b = WD.config3(n, start=0, seed=W.SEEDS[3], device=0)
    row = []
    for tune in (dict(force_dense=1), dict(force_dense=1, group=1), dict(force_dense=1, group=1, chunk=32), dict(force_dense=1, group=1, chunk=16), dict(force_dense=1, group=4)):
        ctl = q.BalanceController.from_params(P).set_tuning(**tune)
        try:
            info = ctl.query_launch(n); us, out = timeit(ctl, b)
            row.append("%s G%d chunk%d: %.1f us (max it %d)" % ({k: v for k, v in tune.items() if k != "force_dense"}, info["lanes_per_robot"], info["chunk"], us, int(out["iterations"].max())))
        except Exception as e:
            row.append("%s -> %s" % (tune, str(e)[:60]))
    print(n, " | ".join(row), flush=True)
