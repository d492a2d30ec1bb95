"""Development tool: clamp_scan.py over several seeds (config-3 distribution, cold start) - the batch maximum of recalculations
moves the kernel time, so one seed says little.  usage: python tools/clamp_scan_seeds.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads_device as WD
P = q.cheetah_params(0.6)
def timeit(ctl, b, reps=20):
    launch, out = ctl.plan_batch(b, want_iterations=True)
    for _ in range(3): launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    return e0.elapsed_time(e1) / reps * 1e3, int(out["iterations"].max())
KS = tuple(int(a) for a in sys.argv[1:]) or (1, 2, 3, 4)
SEEDS = [0x5EED0003 + 0x1000 * k for k in range(8)]
for n, tune in ((4096, dict(race=0)), (8192, {}), (16384, {}), (32768, {}), (65536, {}), (131072, {}), (262144, {}), (1048576, {})):
    rows = {k: [] for k in KS}
    for seed in SEEDS:
        b = WD.config3(n, seed=seed)
        for k in rows:
            rows[k].append(timeit(q.BalanceController.from_params(P).set_tuning(clamp_steps=k, **tune), b))
    print("n=%8d %s: " % (n, tune) + " | ".join("%d: %6.1f us (max it %s)" % (k, np.mean([r[0] for r in v]), ",".join(str(r[1]) for r in v)) for k, v in rows.items()), flush=True)
