"""Round 5: the one-lane dense 12x12 kernel (general SPD W, `force_dense`) as one-fill workgroups vs persistent waves over batch sizes,
with the resident workgroups the occupancy query reports (the LDS diet: 58 KB -> 39 KB per workgroup = 2 -> 4 per CU).
usage: python tools/dense_sizes.py [sizes...]"""
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
sizes = [int(a) for a in sys.argv[1:]] or [4096, 16384, 32768, 65536, 131072, 262144, 1048576]
for n in sizes:
    b = WD.config3(n, start=0, seed=W.SEEDS[3], device=0)
    ref = None
    for tune in (dict(), dict(force_dense=1), dict(force_dense=1, group=1, one_fill=1), dict(force_dense=1, group=1, one_fill=0), dict(force_dense=1, group=4)):
        ctl = q.BalanceController.from_params(P).set_tuning(**tune)
        try:
            info = ctl.query_launch(n)
            us, out = timeit(ctl, b)
        except Exception as e:
            print(n, tune, "->", str(e)[:100]); continue
        g = out["grf_body"]
        ref = g.clone() if ref is None else ref
        print("config3 %8d %-46s form=%d G=%d mode=%d resident=%5d lds=%6d : %8.1f us  %.3e QP/s  max iters %d  maxdiff vs uniform %.1e" %
              (n, tune, info["form"], info["lanes_per_robot"], info["mode"], info["resident_workgroups"], info["lds_bytes"], us, n / us * 1e6,
               int(out["iterations"].max()), float((g - ref).abs().max())), flush=True)
