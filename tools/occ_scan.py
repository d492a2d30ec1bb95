"""Development tool: A/B of lanes per robot x kernel mode x register cap (waves per SIMD) on the throughput configs.
Needs a library built with -DQC_EXPERIMENTAL_OCC (tools/_build/libqc_balance_occ.so, see tools/README.md) for the
min_waves > 2 rows.  usage: python tools/occ_scan.py [lib] ; prints one line per variant: kernel time (HIP events)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from quadruped_control_amd import _lib
if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W

P = q.cheetah_params(0.6)


def timeit(ctl, b, warm, reps=20):
    launch, out = ctl.plan_batch(b, warm=warm)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record(); torch.cuda.synchronize()
    assert int((out["status"] != 0).sum()) == 0
    return e0.elapsed_time(e1) / reps * 1e3, out


VARIANTS = [dict(), dict(group=1, one_fill=1), dict(group=2, one_fill=1), dict(group=2, one_fill=1, min_waves=3), dict(group=2, one_fill=1, min_waves=4),
            dict(group=4, one_fill=1), dict(group=4, one_fill=1, min_waves=3), dict(group=4, one_fill=1, min_waves=4),
            dict(group=2, one_fill=0), dict(group=4, one_fill=0), dict(group=1)]
work = []
b3 = q.to_device(W.config3(65536)); work.append(("cfg3 65536 cold", b3, None, 65536))
t0, t1 = W.config4(262144)
d0, d1 = q.to_device(t0), q.to_device(t1)
w = q.BalanceController.from_params(P).control_batch(d0, want_active_set=True)["active_set"]
work.append(("cfg4 262144 warm", d1, w, 262144))
b5 = q.to_device(W.config5(262144)); work.append(("cfg5-shard 262144 cold", b5, None, 262144))
if "--big" in sys.argv:
    b5b = q.to_device(W.config5(1048576)); work.append(("cfg5 1M cold", b5b, None, 1048576))
ref = {}
for name, b, warm, n in work:
    for v in VARIANTS:
        ctl = q.BalanceController.from_params(P)
        try:
            ctl.set_tuning(**v)
        except ValueError as e:
            print(name, v, "skipped:", e); continue
        try:
            info = ctl.query_launch(n, warm=warm is not None)
        except RuntimeError as e:  # persistent 6x6 kernels: -DQC_PERSISTENT_6X6=1 builds only
            print(name, v, "skipped:", str(e).split(":")[-1].strip()[:90]); continue
        try:
            us, out = timeit(ctl, b, warm)
        except Exception as e:
            print(name, v, "FAILED", e); continue
        g = out["grf_body"]
        if name not in ref:
            ref[name] = g.clone()
        err = float((g - ref[name]).abs().max())
        print("%-24s %-44s G=%d mode=%d resident=%5d lds=%5d : %8.1f us  %.3e QP/s  maxdiff %.1e" %
              (name, json.dumps(v), info["lanes_per_robot"], info["mode"], info["resident_workgroups"], info["lds_bytes"], us, n / us * 1e6, err), flush=True)
