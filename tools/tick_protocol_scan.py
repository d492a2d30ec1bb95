"""where do the 6-9 % between the profiled run (bench.py --tick full, 200 steps after 20 warm-up launches:
143-148 us) and the sweep entry `full_tick_262144` (50 steps after 3 warm-up launches: 157-160 us) come from?

Builds the entry's launches once (bench.run_config with time_launches intercepted) and times the SAME launches under
different (steps, warm-up launches, warm-up milliseconds, idle seconds before) protocols, then prints a per-launch trace
(one HIP event per launch) taken right after an idle gap: a device that is still raising its clocks shows as a ramp.

  python tools/tick_protocol_scan.py [robots] [full|full-frozen]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import quadruped_control_amd as q  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
tick = sys.argv[2] if len(sys.argv) > 2 else "full-frozen"
COMBOS = [  # steps, warm-up launches, warm-up ms, idle seconds before
    (50, 3, 0.0, 0.0), (50, 3, 0.0, 2.0), (50, 3, 0.0, 2.0), (50, 20, 0.0, 2.0), (50, 20, 25.0, 2.0), (50, 20, 100.0, 2.0),
    (200, 20, 0.0, 2.0), (200, 20, 25.0, 2.0), (200, 3, 0.0, 2.0), (20, 3, 0.0, 2.0), (20, 20, 25.0, 2.0), (50, 3, 0.0, 0.0), (50, 20, 25.0, 0.0)]
orig = bench.time_launches


def scan(launches, steps, warmup, dist=None, warm_all=False, warm_ms=0.0, before_timed=None):
    m = len(launches)
    print(f"# {tick} tick, {n} robots, {m} rotating sets", flush=True)
    for st, wu, wm, idle in COMBOS:
        time.sleep(idle)
        wall, ev = orig(launches, st, wu, None, warm_all, wm, None)
        print(json.dumps({"steps": st, "warmup": wu, "warm_ms": wm, "idle_s": idle, "kernel_us": round(ev / st * 1e6, 2),
                          "wall_us_per_step": round(wall / st * 1e6, 2)}), flush=True)
    # per-launch trace right after an idle gap
    for idle in (2.0, 0.0):
        time.sleep(idle)
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(121)]
        evs[0].record()
        for i in range(120):
            launches[i % m]()
            evs[i + 1].record()
        torch.cuda.synchronize()
        d = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(120)]
        print(f"# per-launch us after {idle:.0f} s idle: first 12 = {[round(x, 1) for x in d[:12]]}; launches 12-59 mean {sum(d[12:60]) / 48:.1f}; "
              f"60-119 mean {sum(d[60:]) / 60:.1f}", flush=True)
    return orig(launches, steps, warmup, dist, warm_all, warm_ms, before_timed)


bench.time_launches = scan
P = q.cheetah_params(mu=0.6)
ctl = q.BalanceController.from_params(P, device=0)
bench.run_config(ctl, q, 3, n, 0, 50, 3, None, 0, fused=tick, protocols=("cold",))
