"""Complete-tick throughput vs batch size and lanes per robot (development tool): python tools/tick_scan.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import quadruped_control_amd as q
import bench
P = q.cheetah_params(0.6)
for n in (4096, 16384, 32768, 65536, 131072, 262144, 1048576):
    for mode in (True, "full"):
        row = []
        for g in (0, 1, 2, 4):
            if g == 4 and n > 262144:
                row.append("      -"); continue
            ctl = q.BalanceController.from_params(P).set_tuning(group=g)
            r = bench.run_config(ctl, q, 3, n, 0, 20, 5, None, 0, fused=mode, protocols=("warm",))
            row.append("%7.1f" % (r["warm_cache"][1] / 20 * 1e6))
        print("n=%8d %-5s  auto / G1 / G2 / G4 (us): %s" % (n, "FK+JT" if mode is True else "full", " ".join(row)), flush=True)
