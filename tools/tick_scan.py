"""Complete-tick throughput vs batch size (development tool): python tools/tick_scan.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import quadruped_control_amd as q
import bench
ctl = q.BalanceController.from_params(q.cheetah_params(0.6))
for n in (4096, 65536, 131072, 262144, 1048576):
    for mode in (True, "full"):
        r = bench.run_config(ctl, q, 3, n, 0, 20, 3, None, 0, fused=mode)
        print("n=%8d %-5s %.3e ticks/s  %.1f us" % (n, "FK+JT" if mode is True else "full", n * 20 / r["wall"], r["wall"] / 20 * 1e6))
