"""Development tool: per-step time of the config-2 cold-cache loop vs the number of timed steps K and an untimed spin-up before them
(is a 20-step measurement taken at ramped-up clocks?).  After 0.3 s of idling a 20-step loop is 20-40 % slower per step than a
2 000-step one; inside bench.py the set-up work keeps the device awake and a spin-up changes nothing (tried, not kept): what remains
at K = 20 is ~40 us of fixed cost per timed region (first launch after the synchronise, the wake-up of the final one) = 2 us per step.
usage: python tools/steps_scan.py"""
import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
ctl = q.BalanceController.from_params(P)
sets = [q.to_device(W.config2(4096, seed=0x5EED0002 + 7 * i)) for i in range(269)]
launches = [ctl.plan_batch(b)[0] for b in sets]
def run(K, spin_ms, W_=5):
    for i in range(max(W_, 8)): launches[i % 269]()
    t = time.perf_counter(); i = 0
    while (time.perf_counter() - t) * 1e3 < spin_ms:
        launches[i % 269](); i += 1
        if i % 64 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for i in range(K): launches[(W_ + i) % 269]()
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6, e0.elapsed_time(e1) / K * 1e3
for trial in range(3):
    for K, spin in ((20, 0), (20, 5), (20, 30), (200, 0), (200, 30), (2000, 0)):
        time.sleep(0.3)  # let the device idle, as after the set-up phase of bench.py
        w, e = run(K, spin)
        print("K=%4d spin %2d ms: wall %.2f us/step, events %.2f us/step" % (K, spin, w, e), flush=True)
