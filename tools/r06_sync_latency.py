"""Round 6 development tool: where the driver's 20-step region spends the time outside its kernels - wall clock around K launches
(barrier-free, one process) against the HIP-event time of the same launches, with the runtime's default wait and with
hipDeviceScheduleSpin set before the device is first touched.  usage: python tools/r06_sync_latency.py [spin]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
spin = len(sys.argv) > 1 and sys.argv[1] == "spin"
if spin:
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", hip.hipSetDeviceFlags(ctypes.c_uint(1)))
import torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
ctl = q.BalanceController.from_params(P)
sets = []
NSETS = int(os.environ.get("NSETS", "32"))
for j in range(NSETS):
    b = q.to_device(W.config2(4096, seed=W.SEEDS[2] + 0x100 * j))
    o = {"grf_body": torch.empty((4096, 12), dtype=torch.float64, device="cuda"), "status": torch.empty((4096,), dtype=torch.int32, device="cuda")}
    sets.append(ctl.plan_batch(b, out=o)[0])
for K in (20,):
    for rep in range(5):
        for i in range(300): sets[i % NSETS]()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); e0.record()
        for i in range(K): sets[i % NSETS]()
        t1 = time.perf_counter()
        e1.record(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        ev = e0.elapsed_time(e1) * 1e3
        print("%s K=%3d: wall %.1f us = %.2f us/step; events %.1f us = %.2f us/step; enqueue of the K launches %.1f us; wall - events %.1f us" %
              ("spin" if spin else "default", K, (t2 - t0) * 1e6, (t2 - t0) * 1e6 / K, ev, ev / K, (t1 - t0) * 1e6, (t2 - t0) * 1e6 - ev), flush=True)
