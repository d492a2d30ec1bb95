"""Single-robot latency breakdown (development tool): python wrapper vs raw C call vs device-resident launch+sync."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import ctypes as C
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W, _lib

P = q.cheetah_params(0.6)
ctl = q.BalanceController.from_params(P)
b = W.config1()
names = ctl.leg_names
fm = {nm: b["feet"].reshape(-1, 4, 3)[0, i] for i, nm in enumerate(names)}
a = [b[k][0] for k in ("Rwb", "Rwb_d", "x", "xdot", "w", "x_d", "xdot_d", "w_d")]

def timeit(f, reps=300):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6

print("python control():        %.1f us" % timeit(lambda: ctl.control(*a, fm)))
arrs = [np.ascontiguousarray(v.reshape(-1)) for v in a] + [np.ascontiguousarray(b["feet"][0]), np.ones(4, np.uint8)]
grf = np.zeros(12); st = np.zeros(1, np.int32)
ptrs = [C.c_void_p(v.ctypes.data) for v in arrs] + [C.c_void_p(grf.ctypes.data), C.c_void_p(st.ctypes.data)]
lib = ctl._lib
print("raw qc_control:          %.1f us" % timeit(lambda: lib.qc_control(ctl._h, *ptrs)))
db = q.to_device(b)
launch, out = ctl.plan_batch(db)
def dev():
    launch(); torch.cuda.synchronize()
print("device launch + sync:    %.1f us" % timeit(dev))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
launch(); torch.cuda.synchronize()
e0.record()
for _ in range(100): launch()
e1.record(); torch.cuda.synchronize()
print("device kernel (events):  %.1f us" % (e0.elapsed_time(e1) * 10))
for n in (1, 16, 64):
    hb = W.config2(n)
    print("control_batch_host n=%d: %.1f us" % (n, timeit(lambda: ctl.control_batch_host(hb))))
