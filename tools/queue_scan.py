"""Development tool (round 3): the straggler-queue kernel (MODE 3, qc_set_tuning("queue")) against the one-fill kernel
it replaces - results (max difference, status, iteration semantics) and kernel time, warm-started and cold, over the
list length K and the hand-over threshold.  usage: python tools/queue_scan.py [quick]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
from quadruped_control_amd import workloads_device as WD
P = q.cheetah_params(0.6)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
def timeit(ctl, sets, reps):
    ls = [ctl.plan_batch(b, warm=w, out=o)[0] for b, w, o in sets]
    for i in range(max(3, len(ls))): ls[i % len(ls)]()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): ls[i % len(ls)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def outs(n, warm):
    o = {"grf_body": torch.empty((n, 12), dtype=torch.float64, device="cuda"), "status": torch.empty((n,), dtype=torch.int32, device="cuda"),
         "iterations": torch.empty((n,), dtype=torch.int32, device="cuda")}
    if warm: o["active_set"] = torch.empty((n,), dtype=torch.int32, device="cuda")
    return o
def make_sets(kind, n, nsets):
    sets = []
    for j in range(nsets):
        if kind == "warm":
            t0, t1 = W.config4(n, seed=W.SEEDS[4] + 0x100 * j)
            w = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
            sets.append((q.to_device(t1), w, outs(n, True)))
        else:
            sets.append((WD.config3(n, seed=W.SEEDS[5] + 0x100 * j, device=0), None, outs(n, False)))
    return sets
for kind, n in (("warm", 262144), ("cold", 262144), ("cold", 2097152), ("warm", 1048576), ("cold", 70000), ("warm", 150001)):
    if quick and n > 262144: continue
    nsets = 1 if n > 600000 else 5
    sets = make_sets(kind, n, nsets)
    base = q.BalanceController.from_params(P).set_tuning(queue=0)
    reps = 20 if n <= 262144 else 6
    t_base = timeit(base, sets, reps)
    t_base1 = timeit(base, sets[:1], reps)
    ref = {k: v.clone() for k, v in sets[0][2].items()}
    assert int((ref["status"] != 0).sum()) == 0
    print("%s n=%d: one-fill (queue=0) %.1f us rotating / %.1f us one set, mean iterations %.2f" % (kind, n, t_base, t_base1, ref["iterations"].float().mean().item()), flush=True)
    grid = [(K, th, rf) for K in ((4, 8, 16) if not quick else (8,)) for th in (16, 24, 32) for rf in (4,)] + [(8, 24, 1), (8, 24, 8), (2, 24, 4), (1, 24, 4)]
    for K, th, rf in grid:
        ctl = q.BalanceController.from_params(P).set_tuning(queue=1, queue_group=K, queue_th=th, queue_refill=rf)
        info = ctl.query_launch(n, warm=(kind == "warm"))
        for s in sets: s[2]["status"].fill_(-7)
        t = timeit(ctl, sets, reps)
        t1 = timeit(ctl, sets[:1], reps)
        o = sets[0][2]
        bad = int((o["status"] != 0).sum())
        scale = ref["grf_body"].abs().amax(dim=1, keepdim=True).clamp(min=1.0)
        err = ((o["grf_body"] - ref["grf_body"]).abs() / scale).max().item()
        print("   queue K=%2d th=%2d refill=%d (mode %d): %.1f us rotating / %.1f us one set | status!=0: %d  max rel diff vs one-fill %.1e  mean iterations %.2f" %
              (K, th, rf, info["mode"], t, t1, bad, err, o["iterations"].float().mean().item()), flush=True)
    del sets
    torch.cuda.empty_cache()
