"""Development tool (round 3): kernel times (HIP events) of the throughput configs for whatever library QC_LIB_PATH names -
config 3 (65 536 cold), config 4 (262 144 warm-started), config 5's shard (262 144 cold), 1 M and 2 M robots - under the
cold-cache protocol (rotating sets > 512 MiB) and as a replay of one resident set.  For A/B runs of two builds:
  for i in 1 2; do python tools/cfg_times.py; QC_LIB_PATH=tools/_build/libX.so python tools/cfg_times.py; done
usage: python tools/cfg_times.py [cfg3|cfg4|cfg5s|1M|2M ...] [tuning key=value ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
from quadruped_control_amd import workloads_device as WD
P = q.cheetah_params(0.6)
tune = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
sizes = [a for a in sys.argv[1:] if "=" not in a]
def timeit(ls, reps):
    for i in range(max(4, len(ls))): ls[i % len(ls)]()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for i in range(reps): ls[i % len(ls)]()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
def outs(n, warm):
    o = {"grf_body": torch.empty((n, 12), dtype=torch.float64, device="cuda"), "status": torch.empty((n,), dtype=torch.int32, device="cuda")}
    if warm: o["active_set"] = torch.empty((n,), dtype=torch.int32, device="cuda")
    return o
row = []
CASES = [("cfg3", "cold", 65536), ("cfg4", "warm", 262144), ("cfg5s", "cold", 262144), ("1M", "cold", 1048576), ("2M", "cold", 2097152)]
CASES += [(a, "cold" if a[0] == "N" else "warm", int(a[1:])) for a in sizes if a[0] in "NW" and a[1:].isdigit()]  # N393216 = cold, W524288 = warm-started
for name, kind, n in CASES:
    if sizes and name not in sizes: continue
    nsets = max(1, (512 << 20) // (488 * n) + 1) if n <= 262144 else 1
    ctl = q.BalanceController.from_params(P).set_tuning(**tune)
    sets = []
    for j in range(nsets):
        if kind == "warm":
            t0, t1 = W.config4(n, seed=W.SEEDS[4] + 0x100 * j)
            w = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
            sets.append((q.to_device(t1), w, outs(n, True)))
        else:
            sets.append((WD.config3(n, seed=W.SEEDS[5 if n > 65536 else 3] + 0x100 * j, device=0), None, outs(n, False)))
    ls = [ctl.plan_batch(b, warm=w, out=o)[0] for b, w, o in sets]
    reps = 40 if n <= 262144 else 8
    t_rot = timeit(ls, reps) if nsets > 1 else float("nan")
    t_one = timeit(ls[:1], reps)
    assert int((sets[0][2]["status"] != 0).sum()) == 0
    row.append("%s %.1f/%.1f" % (name, t_rot, t_one))
    del sets, ls
    torch.cuda.empty_cache()
print("%-28s us rotating/one-set: %s" % (os.path.basename(os.environ.get("QC_LIB_PATH", "in-tree")), "  ".join(row)), flush=True)
