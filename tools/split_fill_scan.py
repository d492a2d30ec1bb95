"""Round 5: the fill / assemble role at three or more waves per SIMD.

gfx950 allocates registers per KERNEL (one .amdhsa_next_free_vgpr per kernel descriptor; no s_alloc_vgpr before gfx12), so two
roles inside one launch cannot have different register budgets: the 244-register solve pins every wave of its kernel at two per
SIMD.  What can be measured is the role as a kernel of its own (tools/experiments/split_fill_role.patch, built with
-DQC_EXPERIMENT_SPLIT_FILL into tools/_build/libqc_split.so):

  split_fill = 2   the fill kernel ALONE, storing like the batch-load probe: what occupancy does to the load -> assemble -> store phase
                   (124 registers = 4 waves per SIMD by itself; fill_minw = 6 / 8 caps it at 80 / 64 registers, with scratch)
  split_fill = 1   fill kernel (parks 28 doubles per robot, [plane][n]) + the one-lane solve kernel fed from those records
                   (28 coalesced 512-byte loads per wave instead of 48 row-per-lane gathers + the assembly)

against the product (one launch) on config 5's shard (262 144 cold), config 4 (262 144 warm-started) and 2 M robots, cold-cache
rotation.  usage: QC_LIB_PATH=tools/_build/libqc_split.so python tools/split_fill_scan.py [cfg5s cfg4 2M]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
from quadruped_control_amd import workloads_device as WD

P = q.cheetah_params(0.6)
want = [a for a in sys.argv[1:]] or ["cfg5s", "cfg4", "2M"]


def timeit(ls, reps):
    for i in range(max(6, len(ls))): ls[i % len(ls)]()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(4):
        e0.record()
        for i in range(reps): ls[i % len(ls)]()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]


def outs(n, warm):
    o = {"grf_body": torch.zeros((n, 12), dtype=torch.float64, device="cuda"), "status": torch.zeros((n,), dtype=torch.int32, device="cuda")}
    if warm: o["active_set"] = torch.zeros((n,), dtype=torch.int32, device="cuda")
    return o


VARIANTS = [("product", {}), ("one-fill kernel (pair = 0)", dict(pair=0)),
            ("split: fill (4 waves/SIMD) + record-fed solve", dict(pair=0, split_fill=1, fill_minw=1)),
            ("split: fill capped at 6 waves/SIMD + solve", dict(pair=0, split_fill=1, fill_minw=6)),
            ("product probe (same kernel, no iterations: 2 waves/SIMD)", dict(pair=0, probe_batch_load=1)),
            ("fill kernel alone, 4 waves/SIMD (124 VGPRs)", dict(pair=0, split_fill=2, fill_minw=1)),
            ("fill kernel alone, capped at 2 waves/SIMD", dict(pair=0, split_fill=2, fill_minw=2)),
            ("fill kernel alone, 6 waves/SIMD (80 VGPRs + scratch)", dict(pair=0, split_fill=2, fill_minw=6)),
            ("fill kernel alone, 8 waves/SIMD (64 VGPRs + scratch)", dict(pair=0, split_fill=2, fill_minw=8))]
for name, kind, n in (("cfg5s", "cold", 262144), ("cfg4", "warm", 262144), ("2M", "cold", 2097152)):
    if name not in want: continue
    nsets = max(1, (512 << 20) // (488 * n) + 1) if n <= 262144 else 2
    sets = []
    for j in range(nsets):
        if kind == "warm":
            t0, t1 = W.config4(n, seed=W.SEEDS[4] + 0x100 * j)
            w = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
            sets.append((q.to_device(t1), w, outs(n, True)))
        else:
            sets.append((WD.config3(n, seed=W.SEEDS[5] + 0x100 * j, device=0), None, outs(n, False)))
    ref = None
    print(f"== {name}: {n} robots, {kind}, {nsets} rotating sets", flush=True)
    for label, tune in VARIANTS:
        ctl = q.BalanceController.from_params(P)
        try:
            ctl.set_tuning(**tune)
        except ValueError as e:
            print(f"   {label:62s} -> {str(e)[:80]}"); continue
        ls = [ctl.plan_batch(b, warm=w, out=o)[0] for b, w, o in sets]
        best, med = timeit(ls, 40 if n <= 262144 else 10)
        g = sets[0][2]["grf_body"]
        solved = int((sets[0][2]["status"] == 0).sum())
        note = ""
        if tune.get("probe_batch_load") or tune.get("split_fill") == 2:
            note = f"   [{488 * n / best / 1e3:.0f} GB/s = {488 * n / best / 1e3 / 80:.1f} % of HBM peak]"
        else:
            assert solved == n, (label, solved)
            if ref is None: ref = g.clone()
            note = f"   max |GRF - product| = {float((g - ref).abs().max()):.1e}"
        print(f"   {label:62s} best {best:8.1f} us  median {med:8.1f} us{note}", flush=True)
        del ls, ctl
    del sets
    torch.cuda.empty_cache()
