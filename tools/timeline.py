"""Development tool (round 3): timeline of one launch, from tools/_build/libqc_timeline.so (tools/timeline.hip) - per
workgroup: entry, inputs landed, solve finished, end (100 MHz s_memrealtime) and the hardware slot.  Prints, per
microsecond of the launch: workgroups started, workgroups in their load / solve / flush phase, input bytes that landed
(as TB/s), and the phase-length statistics.  usage: python tools/timeline.py [cfg4|cfg5s|cfg3|2M] [rotate] [key=value ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from quadruped_control_amd import _lib
_lib.LIB_PATH = os.environ.get("QC_TL_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libqc_timeline.so")
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
from quadruped_control_amd import workloads_device as WD

what = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
rotate = "rotate" in sys.argv[2:]
sys.argv = [a for a in sys.argv if a not in ("rotate",)]
tune = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[2:] if "=" in a}
brief = "brief" in sys.argv[2:]
kind, n = {"cfg4": ("warm", 262144), "cfg5s": ("cold", 262144), "cfg3": ("cold", 65536), "2M": ("cold", 2097152), "1M": ("cold", 1048576),
           "16k": ("cold", 16384), "4k": ("cold", 4096), "32k": ("cold", 32768)}[what]
P = q.cheetah_params(0.6)
ctl = q.BalanceController.from_params(P).set_tuning(**tune)
nsets = ((512 << 20) // (488 * n) + 1) if rotate else 1
ls = []
keep = []
for j in range(nsets):
    if kind == "warm":
        t0, t1 = W.config4(n, seed=W.SEEDS[4] + 0x100 * j)
        w = q.BalanceController.from_params(P).control_batch(q.to_device(t0), want_active_set=True)["active_set"]
        b = q.to_device(t1)
    else:
        b, w = WD.config3(n, seed=W.SEEDS[5] + 0x100 * j, device=0), None
    l, o = ctl.plan_batch(b, warm=w, want_active_set=w is not None)
    ls.append(l); keep.append((b, w, o))
for i in range(3 * len(ls) + 1):
    ls[i % len(ls)]()
torch.cuda.synchronize()
info = ctl.query_launch(n, warm=kind == "warm")
blocks = info["blocks"]
buf = (C.c_ulonglong * (5 * blocks))()
assert ctl._lib.qc_timeline_read(buf, blocks) == 0
t = np.array(list(buf), dtype=np.uint64).reshape(blocks, 5)
hw = t[:, 4]
ts = t[:, :4].astype(np.int64)
t00 = ts[:, 0].min()
us = (ts - t00) / 100.0  # 100 MHz -> us
span = us[:, 3].max()
per_wg = info["chunk"] * (392 + (4 if kind == "warm" else 0))
print("%s n=%d %s: %d workgroups (G=%d mode %d), launch span %.1f us (first entry -> last end)" %
      (what, n, "rotating sets (cold cache)" if rotate else "one resident set", blocks, info["lanes_per_robot"], info["mode"], span))
ph = np.stack([us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2], us[:, 3] - us[:, 0]], 1)
for k, nm in enumerate(("entry -> inputs landed", "inputs landed -> solve done", "flush (transform + stores acked)", "whole workgroup")):
    v = ph[:, k]
    print("  %-34s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (nm, v.mean(), *np.percentile(v, [10, 50, 90]), v.max()))
if info["mode"] == 3:  # two-wave workgroups: do the two waves of a workgroup share a SIMD?  (HW_ID: simd_id = bits 5:4, cu_id 11:8, se 15:13; XCC in the high word)
    b1 = (C.c_ulonglong * blocks)()
    assert ctl._lib.qc_timeline_read_w1(b1, blocks) == 0
    h1 = np.array(list(b1), dtype=np.uint64)
    same_cu = ((hw >> np.uint64(8)) & np.uint64(0xFF)) == ((h1 >> np.uint64(8)) & np.uint64(0xFF))
    same_simd = same_cu & (((hw >> np.uint64(4)) & np.uint64(3)) == ((h1 >> np.uint64(4)) & np.uint64(3)))
    simd_of = lambda h: (h >> np.uint64(32)) * np.uint64(1 << 16) + ((h >> np.uint64(4)) & np.uint64(0xFFF))  # (xcc, se, cu, simd)
    occ = np.unique(np.concatenate([simd_of(hw), simd_of(h1)]), return_counts=True)[1]
    print("  two-wave workgroups: both waves on one SIMD in %.1f %% of the workgroups; SIMDs used %d, waves per used SIMD: mean %.2f max %d" %
          (100.0 * same_simd.mean(), len(occ), occ.mean(), occ.max()))
simd_key = (hw >> np.uint64(32)) * np.uint64(1 << 16) + ((hw >> np.uint64(4)) & np.uint64(0xFFF))  # (xcc, se / sh, cu, simd) of the workgroup's first wave
cu_key = (hw >> np.uint64(32)) * np.uint64(1 << 16) + ((hw >> np.uint64(8)) & np.uint64(0xFF))
first = np.argsort(us[:, 0])[:min(blocks, 2048)]  # the workgroups that started on an empty chip
oc = np.unique(simd_key[first], return_counts=True)[1]
occ_cu = np.unique(cu_key[first], return_counts=True)[1]
print("  first %d workgroups (first waves): %d SIMDs used, workgroups per used SIMD mean %.2f max %d (histogram %s); %d CUs used, per CU min %d max %d" %
      (len(first), len(oc), oc.mean(), oc.max(), np.bincount(oc).tolist(), len(occ_cu), occ_cu.min(), occ_cu.max()))
slots = len(np.unique(hw))
print("  distinct hardware wave slots used: %d; workgroups per slot: mean %.2f max %d" % (slots, blocks / slots, np.unique(hw, return_counts=True)[1].max()))
# Where the launch's wave-slot time goes: a slot is USEFUL while its workgroup solves or flushes; what is
# lost splits into (a) the first round's load burst - every slot of the first round waiting for its inputs -, (b) the loads of
# the later rounds (a slot held by a workgroup whose inputs have not landed, while the chip is otherwise busy) and (c) the
# final tail - slots that stay empty because no workgroup is left to start.  In microseconds of the whole chip
# (slot-microseconds / slots), so the three figures plus "useful" add up to the launch span.
first_round = np.argsort(us[:, 0])[:slots]  # the workgroups that started on an empty chip
is_first = np.zeros(blocks, bool); is_first[first_round] = True
load = us[:, 1] - us[:, 0]
burst = load[is_first].sum() / slots
later = load[~is_first].sum() / slots
useful = (us[:, 3] - us[:, 1]).sum() / slots
print("  slot-time split [us of the whole chip]: useful (solve + flush) %.1f | first-round load burst %.1f | loads of later rounds %.1f | "
      "empty slots (final tail + gaps between workgroups) %.1f  (sum = span %.1f)" % (useful, burst, later, span - useful - burst - later, span))
last_start = us[:, 0].max()
print("  last workgroup starts at %.1f us; from then on %.1f us of tail; mean resident workgroups over the launch %.2f per slot" %
      (last_start, span - last_start, (us[:, 3] - us[:, 0]).sum() / slots / span))
if brief:
    sys.exit(0)
edges = np.arange(0, np.ceil(span) + 1)
print("  t[us]  started  in-load  in-solve  in-flush  resident  inputs landed [TB/s]  results stored [TB/s]")
for a in edges[:-1]:
    b_ = a + 1
    started = int(((us[:, 0] >= a) & (us[:, 0] < b_)).sum())
    mid = a + 0.5
    inload = int(((us[:, 0] <= mid) & (us[:, 1] > mid)).sum())
    insolve = int(((us[:, 1] <= mid) & (us[:, 2] > mid)).sum())
    inflush = int(((us[:, 2] <= mid) & (us[:, 3] > mid)).sum())
    landed = ((us[:, 1] >= a) & (us[:, 1] < b_)).sum() * per_wg / 1e-6 / 1e12
    stored = ((us[:, 3] >= a) & (us[:, 3] < b_)).sum() * info["chunk"] * 100 / 1e-6 / 1e12
    print("  %5.0f  %7d  %7d  %8d  %8d  %8d  %20.2f  %21.2f" % (a, started, inload, insolve, inflush, inload + insolve + inflush, landed, stored))
