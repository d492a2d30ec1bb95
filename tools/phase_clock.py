"""Development tool: where the cycles of one wave go (needs tools/_build/libqc_balance_clk.so, see phase_clock.hip).
usage: python tools/phase_clock.py [n=4096] [config=2] [lanes per robot=4] [persistent=0] [key=value ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from quadruped_control_amd import _lib
_lib.LIB_PATH = os.environ.get("QC_CLK_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libqc_balance_clk.so")
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
# markers bracket the recalculation in the persistent loop (mode 0) and in the strided one-fill loops (G = 4, modes 1 / 2);
# argv[3] = lanes per robot, argv[4] = 1 forces the persistent kernel
tune = dict(group=int(sys.argv[3]) if len(sys.argv) > 3 else 4)
if len(sys.argv) > 4 and sys.argv[4] == "1":
    tune["one_fill"] = 0  # (6x6 forms: needs a -DQC_PERSISTENT_6X6=1 build of the clock library, otherwise the launch is refused)
for kv in sys.argv[5:]:  # further tuning keys, e.g. race=0
    k, v = kv.split("=")
    tune[k] = float(v)
ctl = q.BalanceController.from_params(q.cheetah_params(0.6)).set_tuning(**tune)
lib = ctl._lib
b = q.to_device({2: W.config2, 3: W.config3}[cfg](n))
launch, out = ctl.plan_batch(b)
NAMES = ["first restock", "loop/refill/push", "coef + local M", "group reduce", "LDL^T", "tri solves", "forces+grad",
         "ratio/mult/update", "final flush", "(one marker)"]
buf = (C.c_ulonglong * 32)()
for _ in range(3):
    launch()
torch.cuda.synchronize()
lib.qc_clk_read(buf, 1)
reps = 20
for _ in range(reps):
    launch()
torch.cuda.synchronize()
lib.qc_clk_read(buf, 1)
v = np.array(list(buf), dtype=np.uint64).astype(np.int64) / reps
if os.environ.get("QC_CLK_ALL"):  # library built with -DQC_CLK_BLOCK=blockIdx.x: every workgroup adds its clocks; print the average
    v = v / ctl.query_launch(n)["blocks"]
its = v[10]
print("kernel %s, n=%d: block 0 made %.1f iterate calls per launch; %.0f cycles = %.2f us (s_memtime %.0f MHz)" %
      (ctl.kernel_name, n, its, v[12], v[11] / 100.0, v[12] / (v[11] / 100.0)))
for k, nm in enumerate(NAMES):
    per = v[k] / its if 2 <= k <= 7 or k == 9 else float("nan")
    print("  %-20s %9.0f cycles total  %8.0f per iterate" % (nm, v[k], per))
if v[26] > 0:
    print("  4-lane tail: %.1f iterate calls per launch, re-pack %.0f cycles" % (v[26], v[1]))
    for k in range(2, 8):
        print("    %-18s %9.0f cycles total  %8.0f per iterate" % (NAMES[k], v[16 + k], v[16 + k] / v[26]))
    print("    tail total %.0f cycles (+ re-pack) of %.0f" % (v[18:24].sum(), v[12]))
mk = v[9] / its
print("  sum of iterate phases per call: %.0f cycles, of which markers ~%.0f (8 x %.0f)" % ((v[2:8].sum() + v[9]) / its, 8 * mk, mk))
