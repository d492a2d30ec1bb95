"""Development tool: where the cycles of a wave of the widened tick go (SURVEY 8f kernels; needs
tools/_build/libqc_balance_clk_all.so = tools/phase_clock.hip built with -DQC_CLK_BLOCK=blockIdx.x, every workgroup adds
its clocks and the average is printed).
usage: python tools/tick_clock.py [n=65536] [fused|full] [key=value ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from quadruped_control_amd import _lib
_lib.LIB_PATH = os.environ.get("QC_CLK_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libqc_balance_clk_all.so")
import quadruped_control_amd as q
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kind = sys.argv[2] if len(sys.argv) > 2 else "full"
tune = {}
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    tune[k] = float(v)
ctl = q.BalanceController.from_params(q.cheetah_params(0.6)).set_tuning(**tune)
lib = ctl._lib
fused = "full" if kind == "full" else (True if kind == "fused" else False)
if fused:
    hb = bench.make_tick_batch(3, n, 0, fused)
else:
    hb, _ = bench.make_batch(3, n, 0)
b = q.to_device(hb)
if fused == "full":
    b["swing_state"] = torch.from_numpy(q.new_swing_states(n).view("uint8").reshape(-1).copy()).cuda()
launch, out = ctl.plan_batch(b, want_torques=bool(fused))
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
info = ctl.query_launch(n, kin=bool(fused))
v = np.array(list(buf), dtype=np.uint64).astype(np.int64) / reps / info["blocks"]
print("tick=%s n=%d kernel %s G=%d mode=%d: per workgroup %.0f cycles = %.2f us (s_memtime %.0f MHz); solved %.4f" %
      (kind, n, ctl.kernel_name, info["lanes_per_robot"], info["mode"], v[12], v[11] / 100.0, v[12] / max(v[11] / 100.0, 1e-9),
       float((out["status"] == 0).float().mean())))
rows = [("fill (load + assembly)", v[0]), ("one-lane recalculations (%.1f)" % v[10], v[2:8].sum() + v[9]), ("re-pack", v[1]),
        ("4-lane tail recalculations (%.1f)" % v[26], v[18:24].sum() + v[25]), ("GRF stores", v[8]), ("torque pass: task lists", v[13]),
        ("torque pass: swing-leg tasks", v[14]), ("torque pass: stance-leg tasks", v[15])]
for nm, c in rows:
    print("  %-36s %9.0f cycles  %5.1f %%" % (nm, c, 100.0 * c / v[12]))
