import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import quadruped_control_amd as q
from quadruped_control_amd import workloads as W
P = q.cheetah_params(0.6)
ctl = q.BalanceController.from_params(P)
t0, t1 = W.config4(262144)
o0 = ctl.control_batch(q.to_device(t0), want_active_set=True)
d1 = q.to_device(t1)
cold = ctl.control_batch(d1, want_iterations=True, want_active_set=True)
warm = ctl.control_batch(d1, warm=o0["active_set"], want_iterations=True, want_active_set=True)
torch.cuda.synchronize()
gc, gw = cold["grf_body"].cpu().numpy(), warm["grf_body"].cpu().numpy()
scale = np.maximum(1.0, np.abs(gc).max(axis=1, keepdims=True))
err = (np.abs(gc - gw) / scale).max(1)
idx = np.argsort(-err)[:200]
print("tol", os.environ.get("QC_TOL_D"), "iters cold mean %.3f max %d warm mean %.3f max %d" % (cold["iterations"].float().mean().item(), cold["iterations"].max().item(), warm["iterations"].float().mean().item(), warm["iterations"].max().item()), "fail", int((cold["status"]!=0).sum()), int((warm["status"]!=0).sum()))
print("n>1e-7:", (err > 1e-7).sum(), "n>1e-9:", (err > 1e-9).sum(), "max", err.max())
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/mismatch.npz", idx=idx, err=err[idx], gc=gc[idx], gw=gw[idx],
         ac=cold["active_set"].cpu().numpy()[idx], aw=warm["active_set"].cpu().numpy()[idx],
         ic=cold["iterations"].cpu().numpy()[idx], iw=warm["iterations"].cpu().numpy()[idx],
         **{k: v[idx] for k, v in t1.items()})
