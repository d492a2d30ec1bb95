"""Round-2 design prototype (development tool, uses the oracle): would racing strategies in the 4-lane TAIL of a throughput
wave (robots still running after K recalculations fork into variants; the first to finish wins) shorten the slowest
robot's chain on config-3 inputs?  Base strategy = the product's (five clamp steps, drop the most negative multiplier)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from oracle import numpy_restatement as R
from quadruped_control_amd import workloads as W
from oracle.prototypes.prototype_solver import assemble_batch
from oracle.prototypes.prototype_as import QP
from oracle.prototypes.proto_race_strategies_lib import solve_policy

P = R.cheetah_params(mu=0.6)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
B = W.config3(n)
Q, c = assemble_batch(P, B)
base = np.zeros(n, int)
qps = [QP(Q[i], c[i], B["stance"][i], P["mu"], P["fzmin"], P["fzmax"]) for i in range(n)]
for i in range(n):
    base[i] = solve_policy(qps[i], lambda it: it < 5, lambda it: "most")[1]
print("base: mean %.2f max %d, per-64 max mean %.2f, hist %s" % (base.mean(), base.max(), base.reshape(-1, 64).max(1).mean(), np.bincount(base)), flush=True)
for K in (6, 8, 10):
    late = np.flatnonzero(base > K)
    VAR = {"all": (lambda it, K=K: it < 5, lambda it, K=K: "all" if it >= K else "most"),
           "clamp1+most": (lambda it, K=K: it < 5 or it == K, lambda it: "most"),
           "clamp1+all": (lambda it, K=K: it < 5 or it == K, lambda it, K=K: "all" if it >= K else "most"),
           "clamp2+most": (lambda it, K=K: it < 5 or K <= it < K + 2, lambda it: "most")}
    res = {k: base.copy() for k in VAR}
    for i in late:
        for k, (ca, da) in VAR.items():
            res[k][i] = solve_policy(qps[i], ca, da)[1]
    best4 = np.minimum.reduce([base] + [res[k] for k in ("all", "clamp1+most", "clamp1+all")])
    best2 = np.minimum(base, res["all"])
    print("fork after %d (%d robots, %.1f %%):" % (K, len(late), 100.0 * len(late) / n), {k: (round(float(v[late].mean()), 2), int(v.max())) for k, v in res.items()},
          "| base on those: mean %.2f" % base[late].mean(), flush=True)
    print("   race of {most, all}: max %d per-64 max mean %.2f | race of 4: max %d per-64 max mean %.2f (base %d / %.2f)" %
          (best2.max(), best2.reshape(-1, 64).max(1).mean(), best4.max(), best4.reshape(-1, 64).max(1).mean(), base.max(), base.reshape(-1, 64).max(1).mean()), flush=True)
