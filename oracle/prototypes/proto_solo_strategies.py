"""Round-2 design prototype (development tool, uses the oracle): would ONE other pivoting strategy beat the classic one where no
SIMD is idle (one lane per robot, 64 robots per wave)?  Per strategy: mean recalculations, mean over waves of the slowest of 64
robots, batch maximum - config 2 and config 3 inputs.  See DESIGN.md 7.1."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from oracle import numpy_restatement as R
from quadruped_control_amd import workloads as W
from oracle.prototypes.prototype_solver import assemble_batch
from oracle.prototypes.prototype_as import QP
from oracle.prototypes.proto_race_strategies_lib import solve

P = R.cheetah_params(mu=0.6)
VAR = [(1, "most"), (1, "all"), (2, "most"), (2, "all"), (3, "most"), (3, "all")]
for name, gen, n in (("config3", W.config3, 8192), ("config2", W.config2, 4096)):
    B = gen(n)
    Q, c = assemble_batch(P, B)
    res = {v: np.zeros(n, int) for v in VAR}
    for i in range(n):
        qp = QP(Q[i], c[i], B["stance"][i], P["mu"], P["fzmin"], P["fzmax"])
        for v in VAR: res[v][i] = solve(qp, *v)[1]
    for v in VAR:
        it = res[v]
        print("%s %-12s mean %.2f  slowest of 64 (mean over waves) %.2f  slowest of 16 %.2f  max %d" %
              (name, v, it.mean(), it.reshape(-1, 64).max(1).mean(), it.reshape(-1, 16).max(1).mean(), it.max()), flush=True)
