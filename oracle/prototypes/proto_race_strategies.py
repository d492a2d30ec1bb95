"""Round-2 design prototype (development tool, uses the oracle): which pivoting strategies to race per robot in the mode-2 kernel.
Strategy = (clamp steps before the first ratio test, drop rule); prints the batch maximum of recalculations for every race of 2 and 4
over eight config-2 batches.  See DESIGN.md 7.1."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, itertools
from oracle import numpy_restatement as R
from quadruped_control_amd import workloads as W
from oracle.prototypes.prototype_solver import assemble_batch
from oracle.prototypes.prototype_as import QP
from oracle.prototypes.proto_race_strategies_lib import solve

P = R.cheetah_params(mu=0.6)
n = 4096
VAR = [(1, "most"), (1, "all"), (2, "most"), (2, "all"), (3, "most"), (3, "all")]
allres = []
seeds = [0x5EED0002 + 0x100 * k for k in range(8)]
for seed in seeds:
    B = W.config2(n, seed=seed)
    Q, c = assemble_batch(P, B)
    res = {v: [] for v in VAR}
    for i in range(n):
        qp = QP(Q[i], c[i], B["stance"][i], P["mu"], P["fzmin"], P["fzmax"])
        for v in VAR: res[v].append(solve(qp, *v)[1])
    allres.append({k: np.array(v) for k, v in res.items()})
    print("seed %x done:" % seed, {k: int(np.max(v)) for k, v in res.items()}, flush=True)
for r in (2, 4):
    rows = []
    for combo in itertools.combinations(VAR, r):
        if (1, "most") not in combo: continue
        mx = [int(np.min(np.stack([res[k] for k in combo]), axis=0).max()) for res in allres]
        rows.append((np.mean(mx), max(mx), combo, mx))
    rows.sort(key=lambda t: (t[0], t[1]))
    print("races of", r)
    for row in rows[:6]: print("   mean max %.2f worst %d %s %s" % row)
print("plain:", [int(res[(1, "most")].max()) for res in allres])
