"""proto_race_strategies.py with a wider pool - clamp steps 1 ... 5 x drop rule {most, all} - and no strategy forced into the race:
is the product's set {(1, most), (1, all), (2, all), (3, most)} still the one to run?  (config 2, eight 4 096-robot batches)"""
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
VAR = [(k, d) for k in (1, 2, 3, 4, 5) for d in ("most", "all")]
allres = []
for seed in [0x5EED0002 + 0x100 * k for k in range(8)]:
    B = W.config2(n, seed=seed)
    Q, c = assemble_batch(P, B)
    res = {v: np.zeros(n, int) for v in VAR}
    for i in range(n):
        qp = QP(Q[i], c[i], B["stance"][i], P["mu"], P["fzmin"], P["fzmax"])
        for v in VAR: res[v][i] = solve(qp, *v)[1]
    allres.append(res)
    print("seed %x:" % seed, {k: int(v.max()) for k, v in res.items()}, flush=True)
for r in (1, 2, 4):
    rows = []
    for combo in itertools.combinations(VAR, r):
        mx = [int(np.min(np.stack([res[k] for k in combo]), axis=0).max()) for res in allres]
        rows.append((np.mean(mx), max(mx), combo, mx))
    rows.sort(key=lambda t: (t[0], t[1]))
    print("races of", r)
    for row in rows[:8]: print("   mean max %.2f worst %d %s %s" % row)
prod = ((1, "most"), (1, "all"), (2, "all"), (3, "most"))
mx = [int(np.min(np.stack([res[k] for k in prod]), axis=0).max()) for res in allres]
print("product's set: mean max %.2f %s" % (np.mean(mx), mx))
