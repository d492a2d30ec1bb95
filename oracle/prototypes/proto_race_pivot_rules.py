"""Round-2 design prototype (development tool, uses the oracle): iteration statistics of an active-set variant on config 2. See DESIGN.md 7.1."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import numpy_restatement as R
from quadruped_control_amd import workloads as W
from oracle.prototypes.prototype_solver import assemble_batch
from oracle.prototypes.prototype_as import QP

def solve(qp, drop="most", addrule="first", maxit=200, tol=1e-9, clampmode="all"):
    S0 = [np.zeros(4, int), np.zeros(4, int), np.zeros(4, int)]
    fu, *_ = qp.eqp(S0); n = 1
    f, S = qp.clamp(fu)
    if np.allclose(f, fu, atol=1e-12): return f, n
    if clampmode == "zonly":   # keep only the z faces of the clamp, x/y faces free (point stays feasible)
        S = [np.zeros(4, int), np.zeros(4, int), S[2]]
    gs = 1 + np.max(np.abs(qp.c))
    for it in range(maxit):
        fh, lx, ly, lz = qp.eqp(S); n += 1
        d = fh - f
        al, who = qp.ratio(f, d, S)
        if al < 1.0:
            f = f + al * d
            S[who[0]][who[1]] = who[2]
            continue
        f = fh
        lam = np.stack([np.where(S[0] != 0, lx, np.inf), np.where(S[1] != 0, ly, np.inf), np.where(S[2] != 0, lz, np.inf)])
        neg = lam < -tol * gs
        if not neg.any(): return f, n
        if drop == "most": k = np.unravel_index(np.argmin(lam), lam.shape)
        elif drop == "least":  # least negative
            l2 = np.where(neg, lam, -np.inf); k = np.unravel_index(np.argmax(l2), lam.shape)
        elif drop == "zfirst":  # prefer z faces
            l2 = lam.copy(); l2[2][neg[2]] -= 1e12; k = np.unravel_index(np.argmin(l2), lam.shape)
        elif drop == "xyfirst":
            l2 = lam.copy(); l2[0][neg[0]] -= 1e12; l2[1][neg[1]] -= 1e12; k = np.unravel_index(np.argmin(l2), lam.shape)
        elif drop == "all":
            for a in range(3): S[a][neg[a]] = 0
            continue
        S[k[0]][k[1]] = 0
    return f, n

P = R.cheetah_params(mu=0.6)
n = 4096
B = W.config2(n)
Q, c = assemble_batch(P, B)
variants = [("most", "all"), ("least", "all"), ("zfirst", "all"), ("xyfirst", "all"), ("all", "all"), ("most", "zonly")]
res = {}
for v in variants:
    ks = []
    for i in range(n):
        qp = QP(Q[i], c[i], B["stance"][i], P["mu"], P["fzmin"], P["fzmax"])
        f, k = solve(qp, drop=v[0], clampmode=v[1])
        ks.append(k)
    res[v] = np.array(ks)
    print(v, "mean %.2f max %d  p99.9 %d" % (res[v].mean(), res[v].max(), np.sort(res[v])[-5]), flush=True)
base = res[("most", "all")]
import itertools
for r in (2, 3, 4):
    for combo in itertools.combinations(variants, r):
        if ("most", "all") not in combo: continue
        m = np.min(np.stack([res[v] for v in combo]), axis=0)
        print(combo, "race: mean %.2f max %d; per-16 max mean %.2f" % (m.mean(), m.max(), m.reshape(-1, 16).max(1).mean()))
