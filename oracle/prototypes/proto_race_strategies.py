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
def clamp_keep(qp, fh, S):
    """the device's clamp_foot: clamp fh into the frusta, keeping the faces of S (point stays on them)"""
    f = fh.copy(); Sn = [s.copy() for s in S]; moved = False
    for i in range(4):
        if not qp.st[i]:
            f[3*i:3*i+3] = 0; continue
        fz = f[3*i+2]
        if fz > qp.hi: fz = qp.hi; Sn[2][i] = 1; moved = True
        elif fz < qp.lo: fz = qp.lo; Sn[2][i] = -1; moved = True
        f[3*i+2] = fz
        m = qp.mu * fz
        for a in (0, 1):
            v = f[3*i+a]
            if S[a][i] != 0:
                nv = S[a][i] * m
            else:
                if v > m: nv = m; Sn[a][i] = 1
                elif v < -m: nv = -m; Sn[a][i] = -1
                else: nv = v
            if nv != v: moved = True
            f[3*i+a] = nv
    return f, Sn, moved

def solve(qp, nclamp=1, drop="most", maxit=200, tol=1e-13):
    S = [np.zeros(4, int), np.zeros(4, int), np.zeros(4, int)]
    gs = 1 + np.max(np.abs(qp.c))
    f = None; n = 0
    for it in range(maxit):
        fh, lx, ly, lz = qp.eqp(S); n += 1
        at_fh = False
        if it < nclamp:
            fc, Sc, moved = clamp_keep(qp, fh, S)
            f, S = fc, Sc
            at_fh = not moved
            if moved: continue
        else:
            d = fh - f
            al, who = qp.ratio(f, d, S)
            if al < 1.0:
                f = f + al * d; S[who[0]][who[1]] = who[2]; continue
            f = fh; at_fh = True
        lam = np.stack([np.where(S[0] != 0, lx, np.inf), np.where(S[1] != 0, ly, np.inf), np.where(S[2] != 0, lz, np.inf)])
        neg = lam < -tol * gs
        if not neg.any(): return f, n
        if drop == "all":
            for a in range(3): S[a][neg[a]] = 0
        else:
            k = np.unravel_index(np.argmin(lam), lam.shape); S[k[0]][k[1]] = 0
    return f, n


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
