"""Shared by the strategy prototypes: a primal active-set solve with a configurable number of clamp steps and drop rule."""
import numpy as np
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




def solve_policy(qp, clamp_at, drop_at, maxit=200, tol=1e-13):
    """solve() with the step kind and the drop rule as functions of the recalculation number (0-based):
    clamp_at(it) -> bool (clamp step instead of a ratio-test step), drop_at(it) -> "most" | "all"."""
    S = [np.zeros(4, int), np.zeros(4, int), np.zeros(4, int)]
    gs = 1 + np.max(np.abs(qp.c))
    f = None; n = 0
    for it in range(maxit):
        fh, lx, ly, lz = qp.eqp(S); n += 1
        if clamp_at(it) or f is None:
            fc, Sc, moved = clamp_keep(qp, fh, S)
            f, S = fc, Sc
            if moved: continue
        else:
            d = fh - f
            al, who = qp.ratio(f, d, S)
            if al < 1.0:
                f = f + al * d; S[who[0]][who[1]] = who[2]; continue
            f = fh
        lam = np.stack([np.where(S[0] != 0, lx, np.inf), np.where(S[1] != 0, ly, np.inf), np.where(S[2] != 0, lz, np.inf)])
        neg = lam < -tol * gs
        if not neg.any(): return f, n
        if drop_at(it) == "all":
            for a in range(3): S[a][neg[a]] = 0
        else:
            k = np.unravel_index(np.argmin(lam), lam.shape); S[k[0]][k[1]] = 0
    return f, n
