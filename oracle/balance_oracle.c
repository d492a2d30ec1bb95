/* TEST INFRASTRUCTURE - see balance_oracle.h for scope, provenance and the
 * "parity unpinned" statement.  Plain C99, FP64, no third-party code.
 * BC.cpp / BC.hpp citations are relative to /root/reference/quadruped_controller. */
#include "balance_oracle.h"

#include <math.h>
#include <string.h>

#define NV 12 /* BC.hpp:157 num_variables_qp_   */
#define NC 20 /* BC.hpp:158 num_constraints_qp_ */
#define NE 6  /* BC.hpp:156 num_equations_qp_   */

/* ---------------------------------------------------------------- helpers */
static void mat3_mul(const double* a, const double* b, double* c) { /* c = a b */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * k + j];
      c[3 * i + j] = s;
    }
}
static void mat3_mul_bt(const double* a, const double* b, double* c) { /* c = a b' */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * j + k];
      c[3 * i + j] = s;
    }
}
static void mat3_vec(const double* a, const double* v, double* o) {
  for (int i = 0; i < 3; i++) o[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
}

/* rigid3d.cpp:61-74 */
static void skew_symmetric(const double* v, double* m) {
  m[0] = 0.0;   m[1] = -v[2]; m[2] = v[1];
  m[3] = v[2];  m[4] = 0.0;   m[5] = -v[0];
  m[6] = -v[1]; m[7] = v[0];  m[8] = 0.0;
}

/* rigid3d.cpp:177-179 + 198-203: Rotation3d(mat).angleAxisTotal() ->
 * drake::math::RotationMatrix::ToAngleAxis() -> Eigen::AngleAxisd(matrix):
 * matrix -> quaternion -> angle/axis (published Eigen 3.3 algorithm restated;
 * Drake v0.26.0 per README.md:101, not vendored). */
void oracle_angle_axis_total(const double* m, double* out) {
  double q[4]; /* x y z w */
  double t = m[0] + m[4] + m[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
    q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
    q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
  }
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  if (n != 0.0) {
    double angle = 2.0 * atan2(n, fabs(q[3]));
    if (q[3] < 0.0) n = -n;
    out[0] = q[0] / n * angle;
    out[1] = q[1] / n * angle;
    out[2] = q[2] / n * angle;
  } else { /* angle 0, axis (1,0,0) */
    out[0] = out[1] = out[2] = 0.0;
  }
}

/* ------------------------------------------------------------- assembly */
void oracle_assemble(const oracle_params* P, const double* Rwb, const double* Rwb_d, const double* x,
                     const double* xdot, const double* w, const double* x_d, const double* xdot_d,
                     const double* w_d, const double* feet, const unsigned char* stance, double* H,
                     double* g, double* C, double* lb, double* ub, double* A_out, double* b_out) {
  /* frictionConeConstraint(), BC.cpp:274-292 (copied into qp_C_ each tick, :327) */
  const double mu = P->mu;
  const double Cf[5][3] = {{1.0, 0.0, -mu}, {0.0, 1.0, -mu}, {0.0, 1.0, mu}, {1.0, 0.0, mu}, {0.0, 0.0, 1.0}};
  memset(C, 0, sizeof(double) * NC * NV);
  for (int i = 0; i < 4; i++)
    for (int r = 0; r < 5; r++)
      for (int c = 0; c < 3; c++) C[(5 * i + r) * NV + 3 * i + c] = Cf[r][c];

  /* frictionConeBounds(), BC.cpp:294-330 */
  const double upper = 1000000.0, lower = -1000000.0;
  const double lbf[5] = {lower, lower, 0.0, 0.0, P->fzmin};
  const double ubf[5] = {0.0, 0.0, upper, upper, P->fzmax};
  for (int i = 0; i < 4; i++)
    for (int r = 0; r < 5; r++) {
      lb[5 * i + r] = stance[i] ? lbf[r] : 0.0;
      ub[5 * i + r] = stance[i] ? ubf[r] : 0.0;
    }

  /* PD law, BC.cpp:126-129 */
  double xddot_d[3], wdot_d[3];
  for (int k = 0; k < 3; k++) xddot_d[k] = P->kp_p[k] * (x_d[k] - x[k]) + P->kd_p[k] * (xdot_d[k] - xdot[k]);
  xddot_d[0] += P->kff[0] * xdot_d[0];
  xddot_d[1] += P->kff[1] * xdot_d[1];
  xddot_d[2] += P->kff[2] * P->mass * 9.81;

  /* rotation error + angular PD, BC.cpp:133-139 */
  double R_error[9], aa[3];
  mat3_mul_bt(Rwb_d, Rwb, R_error);
  oracle_angle_axis_total(R_error, aa);
  for (int k = 0; k < 3; k++) wdot_d[k] = P->kp_w[k] * aa[k] + P->kd_w[k] * (w_d[k] - w[k]);
  wdot_d[0] += P->kff[3] * w_d[0];
  wdot_d[1] += P->kff[4] * w_d[1];
  wdot_d[1] += P->kff[5] * w_d[2]; /* sic: index 1, BC.cpp:139 */

  /* dynamics(), BC.cpp:237-272 (its `x` argument is unused) */
  double A[NE * NV], b[NE];
  memset(A, 0, sizeof(A));
  for (int i = 0; i < 4; i++) {
    double r[3], sk[9];
    mat3_vec(Rwb, feet + 3 * i, r); /* :244-248 */
    skew_symmetric(r, sk);
    for (int k = 0; k < 3; k++) {
      A[k * NV + 3 * i + k] = 1.0; /* :254-257 */
      for (int c = 0; c < 3; c++) A[(3 + k) * NV + 3 * i + c] = sk[3 * k + c]; /* :259-262 */
    }
  }
  double RI[9], Iw[9];
  mat3_mul(Rwb, P->Ib, RI);
  mat3_mul_bt(RI, Rwb, Iw); /* :251 */
  const double grav[3] = {0.0, 0.0, -9.81}; /* BC.cpp:76 */
  for (int k = 0; k < 3; k++) b[k] = P->mass * (xddot_d[k] + grav[k]); /* :265 */
  double Iwd[3], Iww[3];
  mat3_vec(Iw, wdot_d, Iwd);
  mat3_vec(Iw, w_d, Iww);
  b[3] = Iwd[0] + (w_d[1] * Iww[2] - w_d[2] * Iww[1]); /* :269 */
  b[4] = Iwd[1] + (w_d[2] * Iww[0] - w_d[0] * Iww[2]);
  b[5] = Iwd[2] + (w_d[0] * Iww[1] - w_d[1] * Iww[0]);

  /* Q = 2(A'SA + W), c = -2 A'S b, BC.cpp:152-153 */
  double SA[NE * NV], Sb[NE];
  for (int i = 0; i < NE; i++) {
    for (int j = 0; j < NV; j++) {
      double s = 0.0;
      for (int k = 0; k < NE; k++) s += P->S[i * NE + k] * A[k * NV + j];
      SA[i * NV + j] = s;
    }
    double s = 0.0;
    for (int k = 0; k < NE; k++) s += P->S[i * NE + k] * b[k];
    Sb[i] = s;
  }
  for (int i = 0; i < NV; i++) {
    for (int j = 0; j < NV; j++) {
      double s = 0.0;
      for (int k = 0; k < NE; k++) s += A[k * NV + i] * SA[k * NV + j];
      H[i * NV + j] = 2.0 * (s + P->W[i * NV + j]);
    }
    double s = 0.0;
    for (int k = 0; k < NE; k++) s += A[k * NV + i] * Sb[k];
    g[i] = -2.0 * s;
  }
  if (A_out) memcpy(A_out, A, sizeof(A));
  if (b_out) memcpy(b_out, b, sizeof(b));
}

/* --------------------------------------------- dense primal active set QP */
#define KMAX (NV + NV)

/* Gaussian elimination with partial pivoting, n<=KMAX, in place. 0 on success. */
static int lin_solve(double* M, double* rhs, int n) {
  for (int c = 0; c < n; c++) {
    int p = c;
    double best = fabs(M[c * KMAX + c]);
    for (int r = c + 1; r < n; r++)
      if (fabs(M[r * KMAX + c]) > best) { best = fabs(M[r * KMAX + c]); p = r; }
    if (best < 1e-300) return 1;
    if (p != c) {
      for (int k = 0; k < n; k++) { double t = M[c * KMAX + k]; M[c * KMAX + k] = M[p * KMAX + k]; M[p * KMAX + k] = t; }
      double t = rhs[c]; rhs[c] = rhs[p]; rhs[p] = t;
    }
    for (int r = c + 1; r < n; r++) {
      double m = M[r * KMAX + c] / M[c * KMAX + c];
      if (m == 0.0) continue;
      for (int k = c; k < n; k++) M[r * KMAX + k] -= m * M[c * KMAX + k];
      rhs[r] -= m * rhs[c];
    }
  }
  for (int r = n - 1; r >= 0; r--) {
    double s = rhs[r];
    for (int k = r + 1; k < n; k++) s -= M[r * KMAX + k] * rhs[k];
    rhs[r] = s / M[r * KMAX + r];
  }
  return 0;
}

/* The same elimination in long double (x87 extended precision here): used ONCE per QP, to recompute the accepted point.
 * With W = w I and w <= 2e-7 the KKT matrix has a condition number of 1e9 and more, and the double-precision solve
 * leaves up to 3e-4 N on the weakly determined force components - the NNLS restatement (oracle/numpy_restatement.py)
 * put the GPU path's 6x6 form closer to the minimiser than this oracle on such problems (profiles/r02_stress_extended.log).
 * Decisions (ratio tests, multiplier signs) stay in double; only the reported point is refined. */
static int g_refine = 1;
/* 0: skip the long-double recomputation (bench.py times the port that way: the refinement is checker accuracy, not part of
 * the reference path, and x87 arithmetic would make the CPU baseline 35 % slower than the algorithm is) */
void oracle_set_refine(int on) { g_refine = on; }
static int lin_solve_ld(const double* M0, const double* rhs0, int n, double* x) {
  long double M[KMAX * KMAX], rhs[KMAX];
  for (int r = 0; r < n; r++) {
    for (int k = 0; k < n; k++) M[r * KMAX + k] = M0[r * KMAX + k];
    rhs[r] = rhs0[r];
  }
  for (int c = 0; c < n; c++) {
    int p = c;
    long double best = fabsl(M[c * KMAX + c]);
    for (int r = c + 1; r < n; r++)
      if (fabsl(M[r * KMAX + c]) > best) { best = fabsl(M[r * KMAX + c]); p = r; }
    if (best < 1e-300L) return 1;
    if (p != c) {
      for (int k = 0; k < n; k++) { long double t = M[c * KMAX + k]; M[c * KMAX + k] = M[p * KMAX + k]; M[p * KMAX + k] = t; }
      long double t = rhs[c]; rhs[c] = rhs[p]; rhs[p] = t;
    }
    for (int r = c + 1; r < n; r++) {
      long double m = M[r * KMAX + c] / M[c * KMAX + c];
      if (m == 0.0L) continue;
      for (int k = c; k < n; k++) M[r * KMAX + k] -= m * M[c * KMAX + k];
      rhs[r] -= m * rhs[c];
    }
  }
  for (int r = n - 1; r >= 0; r--) {
    long double s = rhs[r];
    for (int k = r + 1; k < n; k++) s -= M[r * KMAX + k] * rhs[k];
    rhs[r] = s / M[r * KMAX + r];
  }
  for (int r = 0; r < n; r++) x[r] = (double)rhs[r];
  return 0;
}

/* is row `a` linearly independent of the rows listed in idx[0..m)? */
static int independent(const double* C, const int* idx, int m, const double* a) {
  double basis[NV][NV];
  int nb = 0;
  for (int t = 0; t <= m; t++) {
    double v[NV];
    const double* src = (t < m) ? C + idx[t] * NV : a;
    double n0 = 0.0;
    for (int k = 0; k < NV; k++) { v[k] = src[k]; n0 += v[k] * v[k]; }
    for (int pass = 0; pass < 2; pass++)
      for (int q = 0; q < nb; q++) {
        double d = 0.0;
        for (int k = 0; k < NV; k++) d += basis[q][k] * v[k];
        for (int k = 0; k < NV; k++) v[k] -= d * basis[q][k];
      }
    double n1 = 0.0;
    for (int k = 0; k < NV; k++) n1 += v[k] * v[k];
    int indep = n1 > 1e-20 * (n0 > 0 ? n0 : 1.0);
    if (t == m) return indep;
    if (indep && nb < NV) {
      double s = 1.0 / sqrt(n1);
      for (int k = 0; k < NV; k++) basis[nb][k] = v[k] * s;
      nb++;
    }
  }
  return 0;
}

/* KKT system of the equality-constrained subproblem on the working set idx[0..m) */
static void build_kkt(const double* H, const double* g, const double* C, const double* lb, const double* ub, const int* ws, const int* idx, int m,
                      double* M, double* rhs) {
  memset(M, 0, sizeof(double) * KMAX * KMAX);
  for (int i = 0; i < NV; i++) {
    for (int j = 0; j < NV; j++) M[i * KMAX + j] = H[i * NV + j];
    rhs[i] = -g[i];
  }
  for (int t = 0; t < m; t++) {
    int r = idx[t];
    for (int k = 0; k < NV; k++) { M[(NV + t) * KMAX + k] = C[r * NV + k]; M[k * KMAX + NV + t] = C[r * NV + k]; }
    rhs[NV + t] = (ws[r] == -1) ? lb[r] : ub[r];
  }
}

int oracle_qp_solve(const double* H, const double* g, const double* C, const double* lb,
                    const double* ub, int max_iter, double* f, double* lam_out, int* iters_out) {
  /* Feasible start: the reference's bounds always admit f_i = (0,0,lb of the
   * fz row) per foot (BC.cpp:300-301: lbf[4]=fzmin; swing rows are all 0). */
  for (int k = 0; k < NV; k++) f[k] = 0.0;
  /* fz = fzmin when that is a regular point of the foot's feasible set;
   * with fzmin = 0 it would be the cone apex (five rows active on three
   * variables, a degenerate vertex where the textbook method can cycle), so
   * start from the middle of the fz range instead. */
  for (int i = 0; i < 4; i++) f[3 * i + 2] = (lb[5 * i + 4] > 0.0 || lb[5 * i + 4] == ub[5 * i + 4]) ? lb[5 * i + 4] : 0.5 * (lb[5 * i + 4] + ub[5 * i + 4]);
  int ws[NC];      /* 0 free, +1 at ub, -1 at lb, 2 equality, 3 redundant equality */
  int idx[NC], m = 0;
  const double ftol = 1e-9;
  for (int r = 0; r < NC; r++) {
    double cf = 0.0;
    for (int k = 0; k < NV; k++) cf += C[r * NV + k] * f[k];
    if (cf > ub[r] + ftol * (1 + fabs(ub[r])) || cf < lb[r] - ftol * (1 + fabs(lb[r]))) return ORACLE_INFEASIBLE;
    ws[r] = 0;
  }
  for (int pass = 0; pass < 2; pass++) /* equalities first, then rows active at f */
    for (int r = 0; r < NC; r++) {
      if (ws[r]) continue;
      double cf = 0.0;
      for (int k = 0; k < NV; k++) cf += C[r * NV + k] * f[k];
      int kind = 0;
      if (pass == 0 && lb[r] == ub[r]) kind = 2;
      if (pass == 1 && lb[r] != ub[r]) {
        if (fabs(cf - lb[r]) <= 1e-12 * (1 + fabs(lb[r]))) kind = -1;
        else if (fabs(cf - ub[r]) <= 1e-12 * (1 + fabs(ub[r]))) kind = 1;
      }
      if (kind && m < NV && independent(C, idx, m, C + r * NV)) { ws[r] = kind; idx[m++] = r; }
      else if (kind == 2) ws[r] = 3; /* equality row implied by earlier equality rows: satisfied
                                        on the whole working manifold, never blocks, no multiplier */
    }

  double lam[NC];
  int it;
  int stalled = 0; /* consecutive working-set changes without movement; > 12 switches to Bland's
                      least-index rule, which cannot cycle at a degenerate vertex (cone apex) */
  for (it = 0; it < max_iter; it++) {
    const int bland = stalled > 12;
    /* KKT system of the equality-constrained subproblem */
    double M[KMAX * KMAX], rhs[KMAX];
    int n = NV + m;
    build_kkt(H, g, C, lb, ub, ws, idx, m, M, rhs);
    if (lin_solve(M, rhs, n)) return ORACLE_NOT_PD;
    double d[NV];
    for (int k = 0; k < NV; k++) d[k] = rhs[k] - f[k];
    /* ratio test over rows outside the working set.  A row that is linearly
     * dependent on the working set cannot really block (its c.d is rounding
     * noise, e.g. the fz>=0 row once three cone rows pin a foot to the apex):
     * it is skipped and the test repeated. */
    int skip[NC] = {0};
    double alpha;
    int block, bside;
    for (;;) {
      alpha = 1.0; block = -1; bside = 0;
      for (int r = 0; r < NC; r++) {
        if (ws[r] || skip[r]) continue;
        double cd = 0.0, cf = 0.0;
        for (int k = 0; k < NV; k++) { cd += C[r * NV + k] * d[k]; cf += C[r * NV + k] * f[k]; }
        if (cd > 1e-13) {
          double a = (ub[r] - cf) / cd;
          if (a < 0) a = 0;
          if (a < alpha && !(bland && block >= 0 && alpha <= 1e-12)) { alpha = a; block = r; bside = 1; }
        } else if (cd < -1e-13) {
          double a = (lb[r] - cf) / cd;
          if (a < 0) a = 0;
          if (a < alpha && !(bland && block >= 0 && alpha <= 1e-12)) { alpha = a; block = r; bside = -1; }
        }
      }
      if (block < 0 || (m < NV && independent(C, idx, m, C + block * NV))) break;
      skip[block] = 1;
    }
    if (block >= 0) {
      if (alpha <= 1e-12) stalled++; else stalled = 0;
      for (int k = 0; k < NV; k++) f[k] += alpha * d[k];
      ws[block] = bside; idx[m++] = block;
      continue;
    }
    for (int k = 0; k < NV; k++) f[k] = rhs[k];
    /* multipliers: H f + g + sum lam_t a_t = 0; need lam>=0 at ub, <=0 at lb */
    int worst = -1;
    double wv = 0.0, gs = 1.0;
    for (int k = 0; k < NV; k++) if (fabs(g[k]) > gs) gs = fabs(g[k]);
    for (int r = 0; r < NC; r++) lam[r] = 0.0;
    for (int t = 0; t < m; t++) {
      int r = idx[t];
      lam[r] = rhs[NV + t];
      double viol = (ws[r] == 1) ? -lam[r] : (ws[r] == -1 ? lam[r] : 0.0);
      /* W ~ 1e-5 (and smaller) makes the primal very sensitive to a wrongly kept weakly-active row: the objective
       * curves only with 2w along it, so a multiplier of -t moves a force by t/(2w).  -2e-7 gave 3e-4 relative in
       * testing, 1e-13*|c| still 7e-3 N on w = 2e-7 problems: the threshold sits just above the rounding noise. */
      if (viol > 1e-15 * gs && (bland ? (worst < 0 || idx[t] < idx[worst]) : viol > wv)) { wv = viol; worst = t; }
    }
    if (worst < 0) {
      if (g_refine) {  /* (lin_solve worked in place: the system of the accepted working set is built once more) */
        double M0[KMAX * KMAX], rhs0[KMAX], xr[KMAX];
        build_kkt(H, g, C, lb, ub, ws, idx, m, M0, rhs0);
        if (lin_solve_ld(M0, rhs0, n, xr) == 0)
          for (int k = 0; k < NV; k++) f[k] = xr[k];
      }
      /* a checker must not certify an infeasible point: nearly parallel rows (mu -> 0) can defeat the
       * dependent-row logic above; report failure instead of a wrong ORACLE_OK */
      double fs = 1.0; /* scale of the solution: the KKT solves leave ~cond*eps*|f| on the pinned rows */
      for (int k = 0; k < NV; k++) if (fabs(f[k]) > fs) fs = fabs(f[k]);
      for (int r = 0; r < NC; r++) {
        double cf = 0.0;
        for (int k = 0; k < NV; k++) cf += C[r * NV + k] * f[k];
        if (cf > ub[r] + 1e-7 * (fs + fabs(ub[r])) || cf < lb[r] - 1e-7 * (fs + fabs(lb[r]))) {
          if (iters_out) *iters_out = it + 1;
          return ORACLE_MAX_ITER;
        }
      }
      if (lam_out) memcpy(lam_out, lam, sizeof(lam));
      if (iters_out) *iters_out = it + 1;
      return ORACLE_OK;
    }
    ws[idx[worst]] = 0;
    for (int t = worst; t + 1 < m; t++) idx[t] = idx[t + 1];
    m--;
  }
  if (iters_out) *iters_out = it;
  return ORACLE_MAX_ITER;
}

void oracle_kkt(const double* H, const double* g, const double* C, const double* lb, const double* ub,
                const double* f, const double* lam, double* stationarity, double* primal, double* dual) {
  double st = 0.0, pr = 0.0, du = 0.0;
  for (int i = 0; i < NV; i++) {
    double s = g[i];
    for (int j = 0; j < NV; j++) s += H[i * NV + j] * f[j];
    for (int r = 0; r < NC; r++) s += C[r * NV + i] * lam[r];
    if (fabs(s) > st) st = fabs(s);
  }
  for (int r = 0; r < NC; r++) {
    double cf = 0.0;
    for (int k = 0; k < NV; k++) cf += C[r * NV + k] * f[k];
    if (cf - ub[r] > pr) pr = cf - ub[r];
    if (lb[r] - cf > pr) pr = lb[r] - cf;
    if (lb[r] == ub[r]) continue;
    double comp = (lam[r] > 0) ? lam[r] * fabs(ub[r] - cf) : -lam[r] * fabs(cf - lb[r]);
    if (comp > du) du = comp;
  }
  *stationarity = st; *primal = pr; *dual = du;
}

/* ------------------------------------------------------- control() */
int oracle_control(const oracle_params* P, const double* Rwb, const double* Rwb_d, const double* x,
                   const double* xdot, const double* w, const double* x_d, const double* xdot_d,
                   const double* w_d, const double* feet, const unsigned char* stance,
                   double* grf_body, double* f_world, int* iters) {
  double H[NV * NV], g[NV], C[NC * NV], lb[NC], ub[NC], fw[NV];
  oracle_assemble(P, Rwb, Rwb_d, x, xdot, w, x_d, xdot_d, w_d, feet, stance, H, g, C, lb, ub, 0, 0);
  /* non-finite inputs poison H or g: qpOASES cannot return SUCCESSFUL_RETURN on such data, i.e. the reference
   * ends with an empty ForceMap (BC.cpp:182-216); report it as a failed instance instead of iterating on NaNs */
  int finite = 1;
  for (int k = 0; k < NV * NV; k++) finite &= isfinite(H[k]) ? 1 : 0;
  for (int k = 0; k < NV; k++) finite &= isfinite(g[k]) ? 1 : 0;
  for (int k = 0; k < 9; k++) finite &= isfinite(Rwb[k]) ? 1 : 0;
  int st = finite ? oracle_qp_solve(H, g, C, lb, ub, P->max_iter > 0 ? P->max_iter : 200, fw, 0, iters) : ORACLE_NOT_PD;
  if (!finite && iters) *iters = 0;
  for (int k = 0; k < NV; k++) grf_body[k] = 0.0;
  if (f_world) for (int k = 0; k < NV; k++) f_world[k] = (st == ORACLE_OK) ? fw[k] : 0.0;
  if (st != ORACLE_OK) return st; /* reference: empty ForceMap, BC.cpp:182-216 */
  for (int i = 0; i < 4; i++) {
    if (!stance[i]) continue; /* swing legs omitted, BC.cpp:222 */
    for (int r = 0; r < 3; r++) /* fb = -Rwb' fw(3i..3i+2), BC.cpp:225 */
      grf_body[3 * i + r] = -(Rwb[0 + r] * fw[3 * i] + Rwb[3 + r] * fw[3 * i + 1] + Rwb[6 + r] * fw[3 * i + 2]);
  }
  return st;
}

void oracle_control_batch(const oracle_params* P, long n, const double* Rwb, const double* Rwb_d,
                          const double* x, const double* xdot, const double* w, const double* x_d,
                          const double* xdot_d, const double* w_d, const double* feet,
                          const unsigned char* stance, double* grf_body, int* status, int* iters,
                          int threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long i = 0; i < n; i++) {
    int it = 0;
    int st = oracle_control(P, Rwb + 9 * i, Rwb_d + 9 * i, x + 3 * i, xdot + 3 * i, w + 3 * i, x_d + 3 * i,
                            xdot_d + 3 * i, w_d + 3 * i, feet + 12 * i, stance + 4 * i, grf_body + 12 * i, 0, &it);
    if (status) status[i] = st;
    if (iters) iters[i] = it;
  }
}

/* ------------------------------------------------ kinematics either side of control() */
void oracle_default_kinematics(oracle_kinematics* k) {
  /* kinematics.cpp:22-42 */
  const double xbh = 0.196, ybh = 0.050, zbh = 0.0;
  const double l1 = 0.077, l2 = 0.211, l3 = 0.230;
  const double trans[4][3] = {{-xbh, ybh, zbh}, {xbh, ybh, zbh}, {-xbh, -ybh, zbh}, {xbh, -ybh, zbh}}; /* RL FL RR FR */
  const double left_links[3] = {l1, -l2, -l3}, right_links[3] = {-l1, -l2, -l3};
  for (int leg = 0; leg < 4; leg++)
    for (int c = 0; c < 3; c++) {
      k->hip[3 * leg + c] = trans[leg][c];
      k->links[3 * leg + c] = (leg < 2) ? left_links[c] : right_links[c];
    }
  k->tau_min = -20.0; /* commander_node.cpp:324 */
  k->tau_max = 20.0;  /* commander_node.cpp:325 */
  /* joint_control gains, mit_cheetah_config.yaml:50-53 */
  k->jc_kff[0] = k->jc_kff[1] = k->jc_kff[2] = 0.0;
  k->jc_kp[0] = 40.0; k->jc_kp[1] = 40.0; k->jc_kp[2] = 50.0;
  k->jc_kd[0] = k->jc_kd[1] = k->jc_kd[2] = 1.0;
  {
    const double xbt = 0.196, ybt = 0.127, zbt = 0.0; /* foot_planner.cpp:27-42 */
    const double hp[4][3] = {{-xbt, ybt, zbt}, {xbt, ybt, zbt}, {-xbt, -ybt, zbt}, {xbt, -ybt, zbt}};
    for (int leg = 0; leg < 4; leg++)
      for (int c = 0; c < 3; c++) k->planner_hip[3 * leg + c] = hp[leg][c];
  }
  k->planner_k = 0.01;    /* foot_planner.cpp:25 */
  k->swing_height = 0.08; /* gait/height */
  k->t_swing = 0.18;      /* mit_cheetah_config.yaml:17-18 */
  k->t_stance = 0.8;
}

void oracle_leg_fk(const oracle_kinematics* k, int leg, const double* q, double* p) {
  /* kinematics.cpp:81-103 */
  const double l1 = k->links[3 * leg], l2 = k->links[3 * leg + 1], l3 = k->links[3 * leg + 2];
  const double t1 = q[0], t2 = q[1], t3 = q[2];
  p[0] = l2 * sin(t2) + l3 * sin(t2 + t3) + k->hip[3 * leg];
  p[1] = l1 * cos(t1) - l2 * sin(t1) * cos(t2) - l3 * sin(t1) * cos(t2 + t3) + k->hip[3 * leg + 1];
  p[2] = l1 * sin(t1) + l2 * cos(t1) * cos(t2) + l3 * cos(t1) * cos(t2 + t3) + k->hip[3 * leg + 2];
}

void oracle_leg_jacobian(const oracle_kinematics* k, int leg, const double* q, double* jac) {
  /* kinematics.cpp:162-188 */
  const double l1 = k->links[3 * leg], l2 = k->links[3 * leg + 1], l3 = k->links[3 * leg + 2];
  const double t1 = q[0], t2 = q[1], t3 = q[2];
  jac[0] = 0.0;
  jac[1] = l2 * cos(t2) + l3 * cos(t2 + t3);
  jac[2] = l3 * cos(t2 + t3);
  jac[3] = -l1 * sin(t1) - l2 * cos(t1) * cos(t2) - l3 * cos(t1) * cos(t2 + t3);
  jac[4] = (l2 * sin(t2) + l3 * sin(t2 + t3)) * sin(t1);
  jac[5] = l3 * sin(t1) * sin(t2 + t3);
  jac[6] = l1 * cos(t1) - l2 * sin(t1) * cos(t2) - l3 * sin(t1) * cos(t2 + t3);
  jac[7] = -(l2 * sin(t2) + l3 * sin(t2 + t3)) * cos(t1);
  jac[8] = -l3 * sin(t2 + t3) * cos(t1);
}

void oracle_tick_batch(const oracle_params* P, const oracle_kinematics* K, long n, const double* Rwb,
                       const double* Rwb_d, const double* x, const double* xdot, const double* w,
                       const double* x_d, const double* xdot_d, const double* w_d, const double* joint_q,
                       const unsigned char* stance, double* feet_out, double* grf_body, double* joint_tau,
                       int* status, int threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long i = 0; i < n; i++) {
    double feet[12];
    for (int leg = 0; leg < 4; leg++) oracle_leg_fk(K, leg, joint_q + 12 * i + 3 * leg, feet + 3 * leg); /* commander_node.cpp:383-384 */
    if (feet_out) memcpy(feet_out + 12 * i, feet, sizeof(feet));
    int it = 0;
    int st = oracle_control(P, Rwb + 9 * i, Rwb_d + 9 * i, x + 3 * i, xdot + 3 * i, w + 3 * i, x_d + 3 * i,
                            xdot_d + 3 * i, w_d + 3 * i, feet, stance + 4 * i, grf_body + 12 * i, 0, &it);
    if (status) status[i] = st;
    for (int leg = 0; leg < 4; leg++) {
      double J[9];
      oracle_leg_jacobian(K, leg, joint_q + 12 * i + 3 * leg, J);
      const double* f = grf_body + 12 * i + 3 * leg;
      for (int c = 0; c < 3; c++) {
        double t = J[c] * f[0] + J[3 + c] * f[1] + J[6 + c] * f[2]; /* tau = J^T f, kinematics.cpp:226 */
        if (t < K->tau_min) t = K->tau_min;                          /* arma::clamp, commander_node.cpp:526 */
        if (t > K->tau_max) t = K->tau_max;
        joint_tau[12 * i + 3 * leg + c] = (st == ORACLE_OK && stance[4 * i + leg]) ? t : 0.0;
      }
    }
  }
}

/* ------------------------------------------------ swing legs: IK + J^-1 + joint PD */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
static double normalize_angle_2PI(double angle) { /* math/numerics.cpp:23-35 */
  const double q = floor(angle / (2.0 * M_PI));
  angle -= q * 2.0 * M_PI;
  if (angle < 0.0) angle += 2.0 * M_PI;
  return angle;
}
static double normalize_angle_PI(double rad) { /* math/numerics.cpp:37-50 */
  const double q = floor((rad + M_PI) / (2.0 * M_PI));
  rad = (rad + M_PI) - q * 2.0 * M_PI;
  if (rad < 0) rad += 2.0 * M_PI;
  return rad - M_PI;
}

void oracle_leg_ik(const oracle_kinematics* k, int leg, const double* foothold, double* q) {
  /* kinematics.cpp:117-160; links_ there are the unsigned lengths, right legs are "FR"/"RR" (legs 2, 3) */
  const double x = foothold[0] - k->hip[3 * leg], y = foothold[1] - k->hip[3 * leg + 1], z = foothold[2] - k->hip[3 * leg + 2];
  const double l1 = fabs(k->links[3 * leg]), l2 = fabs(k->links[3 * leg + 1]), l3 = fabs(k->links[3 * leg + 2]);
  double d = (x * x + y * y + z * z - l1 * l1 - l2 * l2 - l3 * l3) / (2.0 * l2 * l3);
  if (d > 1.0) d = 1.0;
  double sqrt_component = y * y + z * z - l1 * l1;
  if (sqrt_component < 0.0) sqrt_component = 0.0;
  if (k->links[3 * leg] < 0.0) q[0] = atan2(z, y) + atan2(sqrt(sqrt_component), -l1);
  else q[0] = -(atan2(z, -y) + atan2(sqrt(sqrt_component), -l1));
  q[2] = atan2(-sqrt(1.0 - d * d), d);
  q[1] = -atan2(x, sqrt(sqrt_component)) - atan2(l3 * sin(q[2]), l2 + l3 * cos(q[2]));
}

/* arma::pinv of a 3x3 (kinematics.cpp:196): Moore-Penrose inverse from the singular value decomposition with
 * Armadillo's default tolerance max(m, n) * sigma_max * epsilon.  The SVD here is the one-sided Jacobi method
 * (Hestenes): columns of A = J V are rotated pairwise until orthogonal, then sigma_i = |a_i|, u_i = a_i / sigma_i,
 * which keeps tiny singular values accurate (nothing is squared).  pinv = sum over sigma_i > tol of v_i u_i^T / sigma_i.
 * Returns 0 if the sweeps do not converge (never observed; the caller then falls through to J^T as the reference does). */
static int pinv3_keep(const double* J, int keep, double* Jp, int* rank_arma) {
  double A[3][3], V[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { A[i][j] = J[3 * i + j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
  /* A column whose norm has fallen below the tolerance that drops its singular value anyway (3 sigma_max epsilon, with the
   * largest current column norm for sigma_max) is numerically zero and its DIRECTION is rounding noise: on an exactly
   * rank-deficient J - a stretched leg, or the lateral clamp of kinematics.cpp:137-140, on some random kinematic models - the
   * orthogonality test against such a column never settles (found by the round-4 tick fuzz: 60 sweeps ran out and the J^T
   * fallback answered where Armadillo's pinv, numpy's and the device's return the rank-2 pseudo-inverse).  Pairs with such a
   * column count as converged. */
  int converged = 0;
  for (int sweep = 0; sweep < 60 && !converged; sweep++) {
    converged = 1;
    double cmax = 0.0;
    for (int j = 0; j < 3; j++) {
      const double c2 = A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j];
      if (c2 > cmax) cmax = c2;
    }
    const double zero2 = 9.0 * 2.220446049250313e-16 * 2.220446049250313e-16 * cmax;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
        for (int i = 0; i < 3; i++) { alpha += A[i][p] * A[i][p]; beta += A[i][q] * A[i][q]; gamma += A[i][p] * A[i][q]; }
        if (alpha <= zero2 || beta <= zero2) continue;
        /* orthogonal to working precision: a three-term inner product carries up to ~3 epsilon of relative rounding error, so a
         * bare epsilon makes two large columns at the noise floor swap the sign of gamma for ever (the same fuzz finding) */
        if (gamma == 0.0 || fabs(gamma) <= 4.0 * 2.220446049250313e-16 * sqrt(alpha * beta)) continue;
        converged = 0;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; i++) {
          const double ap = A[i][p], aq = A[i][q];
          A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
  }
  if (!converged) return 0;
  double sig[3], smax = 0.0;
  for (int j = 0; j < 3; j++) {
    sig[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    if (sig[j] > smax) smax = sig[j];
  }
  const double tol = 3.0 * smax * 2.220446049250313e-16;
  for (int i = 0; i < 9; i++) Jp[i] = 0.0;
  /* keep <= 3 (oracle_pinv3): Armadillo's tolerance decides alone.  keep < 3 (oracle_pinv3_band): at most the `keep` largest
   * singular values survive, whatever their size. */
  int order[3] = {0, 1, 2};
  for (int a = 0; a < 2; a++)
    for (int b = a + 1; b < 3; b++)
      if (sig[order[b]] > sig[order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
  if (rank_arma) { /* singular values Armadillo's own tolerance keeps */
    *rank_arma = 0;
    for (int j = 0; j < 3; j++) *rank_arma += sig[j] > tol ? 1 : 0;
  }
  for (int m = 0; m < 3 && m < keep; m++) {
    const int j = order[m];
    if (!(sig[j] > tol)) continue;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Jp[3 * r + c] += V[r][j] * A[c][j] / (sig[j] * sig[j]); /* v_j u_j^T / sigma_j, u_j = a_j / sigma_j */
  }
  return 1;
}

int oracle_pinv3(const double* J, double* Jp) { return pinv3_keep(J, 3, Jp, 0); }

/* Numerical rank the way this build decides it INSIDE the band where it answers with the pseudo-inverse (see
 * oracle_swing_torque): elimination with complete pivoting, a second pivot below 1e-9 of the first counts as zero and a
 * third pivot is never taken - in that band |det| < 64 epsilon (sum |l|)^3, so the third direction is the one that has
 * collapsed (a stretched knee, the lateral clamp) and 1 / sigma_3 would be a 1e13 ... 1e17-sized, noise-signed gain.
 * Armadillo's own tolerance (3 sigma_max epsilon) would keep sigma_3 for |det| between ~1e-17 and the band's upper end; the
 * granularity of the knee cosine (|det| jumps from rounding noise to ~3e-10 one ulp below d = 1) makes that window all but
 * unreachable, but device and checker must not depend on that: both take the rank from THIS rule (ADVICE r4;
 * qc_device.hpp pinv3_apply is the same elimination). */
int oracle_cp_rank3(const double* J) {
  double A[9];
  for (int i = 0; i < 9; i++) A[i] = J[i];
  double piv1 = 0.0;
  int rank = 0;
  for (int k = 0; k < 2; k++) {
    double best = -1.0;
    int bi = 0, bj = 0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        if (fabs(A[3 * i + j]) > best) { best = fabs(A[3 * i + j]); bi = i; bj = j; }
    if (k == 0) piv1 = best;
    if (!(k == 0 ? best > 0.0 : (rank == 1 && best > 1.0e-9 * piv1))) break;
    double col[3], row[3];
    for (int i = 0; i < 3; i++) { col[i] = A[3 * i + bj]; row[i] = A[3 * bi + i]; }
    const double ip = 1.0 / col[bi];
    for (int i = 0; i < 3; i++) row[i] *= ip;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) A[3 * i + j] -= col[i] * row[j];
    rank = k + 1;
  }
  return rank;
}

int oracle_pinv3_band(const double* J, double* Jp) { return pinv3_keep(J, oracle_cp_rank3(J), Jp, 0); }

/* ADVICE r5.  oracle_swing_torque restates the REFERENCE: in the singular band it answers with arma::pinv's own rule (tolerance
 * 3 sigma_max epsilon, oracle_pinv3).  The device decides the rank differently (complete pivoting, never a third pivot, a second
 * one only above 1e-9 of the first: qc_device.hpp pinv3_apply, restated as oracle_cp_rank3) - a deliberate deviation
 * (INTEGRATION.md): where Armadillo keeps a sigma between ~1e-16 and 1e-9 of sigma_max, 1 / sigma is a 1e9 ... 1e16-sized gain
 * on rounding noise.  Every swing leg on which the two rules would keep a different number of singular values is COUNTED here,
 * so that the parity tests and fuzz campaigns can assert that the window stays unvisited instead of having the checker agree
 * with the device by construction. */
static long pinv_rule_disagreements = 0;
long oracle_pinv_rule_disagreements(int reset) {
  long v;
#pragma omp atomic capture
  { v = pinv_rule_disagreements; pinv_rule_disagreements += 0; }
  if (reset) {
#pragma omp atomic write
    pinv_rule_disagreements = 0;
  }
  return v;
}

void oracle_swing_torque(const oracle_kinematics* k, int leg, const double* Rwb, const double* x, const double* pos,
                         const double* vel, const double* q, const double* qdot, double* tau) {
  double pb[3], vb[3], qr[3], J[9], Jinv[9], qd[3];
  for (int r = 0; r < 3; r++) { /* commander_node.cpp:492-493: Rwb.t() * position - x ; Rwb.t() * velocity */
    pb[r] = Rwb[r] * pos[0] + Rwb[3 + r] * pos[1] + Rwb[6 + r] * pos[2] - x[r];
    vb[r] = Rwb[r] * vel[0] + Rwb[3 + r] * vel[1] + Rwb[6 + r] * vel[2];
  }
  oracle_leg_ik(k, leg, pb, qr);          /* :495 */
  oracle_leg_jacobian(k, leg, qr, J);     /* legJacobianInverse, kinematics.cpp:190-204 */
  {
    /* legJacobianInverse, kinematics.cpp:190-204: arma::inv, if that fails arma::pinv, if that fails J^T.
     * arma::inv of a 3x3 is Armadillo's closed-form "tiny" inverse for epsilon <= |det| <= 1 / epsilon (recalled from
     * Armadillo 10.2's op_inv, not verifiable here); outside that band it calls LAPACK, whose LU returns a 1/sigma_3-sized
     * "inverse" unless a pivot is exactly zero, and only then arma::pinv answers (:196).  Restated here: the closed-form band
     * is inverted (Gauss-Jordan with partial pivoting - same values as the cofactor formula to rounding), everything
     * outside it takes the pinv branch.  A singular J - leg fully stretched: the reference point is out of reach and IK
     * clamps d to 1, so q3 = 0 and the last two columns are parallel (|det| ~ 1e-18 of rounding noise; rank 1 when
     * y^2 + z^2 < l1^2 is clamped as well) - therefore always gets pinv, where the reference gets pinv or a full-scale
     * clamped torque depending on rounding inside LAPACK (tests/test_oracle_cpu.py puts numbers on it). */
    double M[3][6];
    const double det = J[0] * (J[4] * J[8] - J[5] * J[7]) + J[1] * (J[5] * J[6] - J[3] * J[8]) + J[2] * (J[3] * J[7] - J[4] * J[6]);
    /* (lower end raised to det's own rounding noise, 64 epsilon (sum |l|)^3, where that is larger: below it the sign of
     * det - and of the saturated torque it leads to - is noise on any implementation) */
    const double lsum = fabs(k->links[3 * leg]) + fabs(k->links[3 * leg + 1]) + fabs(k->links[3 * leg + 2]);
    const double det_lo = fmax(2.220446049250313e-16, 1.4210854715202004e-14 * lsum * lsum * lsum);
    /* (a NaN determinant - d < -1 in legInverseKinematics - is neither below nor above the band: the closed form divides the
     * cofactors by NaN, and so does the elimination below: every entry of the inverse, hence every torque of the leg, is NaN) */
    int singular = (fabs(det) < det_lo) || (fabs(det) > 4503599627370496.0);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) { M[i][j] = J[3 * i + j]; M[i][3 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 3 && !singular; c++) {
      int p = c;
      for (int r = c + 1; r < 3; r++) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
      if (M[p][c] == 0.0) { singular = 1; break; }
      if (p != c) for (int j = 0; j < 6; j++) { double t = M[c][j]; M[c][j] = M[p][j]; M[p][j] = t; }
      const double piv = M[c][c];
      for (int j = 0; j < 6; j++) M[c][j] /= piv;
      for (int r = 0; r < 3; r++) {
        if (r == c) continue;
        const double m = M[r][c];
        for (int j = 0; j < 6; j++) M[r][j] -= m * M[c][j];
      }
    }
    if (!singular) {
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Jinv[3 * i + j] = M[i][3 + j];
    } else {
      int rank_arma = -1;
      const int ok = pinv3_keep(J, 3, Jinv, &rank_arma); /* arma::pinv, :196 - Armadillo's tolerance decides the rank */
      if (ok && rank_arma != oracle_cp_rank3(J)) {        /* the device's rule would keep another rank: counted, see above */
#pragma omp atomic update
        pinv_rule_disagreements += 1;
      }
      if (!ok)
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) Jinv[3 * i + j] = J[3 * j + i]; /* :198 */
    }
  }
  for (int r = 0; r < 3; r++) qd[r] = Jinv[3 * r] * vb[0] + Jinv[3 * r + 1] * vb[1] + Jinv[3 * r + 2] * vb[2]; /* :496-497 */
  for (int c = 0; c < 3; c++) { /* joint_controller.cpp:28-36 */
    const double q_error_normalized = normalize_angle_2PI(qr[c]) - normalize_angle_2PI(q[c]);
    const double q_error = normalize_angle_PI(q_error_normalized);
    tau[c] = k->jc_kp[c] * q_error + k->jc_kd[c] * (qd[c] - qdot[c]) + k->jc_kff[c];
  }
}

void oracle_tick_swing_batch(const oracle_params* P, const oracle_kinematics* K, long n, const double* Rwb,
                             const double* Rwb_d, const double* x, const double* xdot, const double* w,
                             const double* x_d, const double* xdot_d, const double* w_d, const double* joint_q,
                             const double* joint_qdot, const double* swing_pos, const double* swing_vel,
                             const unsigned char* stance, double* grf_body, double* joint_tau, int* status, int threads) {
  oracle_tick_batch(P, K, n, Rwb, Rwb_d, x, xdot, w, x_d, xdot_d, w_d, joint_q, stance, 0, grf_body, joint_tau, status, threads);
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long i = 0; i < n; i++)
    for (int leg = 0; leg < 4; leg++) {
      if (stance[4 * i + leg]) continue; /* commander_node.cpp:485: swing legs only */
      double tau[3];
      oracle_swing_torque(K, leg, Rwb + 9 * i, x + 3 * i, swing_pos + 12 * i + 3 * leg, swing_vel + 12 * i + 3 * leg,
                          joint_q + 12 * i + 3 * leg, joint_qdot + 12 * i + 3 * leg, tau);
      for (int c = 0; c < 3; c++) { /* merged map clamped at commander_node.cpp:526 */
        double t = tau[c];
        if (t < K->tau_min) t = K->tau_min;
        if (t > K->tau_max) t = K->tau_max;
        joint_tau[12 * i + 3 * leg + c] = t;
      }
    }
}

/* ------------------------------------------------ swing references: planner + sextic trajectories */
static void single_foot(const oracle_kinematics* K, int leg, const double* Rwb, const double* x, const double* xdot,
                        const double* w, const double* xdot_d, const double* foot_position, double* foothold) {
  /* FootPlanner::singleFoot, foot_planner.cpp:76-104 */
  double p_thigh[3], pcom_foot[3], tang_vel[3];
  mat3_vec(Rwb, K->planner_hip + 3 * leg, p_thigh);
  for (int r = 0; r < 3; r++) p_thigh[r] += x[r];
  mat3_vec(Rwb, foot_position, pcom_foot);
  tang_vel[0] = w[1] * pcom_foot[2] - w[2] * pcom_foot[1];
  tang_vel[1] = w[2] * pcom_foot[0] - w[0] * pcom_foot[2];
  tang_vel[2] = w[0] * pcom_foot[1] - w[1] * pcom_foot[0];
  for (int r = 0; r < 3; r++) {
    const double p_linear = (K->t_stance / 2.0) * xdot[r] + K->planner_k * (xdot[r] - xdot_d[r]);
    const double p_tangent = (K->t_stance / 2.0) * tang_vel[r];
    const double p_lip = 0.5 * sqrt(x[2] / 9.81) * xdot[r];
    foothold[r] = p_thigh[r] + p_linear + p_tangent + p_lip;
  }
  foothold[2] = 0.0;
}

static void sextic_coefficients(const double* p_start, const double* p_center, const double* p_final, double coef[7][3]) {
  /* FootTrajectory::initSystem / constantTerms / generateTrajetory, trajectory.cpp:220-225, 256-296:
   * arma::solve(A, B) restated as Gaussian elimination with partial pivoting */
  double M[7][10] = {{1, 0, 0, 0, 0, 0, 0}, {1, 1, 1, 1, 1, 1, 1}, {1, 0.5, 0.25, 0.125, 0.0625, 0.03125, 0.015625},
                     {0, 1, 0, 0, 0, 0, 0}, {0, 1, 2, 3, 4, 5, 6}, {0, 0, 2, 0, 0, 0, 0}, {0, 0, 2, 6, 12, 20, 30}};
  for (int i = 0; i < 7; i++)
    for (int c = 0; c < 3; c++) M[i][7 + c] = (i == 0) ? p_start[c] : (i == 1 ? p_final[c] : (i == 2 ? p_center[c] : 0.0));
  for (int c = 0; c < 7; c++) {
    int p = c;
    for (int r = c + 1; r < 7; r++) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
    if (p != c) for (int j = 0; j < 10; j++) { double t = M[c][j]; M[c][j] = M[p][j]; M[p][j] = t; }
    for (int r = c + 1; r < 7; r++) {
      const double m = M[r][c] / M[c][c];
      for (int j = c; j < 10; j++) M[r][j] -= m * M[c][j];
    }
  }
  for (int r = 6; r >= 0; r--)
    for (int c = 0; c < 3; c++) {
      double s2 = M[r][7 + c];
      for (int j = r + 1; j < 7; j++) s2 -= M[r][j] * coef[j][c];
      coef[r][c] = s2 / M[r][r];
    }
}

static void reference_state(const oracle_kinematics* K, const oracle_swing_state* st, int leg, double phase, double* pos, double* vel) {
  /* FootTrajectoryManager::referenceState, trajectory.cpp:360-388 */
  pos[0] = pos[1] = pos[2] = vel[0] = vel[1] = vel[2] = 0.0;
  if (!st->has_traj[leg]) return; /* FootState() */
  const double stance_phase = K->t_stance / (K->t_swing + K->t_stance); /* :303 */
  const double slope = 1.0 / (1.0 - stance_phase), y_intercept = 1.0 - slope;
  double t = slope * phase + y_intercept;
  if (t < 0.0) t = 0.0;
  if (t > 1.0) t = 1.0;
  double p_center[3], coef[7][3];
  for (int c = 0; c < 3; c++) p_center[c] = (st->p_start[3 * leg + c] + st->p_final[3 * leg + c]) / 2.0; /* :325 */
  p_center[2] = K->swing_height;                                                                         /* :326 */
  sextic_coefficients(st->p_start + 3 * leg, p_center, st->p_final + 3 * leg, coef);
  /* trackTrajectory, trajectory.cpp:227-254 */
  const double pf[7] = {1.0, t, pow(t, 2), pow(t, 3), pow(t, 4), pow(t, 5), pow(t, 6)};
  const double vf[7] = {0.0, 1.0, 2.0 * t, 3.0 * pow(t, 2), 4.0 * pow(t, 3), 5.0 * pow(t, 4), 6.0 * pow(t, 5)};
  for (int j = 0; j < 7; j++)
    for (int c = 0; c < 3; c++) { pos[c] += pf[j] * coef[j][c]; vel[c] += vf[j] * coef[j][c]; }
}

void oracle_swing_references(const oracle_kinematics* K, oracle_swing_state* st, const double* Rwb, const double* x,
                             const double* xdot, const double* w, const double* xdot_d, const double* feet_body,
                             const unsigned char* stance, const double* phase, double* pos, double* vel) {
  /* FootPlanner::updateStates, foot_planner.cpp:106-157 */
  int plan[4] = {0, 0, 0, 0}, any = 0;
  const int empty = st->leg_state[0] < 0;
  for (int leg = 0; leg < 4; leg++) {
    const int now = stance[leg] ? 1 : 0;
    if (empty) { if (!now) plan[leg] = 1; }
    else if (st->leg_state[leg] == 1 && !now) plan[leg] = 1;
    st->leg_state[leg] = now;
    any |= plan[leg];
  }
  if (any) { /* commander_node.cpp:450-462 -> referenceStates(gait_map, bounds): traj_map_.clear() then the planned legs */
    for (int leg = 0; leg < 4; leg++) {
      st->has_traj[leg] = plan[leg];
      if (!plan[leg]) continue;
      double ps[3];
      mat3_vec(Rwb, feet_body + 3 * leg, ps);
      for (int r = 0; r < 3; r++) st->p_start[3 * leg + r] = ps[r] + x[r]; /* :456 */
      single_foot(K, leg, Rwb, x, xdot, w, xdot_d, feet_body + 3 * leg, st->p_final + 3 * leg);
    }
  }
  for (int leg = 0; leg < 4; leg++) {
    for (int c = 0; c < 3; c++) pos[3 * leg + c] = vel[3 * leg + c] = 0.0;
    if (!stance[leg]) reference_state(K, st, leg, phase[leg], pos + 3 * leg, vel + 3 * leg); /* :485-489 */
  }
}

void oracle_tick_planned_batch(const oracle_params* P, const oracle_kinematics* K, long n, oracle_swing_state* states,
                               const double* Rwb, const double* Rwb_d, const double* x, const double* xdot,
                               const double* w, const double* x_d, const double* xdot_d, const double* w_d,
                               const double* joint_q, const double* joint_qdot, const double* gait_phase,
                               double* grf_body, double* joint_tau, int* status, int threads) {
  if (threads < 1) threads = 1;
  const double duty = K->t_stance / (K->t_swing + K->t_stance); /* gait.cpp:45 */
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long i = 0; i < n; i++) {
    unsigned char stance[4];
    double feet[12], pos[12], vel[12];
    for (int leg = 0; leg < 4; leg++) {
      const double ph = gait_phase[4 * i + leg]; /* GaitScheduler::phase, gait.cpp:125-134 */
      stance[leg] = ((ph > 0.0 || fabs(ph - 0.0) < 1.0e-12) && (ph < duty || fabs(ph - duty) < 1.0e-12)) ? 1 : 0;
      oracle_leg_fk(K, leg, joint_q + 12 * i + 3 * leg, feet + 3 * leg);
    }
    oracle_swing_references(K, states + i, Rwb + 9 * i, x + 3 * i, xdot + 3 * i, w + 3 * i, xdot_d + 3 * i, feet, stance,
                            gait_phase + 4 * i, pos, vel);
    oracle_tick_swing_batch(P, K, 1, Rwb + 9 * i, Rwb_d + 9 * i, x + 3 * i, xdot + 3 * i, w + 3 * i, x_d + 3 * i, xdot_d + 3 * i,
                            w_d + 3 * i, joint_q + 12 * i, joint_qdot + 12 * i, pos, vel, stance, grf_body + 12 * i,
                            joint_tau + 12 * i, status ? status + i : 0, 1);
  }
}

/* gait.cpp:113-123 */
void oracle_gait_update(const oracle_kinematics* K, long n, double* phases, const double* dt) {
  for (long i = 0; i < n; i++) {
    const double step = 1.0 / (K->t_swing + K->t_stance) * dt[i];
    for (int l = 0; l < 4; l++) phases[4 * i + l] = fmod(phases[4 * i + l] + step, 1.0);
  }
}
