// qc_balance.hip - kernels and C ABI (include/qc_balance.h) of the MI355X-native
// batched balance controller.  gfx950 only; fails loudly without a HIP device.
//
// Replaces BalanceController::control()
//   quadruped_controller/src/quadruped_controller/balance_controller.cpp:98-330 (BC.cpp)
// for n independent robots per launch.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>

#include "../../include/qc_balance.h"
#include "qc_device.hpp"

namespace qc {

// ------------------------------------------------------------------ the kernel
// A group of G lanes = one robot (qc_device.hpp).  Primal active-set method on
// the per-foot cube states:
//   fresh robot  f^ = EQP(S0)  (S0 = no active face, or the warm-start word);
//                f = clamp(f^); if nothing was clamped f^ is feasible.
//   afterwards   f^ = EQP(S);  step f -> f^ until the first blocking face (add
//                it), or, after a full step, drop the face with the most
//                negative multiplier; stop when all multipliers are >= 0 (KKT).
// The QP is strictly convex (W > 0), so the KKT point is THE minimiser qpOASES
// returns in the reference (BC.cpp:177-210).
//
// A wavefront owns a contiguous chunk of robots and walks through it: all
// lanes execute the same working-set recalculation in lockstep (straight-line,
// select-based code, no per-lane branches); a group whose robot has converged
// pushes the result to the wave's output stock and, once `refill_t` lanes are
// free, free groups pull the next robots from the wave's input stock.  This
// keeps the lanes busy although robots need between 1 and ~20 recalculations.
//
// LDS stock planes (see "Dense phases around the divergent solve" below).
// SP = plane stride in doubles = slots + 1 (odd), so that the lanes of one group (same
// slot, planes 3*FPL apart) fall into different banks; the dense side
// (plane[f][lane]) is conflict-free either way.  The stock is sized per kernel mode: a persistent wave
// (MODE 0) restocks 64 robots at a time, a one-fill wave (MODE 1/2) only ever holds its 64/G robots, so its
// stock is 64/G slots - 9.2 KB instead of 18 KB at G = 2, which is what lets more than two workgroups per
// SIMD share the CU's 160 KB (the 92-VGPR one-fill kernels are register-good for five).
// "this robot still has its polish release" (Lane::iterate): bit 9 of the stance word in the hand-over records of a RUNNING robot
// (re-pack areas, the paired waves' list); inside a lane it is the sign of the lane's multiplier tolerance `tol_s`, and a fresh
// robot starts from DevParams::tol_start
constexpr uint32_t QC_POLISH_ONE = 1u << 9;
enum { IN_B = 0, IN_R = 6, IN_FLAGS = 18, IN_IDX = 19, IN_PLANES = 20,
       OUT_F = 0, OUT_STAT = 12, OUT_WORD = 13, OUT_IDX = 14, OUT_PLANES = 15,
       STOCK_PLANES = IN_PLANES + OUT_PLANES };
constexpr int stock_slots(int G, int MODE) { return MODE == 0 ? 64 : 64 / G; }
constexpr int TASK_DOUBLES = 32;  // 256 one-byte (robot, leg) tasks of the torque pass, behind the planes
constexpr int stock_doubles(int slots) { return ((STOCK_PLANES * (slots + 1) + TASK_DOUBLES + 63) / 64) * 64; }
// QPARK - the one-lane one-fill joint_q kernels of the 6x6 forms (round 6).  The torque pass needs every robot's twelve joint angles a
// second time, tens of microseconds after the fill read them - by then the wave's lines have left the XCD's 4 MB L2 (a resident round of
// workgroups streams 16 MB through it), so they came from the fabric again: 96 B per robot on top of the algorithmic bytes
// (profiles/r06b_tick_full262144: FETCH 1.15x).  They are parked in LDS instead, twelve planes next to the nine of Rwb.  The room: the
// re-pack records of the 4-lane tail move UNDER the output stock (nobody parks a result before the last record has been read:
// finish_on_four_lanes<HOLD>, and the lanes that finished on the one-lane body push after the tail), which frees the input-stock area:
//   [0, 9) Rwb   [9, 21) joint_q   [21, 36) output stock (= re-pack records during the tail)   + the task bytes
// 36 x 65 + 32 doubles = 19 KB: eight workgroups per CU still fit the 160 KB.
constexpr int QPARK_R = 0, QPARK_Q = 9, QPARK_OUT = 21, QPARK_PLANES = 36;
constexpr int qpark_doubles() { return ((QPARK_PLANES * 65 + TASK_DOUBLES + 63) / 64) * 64; }

// RACE (mode-2 kernels of batches that leave most SIMDs idle): several lane groups solve the SAME robot with
// different pivoting strategies and the first to reach the KKT point wins.  A strategy is (nclamp, drop_all):
// the first `nclamp` recalculations clamp f^ into the frusta keeping the faces already in the set (1 = the
// classic method; more = further projection steps before the first ratio test), and after a full step either the
// most negative multiplier is dropped or all negative ones at once.  Every strategy is a primal active-set method
// that stops only at the KKT point of the strictly convex QP, i.e. at the same minimiser.
template <class Eqp, bool KIN, bool RACE = false>
struct Lane {
  static constexpr int G = Eqp::G;
  static constexpr int FPL = 4 / G;  // feet per lane
  static constexpr bool S = Eqp::kStrided;
  Wrench<FPL> Wr;
  Cube<FPL> C;
  double f[3 * FPL];
  long idx;
  uint32_t stance;  // bits 0-3: LegState per foot, bit 8: non-finite input
  int foot0;        // first foot of this lane
  int status, iters;
  double tol_s;  // relative multiplier tolerance of this robot (tol_d x Eqp::kTolScale), signed: + while it still has its polish release, - after
  bool have_f;
  int nclamp = 1;        // RACE: recalculations that clamp instead of stepping
  bool drop_all = false;  // RACE: drop every negative multiplier after a full step

  template <class PT>
  QC_DEV double lo(const PT& P, int i) const { return ((stance >> (foot0 + i)) & 1u) ? P.fzmin : 0.0; }
  template <class PT>
  QC_DEV double hi(const PT& P, int i) const { return ((stance >> (foot0 + i)) & 1u) ? P.fzmax : 0.0; }

  // multiplier test on the current face: true if all active faces have
  // lambda >= -tol; otherwise wcode = 3*foot+axis of the most negative one.
  // (`worst`: the smallest multiplier, face code in its low bits; `thr` = the bar below which a multiplier is released:
  // -tol |g|, or +tol |g| while the robot has its polish release - see iterate)
  template <class PT>
  QC_DEV bool multipliers_ok(const PT& P, const double (&g)[3 * FPL], double gscale, double& worst, double& thr, bool (&neg)[3 * FPL], uint32_t& negbits) const {
    negbits = 0u;
    // the tolerance is relative to the scale of the multipliers' rounding noise: what the 6x6 forms' solve hands over
    // (eqp_diagw: group-uniform, ready before the gradient is), max(1, |g|_inf) over the group's feet for the dense forms
    double gs = gscale;
    if constexpr (!Eqp::kHasScale) {
      gs = 1.0;
#pragma unroll
      for (int k = 0; k < 3 * FPL; k++) gs = max_abs_nn(gs, g[k]);
      gs = group_max<G, S>(gs);
    }
    thr = tol_s * gs;
    double cand[3 * FPL];
#pragma unroll
    for (int i = 0; i < FPL; i++) {
      const double lx = -(double)C.sx[i] * g[3 * i];
      const double ly = -(double)C.sy[i] * g[3 * i + 1];
      const double lz = (double)C.sz[i] * (P.mu * (lx + ly) - g[3 * i + 2]);
      const int c0 = 3 * (foot0 + i);
      cand[3 * i + 0] = tag(C.sx[i] != 0 ? lx : QC_BIG, c0 + 0);
      cand[3 * i + 1] = tag(C.sy[i] != 0 ? ly : QC_BIG, c0 + 1);
      cand[3 * i + 2] = tag(C.sz[i] != 0 ? lz : QC_BIG, c0 + 2);
      if constexpr (RACE) {  // per-axis "this multiplier is negative" for the drop-all strategy (compares; the masks live in SGPRs)
        neg[3 * i + 0] = (C.sx[i] != 0) & (lx < thr);
        neg[3 * i + 1] = (C.sy[i] != 0) & (ly < thr);
        neg[3 * i + 2] = (C.sz[i] != 0) & (lz < thr);
        if constexpr (FPL == 4) {
          // one lane per robot (the dense form's twin race): twelve lane masks = twelve SGPR pairs across the state update, next to a
          // kernel whose pointers already fill half of the SGPR file - they spill, and SGPR pairs that spill leave frame slots.
          // The same twelve answers as bits of one VGPR (negbits); the bool array is dead on this path.
          negbits |= (neg[3 * i + 0] ? 1u : 0u) << (3 * i) | (neg[3 * i + 1] ? 2u : 0u) << (3 * i) | (neg[3 * i + 2] ? 4u : 0u) << (3 * i);
          neg[3 * i + 0] = neg[3 * i + 1] = neg[3 * i + 2] = false;
        }
      } else {
        neg[3 * i + 0] = neg[3 * i + 1] = neg[3 * i + 2] = false;
      }
    }
#pragma unroll
    for (int w = 3 * FPL; w > 1; w = (w + 1) / 2)
#pragma unroll
      for (int k = 0; k < w / 2; k++) cand[k] = min_nn(cand[k], cand[k + (w + 1) / 2]);
    worst = group_min<G, S>(cand[0]);
    return !(worst < thr);
  }

  // one working-set recalculation; returns true when the robot is finished.
  // PHASE: MIXED = fresh and running robots share the wave (persistent waves with lane refill: one straight-line
  // body, the clamp of the fresh ones behind a wave-uniform branch); FIRST / STEADY = the one-fill modes, where
  // every robot of the wave is fresh in the first recalculation and none afterwards, so the first is peeled and
  // the steady body carries no clamp, no fresh/running selects and no ratio test for nothing.
  // `live`: the lane's robot is still running.  The strided 4-lane kernels iterate every lane through a
  // wave-uniform loop (MFMA sums read all 64 lanes), finished robots included: a finished robot recomputes
  // the same f^ from the same working set (idempotent), so only what must not move once it has finished -
  // working set, status, iteration count - is held back by `live`, instead of copying the whole lane state
  // and selecting it back.  Callers whose EXEC mask already excludes finished robots pass true.
  // What makes that sound (the invariants behind `live`):
  //  * a SOLVED robot is a fixed point: the same working set gives the same f^ bit for bit, no face blocks (every
  //    candidate is +BIG, alpha clamps to 1, beta = 0), so f = f^ is rewritten with the value it already holds;
  //  * a robot that is finished WITHOUT being solved (QC_NOT_PD, the iteration cap, a racing group whose partner won)
  //    does keep moving f while C stays frozen - and nobody reads that f: store_result() writes zero forces for every
  //    status other than QC_SOLVED, and a racing loser never pushes (the winner's group does);
  //  * the cap is tested per lane (`iters >= P.max_iter` in the return value), so it holds for robots that did not start
  //    together as well - the refilled lane groups of the paired-waves consumer, whose records carry their own count; in a
  //    one-fill wave all robots happen to reach it in the same recalculation.
  // tests/test_gpu_matrix.py::test_iteration_cap_and_bad_inputs_agree_across_widths holds status, iteration count and the
  // (zero) forces of capped / bad robots equal across the lane-group widths.
  enum { MIXED = 0, FIRST = 1, STEADY = 2 };
  // `empty_set` (wave-uniform): every robot of the wave still has the EMPTY working set - the first recalculation of a cold
  // fill.  With no active face there is no multiplier to test (every candidate is +BIG), so the test is skipped.
  template <int PHASE = MIXED, class PT>
  QC_DEV bool iterate(const PT& P, Eqp& eqp, const bool live = true, const bool empty_set = false) {
    double fh[3 * FPL], g[3 * FPL];
    iters += live ? 1 : 0;
    const bool pd = eqp.solve(P, Wr, C, stance, foot0, fh, g);
    // RACE, MIXED: the lane's strategy decides (its first `nclamp` recalculations are clamp steps)
    const bool fresh = PHASE == FIRST ? true : (PHASE == STEADY ? false : (RACE ? iters <= nclamp : !have_f));
    have_f = true;
    // (a) fresh robot: clamp f^ into the frusta.  In MIXED waves fresh robots exist only right
    // after a (re)fill, so the whole block sits behind a wave-uniform branch.
    double fc[3 * FPL];
    Cube<FPL> Cc;
    bool changed = false;
#pragma unroll
    for (int i = 0; i < FPL; i++) {
      fc[3 * i] = fh[3 * i]; fc[3 * i + 1] = fh[3 * i + 1]; fc[3 * i + 2] = fh[3 * i + 2];
      Cc.sx[i] = Cc.sy[i] = Cc.sz[i] = 0;
    }
    if (PHASE == FIRST || (PHASE == MIXED && __builtin_amdgcn_ballot_w64(fresh) != 0)) {
      int ch = 0;
#pragma unroll
      for (int i = 0; i < FPL; i++)
        ch |= (int)clamp_foot(P.mu, lo(P, i), hi(P, i), C.sx[i], C.sy[i], C.sz[i], fc[3 * i], fc[3 * i + 1], fc[3 * i + 2], Cc.sx[i], Cc.sy[i],
                              Cc.sz[i]);
      changed = group_or<G, S>(ch) != 0;
    }
    // (b) otherwise: ratio test over the faces outside the working set (tree min, face code in the low bits)
    bool blocked = false;
    int bcode = -1;
    if constexpr (PHASE != FIRST) {
      double cand[6 * FPL];
#pragma unroll
      for (int i = 0; i < FPL; i++) {
        const double fx = f[3 * i], fy = f[3 * i + 1], fz = f[3 * i + 2];
        const double dx = fh[3 * i] - fx, dy = fh[3 * i + 1] - fy, dz = fh[3 * i + 2] - fz;
        const double m = P.mu * fz, md = P.mu * dz;
        const bool zf = C.sz[i] == 0, xf = C.sx[i] == 0, yf = C.sy[i] == 0;
        const int c0 = 6 * (foot0 + i);
        cand[6 * i + 0] = step_cand(xf, m + fx, -dx - md, c0 + 0);     // X-
        cand[6 * i + 1] = step_cand(xf, m - fx, dx - md, c0 + 1);      // X+
        cand[6 * i + 2] = step_cand(yf, m + fy, -dy - md, c0 + 2);     // Y-
        cand[6 * i + 3] = step_cand(yf, m - fy, dy - md, c0 + 3);      // Y+
        cand[6 * i + 4] = step_cand(zf, fz - lo(P, i), -dz, c0 + 4);   // Z-
        cand[6 * i + 5] = step_cand(zf, hi(P, i) - fz, dz, c0 + 5);    // Z+
      }
#pragma unroll
      for (int w = 6 * FPL; w > 1; w = (w + 1) / 2)
#pragma unroll
        for (int k = 0; k < w / 2; k++) cand[k] = min_nn(cand[k], cand[k + (w + 1) / 2]);
      const double amin = group_min<G, S>(cand[0]);
      blocked = !fresh & (amin < 1.0e299);
      bcode = blocked ? tag_code(amin) : -1;
      // f <- f^ + (1 - alpha)(f - f^): exactly f^ for a full step.  No face blocks: amin = +BIG, clamped to alpha = 1
      // (branch-free; a blocking ratio that the approximate reciprocal rounds up to 1 is clamped too)
      const double beta = 1.0 - min_nn(max_nn(amin, 0.0), 1.0);
#pragma unroll
      for (int k = 0; k < 3 * FPL; k++) f[k] = fresh ? fc[k] : __builtin_fma(beta, f[k] - fh[k], fh[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 3 * FPL; k++) f[k] = fc[k];
    }
    // multiplier test, meaningful when f landed on f^
    const bool at_fh = fresh ? !changed : !blocked;
    bool neg[3 * FPL];
#pragma unroll
    for (int k = 0; k < 3 * FPL; k++) neg[k] = false;
    bool opt = true;
    uint32_t negbits = 0u;  // (FPL == 4 && RACE: `neg` as bits of one register)
    double worst = QC_BIG, thr = 0.0;
    if (!(PHASE == FIRST && empty_set)) opt = multipliers_ok(P, g, eqp.gscale, worst, thr, neg, negbits);
    if constexpr (RACE && FPL == 4) asm volatile("" : "+v"(negbits));
    // POLISH.  The bar -tol |g| sits just above the multipliers' rounding noise, and along a weakly active face the objective
    // curves only with 2w: a multiplier accepted anywhere in (-tol |g|, 0) leaves the force up to tol |g| / (2w) from the
    // minimiser (w ~ 1e-7: 3e-3 N = 6.9e-5 relative, profiles/r05_fuzz_dig_138.log), and one whose true value is negative but
    // inside the noise about as far.  So a robot starts with the bar at +tol |g| (`thr` = tol_s x scale, tol_s > 0): a face
    // whose multiplier lies inside the band (-tol |g|, +tol |g|) counts as not optimal and is released like a negative one, and
    // the walk goes on as the primal active-set method it is (the objective never increases) - either the larger subproblem's
    // minimiser is feasible, which is the better point, or the released face blocks at a zero-length step and comes back.
    // The first release that was NOT of a clearly negative multiplier (|worst| <= tol |g|) flips the robot's tolerance for good:
    // from then on the rule is round 5's, so a weakly active face is released once and cannot cycle.  Clearly negative
    // multipliers are dropped as ever and leave the polish release in place.  Per recalculation: one compare and one select
    // (a budget counter in a register of its own cost the chain-bound kernels 2 %, the flag inside `status` 2.7 %, an EXEC
    // branch around the test 2-4.5 %: profiles/r06_polish_ab.log).
    const int wcode = (at_fh & !opt) ? tag_code(worst) : -1;
    {
      const bool polish = at_fh & !opt & (__builtin_fabs(worst) <= __builtin_fabs(thr));  // (false once tol_s < 0: then !opt means worst < -tol |g|)
      const unsigned long long tb = (unsigned long long)__double_as_longlong(tol_s);
      const uint32_t hi = (uint32_t)(tb >> 32) | (polish ? 0x80000000u : 0u);  // (select between constants: written as a select between the two words it became an EXEC branch)
      tol_s = __longlong_as_double((long long)(((unsigned long long)hi << 32) | (uint32_t)tb));
    }
    const bool dall = RACE & drop_all & at_fh;  // this lane's strategy drops every multiplier below the bar at once
    const bool take_clamp = fresh & changed;
#pragma unroll
    for (int i = 0; i < FPL; i++) {
      int sx = C.sx[i], sy = C.sy[i], sz = C.sz[i];
      const int b0 = 6 * (foot0 + i), w0 = 3 * (foot0 + i);
      const bool nx = (RACE && FPL == 4) ? ((negbits >> (3 * i)) & 1u) != 0 : neg[3 * i + 0];
      const bool ny = (RACE && FPL == 4) ? ((negbits >> (3 * i + 1)) & 1u) != 0 : neg[3 * i + 1];
      const bool nz = (RACE && FPL == 4) ? ((negbits >> (3 * i + 2)) & 1u) != 0 : neg[3 * i + 2];
      const bool px = RACE ? (dall ? nx : wcode == w0 + 0) : wcode == w0 + 0;
      const bool py = RACE ? (dall ? ny : wcode == w0 + 1) : wcode == w0 + 1;
      const bool pz = RACE ? (dall ? nz : wcode == w0 + 2) : wcode == w0 + 2;
      sx = (bcode == b0 + 0) ? -1 : ((bcode == b0 + 1) ? 1 : (px ? 0 : sx));
      sy = (bcode == b0 + 2) ? -1 : ((bcode == b0 + 3) ? 1 : (py ? 0 : sy));
      sz = (bcode == b0 + 4) ? -1 : ((bcode == b0 + 5) ? 1 : (pz ? 0 : sz));
      C.sx[i] = live ? (take_clamp ? Cc.sx[i] : sx) : C.sx[i];
      C.sy[i] = live ? (take_clamp ? Cc.sy[i] : sy) : C.sy[i];
      C.sz[i] = live ? (take_clamp ? Cc.sz[i] : sz) : C.sz[i];
    }
    // (bitwise on purpose: short-circuit forms become exec-mask branches on the serial chain)
    const bool solved = pd & at_fh & opt;
    status = live ? (!pd ? (int)QC_NOT_PD : (solved ? (int)QC_SOLVED : status)) : status;  // otherwise it stays QC_MAX_ITER
    return (!pd) | solved | (iters >= P.max_iter);
  }

  // take the robot staged in `slot` of the wave's input stock into this lane's group
  // (`tol` = DevParams::tol_start: a fresh robot has its polish release unless the handle switched the polish off)
  template <int SP>
  QC_DEV void load_from_stock(const double* __restrict__ sin, int slot, int member, double tol) {
    foot0 = member * FPL;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      Wr.b[k] = Eqp::kNegB ? -sin[(IN_B + k) * SP + slot] : sin[(IN_B + k) * SP + slot];
      // (opaque: otherwise the compiler keeps +b and re-materialises the negation - an MFMA addend takes no source
      // modifier - inside every recalculation)
      if constexpr (Eqp::kNegB) asm volatile("" : "+v"(Wr.b[k]));
    }
#pragma unroll
    for (int i = 0; i < FPL; i++)
#pragma unroll
      for (int k = 0; k < 3; k++) Wr.r[i][k] = sin[(IN_R + 3 * (foot0 + i) + k) * SP + slot];
    const unsigned long long fl = (unsigned long long)__double_as_longlong(sin[IN_FLAGS * SP + slot]);
    stance = (uint32_t)fl;
    tol_s = Eqp::kTolScale * tol;
    const uint32_t wv = (uint32_t)(fl >> 32);
    idx = __double_as_longlong(sin[IN_IDX * SP + slot]);
    const bool use_warm = (wv & 0x80000000u) != 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) {
      const uint32_t fb = wv >> (6 * (foot0 + i));
      const bool st = use_warm && ((stance >> (foot0 + i)) & 1u);
      C.sx[i] = st ? dec2(fb) : 0;
      C.sy[i] = st ? dec2(fb >> 2) : 0;
      C.sz[i] = st ? dec2(fb >> 4) : 0;
    }
#pragma unroll
    for (int k = 0; k < 3 * FPL; k++) f[k] = 0.0;
    status = QC_MAX_ITER;
    iters = 0;
    have_f = false;
  }

  // take a freshly assembled robot straight from registers (one lane per robot, no stock in between)
  QC_DEV void load_direct(const Wrench<FPL>& W, uint32_t st, uint32_t wv, long robot, int member, double tol) {
    static_assert(G == 1, "one lane assembles and solves the whole robot");
    foot0 = member * FPL;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      Wr.b[k] = Eqp::kNegB ? -W.b[k] : W.b[k];
      if constexpr (Eqp::kNegB) asm volatile("" : "+v"(Wr.b[k]));
    }
#pragma unroll
    for (int i = 0; i < FPL; i++)
#pragma unroll
      for (int k = 0; k < 3; k++) Wr.r[i][k] = W.r[i][k];
    stance = st;
    tol_s = Eqp::kTolScale * tol;
    idx = robot;
    const bool use_warm = (wv & 0x80000000u) != 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) {
      const uint32_t fb = wv >> (6 * (foot0 + i));
      const bool on = use_warm && ((stance >> (foot0 + i)) & 1u);
      C.sx[i] = on ? dec2(fb) : 0;
      C.sy[i] = on ? dec2(fb >> 2) : 0;
      C.sz[i] = on ? dec2(fb >> 4) : 0;
    }
#pragma unroll
    for (int k = 0; k < 3 * FPL; k++) f[k] = 0.0;
    status = QC_MAX_ITER;
    iters = 0;
    have_f = false;
  }
  // bit 9 of the stance word a RUNNING robot travels with (hand-over records)
  QC_DEV uint32_t polish_bit() const { return tol_s > 0.0 ? QC_POLISH_ONE : 0u; }
  // the working-set word of this lane's feet (bit 31 = valid is added by the caller)
  QC_DEV uint32_t word_bits() const {
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) word |= encode_foot(C.sx[i], C.sy[i], C.sz[i]) << (6 * (foot0 + i));
    return word;
  }

  // park the finished robot's result in `slot` of the wave's output stock
  template <int SP>
  QC_DEV void push_result(double* __restrict__ sout, int slot) const {
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) {
#pragma unroll
      for (int k = 0; k < 3; k++) sout[(OUT_F + 3 * (foot0 + i) + k) * SP + slot] = f[3 * i + k];
      word |= encode_foot(C.sx[i], C.sy[i], C.sz[i]) << (6 * (foot0 + i));
    }
    word = (uint32_t)group_or<G, S>((int)word) | 0x80000000u;
    if (foot0 == 0) {
      sout[OUT_STAT * SP + slot] = __longlong_as_double((long long)(((unsigned long long)(uint32_t)iters << 32) | (uint32_t)status));
      sout[OUT_WORD * SP + slot] = __longlong_as_double((long long)(((unsigned long long)stance << 32) | word));
      sout[OUT_IDX * SP + slot] = __longlong_as_double(idx);
    }
  }
};

// Dense phases around the divergent solve.  Robots need 1 ... ~20
// recalculations, so the solver lanes are refilled a few at a time; running the
// expensive per-robot assembly (rotation log with atan2/sqrt, Newton-Euler
// right-hand side, optional forward kinematics: ~1400 instructions) and the
// output transform under such sparse exec masks would waste most of the wave.
// Instead every LANE assembles one whole robot for a stock of 64 robots in LDS,
// groups pull from the stock with a handful of ds_reads, finished groups push
// their world-frame forces to an output stock, and a dense pass (again one robot
// per lane) rotates them to the body frame, applies J^T and stores.
// Stock planes are [field][64 slots] doubles: the dense side touches
// plane[field][lane] (conflict-free), the group side one slot per group.
// Swing-planning state update of one robot at the start of its tick, commander_node.cpp:432-471:
// FootPlanner::updateStates (foot_planner.cpp:106-157) decides which legs need a new foothold (stance ->
// swing edge, or any swinging leg on the very first call); if there is one, FootTrajectoryManager::
// referenceStates(gait_map, bounds) (trajectory.cpp:308-344) CLEARS every stored trajectory and creates
// those of the planned legs from p_start = Rwb foot + x (commander_node.cpp:456) and the planned foothold.
// Returns the has_traj bits of this lane's legs after the update (bit = leg number): they travel in the robot's stance word (bits
// 12-15) to the torque pass, which used to read them back - one agent-scope load per swing leg, of a record line the wave
// already had - from memory.
template <int FPL, bool STR = false>
QC_DEV uint32_t swing_plan(CParams& P, const BatchIn& in, long robot, int foot0, uint32_t stance, const RawState& St, const Wrench<FPL>& W,
                           const TickExtra& X) {
  constexpr int GG = 4 / FPL;
  SwingState* S = in.swing_state + robot;
  const bool first = X.leg_state[0] < 0;  // state_map_.empty()
  int plan = 0;
  int had[FPL];
#pragma unroll
  for (int i = 0; i < FPL; i++) {
    // (selects over the four fetched words: foot0 is a per-lane value in the lane-group layouts)
    const int f = foot0 + i;
    const int prev = f == 0 ? X.leg_state[0] : (f == 1 ? X.leg_state[1] : (f == 2 ? X.leg_state[2] : X.leg_state[3]));
    had[i] = f == 0 ? X.has[0] : (f == 1 ? X.has[1] : (f == 2 ? X.has[2] : X.has[3]));
    const bool swing_now = !((stance >> f) & 1u);
    if (swing_now && (first || prev == 1)) plan |= 1 << i;
  }
  const bool any = group_or<GG, STR>(plan) != 0;
  uint32_t has_bits = 0u;
#pragma unroll
  for (int i = 0; i < FPL; i++) {
    const int leg = foot0 + i;
    if ((plan >> i) & 1) {
      // foot_actual_map (commander_node.cpp:383-384) went through Rwb in the wrench assembly already: W.r[i] = Rwb foot_body
      double fh[3];
      plan_foothold(P, leg, St.R, St.x, St.xdot, St.w, St.xdotd, W.r[i], fh);
#pragma unroll
      for (int r = 0; r < 3; r++) {
        S->p_start[3 * leg + r] = W.r[i][r] + St.x[r];  // Rwb foot + x, :456
        S->p_final[3 * leg + r] = fh[r];
      }
    }
    const int has_now = any ? ((plan >> i) & 1) : had[i];
    S->has_traj[leg] = has_now;
    S->leg_state[leg] = ((stance >> leg) & 1u) ? 1 : 0;
    has_bits |= has_now ? (1u << leg) : 0u;
  }
  return has_bits;
}

// the assembly of one robot (or of this lane's feet of it): wrench target and lever arms, contact state, optional
// gait clock and swing planning.  Returns the stance word (bits 0-3 LegState, bit 8 = non-finite input).
// (`S`, `fp`: what fetch_state() read; `sw`: the robot's four LegState bytes if in.stance is given)
template <bool KIN, int FPL, bool STR>
QC_DEV uint32_t assemble_from_state(CParams& P, const BatchIn& in, long robot, int member, const RawState& S, const double (&fp)[3 * FPL], uint32_t sw,
                                    const TickExtra& X, Wrench<FPL>& W) {
  constexpr int GG = 4 / FPL;
  const int foot0 = member * FPL;
  const double fin = wrench_from_state<FPL, KIN>(P, S, fp, foot0, W);
  uint32_t stance = 0xFu;  // make_stance_gait(), gait.cpp:24-34
  if (in.stance) {
    stance = ((sw & 0xFFu) ? 1u : 0u) | ((sw & 0xFF00u) ? 2u : 0u) | ((sw & 0xFF0000u) ? 4u : 0u) | ((sw & 0xFF000000u) ? 8u : 0u);
  } else if (in.gait_phase) {
    // GaitScheduler::phase(), gait.cpp:125-134 (almost_equal = |a-b| < 1e-12, math/numerics.cpp:18-21)
    const double duty = in.gait_duty ? X.duty : P.stance_phase;
    stance = 0;
    double phs[4];
#pragma unroll
    for (int i = 0; i < 4; i++) phs[i] = X.ph[i];
    if (in.gait_dt) {  // GaitScheduler::update(dt), gait.cpp:113-123: the clock of this robot advances first
      const double step = 1.0 / (P.t_swing + P.t_stance) * X.dt;
      double* wp = in.gait_phase + 4 * robot;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        // fmod(v, 1.0), bit for bit, without the library's division loop (four of them per robot were ~2 us of a wave's fill): the
        // integer part of a double subtracts exactly, and fmod's result carries the sign of v (-1.0 -> -0.0; inf -> NaN both ways)
        const double v = phs[i] + step;
        phs[i] = __builtin_copysign(v - __builtin_trunc(v), v);
        // every lane of the group computes the same four values; the first one stores them (read again by this wave's
        // store phase for the swing trajectories - workgroup scope, see swing_fetch)
        if (member == 0) __hip_atomic_store(wp + i, phs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double ph = phs[i];
      const bool ge0 = (ph > 0.0) || (fabs(ph) < 1.0e-12);
      const bool le = (ph < duty) || (fabs(ph - duty) < 1.0e-12);
      stance |= (ge0 && le) ? (1u << i) : 0u;
    }
  }
  if (KIN && in.swing_state)  // (bits 12-15: has_traj per leg, for the torque pass; only member 0's word reaches the stock: the group's bits are or-ed)
    stance |= (uint32_t)group_or<GG, STR>((int)swing_plan<FPL, STR>(P, in, robot, foot0, stance, S, W, X)) << 12;
  // non-finite inputs poison b, r or R: report QC_NOT_PD instead of iterating on NaNs
  const bool bad = group_or<GG, STR>(!(fin == 0.0) ? 1 : 0) != 0;
  if (bad) {
    stance |= 0x100u;
#pragma unroll
    for (int k = 0; k < 6; k++) W.b[k] = 0.0;
#pragma unroll
    for (int i = 0; i < FPL; i++) W.r[i][0] = W.r[i][1] = W.r[i][2] = 0.0;
  }
  return stance;
}

template <bool KIN, int FPL, bool STR>
QC_DEV uint32_t assemble_robot(CParams& P, const BatchIn& in, long robot, int member, Wrench<FPL>& W) {
  // all loads first, back to back (see the one-lane fill in balance_kernel): one memory round trip instead of four
  const uint32_t sw = in.stance ? *reinterpret_cast<const uint32_t*>(in.stance + 4 * robot) : 0u;
  RawState S;
  TickExtra X;
  double fp[3 * FPL];
  fetch_extra<KIN>(in, robot, X);
  fetch_state<FPL, KIN>(in, robot, member * FPL, S, fp);
  asm volatile("" ::: "memory");
  return assemble_from_state<KIN, FPL, STR>(P, in, robot, member, S, fp, sw, X, W);
}

// FPL = 4: one lane assembles a whole robot (dense restock of a big batch);
// FPL = 4/G: the G lanes of a group share a robot, each doing its own feet
// (small fills, where latency matters more than lane efficiency).
template <bool KIN, int FPL, bool STR, int SP>
QC_DEV void assemble_to_stock(CParams& P, const BatchIn& in, const uint32_t* __restrict__ warm, long robot, int slot, int member,
                              double* __restrict__ sin) {
  Wrench<FPL> W;
  const int foot0 = member * FPL;
  const uint32_t wv = warm ? warm[robot] : 0u;  // (issued with the robot's other loads, not behind the assembly)
  const uint32_t stance = assemble_robot<KIN, FPL, STR>(P, in, robot, member, W);
#pragma unroll
  for (int i = 0; i < FPL; i++)
#pragma unroll
    for (int k = 0; k < 3; k++) sin[(IN_R + 3 * (foot0 + i) + k) * SP + slot] = W.r[i][k];
  if (member == 0) {
#pragma unroll
    for (int k = 0; k < 6; k++) sin[(IN_B + k) * SP + slot] = W.b[k];
    sin[IN_FLAGS * SP + slot] = __longlong_as_double((long long)(((unsigned long long)wv << 32) | stance));
    sin[IN_IDX * SP + slot] = __longlong_as_double(robot);
  }
}

// output transform, BC.cpp:218-232: fb = -Rwb^T fw for stance legs.  `fw` = world-frame forces of
// the FPL feet from foot0 on; `member` 0 also writes the per-robot words.  (The joint torques of the widened tick are the
// torque pass's, below.)
// `Rl` != nullptr: the robot's Rwb is parked in LDS (entry k at Rl[k * rstride]) by the wave's assembly phase - the one-lane
// one-fill kernel keeps it there instead of reading the 72-byte row from global memory a second time.
// (RLDS is a template parameter, not a null test on `Rl`: a load through a pointer selected between LDS and global memory is
// a FLAT load - it was one, nine times per robot, in every one-lane kernel of rounds 2 and 3.)
template <bool KIN, int FPL, bool RLDS = false>
QC_DEV void store_result(CParams& P, const BatchIn& in, const BatchOut& out, long idx, uint32_t stance, int status, int iters, uint32_t word,
                         const double (&fw)[3 * FPL], int member, const double* __restrict__ Rl = nullptr, int rstride = 0) {
  const int foot0 = member * FPL;
  double R[9];
  if constexpr (RLDS) {
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rl[k * rstride];
  } else {
    const double* Rp = in.Rwb + 9 * idx;
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rp[k];
  }
  const int st_out = (stance & 0x100u) ? (int)QC_NOT_PD : status;
  double* o = out.grf_body + 12 * idx + 3 * foot0;
#pragma unroll
  for (int i = 0; i < FPL; i++) {
    const bool st = ((stance >> (foot0 + i)) & 1u) && st_out == QC_SOLVED;
    double f[3];
#pragma unroll
    for (int k = 0; k < 3; k++) f[k] = fw[3 * i + k];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const double v = -(R[r] * f[0] + R[3 + r] * f[1] + R[6 + r] * f[2]);
      o[3 * i + r] = st ? v : 0.0;
    }
  }
  if (member == 0) {
    out.status[idx] = st_out;
    if (out.active_set) out.active_set[idx] = word;
    if (out.iterations) out.iterations[idx] = iters;
  }
}

template <bool KIN, int FPL, int SP, bool RLDS = false>
QC_DEV void store_from_stock(CParams& P, const BatchIn& in, const BatchOut& out, const double* __restrict__ sout, int slot, int member,
                             const double* __restrict__ Rl = nullptr) {
  const int foot0 = member * FPL;
  const long idx = __double_as_longlong(sout[OUT_IDX * SP + slot]);
  const unsigned long long sw = (unsigned long long)__double_as_longlong(sout[OUT_STAT * SP + slot]);
  const unsigned long long ww = (unsigned long long)__double_as_longlong(sout[OUT_WORD * SP + slot]);
  double fw[3 * FPL];
#pragma unroll
  for (int k = 0; k < 3 * FPL; k++) fw[k] = sout[(OUT_F + 3 * foot0 + k) * SP + slot];
  store_result<KIN, FPL, RLDS>(P, in, out, idx, (uint32_t)(ww >> 32), (int)(uint32_t)sw, (int)(uint32_t)(sw >> 32), (uint32_t)ww, fw, member, Rl, SP);
}

// dense assembly of the next (up to 64) robots of the chunk into the input stock; returns how many
template <int G, bool KIN, bool STR, int SP>
QC_DEV int restock(const DevParams* __restrict__ Pg, const BatchIn& in, const uint32_t* __restrict__ warm, long cursor, long end, int lane,
                   int member, double* __restrict__ sin) {
  const long left = end - cursor;
  const int k = left < 64 ? (int)left : 64;
  if (G > 1 && k <= 64 / G) {  // few robots: the lanes of a group share one
    const int grp = lane_group<G, STR>(lane);
    if (grp < k) {
      CParams& P = *QC_PARAMS_HERE(Pg);
      assemble_to_stock<KIN, 4 / G, STR, SP>(P, in, warm, cursor + grp, grp, member, sin);
    }
  } else if (lane < k) {
    CParams& P = *QC_PARAMS_HERE(Pg);
    assemble_to_stock<KIN, 4, false, SP>(P, in, warm, cursor + lane, lane, 0, sin);
  }
  __syncthreads();
  return k;
}

// ---- the torque pass of the widened tick (SURVEY 8f rows 1 + 4): joint_tau of the robots parked in the output stock ----------
// commander_node.cpp:482-531: swing legs get IK -> J^-1 -> joint PD torques, stance legs tau = J^T f_body, all clamped.
// Round 3 looped over the four legs with one robot per lane and a per-lane `if (swing) ... else ...`: in a batch of mixed
// contact states every leg is swinging in SOME lane, so every lane walked through IK + J^-1 + PD and through J^T four times.
// Now a swinging (robot, leg) pair is a TASK: the wave's swing tasks are listed (one byte each: slot << 2 | leg, in a 256-byte
// array behind the stock planes; built with ballots and mbcnt, leg-major, so neighbouring lanes still touch neighbouring
// robots) and run in consecutive lanes - one or two passes of the ~650-instruction swing chain instead of four.  The cheap
// stance map (J^T f: three sincos and a dozen products per leg) stays one robot per lane: its loads are one coalesced batch and
// its per-leg constants scalar operands.
// What makes the pass cheap is as much WHEN its loads are issued (a wave of a one-round launch has its SIMD to itself: a
// dependent global load is ~1 us of nothing): the stance legs' joint angles are requested before the force stores, the first
// swing pass's inputs before the stance legs are computed, and every further swing pass's inputs while the previous one computes.
QC_DEV void store_tau(CParams& P, const BatchOut& out, long idx, int leg, const double (&tau)[3], bool emit) {
  double* to = out.joint_tau + 12 * idx + 3 * leg;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    // arma::clamp (commander_node.cpp:526) is two compares: a NaN torque (NaN joint state) stays NaN, as in
    // the reference, instead of turning into a full-scale command the way fmin(fmax()) would make it
    const double t = tau[r];
    to[r] = emit ? (t < P.tau_min ? P.tau_min : (t > P.tau_max ? P.tau_max : t)) : 0.0;
  }
}
// what one swing-leg task reads from memory
struct SwingIn {
  long idx;
  int leg, has;
  double q[3], qdot[3], x[3], a[3], b[3], ph;  // a, b: trajectory end points (swing_state) or reference position / velocity (swing_pos / swing_vel)
};
// (`stance_word`: the robot's stance word from the output stock - bits 12-15 carry has_traj as the assembly phase left it)
// (`qslot` != nullptr - QPARK: the robot's parked joint angles, entry k at qslot[k * qstride])
QC_DEV void swing_fetch(const BatchIn& in, long idx, int leg, uint32_t stance_word, SwingIn& T, const double* __restrict__ qslot = nullptr, int qstride = 0) {
  T.idx = idx;
  T.leg = leg;
  T.has = 1;
  T.ph = 0.0;
  if (in.swing_state) {  // FootTrajectoryManager::referenceState(leg, phase), trajectory.cpp:360-388
    const SwingState* S = in.swing_state + idx;
    T.has = (int)((stance_word >> (12 + leg)) & 1u);
    // p_start / p_final / the advanced phases may have been written by THIS wave's assembly phase (another lane of it): workgroup
    // scope is all that takes - one wave, one CU, one write-through vector L1, a barrier's release / acquire in between.  Rounds 4-5
    // used agent scope here, which on this multi-die part means "visible across XCDs": every one of the four 8-byte phase stores
    // went out as its own write-through (WRITE_SIZE 1.40x the bytes written: profiles/r06a_tick_full262144) and every load
    // bypassed the L1 that held the line.
#pragma unroll
    for (int r = 0; r < 3; r++) {
      T.a[r] = __hip_atomic_load(&S->p_start[3 * leg + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      T.b[r] = __hip_atomic_load(&S->p_final[3 * leg + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    T.ph = in.gait_dt ? __hip_atomic_load(in.gait_phase + 4 * idx + leg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : in.gait_phase[4 * idx + leg];
  } else {
#pragma unroll
    for (int r = 0; r < 3; r++) {
      T.a[r] = in.swing_pos[12 * idx + 3 * leg + r];
      T.b[r] = in.swing_vel[12 * idx + 3 * leg + r];
    }
  }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    T.q[r] = qslot ? qslot[(3 * leg + r) * qstride] : in.joint_q[12 * idx + 3 * leg + r];
    T.qdot[r] = in.joint_qdot[12 * idx + 3 * leg + r];
    T.x[r] = in.x[3 * idx + r];
  }
}
// swing leg: reference foot state -> IK -> J^-1 -> joint PD, commander_node.cpp:482-504; independent of the QP's status
QC_DEV void swing_leg_task(CParams& P, const BatchIn& in, const BatchOut& out, const SwingIn& T, const double (&R)[9]) {
  double sp[3], sv[3];
  if (in.swing_state) {
    track_swing(P, T.ph, T.a, T.b, sp, sv);
    if (!T.has) sp[0] = sp[1] = sp[2] = sv[0] = sv[1] = sv[2] = 0.0;  // no trajectory: FootState() (:387)
  } else {
#pragma unroll
    for (int r = 0; r < 3; r++) { sp[r] = T.a[r]; sv[r] = T.b[r]; }
  }
  double pb[3], vb[3], tau[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    pb[r] = R[r] * sp[0] + R[3 + r] * sp[1] + R[6 + r] * sp[2] - T.x[r];  // Rwb^T pos - x (sic, :492)
    vb[r] = R[r] * sv[0] + R[3 + r] * sv[1] + R[6 + r] * sv[2];          // Rwb^T vel (:493)
  }
  const LegGeom g = leg_geom(P, T.leg);
  leg_swing_torque(P, g, pb, vb, T.q, T.qdot, tau);
  store_tau(P, out, T.idx, T.leg, tau, true);
}
// Rwb of the robot in `slot`: the wave's LDS rows (RLDS: the one-lane one-fill kernel parked it there) or global memory
template <int SP, bool RLDS>
QC_DEV void task_rwb(const BatchIn& in, const double* __restrict__ Rplanes, int slot, long idx, double (&R)[9]) {
  if constexpr (RLDS) {
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rplanes[k * SP + slot];
  } else {
    const double* Rp = in.Rwb + 9 * idx;
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rp[k];
  }
}

// joint angles of this lane's robot, requested before anything else in the flush (consumed by torque_pass)
struct TorquePre {
  double q[12];
};
// Few robots in the stock of a lane-group kernel (chain-bound batches: four racing lane groups hold FOUR robots): the legs are
// spread over the lanes of the robot's group, as the force stores are - one robot per lane would walk a lane through all
// four legs while 60 lanes idle (measured: +2 us on the 4 096-robot fused tick).
template <int G>
QC_DEV bool torque_by_group(int out_n) { return G > 1 && out_n <= 64 / G; }
// (`Qplanes` != nullptr - QPARK: the joint angles parked by the fill, plane k of slot s at Qplanes[k * SP + s])
template <int SP, int G, bool STR>
QC_DEV void torque_prefetch(const BatchIn& in, const double* __restrict__ sout, int out_n, int lane, TorquePre& T, const double* __restrict__ Qplanes = nullptr) {
  // (defined in every lane and on both paths: an array written to different extents on two paths stays in memory - scratch)
#pragma unroll
  for (int k = 0; k < 12; k++) T.q[k] = 0.0;
  if (torque_by_group<G>(out_n)) {
    constexpr int FPL = 4 / G;
    const int grp = lane_group<G, STR>(lane);
    if (grp < out_n) {
      const long idx = __double_as_longlong(sout[OUT_IDX * SP + grp]);
      const double* qp = in.joint_q + 12 * idx + 3 * FPL * lane_member<G, STR>(lane);
#pragma unroll
      for (int k = 0; k < 3 * FPL; k++) T.q[k] = qp[k];
    }
  } else if (lane < out_n) {
    if (Qplanes) {  // (wave-uniform)
#pragma unroll
      for (int k = 0; k < 12; k++) T.q[k] = Qplanes[k * SP + lane];
    } else {
      const long idx = __double_as_longlong(sout[OUT_IDX * SP + lane]);
      const double* qp = in.joint_q + 12 * idx;
#pragma unroll
      for (int k = 0; k < 12; k++) T.q[k] = qp[k];
    }
  }
}
// the stance-leg torque map of NL legs of the robot in `slot`, from leg0 on (their joint angles in q[0 .. 3 NL))
template <int SP, bool RLDS, int NL, bool RUNTIME_LEG>
QC_DEV void stance_legs(CParams& P, const BatchIn& in, const BatchOut& out, const double* __restrict__ sout, const double* __restrict__ Rplanes, int slot,
                        int leg0, bool have_swing, const double (&q)[12]) {
  const long idx = __double_as_longlong(sout[OUT_IDX * SP + slot]);
  const unsigned long long sw = (unsigned long long)__double_as_longlong(sout[OUT_STAT * SP + slot]);
  const uint32_t stance = (uint32_t)((unsigned long long)__double_as_longlong(sout[OUT_WORD * SP + slot]) >> 32);
  const int st_out = (stance & 0x100u) ? (int)QC_NOT_PD : (int)(uint32_t)sw;
  double R[9];
  task_rwb<SP, RLDS>(in, Rplanes, slot, idx, R);
  // Branch-free over the legs on purpose: a wave of a one-round launch has its SIMD to itself and a leg's three sincos are
  // serial polynomial chains - interleaved they issue back to back, one leg at a time behind a "does any lane need this
  // leg" branch they ran at ~10 cycles per instruction (8.8 k cycles for the four legs, 6.2 k now).
  LegTrig tr[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const double qa[3] = {q[3 * i], q[3 * i + 1], q[3 * i + 2]};
    tr[i] = leg_trig(qa);
  }
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int leg = leg0 + i;
    const bool stl = (stance >> leg) & 1u;
    const bool todo = !(have_swing && !stl);  // (a swing leg with swing references belongs to the swing tasks)
    const bool st = stl && st_out == QC_SOLVED;
    double f[3], fb[3], tau[3];
#pragma unroll
    for (int k = 0; k < 3; k++) f[k] = sout[(OUT_F + 3 * leg + k) * SP + slot];
#pragma unroll
    for (int r = 0; r < 3; r++) fb[r] = st ? -(R[r] * f[0] + R[3 + r] * f[1] + R[6 + r] * f[2]) : 0.0;  // BC.cpp:218-232
    if constexpr (RUNTIME_LEG) leg_jt_force(leg_geom(P, leg), tr[i], fb, tau);
    else leg_jt_force(P, i, tr[i], fb, tau);  // (compile-time leg: the constants are scalar operands)
    if (todo) store_tau(P, out, idx, leg, tau, st);
  }
}
template <int SP, bool RLDS, int G, bool STR>
QC_DEV void torque_pass(const DevParams* __restrict__ Pg, const BatchIn& in, const BatchOut& out, const double* __restrict__ sout, int out_n, int lane,
                        const double* __restrict__ Rplanes, TorquePre& Q, const double* __restrict__ Qplanes = nullptr) {
  QC_CLK_ABS(8, 13);
  unsigned char* const tl = reinterpret_cast<unsigned char*>(const_cast<double*>(sout) + OUT_PLANES * SP);  // (TASK_DOUBLES behind the planes)
  const bool have_swing = in.swing_pos || in.swing_state;
  const bool mine = lane < out_n;
  uint32_t stance = 0xFu;
  if (mine) stance = (uint32_t)((unsigned long long)__double_as_longlong(sout[OUT_WORD * SP + lane]) >> 32);
  // the swing-leg tasks of the wave, leg-major (neighbouring lanes touch neighbouring robots): one byte each, slot << 2 | leg
  int ns = 0;
  if (have_swing) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool sw = mine && !((stance >> i) & 1u);
      const unsigned long long ms = __builtin_amdgcn_ballot_w64(sw);
      if (sw) tl[ns + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ms >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ms, 0))] = (unsigned char)((lane << 2) | i);
      ns += __builtin_popcountll(ms);
    }
    __syncthreads();
  }
  // The joint angles requested at the top of the flush have to BE here before the swing inputs are requested: vector-memory
  // results return in order, so waiting for them later would wait for the (slower, L1-bypassing) swing loads as well.
#pragma unroll
  for (int k = 0; k < 12; k++) asm volatile("" : "+v"(Q.q[k]));
  SwingIn nxt;
  int nslot = 0;
  if (lane < ns) {  // the first swing pass's inputs: in flight while the stance legs are computed
    const int code = tl[lane];
    nslot = code >> 2;
    swing_fetch(in, __double_as_longlong(sout[OUT_IDX * SP + nslot]), code & 3,
                (uint32_t)((unsigned long long)__double_as_longlong(sout[OUT_WORD * SP + nslot]) >> 32), nxt, Qplanes ? Qplanes + nslot : nullptr, SP);
  }
  asm volatile("" ::: "memory");  // (... and requested above this line, not behind the stance legs' arithmetic)
  QC_CLK_ABS(13, 15);
  // stance legs (and the swing legs of a batch without swing references: zero torque), one robot per lane - or, with few
  // robots in a lane-group kernel, the legs of a robot spread over its group: tau = clamp(J^T f_body), kinematics.cpp:219-231,
  // commander_node.cpp:511-526
  {
    CParams& P = *QC_PARAMS_HERE(Pg);
    if (torque_by_group<G>(out_n)) {
      if constexpr (G > 1) {
        const int grp = lane_group<G, STR>(lane);
        if (grp < out_n) stance_legs<SP, RLDS, 4 / G, true>(P, in, out, sout, Rplanes, grp, (4 / G) * lane_member<G, STR>(lane), have_swing, Q.q);
      }
    } else if (mine) {
      stance_legs<SP, RLDS, 4, false>(P, in, out, sout, Rplanes, lane, 0, have_swing, Q.q);
    }
  }
  QC_CLK_ABS(15, 14);
#pragma unroll 1
  for (int t = lane; t < ns; t += 64) {
    const SwingIn cur = nxt;
    const int cslot = nslot;
    if (t + 64 < ns) {  // the next pass's inputs are requested before this pass computes
      const int code = tl[t + 64];
      nslot = code >> 2;
      swing_fetch(in, __double_as_longlong(sout[OUT_IDX * SP + nslot]), code & 3,
                  (uint32_t)((unsigned long long)__double_as_longlong(sout[OUT_WORD * SP + nslot]) >> 32), nxt, Qplanes ? Qplanes + nslot : nullptr, SP);
    }
    double R[9];
    task_rwb<SP, RLDS>(in, Rplanes, cslot, cur.idx, R);
    swing_leg_task(*QC_PARAMS_HERE(Pg), in, out, cur, R);
  }
  QC_CLK_ABS(14, 8);
}

// store the robots parked in the output stock: one per lane, or one per lane group when there are few
template <int G, bool KIN, bool STR, int SP, bool RLDS = false>
QC_DEV void flush_out(const DevParams* __restrict__ Pg, const BatchIn& in, const BatchOut& out, const double* __restrict__ sout, int out_n, int lane,
                      const double* __restrict__ Rplanes = nullptr, const double* __restrict__ Qplanes = nullptr) {
  TorquePre tq;
  if constexpr (KIN) {
    if (out.joint_tau) {
      torque_prefetch<SP, G, STR>(in, sout, out_n, lane, tq, Qplanes);
      asm volatile("" ::: "memory");  // (requested HERE: left alone, the compiler sinks these loads to their first use, behind the swing inputs)
    }
  }
  if (G > 1 && out_n <= 64 / G) {
    const int grp = lane_group<G, STR>(lane);
    if (grp < out_n) {
      CParams& P = *QC_PARAMS_HERE(Pg);
      store_from_stock<KIN, 4 / G, SP>(P, in, out, sout, grp, lane_member<G, STR>(lane));
    }
  } else if (lane < out_n) {
    CParams& P = *QC_PARAMS_HERE(Pg);
    store_from_stock<KIN, 4, SP, RLDS>(P, in, out, sout, lane, 0, RLDS ? Rplanes + lane : nullptr);
  }
  if constexpr (KIN) {
    if (out.joint_tau) torque_pass<SP, RLDS, G, STR>(Pg, in, out, sout, out_n, lane, Rplanes, tq, Qplanes);
  }
}

// record of one running robot in the re-pack area (37 doubles: odd, so the four lanes of a group read different
// banks): 0-5 b, 6 idx, 7 {iters, output slot, stance}, 8-19 r, 20-31 f, 32-35 face code per foot
constexpr int REPACK_RS = 37;
template <class LaneX>
QC_DEV void repack_write(const LaneX& L, double* __restrict__ rec, int slot, int member) {
  constexpr int FPL = LaneX::FPL;
  if (member == 0) {
#pragma unroll
    for (int k = 0; k < 6; k++) rec[k] = L.Wr.b[k];
    rec[6] = __longlong_as_double(L.idx);
    rec[7] = __longlong_as_double((long long)(((unsigned long long)(uint32_t)((L.iters << 8) | slot) << 32) | L.stance | L.polish_bit()));
  }
#pragma unroll
  for (int i = 0; i < FPL; i++) {
    const int ft = L.foot0 + i;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      rec[8 + 3 * ft + k] = L.Wr.r[i][k];
      rec[20 + 3 * ft + k] = L.f[3 * i + k];
    }
    rec[32 + ft] = __longlong_as_double((long long)encode_foot(L.C.sx[i], L.C.sy[i], L.C.sz[i]));
  }
}
// ... and into member j4 of a 4-lane group; returns the robot's slot in the output stock
template <class Lane4X, class Eqp4X>
QC_DEV int repack_read(const DevParams* __restrict__ Pg, Lane4X& L4, Eqp4X& eqp4, const double* __restrict__ rec, int j4) {
#pragma unroll
  for (int k = 0; k < 6; k++) {
    L4.Wr.b[k] = rec[k];  // (already -b: both sides are 6x6 forms)
    asm volatile("" : "+v"(L4.Wr.b[k]));
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    L4.Wr.r[0][k] = rec[8 + 3 * j4 + k];
    L4.f[k] = rec[20 + 3 * j4 + k];
  }
  const uint32_t fw = (uint32_t)__double_as_longlong(rec[32 + j4]);
  L4.C.sx[0] = dec2(fw); L4.C.sy[0] = dec2(fw >> 2); L4.C.sz[0] = dec2(fw >> 4);
  L4.idx = __double_as_longlong(rec[6]);
  const unsigned long long fl = (unsigned long long)__double_as_longlong(rec[7]);
  L4.stance = (uint32_t)fl & ~QC_POLISH_ONE;
  L4.iters = (int)(fl >> 40);
  L4.foot0 = j4;
  L4.status = QC_MAX_ITER;
  {
    const double tol = Eqp4X::kTolScale * QC_PARAMS_HERE(Pg)->tol_d;
    L4.tol_s = ((uint32_t)fl & QC_POLISH_ONE) ? tol : -tol;
  }
  L4.have_f = true;
  eqp4.setup(*QC_PARAMS_HERE(Pg), L4.Wr, j4);
  return (int)((fl >> 32) & 0xFFu);
}

// One or two lanes per robot: once at most 16 robots of a wave are still running they fit a 4-lanes-per-robot layout,
// whose recalculation is much shorter (487 instructions against 730 at two lanes and ~1170 at one) - and the wave
// waits for exactly these stragglers.  The running robots are re-packed through the (now idle) input stock and finish
// on the G = 4 body; `slot` is where the robot's result goes in the output stock.  `bm` = ballot(busy), at most
// 16 robots.
// Two stages (strided layout): while more than 8 robots run, the classic body.  Then the survivors are re-packed once
// more, each into TWO groups 8 lanes apart that continue with different drop rules - most negative multiplier / all
// negative multipliers - and the first at the KKT point wins (the racing strategies of the mode-2 kernel, applied to
// the robots every other lane of the wave is waiting for).  Forking after ~6 recalculations still shortens the
// slowest robot's chain: 20 -> 16 on config 3's 16 384 first robots, mean of the per-wave maximum 11.5 -> 10.6
// (oracle/prototypes/proto_tail_fork.py); clamp-step variants add nothing at that point.
// HOLD (the joint_q one-lane kernel, whose re-pack records ALIAS the output stock - see QPARK in balance_kernel): no result is parked
// before the last read of a record; the classic stage's finishers keep theirs in registers (three doubles and a few words per lane) until
// the race stage is over.
template <bool KIN, bool UNIFORM, int SP, bool HOLD = false, class LaneG>
QC_DEV void finish_on_four_lanes(const DevParams* __restrict__ Pg, const LaneG& L, bool busy, unsigned long long bm, int slot, int member, int lane,
                                 double* __restrict__ sin, double* __restrict__ sout, const bool cold) {
  using Eqp4 = EqpDiagW<UNIFORM, 4, !QC_NO_STRIDED>;
  using Lane4 = Lane<Eqp4, KIN>;
  using LaneR = Lane<Eqp4, KIN, true>;  // with a per-lane drop rule
  constexpr bool STR4 = Eqp4::kStrided;
  constexpr int GS = LaneG::G;  // lanes per robot of the layout being left
  constexpr int RS = REPACK_RS;
  const int nb = __builtin_popcountll(bm) / GS;  // running robots
  if (nb == 0) return;
  QC_CLK_TAIL_BEGIN();
  static_assert(16 * RS <= (HOLD ? OUT_PLANES : IN_PLANES) * SP, "the re-pack records live in the idle input stock (HOLD: under the output stock)");
  const int rank2 = __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0)) / GS;
  __syncthreads();  // nobody reads the input stock any more
  if (busy) repack_write(L, sin + rank2 * RS, slot, member);
  __syncthreads();
  const int g4 = lane_group<4, STR4>(lane), j4 = lane_member<4, STR4>(lane);
  // (a warm-started robot that is not done after its first recalculation is a face or two away: nothing to race for;
  // measured neutral to +0.5 us on config 4's tick, tools/tail_race_scan.py)
  const bool race = STR4 && cold && QC_PARAMS_HERE(Pg)->tail_race != 0;
  int nrun = nb;  // robots the race stage starts with (the first re-pack's records if the classic stage is skipped)
  QC_CLK_TAIL_LOOP();
  Lane4 L4;
  bool held = false;  // HOLD: this group's classic-stage result is still in L4
  int slot_held = 0;
  if (!race || nb > 8) {
    Eqp4 eqp4(nullptr);
    bool busy4 = g4 < nb;
    // groups beyond the running robots shadow record 0: the strided layout keeps every lane in the loop (MFMA)
    const int slot4 = repack_read(Pg, L4, eqp4, sin + (busy4 || !STR4 ? g4 : 0) * RS, j4);
    const int stop = race ? 8 : 0;
    if constexpr (STR4 && UNIFORM) {
      // the tail has the registers to keep the recalculation's constants resident, as the mode-2 kernel does
      UConst uc = load_uconst(*QC_PARAMS_HERE(Pg));
      while (__builtin_popcountll(__builtin_amdgcn_ballot_w64(busy4) & 0xFFFFull) > stop) {
        QC_CLK(7, 2);
        pin_uconst(uc);
        const bool done = L4.template iterate<Lane4::STEADY>(uc, eqp4, busy4);
        busy4 = busy4 & !done;
      }
    } else if constexpr (STR4) {
      while (__builtin_popcountll(__builtin_amdgcn_ballot_w64(busy4) & 0xFFFFull) > stop) {
        QC_CLK(7, 2);
        const bool done = L4.template iterate<Lane4::STEADY>(*QC_PARAMS_HERE(Pg), eqp4, busy4);
        busy4 = busy4 & !done;
      }
    } else {
      while (busy4) busy4 = !L4.template iterate<Lane4::STEADY>(*QC_PARAMS_HERE(Pg), eqp4);
    }
    if constexpr (HOLD) {
      held = g4 < nb && !busy4;
      slot_held = slot4;
    } else {
      if (g4 < nb && !busy4) L4.template push_result<SP>(sout, slot4);
    }
    const unsigned run16 = (unsigned)(__builtin_amdgcn_ballot_w64(busy4) & 0xFFFFull);  // (member 0 of the strided groups = lanes 0-15)
    nrun = __builtin_popcount(run16);
    if (nrun == 0) {
      if constexpr (HOLD) {
        __syncthreads();  // every record has been read
        if (held) L4.template push_result<SP>(sout, slot_held);
      }
      QC_CLK_TAIL_END();
      return;
    }
    // second re-pack: the <= 8 survivors
    const int rank = __builtin_popcount(run16 & ((1u << g4) - 1u));
    __syncthreads();
    if (busy4) repack_write(L4, sin + rank * RS, slot4, j4);
    __syncthreads();
  }
  if constexpr (STR4) {
    LaneR LR;
    Eqp4 eqpR(nullptr);
    const int r = g4 & 7, sid = g4 >> 3;  // robot, strategy of this group
    bool busyR = r < nrun;
    const int slotR = repack_read(Pg, LR, eqpR, sin + (busyR ? r : 0) * RS, j4);
    LR.drop_all = sid == 1;
    unsigned solved_mask = 0;  // strategies of this lane's robot that have reached the KKT point
    auto after = [&](bool done) {
      const int mine = (busyR & done & (LR.status == QC_SOLVED)) ? (1 << sid) : 0;
      const int m = mine | __builtin_amdgcn_update_dpp(0, mine, 0x120 + 8, 0xF, 0xF, true);  // row_ror 8: the partner group
      solved_mask |= (unsigned)m;
      busyR = busyR & !done & (solved_mask == 0);
    };
    if constexpr (UNIFORM) {
      UConst uc = load_uconst(*QC_PARAMS_HERE(Pg));
      while (__builtin_amdgcn_ballot_w64(busyR) != 0) {
        QC_CLK(7, 2);
        pin_uconst(uc);
        after(LR.template iterate<LaneR::STEADY>(uc, eqpR, busyR));
      }
    } else {
      while (__builtin_amdgcn_ballot_w64(busyR) != 0) {
        QC_CLK(7, 2);
        after(LR.template iterate<LaneR::STEADY>(*QC_PARAMS_HERE(Pg), eqpR, busyR));
      }
    }
    QC_CLK_TAIL_END();
    // the winner - the lower-numbered strategy if both got there in the same recalculation, strategy 0 with whatever
    // status it has if none did - parks the result
    const int win = solved_mask ? __builtin_ctz(solved_mask) : 0;
    if constexpr (HOLD) __syncthreads();  // (the race stage's records have been read by every group)
    if (r < nrun && sid == win) LR.template push_result<SP>(sout, slotR);
  }
  if constexpr (HOLD) {
    if constexpr (!STR4) __syncthreads();
    if (held) L4.template push_result<SP>(sout, slot_held);
  }
}

// How many of a cold-started robot's first recalculations are CLAMP steps (project the equality-constrained minimiser
// of the current working set into the frusta, keep the faces it hits) before the ratio-test steps start.  One is the
// classic start.  More of them build the working set several faces at a time and cost less than a ratio-test step:
// five are the measured optimum where the batch fills the chip (one or two lanes per robot: 262 144 robots 104 -> 95 us,
// 1 M robots 345 -> 291 us; more than six start to cycle - profiles/r02_clamp_scan.log); the 4-lane kernels (chain-bound
// batches) do not care, and a warm-started robot already has its working set: further clamp steps only disturb it.
template <int G>
QC_DEV int clamp_steps_for(CParams& P, const uint32_t* warm) {
  const int tuned = P.clamp_steps;  // qc_set_tuning "clamp_steps"; 0 = this rule
  return tuned > 0 ? tuned : ((warm != nullptr || G == 4) ? 1 : 5);
}

// MODE 0: persistent waves (chunks of many fills, lane refill).  MODE 1: the launch gives every wave at most one
// fill (chunk <= 64 / G).  MODE 2: one fill and the SIMD to itself, recalculation constants resident in VGPRs.
// RACE (strided 4-lane one-fill kernels): strategies racing per robot, 1, 2 or 4; a wave then holds 16 / RACE robots (see Lane).

// The kernel's argument block as the launch lays it out (kernarg segment), for re-deriving BatchIn / BatchOut AT THEIR USE in the
// joint_q kernels: 23 array pointers held in SGPRs from the kernel's entry to its flush - across a solve that needs none of
// them - do not fit the 102-SGPR file next to the loop's masks, and the compiler parks them in VGPR lanes (108-175 spilled
// SGPRs, ~250 v_readlane per wave: profiles/r05_kernel_resources.txt).  Read again from the (constant, scalar-cached) kernarg
// segment where they are used, behind an opaque pointer so that they are not kept live in between, they cost a few s_loads.
struct BalanceKernelArgs {
  const DevParams* Pg;
  long n;
  BatchIn in;
  const uint32_t* warm;
  BatchOut out;
  long chunk;
  int refill_t;
};
template <class T>
QC_DEV const T& kernarg_here(unsigned off) {
  const __attribute__((address_space(4))) char* p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  p += off;
  asm volatile("" : "+s"(p));
  return *(const T*)(const __attribute__((address_space(4))) T*)p;
}

template <class Eqp, bool KIN, int MIN_WAVES_PER_SIMD, int MODE = 0, int RACE = 1>
__global__ __launch_bounds__(64, MIN_WAVES_PER_SIMD) void balance_kernel(const DevParams* __restrict__ Pg, const long n, const BatchIn in_k,
                                                                         const uint32_t* __restrict__ warm, const BatchOut out_k, const long chunk,
                                                                         const int refill_t) {
  static_assert(RACE == 1 || (MODE != 0 && Eqp::kStrided && Eqp::G == 4), "racing strategies exist for the strided one-fill kernels");
  constexpr int G = Eqp::G;
  // (KIN: re-read from the kernarg segment at every use, see BalanceKernelArgs; the QP-only kernels hold 14 pointers and keep them,
  // and so does the one-wave-per-SIMD kernel of chain-bound batches - a lone wave waits out every scalar-load round trip: the fused
  // tick on 4 096 robots 27.36 -> 27.55 us with it, profiles/r06_tick_ab.log)
  constexpr bool REREAD = KIN && MODE != 2;
  auto IN = [&]() -> const BatchIn& {
    if constexpr (REREAD) return kernarg_here<BatchIn>((unsigned)offsetof(BalanceKernelArgs, in));
    else return in_k;
  };
  auto OUT = [&]() -> const BatchOut& {
    if constexpr (REREAD) return kernarg_here<BatchOut>((unsigned)offsetof(BalanceKernelArgs, out));
    else return out_k;
  };
  extern __shared__ __attribute__((aligned(16))) double qc_lds[];  // [stock planes][64] (+ the dense form's 78 Hessian planes)
  constexpr int SP = stock_slots(G, MODE) + 1;  // plane stride of this mode's stock
  // One-lane dense form as one-fill workgroups (round 5): the 78 Hessian planes (39 936 B) ARE the workgroup's LDS - four
  // workgroups per CU, one per SIMD.  With the stock in front of them (58.3 KB) a CU held two, i.e. every other SIMD idled
  // behind the 512-register kernel.  The input stock is not used on this path (one lane assembles and solves its robot),
  // Rwb is read again for the output transform instead of parked, and the output stock aliases the first Hessian planes
  // once every robot of the wave is finished.
  constexpr bool HESS_ONLY = Eqp::kHessianInLds && MODE != 0;
  static_assert(!HESS_ONLY || (OUT_PLANES * SP + TASK_DOUBLES <= 78 * 64), "the output stock fits the Hessian planes it aliases");
  // (see QPARK_* above: Rwb and the joint angles parked, the re-pack records under the output stock)
  constexpr bool QPARK = KIN && G == 1 && MODE == 1 && !Eqp::kHessianInLds;
  static_assert(!QPARK || SP == 65, "the QPARK layout is laid out for 64 slots");
  double* const sin = qc_lds;
  double* const sout = HESS_ONLY ? qc_lds : qc_lds + (QPARK ? QPARK_OUT : IN_PLANES) * SP;
  // XCD-aware workgroup -> chunk map.  The dispatcher places workgroup b on XCD b % 8 (observed, MI355X_MICROARCH.md "Workgroup
  // dispatch"; a speed assumption only - any placement computes the same robots), each XCD with an L2 of its own.  A racing wave holds
  // 4 (or 8) robots: 288 contiguous bytes per 72-byte-row array, so neighbouring waves share the 128-byte lines their rows straddle,
  // and with the identity map - neighbours on different XCDs - every shared line is fetched from HBM once per XCD: 1.6x the algorithmic
  // bytes on config 2 (profiles/r02 ... r04_cfg2).  Sixteen robots are line-aligned in every array (16 x 72 B = 9 lines), so the waves
  // of one aligned 16-robot group are mapped to ONE XCD and the groups go round the XCDs as before.  (Mapping whole contiguous eighths of
  // the batch to an XCD removes the same re-fetches but costs config 2 0.2 us, profiles/r05_ab_xcd_map.log: the spreading of
  // neighbouring groups over the XCDs is worth keeping.)  Waves of 16 robots or more share no line: identity.
  unsigned chunk_id = blockIdx.x;
#ifndef QC_NO_XCD_MAP  // (development: -DQC_NO_XCD_MAP compiles the identity map for A/B runs)
  if constexpr (RACE > 1) {  // 16 / RACE robots per wave: RACE waves per aligned group (compile-time powers of two: shifts and masks only -
    // with run-time divisions the map cost the wave's first 0.1 us more than it saved)
    constexpr unsigned gw = (unsigned)RACE, super = 8u * gw;
    const unsigned full = gridDim.x & ~(super - 1u);
    if (blockIdx.x < full && chunk == 16 / RACE) {
      const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
      chunk_id = (slot / gw) * super + xcd * gw + (slot % gw);
    }
  }
#endif
  long cursor = (long)chunk_id * chunk;  // wave-uniform: next robot of this wave's chunk to assemble
  const long end = cursor + chunk < n ? cursor + chunk : n;
  const int lane = threadIdx.x;
  constexpr bool STR = Eqp::kStrided;  // lane layout of a group (one-fill modes only)
  static_assert(!STR || MODE != 0, "the refill bookkeeping of the persistent mode assumes adjacent lanes");
  const int member = lane_member<G, STR>(lane);
  int stock_n = 0, stock_next = 0;  // input stock: slots [stock_next, stock_n) hold assembled robots
  int out_n = 0;                    // output stock: slots [0, out_n) hold finished results
  // one-lane dense form, QP only: stragglers fork into idle lanes with the other drop rule.  (Not the joint_q kernel: with the results
  // parked in the lanes' own dead Hessian columns for the torque pass - tools/experiments/r06_dense_kin_twin.patch - the race bought its
  // tick nothing, 151.4 us with it, 151.4 with race = 0, 147.8 without the machinery: profiles/r06_dense_tick.log.  That kernel's time
  // is its fill and flush at one wave per SIMD, not its stragglers.)
  constexpr bool TWIN = Eqp::kHessianInLds && MODE != 0 && !KIN;
  Lane<Eqp, KIN, (RACE > 1) || TWIN> L;
  L.idx = -1;
  L.foot0 = member * (4 / G);
  Eqp eqp(qc_lds + (HESS_ONLY ? 0 : stock_doubles(stock_slots(G, MODE))) + lane);
  bool busy = false;  // group holds an unfinished robot
  QC_CLK_BEGIN();
  if constexpr (MODE != 0) {
    // One fill per wave: the persistent-wave machinery (refill, stock cursors, a result push after every
    // recalculation) is dead weight on the serial chain that bounds such a batch.  One fill, a divergent solve
    // loop, one push, one flush.
    constexpr bool RESIDENT = MODE == 2;
    if (cursor >= end) return;
    UConst uc;  // RESIDENT: the recalculation's constants live in VGPRs (no scalar load + wait per recalculation)
    if constexpr (RESIDENT) uc = load_uconst(*QC_PARAMS_HERE(Pg));
    // One lane per robot: the lane that assembles a robot is the lane that solves it, so the robot goes straight from the
    // assembly into the solver's registers - no input stock, no barrier - and its Rwb is parked in the first nine planes of
    // the (otherwise idle) input-stock area for the output transform; the planes behind them serve as the re-pack area of the tail.
    constexpr bool DIRECT = G == 1;
    constexpr bool RPARK = DIRECT && !HESS_ONLY;  // Rwb parked in LDS for the output transform
    constexpr int R_PLANES = RPARK ? 9 : 0;
    static_assert(!DIRECT || QPARK || (R_PLANES * SP + 16 * REPACK_RS <= IN_PLANES * SP), "Rwb planes + re-pack records fit the input-stock area");
    if constexpr (DIRECT) {
      const long left = end - cursor;
      stock_n = left < 64 ? (int)left : 64;
      if (lane < stock_n) {
        // Everything the fill reads from memory is issued back to back - the contact bytes and the warm-start word first
        // (behind wave-uniform null tests), then the 48 doubles of the state - and nothing dependent sits in between: a
        // wave's fill used to chain four to five vector-memory round trips (two batches of state, contact bytes, warm word,
        // Rwb again: 2.5 us from kernel entry to "inputs landed" for a lone wave, profiles/r03_timeline.log).
        const BatchIn& ia = IN();
        const uint32_t* wp = warm;
        const long robot = cursor + lane;
        const uint32_t sw = ia.stance ? *reinterpret_cast<const uint32_t*>(ia.stance + 4 * robot) : 0u;
        const uint32_t wv = wp ? wp[robot] : 0u;
        RawState S;
        TickExtra X;
        double fp[12];
        fetch_extra<KIN>(ia, robot, X);
        fetch_state<4, KIN>(ia, robot, 0, S, fp);
        asm volatile("" ::: "memory");  // (the loads above stay above)
        CParams& P = *QC_PARAMS_HERE(Pg);
        Wrench<4> W;
        const uint32_t st = assemble_from_state<KIN, 4, false>(P, ia, robot, 0, S, fp, sw, X, W);
        if constexpr (RPARK) {
#pragma unroll
          for (int k = 0; k < 9; k++) sin[k * SP + lane] = S.R[k];
        }
        if constexpr (QPARK) {  // (fp = the twelve joint angles fetch_state read)
#pragma unroll
          for (int k = 0; k < 12; k++) sin[(QPARK_Q + k) * SP + lane] = fp[k];
        }
        L.load_direct(W, st, wv, robot, 0, P.tol_start);
      } else if constexpr (TWIN) {
        // The wave-uniform loops below run every lane through the recalculation body, the lanes beyond a ragged last fill
        // included: they carry a defined robot - no wrench, no stance foot (every foot eliminated, nothing to iterate on) -
        // instead of whatever the registers held.  They never store.  (Shadowing robot 0 of the wave, as the strided kernels'
        // spare groups do, costs this 512-register kernel 12 B of scratch.)
        Wrench<4> W;
#pragma unroll
        for (int k = 0; k < 6; k++) W.b[k] = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) W.r[i][0] = W.r[i][1] = W.r[i][2] = 0.0;
        L.load_direct(W, 0u, 0u, -1, 0, 0.0);
      }
    } else {
      stock_n = restock<G, KIN, STR, SP>(Pg, IN(), warm, cursor, end, lane, member, sin);
    }
    const int grp = lane_group<G, STR>(lane);
    busy = grp < stock_n;
    // measurement probe (qc_set_tuning "probe_batch_load"): load -> assemble -> store only, no recalculation
    const bool probe = QC_PARAMS_HERE(Pg)->max_iter == 0;
    using LaneT = Lane<Eqp, KIN, (RACE > 1) || TWIN>;
    if constexpr (STR) {
      // Strided layout: the group sums run on the matrix pipe, and an MFMA reads its operands from ALL 64 lanes
      // whatever EXEC says - the all-ones A operand included, which the compiler materialises under the current
      // EXEC.  So nothing here may run under a partial EXEC: every lane carries a robot (groups beyond the fill
      // shadow robot 0) through a wave-uniform loop, and only the lanes of running robots commit what a
      // recalculation produced.
      // RACE > 1: the wave holds 16 / RACE robots; group grp solves robot grp % ROB with strategy sid
      constexpr int ROB = 16 / RACE;
      const int slot = grp & (ROB - 1);
      const int sid = (grp / ROB) * (4 / RACE);  // 4 strategies: 0 .. 3; 2: {0, 2}
      if constexpr (RACE > 1) {
        busy = slot < stock_n;
        // (clamp steps, drop rule) = (1, most negative) (1, all) (2, all) (3, most negative): over eight 4 096-robot
        // batches the slowest robot takes 13.25 recalculations on average against 16.9 with the first strategy alone
        // (13.9 for the pair {0, 2} a 2-way race runs); oracle/prototypes/proto_race_strategies.py
        L.nclamp = sid == 3 ? 3 : (sid == 2 ? 2 : 1);
        L.drop_all = sid == 1 || sid == 2;
      }
      double tol0;  // (RESIDENT: from the register copy - another scalar load here is a memory round trip a lone wave waits out)
      if constexpr (RESIDENT) tol0 = uc.tol_start;
      else tol0 = QC_PARAMS_HERE(Pg)->tol_start;
      L.template load_from_stock<SP>(sin, busy ? slot : 0, member, tol0);
      eqp.setup(*QC_PARAMS_HERE(Pg), L.Wr, L.foot0);
      busy = busy && !probe;
      unsigned solved_mask = 0;  // RACE: strategies of this lane's robot that have reached the KKT point
      // the robot is finished as soon as one strategy has solved it; the others stop with it
      auto after = [&](bool done) {
        if constexpr (RACE > 1) {
          const int mine = (busy & done & (L.status == QC_SOLVED)) ? (1 << sid) : 0;
          int m = mine | __builtin_amdgcn_update_dpp(0, mine, 0x120 + ROB, 0xF, 0xF, true);  // row_ror by ROB lanes: the partner groups
          if constexpr (RACE == 4) {
            m |= __builtin_amdgcn_update_dpp(0, mine, 0x120 + 2 * ROB, 0xF, 0xF, true);
            m |= __builtin_amdgcn_update_dpp(0, mine, 0x120 + 3 * ROB, 0xF, 0xF, true);
          }
          solved_mask |= (unsigned)m;
          busy = busy & !done & (solved_mask == 0);
        } else {
          busy = busy & !done;
        }
      };
      QC_CLK(0, 2);
      // clamp steps: one for the racing lanes (their strategies take more in the MIXED recalculations below)
      const int first_steps = RACE > 1 ? 1 : clamp_steps_for<G>(*QC_PARAMS_HERE(Pg), warm);
#pragma unroll 1
      for (int k = 0; k < first_steps && (k == 0 || __builtin_amdgcn_ballot_w64(busy) != 0); k++) {
        if (k) QC_CLK(7, 2);
        bool done;
        if constexpr (RESIDENT) {
          pin_uconst(uc);
          done = L.template iterate<LaneT::FIRST>(uc, eqp, busy, k == 0 && warm == nullptr);
        } else {
          done = L.template iterate<LaneT::FIRST>(*QC_PARAMS_HERE(Pg), eqp, busy, k == 0 && warm == nullptr);
        }
        after(done);
      }
      if constexpr (RACE > 1) {  // recalculations 2 and 3: clamp steps for the strategies that take them, ordinary steps for the others
#pragma unroll 1
        for (int it = 2; it <= 3 && __builtin_amdgcn_ballot_w64(busy) != 0; it++) {
          QC_CLK(7, 2);
          bool done;
          if constexpr (RESIDENT) {
            pin_uconst(uc);
            done = L.template iterate<LaneT::MIXED>(uc, eqp, busy);
          } else {
            done = L.template iterate<LaneT::MIXED>(*QC_PARAMS_HERE(Pg), eqp, busy);
          }
          after(done);
        }
      }
      while (__builtin_amdgcn_ballot_w64(busy) != 0) {
        QC_CLK(7, 2);
        bool done;
        if constexpr (RESIDENT) {
          pin_uconst(uc);
          done = L.template iterate<LaneT::STEADY>(uc, eqp, busy);
        } else {
          done = L.template iterate<LaneT::STEADY>(*QC_PARAMS_HERE(Pg), eqp, busy);
        }
        after(done);
      }
      if constexpr (RACE > 1) {
        // the winner - the lowest-numbered strategy among those that solved the robot in the deciding
        // recalculation, or strategy 0 with whatever status it has if none did - parks the result
        const int win = solved_mask ? __builtin_ctz(solved_mask) : 0;
        QC_CLK(7, 8);
        if (slot < stock_n && sid == win) L.template push_result<SP>(sout, slot);
        __syncthreads();
        flush_out<Eqp::G, KIN, STR, SP>(Pg, IN(), OUT(), sout, stock_n, lane);
        QC_CLK_END(8);
        return;
      }
      QC_CLK(7, 8);
      if (grp < stock_n) L.template push_result<SP>(sout, grp);
      __syncthreads();
      flush_out<Eqp::G, KIN, STR, SP>(Pg, IN(), OUT(), sout, stock_n, lane);
      QC_CLK_END(8);
      return;
    }
    if (busy || TWIN) {  // (TWIN: the shadow lanes of a ragged fill get their Hessian column too)
      if constexpr (!DIRECT) L.template load_from_stock<SP>(sin, grp, member, QC_PARAMS_HERE(Pg)->tol_start);
      eqp.setup(*QC_PARAMS_HERE(Pg), L.Wr, L.foot0);
    }
    const bool mine = busy;
    busy = busy && !probe;
    QC_CLK(0, 2);
    const int first_steps = clamp_steps_for<G>(*QC_PARAMS_HERE(Pg), warm);
#pragma unroll 1
    for (int k = 0; k < first_steps; k++) {
      if (k) QC_CLK(7, 2);
      if (busy) {  // every robot of a one-fill wave is fresh exactly once: the clamp steps are peeled
        if constexpr (RESIDENT) {
          pin_uconst(uc);
          busy = !L.template iterate<LaneT::FIRST>(uc, eqp, true, k == 0 && warm == nullptr);
        } else {
          busy = !L.template iterate<LaneT::FIRST>(*QC_PARAMS_HERE(Pg), eqp, true, k == 0 && warm == nullptr);
        }
      }
    }
    if constexpr (Eqp::kRepackTail) {
      unsigned long long bm = __builtin_amdgcn_ballot_w64(busy);
      while (__builtin_popcountll(bm) > 16 * G) {  // more than 16 robots still running
        QC_CLK(7, 2);
        if (busy) busy = !L.template iterate<LaneT::STEADY>(*QC_PARAMS_HERE(Pg), eqp);
        bm = __builtin_amdgcn_ballot_w64(busy);
      }
      if constexpr (QPARK) {
        // the re-pack records live under the output stock: the tail parks its results after its last record read, and the robots that
        // finished on the one-lane body after the tail (their results wait in this lane's registers)
        finish_on_four_lanes<KIN, Eqp::kUniform, SP, true>(Pg, L, busy, bm, grp, member, lane, sout, sout, warm == nullptr);
        if (!busy && mine) L.template push_result<SP>(sout, grp);
      } else {
        if (!busy && mine) L.template push_result<SP>(sout, grp);  // finished in the one- / two-lane layout
        finish_on_four_lanes<KIN, Eqp::kUniform, SP>(Pg, L, busy, bm, grp, member, lane, sin + R_PLANES * SP, sout, warm == nullptr);
      }
    } else if constexpr (TWIN) {
      // The one-lane dense form has no 4-lane tail (the exchange tile of the 4-lane dense body does not fit next to the Hessian
      // planes), so a wave walks its one-lane body until its slowest robot is done - at half the lanes or fewer busy for most of
      // that walk.  Those idle lanes race: once at most 32 robots still run, every running robot is copied into an idle lane
      // (working set, point, c; the 78 Hessian entries are READ from the owner's LDS column by both - nobody writes them after
      // setup) which continues with the OTHER drop rule - all negative multipliers at once instead of the most negative one -
      // and whoever reaches the KKT point first ends both (the tail race of the 6x6 forms, profiles/r02_tail_race_scan.log, with
      // lanes instead of lane groups).  Cold batches only: a warm-started robot that is not done after its first recalculation
      // is a face or two away.
      unsigned long long bm = __builtin_amdgcn_ballot_w64(busy);
      const bool fork_ok = warm == nullptr && QC_PARAMS_HERE(Pg)->tail_race != 0;
      // (both loops are wave-uniform and run every lane with `live` = busy - see Lane::iterate on why a finished or empty lane may
      // keep computing: a per-lane branch around the 2 600-instruction body costs exec-mask saves in SGPR pairs, which spill here)
      while (__builtin_popcountll(bm) > (fork_ok ? 32 : 0)) {
        const bool done = L.template iterate<LaneT::STEADY>(*QC_PARAMS_HERE(Pg), eqp, busy);
        busy = busy & !done;
        bm = __builtin_amdgcn_ballot_w64(busy);
      }
      int partner = lane;   // the lane racing on the same robot (itself: no race)
      // per-lane flags of the race, packed into one VGPR (as separate bools they live in SGPR pairs across the 2 600-instruction
      // body, and SGPR pairs that spill leave frame slots): bit 0 this lane still holds a result (its own robot's) that has to
      // reach the output, bit 1 it is a twin, bit 2 it brought its robot to the KKT point
      int role = mine ? 1 : 0;
      if (bm != 0) {
        // lanes whose own robot is finished store it straight from their registers (the output stock aliases Hessian planes that
        // are still being read; Rwb comes from memory) - their registers are about to carry somebody else's robot
        if (mine && !busy) {
          CParams& P = *QC_PARAMS_HERE(Pg);
          store_result<KIN, 4, false>(P, IN(), OUT(), L.idx, L.stance, L.status, L.iters, L.word_bits() | 0x80000000u, L.f, 0);
          role = 0;
        }
        const int nb = __builtin_popcountll(bm);
        const unsigned long long idle = ~bm;
        const int my_rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)((busy ? bm : idle) >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)(busy ? bm : idle), 0));
        // lane id of the rank-th set bit of m (ranks from 0; m has more than `rank` bits set): binary search on popcounts
        auto nth = [](unsigned long long m, int rank) {
          int pos = 0;
#pragma unroll
          for (int w = 32; w >= 1; w >>= 1) {
            const int c = __builtin_popcountll(m & ((1ull << w) - 1ull));
            const bool up = rank >= c;
            m = up ? (m >> w) : m;
            rank -= up ? c : 0;
            pos += up ? w : 0;
          }
          return pos;
        };
        const bool twin = !busy && my_rank < nb;  // the my_rank-th idle lane takes the my_rank-th running robot
        if (busy) partner = nth(idle, my_rank);
        if (twin) partner = nth(bm, my_rank);
        // the owner's state, read by its twin (every lane executes the shuffles; only twins keep what they read)
        const int src = twin ? partner : lane;
        auto take_i = [&](int v) { return __builtin_amdgcn_ds_bpermute(src << 2, v); };
        auto take_d = [&](double v) {
          const unsigned long long u = (unsigned long long)__double_as_longlong(v);
          const unsigned lo = (unsigned)take_i((int)(unsigned)u), hi = (unsigned)take_i((int)(unsigned)(u >> 32));
          return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        };
        const uint32_t wbits = (uint32_t)take_i((int)L.word_bits());
        const uint32_t stance_o = (uint32_t)take_i((int)L.stance);
        const int iters_o = take_i(L.iters);
        const double tol_o = take_d(L.tol_s);
        const int idx_lo = take_i((int)(unsigned)(unsigned long long)L.idx), idx_hi = take_i((int)(unsigned)((unsigned long long)L.idx >> 32));
        // (one value at a time, selected in place: two dozen temporaries next to the 78-entry factor's registers would spill)
#pragma unroll
        for (int k = 0; k < 12; k++) {
          const double tf = take_d(L.f[k]);
          L.f[k] = twin ? tf : L.f[k];
          const double tc = take_d(eqp.c[k]);
          eqp.c[k] = twin ? tc : eqp.c[k];
        }
        if (twin) {
          L.stance = stance_o;
          L.iters = iters_o;
          L.idx = (long)(((unsigned long long)(unsigned)idx_hi << 32) | (unsigned)idx_lo);
          L.status = QC_MAX_ITER;
          L.tol_s = tol_o;
          L.have_f = true;
          L.drop_all = true;
          L.foot0 = 0;
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const uint32_t fb = wbits >> (6 * i);
            L.C.sx[i] = dec2(fb); L.C.sy[i] = dec2(fb >> 2); L.C.sz[i] = dec2(fb >> 4);
          }
          // the robot's Hessian: the twin reads the owner's LDS column (read-only after setup; same address in both lanes = a broadcast)
          eqp.Qs = eqp.Qs - lane + partner;
          role = 2;
          busy = true;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      asm volatile("" : "+v"(role), "+v"(partner));
      while (__builtin_amdgcn_ballot_w64(busy) != 0) {
        const bool done = L.template iterate<LaneT::STEADY>(*QC_PARAMS_HERE(Pg), eqp, busy);
        asm volatile("" : "+v"(role), "+v"(partner));
        const int solved = (busy && done && L.status == QC_SOLVED) ? 1 : 0;
        const int fin = (busy && done) ? 1 : 0;
        const int theirs = __builtin_amdgcn_ds_bpermute(partner << 2, solved);
        // a tie goes to the owner (the classic rule); a twin stops as soon as its owner is finished in any way
        const int owner_done = __builtin_amdgcn_ds_bpermute(partner << 2, fin);
        const bool is_twin = (role & 2) != 0;
        role |= (solved && !(is_twin && theirs)) ? 4 : 0;
        busy = busy && !done && !(partner != lane && theirs) && !(is_twin && owner_done);
      }
      // who stores: the owner, unless its twin got to the KKT point first (then the twin)
      const int twin_won = __builtin_amdgcn_ds_bpermute(partner << 2, ((role & 6) == 6) ? 1 : 0);
      const bool store_own = (role & 1) && !(partner != lane && twin_won);
      if (store_own || (role & 6) == 6) {  // straight from the registers, like the lanes that finished before the fork: no output stock on this path
        CParams& P = *QC_PARAMS_HERE(Pg);
        store_result<KIN, 4, false>(P, IN(), OUT(), L.idx, L.stance, L.status, L.iters, L.word_bits() | 0x80000000u, L.f, 0);
      }
      QC_CLK_END(8);
      return;
    } else {
      while (busy) {
        if constexpr (RESIDENT) {
          pin_uconst(uc);
          busy = !L.template iterate<LaneT::STEADY>(uc, eqp);
        } else {
          busy = !L.template iterate<LaneT::STEADY>(*QC_PARAMS_HERE(Pg), eqp);
        }
      }
      if constexpr (HESS_ONLY) __syncthreads();  // (the dense joint_q kernel) every lane is past its last read of the Hessian planes the output stock aliases
      if (mine) L.template push_result<SP>(sout, grp);
    }
    QC_CLK(7, 8);
    __syncthreads();
    flush_out<Eqp::G, KIN, STR, SP, RPARK>(Pg, IN(), OUT(), sout, stock_n, lane, RPARK ? sin : nullptr, QPARK ? sin + QPARK_Q * SP : nullptr);
    QC_CLK_END(8);
    return;
  }
  // The first restock runs before any solver state is live, so (unlike its copy
  // inside the loop) it needs no spills: batches that fit one fill per wave
  // never execute the in-loop copy.
  if (cursor < end) {
    stock_n = restock<G, KIN, false, SP>(Pg, IN(), warm, cursor, end, lane, member, sin);
    cursor += stock_n;
  }
  QC_CLK(0, 1);
  for (;;) {
    const unsigned long long busy_mask = __builtin_amdgcn_ballot_w64(busy);
    const int n_free = (64 - __builtin_popcountll(busy_mask)) / G;  // free groups
    const long avail = (end - cursor) + (long)(stock_n - stock_next);
    if (avail > 0 && (n_free >= refill_t || busy_mask == 0)) {
      if (stock_next == stock_n) {
        stock_n = restock<G, KIN, false, SP>(Pg, IN(), warm, cursor, end, lane, member, sin);
        stock_next = 0;
        cursor += stock_n;
      }
      const int have = stock_n - stock_next;
      const int take = n_free < have ? n_free : have;
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(~busy_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)~busy_mask, 0)) / G;
      if (!busy && rank < take) {
        L.template load_from_stock<SP>(sin, stock_next + rank, member, QC_PARAMS_HERE(Pg)->tol_start);
        CParams& P = *QC_PARAMS_HERE(Pg);
        eqp.setup(P, L.Wr, L.foot0);
        busy = true;
      }
      stock_next += take;
      continue;
    }
    if (busy_mask == 0) break;
    bool fin = false;
    if (busy) {
      asm volatile("; QC_ITER_BEGIN");
      QC_CLK(1, 2);
      CParams& P = *QC_PARAMS_HERE(Pg);
      if (P.max_iter == 0) fin = true;  // measurement probe: load -> assemble -> store only (wave-uniform branch)
      else fin = L.iterate(P, eqp);
      QC_CLK(7, 1);
      asm volatile("; QC_ITER_END");
    }
    const unsigned long long fin_mask = __builtin_amdgcn_ballot_w64(fin);
    if (fin_mask != 0) {
      const int n_fin = __builtin_popcountll(fin_mask) / G;
      if (out_n + n_fin > 64) {  // dense flush of the output stock, one robot per lane
        __syncthreads();
        flush_out<Eqp::G, KIN, false, SP>(Pg, IN(), OUT(), sout, out_n, lane);
        __syncthreads();
        out_n = 0;
      }
      if (fin) {
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(fin_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fin_mask, 0)) / G;
        L.template push_result<SP>(sout, out_n + rank);
        busy = false;
      }
      out_n += n_fin;
    }
  }
  __syncthreads();
  QC_CLK(1, 8);
  flush_out<Eqp::G, KIN, false, SP>(Pg, IN(), OUT(), sout, out_n, lane);
  QC_CLK_END(8);
}



// ----------------------------------------------------------------------------------------------------------------
// MODE 3, paired waves (round 3): the one-lane kernel of batches that need more than one round of workgroups.
//
// A workgroup is TWO waves = 128 consecutive robots.  Each wave fills and takes its clamp steps exactly as the one-fill
// kernel does (one lane per robot, every load in one batch, Rwb parked in LDS), stores the robots that are finished straight
// from its registers, and writes the state of those still running - 33 doubles each - to a RECORD list in the
// workgroup's LDS.  Then it bumps an LDS counter: the wave that arrives FIRST simply ends (its wave slot is free for
// the next workgroup; nothing ever waits at a barrier after the first microsecond), the one that arrives LAST finishes
// both waves' stragglers four lanes per robot on the strided MFMA body, sixteen at a time, REFILLING lane groups from
// the list as robots finish, and races the two drop rules on the last <= 8 (cold batches).
// Why: in the one-fill kernel every wave finishes its own <= 16 stragglers and stays until the slowest is done - 5.65 tail
// recalculations per wave on config 5, mostly with a few of the 16 lane groups live.  Two waves' stragglers (~38 after
// five clamp steps) keep the 16 groups full for most of the walk: 4.8 tail recalculations per wave in the prototype
// (oracle/prototypes/proto_straggler_queue.py, K = 2) and no one-lane recalculation at 30 % occupancy in between.
// The same idea through a list in GLOBAL memory shared by K workgroups lost (tools/experiments/straggler_queue.patch: a
// memory round trip per refill, and a K-waves-long serial chain at the end of the launch); here a refill is a handful
// of ds_reads and the chain is two waves' worth.  And as ONE persistent wave that alternates fills and list consumption
// (tools/experiments/straggler_list_persistent.patch) it lost to code generation: a loop around the two heavy phases makes
// the compiler hoist invariants above bodies that need all 256 registers - scratch, whose first touch costs a wave ~10 us.
// Measured (profiles/r03_paired_waves.log): -5 % from four rounds of workgroups on (524 288 robots: 164 -> 156 us, 1 M: 300 ->
// 285, 2 M: 545 -> 515), nothing at two rounds - the early wave's slot idles until a second one is free for the next
// two-wave workgroup - which is where the planner draws the line.
constexpr int PAIR_REC = 33;    // doubles per record (odd: the records of the 16 lane groups fall into different banks):
                                // 0-5 -b, 6 {slot | stance << 8 | face codes << 24 | iters << 48}, 7-18 r, 19-30 f
constexpr int PAIR_CAP = 32;    // records a wave may leave (hand-over threshold <= PAIR_CAP)
struct PairLds {
  double Rrows[128 * 9];                 // Rwb of the workgroup's robots, for the output transform
  double rec[2 * PAIR_CAP * PAIR_REC];   // the record list; once it is drained, the race stage's re-pack area
  int count[2], arrived;  // records in the list (paired: count[0]; solo: one list per wave)
};

template <class Eqp, bool KIN>
__global__ __launch_bounds__(128, 2) void balance_pair_kernel(const DevParams* __restrict__ Pg, const long n, const BatchIn in,
                                                              const uint32_t* __restrict__ warm, const BatchOut out, const int th,
                                                              const int refill, const int race, const unsigned solo_from) {
  static_assert(Eqp::G == 1 && Eqp::kNegB, "one lane per robot, 6x6 forms");
  static_assert(8 * REPACK_RS <= PAIR_CAP * PAIR_REC, "the race stage re-packs into the drained list (a solo wave's half of it)");
  constexpr bool UNIFORM = Eqp::kUniform;
  using Lane1 = Lane<Eqp, KIN>;
  using Eqp4 = EqpDiagW<UNIFORM, 4, true>;
  using Lane4 = Lane<Eqp4, KIN>;
  using LaneR = Lane<Eqp4, KIN, true>;  // with a per-lane drop rule
  __shared__ PairLds lds;
  const int lane = threadIdx.x & 63;
  const int slot = threadIdx.x;  // 0 ... 127: this robot's place in the workgroup (= its Rwb row)
  const long base = (long)blockIdx.x * 128;
  const long robot = base + slot;
  const bool mine = robot < n;
  QC_CLK_BEGIN();
  // SOLO: the workgroups of the launch's LAST round keep their stragglers to themselves - each wave is the consumer of its own
  // list.  Behind them no workgroup is waiting for a wave slot, so nothing is gained by ending early, and a wave's own list
  // is the shorter chain at the end of the launch.
  const bool solo = blockIdx.x >= solo_from;
  const int wv_id = threadIdx.x >> 6;
  if (threadIdx.x == 0) { lds.count[0] = 0; lds.count[1] = 0; lds.arrived = 0; }
  __syncthreads();  // the only barrier: both waves have just started
  double* const list = lds.rec + (solo ? wv_id * PAIR_CAP * PAIR_REC : 0);
  int* const list_n = &lds.count[solo ? wv_id : 0];
  // ---------------------------------------------------------------- producer: one lane per robot
  {
    Lane1 L;
    Eqp eqp(nullptr);
    L.idx = -1;
    L.foot0 = 0;
    L.stance = 0;
    L.tol_s = 0.0;
    bool busy = false;
    if (mine) {
      const uint32_t sw = in.stance ? *reinterpret_cast<const uint32_t*>(in.stance + 4 * robot) : 0u;
      const uint32_t wv = warm ? warm[robot] : 0u;
      RawState S;
      TickExtra X;
      double fp[12];
      fetch_extra<KIN>(in, robot, X);
      fetch_state<4, KIN>(in, robot, 0, S, fp);
      asm volatile("" ::: "memory");  // (every load of the fill is issued above this line)
      CParams& P = *QC_PARAMS_HERE(Pg);
      Wrench<4> W;
      const uint32_t st = assemble_from_state<KIN, 4, false>(P, in, robot, 0, S, fp, sw, X, W);
#pragma unroll
      for (int k = 0; k < 9; k++) lds.Rrows[9 * slot + k] = S.R[k];
      L.load_direct(W, st, wv, robot, 0, P.tol_start);
      eqp.setup(P, L.Wr, 0);
      busy = P.max_iter != 0;  // (0: the batch-load probe - load -> assemble -> store only)
    }
    QC_CLK(0, 2);
    const int first_steps = clamp_steps_for<1>(*QC_PARAMS_HERE(Pg), warm);
#pragma unroll 1
    for (int k = 0; k < first_steps; k++)
      if (busy) busy = !L.template iterate<Lane1::FIRST>(*QC_PARAMS_HERE(Pg), eqp, true, k == 0 && warm == nullptr);
    unsigned long long bm = __builtin_amdgcn_ballot_w64(busy);
    while (__builtin_popcountll(bm) > th) {
      if (busy) busy = !L.template iterate<Lane1::STEADY>(*QC_PARAMS_HERE(Pg), eqp);
      bm = __builtin_amdgcn_ballot_w64(busy);
    }
    if (mine && !busy) {
      CParams& P = *QC_PARAMS_HERE(Pg);
      store_result<KIN, 4, true>(P, in, out, robot, L.stance, L.status, L.iters, L.word_bits() | 0x80000000u, L.f, 0, lds.Rrows + 9 * slot, 1);
    }
    const int nb = __builtin_popcountll(bm);
    int first = 0;
    if (nb > 0 && lane == 0) first = __hip_atomic_fetch_add(list_n, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    first = __builtin_amdgcn_readfirstlane(first);
    if (busy) {
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
      double* rec = list + (first + rank) * PAIR_REC;
#pragma unroll
      for (int k = 0; k < 6; k++) rec[k] = L.Wr.b[k];
      const unsigned long long fl = (unsigned long long)slot | ((unsigned long long)((L.stance | L.polish_bit()) & 0x3FFu) << 8) |
                                    ((unsigned long long)(L.word_bits() & 0xFFFFFFu) << 24) | ((unsigned long long)(uint32_t)L.iters << 48);
      rec[6] = __longlong_as_double((long long)fl);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
          rec[7 + 3 * i + k] = L.Wr.r[i][k];
          rec[19 + 3 * i + k] = L.f[3 * i + k];
        }
    }
  }
  // ---------------------------------------------------------------- hand-over inside the workgroup: the last wave to arrive consumes
  int arrived = 1;
  if (!solo) {
    if (lane == 0) arrived = __hip_atomic_fetch_add(&lds.arrived, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
  }
  if (arrived == 0) return;  // (no clock hook here: the development harnesses' end-of-kernel hooks may hold a barrier)
  const int T = __hip_atomic_load(list_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (T == 0) {
    QC_CLK_END(8);
    return;
  }
  QC_CLK(7, 8);
  // ---------------------------------------------------------------- consumer: four lanes per robot, refilled from the list
  const int g4 = lane & 15, j4 = lane >> 4;
  Lane4 L4;
  Eqp4 eqp4(nullptr);
  int rslot = 0;  // the robot's place in the workgroup (its Rwb row)
  auto take = [&](int t) {
    const double* rec = list + t * PAIR_REC;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      L4.Wr.b[k] = rec[k];  // (already -b)
      asm volatile("" : "+v"(L4.Wr.b[k]));
    }
    const unsigned long long fl = (unsigned long long)__double_as_longlong(rec[6]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      L4.Wr.r[0][k] = rec[7 + 3 * j4 + k];
      L4.f[k] = rec[19 + 3 * j4 + k];
    }
    rslot = (int)(fl & 0xFFu);
    L4.idx = base + rslot;
    L4.stance = (uint32_t)((fl >> 8) & 0x1FFu);
    const uint32_t fw = (uint32_t)(fl >> (24 + 6 * j4));
    L4.C.sx[0] = dec2(fw); L4.C.sy[0] = dec2(fw >> 2); L4.C.sz[0] = dec2(fw >> 4);
    L4.iters = (int)(fl >> 48);
    L4.foot0 = j4;
    L4.status = QC_MAX_ITER;
    {
      const double tol = Eqp4::kTolScale * QC_PARAMS_HERE(Pg)->tol_d;
      L4.tol_s = ((fl >> 8) & QC_POLISH_ONE) ? tol : -tol;
    }
    L4.have_f = true;
    eqp4.setup(*QC_PARAMS_HERE(Pg), L4.Wr, j4);
  };
  auto push4 = [&](const Lane4& X, int rs) {  // this lane's foot of a finished robot, straight to the outputs
    const uint32_t word = (uint32_t)group_or<4, true>((int)X.word_bits()) | 0x80000000u;
    CParams& P = *QC_PARAMS_HERE(Pg);
    store_result<KIN, 1, true>(P, in, out, X.idx, X.stance, X.status, X.iters, word, X.f, j4, lds.Rrows + 9 * rs, 1);
  };
  // groups without a robot shadow ticket 0 (the strided layout keeps every lane in the loop: MFMA sums read all 64)
  bool busy4 = g4 < T;
  bool holds = busy4;  // the group holds a robot whose result has not been stored yet
  take(busy4 ? g4 : 0);
  int next = T < 16 ? T : 16;
  const int stop = race ? 8 : 0;
  UConst uc;
  if constexpr (UNIFORM) uc = load_uconst(*QC_PARAMS_HERE(Pg));
  for (;;) {
    bool done;
    if constexpr (UNIFORM) {
      pin_uconst(uc);
      done = L4.template iterate<Lane4::STEADY>(uc, eqp4, busy4);
    } else {
      done = L4.template iterate<Lane4::STEADY>(*QC_PARAMS_HERE(Pg), eqp4, busy4);
    }
    busy4 = busy4 & !done;
    const unsigned run16 = (unsigned)(__builtin_amdgcn_ballot_w64(busy4) & 0xFFFFull);  // member 0 of the groups = lanes 0-15
    const int nrun = __builtin_popcount(run16);
    if (next < T) {
      if (16 - nrun >= refill) {
        if (holds && !busy4) push4(L4, rslot);
        const unsigned free16 = ~run16 & 0xFFFFu;
        const int t = next + __builtin_popcount(free16 & ((1u << g4) - 1u));
        if (!busy4) {
          holds = t < T;
          if (holds) {
            take(t);
            busy4 = true;
          }
        }
        next += __builtin_popcount(free16);
        next = next < T ? next : T;
      }
      continue;
    }
    if (nrun <= stop) break;
  }
  if (holds && !busy4) push4(L4, rslot);
  const unsigned run16 = (unsigned)(__builtin_amdgcn_ballot_w64(busy4) & 0xFFFFull);
  const int nrun = __builtin_popcount(run16);
  if (nrun == 0) {
    QC_CLK_END(8);
    return;
  }
  // race stage (as the one-fill kernels' tail): the <= 8 survivors are re-packed - through the drained list - each into TWO
  // groups 8 lanes apart that continue with different drop rules (most negative multiplier / all negative multipliers);
  // the first at the KKT point wins
  {
    double* rpk = list;
    const int rank = __builtin_popcount(run16 & ((1u << g4) - 1u));
    __builtin_amdgcn_wave_barrier();  // (this wave is alone by now; LDS operations of one wave stay in order)
    if (busy4) repack_write(L4, rpk + rank * REPACK_RS, rslot, j4);
    __builtin_amdgcn_wave_barrier();
    LaneR LR;
    Eqp4 eqpR(nullptr);
    const int r = g4 & 7, sid = g4 >> 3;  // robot, strategy of this group
    bool busyR = r < nrun;
    const int rs = repack_read(Pg, LR, eqpR, rpk + (busyR ? r : 0) * REPACK_RS, j4);
    LR.drop_all = sid == 1;
    unsigned solved_mask = 0;
    auto after = [&](bool fin) {
      const int me = (busyR & fin & (LR.status == QC_SOLVED)) ? (1 << sid) : 0;
      const int m = me | __builtin_amdgcn_update_dpp(0, me, 0x120 + 8, 0xF, 0xF, true);  // row_ror 8: the partner group
      solved_mask |= (unsigned)m;
      busyR = busyR & !fin & (solved_mask == 0);
    };
    while (__builtin_amdgcn_ballot_w64(busyR) != 0) {
      if constexpr (UNIFORM) {
        pin_uconst(uc);
        after(LR.template iterate<LaneR::STEADY>(uc, eqpR, busyR));
      } else {
        after(LR.template iterate<LaneR::STEADY>(*QC_PARAMS_HERE(Pg), eqpR, busyR));
      }
    }
    const int win = solved_mask ? __builtin_ctz(solved_mask) : 0;
    if (r < nrun && sid == win) {
      const uint32_t word = (uint32_t)group_or<4, true>((int)LR.word_bits()) | 0x80000000u;
      CParams& P = *QC_PARAMS_HERE(Pg);
      store_result<KIN, 1, true>(P, in, out, LR.idx, LR.stance, LR.status, LR.iters, word, LR.f, j4, lds.Rrows + 9 * rs, 1);
    }
  }
  QC_CLK_END(8);
}

}  // namespace qc

// =============================================================== host / C ABI
typedef void (*qc_kernel_fn)(const qc::DevParams*, long, qc::BatchIn, const uint32_t*, qc::BatchOut, long, int);

struct qc_handle {
  int device;
  int cus;                  // compute units of the device
  qc::DevParams dp;
  qc::DevParams* d_params;  // device copy of dp (rewritten only by the qc_set_* calls, after a device synchronise)
  bool diag_w;   // W diagonal -> 6x6 formulation
  bool uniform;  // additionally S diagonal and W = w*I -> scalar-constant specialisation
  // what qc_create was given: the tuning overrides (force_general / force_dense / max_iter / probe_batch_load) restore from these
  bool cfg_diag_w, cfg_uniform;
  bool small_w;     // qc_create's rule: max diag(S) / min diag(W) above QC_DENSE_RATIO - the 6x6 dual forms lose digits that matter there
  bool auto_dense;  // ... and such a handle runs the dense 12x12 form (qc_set_tuning "auto_dense", default on)
  int cfg_max_iter;
  bool force_general, force_dense, probing;
  // launch heuristics (defaults from measurements; qc_set_tuning overrides them)
  int refill_t;            // parked lanes that trigger a refill (persistent waves)
  double rounds_cold;      // cold batches up to rounds x (robots resident as one-fill workgroups) run one-fill
  double rounds_warm;      // the same for warm-started batches
  long chunk_override;     // > 0: robots per wave
  int group_override;      // 1, 2, 4: lanes per robot
  int one_fill_override;   // 0: never (persistent waves), 1: always, -1: heuristic
  int wave_slots_override; // > 0: resident workgroups assumed for every kernel instead of the occupancy query
  int race_override;       // -1 heuristic; 0 / 1: no racing strategies; 2, 4: at most that many per robot
  int min_waves;           // development builds (QC_EXPERIMENTAL_OCC): register cap of the one-fill kernels, waves per SIMD
  // resident workgroups per kernel instantiation (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs), filled lazily
  struct { const void* fn; size_t lds; long resident; } occ[40];
  int n_occ;
  // staging buffers for the host-pointer entry points
  void* stage;
  size_t stage_bytes;
  void* pin;  // pinned host buffer the kernel reads/writes in place for small batches
  size_t pin_bytes;
  uint32_t last_word;  // qc_control(): working set of the previous call (hot start)
  bool has_last;
  hipStream_t stream;
  int pair_override;  // MODE 3 (paired waves): -1 heuristic, 0 never, 1 whenever the form allows it
  int pair_th;        // hand-over threshold (0: heuristic)
  int pair_refill;    // free lane groups that trigger a refill (0: heuristic)
  int pair_solo;      // 1: the last round of workgroups keeps each wave's stragglers in the wave (default), 0: pairs everywhere
};

// The 6x6 forms solve the DUAL system M = S^-1 + A~ B^-1 A~^T, whose B^-1 = O(1/w) part has rank = the number of free force
// coordinates: with fewer than six of them free, M is a rank-k term of size 1/w on top of an S^-1 of order one and the forces
// (1/w) A~^T v come out with an absolute error of eps (S/w) |b| - 1e-5 N at S/w = 1e9 (the parameter campaigns' 1.4e-5 ... 2.4e-5
// at w ~ 1e-7, S ~ 100: profiles/r06_fuzz_campaigns.log), 1e-8 N at the reference's S/w = 1e6 (commander_node.cpp:305-307).  The
// dense 12x12 form factorises the PRIMAL reduced Hessian, which is well-conditioned exactly there (tests/test_gpu_parity.py::
// test_small_w_golden: < 5e-6 where the dual forms are at 2.4e-5), at three to five times the time.  Above this ratio a handle
// therefore runs the dense form: accuracy before speed for regularisation weights this far below the reference's.
#ifndef QC_DENSE_RATIO
#define QC_DENSE_RATIO 3.0e8
#endif
// the formulation a handle runs, from what qc_create was given and the tuning flags - in one place, whatever the order of the calls
static void resolve_form(qc_handle* h) {
  h->diag_w = h->cfg_diag_w && !h->force_dense && !(h->small_w && h->auto_dense);
  h->uniform = h->cfg_uniform && !h->force_general;
}

#define QC_COMMA(...) __VA_ARGS__
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define QC_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) return fail(QC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Cholesky-based inverse of a small SPD matrix (host, once per handle)
static bool spd_inverse(const double* A, int n, double* inv) {
  double L[36];
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        L[i * n + i] = std::sqrt(s);
      } else {
        L[i * n + j] = s / L[j * n + j];
      }
    }
  for (int c = 0; c < n; c++) {
    double y[6], x[6];
    for (int i = 0; i < n; i++) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k];
      y[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = y[i];
      for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
      x[i] = s / L[i * n + i];
    }
    for (int i = 0; i < n; i++) inv[i * n + c] = x[i];
  }
  return true;
}

// Columns 0..2 of the inverse of FootTrajectory::initSystem()'s 7x7 matrix (trajectory.cpp:256-277):
// the responses of the sextic coefficients to p_start, p_final and p_centre.  basis[3*j + k].
static void sextic_basis(double* basis) {
  double A[7][7] = {{1, 0, 0, 0, 0, 0, 0}, {1, 1, 1, 1, 1, 1, 1}, {1, 0.5, 0.25, 0.125, 0.0625, 0.03125, 0.015625},
                    {0, 1, 0, 0, 0, 0, 0}, {0, 1, 2, 3, 4, 5, 6}, {0, 0, 2, 0, 0, 0, 0}, {0, 0, 2, 6, 12, 20, 30}};
  double M[7][10];
  for (int i = 0; i < 7; i++) {
    for (int j = 0; j < 7; j++) M[i][j] = A[i][j];
    for (int k = 0; k < 3; k++) M[i][7 + k] = (i == k) ? 1.0 : 0.0;
  }
  for (int c = 0; c < 7; c++) {
    int p = c;
    for (int r = c + 1; r < 7; r++) if (std::fabs(M[r][c]) > std::fabs(M[p][c])) p = r;
    for (int j = 0; j < 10; j++) std::swap(M[c][j], M[p][j]);
    const double piv = M[c][c];
    for (int j = 0; j < 10; j++) M[c][j] /= piv;
    for (int r = 0; r < 7; r++) {
      if (r == c) continue;
      const double m = M[r][c];
      for (int j = 0; j < 10; j++) M[r][j] -= m * M[c][j];
    }
  }
  for (int j = 0; j < 7; j++)
    for (int k = 0; k < 3; k++) basis[3 * j + k] = M[j][7 + k];
}

// Rewrites the device copy of the constants.  Launches in flight - on any stream, non-blocking ones included -
// may still be reading the old copy through scalar loads, so the device is drained first; the copy itself is
// synchronous.  Setters are configuration calls, not per-tick calls.
static int upload_params(qc_handle* h) {
  QC_HIP(hipSetDevice(h->device));
  QC_HIP(hipDeviceSynchronize());
  QC_HIP(hipMemcpy(h->d_params, &h->dp, sizeof(qc::DevParams), hipMemcpyHostToDevice));
  return QC_OK;
}

// ---------------------------------------------------------------- launch planning
// One wave per 64-thread block; a group of G lanes per robot.  The group width trades latency for throughput
// (instructions per steady recalculation: 487 per 16 robots at G = 4, 730 per 32 at G = 2, ~1170 per 64 at G = 1):
// the planner takes the widest group with which the batch still fits ONE wave per SIMD - such a batch cannot fill
// the chip anyway and the slowest robot's serial chain is what is timed - i.e. G = 4 up to CUs x 4 x 16 robots
// (16 384), G = 2 up to twice that, and one lane per robot above (tools/size_scan.py: the cross-overs sit exactly
// there, cold and warm).  Every width finishes its last <= 16 running robots on the 4-lane body.  Batches up to
// `rounds` times the resident one-fill workgroups run as one-fill workgroups - the hardware scheduler does the
// refill; with the one-lane kernel that wins at every size measured (2 M robots cold: 633 us against 676 us as
// persistent waves), so `rounds` is unbounded and the persistent kernels (waves walking contiguous chunks with lane
// refill) serve qc_set_tuning("one_fill", 0) in development builds only.
enum { QC_FORM_UNIFORM = 0, QC_FORM_GENERAL = 1, QC_FORM_DENSE = 2 };
// The persistent-wave (MODE 0) kernels - round 1's large-batch kernels, which the planner has not picked for any batch of a
// 6x6 form since the one-lane one-fill kernel exists (round 2) nor for the one-lane dense form since its LDS diet (round 5:
// one-fill workgroups resident once per SIMD win at every size) - are compiled only into development builds
// (-DQC_PERSISTENT_6X6=1: the A/B scans, QC_TEST_PERSISTENT_6X6=1 in tests/test_gpu_matrix.py): fourteen instantiations with
// up to 672 B of scratch per lane that a default handle can never launch.  Without them every form runs as one-fill
// workgroups and qc_set_tuning("one_fill", 0) / a chunk beyond one fill is an error at launch.
#ifndef QC_PERSISTENT_6X6
#define QC_PERSISTENT_6X6 0
#endif
#ifndef QC_ROUNDS_COLD
#define QC_ROUNDS_COLD 1.0e9
#endif
#ifndef QC_ROUNDS_WARM
#define QC_ROUNDS_WARM 1.0e9
#endif

template <class EQP, int MINW, int MODE, int RACE = 1>
static qc_kernel_fn kernel_of(bool kin) {
  return kin ? (qc_kernel_fn)qc::balance_kernel<EQP, true, MINW, MODE, RACE> : (qc_kernel_fn)qc::balance_kernel<EQP, false, MINW, MODE, RACE>;
}
// the kernel instantiation for (form, lanes per robot, mode); mode 2 exists for the uniform G = 4 form only
static qc_kernel_fn kernel_for(int form, int G, int mode, bool kin, int minw = 2, int race = 1) {
  using namespace qc;
  constexpr bool STR = !QC_NO_STRIDED;
#ifdef QC_EXPERIMENTAL_OCC  // development builds: register-capped one-fill instantiations of the uniform form (tools/occ_scan.py)
  if (form == QC_FORM_UNIFORM && mode == 1 && minw > 2) {
    if (G == 4) return minw >= 4 ? kernel_of<EqpDiagW<true, 4, STR>, 4, 1>(kin) : kernel_of<EqpDiagW<true, 4, STR>, 3, 1>(kin);
    if (G == 2) return minw >= 4 ? kernel_of<EqpDiagW<true, 2>, 4, 1>(kin) : kernel_of<EqpDiagW<true, 2>, 3, 1>(kin);
  }
#endif
  (void)minw;
#ifdef QC_DEV_ONLY_DENSE1  // development: compile the one-lane dense kernels alone (seconds instead of a minute; tools/kernel_resources.py)
  (void)form; (void)G; (void)race;
  return kernel_of<EqpDense, 1, 1>(kin);
#else
#if QC_PERSISTENT_6X6
#define QC_MODE0(EQP) kernel_of<EQP, 2, 0>(kin)
#define QC_MODE0_DENSE kernel_of<EqpDense, 1, 0>(kin)
#else
#define QC_MODE0(EQP) nullptr /* the planner never asks: plan_launch keeps every form on one-fill workgroups */
#define QC_MODE0_DENSE nullptr
#endif
  if (form == QC_FORM_DENSE && G == 4) return race == 4 ? kernel_of<EqpDense4, 1, 1, 4>(kin) : (race == 2 ? kernel_of<EqpDense4, 1, 1, 2>(kin) : kernel_of<EqpDense4, 1, 1>(kin));
  if (form == QC_FORM_DENSE) return mode ? kernel_of<EqpDense, 1, 1>(kin) : QC_MODE0_DENSE;
  if (form == QC_FORM_GENERAL) {
    if (G == 4 && mode && STR && race == 4) return kernel_of<EqpDiagW<false, 4, STR>, 2, 1, (STR ? 4 : 1)>(kin);
    if (G == 4 && mode && STR && race == 2) return kernel_of<EqpDiagW<false, 4, STR>, 2, 1, (STR ? 2 : 1)>(kin);
    if (G == 4) return mode ? kernel_of<EqpDiagW<false, 4, STR>, 2, 1>(kin) : QC_MODE0(QC_COMMA(EqpDiagW<false, 4>));
    if (G == 2) return mode ? kernel_of<EqpDiagW<false, 2>, 2, 1>(kin) : QC_MODE0(QC_COMMA(EqpDiagW<false, 2>));
    return mode ? kernel_of<EqpDiagW<false, 1>, 2, 1>(kin) : QC_MODE0(QC_COMMA(EqpDiagW<false, 1>));
  }
  if (G == 4 && mode == 2 && STR && race == 4) return kernel_of<EqpDiagW<true, 4, STR>, 2, 2, (STR ? 4 : 1)>(kin);
  if (G == 4 && mode == 2 && STR && race == 2) return kernel_of<EqpDiagW<true, 4, STR>, 2, 2, (STR ? 2 : 1)>(kin);
  if (G == 4) return mode == 2 ? kernel_of<EqpDiagW<true, 4, STR>, 2, 2>(kin) : (mode ? kernel_of<EqpDiagW<true, 4, STR>, 2, 1>(kin) : QC_MODE0(QC_COMMA(EqpDiagW<true, 4>)));
  if (G == 2) return mode ? kernel_of<EqpDiagW<true, 2>, 2, 1>(kin) : QC_MODE0(QC_COMMA(EqpDiagW<true, 2>));
  return mode ? kernel_of<EqpDiagW<true, 1>, 2, 1>(kin) : QC_MODE0(QC_COMMA(EqpDiagW<true, 1>));
#undef QC_MODE0
#undef QC_MODE0_DENSE
#endif  // QC_DEV_ONLY_DENSE1
}
typedef void (*qc_pair_kernel_fn)(const qc::DevParams*, long, qc::BatchIn, const uint32_t*, qc::BatchOut, int, int, int, unsigned);
static qc_pair_kernel_fn pair_kernel_for(int form) {
  using namespace qc;
#ifdef QC_DEV_ONLY_DENSE1
  (void)form;
  return nullptr;
#else
  return form == QC_FORM_UNIFORM ? (qc_pair_kernel_fn)balance_pair_kernel<EqpDiagW<true, 1>, false> : (qc_pair_kernel_fn)balance_pair_kernel<EqpDiagW<false, 1>, false>;
#endif
}
static size_t lds_for(int form, int G, int mode, bool kin = false) {
  if (mode == 3) return 0;  // (static LDS: Rwb rows, record list, two counters)
  if (kin && G == 1 && mode == 1 && form != QC_FORM_DENSE) return (size_t)qc::qpark_doubles() * sizeof(double);  // (balance_kernel: QPARK)

  const size_t stock = (size_t)qc::stock_doubles(qc::stock_slots(G, mode)) * sizeof(double);
  if (form == QC_FORM_DENSE && G != 4 && mode != 0) return (size_t)78 * 64 * sizeof(double);  // the Hessian planes alone (balance_kernel: HESS_ONLY)
  if (form == QC_FORM_DENSE) return stock + (G == 4 ? (size_t)qc::EqpDense4::X_DOUBLES : (size_t)78 * 64) * sizeof(double);
  return stock;
}
// workgroups of this kernel the device holds at once (registers, LDS and the 32-waves-per-CU cap, as the runtime sees them)
static long resident_workgroups(qc_handle* h, const void* fn, size_t lds, int threads = 64) {
  if (h->wave_slots_override > 0) return h->wave_slots_override;
  for (int i = 0; i < h->n_occ; i++)
    if (h->occ[i].fn == fn && h->occ[i].lds == lds) return h->occ[i].resident;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess || per_cu <= 0) per_cu = 4;
  const long r = (long)per_cu * h->cus;
  if (h->n_occ < (int)(sizeof(h->occ) / sizeof(h->occ[0]))) {
    h->occ[h->n_occ].fn = fn; h->occ[h->n_occ].lds = lds; h->occ[h->n_occ].resident = r;
    h->n_occ++;
  }
  return r;
}

struct qc_launch_plan {
  qc_pair_kernel_fn pfn;  // mode 3
  int p_th, p_refill;
  qc_kernel_fn fn;
  size_t lds;
  unsigned blocks;
  long chunk;
  int refill_t, G, mode, race;
  long resident;
};

static int plan_launch(qc_handle* h, long n, bool kin, bool warm, qc_launch_plan* lp) {
  const int form = !h->diag_w ? QC_FORM_DENSE : (h->uniform ? QC_FORM_UNIFORM : QC_FORM_GENERAL);
  int G = 1;
  const long simds = (long)h->cus * 4;
  const long cap4 = resident_workgroups(h, (const void*)kernel_for(form, 4, 1, kin, h->min_waves), lds_for(form, 4, 1, kin)) * 16;
  if (form != QC_FORM_DENSE) {
    G = n <= 16 * simds ? 4 : (n <= 32 * simds ? 2 : 1);
    if (h->group_override) G = h->group_override;
  } else {
    // dense form: four lanes per robot (one-fill workgroups only) while the batch fits the resident waves - the
    // latency regime; one lane per robot with the LDS-staged Hessian above that
    G = n <= cap4 ? 4 : 1;
    if (h->group_override) G = h->group_override == 4 ? 4 : 1;
    if (QC_PERSISTENT_6X6 && G == 4 && (h->one_fill_override == 0 || h->chunk_override > 16)) G = 1;
    // a chunk of 17 ... 64 robots is more than a four-lane wave holds and exactly what a one-lane one-fill wave does (ADVICE r5:
    // rounds 2-4 served such a request on the one-lane kernel; it must not become an error because the four-lane one would have been picked)
    if (!h->group_override && G == 4 && h->chunk_override > 16 && h->chunk_override <= 64) G = 1;
  }
  lp->pfn = nullptr;
  lp->p_th = lp->p_refill = 0;
  // A request this build cannot honour is an error, not a silent fall-back to one-fill workgroups: the persistent (mode 0)
  // kernels - of the 6x6 forms and, since round 5, of the one-lane dense form too - exist only in -DQC_PERSISTENT_6X6=1 builds.
  if (!QC_PERSISTENT_6X6 && (h->one_fill_override == 0 || h->chunk_override > 64 / G))
    return fail(QC_ERR_INVALID, "qc_set_tuning: one_fill = 0 / a chunk beyond one fill asks for a persistent-wave kernel, which this "
                                "build does not contain (compile with -DQC_PERSISTENT_6X6=1)");
  // MODE 3 (paired waves): 6x6 forms, one lane per robot, batches of at least four rounds of one-fill workgroups (524 288
  // robots): measured -5 % there and nothing below (profiles/r03_paired_waves.log) - with few rounds the consumer of two
  // waves' stragglers is mostly a longer chain at the end of the launch.  (Not the joint_q variants: they always run as
  // one-fill workgroups.)
  if (form != QC_FORM_DENSE && G == 1 && !kin && h->chunk_override <= 0 && h->one_fill_override != 0) {
    const long res1 = resident_workgroups(h, (const void*)kernel_for(form, 1, 1, kin, h->min_waves), lds_for(form, 1, 1, kin));
    bool use_pair = n >= 4 * res1 * 64;
    if (h->pair_override >= 0) use_pair = h->pair_override != 0;
    if (use_pair) {
      lp->pfn = pair_kernel_for(form);
      lp->fn = nullptr;
      lp->lds = 0;
      lp->blocks = (unsigned)((n + 127) / 128);
      lp->chunk = 128;
      lp->refill_t = 0;
      lp->G = 1;
      lp->mode = 3;
      lp->race = 1;
      // two-wave workgroups of the pair kernel itself (its own registers and 26 KB of static LDS), not half of the one-fill figure
      lp->resident = resident_workgroups(h, (const void*)lp->pfn, 0, 128);
      lp->p_th = h->pair_th > 0 ? (h->pair_th < qc::PAIR_CAP ? h->pair_th : qc::PAIR_CAP) : 24;
      // (1 ... 16 free lane groups: beyond 16 the refill test of the consumer could never fire and it would spin on a full list)
      lp->p_refill = h->pair_refill > 0 ? (h->pair_refill < 16 ? h->pair_refill : 16) : 4;
      return QC_OK;
    }
  }
  const long rpw = 64 / G;  // robots per wave fill
  const bool can_one_fill = true;
  bool one_fill = false;
  long resident = 0;
  if (can_one_fill) {
    resident = resident_workgroups(h, (const void*)kernel_for(form, G, 1, kin, h->min_waves), lds_for(form, G, 1, kin));
    // (round 5: the one-lane dense kernel too - with the Hessian planes as its only LDS it is resident once per SIMD and, as
    // one-fill workgroups, beats its persistent form at every size: 65 536 robots 138 vs 172 us, 262 144 317 vs 415,
    // profiles/r05_dense_sizes.log)
    const double rounds = warm ? h->rounds_warm : h->rounds_cold;
    one_fill = (double)n <= rounds * (double)(resident * rpw);
    // The joint_q / joint_tau variants carry the kinematics through the persistent loop and spill 400-700 B per
    // lane there; as one-fill workgroups they stay at <= 68 B and win at every size.
    if (kin) one_fill = true;
    if (h->one_fill_override >= 0) one_fill = h->one_fill_override != 0;
    if (h->chunk_override > 0) one_fill = h->chunk_override <= rpw;
    if (form == QC_FORM_DENSE && G == 4) one_fill = true;
    if (!QC_PERSISTENT_6X6) one_fill = true;  // (the persistent kernels are not in this build)
  }
  int mode = 0, race = 1;
  long chunk;
  if (one_fill) {
    chunk = (h->chunk_override > 0 && h->chunk_override <= rpw) ? h->chunk_override : rpw;
    const long blocks = (n + chunk - 1) / chunk;
    // one wave per SIMD is enough: the recalculation's constants stay resident in VGPRs (uniform G = 4 form)
    mode = (form == QC_FORM_UNIFORM && G == 4 && blocks <= (long)h->cus * 4) ? 2 : 1;
    // ... and when even that leaves SIMDs idle the 4-lane kernels (all three forms) race pivoting strategies on the
    // same robot (Lane: RACE): four per robot up to CUs x 4 x 4 robots (4 096)
    if (G == 4 && !QC_NO_STRIDED && h->chunk_override <= 0 && (mode == 2 || form != QC_FORM_UNIFORM)) {
      // (a 2-way race up to 8 192 robots is built and selectable - qc_set_tuning("race", 2) - but measured neutral on
      // average: its two fewer recalculations pay for the heavier body, tools/race_scan.py)
      // (cold batches: a warm-started one has little chain to shorten - 10-17 us either way, 0.2-0.6 us dearer with the race,
      // tools/warm_race_scan.py)
      race = (!warm && n <= 4 * simds) ? 4 : 1;
      if (h->race_override >= 0) race = (h->race_override == 4 && n <= 4 * simds) ? 4 : ((h->race_override >= 2 && n <= 8 * simds) ? 2 : 1);
      chunk = 16 / race;
    }
  } else {
    resident = resident_workgroups(h, (const void*)kernel_for(form, G, 0, kin), lds_for(form, G, 0));
    chunk = rpw;
    if (n > rpw * resident) chunk = ((n + resident - 1) / resident + 15) / 16 * 16;
    if (h->chunk_override > 0) chunk = h->chunk_override;
  }
  lp->fn = kernel_for(form, G, mode, kin, h->min_waves, race);
  lp->race = race;
  lp->lds = lds_for(form, G, mode, kin);
  lp->blocks = (unsigned)((n + chunk - 1) / chunk);
  lp->chunk = chunk;
  lp->refill_t = h->refill_t > 0 ? (h->refill_t + G - 1) / G : 1;
  lp->G = G;
  lp->mode = mode;
  lp->resident = resident;
  return QC_OK;
}

extern "C" {

const char* qc_last_error(void) { return g_err.c_str(); }
int qc_abi_version(void) { return QC_ABI_VERSION; }
int qc_check_abi(int abi_version, size_t sizeof_params, size_t sizeof_batch_in, size_t sizeof_batch_out) {
  // A caller compiled against another revision of include/qc_balance.h would hand over structs of another size
  // (qc_batch_in has grown with every ABI revision and carries no size field): refuse it before any of them is read.
  if (abi_version != QC_ABI_VERSION || sizeof_params != sizeof(qc_params) || sizeof_batch_in != sizeof(qc_batch_in) ||
      sizeof_batch_out != sizeof(qc_batch_out)) {
    char msg[256];
    std::snprintf(msg, sizeof(msg), "qc_check_abi: caller built against ABI v%d (qc_params %zu B, qc_batch_in %zu B, qc_batch_out %zu B), library is "
                  "ABI v%d (%zu / %zu / %zu B)", abi_version, sizeof_params, sizeof_batch_in, sizeof_batch_out, QC_ABI_VERSION,
                  sizeof(qc_params), sizeof(qc_batch_in), sizeof(qc_batch_out));
    return fail(QC_ERR_ABI, msg);
  }
  return QC_OK;
}
const char* qc_kernel_name(const qc_handle* h) {
  if (!h) return "";
  return h->diag_w ? (h->uniform ? "diagW-6x6-uniform" : "diagW-6x6") : "dense-12x12";
}

void qc_default_kinematics(qc_kinematics* k) {
  if (!k) return;
  // QuadrupedKinematics::QuadrupedKinematics(), kinematics.cpp:20-47
  const double xbh = 0.196, ybh = 0.050, zbh = 0.0, l1 = 0.077, l2 = 0.211, l3 = 0.230;
  const double hip[12] = {-xbh, ybh, zbh, xbh, ybh, zbh, -xbh, -ybh, zbh, xbh, -ybh, zbh};  // RL FL RR FR
  const double links[12] = {l1, -l2, -l3, l1, -l2, -l3, -l1, -l2, -l3, -l1, -l2, -l3};    // left, left, right, right
  std::memcpy(k->hip, hip, sizeof(hip));
  std::memcpy(k->links, links, sizeof(links));
  k->tau_min = -20.0;  // commander_node.cpp:324-325
  k->tau_max = 20.0;
  const double kff[3] = {0.0, 0.0, 0.0}, kp[3] = {40.0, 40.0, 50.0}, kd[3] = {1.0, 1.0, 1.0};  // mit_cheetah_config.yaml:50-53
  std::memcpy(k->jc_kff, kff, sizeof(kff));
  std::memcpy(k->jc_kp, kp, sizeof(kp));
  std::memcpy(k->jc_kd, kd, sizeof(kd));
  const double xbt = 0.196, ybt = 0.127, zbt = 0.0;  // foot_planner.cpp:27-42
  const double phip[12] = {-xbt, ybt, zbt, xbt, ybt, zbt, -xbt, -ybt, zbt, xbt, -ybt, zbt};
  std::memcpy(k->planner_hip, phip, sizeof(phip));
  k->planner_k = 0.01;     // foot_planner.cpp:25
  k->swing_height = 0.08;  // gait/height, commander_node.cpp:247
}

void qc_swing_state_init(qc_swing_state* s, size_t n) {
  if (!s) return;
  std::memset(s, 0, n * sizeof(qc_swing_state));
  for (size_t i = 0; i < n; i++)
    for (int l = 0; l < 4; l++) s[i].leg_state[l] = -1;
}

int qc_set_kinematics(qc_handle* h, const qc_kinematics* kin) {
  if (!h) return fail(QC_ERR_INVALID, "qc_set_kinematics: null handle");
  qc_kinematics k;
  if (kin) k = *kin; else qc_default_kinematics(&k);
  if (!(k.tau_min <= k.tau_max)) return fail(QC_ERR_INVALID, "qc_set_kinematics: need tau_min <= tau_max");
  std::memcpy(h->dp.hip, k.hip, sizeof(k.hip));
  std::memcpy(h->dp.links, k.links, sizeof(k.links));
  h->dp.tau_min = k.tau_min;
  h->dp.tau_max = k.tau_max;
  std::memcpy(h->dp.jc_kff, k.jc_kff, sizeof(k.jc_kff));
  std::memcpy(h->dp.jc_kp, k.jc_kp, sizeof(k.jc_kp));
  std::memcpy(h->dp.jc_kd, k.jc_kd, sizeof(k.jc_kd));
  std::memcpy(h->dp.planner_hip, k.planner_hip, sizeof(k.planner_hip));
  h->dp.planner_k = k.planner_k;
  h->dp.swing_height = k.swing_height;
  return upload_params(h);
}

int qc_set_gait(qc_handle* h, double t_swing, double t_stance) {
  if (!h) return fail(QC_ERR_INVALID, "qc_set_gait: null handle");
  if (!(t_swing >= 0.0) || !(t_stance >= 0.0) || !(t_swing + t_stance > 0.0)) return fail(QC_ERR_INVALID, "qc_set_gait: need t_swing, t_stance >= 0, not both 0");
  h->dp.stance_phase = t_stance / (t_swing + t_stance);  // gait.cpp:45
  h->dp.t_swing = t_swing;
  h->dp.t_stance = t_stance;
  return upload_params(h);
}

int qc_create_abi(const qc_params* p, int device, qc_handle** out, int abi_version, size_t sizeof_params, size_t sizeof_batch_in,
                  size_t sizeof_batch_out) {
  if (!p || !out) return fail(QC_ERR_INVALID, "qc_create: null argument");
  *out = nullptr;
  // ABI v6: no handle for a caller that fills other structs than this library reads (include/qc_balance.h, qc_create)
  if (const int rc = qc_check_abi(abi_version, sizeof_params, sizeof_batch_in, sizeof_batch_out); rc != QC_OK) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(QC_ERR_NO_DEVICE, "qc_create: no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(QC_ERR_INVALID, "qc_create: device ordinal out of range");
  if (!(p->mu > 0.0) || !(p->mass > 0.0)) return fail(QC_ERR_INVALID, "qc_create: mu and mass must be > 0");
  if (!(p->fzmin >= 0.0) || !(p->fzmax >= p->fzmin)) return fail(QC_ERR_INVALID, "qc_create: need 0 <= fzmin <= fzmax");
  // The cone rows of the reference are two-sided with +-1e6 on the far side (BC.cpp:296-301: -1e6 <= fx - mu fz <= 0, ...).
  // Inside the cone |fx -+ mu fz| <= 2 mu fz <= 2 mu fzmax, so those sides cannot bind - and this solver does not carry
  // them - as long as 2 mu fzmax < 1e6.  Parameters beyond that would make the reference's QP a different one: refused.
  // (the hand-over records of the re-packed tails carry the recalculation count in 16 bits)
  if (p->max_iter > QC_MAX_ITER_LIMIT) return fail(QC_ERR_INVALID, "qc_create: max_iter must be <= 65535");
  if (!(2.0 * p->mu * p->fzmax < 1.0e6))
    return fail(QC_ERR_INVALID, "qc_create: need 2 * mu * fzmax < 1e6 (the +-1e6 sides of the reference's cone rows, balance_controller.cpp:296-301, are not carried)");
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < i; j++)
      if (std::fabs(p->S[6 * i + j] - p->S[6 * j + i]) > 1e-12 * (std::fabs(p->S[6 * i + i]) + std::fabs(p->S[6 * j + j])))
        return fail(QC_ERR_INVALID, "qc_create: S must be symmetric");
  bool diag = true;
  for (int i = 0; i < 12; i++) {
    if (!(p->W[12 * i + i] > 0.0)) return fail(QC_ERR_INVALID, "qc_create: W must be positive definite");
    for (int j = 0; j < 12; j++)
      if (i != j && p->W[12 * i + j] != 0.0) diag = false;
  }
  for (int i = 0; i < 12; i++)
    for (int j = 0; j < i; j++)
      if (std::fabs(p->W[12 * i + j] - p->W[12 * j + i]) > 1e-12 * (p->W[12 * i + i] + p->W[12 * j + j]))
        return fail(QC_ERR_INVALID, "qc_create: W must be symmetric");

  qc_handle* h = new (std::nothrow) qc_handle();
  if (!h) return fail(QC_ERR_INVALID, "qc_create: out of memory");
  h->device = device;
  h->diag_w = diag;
  h->stage = nullptr;
  h->stage_bytes = 0;
  h->pin = nullptr;
  h->pin_bytes = 0;
  h->last_word = 0;
  h->has_last = false;
  h->stream = nullptr;
  h->d_params = nullptr;
  qc::DevParams& d = h->dp;
  std::memset(&d, 0, sizeof(d));
  d.mu = p->mu; d.mass = p->mass; d.fzmin = p->fzmin; d.fzmax = p->fzmax;
  std::memcpy(d.Ib, p->Ib, sizeof(d.Ib));
  std::memcpy(d.S, p->S, sizeof(d.S));
  if (!spd_inverse(p->S, 6, d.V)) { delete h; return fail(QC_ERR_INVALID, "qc_create: S must be positive definite"); }
  for (int i = 0; i < 12; i++) d.w[i] = p->W[12 * i + i];
  std::memcpy(d.W, p->W, sizeof(d.W));
  for (int i = 0; i < 4; i++) {
    d.inv_wx[i] = 1.0 / d.w[3 * i];
    d.inv_wy[i] = 1.0 / d.w[3 * i + 1];
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < 2; b++)
        d.inv_bz[4 * i + 2 * a + b] = 1.0 / (d.w[3 * i + 2] + p->mu * p->mu * (a * d.w[3 * i] + b * d.w[3 * i + 1]));
  }
  bool uni = diag;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++)
      if (i != j && p->S[6 * i + j] != 0.0) uni = false;
  for (int i = 1; i < 12; i++)
    if (p->W[12 * i + i] != p->W[0]) uni = false;
  h->cfg_diag_w = diag;
  h->cfg_uniform = uni;
  h->force_general = h->force_dense = h->probing = false;
  {
    double smax = 0.0, wmin = p->W[0];
    for (int i = 0; i < 6; i++) smax = std::fmax(smax, p->S[6 * i + i]);
    for (int i = 1; i < 12; i++) wmin = std::fmin(wmin, p->W[12 * i + i]);
    h->small_w = diag && smax > QC_DENSE_RATIO * wmin;
    h->auto_dense = true;
  }
  resolve_form(h);
  for (int i = 0; i < 6; i++) d.Vd[i] = d.V[6 * i + i];
  d.w_u = p->W[0];
  d.inv_w_u = 1.0 / p->W[0];
  for (int k = 0; k < 3; k++) d.inv_bz_u[k] = 1.0 / (p->W[0] * (1.0 + p->mu * p->mu * k));
  std::memcpy(d.kff, p->kff, sizeof(d.kff));
  std::memcpy(d.kp_p, p->kp_p, sizeof(d.kp_p));
  std::memcpy(d.kd_p, p->kd_p, sizeof(d.kd_p));
  std::memcpy(d.kp_w, p->kp_w, sizeof(d.kp_w));
  std::memcpy(d.kd_w, p->kd_w, sizeof(d.kd_w));
  {
    qc_kinematics k;
    qc_default_kinematics(&k);
    std::memcpy(d.hip, k.hip, sizeof(k.hip));
    std::memcpy(d.links, k.links, sizeof(k.links));
    d.tau_min = k.tau_min;
    d.tau_max = k.tau_max;
    std::memcpy(d.jc_kff, k.jc_kff, sizeof(k.jc_kff));
    std::memcpy(d.jc_kp, k.jc_kp, sizeof(k.jc_kp));
    std::memcpy(d.jc_kd, k.jc_kd, sizeof(k.jc_kd));
    std::memcpy(d.planner_hip, k.planner_hip, sizeof(k.planner_hip));
    d.planner_k = k.planner_k;
    d.swing_height = k.swing_height;
    d.t_swing = 0.18;  // mit_cheetah_config.yaml:17-18
    d.t_stance = 0.8;
    sextic_basis(d.traj_basis);
  }
  d.stance_phase = 0.8 / (0.18 + 0.8);  // mit_cheetah_config.yaml:17-18
  // Multiplier tolerance, relative to max(1, |grad|_inf).  It has to sit just above the rounding noise of the
  // multipliers, not at a "reasonable" 1e-9: along a weakly active face the objective curves only with 2w, so a
  // multiplier accepted at -tol*|g| leaves the force tol*|g|/(2w) away from the minimiser (w = 2.8e-7, |g| ~ 1e6:
  // 1e-12 gave 0.2 N = 2e-3 relative in the parameter fuzz; 1e-13 ... 1e-15 all agree with the NNLS restatement
  // and change neither the recalculation counts of 1 M robots nor anything else measurable).
  d.tol_d = 1e-14;
  d.tol_start = d.tol_d;  // (polish on)
  d.max_iter = p->max_iter > 0 ? p->max_iter : 200;
  h->cfg_max_iter = d.max_iter;
  d.clamp_steps = 0;  // 0: per kernel (clamp_steps_for)
  d.tail_race = 1;
  d.polish = 1;  // the polish at acceptance (Lane::iterate)
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete h; return fail(QC_ERR_HIP, "qc_create: hipGetDeviceProperties failed"); }
  h->cus = prop.multiProcessorCount;
  h->refill_t = 16;
  h->rounds_cold = QC_ROUNDS_COLD;
  h->rounds_warm = QC_ROUNDS_WARM;
  h->chunk_override = 0;
  h->group_override = 0;
  h->one_fill_override = -1;
  h->wave_slots_override = 0;
  h->min_waves = 2;
  h->race_override = -1;
  h->pair_override = -1;
  h->pair_th = 0;
  h->pair_refill = 0;
  h->pair_solo = 1;
  h->n_occ = 0;
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)&h->d_params, sizeof(qc::DevParams)) != hipSuccess ||
      hipMemcpy(h->d_params, &h->dp, sizeof(qc::DevParams), hipMemcpyHostToDevice) != hipSuccess) {
    delete h;
    return fail(QC_ERR_HIP, "qc_create: could not upload the controller constants");
  }
  *out = h;
  return QC_OK;
}

int qc_set_tuning(qc_handle* h, const char* key, double value) {
  if (!h || !key) return fail(QC_ERR_INVALID, "qc_set_tuning: null argument");
  const std::string k(key);
  bool params = false;
  if (k == "group") {
    const int g = (int)value;
    if (g != 0 && g != 1 && g != 2 && g != 4) return fail(QC_ERR_INVALID, "qc_set_tuning: group is 0 (heuristic), 1, 2 or 4");
    h->group_override = g;
  } else if (k == "one_fill") h->one_fill_override = value < 0 ? -1 : (value != 0 ? 1 : 0);
  else if (k == "chunk") h->chunk_override = value > 0 ? (long)value : 0;
  else if (k == "wave_slots") h->wave_slots_override = value > 0 ? (int)value : 0;
  else if (k == "race") {  // (0 / 1 also switch the race in the 4-lane tail of the wider kernels off)
    h->race_override = value < 0 ? -1 : (int)value;
    h->dp.tail_race = (value < 0 || value >= 2) ? 1 : 0;
    params = true;
  }
  else if (k == "pair") h->pair_override = value < 0 ? -1 : (value != 0 ? 1 : 0);
  else if (k == "pair_th") h->pair_th = value > 0 ? (int)value : 0;
  else if (k == "pair_refill") h->pair_refill = value > 0 ? (int)value : 0;
  else if (k == "pair_solo") h->pair_solo = value != 0 ? 1 : 0;
  else if (k == "min_waves") h->min_waves = value > 2 ? (int)value : 2;
  else if (k == "refill_t") h->refill_t = value > 0 ? (int)value : 16;
  else if (k == "rounds_cold") h->rounds_cold = value;
  else if (k == "rounds_warm") h->rounds_warm = value;
  else if (k == "force_general" || k == "force_dense") {
    // general 6x6 form on uniform weights / dense 12x12 form on a diagonal W (same minimiser).  The form follows from what
    // qc_create was given and the two flags, in one place, whatever the order of the calls.
    (k == "force_general" ? h->force_general : h->force_dense) = value != 0;
    resolve_form(h);
  } else if (k == "auto_dense") {  // 0: a diagonal W keeps its 6x6 form however small it is (QC_DENSE_RATIO above)
    h->auto_dense = value != 0;
    resolve_form(h);
  } else if (k == "tol_d") { h->dp.tol_d = value; h->dp.tol_start = h->dp.polish ? value : -value; params = true; }
  else if (k == "max_iter") {  // <= 0: back to the handle's own cap (qc_params.max_iter)
    if (value > (double)QC_MAX_ITER_LIMIT) return fail(QC_ERR_INVALID, "qc_set_tuning: max_iter must be <= 65535");
    h->probing = false;
    h->dp.max_iter = value > 0 ? (int)value : h->cfg_max_iter; params = true;
  }
  else if (k == "clamp_steps") { h->dp.clamp_steps = value >= 1 ? (int)value : 0; params = true; }
  else if (k == "polish") {  // 0: round 5's acceptance rule
    h->dp.polish = value != 0 ? 1 : 0;
    h->dp.tol_start = h->dp.polish ? h->dp.tol_d : -h->dp.tol_d;
    params = true;
  }
  else if (k == "probe_batch_load") {
    // measurement probe: load -> assemble -> store only (every robot reports QC_MAX_ITER); 0 restores the handle's own cap
    h->probing = value != 0;
    h->dp.max_iter = h->probing ? 0 : h->cfg_max_iter; params = true;
  } else return fail(QC_ERR_INVALID, "qc_set_tuning: unknown key '" + k + "'");
  return params ? upload_params(h) : QC_OK;
}

int qc_query_launch(qc_handle* h, size_t n, int kin, int warm, qc_launch_info* out) {
  if (!h || !out) return fail(QC_ERR_INVALID, "qc_query_launch: null argument");
  QC_HIP(hipSetDevice(h->device));
  qc_launch_plan lp;
  const int rc = plan_launch(h, (long)(n ? n : 1), kin != 0, warm != 0, &lp);
  if (rc != QC_OK) return rc;
  out->lanes_per_robot = lp.G;
  out->mode = lp.mode;
  out->form = !h->diag_w ? QC_FORM_DENSE : (h->uniform ? QC_FORM_UNIFORM : QC_FORM_GENERAL);
  out->strategies = lp.race;
  out->chunk = lp.chunk;
  out->blocks = lp.blocks;
  out->resident_workgroups = lp.mode == 3 ? lp.resident : resident_workgroups(h, (const void*)lp.fn, lp.lds);
  out->lds_bytes = (int64_t)lp.lds;
  return QC_OK;
}

void qc_destroy(qc_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->d_params) (void)hipFree(h->d_params);
  if (h->stage) (void)hipFree(h->stage);
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int qc_control_batch(qc_handle* h, size_t n, const qc_batch_in* in, const uint32_t* warm, const qc_batch_out* out, void* stream) {
  if (!h || !in || !out) return fail(QC_ERR_INVALID, "qc_control_batch: null argument");
  if (n == 0) return QC_OK;
  if (!in->Rwb || !in->Rwb_d || !in->x || !in->xdot || !in->w || !in->x_d || !in->xdot_d || !in->w_d || (!in->feet && !in->joint_q))
    return fail(QC_ERR_INVALID, "qc_control_batch: null input array");
  if (!out->grf_body || !out->status) return fail(QC_ERR_INVALID, "qc_control_batch: grf_body and status are required");
  if (out->joint_tau && !in->joint_q) return fail(QC_ERR_INVALID, "qc_control_batch: joint_tau needs joint_q");
  QC_HIP(hipSetDevice(h->device));
  const bool kin = in->joint_q != nullptr;
  const int n_sw = (in->swing_pos ? 1 : 0) + (in->swing_vel ? 1 : 0) + (in->joint_qdot ? 1 : 0);
  if (!in->swing_state && n_sw != 0 && (n_sw != 3 || !in->joint_q || !out->joint_tau))
    return fail(QC_ERR_INVALID, "qc_control_batch: swing_pos, swing_vel and joint_qdot go together and need joint_q and joint_tau");
  if (in->gait_dt && (!in->gait_phase || in->stance)) return fail(QC_ERR_INVALID, "qc_control_batch: gait_dt advances gait_phase (needed, and stance must be NULL)");
  if (in->swing_state && (!in->joint_q || !in->joint_qdot || !in->gait_phase || !out->joint_tau || in->swing_pos || in->swing_vel))
    return fail(QC_ERR_INVALID, "qc_control_batch: swing_state needs joint_q, joint_qdot, gait_phase and joint_tau, and excludes swing_pos/swing_vel");
  qc::BatchIn bi{in->Rwb, in->Rwb_d, in->x, in->xdot, in->w, in->x_d, in->xdot_d, in->w_d, in->feet, in->stance, in->joint_q,
                 in->gait_phase, in->gait_duty, in->swing_pos, in->swing_vel, in->joint_qdot,
                 reinterpret_cast<qc::SwingState*>(in->swing_state), in->gait_dt};
  qc::BatchOut bo{out->grf_body, out->status, out->active_set, out->iterations, out->joint_tau};
  qc_launch_plan lp;
  const int rc = plan_launch(h, (long)n, kin, warm != nullptr, &lp);
  if (rc != QC_OK) return rc;
  if (lp.mode == 3) {
    // the last round of workgroups (those behind which nothing waits for a slot) runs solo
    const long solo_from = h->pair_solo == 0 ? (long)lp.blocks : ((long)lp.blocks > lp.resident ? (long)lp.blocks - lp.resident : 0);
    lp.pfn<<<dim3(lp.blocks), dim3(128), 0, (hipStream_t)stream>>>(h->d_params, (long)n, bi, warm, bo, lp.p_th, lp.p_refill,
                                                                  h->dp.tail_race && warm == nullptr ? 1 : 0, (unsigned)solo_from);
    QC_HIP(hipGetLastError());
    return QC_OK;
  }
  lp.fn<<<dim3(lp.blocks), dim3(64), lp.lds, (hipStream_t)stream>>>(h->d_params, (long)n, bi, warm, bo, lp.chunk, lp.refill_t);
  QC_HIP(hipGetLastError());
  return QC_OK;
}

// host-pointer variant.  Large batches: one device staging allocation, H2D copies, kernel, D2H copies, sync.
// Small batches (n <= kPinnedMaxN; the reference's own use is one robot per controller tick): the records are packed
// into a pinned, device-visible host buffer that the kernel reads and writes in place over PCIe - one launch and one
// stream synchronise instead of a dozen staged copies.
static constexpr size_t kPinnedMaxN = 8192;  // measured crossover with the staged copies: ~16 384 robots (config 2 records)
int qc_control_batch_host(qc_handle* h, size_t n, const qc_batch_in* in, const uint32_t* warm, const qc_batch_out* out) {
  if (!h || !in || !out) return fail(QC_ERR_INVALID, "qc_control_batch_host: null argument");
  if (n == 0) return QC_OK;
  if (!in->Rwb || !in->Rwb_d || !in->x || !in->xdot || !in->w || !in->x_d || !in->xdot_d || !in->w_d || (!in->feet && !in->joint_q))
    return fail(QC_ERR_INVALID, "qc_control_batch_host: null input array");
  if (out->joint_tau && !in->joint_q) return fail(QC_ERR_INVALID, "qc_control_batch_host: joint_tau needs joint_q");
  if (!out->grf_body || !out->status) return fail(QC_ERR_INVALID, "qc_control_batch_host: grf_body and status are required");
  QC_HIP(hipSetDevice(h->device));
  if (!h->stream) QC_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const bool pinned = n <= kPinnedMaxN;
  // layout (all 16-byte aligned slices): 48 doubles in, 12 doubles out, 4 x 4-byte words, optional arrays
  const size_t per = 48 * 8 + 12 * 8 + 4 * 4 + 8 + 12 * 8 + 12 * 8 + 6 * 8 + 3 * 12 * 8 + sizeof(qc_swing_state);
  const size_t need = n * per + 512;
  char* base;
  if (pinned) {
    if (need > h->pin_bytes) {  // grows on demand: a handle that only ever sees one robot per call keeps ~2 KB pinned
      if (h->pin) QC_HIP(hipHostFree(h->pin));
      h->pin = nullptr; h->pin_bytes = 0;
      QC_HIP(hipHostMalloc(&h->pin, need, hipHostMallocDefault));
      h->pin_bytes = need;
    }
    base = (char*)h->pin;
  } else {
    if (need > h->stage_bytes) {
      if (h->stage) QC_HIP(hipFree(h->stage));
      h->stage = nullptr; h->stage_bytes = 0;
      QC_HIP(hipMalloc(&h->stage, need));
      h->stage_bytes = need;
    }
    base = (char*)h->stage;
  }
  size_t off = 0;
  auto carve = [&](size_t bytes) { char* p = base + off; off += (bytes + 15) & ~(size_t)15; return p; };
  hipError_t cerr = hipSuccess;
  auto put = [&](const void* src, size_t bytes) -> void* {  // input slice: carve + host -> buffer
    if (!src) return nullptr;
    char* d = carve(bytes);
    if (pinned) std::memcpy(d, src, bytes);
    else if (cerr == hipSuccess) cerr = hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, h->stream);
    return d;
  };
  auto get = [&](void* dst, const void* d, size_t bytes) {  // output slice: buffer -> host
    if (!dst || !d) return;
    if (pinned) std::memcpy(dst, d, bytes);
    else if (cerr == hipSuccess) cerr = hipMemcpyAsync(dst, d, bytes, hipMemcpyDeviceToHost, h->stream);
  };
  qc_batch_in din{};
  din.Rwb = (const double*)put(in->Rwb, n * 9 * 8);
  din.Rwb_d = (const double*)put(in->Rwb_d, n * 9 * 8);
  din.x = (const double*)put(in->x, n * 3 * 8);
  din.xdot = (const double*)put(in->xdot, n * 3 * 8);
  din.w = (const double*)put(in->w, n * 3 * 8);
  din.x_d = (const double*)put(in->x_d, n * 3 * 8);
  din.xdot_d = (const double*)put(in->xdot_d, n * 3 * 8);
  din.w_d = (const double*)put(in->w_d, n * 3 * 8);
  din.feet = (const double*)put(in->feet, n * 12 * 8);  // may be absent when joint_q is given
  din.stance = (const uint8_t*)put(in->stance, n * 4);
  din.joint_q = (const double*)put(in->joint_q, n * 12 * 8);
  din.gait_phase = (double*)put(in->gait_phase, n * 4 * 8);
  din.gait_duty = (const double*)put(in->gait_duty, n * 8);
  din.gait_dt = (const double*)put(in->gait_dt, n * 8);
  din.swing_pos = (const double*)put(in->swing_pos, n * 12 * 8);
  din.swing_vel = (const double*)put(in->swing_vel, n * 12 * 8);
  din.joint_qdot = (const double*)put(in->joint_qdot, n * 12 * 8);
  din.swing_state = (qc_swing_state*)put(in->swing_state, n * sizeof(qc_swing_state));  // in/out
  const uint32_t* d_warm = (const uint32_t*)put(warm, n * 4);
  qc_batch_out dout{};
  dout.grf_body = (double*)carve(n * 12 * 8);
  dout.status = (int32_t*)carve(n * 4);
  dout.active_set = out->active_set ? (uint32_t*)carve(n * 4) : nullptr;
  dout.iterations = out->iterations ? (int32_t*)carve(n * 4) : nullptr;
  dout.joint_tau = out->joint_tau ? (double*)carve(n * 12 * 8) : nullptr;
  QC_HIP(cerr);
  if (off > (pinned ? h->pin_bytes : h->stage_bytes)) return fail(QC_ERR_INVALID, "qc_control_batch_host: staging overflow");
  int rc = qc_control_batch(h, n, &din, d_warm, &dout, h->stream);
  if (rc != QC_OK) return rc;
  if (pinned) QC_HIP(hipStreamSynchronize(h->stream));  // kernel-end release makes the in-place results visible
  get(out->grf_body, dout.grf_body, n * 12 * 8);
  get(out->status, dout.status, n * 4);
  get(out->active_set, dout.active_set, n * 4);
  get(out->iterations, dout.iterations, n * 4);
  get(out->joint_tau, dout.joint_tau, n * 12 * 8);
  get(in->swing_state, din.swing_state, n * sizeof(qc_swing_state));
  if (in->gait_dt) get(in->gait_phase, din.gait_phase, n * 4 * 8);  // the advanced clock
  QC_HIP(cerr);
  if (!pinned) QC_HIP(hipStreamSynchronize(h->stream));
  return QC_OK;
}

int qc_control(qc_handle* h, const double* Rwb, const double* Rwb_d, const double* x, const double* xdot,
               const double* w, const double* x_d, const double* xdot_d, const double* w_d, const double* feet,
               const uint8_t* stance, double* grf_body, int32_t* status) {
  if (!h || !status) return fail(QC_ERR_INVALID, "qc_control: null argument");
  qc_batch_in in{Rwb, Rwb_d, x, xdot, w, x_d, xdot_d, w_d, feet, stance, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // The handle remembers the working set of its previous call and starts from it - the per-instance hot-start
  // memory of the reference's SQProblem (BC.hpp:161, BC.cpp:191-202).  Same minimiser either way (strictly convex).
  uint32_t word_in = h->last_word, word_out = 0;
  qc_batch_out out{grf_body, status, &word_out, nullptr, nullptr};
  const int rc = qc_control_batch_host(h, 1, &in, h->has_last ? &word_in : nullptr, &out);
  h->has_last = rc == QC_OK && *status == QC_SOLVED;
  h->last_word = word_out;
  return rc;
}

}  // extern "C"
