// qc_device.hpp - device-side building blocks of the batched balance controller
// (gfx950 / CDNA4, wave64, FP64 VALU).  The arithmetic of the QP is not GEMM-shaped
// (per-robot 6x6 / 12x12 matrices, every lane group owns a different robot), so it stays
// on the vector ALU; the matrix pipe is used only as a cross-lane reduction network
// (v_mfma_f64_4x4x4f64 with an all-ones A operand, see group_sum).
//
// Execution model: a GROUP of G lanes (G = 1, 2 or 4) owns one robot
// (one QP instance); each lane of the group owns 4/G feet.  Per-foot work
// (face coefficients, contributions to the 6x6 system, forces, ratio test,
// multipliers) is split across the group, the small dense solve is replicated,
// and partial sums / minima are combined without LDS: DPP quad permutes for adjacent
// lanes, MFMA sums and row swaps for the stride-16 layout of the 4-lane kernels.
// G = 1 maximises throughput per instruction, G = 4 minimises the serial
// latency of one working-set recalculation (small batches, stragglers).
// All per-robot state lives in VGPRs with compile-time indexing; batch inputs
// are read straight from the per-argument arrays of qc_batch_in, so a wave
// touches one contiguous span per array and consumes every fetched line.
//
// Reference being replaced: BalanceController::control(),
// quadruped_controller/src/quadruped_controller/balance_controller.cpp:98-330
// (cited below as BC.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define QC_DEV __device__ __forceinline__
// Phase-clock hooks: empty in the product; tools/phase_clock.hip defines them to attribute cycles to phases.
#ifndef QC_CLK
#define QC_CLK(from, to)
#define QC_CLK_BEGIN()
#define QC_CLK_END(last)
#define QC_CLK_TAIL_BEGIN()  // harness: the 4-lane tail of a one / two lanes-per-robot wave is clocked into its own slots
#define QC_CLK_TAIL_LOOP()
#define QC_CLK_TAIL_END()
#define QC_CLK_PIN(arr)  // harness: pins the values of `arr` at this point so the scheduler cannot move a phase across its marker
#endif
#ifndef QC_CLK_ABS
#define QC_CLK_ABS(from, to)  // harness: a phase boundary between two absolute clock slots (the torque pass: 13 lists, 14 swing, 15 stance)
#endif

#ifndef QC_NO_STRIDED
#define QC_NO_STRIDED 0  // development: 1 keeps the adjacent-lane (DPP) layout in the 4-lanes-per-robot kernels
#endif

namespace qc {

// Uniform (per-handle) constants, uploaded once by qc_create.
struct DevParams {
  double mu, mass, fzmin, fzmax;
  double Ib[9];
  double S[36];        // wrench weight (general SPD)
  double V[36];        // S^-1 (host-computed), used by the diagW formulation
  double w[12];        // diag(W) (diagW formulation)
  double W[144];       // full W, row-major (dense formulation)
  double inv_wx[4];    // 1 / w_x per foot
  double inv_wy[4];    // 1 / w_y per foot
  double inv_bz[16];   // [foot][|sx|*2+|sy|] 1 / (w_z + mu^2 (|sx| w_x + |sy| w_y))
  // uniform-weight specialisation (S diagonal, W = w*I: every reference configuration)
  double Vd[6];        // diag(S^-1)
  double w_u;          // w
  double inv_w_u;      // 1 / w
  double inv_bz_u[3];  // [|sx|+|sy|] 1 / (w (1 + mu^2 k))
  double kff[6], kp_p[3], kd_p[3], kp_w[3], kd_w[3];
  // kinematic model (joint_q / joint_tau extension): kinematics.cpp:20-47, commander_node.cpp:324-325
  double hip[12];      // [leg][xyz] base -> hip
  double links[12];    // [leg][l1,l2,l3] signed
  double tau_min, tau_max;
  double jc_kff[3], jc_kp[3], jc_kd[3];  // swing-leg joint PD (joint_controller.cpp)
  // swing reference generator (foot_planner.cpp, trajectory.cpp)
  double planner_hip[12], planner_k, swing_height, t_swing, t_stance;
  double traj_basis[21];  // [power j][k]: column k of A^-1 of the sextic system (trajectory.cpp:256-277), k = start, final, centre
  double stance_phase; // gait.cpp:45, default duty of the on-device contact rule
  double tol_d;        // relative multiplier tolerance
  double tol_start;    // what a fresh robot starts with: +tol_d with the polish at acceptance on (DevParams::polish), -tol_d with it off
  int max_iter;
  int clamp_steps;     // clamp steps a fresh robot takes before its first ratio test (one-fill kernels)
  int tail_race;       // the 4-lane tail races two drop rules on its last <= 8 robots
  int polish;          // 1: a robot whose smallest multiplier is inside the noise band at its first acceptance releases that face once
};

// Device code reads the constants through the CONSTANT address space (scalar
// loads).  The kernel re-derives the pointer behind an opaque asm before each
// phase (QC_PARAMS_HERE) so the constants are fetched where they are used
// instead of being hoisted out of the persistent loop, which would overflow the
// 102-SGPR file and spill to VGPR lanes.
typedef const __attribute__((address_space(4))) DevParams CParams;
#define QC_PARAMS_HERE(ptr)                 \
  ({                                       \
    const DevParams* p_ = (ptr);           \
    asm volatile("" : "+s"(p_));          \
    (CParams*)(unsigned long long)p_;      \
  })
// ... and from a reference to the constants (inside the EQP structs, which are handed a CParams&)
#define QC_PARAMS_AGAIN(ref)                             \
  ({                                                     \
    unsigned long long p_ = (unsigned long long)&(ref);  \
    asm volatile("" : "+s"(p_));                         \
    (CParams*)p_;                                        \
  })

// The handful of constants one working-set recalculation of the UNIFORM form reads, as a register-resident
// copy (VGPRs, pinned): used by the one-wave-per-SIMD launch of small batches, where a scalar load + wait at the
// top of every recalculation is exposed latency (no second wave to hide it) and the VGPR budget is 512.
struct UConst {
  double mu, fzmin, fzmax, inv_w_u, w_u, tol_start;
  double inv_bz_u[3], Vd[6];
  int max_iter;
};
QC_DEV UConst load_uconst(CParams& P) {
  UConst u;
  u.mu = P.mu; u.fzmin = P.fzmin; u.fzmax = P.fzmax; u.inv_w_u = P.inv_w_u; u.w_u = P.w_u; u.tol_start = P.tol_start;
#pragma unroll
  for (int k = 0; k < 3; k++) u.inv_bz_u[k] = P.inv_bz_u[k];
#pragma unroll
  for (int k = 0; k < 6; k++) u.Vd[k] = P.Vd[k];
  u.max_iter = P.max_iter;
  return u;
}
// keeps the copy where it is (in VGPRs) instead of letting the compiler re-load it from the constant buffer
QC_DEV void pin_uconst(UConst& u) {
  asm volatile("" : "+v"(u.mu), "+v"(u.fzmin), "+v"(u.fzmax), "+v"(u.inv_w_u), "+v"(u.w_u), "+v"(u.tol_start));
  asm volatile("" : "+v"(u.inv_bz_u[0]), "+v"(u.inv_bz_u[1]), "+v"(u.inv_bz_u[2]));
  asm volatile("" : "+v"(u.Vd[0]), "+v"(u.Vd[1]), "+v"(u.Vd[2]), "+v"(u.Vd[3]), "+v"(u.Vd[4]), "+v"(u.Vd[5]), "+v"(u.max_iter));
}

// mirrors qc_swing_state (include/qc_balance.h)
struct SwingState {
  int32_t leg_state[4];
  int32_t has_traj[4];
  double p_start[12];
  double p_final[12];
};

struct BatchIn {
  const double *Rwb, *Rwb_d, *x, *xdot, *w, *x_d, *xdot_d, *w_d, *feet;
  const uint8_t* stance;
  const double* joint_q;
  double* gait_phase;  // written when gait_dt is given
  const double* gait_duty;
  const double* swing_pos;
  const double* swing_vel;
  const double* joint_qdot;
  struct SwingState* swing_state;
  const double* gait_dt;
};
struct BatchOut {
  double* grf_body;
  int32_t* status;
  uint32_t* active_set;
  int32_t* iterations;
  double* joint_tau;
};

// ------------------------------------------------------------ lane groups
// DPP quad permutes: data of lane^1 / lane^2 inside each aligned quad.
QC_DEV int dpp_xor1_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }  // quad_perm [1,0,3,2]
QC_DEV int dpp_xor2_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); }  // quad_perm [2,3,0,1]
QC_DEV double dpp_xor1(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)dpp_xor1_i((int)(unsigned)b), hi = (unsigned)dpp_xor1_i((int)(unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
QC_DEV double dpp_xor2(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)dpp_xor2_i((int)(unsigned)b), hi = (unsigned)dpp_xor2_i((int)(unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// min/max of values known to be non-NaN (tagged candidates, magnitudes): the bare instruction.  fmin()/fmax()
// make the compiler quiet every operand first (v_max_f64 x, x, x) because it cannot prove there is no sNaN,
// which doubles the length of the arg-min trees on the serial chain.
QC_DEV double min_nn(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
QC_DEV double max_nn(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
QC_DEV double max_abs_nn(double a, double b) {  // max(|a|, |b|)
  double r;
  asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// All-reduce over the G lanes of a group.  x op y is commutative, so every
// lane of the group ends up with the bit-identical result: decisions taken
// from reduced values are uniform inside the group.
// Two lane layouts: adjacent lanes (member = lane & (G-1), DPP quad permutes), and - S = true, G = 4 only - the
// four lanes {i, i+16, i+32, i+48} (member = lane >> 4, group = lane & 15).  In the strided layout the matrix pipe
// does the sums: v_mfma_f64_4x4x4f64 with A = 1 computes, for every lane, the sum of B over the lanes that share
// its (lane & 15) - one instruction per reduced double instead of four DPP moves and two adds (27 doubles per
// recalculation: 412 vs 904 cycles, tools/ubench_mfma_reduce.hip), all four lanes getting the bit-identical dot
// product.  min / max / or cross the 16-lane rows with the gfx950 row swaps (v_permlane16_swap, v_permlane32_swap:
// with both operands equal they return the even and the odd row (half) broadcast, so op(r0, r1) is the reduction).
QC_DEV double swap16_op_min(double v) {
  const long long b = __double_as_longlong(v);
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  return min_nn(__longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])),
                __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1])));
}
QC_DEV double swap32_op_min(double v) {
  const long long b = __double_as_longlong(v);
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  return min_nn(__longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])),
                __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1])));
}
QC_DEV double swap16_op_max(double v) {
  const long long b = __double_as_longlong(v);
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  return max_nn(__longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])),
                __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1])));
}
QC_DEV double swap32_op_max(double v) {
  const long long b = __double_as_longlong(v);
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  return max_nn(__longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])),
                __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1])));
}
template <int G, bool S = false>
QC_DEV double group_sum(double v) {
  static_assert(!S || G == 4, "the strided layout is a 4-lane layout");
  if constexpr (S) {
#ifdef QC_SUM_BY_SWAPS
    {
      const long long b = __double_as_longlong(v);
      const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
      const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
      v = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])) + __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
      const long long c = __double_as_longlong(v);
      const auto lo2 = __builtin_amdgcn_permlane32_swap((unsigned)c, (unsigned)c, false, false);
      const auto hi2 = __builtin_amdgcn_permlane32_swap((unsigned)(c >> 32), (unsigned)(c >> 32), false, false);
      return __longlong_as_double((long long)(((unsigned long long)hi2[0] << 32) | lo2[0])) + __longlong_as_double((long long)(((unsigned long long)hi2[1] << 32) | lo2[1]));
    }
#endif
    return __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, v, 0.0, 0, 0, 0);
  } else {
    if (G >= 2) v += dpp_xor1(v);
    if (G >= 4) v += dpp_xor2(v);
    return v;
  }
}
// group sum plus a group-uniform addend: in the strided layout the addend is the MFMA's C operand
template <int G, bool S = false>
QC_DEV double group_sum_add(double v, double addend) {
  if constexpr (S) {
#ifndef QC_SUM_BY_SWAPS
    return __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, v, addend, 0, 0, 0);
#endif
  }
  return group_sum<G, S>(v) + addend;
}
template <int G, bool S = false>
QC_DEV double group_min(double v) {
  if constexpr (S) {
    return swap32_op_min(swap16_op_min(v));
  } else {
    if (G >= 2) v = min_nn(v, dpp_xor1(v));
    if (G >= 4) v = min_nn(v, dpp_xor2(v));
    return v;
  }
}
template <int G, bool S = false>
QC_DEV double group_max(double v) {
  if constexpr (S) {
    return swap32_op_max(swap16_op_max(v));
  } else {
    if (G >= 2) v = max_nn(v, dpp_xor1(v));
    if (G >= 4) v = max_nn(v, dpp_xor2(v));
    return v;
  }
}
template <int G, bool S = false>
QC_DEV int group_or(int v) {
  if constexpr (S) {
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    const unsigned w = a[0] | a[1];
    const auto c = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return (int)(c[0] | c[1]);
  } else {
    if (G >= 2) v |= dpp_xor1_i(v);
    if (G >= 4) v |= dpp_xor2_i(v);
    return v;
  }
}
// position of a lane in its layout
template <int G, bool S>
QC_DEV int lane_member(int lane) { return S ? lane >> 4 : lane & (G - 1); }
template <int G, bool S>
QC_DEV int lane_group(int lane) { return S ? lane & 15 : lane / G; }

// ---------------------------------------------------------------- small math
QC_DEV double rsqrt_nr(double d) {
  // v_rsq_f64 seed + two Newton steps (full FP64 accuracy for d in normal range)
  double y = __builtin_amdgcn_rsq(d);
  double t = d * y;
  double e = __builtin_fma(-t, y, 1.0);
  y = __builtin_fma(y * 0.5, e, y);
  t = d * y;
  e = __builtin_fma(-t, y, 1.0);
  y = __builtin_fma(y * 0.5, e, y);
  return y;
}

QC_DEV double rcp_nr(double d) {
  // v_rcp_f64 seed (2^-24) + two Newton steps: 2^-48, then full FP64 accuracy (tools/ubench_rsq.hip)
  double y = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  return y;
}

// Rotation3d(mat).angleAxisTotal(): math/rigid3d.cpp:177-179,198-203
// (Drake RotationMatrix::ToAngleAxis -> Eigen matrix->quaternion->angle-axis).
// Eigen's four cases (trace > 0, else pivot on the largest diagonal entry; ties keep the lower index, as its > tests do)
// differ only in WHICH of d = 1 +- m00 +- m11 +- m22 goes under the square root and where the entries land, so the case is
// decided first and there is ONE root: r = rsqrt(d) gives sqrt(d) = d r and 0.5 / sqrt(d) = r / 2 without a division, the
// quaternion entries are selects over six products, and the axis scale angle / +-|q_v| is angle * rsqrt(|q_v|^2).  Written
// as four branches (round 2) the compiler if-converted all of them: 4 square roots + 5 divisions, 396 instructions against
// 238 (tools/isa of a one-function kernel); same values to the last ulp or two.
QC_DEV void angle_axis_total(const double (&m)[9], double (&out)[3]) {
  const double tr = m[0] + m[4] + m[8];
  const bool b0 = tr > 0.0;
  const bool b1 = !b0 && (m[0] >= m[4]) && (m[0] >= m[8]);  // i = 0
  const bool b2 = !b0 && !b1 && (m[4] >= m[8]);             // i = 1 (else i = 2)
  const double d = 1.0 + (b0 ? tr : (b1 ? (m[0] - m[4] - m[8]) : (b2 ? (m[4] - m[8] - m[0]) : (m[8] - m[0] - m[4]))));
  const double r = rsqrt_nr(d);
  const double big = 0.5 * (d * r);  // the entry on the pivot: sqrt(d) / 2
  const double h = 0.5 * r;          // 0.5 / sqrt(d)
  const double a = (m[7] - m[5]) * h, b = (m[2] - m[6]) * h, c = (m[3] - m[1]) * h;        // antisymmetric parts
  const double sxy = (m[3] + m[1]) * h, sxz = (m[6] + m[2]) * h, syz = (m[7] + m[5]) * h;  // symmetric parts
  const double qw = b0 ? big : (b1 ? a : (b2 ? b : c));
  const double qx = b0 ? a : (b1 ? big : (b2 ? sxy : sxz));
  const double qy = b0 ? b : (b1 ? sxy : (b2 ? big : syz));
  const double qz = b0 ? c : (b1 ? sxz : (b2 ? syz : big));
  // quaternion -> angle-axis: n = |q_v|, angle = 2 atan2(n, |w|), axis = q_v / (w < 0 ? -n : n); n = 0: angle 0
  const double n2 = qx * qx + qy * qy + qz * qz;
  const double rn = rsqrt_nr(n2);  // (n2 = 0: not a number, selected away below)
  const double angle = 2.0 * atan2(n2 * rn, fabs(qw));
  const double s = (n2 != 0.0) ? (qw < 0.0 ? -angle : angle) * rn : 0.0;
  out[0] = qx * s;
  out[1] = qy * s;
  out[2] = qz * s;
}

// Per-robot quantities every formulation needs: r_i = Rwb p_i (BC.cpp:244-248)
// for the FPL feet this lane owns, and the wrench target b (BC.cpp:126-139,
// 264-269; replicated in every lane of the group).
template <int FPL>
struct Wrench {
  double r[FPL][3];
  double b[6];
};

QC_DEV void load3(const double* __restrict__ p, long idx, double (&v)[3]) {
  const double* q = p + 3 * idx;
  v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
}
QC_DEV void load9(const double* __restrict__ p, long idx, double (&v)[9]) {
  const double* q = p + 9 * idx;
#pragma unroll
  for (int k = 0; k < 9; k++) v[k] = q[k];
}

// Leg kinematics of the reference (kinematics.cpp), one leg: sines/cosines of
// (t1, t2, t2+t3) from three sincos calls and the angle-addition formulas.
struct LegTrig {
  double s1, c1, s2, c2, s23, c23;
};
// sin and cos of a joint angle.  The device library's sincos() carries its huge-argument reduction (Payne-Hanek, six
// v_trig_preop_f64 per call) along: 24 calls per robot in a fused tick (forward kinematics in front, J^T behind), and even
// kept behind a branch as a fallback it cost the joint_q kernels 5 us per 65 536 robots (code size, registers, scratch).
// Joint angles are a few radians: a Cody-Waite reduction by pi/2 in two fused steps (33 + 53 bits of pi/2; the first product
// is exact for k < 2^20 and rounded once - far below the kernels' error - up to 2^30) and the classic minimax kernels on
// [-pi/4, pi/4] (odd degree 13 for the sine, even degree 14 for the cosine; coefficients as in the public-domain fdlibm
// kernels) give both values to 1 ulp of 1 for |x| < 2^30 in ~40 instructions (tools/sincos_check.hip: max error 2.2e-16 against
// libm over +-1e6 and next to multiples of pi/2).  |x| >= 2^30 rad - where neighbouring doubles are 2^-22 rad apart - and
// non-finite angles give NaN (the reference's libm would still return some value in [-1, 1] for the former: INTEGRATION.md).
QC_DEV void sincos_joint(double x, double* __restrict__ sn, double* __restrict__ cs) {
  const bool ok = fabs(x) < 1073741824.0;                            // (false for NaN too)
  const double fn = __builtin_rint(x * 6.36619772367581382433e-01);  // x * 2/pi
  double r = __builtin_fma(-fn, 1.57079632673412561417e+00, x);     // pi/2, first 33 bits
  r = __builtin_fma(-fn, 6.07710050650619224932e-11, r);            // pi/2 - that
  const int n = (int)(ok ? fn : 0.0);
  const double z = r * r;
  const double ps = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                  2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double pc = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                  -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double s = __builtin_fma(r * z, ps, r);
  const double c = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  const bool swap = n & 1;
  const double ss = swap ? c : s, cc = swap ? s : c;
  const double nan = __builtin_nan("");
  *sn = ok ? ((n & 2) ? -ss : ss) : nan;
  *cs = ok ? (((n + 1) & 2) ? -cc : cc) : nan;
}
QC_DEV LegTrig leg_trig(const double* __restrict__ q) {
  LegTrig t;
  double s3, c3;
  sincos_joint(q[0], &t.s1, &t.c1);
  sincos_joint(q[1], &t.s2, &t.c2);
  sincos_joint(q[2], &s3, &c3);
  t.s23 = t.s2 * c3 + t.c2 * s3;
  t.c23 = t.c2 * c3 - t.s2 * s3;
  return t;
}
// QuadrupedKinematics::forwardKinematics(leg, q), kinematics.cpp:81-103; `leg` may be a runtime value
QC_DEV void leg_fk(CParams& P, int leg, const LegTrig& t, double (&p)[3]) {
  const double l1 = P.links[3 * leg], l2 = P.links[3 * leg + 1], l3 = P.links[3 * leg + 2];
  p[0] = l2 * t.s2 + l3 * t.s23 + P.hip[3 * leg];
  p[1] = l1 * t.c1 - l2 * t.s1 * t.c2 - l3 * t.s1 * t.c23 + P.hip[3 * leg + 1];
  p[2] = l1 * t.s1 + l2 * t.c1 * t.c2 + l3 * t.c1 * t.c23 + P.hip[3 * leg + 2];
}
// tau = J^T f with J = QuadrupedKinematics::legJacobian (kinematics.cpp:162-188), kinematics.cpp:219-231
QC_DEV void leg_jt_force(CParams& P, int leg, const LegTrig& t, const double (&f)[3], double (&tau)[3]) {
  const double l1 = P.links[3 * leg], l2 = P.links[3 * leg + 1], l3 = P.links[3 * leg + 2];
  const double a = l2 * t.c2 + l3 * t.c23;  // jac(0,1)
  const double b = l2 * t.s2 + l3 * t.s23;
  const double j01 = a, j02 = l3 * t.c23;
  const double j10 = -l1 * t.s1 - a * t.c1, j11 = b * t.s1, j12 = l3 * t.s1 * t.s23;
  const double j20 = l1 * t.c1 - a * t.s1, j21 = -b * t.c1, j22 = -l3 * t.s23 * t.c1;
  tau[0] = j10 * f[1] + j20 * f[2];  // jac(0,0) = 0
  tau[1] = j01 * f[0] + j11 * f[1] + j21 * f[2];
  tau[2] = j02 * f[0] + j12 * f[1] + j22 * f[2];
}

// The constants of one leg by a PER-LANE leg number (the torque pass of the joint_tau kernels gives every lane its own
// (robot, leg) task): vector loads from the constant buffer, a handful of L1 / L2 hits per task.
struct LegGeom {
  double L1, L2, L3;  // signed link lengths, kinematics.cpp:20-47
  double hx, hy, hz;  // base -> hip
};
QC_DEV LegGeom leg_geom(CParams& P, int leg) {
  const double* lk = (const double*)(unsigned long long)(&P.links[0]) + 3 * leg;
  const double* hp = (const double*)(unsigned long long)(&P.hip[0]) + 3 * leg;
  LegGeom g;
  g.L1 = lk[0]; g.L2 = lk[1]; g.L3 = lk[2];
  g.hx = hp[0]; g.hy = hp[1]; g.hz = hp[2];
  return g;
}
// tau = J^T f, as leg_jt_force above, for a per-lane leg
QC_DEV void leg_jt_force(const LegGeom& g, const LegTrig& t, const double (&f)[3], double (&tau)[3]) {
  const double a = g.L2 * t.c2 + g.L3 * t.c23;  // jac(0,1)
  const double b = g.L2 * t.s2 + g.L3 * t.s23;
  const double j01 = a, j02 = g.L3 * t.c23;
  const double j10 = -g.L1 * t.s1 - a * t.c1, j11 = b * t.s1, j12 = g.L3 * t.s1 * t.s23;
  const double j20 = g.L1 * t.c1 - a * t.s1, j21 = -b * t.c1, j22 = -g.L3 * t.s23 * t.c1;
  tau[0] = j10 * f[1] + j20 * f[2];  // jac(0,0) = 0
  tau[1] = j01 * f[0] + j11 * f[1] + j21 * f[2];
  tau[2] = j02 * f[0] + j12 * f[1] + j22 * f[2];
}

// math/numerics.cpp:23-50
QC_DEV double normalize_angle_2PI(double angle) {
  const double two_pi = 2.0 * 3.14159265358979323846;
  angle -= floor(angle / two_pi) * two_pi;
  if (angle < 0.0) angle += two_pi;
  return angle;
}
QC_DEV double normalize_angle_PI(double rad) {
  const double pi = 3.14159265358979323846, two_pi = 2.0 * pi;
  const double qf = floor((rad + pi) / two_pi);
  rad = (rad + pi) - qf * two_pi;
  if (rad < 0.0) rad += two_pi;
  return rad - pi;
}
// x = pinv(J) v for a rank-deficient 3x3 J - arma::pinv, the second branch of legJacobianInverse (kinematics.cpp:196).
// No SVD on the device: two steps of Gaussian elimination with COMPLETE pivoting give a full-rank factorisation
// J = C F (C = the pivot columns of the running remainder, 3 x r; F = its pivot rows scaled by the pivots, r x 3), and
// the Moore-Penrose inverse of a full-rank product is F^T (F F^T)^-1 (C^T C)^-1 C^T.  r = 2 for the stretched leg (the
// knee column is parallel to the hip-pitch column), r = 1 when the lateral clamp of kinematics.cpp:137-140 acts as well;
// a second pivot below 1e-9 of the first counts as zero (it is ~1e-17 there, against ~0.1 at rank 2).  Agrees with the
// SVD-based pinv of the oracle / numpy to 3e-14 relative on out-of-reach targets.  Rare, divergent path.
QC_DEV void pinv3_apply(const double (&J)[9], const double (&v)[3], double (&x)[3]) {
  double A[9], c[2][3], f[2][3];
#pragma unroll
  for (int k = 0; k < 9; k++) A[k] = J[k];
  int rank = 0;
  double piv1 = 0.0;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    double best = -1.0;
    int bi = 0, bj = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const double a = fabs(A[3 * i + j]);
        const bool t = a > best;
        best = t ? a : best; bi = t ? i : bi; bj = t ? j : bj;
      }
    if (k == 0) piv1 = best;
    const bool take = k == 0 ? best > 0.0 : (rank == 1 && best > 1.0e-9 * piv1);
    if (take) {
      double col[3], row[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        col[i] = bj == 0 ? A[3 * i] : (bj == 1 ? A[3 * i + 1] : A[3 * i + 2]);
        row[i] = bi == 0 ? A[i] : (bi == 1 ? A[3 + i] : A[6 + i]);
      }
      const double ip = 1.0 / (bi == 0 ? col[0] : (bi == 1 ? col[1] : col[2]));
#pragma unroll
      for (int i = 0; i < 3; i++) { row[i] *= ip; c[k][i] = col[i]; f[k][i] = row[i]; }
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A[3 * i + j] -= col[i] * row[j];
      rank = k + 1;
    }
  }
  x[0] = x[1] = x[2] = 0.0;
  if (rank == 1) {
    const double cv = c[0][0] * v[0] + c[0][1] * v[1] + c[0][2] * v[2];
    const double cc = c[0][0] * c[0][0] + c[0][1] * c[0][1] + c[0][2] * c[0][2];
    const double ff = f[0][0] * f[0][0] + f[0][1] * f[0][1] + f[0][2] * f[0][2];
    const double s = cv / (cc * ff);
#pragma unroll
    for (int i = 0; i < 3; i++) x[i] = f[0][i] * s;
  } else if (rank == 2) {
    auto dot = [](const double (&a)[3], const double (&b)[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    const double g00 = dot(c[0], c[0]), g01 = dot(c[0], c[1]), g11 = dot(c[1], c[1]);  // C^T C
    const double a0 = dot(c[0], v), a1 = dot(c[1], v);
    const double ig = 1.0 / (g00 * g11 - g01 * g01);
    const double b0 = ig * (g11 * a0 - g01 * a1), b1 = ig * (g00 * a1 - g01 * a0);
    const double h00 = dot(f[0], f[0]), h01 = dot(f[0], f[1]), h11 = dot(f[1], f[1]);  // F F^T
    const double ih = 1.0 / (h00 * h11 - h01 * h01);
    const double d0 = ih * (h11 * b0 - h01 * b1), d1 = ih * (h00 * b1 - h01 * b0);
#pragma unroll
    for (int i = 0; i < 3; i++) x[i] = f[0][i] * d0 + f[1][i] * d1;
  }
}

// The same wraps with the quotient taken by a multiplication (the division is ~35 instructions, nine of them per swing leg).
// floor() of the two quotients can differ only when the angle sits within an ulp of a multiple of 2 pi, and there both
// versions land within an ulp of the same end of [0, 2 pi] - the subtraction and the `< 0` fix-up are the reference's.
QC_DEV double wrap_2PI(double angle) {
  const double two_pi = 2.0 * 3.14159265358979323846, inv = 1.0 / two_pi;
  angle = __builtin_fma(-floor(angle * inv), two_pi, angle);
  return angle < 0.0 ? angle + two_pi : angle;
}
QC_DEV double wrap_PI(double rad) {
  const double pi = 3.14159265358979323846, two_pi = 2.0 * pi, inv = 1.0 / two_pi;
  const double s = rad + pi;
  double r = __builtin_fma(-floor(s * inv), two_pi, s);
  r = r < 0.0 ? r + two_pi : r;
  return r - pi;
}

// legJacobianInverse(q_ref) * vb (kinematics.cpp:190-204: arma::inv as the closed-form inverse; arma::pinv if singular) and
// JointController::control (joint_controller.cpp:28-36), from the reference angles `qr` and the sines / cosines of
// (q1, q2, q2 + q3) at q_ref.  FAST: reciprocal by Newton steps and the multiplication-form wraps (the trig-free path of
// leg_swing_torque); otherwise the reference's division and wrap functions (its reference-shaped fallback).
template <bool FAST>
QC_DEV void swing_pd(CParams& P, const LegGeom& g, const LegTrig& t, const double (&qr)[3], const double (&vb)[3], const double* __restrict__ q,
                     const double* __restrict__ qdot, double (&tau)[3]) {
  const double L1 = g.L1, L2 = g.L2, L3 = g.L3;
  const double a = L2 * t.c2 + L3 * t.c23, b = L2 * t.s2 + L3 * t.s23;
  const double J[9] = {0.0, a, L3 * t.c23, -L1 * t.s1 - a * t.c1, b * t.s1, L3 * t.s1 * t.s23, L1 * t.c1 - a * t.s1, -b * t.c1, -L3 * t.s23 * t.c1};
  const double c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
  const double det = J[0] * c00 + J[1] * c01 + J[2] * c02;
  // arma::inv of a 3x3 (kinematics.cpp:194) is Armadillo's closed-form "tiny" inverse whenever epsilon <= |det| <= 1 / epsilon;
  // outside that band it hands the matrix to LAPACK, whose LU succeeds - with a 1/sigma_3 ~ 1e17-sized "inverse" - unless it
  // meets an exactly zero pivot, and only then does arma::pinv answer (:196).  This build (and the checker under oracle/) keeps
  // the closed-form band exactly and answers the whole LAPACK band with pinv: a leg stretched because the reference point is
  // out of reach (IK clamps d to 1, q3 = 0, |det| ~ 1e-18 of rounding noise) gets the pseudo-inverse, deterministically,
  // instead of a full-scale torque whose sign depends on rounding inside LAPACK ("choices of this build").
  // The lower end is raised to the rounding noise of det itself, 64 epsilon (sum |l|)^3, where that is larger (legs longer
  // than ~0.4 m): below it the sign of det - and with it the sign of a saturated torque - is noise on any implementation.
  // Rounds 2-3 switched at |det| <= 1e-9 (sum |l|)^3, five orders of magnitude earlier than the reference stops inverting.
  // Inside the pinv band the RANK is pinv3_apply's: never a third pivot, a second one only above 1e-9 of the first.  Armadillo's
  // pinv tolerance (3 sigma_max epsilon) would keep sigma_3 for |det| between ~1e-17 and `lo` and return a 1e13 ... 1e17-sized,
  // noise-signed gain there; IK cannot normally produce such a J (the knee cosine d takes no value between 1 - 2^-53, |det| ~ 3e-10,
  // and 1, |det| ~ 1e-18; only two coinciding granules - sqrt(1 - d^2) ~ 1e-8 times rt ~ 1e-9 - land inside), and the checker does
  // not rely on that: its band pseudo-inverse takes its rank from the same elimination (the CPU test
  // test_pinv_band_takes_its_rank_from_the_device_rule).  The upper end of the band (|det| > 1 / epsilon) needs
  // (sum |l|)^3 > 4.5e15, links of 1e5 m: the same rule applies there and nothing physical reaches it.
  const double ad = fabs(det), lsum = fabs(L1) + fabs(L2) + fabs(L3);
  const double lo = fmax(2.220446049250313e-16, 1.4210854715202004e-14 * lsum * lsum * lsum);
  double qd[3];
  // (a NaN determinant - a reference inside the inner reach limit, d < -1: sqrt(negative) in legInverseKinematics - is neither
  // below nor above the band: Armadillo's closed form then divides the cofactors by NaN, every entry of the "inverse" is NaN and
  // so are all three torques of the leg; written so that NaN takes this branch)
  if (!(ad < lo) && !(ad > 4503599627370496.0)) {
    const double id = FAST ? rcp_nr(det) : 1.0 / det;
    // inverse = adj / det; row r of the inverse dotted with vb
    qd[0] = id * (c00 * vb[0] + (J[2] * J[7] - J[1] * J[8]) * vb[1] + (J[1] * J[5] - J[2] * J[4]) * vb[2]);
    qd[1] = id * (c01 * vb[0] + (J[0] * J[8] - J[2] * J[6]) * vb[1] + (J[2] * J[3] - J[0] * J[5]) * vb[2]);
    qd[2] = id * (c02 * vb[0] + (J[1] * J[6] - J[0] * J[7]) * vb[1] + (J[0] * J[4] - J[1] * J[3]) * vb[2]);
  } else {
    pinv3_apply(J, vb, qd);
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const double e = FAST ? wrap_PI(wrap_2PI(qr[c]) - wrap_2PI(q[c])) : normalize_angle_PI(normalize_angle_2PI(qr[c]) - normalize_angle_2PI(q[c]));
    tau[c] = P.jc_kp[c] * e + P.jc_kd[c] * (qd[c] - qdot[c]) + P.jc_kff[c];
  }
}

// Swing-leg torque of one leg, commander_node.cpp:482-504 + joint_controller.cpp:21-39.
// pb, vb: desired foot position / velocity in the frame the reference hands to IK.
//
// legInverseKinematics (kinematics.cpp:117-160) writes every reference angle as a sum of atan2's of lengths it has just
// computed, and legJacobianInverse then takes sines and cosines of exactly those angles.  So the Jacobian at q_ref needs NO
// trigonometry: with A = atan2(z, +-y), B = atan2(rt, -l1), C = atan2(x, rt), D = atan2(l3 s3, l2 + l3 c3),
//   q1 = +-(A + B),  q2 = -(C + D),  q3 = atan2(-sqrt(1 - d^2), d):   sin q3 = -sqrt(1 - d^2), cos q3 = d  (unit hypotenuse),
// the sine and cosine of each atan2(v, u) are v / hypot and u / hypot - four rsqrt's - and the sums follow from the angle-addition
// formulas.  The three angles themselves are needed only for the PD error, which the reference wraps into [-pi, pi): one atan2
// each of the (sin, cos) pairs gives them modulo 2 pi, which is all the wraps see.  Round 3 evaluated the chain literally:
// five atan2, the device library's sin / cos (with its Payne-Hanek reduction) and three more sincos - ~1500 instructions per
// swing leg against ~650.  Inputs the shortcuts do not cover - a non-finite target (atan2 has its own rules for infinities), a
// foot exactly on the hip axes - take the reference-shaped evaluation below, so the NaN pattern of the torques stays the
// reference's (tests/test_gpu_properties.py::test_non_finite_*).
QC_DEV void leg_swing_torque(CParams& P, const LegGeom& g, const double (&pb)[3], const double (&vb)[3], const double* __restrict__ q,
                             const double* __restrict__ qdot, double (&tau)[3]) {
  // legInverseKinematics, kinematics.cpp:117-160 (unsigned link lengths; right legs have links[0] < 0)
  const double l1 = fabs(g.L1), l2 = fabs(g.L2), l3 = fabs(g.L3);
  const bool right = g.L1 < 0.0;
  const double x = pb[0] - g.hx, y = pb[1] - g.hy, z = pb[2] - g.hz;
  const double num = x * x + y * y + z * z - l1 * l1 - l2 * l2 - l3 * l3;
  double sc = y * y + z * z - l1 * l1;
  if (sc < 0.0) sc = 0.0;
  const double rho2 = y * y + z * z;
  const double rt2 = sc;  // rt^2
  const double sig2 = x * x + rt2;
  // finite AND representable after squaring: a finite target beyond ~1e154 overflows x * x (num, sig2 or rho2 = inf), rsqrt_nr(inf)
  // is 0 and its Newton step inf * 0 = NaN - the reference clamps d = inf to 1 and commands finite, saturated torques
  // (kinematics.cpp:131-134), which is what the reference-shaped evaluation below returns
  const bool finite = (__builtin_fma(num, 0.0, __builtin_fma(sig2, 0.0, rho2 * 0.0)) == 0.0);
  double d = num / (2.0 * l2 * l3);  // (a true division, as the reference's: within a few ulps of |d| = 1 every ulp of d is a different knee angle)
  if (d > 1.0) d = 1.0;
  // d < -1 (the reference point inside the inner reach limit; the reference clamps d > 1 only, kinematics.cpp:131-134) makes
  // q3 and q2 NaN in the reference: that case goes the reference-shaped way below, where the NaNs propagate as they do there
  if (finite && rho2 > 0.0 && sig2 > 0.0 && d >= -1.0) {
    const double u = __builtin_fma(-d, d, 1.0);  // 1 - d^2 >= 0
    const double s3 = u == 0.0 ? -0.0 : -(u * rsqrt_nr(u)), c3 = d;
    const double rt = rt2 == 0.0 ? 0.0 : rt2 * rsqrt_nr(rt2);
    const double ir = rsqrt_nr(rho2);                 // 1 / |(y, z)|
    const double ib = rsqrt_nr(rt2 + l1 * l1);        // 1 / hypot(rt, l1)
    const double is = rsqrt_nr(sig2);                 // 1 / hypot(x, rt)
    const double kx = __builtin_fma(l3, c3, l2), ky = l3 * s3;
    const double k2 = kx * kx + ky * ky;
    const double ik = rsqrt_nr(k2);                   // 1 / hypot(l3 s3, l2 + l3 c3)  (k2 = 0 only for l2 = l3, d = -1)
    const double cA = (right ? y : -y) * ir, sA = z * ir;
    const double cB = -l1 * ib, sB = rt * ib;
    const double sAB = sA * cB + cA * sB, cAB = cA * cB - sA * sB;
    const double cC = rt * is, sC = x * is;
    const double cD = k2 > 0.0 ? kx * ik : 1.0, sD = k2 > 0.0 ? ky * ik : 0.0;
    const double sCD = sC * cD + cC * sD, cCD = cC * cD - sC * sD;
    LegTrig t;
    t.s1 = right ? sAB : -sAB; t.c1 = cAB;
    t.s2 = -sCD; t.c2 = cCD;
    t.s23 = t.s2 * c3 + t.c2 * s3;
    t.c23 = t.c2 * c3 - t.s2 * s3;
    const double qr[3] = {atan2(t.s1, t.c1), atan2(t.s2, t.c2), atan2(s3, c3)};
    swing_pd<true>(P, g, t, qr, vb, q, qdot, tau);
    return;
  }
  // reference-shaped evaluation (rare, divergent)
  const double rt = sqrt(sc);
  double qr[3], s3, c3;
  qr[0] = right ? atan2(z, y) + atan2(rt, -l1) : -(atan2(z, -y) + atan2(rt, -l1));
  qr[2] = atan2(-sqrt(1.0 - d * d), d);
  sincos_joint(qr[2], &s3, &c3);  // (|q3| <= pi; NaN for a non-finite q3, as sin / cos)
  qr[1] = -atan2(x, rt) - atan2(l3 * s3, l2 + l3 * c3);
  swing_pd<false>(P, g, leg_trig(qr), qr, vb, q, qdot, tau);
}

// ---- swing reference generator ----------------------------------------------
// FootPlanner::singleFoot, foot_planner.cpp:76-104 (world-frame foothold of one leg)
// `pc` = Rwb foot_body (foot_planner.cpp:86): the lever arm r_i the wrench assembly has already computed
QC_DEV void plan_foothold(CParams& P, int leg, const double (&R)[9], const double (&x)[3], const double (&xdot)[3], const double (&w)[3],
                          const double (&xdot_d)[3], const double (&pc)[3], double (&fh)[3]) {
  double pt[3];
#pragma unroll
  for (int r = 0; r < 3; r++)
    pt[r] = R[3 * r] * P.planner_hip[3 * leg] + R[3 * r + 1] * P.planner_hip[3 * leg + 1] + R[3 * r + 2] * P.planner_hip[3 * leg + 2] + x[r];
  const double tv[3] = {w[1] * pc[2] - w[2] * pc[1], w[2] * pc[0] - w[0] * pc[2], w[0] * pc[1] - w[1] * pc[0]};
  const double half = 0.5 * P.t_stance, lip = 0.5 * sqrt(x[2] / 9.81);
#pragma unroll
  for (int r = 0; r < 3; r++) fh[r] = pt[r] + (half * xdot[r] + P.planner_k * (xdot[r] - xdot_d[r])) + half * tv[r] + lip * xdot[r];
  fh[2] = 0.0;
}
// FootTrajectoryManager::referenceState + FootTrajectory::trackTrajectory, trajectory.cpp:234-254, 360-388.
// coefficients = A^-1 B with B = [p_start; p_final; p_centre; 0...] (trajectory.cpp:220-225, 279-296), so
// s(t) = h0(t) p_start + h1(t) p_final + h2(t) p_centre with h_k(t) = sum_j basis[j][k] t^j.
QC_DEV void track_swing(CParams& P, double phase, const double (&p0)[3], const double (&pf)[3], double (&pos)[3], double (&vel)[3]) {
  const double duty = P.t_stance / (P.t_swing + P.t_stance);  // stance_phase_, trajectory.cpp:303
  const double slope = 1.0 / (1.0 - duty), yint = 1.0 - slope; // :304-305
  const double u = slope * phase + yint;
  const double t = u < 0.0 ? 0.0 : (1.0 < u ? 1.0 : u);  // std::clamp, :369 (two compares: a NaN phase stays NaN)
  double h[3] = {0.0, 0.0, 0.0}, dh[3] = {0.0, 0.0, 0.0};
  double tp = 1.0, tpm = 0.0;  // t^j and j t^(j-1)
#pragma unroll
  for (int j = 0; j < 7; j++) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      h[k] = __builtin_fma(P.traj_basis[3 * j + k], tp, h[k]);
      dh[k] = __builtin_fma(P.traj_basis[3 * j + k], tpm, dh[k]);
    }
    tpm = (double)(j + 1) * tp;
    tp *= t;
  }
  const double pc[3] = {0.5 * (p0[0] + pf[0]), 0.5 * (p0[1] + pf[1]), P.swing_height};  // trajectory.cpp:325-326
#pragma unroll
  for (int r = 0; r < 3; r++) {
    pos[r] = h[0] * p0[r] + h[1] * pf[r] + h[2] * pc[r];
    vel[r] = dh[0] * p0[r] + dh[1] * pf[r] + dh[2] * pc[r];
  }
}

// K0 + K2 + K3 of SURVEY.md 2.2: gather, PD wrench law, SRB dynamics rhs - in two steps, so that a caller can issue
// EVERYTHING it reads from memory back to back before the first dependent instruction (fetch_state), and compute
// afterwards (wrench_from_state).  One dependent memory round trip costs 0.7-1 us here and a wave's fill used to chain
// six to eight of them (profiles/r03_timeline.log: 2.5 us from kernel entry to "inputs landed" for a lone wave).
struct RawState {
  double R[9], Rd[9], x[3], xd[3], xdot[3], xdotd[3], w[3], wd[3];
};
// `fp`: the FPL feet from foot0 on (body-frame positions, or joint angles when KIN)
template <int FPL, bool KIN>
QC_DEV void fetch_state(const BatchIn& in, long idx, int foot0, RawState& S, double (&fp)[3 * FPL]) {
  load9(in.Rwb, idx, S.R);
  load9(in.Rwb_d, idx, S.Rd);
  load3(in.x, idx, S.x);
  load3(in.x_d, idx, S.xd);
  load3(in.xdot, idx, S.xdot);
  load3(in.xdot_d, idx, S.xdotd);
  load3(in.w, idx, S.w);
  load3(in.w_d, idx, S.wd);
  const double* q = (KIN ? in.joint_q : in.feet) + 12 * idx + 3 * foot0;
#pragma unroll
  for (int k = 0; k < 3 * FPL; k++) fp[k] = q[k];
}
// What the widened tick reads besides the state: gait phases / duty / clock step (contact rule, gait.cpp) and the words of the
// swing-planning record (foot_planner.cpp state_map_, trajectory.cpp traj_map_) - fetched WITH the state, not behind it.
struct TickExtra {
  double ph[4], duty, dt;
  int leg_state[4], has[4];
};
template <bool KIN>
QC_DEV void fetch_extra(const BatchIn& in, long robot, TickExtra& X) {
  if (!in.stance && in.gait_phase) {
#pragma unroll
    for (int i = 0; i < 4; i++) X.ph[i] = in.gait_phase[4 * robot + i];
    if (in.gait_duty) X.duty = in.gait_duty[robot];
    if (in.gait_dt) X.dt = in.gait_dt[robot];
  }
  if (KIN && in.swing_state) {
    const SwingState* S = in.swing_state + robot;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      X.leg_state[i] = S->leg_state[i];
      X.has[i] = S->has_traj[i];
    }
  }
}
// `foot0` = first foot owned by this lane; KIN: foot positions from joint_q by
// forward kinematics instead of the `feet` array.  Returns 0.0 iff every input was finite.
template <int FPL, bool KIN>
QC_DEV double wrench_from_state(CParams& P, const RawState& S, const double (&fp)[3 * FPL], int foot0, Wrench<FPL>& W) {
  const double (&R)[9] = S.R;
  const double (&Rd)[9] = S.Rd;
  const double (&x)[3] = S.x;
  const double (&xd)[3] = S.xd;
  const double (&xdot)[3] = S.xdot;
  const double (&xdotd)[3] = S.xdotd;
  const double (&w)[3] = S.w;
  const double (&wd)[3] = S.wd;
#pragma unroll
  for (int i = 0; i < FPL; i++) {
    double p0 = fp[3 * i], p1 = fp[3 * i + 1], p2 = fp[3 * i + 2];
    if (KIN) {  // commander_node.cpp:383-384: foot_actual_map = kinematics.forwardKinematics(joint_states_map)
      double pb[3];
      const double qa[3] = {p0, p1, p2};
      leg_fk(P, foot0 + i, leg_trig(qa), pb);
      p0 = pb[0]; p1 = pb[1]; p2 = pb[2];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) W.r[i][k] = R[3 * k] * p0 + R[3 * k + 1] * p1 + R[3 * k + 2] * p2;  // BC.cpp:244-248
  }
  // linear PD, BC.cpp:126-129
  double a[3];
#pragma unroll
  for (int k = 0; k < 3; k++) a[k] = P.kp_p[k] * (xd[k] - x[k]) + P.kd_p[k] * (xdotd[k] - xdot[k]);
  a[0] += P.kff[0] * xdotd[0];
  a[1] += P.kff[1] * xdotd[1];
  a[2] += P.kff[2] * P.mass * 9.81;
  // rotation error R_err = Rwb_d Rwb^T and angular PD, BC.cpp:133-139
  double Re[9], e[3], al[3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) Re[3 * i + j] = Rd[3 * i] * R[3 * j] + Rd[3 * i + 1] * R[3 * j + 1] + Rd[3 * i + 2] * R[3 * j + 2];
  angle_axis_total(Re, e);
#pragma unroll
  for (int k = 0; k < 3; k++) al[k] = P.kp_w[k] * e[k] + P.kd_w[k] * (wd[k] - w[k]);
  al[0] += P.kff[3] * wd[0];
  al[1] += P.kff[4] * wd[1];
  al[1] += P.kff[5] * wd[2];  // sic (index 1), BC.cpp:139
  // b, BC.cpp:264-269; g_ = (0,0,-9.81) BC.cpp:76
  W.b[0] = P.mass * a[0];
  W.b[1] = P.mass * a[1];
  W.b[2] = P.mass * (a[2] - 9.81);
  // Iw = Rwb Ib Rwb^T (BC.cpp:251); Iw v = R (Ib (R^T v))
  double t1[3], t2[3], u1[3], u2[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    t1[k] = R[k] * al[0] + R[3 + k] * al[1] + R[6 + k] * al[2];  // R^T al
    t2[k] = R[k] * wd[0] + R[3 + k] * wd[1] + R[6 + k] * wd[2];  // R^T wd
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    u1[k] = P.Ib[3 * k] * t1[0] + P.Ib[3 * k + 1] * t1[1] + P.Ib[3 * k + 2] * t1[2];
    u2[k] = P.Ib[3 * k] * t2[0] + P.Ib[3 * k + 1] * t2[1] + P.Ib[3 * k + 2] * t2[2];
  }
  double Ia[3], Iw[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    Ia[k] = R[3 * k] * u1[0] + R[3 * k + 1] * u1[1] + R[3 * k + 2] * u1[2];
    Iw[k] = R[3 * k] * u2[0] + R[3 * k + 1] * u2[1] + R[3 * k + 2] * u2[2];
  }
  W.b[3] = Ia[0] + (wd[1] * Iw[2] - wd[2] * Iw[1]);
  W.b[4] = Ia[1] + (wd[2] * Iw[0] - wd[0] * Iw[2]);
  W.b[5] = Ia[2] + (wd[0] * Iw[1] - wd[1] * Iw[0]);
  // finiteness probe: any NaN/Inf among the inputs that reach b, r or R makes it non-zero/NaN
  double fin = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) fin = __builtin_fma(W.b[k], 0.0, fin);
#pragma unroll
  for (int i = 0; i < FPL; i++)
#pragma unroll
    for (int k = 0; k < 3; k++) fin = __builtin_fma(W.r[i][k], 0.0, fin);
#pragma unroll
  for (int k = 0; k < 9; k++) fin = __builtin_fma(R[k], 0.0, fin);
  return fin;
}
template <int FPL, bool KIN>
QC_DEV double build_wrench(CParams& P, const BatchIn& in, long idx, int foot0, Wrench<FPL>& W) {
  RawState S;
  double fp[3 * FPL];
  fetch_state<FPL, KIN>(in, idx, foot0, S, fp);
  return wrench_from_state<FPL, KIN>(P, S, fp, foot0, W);
}

// ------------------------------------------------------------ active-set state
// Each stance foot's feasible set { |fx|<=mu fz, |fy|<=mu fz, fzmin<=fz<=fzmax }
// (BC.cpp:274-330: 5 two-sided rows per foot, of which 6 sides can bind) is
// combinatorially a cube: axis X in {fx=-mu fz, free, fx=+mu fz}, same for Y,
// axis Z in {fz=fzmin, free, fz=fzmax}.  State = (sx,sy,sz) in {-1,0,+1}^3.
// Swing feet (BC.cpp:312-316: all five rows pinned to 0) are eliminated: f_i=0.
template <int FPL>
struct Cube {
  int sx[FPL], sy[FPL], sz[FPL];
};

// 6 bits per foot in the warm-start / active_set word; bit 31 = valid.
QC_DEV uint32_t encode_foot(int sx, int sy, int sz) { return (uint32_t)(sx & 3) | ((uint32_t)(sy & 3) << 2) | ((uint32_t)(sz & 3) << 4); }  // -1 -> 3
QC_DEV int dec2(uint32_t v) { return (v & 3u) == 3u ? -1 : ((v & 3u) == 1u ? 1 : 0); }

// Componentwise clamp of one foot into its frustum (branch-free); returns true if the point moved.  lo/hi are the
// foot's own fz bounds (0,0 for a swing foot).  (wx,wy,wz) is the foot's state in the working set the point was
// computed on (all 0 on a cold start): a face of that set is KEPT - the point stays on it (fx = wx mu fz follows a
// clamped fz) - so a warm start whose working set is off by a face or two continues from that set instead of
// re-discovering it one blocking face at a time (config 4: the 3 % of robots whose set changed between ticks
// took 9-11 recalculations, more than a cold start, when the clamp reported only the violated faces).
QC_DEV bool clamp_foot(double mu, double lo, double hi, int wx, int wy, int wz, double& fx, double& fy, double& fz, int& sx, int& sy,
                       int& sz) {
  const bool zu = fz > hi, zl = fz < lo;
  fz = zu ? hi : (zl ? lo : fz);
  sz = zu ? 1 : (zl ? -1 : wz);
  const double m = mu * fz;
  const bool xu = fx > m, xl = fx < -m;
  const double cx = xu ? m : (xl ? -m : fx);
  const double nx = wx != 0 ? (double)wx * m : cx;
  sx = wx != 0 ? wx : (int)xu - (int)xl;
  const bool yu = fy > m, yl = fy < -m;
  const double cy = yu ? m : (yl ? -m : fy);
  const double ny = wy != 0 ? (double)wy * m : cy;
  sy = wy != 0 ? wy : (int)yu - (int)yl;
  const bool moved = zu | zl | (nx != fx) | (ny != fy);
  fx = nx;
  fy = ny;
  return moved;
}

// A (value, 5-bit code) pair packed into one double: the code replaces the 5
// lowest mantissa bits (relative perturbation 2^-47), so a plain v_min_f64
// tree yields arg-min and min together, without compare/select chains.
QC_DEV double tag(double v, int code) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return __longlong_as_double((long long)((b & ~31ull) | (unsigned long long)code));
}
QC_DEV int tag_code(double v) { return (int)(__double_as_longlong(v) & 31ll); }

// Step-length candidate of one face: slack / nd if the face is outside the
// working set and can block (nd > eps and slack < nd, both tested exactly),
// +BIG otherwise.  The ratio itself only ranks candidates, so the
// 2^-23-accurate v_rcp_f64 is enough; a negative slack (face violated by
// rounding) ranks first and is clipped to a zero-length step by the caller.
#define QC_BIG 1.0e300
QC_DEV double step_cand(bool free_face, double slack, double nd, int code) {
  const bool can = free_face & (nd > 1e-14) & (slack < nd);
  const double a = slack * __builtin_amdgcn_rcp(nd);
  const unsigned long long b = (unsigned long long)__double_as_longlong(a);
  const unsigned hi = can ? (unsigned)(b >> 32) : 0x7E37E43Cu;  // high word of 1e300
  const unsigned lo = ((unsigned)b & ~31u) | (unsigned)code;
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ------------------------------------------------------------ small SPD solve
// M = L D L^T in place (packed lower triangle, index r (r + 1) / 2 + c: unit lower L below the diagonal, 1 / d_k on
// it), then rhs <- M^-1 rhs.  No square roots: a pivot's reciprocal is v_rcp_f64 + two Newton steps, and the
// triangular solves carry no scaling on their serial chain.  Returns false if a pivot is not positive (M not PD).
template <int N>
QC_DEV bool ldlt_solve(double (&M)[N * (N + 1) / 2], double (&x)[N]) {
#define QC_LI(r, c) ((r) * ((r) + 1) / 2 + (c))
  bool ok = true;
#pragma unroll
  for (int k = 0; k < N; k++) {
    double d = M[QC_LI(k, k)];
#pragma unroll
    for (int m = 0; m < k; m++) {
      const double u = M[QC_LI(k, m)];        // still unscaled: u_km = L_km d_m
      M[QC_LI(k, m)] = u * M[QC_LI(m, m)];    // L_km
      d = __builtin_fma(-M[QC_LI(k, m)], u, d);
    }
    ok = ok && (d > 0.0);
    M[QC_LI(k, k)] = rcp_nr(d);
#pragma unroll
    for (int r = k + 1; r < N; r++) {  // u_rk = M_rk - sum_m u_rm L_km (scaled when row r becomes the pivot row)
      double t = M[QC_LI(r, k)];
#pragma unroll
      for (int m = 0; m < k; m++) t = __builtin_fma(-M[QC_LI(r, m)], M[QC_LI(k, m)], t);
      M[QC_LI(r, k)] = t;
    }
  }
#pragma unroll
  for (int k = 1; k < N; k++) {
    double t = x[k];
#pragma unroll
    for (int m = 0; m < k; m++) t = __builtin_fma(-M[QC_LI(k, m)], x[m], t);
    x[k] = t;
  }
#pragma unroll
  for (int k = 0; k < N; k++) x[k] *= M[QC_LI(k, k)];
#pragma unroll
  for (int k = N - 2; k >= 0; k--) {
    double t = x[k];
#pragma unroll
    for (int m = k + 1; m < N; m++) t = __builtin_fma(-M[QC_LI(m, k)], x[m], t);
    x[k] = t;
  }
#undef QC_LI
  return ok;
}

// -------------------------------------------------------------- EQP, diagonal W
// Equality-constrained subproblem on the current face, 6-dimensional form.
// With f = T y + p (T,p from the cube states), diagonal W and u = A f - b:
//   (S^-1 + A~ B^-1 A~^T) v = -(b - A p),   v = S u,   y = -B^-1 A~^T v
// where A~ = A T (6 x 12) and B = T^T W T is DIAGONAL for diagonal W.  One
// 6x6 Cholesky per working-set recalculation; the Hessian Q = 2(A^T S A + W)
// of BC.cpp:152 is never formed.  The gradient needed for the multipliers is
// g = Q f + c = 2 (A^T v + W f).

// Face coefficients of one foot for the current cube state.
struct FootCoef {
  double mx, my, fzfix, ix, iy, iz;
};
// UNIFORM = (S diagonal, W = w*I): a handful of scalar constants instead of
// ~60, so they all stay in SGPRs.  The general diagonal-W form takes the weights of
// the foot from a FootW: built from the constant tables with the compile-time foot
// number when a lane owns all four feet (G = 1: scalar operands), or loaded once per
// robot into the lane's registers when a lane owns the feet foot0 ... (lane groups).
struct FootW {
  double inv_wx, inv_wy, inv_bz[4], w[3];
};
QC_DEV FootW foot_weights(CParams& P, int foot) {
  FootW t;
  t.inv_wx = P.inv_wx[foot];
  t.inv_wy = P.inv_wy[foot];
#pragma unroll
  for (int k = 0; k < 4; k++) t.inv_bz[k] = P.inv_bz[4 * foot + k];
#pragma unroll
  for (int k = 0; k < 3; k++) t.w[k] = P.w[3 * foot + k];
  return t;
}
template <bool UNIFORM, class PT>
QC_DEV FootCoef foot_coef(const PT& P, const FootW& fw, int sx, int sy, int sz, bool st) {
  FootCoef k;
  k.mx = P.mu * (double)sx;
  k.my = P.mu * (double)sy;
  k.fzfix = st ? (sz > 0 ? P.fzmax : (sz < 0 ? P.fzmin : 0.0)) : 0.0;
  if constexpr (UNIFORM) {
    k.ix = (st && sx == 0) ? P.inv_w_u : 0.0;
    k.iy = (st && sy == 0) ? P.inv_w_u : 0.0;
    const double b1 = sy != 0 ? P.inv_bz_u[2] : P.inv_bz_u[1];
    const double b0 = sy != 0 ? P.inv_bz_u[1] : P.inv_bz_u[0];
    k.iz = (st && sz == 0) ? (sx != 0 ? b1 : b0) : 0.0;
  } else {
    k.ix = (st && sx == 0) ? fw.inv_wx : 0.0;
    k.iy = (st && sy == 0) ? fw.inv_wy : 0.0;
    const double b0 = sy != 0 ? fw.inv_bz[1] : fw.inv_bz[0];
    const double b1 = sy != 0 ? fw.inv_bz[3] : fw.inv_bz[2];
    k.iz = (st && sz == 0) ? (sx != 0 ? b1 : b0) : 0.0;
  }
  return k;
}

// `stance` = 4-bit mask of the robot, `foot0` = first foot of this lane.
// `lane_w`: the lane's foot weights (general form with lane groups; ignored otherwise).
template <bool UNIFORM, int G, bool S, class PT>
QC_DEV bool eqp_diagw(const PT& P, const FootW (&lane_w)[4 / G], const Wrench<4 / G>& Wr, const Cube<4 / G>& C, uint32_t stance, int foot0,
                      double (&f)[12 / G], double (&g)[12 / G], double& vmax) {
  constexpr int FPL = 4 / G;
  // weights of foot i of this lane: compile-time table entries when the lane owns all feet, its registers otherwise
  auto weights = [&](int i) -> FootW {
    if constexpr (UNIFORM) return FootW{};
    else if constexpr (G == 1) return foot_weights(P, i);
    else return lane_w[i];
  };
  double M[21];
  // packed lower triangle index r*(r+1)/2 + c
#define MI(r, c) ((r) * ((r) + 1) / 2 + (c))
  double rhs[6];

  // Face coefficients: with one foot per lane (G = 4) they stay live across the
  // factorisation (12 VGPRs); with more feet per lane they are
  // recomputed in pass 2 instead, registers matter more there.
  constexpr bool KEEP = FPL <= 1;
  FootCoef kc[FPL];
  // pass 1: this lane's part of sum_i A~_i B_i^-1 A~_i^T and of A p.  The first foot of the lane writes the
  // entries, later feet accumulate (no 0 + x adds, which strict FP cannot fold).
#pragma unroll
  for (int i = 0; i < FPL; i++) {
    const bool first = i == 0;  // compile-time after unrolling
    const bool st = (stance >> (foot0 + i)) & 1u;
    const FootW fwt = weights(i);
    const FootCoef k = foot_coef<UNIFORM>(P, fwt, C.sx[i], C.sy[i], C.sz[i], st);
    if (KEEP) kc[i] = k;
    const double rx = Wr.r[i][0], ry = Wr.r[i][1], rz = Wr.r[i][2];
    double q[6], tq[6];
    q[0] = k.mx; q[1] = k.my; q[2] = 1.0;
    q[3] = ry - rz * k.my;
    q[4] = rz * k.mx - rx;
    q[5] = rx * k.my - ry * k.mx;
#pragma unroll
    for (int c = 0; c < 6; c++) {
      rhs[c] = first ? k.fzfix * q[c] : __builtin_fma(k.fzfix, q[c], rhs[c]);
      tq[c] = k.iz * q[c];  // z slot column q, scaled
    }
    // x slot column (1,0,0, 0, rz, -ry) and y slot column (0,1,0, -rz, 0, rx)
    const double t4 = k.ix * rz, t5 = -k.ix * ry;
    const double t3 = -k.iy * rz, u5 = k.iy * rx;
#define QC_ACC0(r, c) M[MI(r, c)] = first ? q[r] * tq[c] : __builtin_fma(q[r], tq[c], M[MI(r, c)])
#define QC_ACC1(r, c, base) M[MI(r, c)] = first ? __builtin_fma(q[r], tq[c], (base)) : __builtin_fma(q[r], tq[c], M[MI(r, c)] + (base))
    QC_ACC1(0, 0, k.ix);
    QC_ACC0(1, 0); QC_ACC1(1, 1, k.iy);
    QC_ACC0(2, 0); QC_ACC0(2, 1); QC_ACC0(2, 2);
    QC_ACC0(3, 0); QC_ACC1(3, 1, t3); QC_ACC0(3, 2); QC_ACC1(3, 3, -t3 * rz);
    QC_ACC1(4, 0, t4); QC_ACC0(4, 1); QC_ACC0(4, 2); QC_ACC0(4, 3); QC_ACC1(4, 4, t4 * rz);
    QC_ACC1(5, 0, t5); QC_ACC1(5, 1, u5); QC_ACC0(5, 2); QC_ACC1(5, 3, -u5 * rz); QC_ACC1(5, 4, t5 * rz);
    QC_ACC1(5, 5, __builtin_fma(u5, rx, -t5 * ry));
#undef QC_ACC0
#undef QC_ACC1
  }
  QC_CLK_PIN(M); QC_CLK_PIN(rhs);
  QC_CLK(2, 3);
  // combine the group's partial sums; S^-1 and -b ride along as the addend of the reduction (the C operand of
  // the MFMA in the strided layout: no dependent add behind the matrix pipe)
#pragma unroll
  for (int r = 0; r < 6; r++) {
#pragma unroll
    for (int c = 0; c <= r; c++) {
      if constexpr (!UNIFORM) M[MI(r, c)] = group_sum_add<G, S>(M[MI(r, c)], P.V[6 * r + c]);
      else if (r == c) M[MI(r, c)] = group_sum_add<G, S>(M[MI(r, c)], P.Vd[r]);  // S^-1 is diagonal here
      else M[MI(r, c)] = group_sum<G, S>(M[MI(r, c)]);
    }
    rhs[r] = group_sum_add<G, S>(rhs[r], Wr.b[r]);  // (the 6x6 forms keep -b in Wr.b: kNegB, Lane::load_from_stock)
  }
  QC_CLK_PIN(M); QC_CLK_PIN(rhs);
  QC_CLK(3, 4);
  // M = L D L^T (in place: unit lower L below the diagonal, 1/d_k on it).  No square roots: the pivot's
  // reciprocal is v_rcp_f64 + two Newton steps (5 dependent operations instead of the 7 of rsq + Newton), and
  // the triangular solves carry no scaling on their serial chain.
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double wk[6];  // w_m = L_km d_m
    double d = M[MI(k, k)];
#pragma unroll
    for (int m = 0; m < k; m++) {
      wk[m] = M[MI(k, m)];               // still unscaled: u_km = L_km d_m
      M[MI(k, m)] = wk[m] * M[MI(m, m)]; // L_km
      d = __builtin_fma(-M[MI(k, m)], wk[m], d);
    }
    ok = ok && (d > 0.0);
    M[MI(k, k)] = rcp_nr(d);
    // rows below: u_rk = M_rk - sum_m u_rm L_km (kept unscaled until row r becomes the pivot row)
#pragma unroll
    for (int r = k + 1; r < 6; r++) {
      double t = M[MI(r, k)];
#pragma unroll
      for (int m = 0; m < k; m++) t = __builtin_fma(-M[MI(r, m)], M[MI(k, m)], t);
      M[MI(r, k)] = t;
    }
  }
  QC_CLK_PIN(M);
  QC_CLK(4, 5);
  // solve L D L^T v = rhs
  double v[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double t = rhs[k];
#pragma unroll
    for (int m = 0; m < k; m++) t = __builtin_fma(-M[MI(k, m)], v[m], t);
    v[k] = t;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) v[k] *= M[MI(k, k)];
#pragma unroll
  for (int k = 4; k >= 0; k--) {
    double t = v[k];
#pragma unroll
    for (int m = k + 1; m < 6; m++) t = __builtin_fma(-M[MI(m, k)], v[m], t);
    v[k] = t;
  }
#undef MI
  // Scale of the multipliers' rounding noise, for the caller's acceptance test: g = 2 (A^T v + W f) and every entry of A^T v is
  // v_a + v_b r - v_c r', so the noise in g is eps |v| (1 + 2 |r|) - a multiple of |v|_inf, which every lane of the group holds
  // bit-identically (the 6x6 solve is replicated).  Until round 5 the scale was max(1, |g|_inf) over the group's feet: up to
  // twelve maxima plus a cross-lane reduction at the END of the recalculation's chain; this is six maxima off it.
  // max(0.25, |v|_inf): the caller's tolerance carries the factor 4 (kTolScale), i.e. the scale is max(1, 4 |v|_inf) >= |g|_inf
  // for lever arms up to half a metre and within a small factor of it beyond.
  vmax = max_abs_nn(0.25, v[0]);
#pragma unroll
  for (int k = 1; k < 6; k++) vmax = max_abs_nn(vmax, v[k]);
  QC_CLK_PIN(v);
  QC_CLK(5, 6);
  // pass 2: forces and gradient of this lane's feet
#pragma unroll
  for (int i = 0; i < FPL; i++) {
    const bool st = (stance >> (foot0 + i)) & 1u;
    const FootW fwt = weights(i);
    const FootCoef k = KEEP ? kc[i] : foot_coef<UNIFORM>(P, fwt, C.sx[i], C.sy[i], C.sz[i], st);
    const double rx = Wr.r[i][0], ry = Wr.r[i][1], rz = Wr.r[i][2];
    const double ax = v[0] + v[4] * rz - v[5] * ry;  // (A_i^T v)_x
    const double ay = v[1] + v[5] * rx - v[3] * rz;
    const double az = v[2] + v[3] * ry - v[4] * rx;
    const double qv = __builtin_fma(k.mx, ax, __builtin_fma(k.my, ay, az));  // q_i . v
    const double fz = __builtin_fma(-k.iz, qv, k.fzfix);
    const double fx = __builtin_fma(k.mx, fz, -k.ix * ax);
    const double fy = __builtin_fma(k.my, fz, -k.iy * ay);
    f[3 * i] = fx; f[3 * i + 1] = fy; f[3 * i + 2] = fz;
    double wx, wy, wz;
    if constexpr (UNIFORM) wx = wy = wz = P.w_u;
    else { wx = fwt.w[0]; wy = fwt.w[1]; wz = fwt.w[2]; }
    g[3 * i] = 2.0 * __builtin_fma(wx, fx, ax);
    g[3 * i + 1] = 2.0 * __builtin_fma(wy, fy, ay);
    g[3 * i + 2] = 2.0 * __builtin_fma(wz, fz, az);
  }
  QC_CLK_PIN(f); QC_CLK_PIN(g);
  QC_CLK(6, 7);
  QC_CLK(7, 9);  // empty phase: the cost of one marker
  QC_CLK(9, 7);
  return ok;
}

template <bool UNIFORM, int GROUP, bool STRIDED = false>
struct EqpDiagW {
  static constexpr int G = GROUP;
  static constexpr bool kHessianInLds = false;
  static constexpr bool kStrided = STRIDED;  // lane layout of the group, see group_sum
  static constexpr bool kUniform = UNIFORM;
  static constexpr bool kRepackTail = GROUP <= 2;  // one-fill waves finish their stragglers 4 lanes per robot
  static constexpr bool kNegB = true;  // the lane keeps -b (what the right-hand side adds), not b: no negation per recalculation
  static constexpr bool kHasScale = true;  // solve() leaves the group-uniform scale of the multipliers' noise in `gscale`
  static constexpr double kTolScale = 4.0; // ... as max(0.25, |v|_inf): the lane's tolerance carries the 4 (eqp_diagw)
  double gscale;
  FootW lane_w[4 / GROUP];  // general form with lane groups: the weights of this lane's feet (dead otherwise)
  QC_DEV explicit EqpDiagW(double*) {}
  // called when the lane takes a robot; `foot0` = first foot of the lane
  QC_DEV void setup(CParams& P, const Wrench<4 / GROUP>&, int foot0) {
    if constexpr (!UNIFORM && GROUP > 1) {
#pragma unroll
      for (int i = 0; i < 4 / GROUP; i++) lane_w[i] = foot_weights(P, foot0 + i);
    }
  }
  template <class PT>
  QC_DEV bool solve(const PT& P, const Wrench<4 / GROUP>& Wr, const Cube<4 / GROUP>& C, uint32_t stance, int foot0, double (&f)[12 / GROUP],
                    double (&g)[12 / GROUP]) {
    return eqp_diagw<UNIFORM, GROUP, STRIDED>(P, lane_w, Wr, C, stance, foot0, f, g, gscale);
  }
};

// ------------------------------------------------------------ EQP, general W
// Dense formulation for a general SPD W (the API allows any 12x12 SPD W,
// BC.hpp:77; the reference's own configs use W = w*I and take the diagW path).
// The Hessian Q = 2(A^T S A + W) (BC.cpp:152) is assembled once per robot and
// staged in LDS as 78 packed lower-triangle planes of 64 lanes
// (Qs[k*64 + lane]: consecutive lanes -> consecutive 8-byte words, so every
// ds_read_b64 / ds_write_b64 is bank-conflict free).  Each working-set
// recalculation forms the masked reduced Hessian H = T^T Q T + (I - D) in
// registers (fixed 12x12 shape, identity rows on fixed slots, so indexing stays
// compile-time), factorises it with an unrolled Cholesky and back-substitutes.
// One lane per robot (G = 1).
#define QC_SYM(r, c) ((r) >= (c) ? ((r) * ((r) + 1) / 2 + (c)) : ((c) * ((c) + 1) / 2 + (r)))

struct EqpDense {
  static constexpr int G = 1;
  static constexpr bool kHessianInLds = true;  // 78 planes x 64 lanes behind (one-fill workgroups: instead of) the stock
  static constexpr bool kStrided = false;
  static constexpr bool kRepackTail = false;
  static constexpr bool kNegB = false;
  static constexpr bool kHasScale = false;  // the acceptance test scales with max(1, |g|_inf) of the lane group
  static constexpr double kTolScale = 1.0;
  double gscale;  // (unused)
  double* Qs;    // LDS base of this lane: element k at Qs[k * 64]
  double c[12];  // c = -2 A^T S b (BC.cpp:153)

  QC_DEV double q(int r, int cc) const { return Qs[QC_SYM(r, cc) * 64]; }

  QC_DEV explicit EqpDense(double* lds_lane) : Qs(lds_lane) {}

  // assemble Q (into LDS) and c for the robot this lane just fetched
  QC_DEV void setup(CParams& P0, const Wrench<4>& Wr, int) {
    double Sb[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < 6; m++) t = __builtin_fma(P0.S[6 * k + m], Wr.b[m], t);
      Sb[k] = t;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      // the constants of this block column (S: 36 doubles, three columns of W) are fetched HERE: with one pointer for the whole
      // assembly the compiler issues all 180 scalar loads up front, overflows the SGPR file and parks 16-dword tuples in VGPR lanes
      // (120-270 SGPR spills and their stack slots in the kernels that also carry the kinematic constants)
      CParams& P = *QC_PARAMS_AGAIN(P0);
      const double rx = Wr.r[j][0], ry = Wr.r[j][1], rz = Wr.r[j][2];
      // SA_j = S [I; [r_j]x]   (6x3)
      double SA[6][3];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        SA[k][0] = P.S[6 * k + 0] + P.S[6 * k + 4] * rz - P.S[6 * k + 5] * ry;
        SA[k][1] = P.S[6 * k + 1] - P.S[6 * k + 3] * rz + P.S[6 * k + 5] * rx;
        SA[k][2] = P.S[6 * k + 2] + P.S[6 * k + 3] * ry - P.S[6 * k + 4] * rx;
      }
      // c_j = -2 A_j^T (S b)
      c[3 * j + 0] = -2.0 * (Sb[0] + Sb[4] * rz - Sb[5] * ry);
      c[3 * j + 1] = -2.0 * (Sb[1] - Sb[3] * rz + Sb[5] * rx);
      c[3 * j + 2] = -2.0 * (Sb[2] + Sb[3] * ry - Sb[4] * rx);
#pragma unroll
      for (int i = j; i < 4; i++) {
        const double ix = Wr.r[i][0], iy = Wr.r[i][1], iz = Wr.r[i][2];
#pragma unroll
        for (int b = 0; b < 3; b++) {
          // column b of A_i^T SA_j
          const double t0 = SA[0][b] + SA[4][b] * iz - SA[5][b] * iy;
          const double t1 = SA[1][b] - SA[3][b] * iz + SA[5][b] * ix;
          const double t2 = SA[2][b] + SA[3][b] * iy - SA[4][b] * ix;
          const double t[3] = {t0, t1, t2};
#pragma unroll
          for (int a = 0; a < 3; a++) {
            const int r = 3 * i + a, cc = 3 * j + b;
            if (r >= cc) Qs[(r * (r + 1) / 2 + cc) * 64] = 2.0 * (t[a] + P.W[12 * r + cc]);
          }
        }
      }
    }
  }

  QC_DEV bool solve(CParams& P, const Wrench<4>&, const Cube<4>& C, uint32_t stance, int, double (&f)[12], double (&g)[12]) {
    double ax[4], ay[4], az[4], cx[4], cy[4], mx[4], my[4], fzfix[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bool st = (stance >> i) & 1u;
      const int sx = C.sx[i], sy = C.sy[i], sz = C.sz[i];
      ax[i] = (st && sx == 0) ? 1.0 : 0.0;
      ay[i] = (st && sy == 0) ? 1.0 : 0.0;
      az[i] = (st && sz == 0) ? 1.0 : 0.0;
      mx[i] = P.mu * (double)sx;
      my[i] = P.mu * (double)sy;
      cx[i] = mx[i] * az[i];
      cy[i] = my[i] * az[i];
      fzfix[i] = st ? (sz > 0 ? P.fzmax : (sz < 0 ? P.fzmin : 0.0)) : 0.0;
    }
    double L[78], gp[12];
#pragma unroll
    for (int k = 0; k < 12; k++) gp[k] = c[k];
    // H = T^T Q T (+ identity on fixed slots) and gp = Q p + c, one pass over Q
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j <= i; j++) {
        double Qb[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int b = 0; b < 3; b++) Qb[a][b] = q(3 * i + a, 3 * j + b);
        // gp_i += Q_ij p_j ; gp_j += Q_ij^T p_i (i != j).  tz[a] = row a of Q_ij times (mu sx, mu sy, 1)_j: the z column of T_j
        // before its mask, shared with X below
        double tz[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          tz[a] = __builtin_fma(Qb[a][0], mx[j], __builtin_fma(Qb[a][1], my[j], Qb[a][2]));
          gp[3 * i + a] = __builtin_fma(fzfix[j], tz[a], gp[3 * i + a]);
        }
        if (i != j) {
#pragma unroll
          for (int b = 0; b < 3; b++) {
            const double t = Qb[0][b] * mx[i] + Qb[1][b] * my[i] + Qb[2][b];
            gp[3 * j + b] = __builtin_fma(fzfix[i], t, gp[3 * j + b]);
          }
        }
        // X = Q_ij T_j ; H_ij = T_i^T X
        double X[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          X[a][0] = Qb[a][0] * ax[j];
          X[a][1] = Qb[a][1] * ay[j];
          X[a][2] = tz[a] * az[j];
        }
#pragma unroll
        for (int b = 0; b < 3; b++) {
          const double h0 = ax[i] * X[0][b];
          const double h1 = ay[i] * X[1][b];
          const double h2 = az[i] * __builtin_fma(mx[i], X[0][b], __builtin_fma(my[i], X[1][b], X[2][b]));
          const double h[3] = {h0, h1, h2};
#pragma unroll
          for (int a = 0; a < 3; a++) {
            const int r = 3 * i + a, cc = 3 * j + b;
            if (r >= cc) L[r * (r + 1) / 2 + cc] = h[a];
          }
        }
      }
      L[QC_SYM(3 * i + 0, 3 * i + 0)] += 1.0 - ax[i];
      L[QC_SYM(3 * i + 1, 3 * i + 1)] += 1.0 - ay[i];
      L[QC_SYM(3 * i + 2, 3 * i + 2)] += 1.0 - az[i];
    }
    // rhs = -T^T gp
    double y[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      y[3 * i + 0] = -ax[i] * gp[3 * i];
      y[3 * i + 1] = -ay[i] * gp[3 * i + 1];
      y[3 * i + 2] = -(cx[i] * gp[3 * i] + cy[i] * gp[3 * i + 1] + az[i] * gp[3 * i + 2]);
    }
    const bool ok = ldlt_solve<12>(L, y);  // H = L D L^T, y <- H^-1 rhs
    // f = T y + p, with the face coefficients derived from the cube states a second time (a dozen selects) instead of 20 doubles
    // carried across the factorisation, whose 78-entry factor already fills two thirds of the register file (the laundered
    // copies keep the compiler from merging the two derivations)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int sx = C.sx[i], sy = C.sy[i], sz = C.sz[i];
      asm volatile("" : "+v"(sx), "+v"(sy), "+v"(sz));
      const bool st = (stance >> i) & 1u;
      const double ax2 = (st && sx == 0) ? 1.0 : 0.0, ay2 = (st && sy == 0) ? 1.0 : 0.0, az2 = (st && sz == 0) ? 1.0 : 0.0;
      const double fzf = st ? (sz > 0 ? P.fzmax : (sz < 0 ? P.fzmin : 0.0)) : 0.0;
      const double fz = __builtin_fma(az2, y[3 * i + 2], fzf);
      f[3 * i + 2] = fz;
      f[3 * i + 0] = __builtin_fma(ax2, y[3 * i], (P.mu * (double)sx) * fz);
      f[3 * i + 1] = __builtin_fma(ay2, y[3 * i + 1], (P.mu * (double)sy) * fz);
    }
    // g = Q f + c: Q is read from its LDS planes a second time.  Left alone the compiler keeps the 78 values of the first pass
    // alive across the factorisation - in AGPRs, next to L's 78 in VGPRs: 600 v_accvgpr moves and 400-800 B of scratch per lane in
    // a 2 900-instruction recalculation (rounds 1-4) - to save 39 ds_read2 instructions.
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < 12; k++) g[k] = c[k];
#pragma unroll
    for (int r = 0; r < 12; r++)
#pragma unroll
      for (int cc = 0; cc <= r; cc++) {
        const double v = Qs[(r * (r + 1) / 2 + cc) * 64];
        g[r] = __builtin_fma(v, f[cc], g[r]);
        if (r != cc) g[cc] = __builtin_fma(v, f[r], g[cc]);
      }
    return ok;
  }
};


// The dense form on four lanes per robot (stride-16 layout, lane m = foot m): the latency variant for batches
// that cannot fill the chip, where the slowest robot's serial chain of recalculations is what is timed.
// Split of one recalculation:
//   distributed   block row m of the reduced Hessian H = T^T Q T (3 x 12, from the lane's three rows of Q, which
//                 stay in registers for the robot's lifetime), its part of the right-hand side, the foot's force,
//                 gradient, ratio test and multipliers (Lane::iterate, shared with the 6x6 kernels)
//   exchanged     the 144 + 12 doubles of H and the right-hand side, through a per-robot LDS tile
//                 (entry e of robot g at X[e * 17 + g]: the 16 lanes of one foot write consecutive words, the
//                 four lanes of a robot read the same word - no bank conflicts either way)
//   replicated    the 12 x 12 LDL^T and the two triangular solves (a distributed factorisation would pay one
//                 cross-lane transfer per factor entry, which costs as much as the flops it saves)
// The exchange is wave-synchronous (the four lanes of a robot sit in one wavefront): no barrier, only the
// wave-level fence that keeps the compiler from moving LDS reads across the writes of other lanes.
struct EqpDense4 {
  static constexpr int G = 4;
  static constexpr bool kHessianInLds = false;
  static constexpr bool kStrided = true;
  static constexpr bool kUniform = false;
  static constexpr bool kRepackTail = false;
  static constexpr bool kNegB = false;
  static constexpr bool kHasScale = false;  // the acceptance test scales with max(1, |g|_inf) of the lane group
  static constexpr double kTolScale = 1.0;
  double gscale;  // (unused)
  static constexpr int XS = 17;           // tile stride in doubles (16 robots + 1)
  static constexpr int X_DOUBLES = 156 * XS;
  double* X;        // this robot's column of the wave's exchange tile
  double Qr[3][12]; // rows 3 me .. 3 me + 2 of Q = 2 (A^T S A + W), BC.cpp:152
  double c[3];      // entries of c = -2 A^T S b, BC.cpp:153
  int me;

  QC_DEV explicit EqpDense4(double* lds_lane) {
    const int lane = (int)threadIdx.x;
    X = (lds_lane - lane) + (lane & 15);
    me = lane >> 4;
  }
  static QC_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  // assemble this lane's three rows of Q and of c for the robot the group just fetched
  QC_DEV void setup(CParams& P, const Wrench<1>& Wr, int) {
    double r[4][3];
#pragma unroll
    for (int k = 0; k < 3; k++) X[(3 * me + k) * XS] = Wr.r[0][k];
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int k = 0; k < 3; k++) r[j][k] = X[(3 * j + k) * XS];
    wave_sync();  // the tile is rewritten by the first recalculation
    const double ix = Wr.r[0][0], iy = Wr.r[0][1], iz = Wr.r[0][2];
    double Sb[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < 6; m++) t = __builtin_fma(P.S[6 * k + m], Wr.b[m], t);
      Sb[k] = t;
    }
    c[0] = -2.0 * (Sb[0] + Sb[4] * iz - Sb[5] * iy);
    c[1] = -2.0 * (Sb[1] - Sb[3] * iz + Sb[5] * ix);
    c[2] = -2.0 * (Sb[2] + Sb[3] * iy - Sb[4] * ix);
    const double* Wg = (const double*)(unsigned long long)(&P.W[0]) + 36 * me;  // rows 3 me ... of W: a per-lane (vector) load
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const double rx = r[j][0], ry = r[j][1], rz = r[j][2];
#pragma unroll
      for (int b = 0; b < 3; b++) {
        double SA[6];  // column b of S [I; [r_j]x]
#pragma unroll
        for (int k = 0; k < 6; k++) {
          if (b == 0) SA[k] = P.S[6 * k + 0] + P.S[6 * k + 4] * rz - P.S[6 * k + 5] * ry;
          if (b == 1) SA[k] = P.S[6 * k + 1] - P.S[6 * k + 3] * rz + P.S[6 * k + 5] * rx;
          if (b == 2) SA[k] = P.S[6 * k + 2] + P.S[6 * k + 3] * ry - P.S[6 * k + 4] * rx;
        }
        // column b of A_me^T S A_j
        const double t0 = SA[0] + SA[4] * iz - SA[5] * iy;
        const double t1 = SA[1] - SA[3] * iz + SA[5] * ix;
        const double t2 = SA[2] + SA[3] * iy - SA[4] * ix;
        Qr[0][3 * j + b] = 2.0 * (t0 + Wg[0 * 12 + 3 * j + b]);
        Qr[1][3 * j + b] = 2.0 * (t1 + Wg[1 * 12 + 3 * j + b]);
        Qr[2][3 * j + b] = 2.0 * (t2 + Wg[2 * 12 + 3 * j + b]);
      }
    }
  }

  QC_DEV bool solve(CParams& P, const Wrench<1>&, const Cube<1>& C, uint32_t stance, int, double (&f)[3], double (&g)[3]) {
    // working set of the whole robot: every lane contributes its foot's six bits
    const uint32_t word = (uint32_t)group_or<4, true>((int)(encode_foot(C.sx[0], C.sy[0], C.sz[0]) << (6 * me)));
    double ax[4], ay[4], az[4], cx[4], cy[4], mx[4], my[4], fzfix[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool st = (stance >> j) & 1u;
      const int sx = dec2(word >> (6 * j)), sy = dec2(word >> (6 * j + 2)), sz = dec2(word >> (6 * j + 4));
      ax[j] = (st && sx == 0) ? 1.0 : 0.0;
      ay[j] = (st && sy == 0) ? 1.0 : 0.0;
      az[j] = (st && sz == 0) ? 1.0 : 0.0;
      mx[j] = P.mu * (double)sx;
      my[j] = P.mu * (double)sy;
      cx[j] = mx[j] * az[j];
      cy[j] = my[j] * az[j];
      fzfix[j] = st ? (sz > 0 ? P.fzmax : (sz < 0 ? P.fzmin : 0.0)) : 0.0;
    }
    // this lane's own coefficients (its foot's state is in C)
    const bool sti = (stance >> me) & 1u;
    const double axi = (sti && C.sx[0] == 0) ? 1.0 : 0.0, ayi = (sti && C.sy[0] == 0) ? 1.0 : 0.0, azi = (sti && C.sz[0] == 0) ? 1.0 : 0.0;
    const double mxi = P.mu * (double)C.sx[0], myi = P.mu * (double)C.sy[0];
    const double cxi = mxi * azi, cyi = myi * azi;
    const double fzfi = sti ? (C.sz[0] > 0 ? P.fzmax : (C.sz[0] < 0 ? P.fzmin : 0.0)) : 0.0;
    // block row me of H = T^T Q T + (I - D) and of gp = Q p + c
    double gp[3] = {c[0], c[1], c[2]};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const double onj = (j == me) ? 1.0 : 0.0;  // the identity of the fixed slots sits on the diagonal block
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const double t = Qr[a][3 * j] * mx[j] + Qr[a][3 * j + 1] * my[j] + Qr[a][3 * j + 2];
        gp[a] = __builtin_fma(fzfix[j], t, gp[a]);
      }
      double Xb[3][3];  // Q_(me,j) T_j
#pragma unroll
      for (int a = 0; a < 3; a++) {
        Xb[a][0] = Qr[a][3 * j] * ax[j];
        Xb[a][1] = Qr[a][3 * j + 1] * ay[j];
        Xb[a][2] = Qr[a][3 * j] * cx[j] + Qr[a][3 * j + 1] * cy[j] + Qr[a][3 * j + 2] * az[j];
      }
#pragma unroll
      for (int b = 0; b < 3; b++) {
        double h0 = axi * Xb[0][b];
        double h1 = ayi * Xb[1][b];
        double h2 = cxi * Xb[0][b] + cyi * Xb[1][b] + azi * Xb[2][b];
        if (b == 0) h0 += onj * (1.0 - axi);
        if (b == 1) h1 += onj * (1.0 - ayi);
        if (b == 2) h2 += onj * (1.0 - azi);
        X[(12 * (3 * me + 0) + 3 * j + b) * XS] = h0;
        X[(12 * (3 * me + 1) + 3 * j + b) * XS] = h1;
        X[(12 * (3 * me + 2) + 3 * j + b) * XS] = h2;
      }
    }
    X[(144 + 3 * me + 0) * XS] = -axi * gp[0];
    X[(144 + 3 * me + 1) * XS] = -ayi * gp[1];
    X[(144 + 3 * me + 2) * XS] = -(cxi * gp[0] + cyi * gp[1] + azi * gp[2]);
    wave_sync();
    double L[78], y[12];
#pragma unroll
    for (int r = 0; r < 12; r++)
#pragma unroll
      for (int cc = 0; cc <= r; cc++) L[r * (r + 1) / 2 + cc] = X[(12 * r + cc) * XS];
#pragma unroll
    for (int k = 0; k < 12; k++) y[k] = X[(144 + k) * XS];
    wave_sync();  // the next recalculation (or the next robot's setup) rewrites the tile
    const bool ok = ldlt_solve<12>(L, y);
    // forces of all feet (f = T y + p), then this foot's rows of g = Q f + c
    g[0] = c[0]; g[1] = c[1]; g[2] = c[2];
    double fo[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const double fz = __builtin_fma(az[j], y[3 * j + 2], fzfix[j]);
      const double fx = __builtin_fma(ax[j], y[3 * j], mx[j] * fz);
      const double fy = __builtin_fma(ay[j], y[3 * j + 1], my[j] * fz);
#pragma unroll
      for (int a = 0; a < 3; a++) g[a] = __builtin_fma(Qr[a][3 * j], fx, __builtin_fma(Qr[a][3 * j + 1], fy, __builtin_fma(Qr[a][3 * j + 2], fz, g[a])));
      const bool own = j == me;
      fo[0] = own ? fx : fo[0];
      fo[1] = own ? fy : fo[1];
      fo[2] = own ? fz : fo[2];
    }
    f[0] = fo[0]; f[1] = fo[1]; f[2] = fo[2];
    return ok;
  }
};

}  // namespace qc
