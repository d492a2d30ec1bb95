/* TEST INFRASTRUCTURE - CPU restatement (plain C, FP64) of the reference hot
 * path BalanceController::control().  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (quadruped_control_amd/, include/) never links or calls it.
 *
 * PARITY UNPINNED BY REFERENCE FIXTURES: the reference has no tests/golden
 * vectors for this path and its QP solve lives in qpOASES (README.md:102,
 * master @326a651), absent from /root/reference and this image together with
 * Armadillo, Drake/Eigen and ROS.  The assembly is restated line by line from
 * first-party reference code; the QP (strictly convex => unique minimiser) is
 * solved by a textbook primal active-set method on the LITERAL data the
 * reference hands to qpOASES (H, g, 20x12 C, lbA, ubA - two-sided rows,
 * swing legs as five dependent equality rows), and certified by KKT.
 *
 * Citations: BC.cpp = quadruped_controller/src/quadruped_controller/
 * balance_controller.cpp, BC.hpp = quadruped_controller/include/
 * quadruped_controller/balance_controller.hpp (under /root/reference).
 */
#ifndef BALANCE_ORACLE_H
#define BALANCE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Constructor arguments, BC.hpp:85-88.  Matrices row-major. */
typedef struct oracle_params {
  double mu, mass, fzmin, fzmax;
  double Ib[9], S[36], W[144];
  double kff[6], kp_p[3], kd_p[3], kp_w[3], kd_w[3];
  int max_iter; /* nWSR_, BC.cpp:85 (200) */
} oracle_params;

enum { ORACLE_OK = 0, ORACLE_MAX_ITER = 1, ORACLE_INFEASIBLE = 2, ORACLE_NOT_PD = 3 };

/* BC.cpp:107-161: everything handed to qpOASES.  H 12x12, g 12, C 20x12
 * (row-major, BC.cpp:30-41), lb/ub 20. feet = 4x3 RL,FL,RR,FR body frame,
 * stance[i] = LegState (1 stance, 0 swing). Also returns A (6x12), b (6). */
void oracle_assemble(const oracle_params* P, const double* Rwb, const double* Rwb_d, const double* x,
                     const double* xdot, const double* w, const double* x_d, const double* xdot_d,
                     const double* w_d, const double* feet, const unsigned char* stance, double* H,
                     double* g, double* C, double* lb, double* ub, double* A, double* b);

/* rigid3d.cpp:198-203 (Drake RotationMatrix::ToAngleAxis -> Eigen). */
void oracle_angle_axis_total(const double* R, double* out3);

/* min 1/2 f'Hf + g'f  s.t. lb <= C f <= ub  (BC.cpp:177-210).
 * lam[20] (optional, may be NULL): signed multipliers, >0 at upper, <0 at lower. */
int oracle_qp_solve(const double* H, const double* g, const double* C, const double* lb,
                    const double* ub, int max_iter, double* f, double* lam, int* iters);

/* KKT residuals of (f, lam): stationarity |Hf+g+C'lam|_inf, primal bound
 * violation, dual sign/complementarity violation. */
void oracle_kkt(const double* H, const double* g, const double* C, const double* lb, const double* ub,
                const double* f, const double* lam, double* stationarity, double* primal, double* dual);

/* control() end to end, BC.cpp:98-235.  grf_body[12]: stance legs
 * -Rwb' f_world (BC.cpp:218-232), swing legs 0 (the reference omits them
 * from the map).  Failure -> status != 0 and all-zero forces (the reference
 * returns an empty map, BC.cpp:182-216). */
int oracle_control(const oracle_params* P, const double* Rwb, const double* Rwb_d, const double* x,
                   const double* xdot, const double* w, const double* x_d, const double* xdot_d,
                   const double* w_d, const double* feet, const unsigned char* stance,
                   double* grf_body, double* f_world, int* iters);

/* n robots, arrays as in include/qc_balance.h (host memory). threads<=1: serial. */
/* checker accuracy switch: 1 (default) = the accepted point of every QP is recomputed in long double */
void oracle_set_refine(int on);
void oracle_control_batch(const oracle_params* P, long n, const double* Rwb, const double* Rwb_d,
                          const double* x, const double* xdot, const double* w, const double* x_d,
                          const double* xdot_d, const double* w_d, const double* feet,
                          const unsigned char* stance, double* grf_body, int* status, int* iters,
                          int threads);

/* ---- "next" rows of the scope table: the steps either side of control() ----
 * Kinematic model of QuadrupedKinematics::QuadrupedKinematics(), src/quadruped_controller/kinematics.cpp:20-47
 * (leg order RL, FL, RR, FR). */
typedef struct oracle_kinematics {
  double hip[12];   /* base -> hip translation per leg */
  double links[12]; /* signed (l1, l2, l3) per leg     */
  double tau_min, tau_max; /* commander_node.cpp:324-325 */
  double jc_kff[3], jc_kp[3], jc_kd[3]; /* swing-leg joint PD gains, commander_node.cpp:314-341 */
  double planner_hip[12], planner_k, swing_height; /* foot_planner.cpp:25-42, gait/height commander_node.cpp:247 */
  double t_swing, t_stance;                        /* gait/t_swing, gait/t_stance, commander_node.cpp:245-246 */
} oracle_kinematics;
void oracle_default_kinematics(oracle_kinematics* k);
/* forwardKinematics(leg, q), kinematics.cpp:81-103 */
void oracle_leg_fk(const oracle_kinematics* k, int leg, const double* q3, double* p3);
/* legJacobian(leg, q), kinematics.cpp:162-188 (row-major 3x3) */
void oracle_leg_jacobian(const oracle_kinematics* k, int leg, const double* q3, double* J9);
/* One tick as commander_node.cpp:383-384 + 507-526 chains it:
 * feet = FK(q); forces = control(...); tau = clamp(J^T f_body) for stance legs (kinematics.cpp:219-231),
 * 0 for swing legs and failed instances. */
void oracle_tick_batch(const oracle_params* P, const oracle_kinematics* K, long n, const double* Rwb,
                       const double* Rwb_d, const double* x, const double* xdot, const double* w,
                       const double* x_d, const double* xdot_d, const double* w_d, const double* joint_q,
                       const unsigned char* stance, double* feet_out, double* grf_body, double* joint_tau,
                       int* status, int threads);
/* legInverseKinematics(leg, foothold), kinematics.cpp:117-160 */
void oracle_leg_ik(const oracle_kinematics* k, int leg, const double* p3, double* q3);
/* arma::pinv of a 3x3, the fallback of legJacobianInverse (kinematics.cpp:196): SVD (one-sided Jacobi) with
 * Armadillo's default tolerance 3 * sigma_max * epsilon; Jp9 row-major.  Returns 0 if the SVD did not converge. */
int oracle_pinv3(const double* J9, double* Jp9);
/* the pseudo-inverse of the band where this build answers with pinv: rank by complete pivoting (at most 2, second pivot > 1e-9
 * of the first - the device's rule, qc_device.hpp pinv3_apply), values from the SVD truncated to that rank */
int oracle_cp_rank3(const double* J9);
int oracle_pinv3_band(const double* J9, double* Jp9);
/* swing legs so far on which arma::pinv's tolerance (what oracle_swing_torque applies) and the device's rank rule (oracle_cp_rank3)
 * keep a different number of singular values; reset != 0 clears the count after reading it */
long oracle_pinv_rule_disagreements(int reset);
/* Swing-leg torque of one leg as commander_node.cpp:482-504 + joint_controller.cpp:21-39 compute it (unclamped):
 * pos/vel = world-frame reference foot state, q/qdot = measured joint state of the leg. */
void oracle_swing_torque(const oracle_kinematics* k, int leg, const double* Rwb, const double* x, const double* pos,
                         const double* vel, const double* q3, const double* qdot3, double* tau3);
/* oracle_tick_batch plus the swing-leg torques merged in (commander_node.cpp:514-526). */
void oracle_tick_swing_batch(const oracle_params* P, const oracle_kinematics* K, long n, const double* Rwb,
                             const double* Rwb_d, const double* x, const double* xdot, const double* w,
                             const double* x_d, const double* xdot_d, const double* w_d, const double* joint_q,
                             const double* joint_qdot, const double* swing_pos, const double* swing_vel,
                             const unsigned char* stance, double* grf_body, double* joint_tau, int* status, int threads);

/* FootPlanner::state_map_ + FootTrajectoryManager::traj_map_ of one robot (same layout as qc_swing_state). */
typedef struct oracle_swing_state {
  int leg_state[4]; /* -1 = state_map_ empty */
  int has_traj[4];
  double p_start[12], p_final[12];
} oracle_swing_state;
/* One tick of the reference's swing-reference generation, commander_node.cpp:432-471 + 485-489:
 * foot planner (foot_planner.cpp:46-157), trajectory manager and sextic trajectory (trajectory.cpp:220-388).
 * feet_body = foot_actual_map; phase[4] = per-leg gait phases; outputs world-frame pos/vel [12]
 * (zeros for stance legs and for swing legs without a trajectory). */
void oracle_swing_references(const oracle_kinematics* K, oracle_swing_state* st, const double* Rwb, const double* x,
                             const double* xdot, const double* w, const double* xdot_d, const double* feet_body,
                             const unsigned char* stance, const double* phase, double* pos, double* vel);
/* GaitScheduler::update(dt), gait.cpp:113-123: phases[n][4] += 1 / (t_swing + t_stance) * dt[n], wrapped by fmod(., 1). */
void oracle_gait_update(const oracle_kinematics* K, long n, double* phases, const double* dt);
/* Full tick with on-the-fly swing references: stance from the gait rule, FK, planner/trajectories,
 * control(), J^T torques for stance legs, IK/J^-1/PD torques for swing legs. */
void oracle_tick_planned_batch(const oracle_params* P, const oracle_kinematics* K, long n, oracle_swing_state* states,
                               const double* Rwb, const double* Rwb_d, const double* x, const double* xdot,
                               const double* w, const double* x_d, const double* xdot_d, const double* w_d,
                               const double* joint_q, const double* joint_qdot, const double* gait_phase,
                               double* grf_body, double* joint_tau, int* status, int threads);

#ifdef __cplusplus
}
#endif
#endif
