/* qc_balance.h - C ABI of the MI355X-native batched balance controller.
 *
 * Drop-in boundary for ONE path of bostoncleek/quadruped_control:
 *   quadruped_controller::BalanceController::control()
 *     quadruped_controller/include/quadruped_controller/balance_controller.hpp:85-107
 *     quadruped_controller/src/quadruped_controller/balance_controller.cpp:70-330
 * i.e. PD wrench law -> single-rigid-body Newton-Euler map -> 12-variable
 * friction-cone QP (solved there by qpOASES SQProblem, balance_controller.cpp:177-210)
 * -> body-frame ground reaction forces.  One *instance* = one robot tick; the
 * batch axis is "independent robots".  All arithmetic is FP64, leg order is
 * RL, FL, RR, FR (commander_node.cpp:61), matrices are row-major.
 *
 * Plain C: pointers and sizes only, no C++/torch types, no exceptions.
 * The library is HIP-only (gfx950); there is no CPU fallback behind this ABI.
 */
#ifndef QC_BALANCE_H
#define QC_BALANCE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QC_ABI_VERSION 6

/* Replaces the constructor arguments of BalanceController
 * (balance_controller.hpp:85-88; defaults in commander_node.cpp:289-334 and
 * quadruped_simulation/config/mit_cheetah_config.yaml:66-99). */
typedef struct qc_params {
  double mu;       /* friction coefficient                                  */
  double mass;     /* total mass (kg)                                       */
  double fzmin;    /* min normal force (N), 0 <= fzmin <= fzmax             */
  double fzmax;    /* max normal force (N)                                  */
  double Ib[9];    /* trunk inertia, body frame (3x3)                       */
  double S[36];    /* SPD weight on the wrench residual (6x6)               */
  double W[144];   /* SPD weight on the forces (12x12)                      */
  double kff[6];   /* feed-forward gains                                    */
  double kp_p[3];  /* COM position Kp                                       */
  double kd_p[3];  /* COM linear-velocity Kd                                */
  double kp_w[3];  /* COM orientation Kp                                    */
  double kd_w[3];  /* COM angular-velocity Kd                               */
  int32_t max_iter; /* working-set recalculation cap; <=0 -> 200 (nWSR_, balance_controller.cpp:85); at most QC_MAX_ITER_LIMIT */
  int32_t reserved;
} qc_params;

/* Replaces the per-call arguments of control() (balance_controller.hpp:104-107),
 * one row per robot.  DEVICE pointers for qc_control_batch, HOST pointers for
 * qc_control_batch_host.  8-byte aligned; `stance` may be NULL = make_stance_gait()
 * (gait.cpp:24-34, the default argument of control()). */
typedef struct qc_batch_in {
  const double* Rwb;     /* [n][9]  rotation world<-base                     */
  const double* Rwb_d;   /* [n][9]  desired rotation                         */
  const double* x;       /* [n][3]  COM position                             */
  const double* xdot;    /* [n][3]  COM linear velocity                      */
  const double* w;       /* [n][3]  COM angular velocity                     */
  const double* x_d;     /* [n][3]  desired COM position                     */
  const double* xdot_d;  /* [n][3]  desired COM linear velocity              */
  const double* w_d;     /* [n][3]  desired COM angular velocity             */
  const double* feet;    /* [n][4][3] foot positions, body frame (FootholdMap, types.hpp:108) */
  const uint8_t* stance; /* [n][4]  LegState per leg: 1 stance, 0 swing (GaitMap, types.hpp:91-100) */
  /* ABI v2, optional (NULL = unused).  Joint angles [n][4][3] (RL,FL,RR,FR x hip,thigh,calf; JointStatesMap,
   * types.hpp:127).  When given, the foot positions are computed on the device by the reference's forward
   * kinematics (kinematics.cpp:81-103, what commander_node.cpp:383-384 does before control()) and `feet`
   * is ignored (may be NULL). */
  const double* joint_q;
  /* ABI v2, optional.  Per-leg gait phases [n][4] in [0,1) (GaitScheduler::phases_, gait.cpp:113-123).
   * When given (and `stance` is NULL) the contact state is derived on the device by the reference's rule
   * (GaitScheduler::phase, gait.cpp:125-134): stance iff 0 <= phase <= stance_phase, each comparison with the
   * 1e-12 slack of math::almost_equal.  `gait_duty` [n] = stance_phase = t_stance / (t_swing + t_stance)
   * (gait.cpp:45) per robot, or NULL to use the value installed with qc_set_gait (default 0.8/0.98,
   * mit_cheetah_config.yaml:17-18). */
  double* gait_phase;   /* read-only unless gait_dt is given (then IN/OUT, see below) */
  const double* gait_duty;
  /* ABI v2, optional; all three or none, need joint_q and joint_tau.  Swing-leg references as the reference's
   * FootTrajectoryManager::referenceState() returns them (WORLD frame foot position / velocity, [n][4][3], read
   * for swing legs only) and the measured joint velocities [n][4][3].  The swing legs' entries of joint_tau
   * then hold the reference's swing-leg torque (commander_node.cpp:482-504, 514-526):
   *   p_b = Rwb^T pos - x (sic), v_b = Rwb^T vel, q_ref = legInverseKinematics(p_b) (kinematics.cpp:117-160),
   *   qdot_ref = legJacobianInverse(q_ref) v_b (kinematics.cpp:190-204),
   *   tau = kp o wrapPI(wrap2PI(q_ref) - wrap2PI(q)) + kd o (qdot_ref - qdot) + kff (joint_controller.cpp:21-39),
   * clamped like the stance torques; it does not depend on the QP's status. */
  const double* swing_pos;
  const double* swing_vel;
  const double* joint_qdot;
  /* ABI v2, optional, IN/OUT: persistent per-robot swing-planning state [n] (zero-initialise with
   * qc_swing_state_init before the first tick).  When given (needs joint_q, joint_qdot, gait_phase and
   * joint_tau; swing_pos/swing_vel must be NULL) the swing references are generated on the device the way
   * the reference's loop does it (commander_node.cpp:432-471): FootPlanner::positions / updateStates /
   * singleFoot (foot_planner.cpp:46-157: replan a foothold when a leg goes stance -> swing, Raibert + LIP
   * heuristic), FootTrajectoryManager::referenceStates / referenceState and the sextic FootTrajectory
   * (trajectory.cpp:220-277, 308-388), then fed to the swing-leg torque chain above. */
  struct qc_swing_state* swing_state;
  /* ABI v3, optional: the gait clock.  Time step [n] (seconds) since the robot's previous tick.  When given,
   * `gait_phase` is IN/OUT: at the start of the robot's tick its four phases advance the way
   * GaitScheduler::update does (gait.cpp:113-123): phase += 1 / (t_swing + t_stance) * dt, wrapped by
   * fmod(., 1), with the periods installed by qc_set_gait - and the contact rule, the foothold planner and
   * the swing trajectories of this tick see the advanced phases.  The buffer behind `gait_phase` is written
 * (which is why that member is not const). */
  const double* gait_dt;
} qc_batch_in;

/* FootPlanner::state_map_ (foot_planner.hpp) + FootTrajectoryManager::traj_map_ (trajectory.hpp), per robot. */
typedef struct qc_swing_state {
  int32_t leg_state[4];  /* LegState seen at the previous tick, -1 = none yet (state_map_ empty)     */
  int32_t has_traj[4];   /* leg has an entry in traj_map_                                            */
  double p_start[12];    /* [leg][xyz] world-frame trajectory start  (FootTrajBounds, types.hpp:52-66) */
  double p_final[12];    /* [leg][xyz] world-frame planned foothold                                   */
} qc_swing_state;

/* Replaces the returned ForceMap (types.hpp:119; balance_controller.cpp:218-232). */
typedef struct qc_batch_out {
  double* grf_body;     /* [n][4][3] = -Rwb^T f_world for stance legs; 0 for swing legs
                           (the reference omits swing legs from the map); 0 if status != 0 */
  int32_t* status;      /* [n] qc_status; != 0 is the reference's "empty ForceMap" (balance_controller.cpp:182-216) */
  uint32_t* active_set; /* [n] optional (may be NULL): optimal working set, feed back as `warm` next tick */
  int32_t* iterations;  /* [n] optional (may be NULL): working-set recalculations used */
  /* ABI v2, optional (NULL = unused; needs qc_batch_in.joint_q).  Stance-leg joint torques [n][4][3]:
   * tau = J(q_leg)^T f_body (QuadrupedKinematics::jacobianTransposeControl, kinematics.cpp:219-231 with
   * legJacobian :162-188), clamped to [tau_min, tau_max] as commander_node.cpp:523-526; failed instances get 0.
   * Swing legs get 0 as well unless swing references are given (qc_batch_in.swing_pos / swing_vel / joint_qdot, or
   * swing_state): then their entries hold the reference's swing-leg joint PD torque (commander_node.cpp:482-504),
   * whatever the QP's status. */
  double* joint_tau;
} qc_batch_out;

/* Kinematic constants of QuadrupedKinematics::QuadrupedKinematics() (kinematics.cpp:20-47) and the torque
 * limits of commander_node.cpp:324-325.  qc_create installs the reference's values. */
typedef struct qc_kinematics {
  double hip[12];   /* [leg][xyz] translation base -> hip   (trans_rl/fl/rr/fr)            */
  double links[12]; /* [leg][l1,l2,l3] signed link lengths  (left_links / right_links)     */
  double tau_min;   /* balance_control/torque_min (-20)                                     */
  double tau_max;   /* balance_control/torque_max (+20)                                     */
  double jc_kff[3]; /* joint_control/kff (0,0,0)      swing-leg joint PD, commander_node.cpp:314-341, */
  double jc_kp[3];  /* joint_control/kp  (40,40,50)   mit_cheetah_config.yaml:50-53                   */
  double jc_kd[3];  /* joint_control/kd  (1,1,1)                                                      */
  double planner_hip[12]; /* [leg][xyz] base -> thigh used by FootPlanner (foot_planner.cpp:27-42)          */
  double planner_k;       /* Raibert feedback gain k_ (foot_planner.cpp:25, 0.01)                            */
  double swing_height;    /* gait/height: apex of the swing trajectory (commander_node.cpp:247, 0.08)        */
} qc_kinematics;

typedef enum qc_status {
  QC_SOLVED = 0,
  QC_MAX_ITER = 1,   /* RET_MAX_NWSR_REACHED analogue */
  QC_INFEASIBLE = 2, /* kept for ABI completeness; cannot occur for 0 <= fzmin <= fzmax */
  QC_NOT_PD = 3      /* Hessian not positive definite / non-finite input (balance_controller.cpp:155-158 only logs) */
} qc_status;

/* return codes of the entry points */
#define QC_OK 0
#define QC_ERR_INVALID (-1) /* bad argument (see qc_last_error)  */
#define QC_ERR_HIP (-2)     /* HIP runtime error                 */
#define QC_ERR_NO_DEVICE (-3)
#define QC_ERR_ABI (-4)     /* caller and library were built against different revisions of this header */
#define QC_MAX_ITER_LIMIT 65535 /* largest recalculation cap (qc_params.max_iter, "max_iter" tuning key) */

typedef struct qc_handle qc_handle;

/* BalanceController::BalanceController (balance_controller.cpp:70-96).
 * `device` = HIP device ordinal.  A handle is used by one thread at a time
 * (like the reference object, whose control() mutates `mutable` members,
 * balance_controller.hpp:161-176); distinct handles are independent.
 *
 * ABI v6: the EXPORTED constructor is qc_create_abi, which also takes the caller's view of this header (QC_ABI_VERSION and
 * the sizes of the three structs that cross the boundary) and refuses a mismatch with QC_ERR_ABI before anything is read -
 * qc_batch_in has grown with every revision and carries no size field, so a caller built against an older header would
 * otherwise have its struct read past the end.  qc_create is the inline wrapper below: C and C++ callers keep writing
 * qc_create(&params, device, &handle) and cannot skip the guard; the library no longer exports a symbol of that name, so a
 * binary built against ABI <= 5 fails to link / load instead of running.  Bindings without a C compiler (ctypes, cgo, JNI)
 * call qc_create_abi with the sizes of THEIR mirror structs. */
int qc_create_abi(const qc_params* params, int device, qc_handle** out, int abi_version, size_t sizeof_params,
                  size_t sizeof_batch_in, size_t sizeof_batch_out);
void qc_destroy(qc_handle* h);

/* control() for n robots, device-resident inputs/outputs, asynchronous on
 * `stream` (a hipStream_t, NULL = default stream).  `warm` = device array [n]
 * of active_set words from the previous tick (the qpOASES hotstart analogue,
 * balance_controller.cpp:191-202) or NULL for a cold start. */
int qc_control_batch(qc_handle* h, size_t n, const qc_batch_in* in, const uint32_t* warm,
                     const qc_batch_out* out, void* stream);

/* Same with HOST pointers: stages through handle-owned device buffers and
 * synchronises before returning. */
int qc_control_batch_host(qc_handle* h, size_t n, const qc_batch_in* in, const uint32_t* warm,
                          const qc_batch_out* out);

/* One robot, host arguments laid out exactly like control()'s parameter list;
 * what the C++ BalanceController adapter (include/qc_balance_controller.hpp) calls.
 * Like the reference object (mutable SQProblem, balance_controller.hpp:161; init()/hotstart(),
 * balance_controller.cpp:177-202) the handle keeps the working set of its previous successful call
 * and starts the next one from it; the result is the same unique minimiser either way. */
int qc_control(qc_handle* h, const double* Rwb, const double* Rwb_d, const double* x,
               const double* xdot, const double* w, const double* x_d, const double* xdot_d,
               const double* w_d, const double* feet, const uint8_t* stance, double* grf_body,
               int32_t* status);

/* Kinematic model used by the joint_q / joint_tau extension; NULL restores the reference's constants.
 * qc_set_kinematics / qc_set_gait are configuration calls: they wait for all work on the device
 * (hipDeviceSynchronize) before rewriting the constants, so no launch in flight - on any stream - sees a
 * half-written copy. */
void qc_default_kinematics(qc_kinematics* out);
int qc_set_kinematics(qc_handle* h, const qc_kinematics* kin);
/* Default stance_phase for qc_batch_in.gait_phase: t_stance / (t_swing + t_stance), GaitScheduler ctor gait.cpp:36-46. */
int qc_set_gait(qc_handle* h, double t_swing, double t_stance);
/* Host helper: the "nothing planned yet" value of qc_swing_state for n robots (host memory). */
void qc_swing_state_init(struct qc_swing_state* states, size_t n);

/* Thread-local message of the last failing call (ROS_ERROR replacement,
 * balance_controller.cpp:157,184,199,214). */
const char* qc_last_error(void);

/* Introspection: which device formulation the handle selected
 * ("diagW-6x6-uniform", "diagW-6x6" or "dense-12x12"), and ABI version. */
const char* qc_kernel_name(const qc_handle* h);
int qc_abi_version(void);
/* ABI v5.  Guard for callers compiled separately from the library: pass QC_ABI_VERSION, sizeof(qc_params),
 * sizeof(qc_batch_in) and sizeof(qc_batch_out) as the CALLER's header defines them; anything but QC_OK (QC_ERR_ABI, with
 * the two sides spelled out in qc_last_error) means the structs this library reads are not the ones the caller fills
 * (qc_batch_in has grown with every revision and carries no size field) and no other entry point may be used.  The
 * C++ adapter's constructor and the Python loader call it (it needs no device); since ABI v6 the constructor performs the
 * same check itself (qc_create_abi), so this entry point is a convenience for an early, device-free diagnosis.
 * (The reference's constructor contract, balance_controller.hpp:85-88, has no analogue: it is header-only C++.) */
int qc_check_abi(int abi_version, size_t sizeof_params, size_t sizeof_batch_in, size_t sizeof_batch_out);
#define QC_CHECK_ABI() qc_check_abi(QC_ABI_VERSION, sizeof(qc_params), sizeof(qc_batch_in), sizeof(qc_batch_out))

/* ABI v4.  Which kernel instantiation a batch of n robots would run on (kin = joint_q given, warm = warm-start
 * words given): lanes per robot, kernel mode (0 persistent waves with lane refill, 1 one fill per wave, 2 one fill
 * and one wave per SIMD with register-resident constants; batches that leave SIMDs idle even so race 2 or 4 pivoting
 * strategies per robot there, `strategies`; 3 paired waves - two one-lane waves per workgroup, the last to arrive
 * finishes both waves' stragglers: 6x6 forms from 524 288 robots on), form (0 uniform 6x6, 1 general 6x6, 2 dense 12x12),
 * robots per wave, grid size, and the workgroups of that kernel the device holds at once (the occupancy query the
 * heuristics use).  (Mode 0 is reported by development builds only: the default library runs every form as one-fill workgroups.) */
typedef struct qc_launch_info {
  int32_t lanes_per_robot;
  int32_t mode;
  int32_t form;
  int32_t strategies; /* four lanes per robot, one-fill kernels: pivoting strategies racing per robot (1, 2 or 4) */
  int64_t chunk;
  int64_t blocks;
  int64_t resident_workgroups;
  int64_t lds_bytes;
} qc_launch_info;
int qc_query_launch(qc_handle* h, size_t n, int kin, int warm, qc_launch_info* out);

/* ABI v4.  Development / test interface: explicit overrides of the launch heuristics and solver constants (the
 * library reads NO environment variables).  Keys: "group" (lanes per robot: 0 = heuristic, 1, 2, 4), "one_fill"
 * (-1 heuristic, 0 persistent waves, 1 one-fill workgroups; the persistent kernels exist only in development builds,
 * -DQC_PERSISTENT_6X6=1: elsewhere a launch with 0 fails with QC_ERR_INVALID), "chunk" (robots per wave, 0 = heuristic;
 * beyond one fill only where persistent kernels exist - QC_ERR_INVALID at launch otherwise),
 * "wave_slots" (resident workgroups assumed, 0 = occupancy query), "refill_t", "rounds_cold", "rounds_warm",
 * "race" (-1 heuristic; 0 or 1: one strategy per robot; 2, 4: at most that many racing in the 4-lane one-fill kernels.  With the
 * race on - the default for cold batches - a wave forks its last running robots onto idle lanes with a second drop rule, so a robot's
 * `iterations` (and, under a small max_iter, QC_SOLVED vs QC_MAX_ITER) depend on which robots share its wave, i.e. on the batch
 * layout; the forces agree to the KKT tolerance either way.  0 is the deterministic mode: one walk per robot whatever its neighbours),
 * "pair" (-1 heuristic, 0 never, 1 whenever one lane per robot on a 6x6 form: the paired-waves kernel, mode 3), "pair_th"
 * (its hand-over threshold, <= 32), "pair_refill" (free lane groups that trigger a refill, 1 ... 16), "pair_solo" (0: pairs in the last round of workgroups too),
 * "force_general" / "force_dense" (run the more general formulation on weights that would allow the
 * specialised one; same minimiser), "auto_dense" (1, default: a handle whose max diag(S) / min diag(W) exceeds 3e8 - regularisation
 * weights 300 times further below S than the reference's - runs the dense 12x12 form although its W is diagonal, because the 6x6 dual
 * forms lose digits that matter there, eps (S/w) |b|; 0: it keeps the 6x6 form), "clamp_steps" (clamp steps a cold-started robot takes before its first ratio test in
 * the one-fill kernels; 0 = the kernel's rule: five on one or two lanes per robot, one on four),
 * "tol_d" (relative multiplier tolerance), "polish" (1, default: the first time a robot would be accepted while its
 * smallest multiplier lies inside the noise band +-tol_d |grad|, that face is released once instead; 0: accept at -tol_d |grad|
 * straight away, round 5's rule - up to tol_d |grad| / (2 w) from the minimiser for very small W), "max_iter" (<= 0: back to qc_params.max_iter), "probe_batch_load"
 * (1: skip the solver iterations - load, assemble, store only; every robot then reports QC_MAX_ITER; 0: back to the
 * handle's own cap).  "force_general" / "force_dense" / "max_iter" / "probe_batch_load" restore what qc_create was given,
 * whatever the order of the calls.
 * Calls that change device constants synchronise the device first.  Returns QC_ERR_INVALID for an unknown key. */
int qc_set_tuning(qc_handle* h, const char* key, double value);

#ifdef __cplusplus
}
#endif

/* The constructor every C / C++ caller uses (see qc_create_abi above): the ABI guard travels with it. */
static inline int qc_create(const qc_params* params, int device, qc_handle** out) {
  return qc_create_abi(params, device, out, QC_ABI_VERSION, sizeof(qc_params), sizeof(qc_batch_in), sizeof(qc_batch_out));
}
#endif /* QC_BALANCE_H */
