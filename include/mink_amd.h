/*
 * mink_amd -- C ABI of the MI355X batched differential-IK engine.
 *
 * The reference (priyasundaresan/mink, a vendored copy of mink 0.0.5) is pure
 * Python on top of MuJoCo and qpsolvers; it has no FFI of its own.  Each entry
 * point below replaces one Python-level interface of the reference's hot path
 * (cited per function) with a batched device routine: every array gains a
 * leading batch dimension B and lives in GPU memory owned by the caller.
 *
 * Conventions
 *   - all pointers named d_* are DEVICE pointers (hipMalloc / torch ROCm tensors),
 *     row-major, contiguous, 8-byte aligned; everything else is host memory that
 *     is only read during the call;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *     asynchronous on it and never synchronise;
 *   - functions return MK_OK or a negative error code, never throw; the message of
 *     the last failure on the calling thread is available from mk_last_error();
 *   - a model / problem handle belongs to the device that was current when it was
 *     created; batch calls made while another device is current fail with
 *     MK_ERR_INVALID (one process per GPU never meets this); handles are immutable
 *     after creation and may be used from several threads / streams at once;
 *   - per-problem solver outcome goes to an int32 status array:
 *       0 solved, 1 iteration limit (or an answer of the general solver that fails its final check
 *       of the active rows), 2 infeasible constraints, 3 non-finite input
 *     (a NaN / Inf anywhere in the problem's q row or targets: its velocity is zero and
 *     the iterated entries leave its q as it was; other problems are unaffected; with option
 *     "f32" also a Hessian that float32 arithmetic cannot factor).
 *   - quaternions are (w, x, y, z); poses are (qw, qx, qy, qz, x, y, z) like
 *     mink.SE3.wxyz_xyz (mink/lie/se3.py:20-24); twists are (v; omega).
 */
#ifndef MINK_AMD_H_
#define MINK_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MK_OK 0
#define MK_ERR_INVALID (-1)     /* bad argument / inconsistent descriptor */
#define MK_ERR_UNSUPPORTED (-2) /* valid request outside what the kernels implement */
#define MK_ERR_HIP (-3)         /* a HIP runtime call failed */

/* joint types: MuJoCo mjtJoint values, as assumed by mink/constants.py:27-34 */
enum { MK_JNT_FREE = 0, MK_JNT_BALL = 1, MK_JNT_SLIDE = 2, MK_JNT_HINGE = 3 };
/* frame types: mink/constants.py:3 SUPPORTED_FRAMES */
enum { MK_FRAME_BODY = 0, MK_FRAME_GEOM = 1, MK_FRAME_SITE = 2 };
/* task types: mink/tasks/{frame_task,relative_frame_task,posture_task,com_task}.py;
 * DampingTask (mink/tasks/damping_task.py) is a posture task with gain 0 and
 * target qpos0. */
enum { MK_TASK_FRAME = 0, MK_TASK_RELATIVE_FRAME = 1, MK_TASK_POSTURE = 2, MK_TASK_COM = 3 };
/* limit types: mink/limits/{configuration_limit,velocity_limit,collision_avoidance_limit}.py */
enum { MK_LIMIT_CONFIGURATION = 0, MK_LIMIT_VELOCITY = 1, MK_LIMIT_COLLISION = 2 };

/* Flat kinematic tree; field names follow mujoco.MjModel, which is what the
 * reference reads (mink/configuration.py:86-97, mink/limits/configuration_limit.py:41-56). */
typedef struct mk_model_desc {
  int32_t nq, nv, nbody, njnt, ngeom, nsite;
  const int32_t* body_parentid; /* [nbody] */
  const int32_t* body_jntadr;   /* [nbody] first joint or -1 */
  const int32_t* body_jntnum;   /* [nbody] */
  const int32_t* body_mocapid;  /* [nbody] -1 if not mocap */
  const double* body_pos;       /* [nbody*3] */
  const double* body_quat;      /* [nbody*4] normalised */
  const double* body_mass;      /* [nbody] */
  const double* body_ipos;      /* [nbody*3] */
  const int32_t* jnt_type;      /* [njnt] */
  const int32_t* jnt_qposadr;   /* [njnt] */
  const int32_t* jnt_dofadr;    /* [njnt] */
  const int32_t* jnt_bodyid;    /* [njnt] */
  const int32_t* jnt_limited;   /* [njnt] 0/1 */
  const double* jnt_pos;        /* [njnt*3] */
  const double* jnt_axis;       /* [njnt*3] */
  const double* jnt_range;      /* [njnt*2] */
  const double* qpos0;          /* [nq] */
  const int32_t* dof_parentid;  /* [nv] */
  const int32_t* dof_jntid;     /* [nv] */
  const int32_t* site_bodyid;   /* [nsite] */
  const double* site_pos;       /* [nsite*3] */
  const double* site_quat;      /* [nsite*4] */
  const int32_t* geom_bodyid;   /* [ngeom] */
  const int32_t* geom_type;     /* [ngeom] MuJoCo mjtGeom */
  const double* geom_size;      /* [ngeom*3] */
  const double* geom_pos;       /* [ngeom*3] */
  const double* geom_quat;      /* [ngeom*4] */
  /* convex hulls of the meshes of mesh geoms (may all be 0 / NULL for a model without
   * meshes): vertices in the frame of the geom, as mj_geomDistance sees them */
  int32_t nmesh, nmeshvert;
  const int32_t* geom_dataid;   /* [ngeom] mesh id of a mesh geom, else -1 */
  const int32_t* mesh_vertadr;  /* [nmesh] */
  const int32_t* mesh_vertnum;  /* [nmesh] */
  const double* mesh_vert;      /* [nmeshvert*3] */
  const double* mesh_center;    /* [nmesh*3] centre of a bounding sphere of the hull */
  const double* mesh_radius;    /* [nmesh] */
} mk_model_desc;

/* One kinematic task.  Pose targets of frame tasks are per-problem inputs of the
 * solve calls; everything here is shared by the whole batch. */
typedef struct mk_task_desc {
  int32_t type;            /* MK_TASK_* */
  int32_t frame_type;      /* MK_FRAME_* (frame / relative frame) */
  int32_t frame_id;
  int32_t root_type;       /* relative frame only */
  int32_t root_id;
  int32_t ncost;           /* 6 (frame tasks), nv (posture), 3 (com) */
  double gain;             /* task gain alpha, mink/tasks/task.py:71 */
  double lm_damping;       /* mink/tasks/task.py:74 */
  const double* cost;      /* [ncost] already broadcast */
  const double* target_q;  /* posture: [nq] */
} mk_task_desc;

typedef struct mk_limit_desc {
  int32_t type;                      /* MK_LIMIT_* */
  int32_t n;                         /* velocity: number of dofs; collision: number of pairs */
  double gain;                       /* configuration (default 0.95) / collision (0.85) */
  double min_distance_from_limits;   /* configuration */
  const int32_t* index;              /* velocity: [n] dof ids; collision: [2n] geom id pairs */
  const double* vmax;                /* velocity: [n] */
  double minimum_distance_from_collisions; /* collision */
  double collision_detection_distance;     /* collision */
  double bound_relaxation;                  /* collision */
} mk_limit_desc;

typedef struct mk_model mk_model;
typedef struct mk_problem mk_problem;

typedef struct mk_problem_info {
  int32_t nq, nv;
  int32_t target_stride; /* doubles of per-problem target data (7 per frame / relative
                            frame task, 3 per com task, in task order) */
  int32_t n_rows;        /* rows of the stacked inequality matrix G (mink/solve_ik.py:40) */
  int32_t fused;         /* 1: the single-kernel structured path serves this problem */
  int32_t n_chain;       /* dofs with dense coupling in the fused path */
  int32_t staged;        /* 1: mk_solve_ik_batch runs the structured kernel on every problem, a
                            lane-per-problem kernel with the collision rows on those with a geom pair
                            that may be within its detection distance, and the general kernel on what
                            that one leaves (one frame task, optionally a relative frame task on the
                            same frame, posture tasks, any limits) */
  int32_t staged_rows;   /* staged problems: 1 when that lane-per-problem kernel takes part; 0 when the
                            model is too wide for its rows to fit the LDS of a compute unit (about
                            nv > 60 with a 12-dof chain): every problem with a pair in range then goes
                            straight to the general kernel */
} mk_problem_info;

/* Replaces mujoco.MjModel.from_xml_path + mujoco.MjData construction
 * (interactive_scripts/teleop_mouse.py:29-30, mink/configuration.py:49-51): copies
 * the tree to the device that is current at the time of the call. */
int mk_model_create(const mk_model_desc* desc, mk_model** out);
void mk_model_destroy(mk_model* model);

/* Replaces constructing the task and limit objects handed to mink.solve_ik
 * (interactive_scripts/teleop_mouse.py:46-69; mink/solve_ik.py:28-29 builds the
 * default ConfigurationLimit on every call, here it is built once). nlimit = 0
 * means unconstrained (limits=[]). */
int mk_problem_create(const mk_model* model, const mk_task_desc* tasks, int32_t ntask,
                      const mk_limit_desc* limits, int32_t nlimit, mk_problem** out);
void mk_problem_destroy(mk_problem* problem);
int mk_problem_get_info(const mk_problem* problem, mk_problem_info* info);

/* PostureTask.set_target / set_target_from_configuration (mink/tasks/posture_task.py:62-85)
 * on an existing problem: replaces the target of posture task number `task` (index into the
 * task list given to mk_problem_create) by target_q [nq] (host memory), without rebuilding the
 * problem.  The update is ordered on `stream` like a launch: solves queued before it see the
 * old target, solves queued after it the new one; it must not run concurrently with solves
 * of the same problem on other streams. */
int mk_problem_set_posture_target(mk_problem* problem, int32_t task, const double* target_q, void* stream);

/* mink.solve_ik without the limit check (mink/solve_ik.py:100-105): build the QP
 * of every problem, solve it exactly and return v = dq / dt.
 *   d_q [B, nq], d_targets [B, target_stride], d_v [B, nv], d_status [B]. */
int mk_solve_ik_batch(const mk_problem* problem, const double* d_q, const double* d_targets,
                      double dt, double damping, int32_t B, double* d_v, int32_t* d_status,
                      void* stream);

/* The callers' loop (interactive_scripts/teleop_mouse.py:105-121,
 * envs/mj_env.py:266-276): up to max_iters x (solve_ik -> integrate_inplace ->
 * frame-task error), leaving the loop of a problem once its first frame task has
 * |e[:3]| <= pos_tol and |e[3:]| <= ori_tol (pass negative tolerances to run
 * exactly max_iters solves).  d_q is updated in place; optional outputs (may be
 * NULL): d_v last velocity [B, nv], d_iters solves executed [B], d_err final
 * frame-task error [B, 6].  Problems of the fused path run the loop in one launch;
 * for any other task / limit set the same loop is issued as launches of the general
 * kernels (error, solve, integrate per pass, finished problems skipped), and d_v,
 * d_iters and d_err are then required: they hold the state of the loop. */
int mk_solve_ik_iterate_batch(const mk_problem* problem, double* d_q, const double* d_targets,
                              double dt, double damping, int32_t B, int32_t max_iters,
                              double pos_tol, double ori_tol, double* d_v, int32_t* d_iters,
                              double* d_err, int32_t* d_status, void* stream);

/* Waypoint generator parameters: envs/robot_utils.py:113-118 (LinearWaypointReachConfig;
 * its rot_step_size is not used by the reference's step() and is kept for layout only). */
typedef struct mk_waypoint_config {
  double pos_threshold, pos_step_size, rot_threshold, rot_step_size;
} mk_waypoint_config;

/* Kinematic part of the reference's move_to (envs/mj_env.py:193-276): up to max_steps x
 * { pose of the problem's frame task; LinearWaypointReach.step (envs/robot_utils.py:131-160:
 * bounded linear position step, half-way slerp, reached test on the position distance and
 * the x-y-z Euler angles of goal * current^-1); the callers' IK loop of
 * mk_solve_ik_iterate_batch towards the commanded pose }, a problem stopping after the
 * step in which its generator reports the goal reached.  The configuration is taken to
 * track its IK solution (the reference hands it to the MuJoCo simulation instead).
 * d_q [B, nq] is updated in place; d_goals [B, 7] goal poses (wxyz_xyz) of the frame task;
 * outputs (d_steps, d_reached, d_solves may be NULL): outer steps taken, 1 if reached,
 * IK solves executed, and the worst solver status [B].  Needs a problem of the fused path
 * with a single frame task. */
int mk_move_to_batch(const mk_problem* problem, double* d_q, const double* d_goals,
                     const mk_waypoint_config* wp, double dt, double damping, int32_t B,
                     int32_t max_steps, int32_t max_iters, double pos_tol, double ori_tol,
                     int32_t* d_steps, int32_t* d_reached, int32_t* d_solves, int32_t* d_status,
                     void* stream);

/* The same rollout for a problem outside the fused path that has a frame task (collision
 * limits, relative frame tasks, several tasks, any tree): every outer step is issued as the passes
 * of mk_solve_ik_iterate_batch's general form, so all max_steps x (max_iters + 1) passes are
 * launched and the call lasts accordingly; nothing is allocated and nothing synchronises.
 * `task_index` names the driven task (an MK_TASK_FRAME of the problem).  d_targets [B,
 * target_stride] holds the targets of the other tasks and is working memory for the driven one
 * (its slot is overwritten with the commanded pose of each step; on return it holds the pose
 * commanded in the rollout's last step).  d_steps, d_reached, d_solves,
 * d_status [B] as above but required; d_v [B, nv], d_iters [B], d_err [B, 6] hold the state of the
 * loop and end as after the last step's mk_solve_ik_iterate_batch.
 * Replaces the loop of envs/mj_env.py:212-230 around envs/mj_env.py:266-276 for such problems. */
int mk_move_to_passes_batch(const mk_problem* problem, int32_t task_index, double* d_q, const double* d_goals,
                            double* d_targets, const mk_waypoint_config* wp, double dt, double damping, int32_t B,
                            int32_t max_steps, int32_t max_iters, double pos_tol, double ori_tol, int32_t* d_steps,
                            int32_t* d_reached, int32_t* d_solves, int32_t* d_status, double* d_v, int32_t* d_iters,
                            double* d_err, void* stream);

/* Configuration.integrate_inplace minus the FK refresh (mink/configuration.py:228-236,
 * i.e. mujoco.mj_integratePos): d_q [B, nq] <- d_q (+) dt * d_v [B, nv]. */
int mk_integrate_batch(const mk_model* model, double* d_q, const double* d_v, double dt,
                       int32_t B, void* stream);

/* Configuration.get_transform_frame_to_world (mink/configuration.py:157-185):
 * d_pose [B, 7] of frame (frame_type, frame_id) at d_q [B, nq]. */
int mk_fk_batch(const mk_model* model, const double* d_q, int32_t B, int32_t frame_type,
                int32_t frame_id, double* d_pose, void* stream);

/* Configuration.get_frame_jacobian (mink/configuration.py:112-155): body-frame
 * Jacobian d_jac [B, 6, nv], rows (linear; angular). */
int mk_frame_jacobian_batch(const mk_model* model, const double* d_q, int32_t B,
                            int32_t frame_type, int32_t frame_id, double* d_jac, void* stream);

/* Configuration.check_limits (mink/configuration.py:77-110): d_first_violation [B]
 * receives the id of the first joint outside [lower - tol, upper + tol], or -1. */
int mk_check_limits_batch(const mk_model* model, const double* d_q, double tol, int32_t B,
                          int32_t* d_first_violation, void* stream);

/* mink.build_ik (mink/solve_ik.py:43-65): dense d_P [B, nv, nv], d_c [B, nv],
 * d_G [B, n_rows, nv], d_h [B, n_rows] (G/h may be NULL when n_rows == 0).
 * Rows of inactive collision pairs are zero with h = +inf, as in the reference. */
int mk_build_qp_batch(const mk_problem* problem, const double* d_q, const double* d_targets,
                      double dt, double damping, int32_t B, double* d_P, double* d_c,
                      double* d_G, double* d_h, void* stream);

/* Task.compute_error / Task.compute_jacobian of task number `task` of the problem
 * (mink/tasks/task.py:81-103 and the subclasses): d_err [B, k], d_jac [B, k, nv] with
 * k = 6 (frame tasks), nv (posture) or 3 (com); either output may be NULL. */
int mk_task_eval_batch(const mk_problem* problem, int32_t task, const double* d_q, const double* d_targets,
                       int32_t B, double* d_err, double* d_jac, void* stream);

/* qpsolvers.solve_problem(Problem(P, q, G, h), solver) (mink/solve_ik.py:101):
 * min 1/2 x'Px + c'x s.t. Gx <= h for B independent dense problems, P positive
 * definite; n <= 64 and (4 n^2 + m n) * 8 bytes within the LDS budget (~150 KB).  d_x [B, n].
 * A P that is not positive definite or any non-finite entry gives status 3 and x = 0. */
int mk_qp_solve_batch(const double* d_P, const double* d_c, const double* d_G, const double* d_h,
                      int32_t n, int32_t m, int32_t B, double* d_x, int32_t* d_status,
                      void* stream);

/* Operations of mink.lie (mink/lie/so3.py, mink/lie/se3.py, mink/lie/base.py).  Operand
 * widths in doubles, (a, b) -> out: rotations are 4 (wxyz), poses 7 (wxyz_xyz), tangents
 * 3 / 6 (v; omega), matrices row-major. */
enum {
  MK_SO3_EXP = 0,          /* so3.py:158-173   (3)    -> 4  */
  MK_SO3_LOG = 1,          /* so3.py:176-191   (4)    -> 3  */
  MK_SO3_MULTIPLY = 2,     /* so3.py:148-151   (4, 4) -> 4  */
  MK_SO3_INVERSE = 3,      /* so3.py:136-137   (4)    -> 4  */
  MK_SO3_NORMALIZE = 4,    /* so3.py:139-140   (4)    -> 4  */
  MK_SO3_APPLY = 5,        /* so3.py:143-146   (4, 3) -> 3  */
  MK_SO3_AS_MATRIX = 6,    /* so3.py:111-114, adjoint so3.py:194   (4) -> 9 */
  MK_SO3_FROM_MATRIX = 7,  /* so3.py:79-84 (mju_mat2Quat)          (9) -> 4 */
  MK_SO3_LJAC = 8,         /* so3.py:200-212   (3)    -> 9  */
  MK_SO3_LJACINV = 9,      /* so3.py:214-226   (3)    -> 9  */
  MK_SO3_RJAC = 10,        /* base.py:147      (3)    -> 9  */
  MK_SO3_RJACINV = 11,     /* base.py:151      (3)    -> 9  */
  MK_SO3_JLOG = 12,        /* base.py:155      (4)    -> 9  */
  MK_SO3_RPLUS = 13,       /* base.py:109      (4, 3) -> 4  */
  MK_SO3_LPLUS = 14,       /* base.py:117      (4, 3) -> 4  */
  MK_SO3_RMINUS = 15,      /* base.py:113  a (-) b = log(b^-1 a)   (4, 4) -> 3 */
  MK_SO3_LMINUS = 16,      /* base.py:121  log(a b^-1)             (4, 4) -> 3 */
  MK_SO3_AS_RPY = 17,      /* so3.py:116-133   (4)    -> 3 (roll, pitch, yaw) */
  MK_SO3_FROM_RPY = 18,    /* so3.py:66-77     (3)    -> 4  */
  MK_SE3_EXP = 32,         /* se3.py:112-134   (6)    -> 7  */
  MK_SE3_LOG = 33,         /* se3.py:159-185   (7)    -> 6  */
  MK_SE3_MULTIPLY = 34,    /* se3.py:153-157   (7, 7) -> 7  */
  MK_SE3_INVERSE = 35,     /* se3.py:136-141   (7)    -> 7  */
  MK_SE3_NORMALIZE = 36,   /* se3.py:143-147   (7)    -> 7  */
  MK_SE3_APPLY = 37,       /* se3.py:149-151   (7, 3) -> 3  */
  MK_SE3_AS_MATRIX = 38,   /* se3.py:106-110   (7)    -> 16 */
  MK_SE3_FROM_MATRIX = 39, /* se3.py:68-75     (16)   -> 7  */
  MK_SE3_ADJOINT = 40,     /* se3.py:187-194   (7)    -> 36 */
  MK_SE3_LJAC = 41,        /* se3.py:199-207   (6)    -> 36 */
  MK_SE3_LJACINV = 42,     /* se3.py:210-218   (6)    -> 36 */
  MK_SE3_RJAC = 43,        /* base.py:147      (6)    -> 36 */
  MK_SE3_RJACINV = 44,     /* base.py:151      (6)    -> 36 */
  MK_SE3_JLOG = 45,        /* base.py:155      (7)    -> 36 */
  MK_SE3_RPLUS = 46,       /* base.py:109      (7, 6) -> 7  */
  MK_SE3_LPLUS = 47,       /* base.py:117      (7, 6) -> 7  */
  MK_SE3_RMINUS = 48,      /* base.py:113      (7, 7) -> 6  */
  MK_SE3_LMINUS = 49       /* base.py:121      (7, 7) -> 6  */
};

/* One operation of mink.lie on B independent elements: d_a [B, wa], d_b [B, wb] (NULL for
 * unary operations), d_out [B, wo].  The device functions are the ones the IK kernels use
 * for their frame tasks. */
int mk_lie_batch(int32_t op, const double* d_a, const double* d_b, int32_t B, double* d_out,
                 void* stream);
/* Operand widths of an operation; returns MK_ERR_INVALID for an unknown op. */
int mk_lie_op_shape(int32_t op, int32_t* wa, int32_t* wb, int32_t* wo);

/* Process-wide switches.  No counterpart in the reference (its solver backend is picked by the
 * `solver` string of mink.solve_ik, mink/solve_ik.py:68-77, and has no tuning surface); these
 * choose between equivalent kernel paths and serve the tests and tuning tools.  Every path
 * returns the same results to the stated tolerances, so a caller never needs them -- except
 * "f32", which trades precision:
 *   "f32"             0 (default): float64 arithmetic.  1: mk_solve_ik_batch, mk_solve_ik_iterate_batch and
 *                     mk_move_to_batch run the lane-per-problem kernels of the fused path in float32
 *                     arithmetic (arrays stay float64).  Bound for problems whose frame task has
 *                     lm_damping = 1 at damping >= 1e-3 (the callers' values): |v - v_f64|_inf <=
 *                     1e-3 max(1, |v|_inf) rad/s per step (measured 1.6e-4 on the TidyBot workload, 3.3e-4
 *                     worst over random parameter sets); otherwise the error grows like 1e-7 times the
 *                     condition number of the Hessian, and a factorisation that loses a pivot to rounding
 *                     gives status 3 with zero velocity.  Problems outside the fused path, or with
 *                     collision limits or a relative frame task, are refused (MK_ERR_UNSUPPORTED) while
 *                     the option is set.
 *   "coop"            -1 (default): problems of the fused path with B <= "coop_max_batch" run on the
 *                     cooperative kernels (16 lanes per problem: small batches, the 200 Hz loop of
 *                     interactive_scripts/teleop_mouse.py:99-125); 0 never; 1 always
 *   "coop_max_batch"  default 16384
 *   "handoff"         -1 (default): in a loop with an exit test on more problems than that, a wave of the
 *                     lane-per-problem kernel finishes its last 4 problems still in the loop in the
 *                     cooperative form (within the launch; the streaming and rollout kernels in a
 *                     second launch on the same stream); 0 never; n > 0 its last n
 *   "stream_blocks"   -1 (default): lane streaming of the iterated entry by batch size; 0 never;
 *                     n > 0 always, with n persistent waves
 *   "stream_refill"   default 16: waiting lanes before the streaming kernels hand out problems
 *   "staged"          0: problems with collision limits skip the staged solve (read by mk_problem_create)
 *   "lane_rows", "general_stage", "staged_loop"   0: skip that stage of the staged solve
 *                     ("lane_rows" = 2: only the first of its two launches, the form that solves with two rows)
 *                     (diagnostic; without "general_stage" the problems it would have solved
 *                     report status 1)
 * Unknown names give MK_ERR_INVALID.  Values are read at launch time with relaxed atomics. */
int mk_set_option(const char* name, int64_t value);
int mk_get_option(const char* name, int64_t* value);

/* Message of the last error on the calling thread ("" if none). */
const char* mk_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MINK_AMD_H_ */
