// Device-side model tables and the launchers of the generic (any tree, any task /
// limit set) kernels: utility kernels in mk_generic.hip, the dense QP build + solve in
// mk_dense.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

// Dynamic LDS beyond the 64 KB default needs the kernel's limit raised.  The largest size
// already granted is remembered per kernel and device, so this is one runtime call per kernel
// and device (or per new high-water mark), not one per launch.  Raising the limit is serialised:
// two threads asking for different sizes must not leave the attribute at the smaller one.
struct MkLdsGrant {
  std::atomic<size_t> granted[32];
  std::mutex raise;
};
inline hipError_t mk_grant_lds(const void* kernel, size_t lds, MkLdsGrant& g) {
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds <= 64 * 1024) return hipSuccess;
  int dev = -1;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  const bool tracked = dev >= 0 && dev < 32;
  if (tracked && lds <= g.granted[dev].load(std::memory_order_acquire)) return hipSuccess;
  std::lock_guard<std::mutex> lock(g.raise);
  const size_t have = tracked ? g.granted[dev].load(std::memory_order_relaxed) : 0;
  if (lds <= have) return hipSuccess;
  if (hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess)
    return e;
  if (tracked) g.granted[dev].store(lds, std::memory_order_release);
  return hipSuccess;
}

struct MkDevModel {
  int32_t nq, nv, nbody, njnt, ngeom;
  int32_t npath;   // entries of path_data
  const int32_t *body_parentid, *body_jntadr, *body_jntnum, *body_mocapid;
  const double *body_pos, *body_quat, *body_mass, *body_ipos;
  const int32_t *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_range, *qpos0;
  const int32_t *dof_qadr, *dof_scalar;   // per dof: qpos address, 1 for hinge / slide dofs
  const int32_t* dof_ball;                // per dof: component 0..2 of a ball joint, else -1
  const int32_t *geom_bodyid, *geom_type;
  const double *geom_size, *geom_pos, *geom_quat;
  const int32_t *path_adr, *path_len, *path_data;  // bodies from the root to each body
  const uint64_t* body_dofmask;                     // per body: bit d set when dof d is on the path root -> body
  const double* mesh_vert;                          // hull vertices of all meshes [nmeshvert * 3]
};

constexpr int MK_DENSE_MAX_TASKS = 8;
constexpr int MK_DENSE_MAX_LIMITS = 8;
constexpr int MK_DENSE_MAX_NV = 64;

// One task of the dense path; frames are resolved to (body, local offset).
struct MkDenseTask {
  int32_t type;            // MK_TASK_*
  int32_t body, root_body;
  int32_t target_off;      // offset of this task's per-problem target
  double lpos[3], lquat[4];
  double rpos[3], rquat[4];
  double cost[6];          // frame tasks / com (first 3)
  double gain, lm;
  const double* pcost;     // posture: [nv] device
  const double* ptarget;   // posture: [nq] device
  const int32_t* bodies;   // com: bodies of the subtree (device)
  int32_t nbodies;
  int32_t k;               // error dimension: 6, nv or 3
};

struct MkDenseLimit {
  int32_t type;            // MK_LIMIT_*
  int32_t n;               // velocity: dofs; collision: pairs; configuration: limited joints
  int32_t row0;            // first row in G
  int32_t pad;
  double gain, min_dist;   // configuration
  const int32_t* index;    // configuration: joint ids; velocity: dof ids; collision: 2n geom ids
  const double* vmax;
  double d_min, d_detect, relax;
};

// Box bounds of one dof, merged over the ConfigurationLimit / VelocityLimit objects of the
// problem when that is possible (at most one configuration limit): what lane i needs.
struct MkDenseDofBound {
  double lower, upper, gain;   // configuration limit (range already shrunk by min_distance_from_limits)
  double vmax;                 // velocity limit, +inf when absent
  int32_t qadr, has_cfg;
};

// One collision pair with everything the distance / Jacobian code reads, in one record
// (geoms ordered by type; `swap` tells that the stored pair order is the other one).
struct MkDensePair {
  int32_t t1, t2, b1, b2, swap, row;
  uint64_t mask1, mask2;       // ancestor-dof masks of the two bodies
  double size1[3], size2[3], pos1[3], pos2[3], quat1[4], quat2[4];
  double gain, d_min, d_detect, relax;
  // bounding spheres (centre in the geom frame) for the cheap "far apart" test, and what the
  // support-mapping route needs (convex != 0: no closed form for this pair)
  double bc1[3], bc2[3], rb1, rb2;
  int32_t convex, v1adr, v1num, v2adr, v2num, pad;
  double mean1[3], mean2[3], ext1, ext2;
};

struct MkDenseProblem {
  int32_t ntask, nlimit, n_rows, target_stride;
  int32_t n_general, npairs;   // rows that are not +-e_i (collision pairs)
  const MkDenseDofBound* dof_bound;   // [nv], or null when the limits do not merge
  const MkDensePair* pairs;           // [npairs], in row order
  MkDenseTask task[MK_DENSE_MAX_TASKS];
  MkDenseLimit limit[MK_DENSE_MAX_LIMITS];
};

// State bit of the iterated loop on the general path, kept in the problem's entry of the
// `iters` array while the loop runs: the problem has left the loop.
constexpr int32_t MK_ITER_DONE_BIT = 1 << 30;
// (skip_flags: problems whose entry has MK_ITER_DONE_BIT set are left alone)
hipError_t mk_generic_integrate(const MkDevModel& m, double* q, const double* v, double dt, int B, hipStream_t s,
                                const int32_t* skip_flags = nullptr);
// One step of the loop's bookkeeping: a live problem whose frame-task error meets the
// thresholds (from the second pass on), or that has used up max_iters, gets the done bit;
// any other live problem is about to be solved and its count becomes it + 1.
hipError_t mk_generic_iterate_mark(int32_t* iters, const double* err, int B, int it, int max_iters, double pos_tol,
                                   double ori_tol, bool has_frame, hipStream_t s);
// Status entries that still carry MK_STATUS_PENDING (mk_device.h) become status 1.
hipError_t mk_generic_resolve_pending(int32_t* status, int B, hipStream_t s);
// Clears the done bits: `iters` then holds the solves executed.
hipError_t mk_generic_iterate_finish(int32_t* iters, int B, hipStream_t s);
// Start of one outer step of a waypoint rollout run as passes (mk_move_to_passes_batch) and the
// bookkeeping after the last one; see the kernels.
hipError_t mk_generic_move_to_begin(const MkDevModel& m, const double* q, int B, int body, const double* lpos,
                                    const double* lquat, const double* goals, double pos_thr, double pos_step,
                                    double rot_thr, double* targets, int target_stride, int target_off, int32_t* iters,
                                    int32_t* steps, int32_t* reached, int32_t* solves, bool first, hipStream_t s);
hipError_t mk_generic_move_to_end(int32_t* iters, int32_t* solves, int B, hipStream_t s);
// Pose (d_pose != null) and/or body-frame Jacobian (d_jac != null) of a frame given
// by its body and local offset.
hipError_t mk_generic_frame(const MkDevModel& m, const double* q, int B, int body, const double* lpos,
                            const double* lquat, double* d_pose, double* d_jac, hipStream_t s);
hipError_t mk_generic_check_limits(const MkDevModel& m, const double* q, double tol, int B, int32_t* first,
                                   hipStream_t s);

// Dense path (mk_dense.hip).  Bytes of LDS one problem needs, or 0 if it cannot run
// (n_general = rows other than +-e_i bounds).
size_t mk_dense_lds_bytes(int nv, int n_general, int nq, int target_stride, int nbody, int njnt, int npath);
// Builds the QP of every problem and, depending on the outputs given, writes it out
// (P/c/G/h), solves it (v/status) and/or reports one task's error and Jacobian.
hipError_t mk_dense_launch(const MkDevModel& m, const MkDenseProblem* d_problem, int n_rows, int n_general,
                           const double* q, const double* targets, int target_stride, double dt, double damping, int B,
                           double* P, double* c, double* G, double* h, double* v, int32_t* status, int eval_task,
                           double* err, double* jac, hipStream_t s, const int32_t* skip_flags = nullptr,
                           bool accumulate_status = false, bool only_pending = false);
// Solves caller-provided dense QPs.
hipError_t mk_dense_qp_launch(const double* P, const double* c, const double* G, const double* h, int n, int m, int B,
                              double* x, int32_t* status, hipStream_t s);
