// Generic (any tree, any frame) utility kernels: integrate, frame pose / Jacobian,
// limit check.  One problem per thread; the model tables are read through uniform
// (scalar) loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct MkDevModel {
  int32_t nq, nv, nbody, njnt;
  const int32_t *body_parentid, *body_jntadr, *body_jntnum, *body_mocapid;
  const double *body_pos, *body_quat;
  const int32_t *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_range, *qpos0;
  const int32_t *path_adr, *path_len, *path_data;  // bodies from the root to each body
};

// Placeholder for the dense-path problem description (device pointers).
struct MkGenericProblem {
  int32_t ntask, nlimit;
};

hipError_t mk_generic_integrate(const MkDevModel& m, double* q, const double* v, double dt, int B, hipStream_t s);
// Pose (d_pose != null) and/or body-frame Jacobian (d_jac != null) of a frame given
// by its body and local offset.
hipError_t mk_generic_frame(const MkDevModel& m, const double* q, int B, int body, const double* lpos,
                            const double* lquat, double* d_pose, double* d_jac, hipStream_t s);
hipError_t mk_generic_check_limits(const MkDevModel& m, const double* q, double tol, int B, int32_t* first,
                                   hipStream_t s);
