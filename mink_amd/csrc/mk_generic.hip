// Generic utility kernels (see mk_generic.h).  They restate, per problem:
//   mj_integratePos                     (mink/configuration.py:225,235)
//   mj_kinematics + mj_jac for a frame   (mink/configuration.py:112-185)
//   Configuration.check_limits          (mink/configuration.py:77-110)
#include "mk_generic.h"

#include "../../include/mink_amd.h"
#include "mk_device.h"
#include "mk_tree.h"

namespace {

// q <- q (+) dt v, one thread per (problem, joint).
__global__ void integrate_kernel(MkDevModel m, double* __restrict__ q, const double* __restrict__ v, double dt,
                                 long total, const int32_t* __restrict__ skip_flags) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long b = idx / m.njnt;
  if (skip_flags && (skip_flags[b] & MK_ITER_DONE_BIT)) return;
  const int j = (int)(idx - b * m.njnt);
  double* qq = q + b * m.nq + m.jnt_qposadr[j];
  const double* vv = v + b * m.nv + m.jnt_dofadr[j];
  const int t = m.jnt_type[j];
  if (t == MK_JNT_HINGE || t == MK_JNT_SLIDE) {
    qq[0] += dt * vv[0];
    return;
  }
  if (t == MK_JNT_FREE) {
    qq[0] += dt * vv[0];
    qq[1] += dt * vv[1];
    qq[2] += dt * vv[2];
    qq += 3;
    vv += 3;
  }
  // mju_quatIntegrate: quat <- normalize(quat) * axisangle(w / |w|, dt |w|)
  V3 w = v3(vv[0], vv[1], vv[2]);
  double n = sqrt(dot(w, w));
  V3 ax = n < MK_MINVAL ? v3(1, 0, 0) : (1.0 / n) * w;
  if (n < MK_MINVAL) n = 0.0;
  double ang = dt * n, sn, cs;
  sincos(0.5 * ang, &sn, &cs);
  Q4 rot = ang == 0.0 ? Q4{1, 0, 0, 0} : Q4{cs, sn * ax.x, sn * ax.y, sn * ax.z};
  Q4 r = qmul(qnormalize(Q4{qq[0], qq[1], qq[2], qq[3]}), rot);
  qq[0] = r.w; qq[1] = r.x; qq[2] = r.y; qq[3] = r.z;
}

// Emits body-frame Jacobian columns (rows linear; angular) of the frame at (ps, Rs).
struct FrameJacEmit {
  V3 ps;
  M3 Rs;
  double* jac;
  int nv;
  MK_DEV void operator()(int dof, int kind, V3 axis, V3 anchor) {
    V3 lin = kind == MK_COL_ROT ? cross(axis, ps - anchor) : axis;
    V3 ang = kind == MK_COL_ROT ? axis : v3(0, 0, 0);
    V3 bl = mulT(Rs, lin), ba = mulT(Rs, ang);
    jac[0 * nv + dof] = bl.x; jac[1 * nv + dof] = bl.y; jac[2 * nv + dof] = bl.z;
    jac[3 * nv + dof] = ba.x; jac[4 * nv + dof] = ba.y; jac[5 * nv + dof] = ba.z;
  }
};

__global__ void frame_kernel(MkDevModel m, const double* __restrict__ q, int B, int body, V3 lpos, Q4 lquat,
                             double* __restrict__ pose, double* __restrict__ jac) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* qb = q + b * m.nq;
  const int adr = m.path_adr[body], len = m.path_len[body];
  BodyState st = body_state_identity();
  NoEmit none;
  for (int i = 0; i < len; ++i) advance_body<false>(m, qb, m.path_data[adr + i], st, none);
  V3 ps = st.p + mul(st.R, lpos);
  Q4 Qs = qmul(st.Q, lquat);
  if (pose) {
    // The reference reads the frame's rotation matrix and converts it with mju_mat2Quat
    // (mink/configuration.py:157-185, mink/lie/so3.py:80-84), which fixes the sign of the quaternion:
    // the component its branch takes the square root for is positive.
    const M3 Rm = q2m(Qs);
    const double m0 = Rm.m[0], m4 = Rm.m[4], m8 = Rm.m[8];
    const double lead = (m0 + m4 + m8 > 0.0) ? Qs.w : (m0 > m4 && m0 > m8) ? Qs.x : (m4 > m8) ? Qs.y : Qs.z;
    if (lead < 0.0) Qs = Q4{-Qs.w, -Qs.x, -Qs.y, -Qs.z};
    double* o = pose + b * 7;
    o[0] = Qs.w; o[1] = Qs.x; o[2] = Qs.y; o[3] = Qs.z; o[4] = ps.x; o[5] = ps.y; o[6] = ps.z;
  }
  if (jac) {
    double* jb = jac + b * 6 * m.nv;
    for (int i = 0; i < 6 * m.nv; ++i) jb[i] = 0.0;
    FrameJacEmit em{ps, q2m(Qs), jb, m.nv};
    BodyState s2 = body_state_identity();
    for (int i = 0; i < len; ++i) advance_body<true>(m, qb, m.path_data[adr + i], s2, em);
  }
}

__global__ void check_limits_kernel(MkDevModel m, const double* __restrict__ q, double tol, int B,
                                    int32_t* __restrict__ first) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* qb = q + b * m.nq;
  int bad = -1;
  for (int j = 0; j < m.njnt; ++j) {
    if (m.jnt_type[j] == MK_JNT_FREE || !m.jnt_limited[j]) continue;
    const double val = qb[m.jnt_qposadr[j]];
    if ((val < m.jnt_range[2 * j] - tol || val > m.jnt_range[2 * j + 1] + tol) && bad < 0) bad = j;
  }
  first[b] = bad;
}

__global__ void iterate_mark_kernel(int32_t* __restrict__ iters, const double* __restrict__ err, int B, int it,
                                    int max_iters, double pos_tol, double ori_tol, int has_frame) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int32_t cur = iters[b];
  if (cur & MK_ITER_DONE_BIT) return;
  bool conv = false;
  if (has_frame && it > 0) {
    const double* e = err + (long)b * 6;
    conv = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) <= pos_tol &&
           sqrt(e[3] * e[3] + e[4] * e[4] + e[5] * e[5]) <= ori_tol;
  }
  iters[b] = (conv || it == max_iters) ? (cur | MK_ITER_DONE_BIT) : it + 1;
}

__global__ void iterate_finish_kernel(int32_t* __restrict__ iters, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) iters[b] &= ~MK_ITER_DONE_BIT;
}

// A status entry that still carries the pending bit becomes status 1 (not solved).
__global__ void resolve_pending_kernel(int32_t* __restrict__ status, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B && (status[b] & MK_STATUS_PENDING) != 0) status[b] = 1;
}

// One outer step of the waypoint rollout begins (envs/mj_env.py:212-230): the solves of the step
// before are added to the problem's count; a problem whose generator has reported its goal reached
// is out of the rollout (done bit: every launch of the step leaves it alone); any other gets its
// next commanded pose (LinearWaypointReach.step, envs/robot_utils.py:131-160) written into the driven
// frame task's slot of its target row, one more step on its counter and, when the generator says so,
// the reached flag -- this step's IK still runs for it, like the reference's `break` after `step`.
__global__ void move_to_begin_kernel(MkDevModel m, const double* __restrict__ q, int B, int body, V3 lpos, Q4 lquat,
                                     const double* __restrict__ goals, double pos_thr, double pos_step, double rot_thr,
                                     double* __restrict__ targets, int target_stride, int target_off,
                                     int32_t* __restrict__ iters, int32_t* __restrict__ steps,
                                     int32_t* __restrict__ reached, int32_t* __restrict__ solves, int first) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (!first) solves[b] += iters[b] & ~MK_ITER_DONE_BIT;
  if (reached[b]) {
    iters[b] = MK_ITER_DONE_BIT;
    return;
  }
  const double* qb = q + b * m.nq;
  const int adr = m.path_adr[body], len = m.path_len[body];
  BodyState st = body_state_identity();
  NoEmit none;
  for (int i = 0; i < len; ++i) advance_body<false>(m, qb, m.path_data[adr + i], st, none);
  const V3 tc = st.p + mul(st.R, lpos);
  const Q4 Qc = qmul(st.Q, lquat);
  const double* g = goals + b * 7;
  Q4 Qt;
  V3 pt;
  const bool now = waypoint_step(Qc, tc, Q4{g[0], g[1], g[2], g[3]}, v3(g[4], g[5], g[6]), pos_thr, pos_step, rot_thr, Qt, pt);
  double* t = targets + b * target_stride + target_off;
  t[0] = Qt.w; t[1] = Qt.x; t[2] = Qt.y; t[3] = Qt.z; t[4] = pt.x; t[5] = pt.y; t[6] = pt.z;
  steps[b] += 1;
  if (now) reached[b] = 1;
  iters[b] = 0;
}

// After the last outer step: its solves join the count, and `iters` is left without state bits.
__global__ void move_to_end_kernel(int32_t* __restrict__ iters, int32_t* __restrict__ solves, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int32_t n = iters[b] & ~MK_ITER_DONE_BIT;
  solves[b] += n;
  iters[b] = n;
}

}  // namespace

hipError_t mk_generic_move_to_begin(const MkDevModel& m, const double* q, int B, int body, const double* lpos,
                                    const double* lquat, const double* goals, double pos_thr, double pos_step,
                                    double rot_thr, double* targets, int target_stride, int target_off, int32_t* iters,
                                    int32_t* steps, int32_t* reached, int32_t* solves, bool first, hipStream_t s) {
  hipLaunchKernelGGL(move_to_begin_kernel, dim3((B + 63) / 64), dim3(64), 0, s, m, q, B, body,
                     V3{lpos[0], lpos[1], lpos[2]}, Q4{lquat[0], lquat[1], lquat[2], lquat[3]}, goals, pos_thr, pos_step,
                     rot_thr, targets, target_stride, target_off, iters, steps, reached, solves, first ? 1 : 0);
  return hipGetLastError();
}

hipError_t mk_generic_move_to_end(int32_t* iters, int32_t* solves, int B, hipStream_t s) {
  hipLaunchKernelGGL(move_to_end_kernel, dim3((B + 255) / 256), dim3(256), 0, s, iters, solves, B);
  return hipGetLastError();
}

hipError_t mk_generic_resolve_pending(int32_t* status, int B, hipStream_t s) {
  hipLaunchKernelGGL(resolve_pending_kernel, dim3((B + 255) / 256), dim3(256), 0, s, status, B);
  return hipGetLastError();
}

hipError_t mk_generic_integrate(const MkDevModel& m, double* q, const double* v, double dt, int B, hipStream_t s,
                                const int32_t* skip_flags) {
  const long total = (long)B * m.njnt;
  const int threads = 256;
  hipLaunchKernelGGL(integrate_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0, s, m, q, v,
                     dt, total, skip_flags);
  return hipGetLastError();
}

hipError_t mk_generic_iterate_mark(int32_t* iters, const double* err, int B, int it, int max_iters, double pos_tol,
                                   double ori_tol, bool has_frame, hipStream_t s) {
  hipLaunchKernelGGL(iterate_mark_kernel, dim3((B + 255) / 256), dim3(256), 0, s, iters, err, B, it, max_iters, pos_tol,
                     ori_tol, has_frame ? 1 : 0);
  return hipGetLastError();
}

hipError_t mk_generic_iterate_finish(int32_t* iters, int B, hipStream_t s) {
  hipLaunchKernelGGL(iterate_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, s, iters, B);
  return hipGetLastError();
}

hipError_t mk_generic_frame(const MkDevModel& m, const double* q, int B, int body, const double* lpos,
                            const double* lquat, double* d_pose, double* d_jac, hipStream_t s) {
  const int threads = 64;
  hipLaunchKernelGGL(frame_kernel, dim3((B + threads - 1) / threads), dim3(threads), 0, s, m, q, B, body,
                     V3{lpos[0], lpos[1], lpos[2]}, Q4{lquat[0], lquat[1], lquat[2], lquat[3]}, d_pose, d_jac);
  return hipGetLastError();
}

hipError_t mk_generic_check_limits(const MkDevModel& m, const double* q, double tol, int B, int32_t* first,
                                   hipStream_t s) {
  const int threads = 256;
  hipLaunchKernelGGL(check_limits_kernel, dim3((B + threads - 1) / threads), dim3(threads), 0, s, m, q, tol, B, first);
  return hipGetLastError();
}
