/* Plain C consumer of include/mink_amd.h: no Python, no torch -- device memory from the HIP
 * runtime only.  Builds a 4-dof arm (slide, three hinges, the last with an off-axis tool
 * site) by hand, and checks on a batch of 200 configurations that
 *   1. the fused path (mk_solve_ik_batch) and the explicit route through the dense
 *      (P, c, G, h) of mk_build_qp_batch + mk_qp_solve_batch agree on the velocity,
 *   2. mk_fk_batch / mk_integrate_batch / mk_solve_ik_iterate_batch run and reduce the
 *      frame-task error,
 *   3. waypoint rollouts outside the fused path (mk_move_to_passes_batch) and the process-wide
 *      options work from C,
 *   4. error returns carry a message.
 * Exit code 0 on success; prints what failed otherwise. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mink_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_MK(x) do { int r_ = (x); if (r_ != MK_OK) { printf("mk error %d (%s) at %s:%d\n", r_, mk_last_error(), __FILE__, __LINE__); return 3; } } while (0)

enum { NB = 5, NJ = 4, NQ = 4, NV = 4, B = 200 };

static double urand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (*s >> 8) / 16777216.0; }

static void* to_device(const void* h, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
  if (h && hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(void) {
  /* world, then a chain of 4 bodies, one joint each */
  const int32_t body_parentid[NB] = {0, 0, 1, 2, 3}, body_jntadr[NB] = {-1, 0, 1, 2, 3}, body_jntnum[NB] = {0, 1, 1, 1, 1};
  const int32_t body_mocapid[NB] = {-1, -1, -1, -1, -1};
  const double body_pos[NB * 3] = {0, 0, 0, 0, 0, 0.1, 0, 0, 0.2, 0.3, 0, 0, 0.25, 0, 0.05};
  const double body_quat[NB * 4] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0.9238795325112867, 0, 0.3826834323650898, 0, 1, 0, 0, 0};
  const double body_mass[NB] = {0, 1, 1, 1, 1}, body_ipos[NB * 3] = {0};
  const int32_t jnt_type[NJ] = {MK_JNT_SLIDE, MK_JNT_HINGE, MK_JNT_HINGE, MK_JNT_HINGE};
  const int32_t jnt_qposadr[NJ] = {0, 1, 2, 3}, jnt_dofadr[NJ] = {0, 1, 2, 3}, jnt_bodyid[NJ] = {1, 2, 3, 4};
  const int32_t jnt_limited[NJ] = {1, 0, 1, 1};
  const double jnt_pos[NJ * 3] = {0, 0, 0, 0, 0, 0, 0, 0.01, 0, 0, 0, 0};
  const double jnt_axis[NJ * 3] = {1, 0, 0, 0, 0, 1, 0, 1, 0, 0.6, 0.8, 0};
  const double jnt_range[NJ * 2] = {-0.5, 0.5, 0, 0, -1.5, 1.5, -2.0, 2.0};
  const double qpos0[NQ] = {0, 0, 0, 0};
  const int32_t dof_parentid[NV] = {-1, 0, 1, 2}, dof_jntid[NV] = {0, 1, 2, 3};
  const int32_t site_bodyid[1] = {4};
  const double site_pos[3] = {0.1, 0.02, 0.03}, site_quat[4] = {0.7071067811865476, 0, 0.7071067811865476, 0};
  mk_model_desc md;
  memset(&md, 0, sizeof(md));
  md.nq = NQ; md.nv = NV; md.nbody = NB; md.njnt = NJ; md.ngeom = 0; md.nsite = 1;
  md.body_parentid = body_parentid; md.body_jntadr = body_jntadr; md.body_jntnum = body_jntnum;
  md.body_mocapid = body_mocapid; md.body_pos = body_pos; md.body_quat = body_quat; md.body_mass = body_mass;
  md.body_ipos = body_ipos; md.jnt_type = jnt_type; md.jnt_qposadr = jnt_qposadr; md.jnt_dofadr = jnt_dofadr;
  md.jnt_bodyid = jnt_bodyid; md.jnt_limited = jnt_limited; md.jnt_pos = jnt_pos; md.jnt_axis = jnt_axis;
  md.jnt_range = jnt_range; md.qpos0 = qpos0; md.dof_parentid = dof_parentid; md.dof_jntid = dof_jntid;
  md.site_bodyid = site_bodyid; md.site_pos = site_pos; md.site_quat = site_quat;

  mk_model* model = NULL;
  CHECK_MK(mk_model_create(&md, &model));

  const double frame_cost[6] = {1, 1, 1, 0.5, 0.5, 0.5}, posture_cost[NV] = {1e-2, 1e-2, 1e-2, 1e-2}, home[NQ] = {0.1, 0.2, -0.3, 0.4};
  mk_task_desc tasks[2];
  memset(tasks, 0, sizeof(tasks));
  tasks[0].type = MK_TASK_FRAME; tasks[0].frame_type = MK_FRAME_SITE; tasks[0].frame_id = 0; tasks[0].ncost = 6;
  tasks[0].gain = 1.0; tasks[0].lm_damping = 0.1; tasks[0].cost = frame_cost;
  tasks[1].type = MK_TASK_POSTURE; tasks[1].ncost = NV; tasks[1].gain = 1.0; tasks[1].cost = posture_cost; tasks[1].target_q = home;
  const int32_t vel_index[2] = {0, 2};
  const double vel_max[2] = {0.4, 1.0};
  mk_limit_desc limits[2];
  memset(limits, 0, sizeof(limits));
  limits[0].type = MK_LIMIT_CONFIGURATION; limits[0].gain = 0.95;
  limits[1].type = MK_LIMIT_VELOCITY; limits[1].n = 2; limits[1].index = vel_index; limits[1].vmax = vel_max;
  mk_problem* prob = NULL;
  CHECK_MK(mk_problem_create(model, tasks, 2, limits, 2, &prob));
  mk_problem_info info;
  CHECK_MK(mk_problem_get_info(prob, &info));
  if (info.nq != NQ || info.nv != NV || info.target_stride != 7 || !info.fused || info.n_rows != 2 * 3 + 2 * 2) {
    printf("unexpected problem info: nq %d nv %d stride %d rows %d fused %d\n", info.nq, info.nv, info.target_stride, info.n_rows, info.fused);
    return 4;
  }

  /* inputs: random configurations inside the limits; targets = FK of other configurations */
  static double q[B * NQ], qs[B * NQ], tg[B * 7], v[B * NV], x[B * NV], err0[B * 6], err1[B * 6];
  static int32_t st[B], st2[B], iters[B];
  unsigned seed = 12345u;
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < NQ; ++j) {
      const double lo = jnt_limited[j] ? jnt_range[2 * j] + 0.05 : -2.0, hi = jnt_limited[j] ? jnt_range[2 * j + 1] - 0.05 : 2.0;
      q[b * NQ + j] = lo + (hi - lo) * urand(&seed);
      qs[b * NQ + j] = lo + (hi - lo) * urand(&seed);
    }
  const double dt = 0.01, damping = 1e-4;
  const int m = info.n_rows;
  double *d_q = to_device(q, sizeof(q)), *d_qs = to_device(qs, sizeof(qs)), *d_tg = to_device(NULL, sizeof(tg));
  double *d_v = to_device(NULL, sizeof(v)), *d_x = to_device(NULL, sizeof(x)), *d_err = to_device(NULL, sizeof(err0));
  double *d_P = to_device(NULL, sizeof(double) * B * NV * NV), *d_c = to_device(NULL, sizeof(double) * B * NV);
  double *d_G = to_device(NULL, sizeof(double) * B * m * NV), *d_h = to_device(NULL, sizeof(double) * B * m);
  int32_t *d_st = to_device(NULL, sizeof(st)), *d_st2 = to_device(NULL, sizeof(st2)), *d_it = to_device(NULL, sizeof(iters));
  if (!d_q || !d_qs || !d_tg || !d_v || !d_x || !d_err || !d_P || !d_c || !d_G || !d_h || !d_st || !d_st2 || !d_it) { printf("allocation failed\n"); return 2; }
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_MK(mk_fk_batch(model, d_qs, B, MK_FRAME_SITE, 0, d_tg, stream));

  /* 1. fused solve against build_qp + generic QP */
  CHECK_MK(mk_solve_ik_batch(prob, d_q, d_tg, dt, damping, B, d_v, d_st, stream));
  CHECK_MK(mk_build_qp_batch(prob, d_q, d_tg, dt, damping, B, d_P, d_c, d_G, d_h, stream));
  CHECK_MK(mk_qp_solve_batch(d_P, d_c, d_G, d_h, NV, m, B, d_x, d_st2, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  CHECK_HIP(hipMemcpy(v, d_v, sizeof(v), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(x, d_x, sizeof(x), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(st, d_st, sizeof(st), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(st2, d_st2, sizeof(st2), hipMemcpyDeviceToHost));
  double worst = 0.0, vmax_seen = 0.0;
  for (int b = 0; b < B; ++b) {
    if (st[b] != 0 || st2[b] != 0) { printf("problem %d: status %d / %d\n", b, st[b], st2[b]); return 5; }
    for (int j = 0; j < NV; ++j) {
      worst = fmax(worst, fabs(v[b * NV + j] - x[b * NV + j] / dt));
      vmax_seen = fmax(vmax_seen, fabs(v[b * NV + j]));
    }
    if (fabs(v[b * NV + 0]) > 0.4 + 1e-9 || fabs(v[b * NV + 2]) > 1.0 + 1e-9) { printf("problem %d breaks a velocity limit\n", b); return 5; }
  }
  if (!(worst <= 1e-8) || !(vmax_seen > 0.1)) { printf("fused and dense routes differ by %.3e (largest |v| %.3f)\n", worst, vmax_seen); return 5; }

  /* 2. the iterated loop brings the frame closer to its target */
  CHECK_MK(mk_solve_ik_iterate_batch(prob, d_q, d_tg, dt, damping, B, 0, -1.0, -1.0, NULL, d_it, d_err, d_st, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  CHECK_HIP(hipMemcpy(err0, d_err, sizeof(err0), hipMemcpyDeviceToHost));
  CHECK_MK(mk_solve_ik_iterate_batch(prob, d_q, d_tg, dt, damping, B, 40, 1e-4, 1e-4, d_v, d_it, d_err, d_st, stream));
  CHECK_MK(mk_integrate_batch(model, d_q, d_v, 0.0, B, stream));   /* dt = 0: a no-op that must succeed */
  CHECK_HIP(hipStreamSynchronize(stream));
  CHECK_HIP(hipMemcpy(err1, d_err, sizeof(err1), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(iters, d_it, sizeof(iters), hipMemcpyDeviceToHost));
  int closer = 0;
  for (int b = 0; b < B; ++b) {
    double n0 = 0, n1 = 0;
    for (int k = 0; k < 6; ++k) { n0 += err0[b * 6 + k] * err0[b * 6 + k]; n1 += err1[b * 6 + k] * err1[b * 6 + k]; }
    closer += n1 < n0;
    if (iters[b] < 1 || iters[b] > 40) { printf("problem %d: %d iterations\n", b, iters[b]); return 6; }
  }
  if (closer < B * 9 / 10) { printf("only %d of %d problems approached their target\n", closer, B); return 6; }

  /* 2b. the same loop for a task set outside the fused path (an extra relative-frame task):
   *     issued as launches of the general kernels, state in d_v / d_iters / d_err */
  {
    const double rel_cost[6] = {0.1, 0.1, 0.1, 0.1, 0.1, 0.1};
    mk_task_desc t3[3];
    memcpy(t3, tasks, sizeof(tasks));
    memset(&t3[2], 0, sizeof(t3[2]));
    t3[2].type = MK_TASK_RELATIVE_FRAME; t3[2].frame_type = MK_FRAME_SITE; t3[2].frame_id = 0;
    t3[2].root_type = MK_FRAME_BODY; t3[2].root_id = 2; t3[2].ncost = 6; t3[2].gain = 1.0; t3[2].cost = rel_cost;
    mk_problem* gen = NULL;
    CHECK_MK(mk_problem_create(model, t3, 3, limits, 2, &gen));
    mk_problem_info gi;
    CHECK_MK(mk_problem_get_info(gen, &gi));
    if (gi.fused || gi.target_stride != 14) { printf("unexpected general problem: fused %d stride %d\n", gi.fused, gi.target_stride); return 8; }
    static double tg2[B * 14];
    CHECK_HIP(hipMemcpy(tg, d_tg, sizeof(tg), hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
      memcpy(tg2 + 14 * b, tg + 7 * b, 7 * sizeof(double));
      const double rel_target[7] = {1, 0, 0, 0, 0.3, 0.0, 0.2};
      memcpy(tg2 + 14 * b + 7, rel_target, sizeof(rel_target));
    }
    double* d_tg2 = to_device(tg2, sizeof(tg2));
    CHECK_HIP(hipMemcpy(d_q, q, sizeof(q), hipMemcpyHostToDevice));
    if (mk_solve_ik_iterate_batch(gen, d_q, d_tg2, dt, damping, B, 5, 1e-4, 1e-4, NULL, d_it, d_err, d_st, stream) != MK_ERR_INVALID) {
      printf("the general loop accepted a missing state array\n");
      return 8;
    }
    CHECK_MK(mk_solve_ik_iterate_batch(gen, d_q, d_tg2, dt, damping, B, 5, -1.0, -1.0, d_v, d_it, d_err, d_st, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(err1, d_err, sizeof(err1), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(iters, d_it, sizeof(iters), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(st, d_st, sizeof(st), hipMemcpyDeviceToHost));
    int closer2 = 0;
    for (int b = 0; b < B; ++b) {
      double n0 = 0, n1 = 0;
      for (int k = 0; k < 6; ++k) { n0 += err0[b * 6 + k] * err0[b * 6 + k]; n1 += err1[b * 6 + k] * err1[b * 6 + k]; }
      closer2 += n1 < n0;
      if (iters[b] != 5 || st[b] != 0) { printf("general loop, problem %d: %d iterations, status %d\n", b, iters[b], st[b]); return 8; }
    }
    if (closer2 < B * 8 / 10) { printf("general loop: only %d of %d problems approached their target\n", closer2, B); return 8; }
    /* waypoint rollouts of the same problem (outside the fused path) as queued passes: goals = the
     * frame-task targets; every rollout takes at least one step and its solves are counted */
    {
      static int32_t steps[B], reached[B], solves[B];
      int32_t *d_steps = to_device(NULL, sizeof(steps)), *d_reached = to_device(NULL, sizeof(reached)),
              *d_solves = to_device(NULL, sizeof(solves));
      const mk_waypoint_config wp = {0.01, 0.08, 0.05, 0.3};
      CHECK_HIP(hipMemcpy(d_q, q, sizeof(q), hipMemcpyHostToDevice));
      if (mk_move_to_passes_batch(gen, 1, d_q, d_tg, d_tg2, &wp, dt, damping, B, 3, 4, 1e-4, 1e-4, d_steps, d_reached,
                                  d_solves, d_st, d_v, d_it, d_err, stream) != MK_ERR_INVALID) {
        printf("a posture task was accepted as the driven frame task\n");
        return 9;
      }
      CHECK_MK(mk_move_to_passes_batch(gen, 0, d_q, d_tg, d_tg2, &wp, dt, damping, B, 3, 4, 1e-4, 1e-4, d_steps, d_reached,
                                       d_solves, d_st, d_v, d_it, d_err, stream));
      CHECK_HIP(hipStreamSynchronize(stream));
      CHECK_HIP(hipMemcpy(steps, d_steps, sizeof(steps), hipMemcpyDeviceToHost));
      CHECK_HIP(hipMemcpy(reached, d_reached, sizeof(reached), hipMemcpyDeviceToHost));
      CHECK_HIP(hipMemcpy(solves, d_solves, sizeof(solves), hipMemcpyDeviceToHost));
      CHECK_HIP(hipMemcpy(st, d_st, sizeof(st), hipMemcpyDeviceToHost));
      for (int b = 0; b < B; ++b) {
        if (steps[b] < 1 || steps[b] > 3 || solves[b] < steps[b] || solves[b] > 4 * steps[b] || st[b] != 0 ||
            (reached[b] != 0 && reached[b] != 1)) {
          printf("rollout %d: %d steps, %d solves, reached %d, status %d\n", b, steps[b], solves[b], reached[b], st[b]);
          return 9;
        }
      }
      (void)hipFree(d_steps); (void)hipFree(d_reached); (void)hipFree(d_solves);
    }
    mk_problem_destroy(gen);
    (void)hipFree(d_tg2);
  }

  /* process-wide options: round trip, unknown names, and the float32 switch refusing what has no such kernels */
  {
    int64_t v0 = 0, v1 = 0;
    CHECK_MK(mk_get_option("coop_max_batch", &v0));
    CHECK_MK(mk_set_option("coop_max_batch", 123));
    CHECK_MK(mk_get_option("coop_max_batch", &v1));
    CHECK_MK(mk_set_option("coop_max_batch", v0));
    if (v0 <= 0 || v1 != 123) { printf("option round trip: %lld -> %lld\n", (long long)v0, (long long)v1); return 10; }
    if (mk_set_option("no_such_option", 1) != MK_ERR_INVALID) { printf("an unknown option was accepted\n"); return 10; }
    CHECK_MK(mk_set_option("f32", 1));
    CHECK_MK(mk_solve_ik_batch(prob, d_q, d_tg, dt, damping, B, d_v, d_st, stream));   /* fused: float32 kernels exist */
    CHECK_MK(mk_set_option("f32", 0));
    CHECK_HIP(hipStreamSynchronize(stream));
  }

  /* 4. errors are reported, not thrown */
  if (mk_solve_ik_batch(prob, d_q, NULL, dt, damping, B, d_v, d_st, stream) != MK_ERR_INVALID || strlen(mk_last_error()) == 0) {
    printf("missing targets were not rejected\n");
    return 7;
  }
  if (mk_fk_batch(model, d_q, B, MK_FRAME_SITE, 3, d_tg, stream) == MK_OK) { printf("bad site id was accepted\n"); return 7; }

  mk_problem_destroy(prob);
  mk_model_destroy(model);
  printf("cabi_check ok: fused vs dense route %.2e rad/s apart, %d of %d problems closer after the loop\n", worst, closer, B);
  return 0;
}
