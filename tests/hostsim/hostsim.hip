// TEST-ONLY host build of the fused IK device functions.
//
// The product library (mink_amd/libmink_amd.so) runs solve_problem<NC>() inside HIP
// kernels only.  This translation unit includes the same sources and calls the very
// same templates lane by lane on the CPU, so the structured QP build / active-set
// logic can be checked against the oracle in the CPU test suite (no GPU needed).
// It is never loaded by the mink_amd package.
#define MK_QP_STATS 1
#define MK_COOP_HOSTSIM 1
#include "coop_sim.h"
#include "../../mink_amd/csrc/mk_api.hip"
#include "../../mink_amd/csrc/mk_fused.hip"
#include "../../mink_amd/csrc/mk_coop.hip"

namespace {

// optional per-(problem, iteration) record of the QP work: phase-1 rounds, phase-2 steps
int32_t* g_trace = nullptr;
int g_trace_iters = 0;

template <int NC>
int run_step(const MkFusedDesc& D, const double* q, const double* targets, int stride, double dt, double damping,
             int B, double* v, int32_t* status) {
  std::vector<double> row(D.nq + 1);
  double hbuf[MK_MAX_CHAIN * (MK_MAX_CHAIN + 1) / 2];
  for (int b = 0; b < B; ++b) {
    memcpy(row.data(), q + (size_t)b * D.nq, D.nq * sizeof(double));
    const double* tg = targets + (size_t)b * stride + D.target_off;
    double qc[NC], x[NC];
    load_chain_q<NC>(&D, row.data(), qc);
    ChainJac<NC> fk;
    chain_jac<NC>(&D, qc, Q4{tg[0], tg[1], tg[2], tg[3]}, v3(tg[4], tg[5], tg[6]), fk);
    uint32_t act = 0u, upper = 0u;
    status[b] = solve_problem<NC, false>(&D, row.data(), qc, fk, dt, 1.0 / dt, damping, x, nullptr, act, upper, false, hbuf, 1,
                                         rest_nonfinite(&D, row.data()));
    for (int k = 0; k < NC; ++k)
      if (D.seg_dof[k] >= 0) row[D.seg_dof[k]] = x[k] * (1.0 / dt);
    memcpy(v + (size_t)b * D.nv, row.data(), D.nv * sizeof(double));
  }
  return 0;
}

template <int NC>
int run_iterate(const MkFusedDesc& D, double* q, const double* targets, int stride, double dt, double damping, int B,
                int max_iters, double pos_tol, double ori_tol, double* v, int32_t* iters_out, double* err,
                int32_t* status_out) {
  std::vector<double> row(D.nq + 1);
  double hbuf[MK_MAX_CHAIN * (MK_MAX_CHAIN + 1) / 2];
  for (int b = 0; b < B; ++b) {
    memcpy(row.data(), q + (size_t)b * D.nq, D.nq * sizeof(double));
    const double* tg = targets + (size_t)b * stride + D.target_off;
    Q4 Qt{tg[0], tg[1], tg[2], tg[3]};
    V3 pt = v3(tg[4], tg[5], tg[6]);
    double qc[NC], x[NC];
    load_chain_q<NC>(&D, row.data(), qc);
    for (int k = 0; k < NC; ++k) x[k] = 0.0;
    double* vrow = v ? v + (size_t)b * D.nv : nullptr;
    Twist e{};
    int status = 0, iters = 0;
    uint32_t act = 0u, upper = 0u;
    const bool input_bad = rest_nonfinite(&D, row.data());
    for (int it = 0; it <= max_iters; ++it) {
      ChainJac<NC> fk;
      chain_jac<NC>(&D, qc, Qt, pt, fk);
      e = fk.e;
      const bool conv = it > 0 && sqrt(dot(e.v, e.v)) <= pos_tol && sqrt(dot(e.w, e.w)) <= ori_tol;
      if (conv || it == max_iters) break;
#if !defined(__HIP_DEVICE_COMPILE__)
      const long r0 = g_qp_stats[1], p0 = g_qp_stats[2], s0 = g_qp_stats[4];
#endif
      int st = solve_problem<NC, true>(&D, row.data(), qc, fk, dt, 1.0 / dt, damping, x, vrow, act, upper, it > 0, hbuf, 1,
                                       input_bad);
#if !defined(__HIP_DEVICE_COMPILE__)
      if (g_trace && it < g_trace_iters) {
        int32_t* tr = g_trace + ((size_t)b * g_trace_iters + it) * 3;
        tr[0] = (int32_t)(g_qp_stats[1] - r0);
        tr[1] = (int32_t)(g_qp_stats[2] - p0);
        tr[2] = (int32_t)(g_qp_stats[4] - s0);
      }
#endif
      status = std::max(status, st);
      iters = it + 1;
      for (int k = 0; k < NC; ++k) {
        double vk = x[k] * (1.0 / dt);
        x[k] = vk;
        qc[k] = qc[k] + dt * vk;
        if (D.seg_dof[k] >= 0) row[D.seg_qadr[k]] = qc[k];
      }
    }
    if (vrow)
      for (int k = 0; k < NC; ++k)
        if (D.seg_dof[k] >= 0) vrow[D.seg_dof[k]] = x[k];
    status_out[b] = status;
    if (iters_out) iters_out[b] = iters;
    if (err) {
      double* eo = err + (size_t)b * 6;
      eo[0] = e.v.x; eo[1] = e.v.y; eo[2] = e.v.z; eo[3] = e.w.x; eo[4] = e.w.y; eo[5] = e.w.z;
    }
    memcpy(q + (size_t)b * D.nq, row.data(), D.nq * sizeof(double));
  }
  return 0;
}

int build(const mk_model_desc* md, const mk_task_desc* tasks, int ntask, const mk_limit_desc* limits, int nlimit,
          MkFusedDesc& D, int* nk, int* stride) {
  HostModel h;
  int rc = host_model_from_desc(md, h);
  if (rc != MK_OK) return rc;
  std::string why;
  if (!build_fused(h, tasks, ntask, limits, nlimit, D, nk, why)) return fail(MK_ERR_UNSUPPORTED, "%s", why.c_str());
  *stride = 0;
  for (int t = 0; t < ntask; ++t)
    *stride += (tasks[t].type == MK_TASK_FRAME || tasks[t].type == MK_TASK_RELATIVE_FRAME) ? 7
               : tasks[t].type == MK_TASK_COM ? 3 : 0;
  return MK_OK;
}

}  // namespace

#define HS_DISPATCH(fn, ...)            \
  switch (nk) {                         \
    case 4: return fn<4>(__VA_ARGS__);  \
    case 6: return fn<6>(__VA_ARGS__);  \
    case 7: return fn<7>(__VA_ARGS__);  \
    case 8: return fn<8>(__VA_ARGS__);  \
    case 10: return fn<10>(__VA_ARGS__); \
    case 12: return fn<12>(__VA_ARGS__); \
    default: return MK_ERR_UNSUPPORTED; \
  }

extern "C" int hostsim_solve_ik(const mk_model_desc* md, const mk_task_desc* tasks, int ntask,
                                const mk_limit_desc* limits, int nlimit, const double* q, const double* targets,
                                double dt, double damping, int B, double* v, int32_t* status) {
  static MkFusedDesc D;
  int nk, stride;
  int rc = build(md, tasks, ntask, limits, nlimit, D, &nk, &stride);
  if (rc != MK_OK) return rc;
  HS_DISPATCH(run_step, D, q, targets, stride, dt, damping, B, v, status)
}

extern "C" int hostsim_iterate(const mk_model_desc* md, const mk_task_desc* tasks, int ntask,
                               const mk_limit_desc* limits, int nlimit, double* q, const double* targets, double dt,
                               double damping, int B, int max_iters, double pos_tol, double ori_tol, double* v,
                               int32_t* iters, double* err, int32_t* status) {
  static MkFusedDesc D;
  int nk, stride;
  int rc = build(md, tasks, ntask, limits, nlimit, D, &nk, &stride);
  if (rc != MK_OK) return rc;
  HS_DISPATCH(run_iterate, D, q, targets, stride, dt, damping, B, max_iters, pos_tol, ori_tol, v, iters, err, status)
}

// The lane form of the staged solve's middle stage on every problem: rows of the collision pairs
// in range + the QP with them.  good[b] = 0 where the lane form hands the problem on.
template <int NC>
int run_rows(const MkFusedDesc& D, const MkFusedExt& X, const double* q, const double* targets, int stride, double dt,
             double damping, int B, double* v, int32_t* good, int32_t* nrows) {
  double hbuf[MK_MAX_CHAIN * (MK_MAX_CHAIN + 1) / 2];
  std::vector<double> gbuf((size_t)MK_LANE_ROWS * D.nv);
  double cbuf[MK_LANE_ROWS * NC] = {};   // (written only by the forms that keep the chain part apart)
  for (int b = 0; b < B; ++b) {
    good[b] = rows_problem<NC>(&D, &X, q + (size_t)b * D.nq, targets + (size_t)b * stride, dt, 1.0 / dt, damping,
                               v + (size_t)b * D.nv, hbuf, 1, cbuf, 1, gbuf.data(), 1) ? 1 : 0;
    int n = 0;
    for (int k = 0; k < MK_LANE_ROWS; ++k) {
      bool any = false;
      for (int i = 0; i < D.nv; ++i) any |= gbuf[(size_t)k * D.nv + i] != 0.0;
      for (int i = 0; i < NC; ++i) any |= cbuf[k * NC + i] != 0.0;
      n += any;
    }
    if (nrows) nrows[b] = n;
  }
  return 0;
}

extern "C" int hostsim_solve_rows(const mk_model_desc* md, const mk_task_desc* tasks, int ntask,
                                  const mk_limit_desc* limits, int nlimit, const double* q, const double* targets,
                                  double dt, double damping, int B, double* v, int32_t* good, int32_t* nrows) {
  static MkFusedDesc D;
  static MkFusedExt X;
  HostModel h;
  int rc = host_model_from_desc(md, h);
  if (rc != MK_OK) return rc;
  std::string why;
  int nk;
  bool needs_ext = false;
  if (!build_fused(h, tasks, ntask, limits, nlimit, D, &nk, why, &X, &needs_ext)) return fail(MK_ERR_UNSUPPORTED, "%s", why.c_str());
  int stride = 0;
  for (int t = 0; t < ntask; ++t)
    stride += (tasks[t].type == MK_TASK_FRAME || tasks[t].type == MK_TASK_RELATIVE_FRAME) ? 7
              : tasks[t].type == MK_TASK_COM ? 3 : 0;
  HS_DISPATCH(run_rows, D, X, q, targets, stride, dt, damping, B, v, good, nrows)
}

// box_box_lane on B box pairs: c [B,3], quaternion [B,4] (wxyz), half sizes [B,3] per box;
// out [B,8] = (in range, dist, p1, p2).
extern "C" void hostsim_box_box(const double* c1, const double* q1, const double* h1, const double* c2, const double* q2,
                                const double* h2, int B, double distmax, double* out) {
  for (int b = 0; b < B; ++b) {
    const double *a = q1 + 4 * b, *e = q2 + 4 * b;
    Contact ct;
    ct.dist = 0.0;
    ct.p1 = ct.p2 = v3(0, 0, 0);
    const bool in = box_box_lane(v3(c1[3 * b], c1[3 * b + 1], c1[3 * b + 2]), q2m(Q4{a[0], a[1], a[2], a[3]}),
                                 v3(h1[3 * b], h1[3 * b + 1], h1[3 * b + 2]), v3(c2[3 * b], c2[3 * b + 1], c2[3 * b + 2]),
                                 q2m(Q4{e[0], e[1], e[2], e[3]}), v3(h2[3 * b], h2[3 * b + 1], h2[3 * b + 2]), distmax, ct);
    double* o = out + 8 * b;
    o[0] = in ? 1.0 : 0.0; o[1] = ct.dist;
    o[2] = ct.p1.x; o[3] = ct.p1.y; o[4] = ct.p1.z; o[5] = ct.p2.x; o[6] = ct.p2.y; o[7] = ct.p2.z;
  }
}

// geom_distance_lane on B pairs of primitives of types (t1 <= t2): centre [B,3], quaternion [B,4]
// (wxyz), MuJoCo size [B,3] per geom; out [B,8] = (in range, dist, p1, p2).
extern "C" void hostsim_geom_distance(int t1, const double* c1, const double* q1, const double* s1, int t2,
                                      const double* c2, const double* q2, const double* s2, int B, double distmax,
                                      double* out) {
  for (int b = 0; b < B; ++b) {
    const double *a = q1 + 4 * b, *e = q2 + 4 * b;
    Contact ct;
    ct.dist = 0.0;
    ct.p1 = ct.p2 = v3(0, 0, 0);
    const bool in = geom_distance_lane(t1, v3(c1[3 * b], c1[3 * b + 1], c1[3 * b + 2]), q2m(Q4{a[0], a[1], a[2], a[3]}),
                                       v3(s1[3 * b], s1[3 * b + 1], s1[3 * b + 2]), t2,
                                       v3(c2[3 * b], c2[3 * b + 1], c2[3 * b + 2]), q2m(Q4{e[0], e[1], e[2], e[3]}),
                                       v3(s2[3 * b], s2[3 * b + 1], s2[3 * b + 2]), distmax, ct);
    double* o = out + 8 * b;
    o[0] = in ? 1.0 : 0.0; o[1] = ct.dist;
    o[2] = ct.p1.x; o[3] = ct.p1.y; o[4] = ct.p1.z; o[5] = ct.p2.x; o[6] = ct.p2.y; o[7] = ct.p2.z;
  }
}

// The device waypoint generator on B (current pose, goal pose) pairs (wxyz_xyz each):
// commanded poses [B, 7] and reached flags [B].
extern "C" void hostsim_waypoint_step(const double* cur, const double* goal, int B, double pos_thr, double pos_step,
                                      double rot_thr, double* cmd, int32_t* reached) {
  for (int b = 0; b < B; ++b) {
    const double *c = cur + 7 * b, *g = goal + 7 * b;
    Q4 Qt;
    V3 pt;
    reached[b] = waypoint_step(Q4{c[0], c[1], c[2], c[3]}, v3(c[4], c[5], c[6]), Q4{g[0], g[1], g[2], g[3]},
                               v3(g[4], g[5], g[6]), pos_thr, pos_step, rot_thr, Qt, pt) ? 1 : 0;
    double* o = cmd + 7 * b;
    o[0] = Qt.w; o[1] = Qt.x; o[2] = Qt.y; o[3] = Qt.z; o[4] = pt.x; o[5] = pt.y; o[6] = pt.z;
  }
}

// buf [B, iters, 3] receives (phase-1 rounds, phase-2 steps, single-change steps) of every solve of the next
// hostsim_iterate calls; null switches the record off
extern "C" void hostsim_set_trace(int32_t* buf, int iters) {
  g_trace = buf;
  g_trace_iters = iters;
}

// solves, phase-1 rounds, phase-2 steps, factorizations since the last reset
extern "C" void hostsim_qp_stats(long* out, int reset) {
#if !defined(__HIP_DEVICE_COMPILE__)
  for (int i = 0; i < 4; ++i) { out[i] = g_qp_stats[i]; if (reset) g_qp_stats[i] = 0; }
#endif
}

// ---- the cooperative (16 lanes per problem) form, one row of fibers per problem ----
template <int NC>
int run_coop_step(const MkFusedDesc& D, const double* q, const double* targets, int stride, double dt, double damping,
                  int B, double* v, int32_t* status, int refine) {
  for (long g = 0; g < B; ++g)
    cosim::run_row([&] {
      (void)refine;
      coop_step_body<NC>(&D, q, targets, stride, dt, damping, B, v, status, g);
    });
  return 0;
}

template <int NC>
int run_coop_iterate(const MkFusedDesc& D, double* q, const double* targets, int stride, double dt, double damping, int B,
                     int max_iters, double pos_tol, double ori_tol, double* v, int32_t* iters, double* err,
                     int32_t* status, int refine) {
  for (long g = 0; g < B; ++g)
    cosim::run_row([&] {
      (void)refine;
      coop_iterate_body<NC>(&D, q, targets, stride, dt, damping, max_iters, pos_tol, ori_tol, v, iters, err, status, g,
                            true, 0, 0);
    });
  return 0;
}

extern "C" int hostsim_coop_solve_ik(const mk_model_desc* md, const mk_task_desc* tasks, int ntask,
                                     const mk_limit_desc* limits, int nlimit, const double* q, const double* targets,
                                     double dt, double damping, int B, double* v, int32_t* status, int refine) {
  static MkFusedDesc D;
  int nk, stride;
  int rc = build(md, tasks, ntask, limits, nlimit, D, &nk, &stride);
  if (rc != MK_OK) return rc;
  HS_DISPATCH(run_coop_step, D, q, targets, stride, dt, damping, B, v, status, refine)
}

extern "C" int hostsim_coop_iterate(const mk_model_desc* md, const mk_task_desc* tasks, int ntask,
                                    const mk_limit_desc* limits, int nlimit, double* q, const double* targets, double dt,
                                    double damping, int B, int max_iters, double pos_tol, double ori_tol, double* v,
                                    int32_t* iters, double* err, int32_t* status, int refine) {
  static MkFusedDesc D;
  int nk, stride;
  int rc = build(md, tasks, ntask, limits, nlimit, D, &nk, &stride);
  if (rc != MK_OK) return rc;
  HS_DISPATCH(run_coop_iterate, D, q, targets, stride, dt, damping, B, max_iters, pos_tol, ori_tol, v, iters, err, status,
              refine)
}
