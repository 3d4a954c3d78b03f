// Dense (general) path: one IK problem per wavefront.
//
// Used for task / limit sets outside the structured form of the fused kernels
// (RelativeFrameTask, ComTask, several frame tasks, CollisionAvoidanceLimit, free or
// ball joints on a task chain).  Lane i of the wave owns dof i: column i of every task
// Jacobian, row i of the Hessian.  All lanes walk the same kinematic path, so control
// flow is wave-uniform and the stacked QP
//     P = damping I + sum_t (Jw'Jw + mu I),  c = -sum_t ew'Jw      (mink/solve_ik.py:13-22)
//     G, h = vstack / hstack of the limits                          (mink/solve_ik.py:25-40)
// is assembled in LDS, then either written out (mink.build_ik), or solved in place by
// the Goldfarb-Idnani dual active-set method (the algorithm of the reference's QP
// backend, quadprog) with wave-level reductions, giving v = dq / dt.
#include "mk_generic.h"

#include "../../include/mink_amd.h"
#include "mk_device.h"
#include "mk_tree.h"

namespace {

#define MK_GPU __device__ __forceinline__

constexpr int GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_BOX = 6;

MK_GPU double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
MK_GPU double bcast(double v, int lane) { return __shfl(v, lane); }

// ------------------------------------------------------------- kinematics ----
// Captures, for the lane's own dof, the joint axis / anchor met on a path walk.
struct LaneCapture {
  int lane;
  int on_path;
  int kind;
  V3 axis, anchor;
  MK_GPU void operator()(int dof, int k, V3 ax, V3 an) {
    if (dof == lane) { on_path = 1; kind = k; axis = ax; anchor = an; }
  }
  // world-aligned Jacobian column of a point rigidly attached to the walked body
  MK_GPU V3 jp(V3 point) const {
    if (!on_path) return v3(0, 0, 0);
    return kind == MK_COL_ROT ? cross(axis, point - anchor) : axis;
  }
  MK_GPU V3 jr() const { return (on_path && kind == MK_COL_ROT) ? axis : v3(0, 0, 0); }
};

MK_GPU BodyState walk(const MkDevModel& m, const double* qb, int body, LaneCapture& cap) {
  cap.on_path = 0;
  cap.kind = MK_COL_LIN;
  cap.axis = v3(0, 0, 0);
  cap.anchor = v3(0, 0, 0);
  BodyState st = body_state_identity();
  const int adr = m.path_adr[body], len = m.path_len[body];
  for (int i = 0; i < len; ++i) advance_body<true>(m, qb, m.path_data[adr + i], st, cap);
  return st;
}

struct Frame {
  V3 ps;
  Q4 Qs;
  M3 Rs;
  V3 bl, ba;   // this lane's body-frame Jacobian column (linear; angular)
};

MK_GPU Frame eval_frame(const MkDevModel& m, const double* qb, int body, const double* lpos, const double* lquat,
                        int lane) {
  LaneCapture cap;
  cap.lane = lane;
  BodyState st = walk(m, qb, body, cap);
  Frame f;
  f.ps = st.p + mul(st.R, v3(lpos[0], lpos[1], lpos[2]));
  f.Qs = qmul(st.Q, Q4{lquat[0], lquat[1], lquat[2], lquat[3]});
  f.Rs = q2m(f.Qs);
  f.bl = mulT(f.Rs, cap.jp(f.ps));
  f.ba = mulT(f.Rs, cap.jr());
  return f;
}

// ------------------------------------------------------------ geom distance ----
struct Contact {
  double dist;
  V3 p1, p2;
};

MK_GPU V3 box_closest(V3 p, V3 c, const M3& r, V3 h) {
  V3 loc = mulT(r, p - c);
  loc.x = fmin(fmax(loc.x, -h.x), h.x);
  loc.y = fmin(fmax(loc.y, -h.y), h.y);
  loc.z = fmin(fmax(loc.z, -h.z), h.z);
  return c + mul(r, loc);
}

MK_GPU void seg_seg(V3 p1, V3 q1, V3 p2, V3 q2, V3& c1, V3& c2) {
  const V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
  const double a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
  const double eps = 1e-30;
  double s, t;
  if (a <= eps && e <= eps) {
    s = t = 0.0;
  } else if (a <= eps) {
    s = 0.0;
    t = fmin(fmax(f / e, 0.0), 1.0);
  } else {
    const double c = dot(d1, r);
    if (e <= eps) {
      t = 0.0;
      s = fmin(fmax(-c / a, 0.0), 1.0);
    } else {
      const double b = dot(d1, d2);
      const double den = a * e - b * b;
      s = den > eps ? fmin(fmax((b * f - c * e) / den, 0.0), 1.0) : 0.0;
      t = (b * s + f) / e;
      if (t < 0.0) {
        t = 0.0;
        s = fmin(fmax(-c / a, 0.0), 1.0);
      } else if (t > 1.0) {
        t = 1.0;
        s = fmin(fmax((b - c) / a, 0.0), 1.0);
      }
    }
  }
  c1 = p1 + s * d1;
  c2 = p2 + t * d2;
}

MK_GPU V3 col(const M3& r, int i) { return v3(r.m[i], r.m[3 + i], r.m[6 + i]); }
MK_GPU double comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
MK_GPU V3 box_vertex(V3 c, const M3& r, V3 h, int k) {
  // k = 4*bx + 2*by + bz with sign -1 for bit 0, +1 for bit 1
  V3 s = v3((k & 4) ? h.x : -h.x, (k & 2) ? h.y : -h.y, (k & 1) ? h.z : -h.z);
  return c + mul(r, s);
}
MK_GPU void box_edge(V3 c, const M3& r, V3 h, int e, V3& a0, V3& a1) {
  // e = 4*a + 2*sb + sc over axis a and the signs of the two other axes b=(a+1)%3, c=(a+2)%3
  const int a = e >> 2, b = (a + 1) % 3, cc = (a + 2) % 3;
  const double sb = (e & 2) ? 1.0 : -1.0, sc = (e & 1) ? 1.0 : -1.0;
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  const double hh[3] = {h.x, h.y, h.z};
  lo[b] = hi[b] = sb * hh[b];
  lo[cc] = hi[cc] = sc * hh[cc];
  lo[a] = -hh[a];
  hi[a] = hh[a];
  a0 = c + mul(r, v3(lo[0], lo[1], lo[2]));
  a1 = c + mul(r, v3(hi[0], hi[1], hi[2]));
}

// Signed distance and witness points of two oriented boxes: separating-axis test for
// overlap (penetration = smallest overlap, witnesses around the midpoint of the two
// support points), otherwise exact closest points by enumerating vertex-box and
// edge-edge pairs in a fixed order (first strict minimum wins).
MK_GPU Contact box_box(V3 c1, const M3& r1, V3 h1, V3 c2, const M3& r2, V3 h2) {
  const V3 d = c2 - c1;
  double best_overlap = INFINITY;
  V3 best_axis = v3(1, 0, 0);
  bool separated = false;
  for (int k = 0; k < 15 && !separated; ++k) {
    V3 ax;
    if (k < 3) ax = col(r1, k);
    else if (k < 6) ax = col(r2, k - 3);
    else ax = cross(col(r1, (k - 6) / 3), col(r2, (k - 6) % 3));
    const double n2 = dot(ax, ax);
    if (n2 < 1e-12) continue;
    ax = (1.0 / sqrt(n2)) * ax;
    const V3 a1 = mulT(r1, ax), a2 = mulT(r2, ax);
    const double ra = fabs(a1.x) * h1.x + fabs(a1.y) * h1.y + fabs(a1.z) * h1.z;
    const double rb = fabs(a2.x) * h2.x + fabs(a2.y) * h2.y + fabs(a2.z) * h2.z;
    const double proj = dot(d, ax);
    const double overlap = ra + rb - fabs(proj);
    if (overlap < 0.0) {
      separated = true;
    } else if (overlap < best_overlap) {
      best_overlap = overlap;
      best_axis = proj >= 0.0 ? ax : neg(ax);
    }
  }
  Contact ct;
  if (!separated) {
    const V3 n = best_axis;
    const V3 a1 = mulT(r1, n), a2 = mulT(r2, n);
    // support features along n: a component of n that vanishes in a box frame selects the
    // middle of that extent (face / edge centre), not an arbitrary vertex
    auto sg = [](double a) { return fabs(a) < 1e-9 ? 0.0 : (a > 0.0 ? 1.0 : -1.0); };
    const V3 s1 = c1 + mul(r1, v3(sg(a1.x) * h1.x, sg(a1.y) * h1.y, sg(a1.z) * h1.z));
    const V3 s2 = c2 + mul(r2, v3(-sg(a2.x) * h2.x, -sg(a2.y) * h2.y, -sg(a2.z) * h2.z));
    const V3 pos = 0.5 * (s1 + s2);
    ct.dist = -best_overlap;
    ct.p1 = pos - (0.5 * ct.dist) * n;
    ct.p2 = pos + (0.5 * ct.dist) * n;
    return ct;
  }
  double best = INFINITY;
  ct.p1 = c1;
  ct.p2 = c2;
  for (int k = 0; k < 8; ++k) {
    const V3 v = box_vertex(c1, r1, h1, k), w = box_closest(v, c2, r2, h2);
    const double dd = dot(v - w, v - w);
    if (dd < best) { best = dd; ct.p1 = v; ct.p2 = w; }
  }
  for (int k = 0; k < 8; ++k) {
    const V3 v = box_vertex(c2, r2, h2, k), w = box_closest(v, c1, r1, h1);
    const double dd = dot(v - w, v - w);
    if (dd < best) { best = dd; ct.p1 = w; ct.p2 = v; }
  }
  for (int ea = 0; ea < 12; ++ea) {
    V3 a0, a1;
    box_edge(c1, r1, h1, ea, a0, a1);
    for (int eb = 0; eb < 12; ++eb) {
      V3 b0, b1, u, w;
      box_edge(c2, r2, h2, eb, b0, b1);
      seg_seg(a0, a1, b0, b1, u, w);
      const double dd = dot(u - w, u - w);
      if (dd < best) { best = dd; ct.p1 = u; ct.p2 = w; }
    }
  }
  ct.dist = sqrt(best);
  return ct;
}

// mj_geomDistance for the supported primitive pairs (types already ordered t1 <= t2).
// Returns false for "farther than distmax".
MK_GPU bool geom_distance(int t1, V3 c1, const M3& r1, V3 s1, int t2, V3 c2, const M3& r2, V3 s2, double distmax,
                          Contact& ct) {
  if (t1 == GEOM_PLANE && t2 == GEOM_BOX) {
    const V3 n = col(r1, 2);
    double best = INFINITY;
    V3 pbest = c2;
    for (int k = 0; k < 8; ++k) {
      const V3 v = box_vertex(c2, r2, s2, k);
      const double depth = dot(v - c1, n);
      if (depth < best) { best = depth; pbest = v; }
    }
    ct.dist = best;
    ct.p2 = pbest;
    ct.p1 = pbest - best * n;
  } else if (t1 == GEOM_PLANE && t2 == GEOM_SPHERE) {
    const V3 n = col(r1, 2);
    ct.dist = dot(c2 - c1, n) - s2.x;
    ct.p2 = c2 - s2.x * n;
    ct.p1 = ct.p2 - ct.dist * n;
  } else if (t1 == GEOM_SPHERE && t2 == GEOM_SPHERE) {
    V3 d = c2 - c1;
    double nrm = sqrt(dot(d, d));
    V3 ax = nrm < MK_MINVAL ? v3(1, 0, 0) : (1.0 / nrm) * d;
    if (nrm < MK_MINVAL) nrm = 0.0;
    ct.dist = nrm - s1.x - s2.x;
    ct.p1 = c1 + s1.x * ax;
    ct.p2 = c2 - s2.x * ax;
  } else {  // box - box
    // bounding spheres first: most pairs are far apart
    const V3 d = c2 - c1;
    const double far = sqrt(dot(d, d)) - sqrt(dot(s1, s1)) - sqrt(dot(s2, s2));
    if (far >= distmax) return false;
    ct = box_box(c1, r1, s1, c2, r2, s2);
  }
  return ct.dist < distmax;
}

// ------------------------------------------------------------------ QP ----
// LDS carve-up for one problem (doubles unless noted).
struct Lds {
  double *P, *HiN, *M, *LM, *G, *h, *rn, *c, *xs, *Linv, *u, *rv;
  int32_t* act;
  uint8_t* is_act;
};

__host__ __device__ inline size_t lds_doubles(int n, int m) {
  return (size_t)4 * n * n + (size_t)m * n + 2 * (size_t)m + 5 * (size_t)n;
}

MK_GPU Lds carve(double* base, int n, int m) {
  Lds s;
  s.P = base;
  s.HiN = s.P + n * n;
  s.M = s.HiN + n * n;
  s.LM = s.M + n * n;
  s.G = s.LM + n * n;
  s.h = s.G + (size_t)m * n;
  s.rn = s.h + m;
  s.c = s.rn + m;
  s.xs = s.c + n;
  s.Linv = s.xs + n;
  s.u = s.Linv + n;
  s.rv = s.u + n;
  s.act = reinterpret_cast<int32_t*>(s.rv + n);
  s.is_act = reinterpret_cast<uint8_t*>(s.act + n);
  return s;
}

// In-place Cholesky of the n x n matrix in LDS (lower triangle, row-major); lane i
// owns row i.  Returns false when a pivot is not positive.
MK_GPU bool chol_wave(double* A, double* Linv, int n, int lane) {
  for (int j = 0; j < n; ++j) {
    double t = 0.0;
    if (lane >= j && lane < n) {
      t = A[lane * n + j];
      for (int k = 0; k < j; ++k) t -= A[lane * n + k] * A[j * n + k];
    }
    const double d = bcast(t, j);
    if (!(d > 0.0)) return false;
    const double r = 1.0 / sqrt(d);
    __syncthreads();
    if (lane >= j && lane < n) A[lane * n + j] = t * r;
    if (lane == j) Linv[j] = r;
    __syncthreads();
  }
  return true;
}

// Solves (L L') y = b for the per-lane vector b (lane i holds element i).
MK_GPU double chol_solve_wave(const double* L, const double* Linv, int n, int lane, double b) {
  for (int j = 0; j < n; ++j) {
    const double y = bcast(b, j) * Linv[j];
    if (lane == j) b = y;
    else if (lane > j && lane < n) b -= L[lane * n + j] * y;
  }
  for (int j = n - 1; j >= 0; --j) {
    const double y = bcast(b, j) * Linv[j];
    if (lane == j) b = y;
    else if (lane < j) b -= L[j * n + lane] * y;
  }
  return b;
}

// min 1/2 x'Px + c'x s.t. Gx <= h; P (LDS, overwritten by its Cholesky factor), c, G,
// h in LDS.  Returns the status; lane i receives x_i.
MK_GPU int gi_solve(Lds& s, int n, int m, int lane, double& x_out) {
  if (!chol_wave(s.P, s.Linv, n, lane)) return 3;
  double x = chol_solve_wave(s.P, s.Linv, n, lane, lane < n ? -s.c[lane] : 0.0);
  if (m == 0) { x_out = x; return 0; }
  for (int j = lane; j < m; j += 64) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += s.G[(size_t)j * n + i] * s.G[(size_t)j * n + i];
    s.rn[j] = sqrt(acc);
    s.is_act[j] = 0;
  }
  int na = 0;
  int status = 1;
  const int max_iter = 10 * (m + n);
  for (int it = 0; it < max_iter; ++it) {
    __syncthreads();
    if (lane < n) s.xs[lane] = x;
    __syncthreads();
    // most violated constraint (normalised by the row norm)
    double worst = 1e-12;
    int p = -1;
    for (int j = lane; j < m; j += 64) {
      if (s.is_act[j] || !(s.rn[j] > 0.0)) continue;
      double v = -s.h[j];
      for (int i = 0; i < n; ++i) v += s.G[(size_t)j * n + i] * s.xs[i];
      v /= s.rn[j];
      if (v > worst) { worst = v; p = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ow = __shfl_xor(worst, o);
      const int op = __shfl_xor(p, o);
      if (ow > worst || (ow == worst && op >= 0 && (p < 0 || op < p))) { worst = ow; p = op; }
    }
    if (p < 0) { status = 0; break; }
    const double np = lane < n ? s.G[(size_t)p * n + lane] : 0.0;
    double up = 0.0;
    bool failed = false;
    for (;;) {
      const double hin = chol_solve_wave(s.P, s.Linv, n, lane, np);
      const double nHn = wave_sum(np * hin);
      // r = M^-1 N' H^-1 n+ ; M = N' H^-1 N is kept up to date in LDS
      for (int a = 0; a < na; ++a) {
        const double ga = lane < n ? s.G[(size_t)s.act[a] * n + lane] : 0.0;
        const double v = wave_sum(ga * hin);
        if (lane == 0) s.rv[a] = v;
      }
      __syncthreads();
      if (na > 0) {
        // the active set is small: lane 0 factors M and solves for r
        int ok = 1;
        if (lane == 0) {
          for (int j = 0; j < na && ok; ++j) {
            double d = s.M[j * n + j];
            for (int k = 0; k < j; ++k) d -= s.LM[j * n + k] * s.LM[j * n + k];
            if (!(d > 0.0)) { ok = 0; break; }
            const double sd = sqrt(d);
            s.LM[j * n + j] = sd;
            for (int i = j + 1; i < na; ++i) {
              double t = s.M[i * n + j];
              for (int k = 0; k < j; ++k) t -= s.LM[i * n + k] * s.LM[j * n + k];
              s.LM[i * n + j] = t / sd;
            }
          }
          if (ok) {
            for (int i = 0; i < na; ++i) {
              double t = s.rv[i];
              for (int k = 0; k < i; ++k) t -= s.LM[i * n + k] * s.rv[k];
              s.rv[i] = t / s.LM[i * n + i];
            }
            for (int i = na - 1; i >= 0; --i) {
              double t = s.rv[i];
              for (int k = i + 1; k < na; ++k) t -= s.LM[k * n + i] * s.rv[k];
              s.rv[i] = t / s.LM[i * n + i];
            }
          }
        }
        ok = __shfl(ok, 0);
        if (!ok) { failed = true; break; }
        __syncthreads();
      }
      double z = hin;
      for (int a = 0; a < na; ++a) z -= (lane < n ? s.HiN[a * n + lane] : 0.0) * s.rv[a];
      z = -z;
      const double nz = -wave_sum(np * z);
      const bool dependent = nz <= 1e-13 * nHn;
      double t1 = INFINITY;
      int kb = -1;
      for (int a = 0; a < na; ++a) {
        const double r = s.rv[a];
        if (r > 0.0 && s.u[a] / r < t1) { t1 = s.u[a] / r; kb = a; }
      }
      const double viol = wave_sum(np * x) - s.h[p];
      const double t2 = dependent ? INFINITY : viol / nz;
      const double t = fmin(t1, t2);
      if (!(t < INFINITY)) { status = 2; failed = true; break; }
      if (!dependent) x += t * z;
      __syncthreads();
      if (lane == 0)
        for (int a = 0; a < na; ++a) s.u[a] -= t * s.rv[a];
      up += t;
      __syncthreads();
      if (!dependent && t2 <= t1) {
        // full step: p joins the active set; extend M by its row / column
        if (lane < n) s.HiN[na * n + lane] = hin;
        __syncthreads();
        for (int b2 = 0; b2 <= na; ++b2) {
          const double v = wave_sum(np * (lane < n ? s.HiN[b2 * n + lane] : 0.0));
          if (lane == 0) { s.M[na * n + b2] = v; s.M[b2 * n + na] = v; }
        }
        if (lane == 0) { s.act[na] = p; s.u[na] = up; s.is_act[p] = 1; }
        ++na;
        __syncthreads();
        break;
      }
      // partial step: the blocking constraint kb leaves the active set
      __syncthreads();
      if (lane == 0) s.is_act[s.act[kb]] = 0;
      for (int a = kb; a < na - 1; ++a) {
        if (lane < n) s.HiN[a * n + lane] = s.HiN[(a + 1) * n + lane];
        if (lane == 0) { s.act[a] = s.act[a + 1]; s.u[a] = s.u[a + 1]; }
      }
      __syncthreads();
      // delete row / column kb of M
      if (lane == 0) {
        for (int a = kb; a < na - 1; ++a)
          for (int b2 = 0; b2 < na; ++b2) s.M[a * n + b2] = s.M[(a + 1) * n + b2];
        for (int a = 0; a < na - 1; ++a)
          for (int b2 = kb; b2 < na - 1; ++b2) s.M[a * n + b2] = s.M[a * n + b2 + 1];
      }
      --na;
      __syncthreads();
    }
    if (failed) { if (status == 1) status = 3; break; }
  }
  x_out = x;
  return status;
}

// ---------------------------------------------------------------- kernels ----
__global__ __launch_bounds__(64) void dense_kernel(MkDevModel m, const MkDenseProblem* __restrict__ prob, int n_rows,
                                                   const double* __restrict__ q, const double* __restrict__ targets,
                                                   int target_stride, double dt, double damping, int B,
                                                   double* __restrict__ P_out, double* __restrict__ c_out,
                                                   double* __restrict__ G_out, double* __restrict__ h_out,
                                                   double* __restrict__ v_out, int32_t* __restrict__ status_out,
                                                   int eval_task, double* __restrict__ err_out,
                                                   double* __restrict__ jac_out) {
  extern __shared__ double lds_base[];
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  const int n = m.nv, mrows = n_rows;
  Lds s = carve(lds_base, n, mrows);
  const double* qb = q + b * m.nq;
  const double* tg = targets ? targets + b * target_stride : nullptr;
  for (int idx = lane; idx < n * n; idx += 64) s.P[idx] = (idx / n == idx % n) ? damping : 0.0;
  if (lane < n) s.c[lane] = 0.0;
  for (int idx = lane; idx < mrows * n; idx += 64) s.G[idx] = 0.0;
  __syncthreads();
  const bool own = lane < n;

  // ---- tasks ----
  for (int t = 0; t < prob->ntask; ++t) {
    const MkDenseTask& T = prob->task[t];
    double Jw[6] = {0, 0, 0, 0, 0, 0}, ew[6] = {0, 0, 0, 0, 0, 0};
    double Jraw[6] = {0, 0, 0, 0, 0, 0}, eraw[6] = {0, 0, 0, 0, 0, 0};
    int k = 6;
    if (T.type == MK_TASK_FRAME || T.type == MK_TASK_RELATIVE_FRAME) {
      const double* tt = tg + T.target_off;
      const Q4 Qt = Q4{tt[0], tt[1], tt[2], tt[3]};
      const V3 pt = v3(tt[4], tt[5], tt[6]);
      Frame f = eval_frame(m, qb, T.body, T.lpos, T.lquat, lane);
      Twist e;
      V3 dl, da;
      double sign;
      if (T.type == MK_TASK_FRAME) {
        // e = log(T_wb^-1 T_wt), J = -jlog(T_wt^-1 T_wb) J_b = -ljacinv(e) J_b
        e = se3_log(qmul(qconj(f.Qs), Qt), mulT(f.Rs, pt - f.ps));
        dl = f.bl;
        da = f.ba;
        sign = -1.0;
      } else {
        // e = log(T_rt^-1 T_rf), J = jlog(T_rt^-1 T_rf) (J_f - Ad(T_rf^-1) J_r) = ljacinv(-e) (...)
        Frame r = eval_frame(m, qb, T.root_body, T.rpos, T.rquat, lane);
        const Q4 Qrf = qmul(qconj(r.Qs), f.Qs);
        const V3 trf = mulT(r.Rs, f.ps - r.ps);
        const M3 Rt = q2m(Qt);
        e = se3_log(qmul(qconj(Qt), Qrf), mulT(Rt, trf - pt));
        const M3 Rrf = q2m(Qrf);
        dl = f.bl - mulT(Rrf, r.bl - cross(trf, r.ba));
        da = f.ba - mulT(Rrf, r.ba);
        e = Twist{e.v, e.w};
        sign = 1.0;
      }
      // ljacinv of e (frame task) or of -e (relative frame task)
      Twist arg = sign < 0 ? e : Twist{neg(e.v), neg(e.w)};
      M3 Ji, Bm;
      V3 jl, ja;
      se3_ljacinv_blocks(arg, Ji, Bm);   // I, 0 below the small-angle switch
      jl = mul(Ji, dl) + mul(Bm, da);
      ja = mul(Ji, da);
      Jraw[0] = sign * jl.x; Jraw[1] = sign * jl.y; Jraw[2] = sign * jl.z;
      Jraw[3] = sign * ja.x; Jraw[4] = sign * ja.y; Jraw[5] = sign * ja.z;
      eraw[0] = e.v.x; eraw[1] = e.v.y; eraw[2] = e.v.z; eraw[3] = e.w.x; eraw[4] = e.w.y; eraw[5] = e.w.z;
    } else if (T.type == MK_TASK_COM) {
      k = 3;
      V3 com = v3(0, 0, 0), acc = v3(0, 0, 0);
      double mass = 0.0;
      for (int bi = 0; bi < T.nbodies; ++bi) {
        const int body = T.bodies[bi];
        LaneCapture cap;
        cap.lane = lane;
        BodyState st = walk(m, qb, body, cap);
        const V3 xi = st.p + mul(st.R, v3(m.body_ipos[3 * body], m.body_ipos[3 * body + 1], m.body_ipos[3 * body + 2]));
        const double mb = m.body_mass[body];
        com = com + mb * xi;
        acc = acc + mb * cap.jp(xi);
        mass += mb;
      }
      const double inv = mass >= MK_MINVAL ? 1.0 / mass : 1.0;
      const double* tt = tg + T.target_off;
      eraw[0] = inv * com.x - tt[0]; eraw[1] = inv * com.y - tt[1]; eraw[2] = inv * com.z - tt[2];
      Jraw[0] = inv * acc.x; Jraw[1] = inv * acc.y; Jraw[2] = inv * acc.z;
    }
    if (T.type == MK_TASK_POSTURE) {
      // diagonal task: J = -I on scalar dofs, e = q* - q (free-joint dofs zeroed)
      double cost = 0.0, e = 0.0;
      if (own && m.dof_scalar[lane]) {
        const int qa = m.dof_qadr[lane];
        cost = T.pcost[lane];
        e = T.ptarget[qa] - qb[qa];
      }
      const double ewi = cost * (-T.gain * e);
      const double mu = T.lm * wave_sum(ewi * ewi);
      if (own) {
        s.P[lane * n + lane] += cost * cost + mu;
        s.c[lane] += ewi * cost;
      }
      if (eval_task == t && own) {
        if (err_out) err_out[b * n + lane] = e;
        if (jac_out)
          for (int j = 0; j < n; ++j) jac_out[(b * n + j) * n + lane] = (j == lane && m.dof_scalar[lane]) ? -1.0 : 0.0;
      }
      __syncthreads();
      continue;
    }
    if (!own)
      for (int r = 0; r < 6; ++r) Jraw[r] = 0.0;
    double mu = 0.0;
    for (int r = 0; r < k; ++r) {
      Jw[r] = T.cost[r] * Jraw[r];
      ew[r] = T.cost[r] * (-T.gain * eraw[r]);
      mu += ew[r] * ew[r];
    }
    mu *= T.lm;
    for (int j = 0; j < n; ++j) {
      double acc = 0.0;
      for (int r = 0; r < k; ++r) acc += Jw[r] * bcast(Jw[r], j);
      if (own) s.P[lane * n + j] += acc;
    }
    if (own) {
      s.P[lane * n + lane] += mu;
      double acc = 0.0;
      for (int r = 0; r < k; ++r) acc += ew[r] * Jw[r];
      s.c[lane] -= acc;
    }
    if (eval_task == t) {
      if (err_out && lane < k) err_out[b * k + lane] = eraw[lane];
      if (jac_out && own)
        for (int r = 0; r < k; ++r) jac_out[(b * k + r) * n + lane] = Jraw[r];
    }
    __syncthreads();
  }

  // ---- limits ----
  for (int l = 0; l < prob->nlimit; ++l) {
    const MkDenseLimit& L = prob->limit[l];
    if (L.type == MK_LIMIT_CONFIGURATION) {
      for (int kk = 0; kk < L.n; ++kk) {
        const int j = L.index[kk];
        const int dof = m.jnt_dofadr[j], qa = m.jnt_qposadr[j];
        const double lo = m.jnt_range[2 * j] + L.min_dist, hi = m.jnt_range[2 * j + 1] - L.min_dist;
        if (lane == dof) {
          s.G[(size_t)(L.row0 + kk) * n + dof] = 1.0;
          s.G[(size_t)(L.row0 + L.n + kk) * n + dof] = -1.0;
          s.h[L.row0 + kk] = L.gain * (hi - qb[qa]);
          s.h[L.row0 + L.n + kk] = L.gain * (qb[qa] - lo);
        }
      }
    } else if (L.type == MK_LIMIT_VELOCITY) {
      for (int kk = 0; kk < L.n; ++kk) {
        const int dof = L.index[kk];
        if (lane == dof) {
          s.G[(size_t)(L.row0 + kk) * n + dof] = 1.0;
          s.G[(size_t)(L.row0 + L.n + kk) * n + dof] = -1.0;
          s.h[L.row0 + kk] = dt * L.vmax[kk];
          s.h[L.row0 + L.n + kk] = dt * L.vmax[kk];
        }
      }
    } else {  // collision avoidance: one row per geom pair
      for (int kk = 0; kk < L.n; ++kk) {
        int g1 = L.index[2 * kk], g2 = L.index[2 * kk + 1];
        const bool swap = m.geom_type[g1] > m.geom_type[g2];
        if (swap) { const int tmp = g1; g1 = g2; g2 = tmp; }
        LaneCapture c1, c2;
        c1.lane = c2.lane = lane;
        const int b1 = m.geom_bodyid[g1], b2 = m.geom_bodyid[g2];
        BodyState s1 = walk(m, qb, b1, c1);
        BodyState s2 = walk(m, qb, b2, c2);
        const V3 x1 = s1.p + mul(s1.R, v3(m.geom_pos[3 * g1], m.geom_pos[3 * g1 + 1], m.geom_pos[3 * g1 + 2]));
        const V3 x2 = s2.p + mul(s2.R, v3(m.geom_pos[3 * g2], m.geom_pos[3 * g2 + 1], m.geom_pos[3 * g2 + 2]));
        const M3 r1 = q2m(qmul(s1.Q, Q4{m.geom_quat[4 * g1], m.geom_quat[4 * g1 + 1], m.geom_quat[4 * g1 + 2], m.geom_quat[4 * g1 + 3]}));
        const M3 r2 = q2m(qmul(s2.Q, Q4{m.geom_quat[4 * g2], m.geom_quat[4 * g2 + 1], m.geom_quat[4 * g2 + 2], m.geom_quat[4 * g2 + 3]}));
        Contact ct;
        const bool active = geom_distance(
            m.geom_type[g1], x1, r1, v3(m.geom_size[3 * g1], m.geom_size[3 * g1 + 1], m.geom_size[3 * g1 + 2]),
            m.geom_type[g2], x2, r2, v3(m.geom_size[3 * g2], m.geom_size[3 * g2 + 1], m.geom_size[3 * g2 + 2]),
            L.d_detect, ct);
        const int row = L.row0 + kk;
        if (!active) {
          if (lane == 0) s.h[row] = INFINITY;
          continue;
        }
        // witness points are (on geom1, on geom2) of the *stored* pair order
        V3 pa = ct.p1, pb = ct.p2;
        LaneCapture *ca = &c1, *cb = &c2;
        if (swap) { pa = ct.p2; pb = ct.p1; ca = &c2; cb = &c1; }
        V3 dn = pb - pa;
        const double nn = sqrt(dot(dn, dn));
        const V3 nrm = nn < MK_MINVAL ? v3(1, 0, 0) : (1.0 / nn) * dn;
        const V3 dj = cb->jp(pb) - ca->jp(pa);
        if (own) s.G[(size_t)row * n + lane] = -dot(nrm, dj);
        if (lane == 0)
          s.h[row] = ct.dist > L.d_min ? L.gain * (ct.dist - L.d_min) / dt + L.relax : L.relax;
      }
    }
  }
  __syncthreads();

  // ---- outputs ----
  if (P_out) {
    for (int idx = lane; idx < n * n; idx += 64) P_out[b * n * n + idx] = s.P[idx];
    if (own) c_out[b * n + lane] = s.c[lane];
    if (G_out) {
      for (int idx = lane; idx < mrows * n; idx += 64) G_out[b * mrows * n + idx] = s.G[idx];
      for (int idx = lane; idx < mrows; idx += 64) h_out[b * mrows + idx] = s.h[idx];
    }
  }
  if (v_out) {
    double x;
    const int st = gi_solve(s, n, mrows, lane, x);
    if (own) v_out[b * n + lane] = x / dt;
    if (lane == 0) status_out[b] = st;
  }
}

__global__ __launch_bounds__(64) void dense_qp_kernel(const double* __restrict__ P, const double* __restrict__ c,
                                                      const double* __restrict__ G, const double* __restrict__ h,
                                                      int n, int mrows, double* __restrict__ x_out,
                                                      int32_t* __restrict__ status_out) {
  extern __shared__ double lds_base[];
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  Lds s = carve(lds_base, n, mrows);
  for (int idx = lane; idx < n * n; idx += 64) s.P[idx] = P[b * n * n + idx];
  if (lane < n) s.c[lane] = c[b * n + lane];
  for (int idx = lane; idx < mrows * n; idx += 64) s.G[idx] = G[b * mrows * n + idx];
  for (int idx = lane; idx < mrows; idx += 64) s.h[idx] = h[b * mrows + idx];
  __syncthreads();
  double x;
  const int st = gi_solve(s, n, mrows, lane, x);
  if (lane < n) x_out[b * n + lane] = x;
  if (lane == 0) status_out[b] = st;
}

}  // namespace

size_t mk_dense_lds_bytes(int nv, int n_rows) {
  if (nv > MK_DENSE_MAX_NV) return 0;
  size_t bytes = lds_doubles(nv, n_rows) * sizeof(double) + (size_t)nv * sizeof(int32_t) + (size_t)n_rows + 64;
  return bytes <= 150 * 1024 ? bytes : 0;
}

hipError_t mk_dense_launch(const MkDevModel& m, const MkDenseProblem* d_problem, int n_rows, const double* q,
                           const double* targets, int target_stride, double dt, double damping, int B, double* P,
                           double* c, double* G, double* h, double* v, int32_t* status, int eval_task, double* err,
                           double* jac, hipStream_t s) {
  const size_t lds = mk_dense_lds_bytes(m.nv, n_rows);
  if (lds == 0) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dense_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(dense_kernel, dim3(B), dim3(64), lds, s, m, d_problem, n_rows, q, targets, target_stride, dt,
                     damping, B, P, c, G, h, v, status, eval_task, err, jac);
  return hipGetLastError();
}

hipError_t mk_dense_qp_launch(const double* P, const double* c, const double* G, const double* h, int n, int m, int B,
                              double* x, int32_t* status, hipStream_t s) {
  const size_t lds = mk_dense_lds_bytes(n, m);
  if (lds == 0) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dense_qp_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(dense_qp_kernel, dim3(B), dim3(64), lds, s, P, c, G, h, n, m, x, status);
  return hipGetLastError();
}
