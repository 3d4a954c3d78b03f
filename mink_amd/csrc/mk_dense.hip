// Dense (general) path: one IK problem per wavefront.
//
// Used for task / limit sets outside the structured form of the fused kernels
// (RelativeFrameTask, ComTask, several frame tasks, CollisionAvoidanceLimit, free or
// ball joints on a task chain) and for mink.build_ik / Task.compute_* / qpsolvers-style
// QP solves.  Per problem, with lane i of the wave owning dof i:
//   1. forward kinematics of the whole tree, one body per lane (each lane walks the path
//      root -> its body), body poses and per-dof world axes / anchors parked in LDS
//      (mj_kinematics; the anchors and axes replace cdof of mj_comPos);
//   2. every frame / contact-point Jacobian column is then O(1) per lane: the dof is on
//      the path of the body iff its bit is set in the body's ancestor-dof mask (mj_jac);
//   3. P = damping I + sum_t (Jw'Jw + mu I), c = -sum_t ew'Jw  (mink/solve_ik.py:13-22),
//      row i of P accumulated by lane i in LDS;
//   4. limits: ConfigurationLimit / VelocityLimit rows are +-e_i, kept as two bounds per
//      lane; CollisionAvoidanceLimit rows (plane / sphere / capsule / box primitives) are
//      general rows in LDS, with the box-box
//      closest-feature search spread over the lanes (mink/solve_ik.py:25-40);
//   5. the Goldfarb-Idnani dual active-set method (the algorithm of the reference's QP
//      backend, quadprog), v = dq / dt.
#include "mk_generic.h"

#include "../../include/mink_amd.h"
#include "mk_device.h"
#include "mk_tree.h"

#include <algorithm>

namespace {

#define MK_GPU __device__ __forceinline__

constexpr int GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_BOX = 6;

// Wave-wide reductions with DPP lane moves (row shifts inside the 16-lane rows, then the
// two row broadcasts): ALU speed, where shuffles through the LDS crossbar cost an LDS
// round trip per step.  After the scan lane 63 holds the result, read back for all lanes.
template <int CTRL, int ROW_MASK>
MK_GPU int dpp_i32(int v, int fill) {
  return __builtin_amdgcn_update_dpp(fill, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
MK_GPU double dpp_f64(double v, double fill) {
  const int lo = dpp_i32<CTRL, ROW_MASK>(__double2loint(v), __double2loint(fill));
  const int hi = dpp_i32<CTRL, ROW_MASK>(__double2hiint(v), __double2hiint(fill));
  return __hiloint2double(hi, lo);
}
MK_GPU double readlane_f64(double v, int src);

MK_GPU double wave_sum(double v) {
  v += dpp_f64<0x111, 0xf>(v, 0.0);   // row_shr:1 (lanes without a source add the fill value)
  v += dpp_f64<0x112, 0xf>(v, 0.0);   // row_shr:2
  v += dpp_f64<0x114, 0xf>(v, 0.0);   // row_shr:4
  v += dpp_f64<0x118, 0xf>(v, 0.0);   // row_shr:8
  v += dpp_f64<0x142, 0xa>(v, 0.0);   // row_bcast:15 into rows 1 and 3
  v += dpp_f64<0x143, 0xc>(v, 0.0);   // row_bcast:31 into rows 2 and 3
  return readlane_f64(v, 63);
}
// Maximum over the wave (NaN entries are ignored by fmax).
MK_GPU double wave_max(double v) {
  v = fmax(v, dpp_f64<0x111, 0xf>(v, -INFINITY));
  v = fmax(v, dpp_f64<0x112, 0xf>(v, -INFINITY));
  v = fmax(v, dpp_f64<0x114, 0xf>(v, -INFINITY));
  v = fmax(v, dpp_f64<0x118, 0xf>(v, -INFINITY));
  v = fmax(v, dpp_f64<0x142, 0xa>(v, -INFINITY));
  v = fmax(v, dpp_f64<0x143, 0xc>(v, -INFINITY));
  return readlane_f64(v, 63);
}
MK_GPU double bcast(double v, int lane) { return __shfl(v, lane); }
MK_GPU V3 bcast3(V3 v, int lane) { return v3(__shfl(v.x, lane), __shfl(v.y, lane), __shfl(v.z, lane)); }

// Wave-wide arg-min / arg-max of (key, id); ties go to the smaller id (ids < 0 mean "no
// candidate" for arg-max).  Every lane receives the result.
struct KeyId {
  double key;
  int id;
};
MK_GPU KeyId min_of(KeyId a, KeyId b) { return (b.key < a.key || (b.key == a.key && b.id < a.id)) ? b : a; }
MK_GPU KeyId max_of(KeyId a, KeyId b) {
  return (b.key > a.key || (b.key == a.key && b.id >= 0 && (a.id < 0 || b.id < a.id))) ? b : a;
}
template <int CTRL, int ROW_MASK>
MK_GPU KeyId dpp_keyid(KeyId v, KeyId fill) {
  return KeyId{dpp_f64<CTRL, ROW_MASK>(v.key, fill.key), dpp_i32<CTRL, ROW_MASK>(v.id, fill.id)};
}
#define MK_WAVE_SCAN(v, fill, OP)                  \
  v = OP(v, dpp_keyid<0x111, 0xf>(v, fill));       \
  v = OP(v, dpp_keyid<0x112, 0xf>(v, fill));       \
  v = OP(v, dpp_keyid<0x114, 0xf>(v, fill));       \
  v = OP(v, dpp_keyid<0x118, 0xf>(v, fill));       \
  v = OP(v, dpp_keyid<0x142, 0xa>(v, fill));       \
  v = OP(v, dpp_keyid<0x143, 0xc>(v, fill));
MK_GPU void wave_argmin(double& key, int& id) {
  // the extreme key first (one value per step); only a tie between lanes -- rare -- needs the scan
  // over (key, id) pairs
  const double best = -wave_max(-key);
  const uint64_t who = __ballot(key == best);
  if (__popcll(who) == 1) {
    id = __builtin_amdgcn_readlane(id, __ffsll((unsigned long long)who) - 1);
    key = best;
    return;
  }
  KeyId v{key, id};
  const KeyId fill{INFINITY, 0x7fffffff};
  MK_WAVE_SCAN(v, fill, min_of)
  key = readlane_f64(v.key, 63);
  id = __builtin_amdgcn_readlane(v.id, 63);
}
MK_GPU void wave_argmax(double& key, int& id) {
  const double best = wave_max(key);
  const uint64_t who = __ballot(key == best);
  if (__popcll(who) == 1) {
    id = __builtin_amdgcn_readlane(id, __ffsll((unsigned long long)who) - 1);
    key = best;
    return;
  }
  KeyId v{key, id};
  const KeyId fill{-INFINITY, -1};
  MK_WAVE_SCAN(v, fill, max_of)
  key = readlane_f64(v.key, 63);
  id = __builtin_amdgcn_readlane(v.id, 63);
}

}  // namespace
#include "mk_convex.h"
#include "mk_geom.h"
MK_CVX void cvx_wave_argmax(double& key, int& id) { wave_argmax(key, id); }
namespace {

// ------------------------------------------------------------ geom distance ----
// Signed distance and witness points of two oriented boxes, computed by the whole wave:
// separating-axis test for overlap (one candidate axis per lane; penetration = smallest
// overlap, witnesses around the midpoint of the two support points), otherwise the closest
// points over the vertex-box candidates and the support edges of the widest cross-product axis,
// one candidate per lane.  Candidates are numbered in a fixed order and the first strict minimum
// wins, so the result does not depend on how they are distributed (box_box_lane is the same
// search by one lane).  A separated pair whose gap along some candidate axis is already
// >= distmax returns that gap (the caller only needs to know that the pair is out of range).
MK_GPU Contact box_box_wave(int lane, V3 c1, const M3& r1, V3 h1, V3 c2, const M3& r2, V3 h2, double distmax) {
  const V3 d = c2 - c1;
  double overlap = INFINITY;
  V3 axis = v3(1, 0, 0);
  bool sep = false;
  if (lane < 15 && box_box_axis(lane, d, r1, h1, r2, h2, overlap, axis)) sep = overlap < 0.0;
  else overlap = INFINITY;
  Contact ct;
  if (__ballot(sep) == 0ull) {
    int id = lane;
    wave_argmin(overlap, id);
    return box_box_penetration(c1, r1, h1, c2, r2, h2, overlap, bcast3(axis, id));
  }
  // the gap along a separating axis is a lower bound of the distance: a pair that is
  // farther apart than distmax along any of them is done without the closest-point search
  double gap = (lane < 15 && overlap < INFINITY) ? -overlap : -INFINITY;
  int gid = lane;
  wave_argmax(gap, gid);
  if (gap >= distmax) {
    ct.dist = gap;
    ct.p1 = c1;
    ct.p2 = c2;
    return ct;
  }
  // the cross-product axis with the widest gap (first one among equals)
  double xgap = (lane >= 6 && lane < 15 && overlap < INFINITY) ? -overlap : -INFINITY;
  int kx = (lane >= 6 && lane < 15 && overlap < INFINITY) ? lane : -1;
  wave_argmax(xgap, kx);
  const V3 nx = bcast3(axis, kx >= 0 ? kx : 0);
  double best = INFINITY;
  int bi = 1 << 30;
  V3 p1 = c1, p2 = c2;
  if (lane < MK_BOX_CANDIDATES) {
    box_box_candidate(lane, kx, nx, c1, r1, h1, c2, r2, h2, p1, p2);
    best = dot(p1 - p2, p1 - p2);
    bi = lane;
  }
  wave_argmin(best, bi);
  ct.p1 = bcast3(p1, bi);
  ct.p2 = bcast3(p2, bi);
  ct.dist = sqrt(best);
  return ct;
}

// mj_geomDistance for the supported primitive pairs (types already ordered t1 <= t2);
// all arguments are wave-uniform.  Returns false for "farther than distmax".
MK_GPU bool geom_distance(int lane, int t1, V3 c1, const M3& r1, V3 s1, int t2, V3 c2, const M3& r2, V3 s2,
                          double distmax, Contact& ct) {
  if (t1 == GEOM_PLANE && t2 == GEOM_BOX) {
    plane_box(c1, r1, c2, r2, s2, ct);
  } else if (t1 == GEOM_PLANE && t2 == GEOM_SPHERE) {
    plane_sphere(c1, r1, c2, s2, ct);
  } else if (t1 == GEOM_SPHERE && t2 == GEOM_SPHERE) {
    sphere_pair(c1, s1.x, c2, s2.x, ct);
  } else if (t1 == GEOM_PLANE && t2 == GEOM_CAPSULE) {
    plane_capsule(c1, r1, c2, r2, s2, ct);
  } else if ((t1 == GEOM_SPHERE || t1 == GEOM_CAPSULE) && t2 == GEOM_CAPSULE) {
    round_capsule(t1 == GEOM_CAPSULE, c1, r1, s1, c2, r2, s2, ct);
  } else if (t1 == GEOM_SPHERE && t2 == GEOM_BOX) {
    sphere_box(c1, s1, c2, r2, s2, ct);
  } else if (t1 == GEOM_CAPSULE && t2 == GEOM_BOX) {
    // Axis segment against the box: one candidate per lane, first strict minimum in their numbering.
    const V3 ax = col(r1, 2);
    const V3 a = c1 - s1.y * ax, b = c1 + s1.y * ax;
    double best = INFINITY;
    int bi = lane;
    V3 u = a, w = c2;
    if (lane < MK_CAPSULE_BOX_CANDIDATES) {
      capsule_box_candidate(lane, a, b, c2, r2, s2, u, w);
      best = dot(u - w, u - w);
    }
    wave_argmin(best, bi);
    u = bcast3(u, bi);
    w = bcast3(w, bi);
    capsule_box_finish(a, b, u, w, s1, c2, r2, s2, ct);
  } else {  // box - box
    // bounding spheres first: most pairs are far apart
    const V3 d = c2 - c1;
    const double far = sqrt(dot(d, d)) - sqrt(dot(s1, s1)) - sqrt(dot(s2, s2));
    if (far >= distmax) return false;
    ct = box_box_wave(lane, c1, r1, s1, c2, r2, s2, distmax);
  }
  return ct.dist < distmax;
}

// ------------------------------------------------------------------ LDS ----
// Carve-up of the block's LDS for one problem (n dofs, mg general rows).  Matrices
// are stored by rows with an odd stride, so that "lane i reads element k of row i" and
// "all lanes read element (j, k)" are both free of bank conflicts.
struct WLds {
  double *q, *tg, *bpos, *bquat, *daxis, *danch;
  double *P;     // Hessian rows (the solver leaves them as they are)
  double *W;     // task-row exchange buffer while building, then the solver's J (J J' = P^-1)
  double *HiN;   // (same space as W)
  double *Minv;  // the solver's triangular factor R of the active constraints; Cholesky work space of wide problems
  double *Gg, *hg, *gn;   // general rows, their bounds and norms
  double *c, *rinv, *xs, *hin, *u, *rr, *rv;
  int32_t *dkind, *act;
  uint8_t* gact;
  int npad, na1;
};

// Buffers that are only live while the QP is being built (inputs, body poses, dof axes,
// task rows) share their space with buffers that only the solver uses (Minv, W / HiN).
__host__ __device__ inline size_t wave_lds_doubles(int n, int mg, int nq, int ts, int nbody) {
  const size_t npad = (size_t)n | 1, na1 = (size_t)(n + 1) | 1;
  const size_t build = (size_t)nq + ts + 7 * (size_t)nbody + 6 * (size_t)n + (n + 1) / 2 + 1;   // q, tg, poses, axes, anchors, dkind
  const size_t qp1 = n * na1;                                                                    // Minv
  const size_t shared1 = build > qp1 ? build : qp1;
  const size_t w = n * npad > 6 * (size_t)n ? n * npad : 6 * (size_t)n;                          // W = L^-1, later HiN; task rows
  return shared1 + w + n * npad + mg * npad + 2 * (size_t)mg     // P, Gg, hg, gn
         + 7 * (size_t)n                                         // c, rinv, xs, hin, u, rr, rv
         + (size_t)(n + 1) / 2 + (mg + 7) / 8 + 2;               // act (int32), gact (bytes)
}

MK_GPU WLds carve(double* base, int n, int mg, int nq, int ts, int nbody) {
  WLds s;
  s.npad = n | 1;
  s.na1 = (n + 1) | 1;
  double* p = base;
  // region shared by the build-time buffers and Minv
  s.q = p;
  s.tg = s.q + nq;
  s.bpos = s.tg + ts;
  s.bquat = s.bpos + 3 * nbody;
  s.daxis = s.bquat + 4 * nbody;
  s.danch = s.daxis + 3 * n;
  s.dkind = reinterpret_cast<int32_t*>(s.danch + 3 * n);
  s.Minv = p;
  {
    const size_t build = (size_t)nq + ts + 7 * (size_t)nbody + 6 * (size_t)n + (n + 1) / 2 + 1;
    const size_t qp1 = (size_t)n * s.na1;
    p += build > qp1 ? build : qp1;
  }
  s.W = p;
  s.HiN = p;   // W = L^-1 is dead once H^-1 is formed; HiN is first written after that
  p += (n * s.npad > 6 * n) ? n * s.npad : 6 * n;
  s.P = p; p += n * s.npad;
  s.Gg = p; p += mg * s.npad;
  s.hg = p; p += mg;
  s.gn = p; p += mg;
  s.c = p; p += n;
  s.rinv = p; p += n;
  s.xs = p; p += n;
  s.hin = p; p += n;
  s.u = p; p += n;
  s.rr = p; p += n;
  s.rv = p; p += n;
  s.act = reinterpret_cast<int32_t*>(p);
  s.gact = reinterpret_cast<uint8_t*>(s.act + n + (n & 1));
  return s;
}

// ------------------------------------------------------------------ QP ----
// Value of `v` on lane `src` (wave-uniform index), delivered through scalar registers:
// no LDS round trip, and the result can feed vector instructions directly.
MK_GPU double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Cholesky factor of the n x n matrix whose rows sit in LDS at H (stride np; left as it is),
// n <= NP, taken in the elimination order `order` (position -> variable), and J = L^-T written to
// Jm: row `variable`, with the column of elimination position k stored at n - 1 - k (the variables
// eliminated last own the first columns, see the block start of gi_solve).  Lane i keeps row i of
// the permuted matrix / of L in registers, then column i of L^-1, which is a row of J, and reads
// other lanes' entries with readlane.  Fully unrolled: about n^2 fused multiply-adds and as many
// lane reads, no barriers and no LDS traffic until J is written out.  False when a pivot is not
// positive.
template <int NP>
__device__ __attribute__((noinline)) bool factor_regs(const double* H, double* Jm, const int32_t* order, int np, int n,
                                                      int lane) {
  const int var = lane < n ? order[lane] : 0;
  double A[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) A[j] = (lane < n && j < n) ? H[var * np + order[j]] : 0.0;
  // (lane j keeps 1 / L_jj in place of L_jj: only the reciprocal is used afterwards)
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    if (j < n) {
      // pivot: lane j subtracts the squares of its finished entries
      double t = A[j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= A[k] * A[k];
      const double d = readlane_f64(t, j);
      if (!(d > 0.0)) return false;
      const double r = mk_rsqrt(d);
      // column j of the other rows
      double u = A[j];
#pragma unroll
      for (int k = 0; k < j; ++k) u -= A[k] * readlane_f64(A[k], j);
      A[j] = (lane == j) ? r : ((lane > j) ? u * r : 0.0);
    }
    MK_SCHED_FENCE();
  }
  // column `lane` of L^-1 by forward substitution; L_ik comes from lane i
  double Wc[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    Wc[i] = 0.0;
    if (i < n) {
      double t = (i == lane) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) t -= readlane_f64(A[k], i) * Wc[k];
      Wc[i] = t * readlane_f64(A[i], i);
    }
    MK_SCHED_FENCE();
  }
  __syncthreads();          // every lane has read `order`, which shares its space with the caller's list
#pragma unroll
  for (int i = 0; i < NP; ++i)
    if (lane < n && i < n) Jm[var * np + (n - 1 - i)] = Wc[i];
  __syncthreads();
  return true;
}

// The same through LDS for any n <= 64: the factor of the permuted matrix is formed in `tmp`
// (n x np doubles), lane j then builds its row of J by forward substitution.
MK_GPU bool factor_lds(const double* H, double* Jm, const int32_t* order, double* tmp, double* rinv, int np, int n,
                       int lane) {
  const int var = lane < n ? order[lane] : 0;
  if (lane < n)
    for (int j = 0; j < n; ++j) tmp[lane * np + j] = H[var * np + order[j]];
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    double t = 0.0;
    if (lane >= j && lane < n) {
      const double* ri = tmp + lane * np;
      const double* rj = tmp + j * np;
      t = ri[j];
#pragma unroll 4
      for (int k = 0; k < j; ++k) t -= ri[k] * rj[k];
    }
    const double d = bcast(t, j);
    if (!(d > 0.0)) return false;
    const double r = mk_rsqrt(d);
    if (lane >= j && lane < n) tmp[lane * np + j] = (lane == j) ? d * r : t * r;
    if (lane == j) rinv[j] = r;
    __syncthreads();
  }
  if (lane < n) {
    double* jr = Jm + var * np + (n - 1);      // position k at jr[-k]
    for (int i = 0; i < n; ++i) {
      double t = (i == lane) ? 1.0 : 0.0;
      const double* li = tmp + i * np;
#pragma unroll 4
      for (int k = lane; k < i; ++k) t -= li[k] * jr[-k];
      jr[-i] = t * rinv[i];
    }
  }
  __syncthreads();
  return true;
}

// R y = t and R'y = t for the upper triangular na x na matrix R (rows in LDS, stride na1;
// rdiag holds the reciprocals of its diagonal); entry a of the right-hand side and of the solution
// on lane a.  na sequential steps, each one lane read and one multiply-add per lane.
MK_GPU double tri_back(const double* R, const double* rdiag, int na1, int na, double t, int lane) {
  for (int b = na - 1; b >= 0; --b) {
    const double yb = readlane_f64(t, b) * rdiag[b];
    if (lane == b) t = yb;
    else if (lane < b) t -= R[lane * na1 + b] * yb;
  }
  return lane < na ? t : 0.0;
}
MK_GPU double tri_fwd_t(const double* R, const double* rdiag, int na1, int na, double t, int lane) {
  for (int b = 0; b < na; ++b) {
    const double yb = readlane_f64(t, b) * rdiag[b];
    if (lane == b) t = yb;
    else if (lane > b && lane < na) t -= R[b * na1 + lane] * yb;
  }
  return lane < na ? t : 0.0;
}

// min 1/2 x'Px + c'x  s.t.  lo_i <= x_i <= hi_i (lane i) and Gg x <= hg (mg general rows),
// by the Goldfarb-Idnani dual method in its factorised form (what quadprog, the reference's QP
// backend, implements: mink/solve_ik.py:101).  P, c, Gg, hg in LDS; P is left as it is.
// Constraint ids: i = upper bound of x_i, n + i = lower bound, 2n + r = general row r.
//
// With P = L L', the method keeps J with J J' = P^-1 (J = L^-T at the start) and the upper
// triangular R with J'N = [R; 0] for the normals N of the active constraints, so that for a
// normal n+ and d = J'n+
//   the step direction in x is      -J2 d2        (J2 = columns na.. of J, d2 likewise),
//   its curvature                   n+'J2 J2'n+ = |d2|^2,
//   the multipliers' direction      R^-1 d1.
// A constraint joins by one Householder reflection of the columns J2 (which takes d2 to a multiple
// of e1; the new column of R is (d1; that multiple)); one leaves by deleting its column of R and
// restoring the triangle with Givens rotations of neighbouring rows of R / columns of J.  No
// inverse is formed: errors grow with cond(P), not with its square.
// Lane i owns row i of J (LDS, conflict-free by rows and by columns), lane a entry a of the
// vectors over the active set.  Returns the status; lane i receives x_i.
MK_GPU int gi_solve(WLds& s, int n, int mg, int lane, double lo, double hi, double& x_out) {
  const int np = s.npad, na1 = s.na1;
  double* const J = s.W;
  double* const R = s.Minv;
  double* const rdiag = s.rinv;
  // J for the elimination order in s.act (position -> variable)
  auto factor = [&]() -> bool {
    if (n <= 16) return factor_regs<16>(s.P, J, s.act, np, n, lane);
    if (n <= 24) return factor_regs<24>(s.P, J, s.act, np, n, lane);
    if (n <= 32) return factor_regs<32>(s.P, J, s.act, np, n, lane);
    return factor_lds(s.P, J, s.act, s.Minv, s.rinv, np, n, lane);
  };
  if (lane < n) s.act[lane] = lane;
  __syncthreads();
  if (!factor()) return 3;
  const double* jrow = J + (lane < n ? lane : 0) * np;   // the lane's row of J
  // x0 = -P^-1 c = -J (J'c)
  {
    double t = 0.0;
    if (lane < n) {
#pragma unroll 4
      for (int i = 0; i < n; ++i) t += J[i * np + lane] * s.c[i];
    }
    if (lane < n) s.rr[lane] = t;
    __syncthreads();
  }
  double x = 0.0;
  if (lane < n) {
#pragma unroll 4
    for (int l = 0; l < n; ++l) x -= jrow[l] * s.rr[l];
  }
  for (int r = lane; r < mg; r += 64) {
    double acc = 0.0;
    for (int j = 0; j < n; ++j) acc += s.Gg[r * np + j] * s.Gg[r * np + j];
    s.gn[r] = sqrt(acc);
    s.gact[r] = 0;
  }
#if defined(MK_DENSE_ABLATE) && MK_DENSE_ABLATE == 3
  x_out = x;   // tuning build only: factorisation and the unconstrained minimiser
  return 0;
#endif
  bool act_hi = false, act_lo = false;
  int na = 0;
  int status = 1;

  // Constraint with normal d = J'n+ (in s.hin) joins as number na: jd = J2 d2 of the lane's row,
  // dd2 = |d2|^2 > 0.
  auto join = [&](int p, double up, double jd, double dd2) {
    const double dna = s.hin[na];
    const double nrm = dd2 * mk_rsqrt(dd2);
    const double alpha = dna > 0.0 ? -nrm : nrm;
    const double beta = mk_rcp(dd2 - alpha * dna);          // 2 / v'v for v = d2 - alpha e1 (dd2 + |dna| nrm > 0)
    if (lane < n) {
      double* __restrict__ jr = J + lane * np;
      const double* __restrict__ dv = s.hin;
      const double bw = beta * (jd - alpha * jr[na]);        // beta (J2 v) of this row
      jr[na] -= bw * (dna - alpha);
      // (loads of a group of eight before its stores: the row and d share the LDS, and a
      // read-modify-write loop would otherwise wait for every element's round trip in turn)
      int l = na + 1;
      for (; l + 8 <= n; l += 8) {
        double a[8], h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = jr[l + k]; h[k] = dv[l + k]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) jr[l + k] = a[k] - bw * h[k];
      }
      double a[8], h[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { a[k] = l + k < n ? jr[l + k] : 0.0; h[k] = l + k < n ? dv[l + k] : 0.0; }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (l + k < n) jr[l + k] = a[k] - bw * h[k];
    }
    if (lane < na) R[lane * na1 + na] = s.hin[lane];
    if (lane == 0) {
      R[na * na1 + na] = alpha;
      rdiag[na] = alpha > 0.0 ? mk_rcp(alpha) : -mk_rcp(-alpha);
      s.act[na] = p;
      s.u[na] = up;
      if (p >= 2 * n) s.gact[p - 2 * n] = 1;
    }
    if (p < n && lane == p) act_hi = true;
    if (p >= n && p < 2 * n && lane == p - n) act_lo = true;
    ++na;
    __syncthreads();
  };
  // Active constraint number kb leaves.
  auto leave = [&](int kb) {
    const int idk = s.act[kb];
    if (idk < n) { if (lane == idk) act_hi = false; }
    else if (idk < 2 * n) { if (lane == idk - n) act_lo = false; }
    else if (lane == 0) s.gact[idk - 2 * n] = 0;
    for (int a = kb; a < na - 1; ++a) {
      __syncthreads();
      // rotation of rows (a, a + 1) of R that clears the subdiagonal entry of column a + 1
      const double f = R[a * na1 + a + 1], g = R[(a + 1) * na1 + a + 1];
      const double ih = mk_rsqrt(f * f + g * g);
      const double cs = f * ih, sn = g * ih;
      if (lane > a && lane < na) {
        const double ra = R[a * na1 + lane], rb = R[(a + 1) * na1 + lane];
        R[a * na1 + lane] = cs * ra + sn * rb;
        R[(a + 1) * na1 + lane] = cs * rb - sn * ra;
      }
      if (lane < n) {
        double* jr = J + lane * np;
        const double ja = jr[a], jb = jr[a + 1];
        jr[a] = cs * ja + sn * jb;
        jr[a + 1] = cs * jb - sn * ja;
      }
    }
    __syncthreads();
    // columns kb + 1 .. of R move one to the left (lane = row), and with them act / u
    const int idn = (lane >= kb && lane < na - 1) ? s.act[lane + 1] : 0;
    const double un = (lane >= kb && lane < na - 1) ? s.u[lane + 1] : 0.0;
    if (lane < na) {
      double* rr = R + lane * na1;
      for (int b = kb; b < na - 1; ++b) rr[b] = rr[b + 1];
    }
    __syncthreads();
    if (lane >= kb && lane < na - 1) {
      s.act[lane] = idn;
      s.u[lane] = un;
      rdiag[lane] = 1.0 / R[lane * na1 + lane];
    }
    --na;
    __syncthreads();
  };
  // d = J'n+ for constraint p to s.hin; returns the lane's entry
  auto normal_to_d = [&](int p) -> double {
    double d = 0.0;
    if (p < 2 * n) {
      const int j = p < n ? p : p - n;
      if (lane < n) d = (p < n) ? J[j * np + lane] : -J[j * np + lane];
    } else if (lane < n) {
      const double* g = s.Gg + (p - 2 * n) * np;
#pragma unroll 4
      for (int i = 0; i < n; ++i) d += J[i * np + lane] * g[i];
    }
    __syncthreads();          // readers of the previous contents of hin are done
    if (lane < n) s.hin[lane] = d;
    __syncthreads();
    return d;
  };

  // ---- block start: all bounds violated at the unconstrained minimiser at once ----
  // With A = those bounds, x = x0 - P^-1 N u, u = (N'P^-1N)^-1 (N'x0 - b) is the minimiser on the
  // face A; if every u_a >= 0 this is an S-pair and the dual method may start from it.  The factors
  // for A come without a single reflection: with the variables of A eliminated LAST, the rows of
  // J = L^-T that belong to them are zero outside the columns of A, so J'N is triangular as it
  // stands -- constraint a sits at elimination position n - 1 - a, whose column is stored as
  // column a -- and R is read off those rows.  One more factorisation (in registers) instead of
  // one reflection of J through the LDS per bound.  Bounds whose multiplier comes out negative are
  // taken out and the face is solved again; after three tries the method starts from the empty
  // set as usual (J is a valid factor whatever the order it was formed in).
  {
    uint64_t mh = __ballot(lane < n && x - hi > 1e-12);
    uint64_t ml = __ballot(lane < n && lo - x > 1e-12);
    bool accepted = false;
#if defined(MK_DENSE_ABLATE) && MK_DENSE_ABLATE == 5
    mh = ml = 0ull;    // tuning build only: no block start
#endif
    for (int attempt = 0; attempt < 3 && (mh | ml) != 0ull; ++attempt) {
      const uint64_t mm = mh | ml;
      const int cnt = __popcll(mm);
      const uint64_t below = (1ull << lane) - 1ull;
      const bool mine = lane < n && ((mm >> lane) & 1ull);
      const int a_mine = __popcll(mm & below);                 // number of my bound among the active ones
      const int pos = mine ? n - 1 - a_mine : __popcll(~mm & below);
      __syncthreads();
      if (lane < n) s.act[pos] = lane;
      __syncthreads();
      if (!factor()) break;                                    // (cannot happen: the first one went through)
      // R, the constraint list, and the violations
      const double sg = ((mh >> lane) & 1ull) ? 1.0 : -1.0;
      if (mine) {
        for (int b = 0; b <= a_mine; ++b) R[b * na1 + a_mine] = sg * jrow[b];
        const double dg = sg * jrow[a_mine];
        rdiag[a_mine] = dg > 0.0 ? mk_rcp(dg) : -mk_rcp(-dg);
        s.act[a_mine] = sg > 0.0 ? lane : n + lane;
      }
      __syncthreads();
      const int ida = lane < cnt ? s.act[lane] : 0;
      const int ja = ida < n ? ida : ida - n;
      const double xj = __shfl(x, ja), hj = __shfl(hi, ja), lj = __shfl(lo, ja);
      const double viol = lane < cnt ? (ida < n ? xj - hj : lj - xj) : 0.0;
      const double ua = tri_back(R, rdiag, na1, cnt, tri_fwd_t(R, rdiag, na1, cnt, viol, lane), lane);
      const uint64_t negm = __ballot(lane < cnt && ua < 0.0);
      if (negm != 0ull) {
        // take the bounds with a negative multiplier out of the candidate set
        const double us = __shfl(ua, a_mine);
        const uint64_t dropm = __ballot(mine && us < 0.0);
        mh &= ~dropm;
        ml &= ~dropm;
        continue;
      }
      // accepted: x = x0 - J1 (R u), multipliers, flags
      if (lane < cnt) s.u[lane] = ua;
      __syncthreads();
      double ru = 0.0;
      if (lane < cnt)
        for (int b = lane; b < cnt; ++b) ru += R[lane * na1 + b] * s.u[b];
      if (lane < cnt) s.rr[lane] = ru;
      __syncthreads();
      if (lane < n) {
#pragma unroll 4
        for (int a = 0; a < cnt; ++a) x -= jrow[a] * s.rr[a];
      }
      act_hi = mine && sg > 0.0;
      act_lo = mine && sg < 0.0;
      na = cnt;
      accepted = true;
      __syncthreads();
      break;
    }
    if (!accepted) {
      na = 0;
      act_hi = act_lo = false;
    }
  }
#if defined(MK_DENSE_ABLATE) && MK_DENSE_ABLATE == 6
  x_out = x;    // tuning build only: stop after the block start
  return 0;
#endif
#if defined(MK_DENSE_COUNT_STEPS)
  int n_full = 0, n_partial = 0;   // tuning build only: step counts returned in place of the status
#endif
  const int max_iter = 10 * (mg + 3 * n);
  for (int it = 0; it < max_iter; ++it) {
    // ---- most violated constraint (normalised by the row norm) ----
    if (lane < n) s.xs[lane] = x;
    __syncthreads();
    double worst = 1e-12;
    int p = -1;
    if (lane < n) {
      if (!act_hi && x - hi > worst) { worst = x - hi; p = lane; }
      if (!act_lo && lo - x > worst) { worst = lo - x; p = n + lane; }
    }
    for (int r = lane; r < mg; r += 64) {
      if (s.gact[r] || !(s.gn[r] > 0.0)) continue;
      double v = -s.hg[r];
#pragma unroll 4
      for (int j = 0; j < n; ++j) v += s.Gg[r * np + j] * s.xs[j];
      v /= s.gn[r];
      if (v > worst) { worst = v; p = 2 * n + r; }
    }
    wave_argmax(worst, p);
    if (p < 0) { status = 0; break; }
    // ---- its normal n+ (the lane's component) and bound ----
    double npl = 0.0, hp;
    if (p < 2 * n) {
      const int j = p < n ? p : p - n;
      if (lane == j) npl = p < n ? 1.0 : -1.0;
      hp = readlane_f64(p < n ? hi : -lo, j);
    } else {
      if (lane < n) npl = s.Gg[(p - 2 * n) * np + lane];
      hp = s.hg[p - 2 * n];
    }
    double up = 0.0;
    bool failed = false;
    for (;;) {
      const double d = normal_to_d(p);
      const double dd = wave_sum(lane < n ? d * d : 0.0);                 // n+'P^-1 n+
      const double dd2 = wave_sum(lane >= na && lane < n ? d * d : 0.0);  // curvature along the step
      double jd = 0.0;
      if (lane < n) {
#pragma unroll 4
        for (int l = na; l < n; ++l) jd += jrow[l] * s.hin[l];
      }
      const double z = -jd;                                                // step direction in x
      const double ra = tri_back(R, rdiag, na1, na, lane < na ? d : 0.0, lane);
      const bool dependent = dd2 <= 1e-13 * dd;
      // largest step that keeps the multipliers non-negative
      double t1 = INFINITY;
      int kb = 1 << 30;
      if (lane < na && ra > 0.0) { t1 = s.u[lane] / ra; kb = lane; }
      wave_argmin(t1, kb);
      const double viol = (p < 2 * n ? (p < n ? readlane_f64(x, p) : -readlane_f64(x, p - n)) : wave_sum(npl * x)) - hp;
      const double t2 = dependent ? INFINITY : viol * mk_rcp(dd2);
      const double t = fmin(t1, t2);
      if (!(t < INFINITY)) { status = 2; failed = true; break; }
      if (!dependent) x += t * z;
      if (lane < na) s.u[lane] -= t * ra;
      up += t;
      if (!dependent && t2 <= t1) {
        // full step: p joins the active set
        join(p, up, jd, dd2);
#if defined(MK_DENSE_COUNT_STEPS)
        ++n_full;
#endif
        break;
      }
      // partial step: the blocking constraint kb leaves the active set
#if defined(MK_DENSE_COUNT_STEPS)
      ++n_partial;
#endif
      __syncthreads();
      leave(kb);
    }
    if (failed) break;
  }
  // ---- refinement on the final active set ----
  // With A the final active set, up to two passes of iterative refinement on its KKT system --
  // residuals from P, c and the rows themselves, corrections through the factors,
  //   r = P x + c + N u,  s = N'x - b,  g = J'r,  y = R^-T s,
  //   du = R^-1 (y - g1),  dx = -J (y; g2)
  // -- bring x to working accuracy; active bounds are then met exactly.  A pass is kept only if
  // the residuals it was computed from are smaller than those of the pass before.
#if defined(MK_DENSE_ABLATE) && MK_DENSE_ABLATE == 4
  x_out = x;    // tuning build only: no refinement, no final check
  return status;
#endif
  if (status == 0) {
    double x_prev = x, m_prev = INFINITY, s_kept = 0.0;
    // largest diagonal entry of P: x units -> gradient units
    const double hmax = wave_max(lane < n ? s.P[lane * np + lane] : 0.0);
    for (int pass = 0; pass < 3; ++pass) {
      __syncthreads();
      if (lane < n) s.xs[lane] = x;
      __syncthreads();
      double r = 0.0;
      if (lane < n) {
        r = s.c[lane];
        const double* pr = s.P + lane * np;
#pragma unroll 4
        for (int j = 0; j < n; ++j) r += pr[j] * s.xs[j];
      }
      for (int a = 0; a < na; ++a) {
        const int id = s.act[a];
        const double ua = s.u[a];
        if (id < n) { if (lane == id) r += ua; }
        else if (id < 2 * n) { if (lane == id - n) r -= ua; }
        else if (lane < n) r += s.Gg[(id - 2 * n) * np + lane] * ua;
      }
      // lane a < na: residual of its own equation n_a'x = b_a
      const int ida = lane < na ? s.act[lane] : 0;
      const int ja = ida < n ? ida : (ida < 2 * n ? ida - n : 0);
      double sres = 0.0;
      {
        const double xj = __shfl(x, ja), hj = __shfl(hi, ja), lj = __shfl(lo, ja);
        if (lane < na) {
          if (ida < n) sres = xj - hj;
          else if (ida < 2 * n) sres = lj - xj;
          else {
            const double* g = s.Gg + (ida - 2 * n) * np;
            sres = -s.hg[ida - 2 * n];
            for (int j = 0; j < n; ++j) sres += g[j] * s.xs[j];
          }
        }
      }
      // the size of this x's residuals decides whether the pass that produced it is kept
      double m = fmax(fabs(r), hmax * fabs(sres));
      if (!(m == m)) m = INFINITY;
      double ms = fabs(sres);
      if (!(ms == ms)) ms = INFINITY;
      m = wave_max(m);
      ms = wave_max(ms);
      if (!(m < m_prev)) {   // wave-uniform
        x = x_prev;
        break;
      }
      x_prev = x;
      m_prev = m;
      s_kept = ms;
      if (pass == 2) break;
      // residuals at rounding level of the gradient's own scale: nothing to refine
      if (m <= 1e-13 * wave_max(lane < n ? fabs(s.c[lane]) + hmax * fabs(x) : 0.0)) break;
      if (lane < n) s.hin[lane] = r;
      __syncthreads();
      double g = 0.0;
      if (lane < n) {
#pragma unroll 4
        for (int i = 0; i < n; ++i) g += J[i * np + lane] * s.hin[i];
      }
      const double y = tri_fwd_t(R, rdiag, na1, na, sres, lane);
      const double du = tri_back(R, rdiag, na1, na, lane < na ? y - g : 0.0, lane);
      if (lane < n) s.rr[lane] = lane < na ? y : g;
      __syncthreads();
      bool small;
      {
        double acc = 0.0;
        if (lane < n) {
#pragma unroll 4
          for (int l = 0; l < n; ++l) acc += jrow[l] * s.rr[l];
        }
        // a correction far below the size of x itself: x was accurate already, nothing to check
        const double dm = wave_max(lane < n ? fabs(acc) : 0.0), xm = wave_max(lane < n ? fabs(x) : 0.0);
        small = dm <= 1e-11 * xm;
        if (lane < n) x -= acc;
      }
      if (small) break;   // wave-uniform
      if (lane < na) s.u[lane] += du;
    }
    if (act_hi && fabs(x - hi) <= 1e-9 * fmax(1.0, fabs(hi))) x = hi;
    if (act_lo && fabs(lo - x) <= 1e-9 * fmax(1.0, fabs(lo))) x = lo;
    // The dual method only looks for violations among the rows it holds inactive; the active ones
    // are met by construction -- up to what the factors lose.  An answer that misses its own
    // active rows by more than 1e-7 of its size is not handed out as solved.
    // Likewise for stationarity: a residual of P x + c + N u above 1e-7 of the gradient's own scale.
    {
      const double xm = wave_max(lane < n ? fabs(x) : 0.0), cm = wave_max(lane < n ? fabs(s.c[lane]) : 0.0);
      if (!(s_kept <= 1e-7 * fmax(1.0, xm))) status = 1;
      if (!(m_prev <= 1e-7 * fmax(cm + hmax * xm, 1e-300))) status = 1;
    }
  }
  x_out = x;
#if defined(MK_DENSE_COUNT_STEPS)
  return n_full + 100 * n_partial;
#endif
  return status;
}

// ------------------------------------------------------------- kinematics ----
// Writes the world axis / anchor of every dof of the walked body to LDS.
struct EmitDof {
  double *daxis, *danch;
  int32_t* dkind;
  bool on;
  MK_GPU void operator()(int dof, int kind, V3 ax, V3 an) {
    if (!on) return;
    daxis[3 * dof] = ax.x; daxis[3 * dof + 1] = ax.y; daxis[3 * dof + 2] = ax.z;
    danch[3 * dof] = an.x; danch[3 * dof + 1] = an.y; danch[3 * dof + 2] = an.z;
    dkind[dof] = kind;
  }
};

// Whole-tree forward kinematics by path walking: lane l walks the path root -> body l
// (+64, ...), so all bodies advance together; the last body of a walk is the lane's own,
// whose dofs it emits.  Any tree size; every step of a walk waits on the table loads of
// its body, so this is the slow path, used when the tree has more than 64 bodies.
MK_GPU void tree_fk_walk(const MkDevModel& m, WLds& s, int lane) {
  EmitDof em{s.daxis, s.danch, s.dkind, false};
  for (int b0 = 0; b0 < m.nbody; b0 += 64) {
    const int body = b0 + lane;
    const bool has = body < m.nbody;
    const int adr = has ? m.path_adr[body] : 0, len = has ? m.path_len[body] : 0;
    int maxlen = len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));
    BodyState st = body_state_identity();
    for (int i = 0; i < maxlen; ++i) {
      if (i < len) {
        em.on = (i == len - 1);
        advance_body<true>(m, s.q, m.path_data[adr + i], st, em);
      }
    }
    if (has) {
      s.bpos[3 * body] = st.p.x; s.bpos[3 * body + 1] = st.p.y; s.bpos[3 * body + 2] = st.p.z;
      s.bquat[4 * body] = st.Q.w; s.bquat[4 * body + 1] = st.Q.x; s.bquat[4 * body + 2] = st.Q.y;
      s.bquat[4 * body + 3] = st.Q.z;
    }
  }
  __syncthreads();
}

// Dof axes / anchors in the frame of the body's parent, tagged with that parent.
struct EmitDofLocal {
  double *daxis, *danch;
  int32_t* dkind;
  int parent;
  MK_GPU void operator()(int dof, int kind, V3 ax, V3 an) {
    daxis[3 * dof] = ax.x; daxis[3 * dof + 1] = ax.y; daxis[3 * dof + 2] = ax.z;
    danch[3 * dof] = an.x; danch[3 * dof + 1] = an.y; danch[3 * dof + 2] = an.z;
    dkind[dof] = kind + 2 * parent;
  }
};

// Whole-tree forward kinematics for trees of at most 64 bodies, lane = body, in three
// passes whose table loads are all issued up front:
//   A  every body's transform relative to its parent (offset, then its joints at q) and
//      its dofs' axes / anchors in the parent's frame -- no dependence on other bodies;
//   B  level by level from the root, world pose = parent's world pose (LDS) o local;
//   C  (by the caller, lane = dof) the dof's axis / anchor moved to the world frame with
//      the parent's world pose.
// A body on a free joint carries its absolute pose and is composed with the world body.
MK_GPU void tree_fk_levels(const MkDevModel& m, WLds& s, int lane) {
  const int body = lane;
  const bool has = body < m.nbody;
  int depth = 0, parent = 0;
  BodyState st = body_state_identity();
  if (has) {
    depth = m.path_len[body];
    const int ja = m.body_jntadr[body], jn = m.body_jntnum[body];
    const bool absolute = jn == 1 && m.jnt_type[ja] == MK_JNT_FREE;
    parent = absolute ? 0 : m.body_parentid[body];
    EmitDofLocal em{s.daxis, s.danch, s.dkind, parent};
    advance_body<true>(m, s.q, body, st, em);
    if (depth == 0) {
      s.bpos[3 * body] = st.p.x; s.bpos[3 * body + 1] = st.p.y; s.bpos[3 * body + 2] = st.p.z;
      s.bquat[4 * body] = st.Q.w; s.bquat[4 * body + 1] = st.Q.x; s.bquat[4 * body + 2] = st.Q.y;
      s.bquat[4 * body + 3] = st.Q.z;
    }
  }
  int maxdepth = depth;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) maxdepth = max(maxdepth, __shfl_xor(maxdepth, o));
  __syncthreads();
  for (int d = 1; d <= maxdepth; ++d) {
    if (has && depth == d) {
      V3 pp;
      Q4 pq;
      pp = v3(s.bpos[3 * parent], s.bpos[3 * parent + 1], s.bpos[3 * parent + 2]);
      pq = Q4{s.bquat[4 * parent], s.bquat[4 * parent + 1], s.bquat[4 * parent + 2], s.bquat[4 * parent + 3]};
      const V3 pw = pp + qrot(pq, st.p);
      const Q4 qw = qnormalize(qmul(pq, st.Q));
      s.bpos[3 * body] = pw.x; s.bpos[3 * body + 1] = pw.y; s.bpos[3 * body + 2] = pw.z;
      s.bquat[4 * body] = qw.w; s.bquat[4 * body + 1] = qw.x; s.bquat[4 * body + 2] = qw.y;
      s.bquat[4 * body + 3] = qw.z;
    }
    __syncthreads();
  }
}

// The lane's own dof: how it moves a point attached to a body whose path contains it.
struct DofCol {
  V3 axis, anchor;
  int kind, lane;
  MK_GPU bool on(uint64_t mask) const { return (mask >> lane) & 1ull; }
  // world-aligned Jacobian column of a point rigidly attached to a body with ancestor mask `mask`
  MK_GPU V3 jp(uint64_t mask, V3 point) const {
    if (!on(mask)) return v3(0, 0, 0);
    return kind == MK_COL_ROT ? cross(axis, point - anchor) : axis;
  }
  MK_GPU V3 jr(uint64_t mask) const { return (on(mask) && kind == MK_COL_ROT) ? axis : v3(0, 0, 0); }
};

struct Frame {
  V3 ps;
  Q4 Qs;
  M3 Rs;
  V3 bl, ba;   // this lane's body-frame Jacobian column (linear; angular)
};

MK_GPU void body_pose(const WLds& s, int body, V3& p, Q4& Q) {
  p = v3(s.bpos[3 * body], s.bpos[3 * body + 1], s.bpos[3 * body + 2]);
  Q = Q4{s.bquat[4 * body], s.bquat[4 * body + 1], s.bquat[4 * body + 2], s.bquat[4 * body + 3]};
}

MK_GPU Frame eval_frame(const MkDevModel& m, const WLds& s, int body, const double* lpos, const double* lquat,
                        const DofCol& dc) {
  V3 bp;
  Q4 bq;
  body_pose(s, body, bp, bq);
  const M3 R = q2m(bq);
  Frame f;
  f.ps = bp + mul(R, v3(lpos[0], lpos[1], lpos[2]));
  f.Qs = qmul(bq, Q4{lquat[0], lquat[1], lquat[2], lquat[3]});
  f.Rs = q2m(f.Qs);
  const uint64_t mask = m.body_dofmask[body];
  f.bl = mulT(f.Rs, dc.jp(mask, f.ps));
  f.ba = mulT(f.Rs, dc.jr(mask));
  return f;
}

// ---------------------------------------------------------------- kernels ----
#ifndef MK_DENSE_WPE
#define MK_DENSE_WPE 2
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MK_DENSE_WPE, MK_DENSE_WPE))) void dense_kernel(MkDevModel m, const MkDenseProblem* __restrict__ prob, int n_rows,
                                                   int n_general, int npath, const double* __restrict__ q,
                                                   const double* __restrict__ targets, int target_stride, double dt,
                                                   double damping, int B, double* __restrict__ P_out,
                                                   double* __restrict__ c_out, double* __restrict__ G_out,
                                                   double* __restrict__ h_out, double* __restrict__ v_out,
                                                   int32_t* status_out, int eval_task,
                                                   double* __restrict__ err_out, double* __restrict__ jac_out,
                                                   const int32_t* __restrict__ skip_flags, int accumulate_status,
                                                   int only_pending) {
  extern __shared__ double lds_base[];
  const int lane = threadIdx.x;
  const int n = m.nv, mrows = n_rows, mg = n_general;
  WLds s = carve(lds_base, n, mg, m.nq, target_stride, m.nbody);
  const int np = s.npad;
  const bool own = lane < n;
  // Problems of this wave: b = blockIdx.x, + gridDim.x, ...; in the last stage of a staged solve
  // only the ones still marked pending, found 64 status entries at a time (chunk c of the status
  // array belongs to block c mod gridDim.x).
  long next = blockIdx.x, chunk = (long)blockIdx.x * 64, chunk_cur = 0;
  uint64_t pend = 0ull;
  for (;;) {
    long b;
    if (only_pending) {
      while (pend == 0ull && chunk < B) {
        const long i = chunk + lane;
        pend = __ballot(i < B && (status_out[i] & MK_STATUS_PENDING) != 0);
        chunk_cur = chunk;
        chunk += (long)gridDim.x * 64;
      }
      if (pend == 0ull) break;
      b = chunk_cur + (__ffsll((unsigned long long)pend) - 1);
      pend &= pend - 1ull;
    } else {
      b = next;
      if (b >= B) break;
      next += gridDim.x;
    }
    // the iterated loop marks the problems that have left it: they are not touched again
    if (skip_flags && (skip_flags[b] & MK_ITER_DONE_BIT)) continue;
    // ---- inputs, kinematics ----
    bool bad_lane = false;
    for (int i = lane; i < m.nq; i += 64) {
      const double qv = q[b * m.nq + i];
      bad_lane |= !(fabs(qv) < INFINITY);
      s.q[i] = qv;
    }
    for (int i = lane; i < target_stride; i += 64) {
      const double tv = targets[b * target_stride + i];
      bad_lane |= !(fabs(tv) < INFINITY);
      s.tg[i] = tv;
    }
    const bool input_bad = __ballot(bad_lane) != 0ull;
    for (int idx = lane; idx < n * np; idx += 64) s.P[idx] = 0.0;
    if (own) { s.c[lane] = 0.0; s.dkind[lane] = MK_COL_LIN; }
    for (int i = lane; i < 3 * n; i += 64) { s.daxis[i] = 0.0; s.danch[i] = 0.0; }
    __syncthreads();
    if (own) s.P[lane * np + lane] = damping;
    DofCol dc;
    dc.lane = lane;
    dc.kind = MK_COL_LIN;
    dc.axis = v3(0, 0, 0);
    dc.anchor = v3(0, 0, 0);
    if (m.nbody <= 64) {
      tree_fk_levels(m, s, lane);
      if (own) {
        // pass C: the lane's dof, from the frame of its body's parent to the world
        const int tag = s.dkind[lane], par = tag >> 1;
        V3 pp;
        Q4 pq;
        body_pose(s, par, pp, pq);
        dc.kind = tag & 1;
        dc.axis = qrot(pq, v3(s.daxis[3 * lane], s.daxis[3 * lane + 1], s.daxis[3 * lane + 2]));
        dc.anchor = pp + qrot(pq, v3(s.danch[3 * lane], s.danch[3 * lane + 1], s.danch[3 * lane + 2]));
      }
    } else {
      tree_fk_walk(m, s, lane);
      if (own) {
        dc.kind = s.dkind[lane];
        dc.axis = v3(s.daxis[3 * lane], s.daxis[3 * lane + 1], s.daxis[3 * lane + 2]);
        dc.anchor = v3(s.danch[3 * lane], s.danch[3 * lane + 1], s.danch[3 * lane + 2]);
      }
    }
    const double* qb = s.q;
    const double* tg = s.tg;
#if defined(MK_DENSE_ABLATE) && MK_DENSE_ABLATE == 1
    // tuning build only: kinematics alone
    if (v_out && own) v_out[b * n + lane] = dc.axis.x + dc.anchor.y + s.bpos[3 * (lane % m.nbody)];
    __syncthreads();
    continue;
#endif

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
        Frame f = eval_frame(m, s, T.body, T.lpos, T.lquat, dc);
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
          Frame r = eval_frame(m, s, T.root_body, T.rpos, T.rquat, dc);
          const Q4 Qrf = qmul(qconj(r.Qs), f.Qs);
          const V3 trf = mulT(r.Rs, f.ps - r.ps);
          const M3 Rt = q2m(Qt);
          e = se3_log(qmul(qconj(Qt), Qrf), mulT(Rt, trf - pt));
          const M3 Rrf = q2m(Qrf);
          dl = f.bl - mulT(Rrf, r.bl - cross(trf, r.ba));
          da = f.ba - mulT(Rrf, r.ba);
          sign = 1.0;
        }
        // ljacinv of e (frame task) or of -e (relative frame task)
        Twist arg = sign < 0 ? e : Twist{neg(e.v), neg(e.w)};
        M3 Ji, Bm;
        se3_ljacinv_blocks(arg, Ji, Bm);   // I, 0 below the small-angle switch
        const V3 jl = mul(Ji, dl) + mul(Bm, da);
        const V3 ja = mul(Ji, da);
        Jraw[0] = sign * jl.x; Jraw[1] = sign * jl.y; Jraw[2] = sign * jl.z;
        Jraw[3] = sign * ja.x; Jraw[4] = sign * ja.y; Jraw[5] = sign * ja.z;
        eraw[0] = e.v.x; eraw[1] = e.v.y; eraw[2] = e.v.z; eraw[3] = e.w.x; eraw[4] = e.w.y; eraw[5] = e.w.z;
      } else if (T.type == MK_TASK_COM) {
        k = 3;
        V3 com = v3(0, 0, 0), acc = v3(0, 0, 0);
        double mass = 0.0;
        for (int bi = 0; bi < T.nbodies; ++bi) {
          const int body = T.bodies[bi];
          V3 bp;
          Q4 bq;
          body_pose(s, body, bp, bq);
          const V3 xi = bp + mul(q2m(bq), v3(m.body_ipos[3 * body], m.body_ipos[3 * body + 1], m.body_ipos[3 * body + 2]));
          const double mb = m.body_mass[body];
          com = com + mb * xi;
          acc = acc + mb * dc.jp(m.body_dofmask[body], xi);
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
        const int ballk = own ? m.dof_ball[lane] : -1;
        if (own && m.dof_scalar[lane]) {
          const int qa = m.dof_qadr[lane];
          cost = T.pcost[lane];
          e = T.ptarget[qa] - qb[qa];
        } else if (ballk >= 0) {
          // mj_differentiatePos for a ball joint: rotation vector of conj(q) * q_target (mju_subQuat)
          const int qa = m.dof_qadr[lane] - ballk;
          const Q4 d = qmul(Q4{qb[qa], -qb[qa + 1], -qb[qa + 2], -qb[qa + 3]},
                            Q4{T.ptarget[qa], T.ptarget[qa + 1], T.ptarget[qa + 2], T.ptarget[qa + 3]});
          double sn = sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
          V3 ax = v3(1, 0, 0);
          if (sn < MK_MINVAL) sn = 0.0; else ax = (1.0 / sn) * v3(d.x, d.y, d.z);
          double speed = 2.0 * atan2(sn, d.w);
          if (speed > MK_PI) speed -= 2.0 * MK_PI;
          cost = T.pcost[lane];
          e = speed * (ballk == 0 ? ax.x : (ballk == 1 ? ax.y : ax.z));
        }
        const double ewi = cost * (-T.gain * e);
        const double mu = T.lm * wave_sum(ewi * ewi);
        if (own) {
          s.P[lane * np + lane] += cost * cost + mu;
          s.c[lane] += ewi * cost;
        }
        if (eval_task == t && own) {
          if (err_out) err_out[b * n + lane] = e;
          if (jac_out)
            for (int j = 0; j < n; ++j)
              jac_out[(b * n + j) * n + lane] = (j == lane && (m.dof_scalar[lane] || ballk >= 0)) ? -1.0 : 0.0;
        }
        continue;
      }
      if (!own)
        for (int r = 0; r < 6; ++r) Jraw[r] = 0.0;
      double mu = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r) {   // rows >= k are zero
        const double cr = r < k ? T.cost[r] : 0.0;
        Jw[r] = cr * Jraw[r];
        ew[r] = cr * (-T.gain * eraw[r]);
        mu += ew[r] * ew[r];
      }
      mu *= T.lm;
      // row i of Jw'Jw on lane i: the weighted rows go through LDS
      __syncthreads();
      if (own) {
#pragma unroll
        for (int r = 0; r < 6; ++r) s.W[r * n + lane] = Jw[r];
      }
      __syncthreads();
      if (own) {
        double* prow = s.P + lane * np;
#pragma unroll 2
        for (int j = 0; j < n; ++j) {
          double acc = 0.0;
#pragma unroll
          for (int r = 0; r < 6; ++r) acc += Jw[r] * s.W[r * n + j];
          prow[j] += acc;
        }
        prow[lane] += mu;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r) acc += ew[r] * Jw[r];
        s.c[lane] -= acc;
      }
      if (eval_task == t) {
        if (err_out && lane < k) {
          double el = eraw[0];
#pragma unroll
          for (int r = 1; r < 6; ++r) el = (lane == r) ? eraw[r] : el;
          err_out[b * k + lane] = el;
        }
        if (jac_out && own) {
#pragma unroll
          for (int r = 0; r < 6; ++r)
            if (r < k) jac_out[(b * k + r) * n + lane] = Jraw[r];
        }
      }
    }
    __syncthreads();

    // ---- limits: two bounds per lane, general rows in LDS ----
    double lo = -INFINITY, hi = INFINITY;
    double* Gb = G_out ? G_out + b * (long)mrows * n : nullptr;
    double* hb = h_out ? h_out + b * (long)mrows : nullptr;
    const MkDenseDofBound* bounds = prob->dof_bound;
    if (bounds && !Gb && !hb) {
      // merged bounds of the lane's own dof
      if (own) {
        const MkDenseDofBound bd = bounds[lane];
        const double vb = dt * bd.vmax;
        hi = vb;
        lo = -vb;
        if (bd.has_cfg) {
          const double qv = qb[bd.qadr];
          hi = fmin(hi, bd.gain * (bd.upper - qv));
          lo = fmax(lo, -(bd.gain * (qv - bd.lower)));
        }
      }
    } else {
      for (int l = 0; l < prob->nlimit; ++l) {
        const MkDenseLimit& L = prob->limit[l];
        if (L.type == MK_LIMIT_CONFIGURATION) {
          for (int kk = 0; kk < L.n; ++kk) {
            const int j = L.index[kk];
            const int dof = m.jnt_dofadr[j], qa = m.jnt_qposadr[j];
            const double up = L.gain * (m.jnt_range[2 * j + 1] - L.min_dist - qb[qa]);
            const double dn = L.gain * (qb[qa] - (m.jnt_range[2 * j] + L.min_dist));
            if (lane == dof) { hi = fmin(hi, up); lo = fmax(lo, -dn); }
            if (Gb && own) {
              Gb[(long)(L.row0 + kk) * n + lane] = lane == dof ? 1.0 : 0.0;
              Gb[(long)(L.row0 + L.n + kk) * n + lane] = lane == dof ? -1.0 : 0.0;
            }
            if (hb && lane == 0) { hb[L.row0 + kk] = up; hb[L.row0 + L.n + kk] = dn; }
          }
        } else if (L.type == MK_LIMIT_VELOCITY) {
          for (int kk = 0; kk < L.n; ++kk) {
            const int dof = L.index[kk];
            const double vb = dt * L.vmax[kk];
            if (lane == dof) { hi = fmin(hi, vb); lo = fmax(lo, -vb); }
            if (Gb && own) {
              Gb[(long)(L.row0 + kk) * n + lane] = lane == dof ? 1.0 : 0.0;
              Gb[(long)(L.row0 + L.n + kk) * n + lane] = lane == dof ? -1.0 : 0.0;
            }
            if (hb && lane == 0) { hb[L.row0 + kk] = vb; hb[L.row0 + L.n + kk] = vb; }
          }
        }
      }
    }
    // collision avoidance: one general row per geom pair
    for (int kk = 0; kk < prob->npairs; ++kk) {
      const MkDensePair& C = prob->pairs[kk];
      V3 p1, p2;
      Q4 q1, q2;
      body_pose(s, C.b1, p1, q1);
      body_pose(s, C.b2, p2, q2);
      const V3 x1 = p1 + qrot(q1, v3(C.pos1[0], C.pos1[1], C.pos1[2]));
      const V3 x2 = p2 + qrot(q2, v3(C.pos2[0], C.pos2[1], C.pos2[2]));
      const V3 sz1 = v3(C.size1[0], C.size1[1], C.size1[2]), sz2 = v3(C.size2[0], C.size2[1], C.size2[2]);
      const M3 r1 = q2m(qmul(q1, Q4{C.quat1[0], C.quat1[1], C.quat1[2], C.quat1[3]}));
      // cheap lower bound of the distance from bounding spheres: most pairs are far apart
      const double rad2 = C.t2 == GEOM_SPHERE ? sz2.x : (C.t2 == GEOM_CAPSULE ? sz2.x + sz2.y : sqrt(dot(sz2, sz2)));
      double lower;
      if (C.t1 == GEOM_PLANE) {
        lower = dot(x2 - x1, col(r1, 2)) - rad2;
      } else {
        const double rad1 = C.t1 == GEOM_SPHERE ? sz1.x : (C.t1 == GEOM_CAPSULE ? sz1.x + sz1.y : sqrt(dot(sz1, sz1)));
        const V3 dd = x2 - x1;
        lower = sqrt(dot(dd, dd)) - rad1 - rad2;
      }
      Contact ct;
      bool active = false;
      if (C.convex) {
        // no closed form for this pair (convex mesh, cylinder, ellipsoid): support mappings
        const M3 r2 = q2m(qmul(q2, Q4{C.quat2[0], C.quat2[1], C.quat2[2], C.quat2[3]}));
        const V3 b2c = x2 + mul(r2, v3(C.bc2[0], C.bc2[1], C.bc2[2]));
        if (C.t1 == GEOM_PLANE) {
          lower = dot(b2c - x1, col(r1, 2)) - C.rb2;
        } else {
          const V3 dd = b2c - (x1 + mul(r1, v3(C.bc1[0], C.bc1[1], C.bc1[2])));
          lower = sqrt(dot(dd, dd)) - C.rb1 - C.rb2;
        }
        if (lower < C.d_detect) {
          CvxShape SB;
          SB.type = C.t2; SB.c = x2; SB.R = r2; SB.size = sz2;
          SB.verts = m.mesh_vert + 3 * (long)C.v2adr; SB.nvert = C.v2num;
          SB.mean = v3(C.mean2[0], C.mean2[1], C.mean2[2]); SB.ext = C.ext2;
          if (C.t1 == GEOM_PLANE) {
            cvx_plane_distance(lane, x1, col(r1, 2), SB, ct.dist, ct.p1, ct.p2);
          } else {
            CvxShape SA;
            SA.type = C.t1; SA.c = x1; SA.R = r1; SA.size = sz1;
            SA.verts = m.mesh_vert + 3 * (long)C.v1adr; SA.nvert = C.v1num;
            SA.mean = v3(C.mean1[0], C.mean1[1], C.mean1[2]); SA.ext = C.ext1;
            cvx_distance(lane, SA, SB, ct.dist, ct.p1, ct.p2);
          }
          active = ct.dist < C.d_detect;
        }
      } else if (lower < C.d_detect) {
        const M3 r2 = q2m(qmul(q2, Q4{C.quat2[0], C.quat2[1], C.quat2[2], C.quat2[3]}));
        active = geom_distance(lane, C.t1, x1, r1, sz1, C.t2, x2, r2, sz2, C.d_detect, ct);
      }
      double gval = 0.0, hval = INFINITY;
      if (active) {
        // witness points are (on geom1, on geom2) of the *stored* pair order
        V3 pa = ct.p1, pb = ct.p2;
        uint64_t ma = C.mask1, mb = C.mask2;
        if (C.swap) { pa = ct.p2; pb = ct.p1; ma = C.mask2; mb = C.mask1; }
        V3 dn = pb - pa;
        const double nn = sqrt(dot(dn, dn));
        const V3 nrm = nn < MK_MINVAL ? v3(1, 0, 0) : (1.0 / nn) * dn;
        const V3 dj = dc.jp(mb, pb) - dc.jp(ma, pa);
        gval = own ? -dot(nrm, dj) : 0.0;
        hval = ct.dist > C.d_min ? C.gain * (ct.dist - C.d_min) / dt + C.relax : C.relax;
      }
      if (own) s.Gg[kk * np + lane] = gval;
      if (lane == 0) s.hg[kk] = hval;
      if (Gb && own) Gb[(long)C.row * n + lane] = gval;
      if (hb && lane == 0) hb[C.row] = hval;
    }
    __syncthreads();

    // ---- outputs ----
    if (P_out) {
      for (int idx = lane; idx < n * n; idx += 64) P_out[b * n * n + idx] = s.P[(idx / n) * np + idx % n];
      if (own) c_out[b * n + lane] = s.c[lane];
    }
#if defined(MK_DENSE_ABLATE) && MK_DENSE_ABLATE == 2
    // tuning build only: kinematics + QP build, no solve
    if (v_out && own) v_out[b * n + lane] = s.P[lane * np + (lane + 1) % n] + s.c[lane] + lo + hi + (mg ? s.Gg[lane] : 0.0);
    __syncthreads();
    continue;
#endif
    if (v_out) {
      double x = 0.0;
      int st = gi_solve(s, n, mg, lane, lo, hi, x);
      // non-finite input shows up as a failed factorisation or a non-finite solution:
      // status 3 and a zero velocity for the whole problem
      if (input_bad || st == 3 || __ballot(own && !(fabs(x) < INFINITY)) != 0ull) {
        st = 3;
        x = 0.0;
      }
      if (own) v_out[b * n + lane] = x / dt;
      if (lane == 0) status_out[b] = accumulate_status ? max(status_out[b] & ~MK_STATUS_PENDING, st) : st;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void dense_qp_kernel(const double* __restrict__ P, const double* __restrict__ c,
                                                      const double* __restrict__ G, const double* __restrict__ h,
                                                      int n, int mrows, int B, double* __restrict__ x_out,
                                                      int32_t* __restrict__ status_out) {
  extern __shared__ double lds_base[];
  const int lane = threadIdx.x;
  WLds s = carve(lds_base, n, mrows, 0, 0, 0);
  const int np = s.npad;
  for (long b = blockIdx.x; b < B; b += gridDim.x) {
    for (int idx = lane; idx < n * n; idx += 64) s.P[(idx / n) * np + idx % n] = P[b * n * n + idx];
    if (lane < n) s.c[lane] = c[b * n + lane];
    for (int idx = lane; idx < mrows * n; idx += 64) s.Gg[(idx / n) * np + idx % n] = G[b * mrows * n + idx];
    for (int idx = lane; idx < mrows; idx += 64) s.hg[idx] = h[b * mrows + idx];
    __syncthreads();
    double x = 0.0;
    int st = gi_solve(s, n, mrows, lane, -INFINITY, INFINITY, x);
    // a factorisation that fails (P not positive definite, non-finite input) or a
    // non-finite solution: status 3 and x = 0
    if (st == 3 || __ballot(lane < n && !(fabs(x) < INFINITY)) != 0ull) {
      st = 3;
      x = 0.0;
    }
    if (lane < n) x_out[b * n + lane] = x;
    if (lane == 0) status_out[b] = st;
    __syncthreads();
  }
}

}  // namespace

size_t mk_dense_lds_bytes(int nv, int n_general, int nq, int target_stride, int nbody, int njnt, int npath) {
  if (nv > MK_DENSE_MAX_NV) return 0;
  (void)njnt; (void)npath;
  const size_t bytes = wave_lds_doubles(nv, n_general, nq, target_stride, nbody) * sizeof(double) + 64;
  return bytes <= 150 * 1024 ? bytes : 0;
}

// Blocks are persistent (each loops over problems), sized to fill the GPU a few times over.
static int dense_grid(int B, size_t lds) {
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
  return std::max(1, std::min(B, 256 * per_cu * 2));
}

hipError_t mk_dense_launch(const MkDevModel& m, const MkDenseProblem* d_problem, int n_rows, int n_general,
                           const double* q, const double* targets, int target_stride, double dt, double damping, int B,
                           double* P, double* c, double* G, double* h, double* v, int32_t* status, int eval_task,
                           double* err, double* jac, hipStream_t s, const int32_t* skip_flags, bool accumulate_status,
                           bool only_pending) {
  const size_t lds = mk_dense_lds_bytes(m.nv, n_general, m.nq, target_stride, m.nbody, m.njnt, m.npath);
  if (lds == 0) return hipErrorInvalidValue;
  static MkLdsGrant grant;
  if (hipError_t e = mk_grant_lds(reinterpret_cast<const void*>(dense_kernel), lds, grant); e != hipSuccess) return e;
  int grid = dense_grid(B, lds);
  // the last stage of a staged solve usually finds little or nothing pending: no more waves than
  // chunks of the status array to scan, and no more than the machine holds at once (a launch of
  // 4096 waves that all leave at once costs 20 us)
  if (only_pending) grid = std::max(1, std::min(grid, std::min((B + 63) / 64, 2048)));
  hipLaunchKernelGGL(dense_kernel, dim3(grid), dim3(64), lds, s, m, d_problem, n_rows, n_general, m.npath, q,
                     targets, target_stride, dt, damping, B, P, c, G, h, v, status, eval_task, err, jac, skip_flags,
                     accumulate_status ? 1 : 0, only_pending ? 1 : 0);
  return hipGetLastError();
}

hipError_t mk_dense_qp_launch(const double* P, const double* c, const double* G, const double* h, int n, int m, int B,
                              double* x, int32_t* status, hipStream_t s) {
  const size_t lds = mk_dense_lds_bytes(n, m, 0, 0, 0, 0, 0);
  if (lds == 0) return hipErrorInvalidValue;
  static MkLdsGrant grant;
  if (hipError_t e = mk_grant_lds(reinterpret_cast<const void*>(dense_qp_kernel), lds, grant); e != hipSuccess) return e;
  hipLaunchKernelGGL(dense_qp_kernel, dim3(dense_grid(B, lds)), dim3(64), lds, s, P, c, G, h, n, m, B, x, status);
  return hipGetLastError();
}
