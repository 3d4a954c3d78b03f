// Distance and penetration of two convex geoms by their support mappings, for the geom
// pairs mj_geomDistance (mink/limits/collision_avoidance_limit.py:214-229) has no closed
// form for: convex meshes (the hulls MuJoCo builds for collision meshes), cylinders,
// ellipsoids, and any of them against spheres, capsules and boxes.
//
// One pair at a time per wave: the simplex / portal logic is wave-uniform, the support
// mapping of a mesh is a wave-wide arg-max over its hull vertices.  Separated pairs get the
// exact Euclidean distance by the Gilbert-Johnson-Keerthi iteration (closest point of the
// simplex by Voronoi regions); overlapping pairs depth, direction and position by Minkowski
// portal refinement.  Spheres and capsules enter as a point / segment core plus a radius.
// (MuJoCo 3.2.6 hands such pairs to libccd with tolerance 1e-6; what is computed here is the
// geometry itself.)
#pragma once
#include "mk_device.h"

#define MK_CVX __device__ __forceinline__

constexpr int CVX_SPHERE = 2, CVX_CAPSULE = 3, CVX_ELLIPSOID = 4, CVX_CYLINDER = 5, CVX_BOX = 6, CVX_MESH = 7;
constexpr int CVX_GJK_ITERATIONS = 128;
constexpr double CVX_GJK_REL = 1e-13;
constexpr double CVX_GJK_TOUCH = 1e-13;
constexpr int CVX_MPR_ITERATIONS = 128;
constexpr double CVX_MPR_TOL = 1e-11;

struct CvxShape {
  int type;
  V3 c;                 // geom frame origin in the world
  M3 R;                 // geom frame orientation
  V3 size;
  const double* verts;  // mesh: hull vertices in the geom frame [nvert * 3]
  int nvert;
  V3 mean;              // mesh: mean of the hull vertices (an interior point), geom frame
  double ext;           // largest coordinate magnitude (scale of the shape)
};

// implemented by the including file: wave-wide arg-max with ties to the smaller id
MK_CVX void cvx_wave_argmax(double& key, int& id);

MK_CVX double cvx_radius(const CvxShape& s) { return (s.type == CVX_SPHERE || s.type == CVX_CAPSULE) ? s.size.x : 0.0; }
MK_CVX V3 cvx_centre(const CvxShape& s) { return s.type == CVX_MESH ? s.c + mul(s.R, s.mean) : s.c; }

// Farthest point of the shape's core along d (world frame).
MK_CVX V3 cvx_support(int lane, const CvxShape& s, V3 d) {
  const V3 dl = mulT(s.R, d);
  V3 loc = v3(0, 0, 0);
  if (s.type == CVX_CAPSULE) {
    loc = v3(0.0, 0.0, dl.z >= 0.0 ? s.size.y : -s.size.y);
  } else if (s.type == CVX_BOX) {
    loc = v3(dl.x >= 0.0 ? s.size.x : -s.size.x, dl.y >= 0.0 ? s.size.y : -s.size.y, dl.z >= 0.0 ? s.size.z : -s.size.z);
  } else if (s.type == CVX_CYLINDER) {
    const double n = sqrt(dl.x * dl.x + dl.y * dl.y);
    const double k = n > 1e-300 ? s.size.x / n : 0.0;
    loc = v3(k * dl.x, k * dl.y, dl.z >= 0.0 ? s.size.y : -s.size.y);
  } else if (s.type == CVX_ELLIPSOID) {
    const V3 sd = v3(s.size.x * dl.x, s.size.y * dl.y, s.size.z * dl.z);
    const double n = sqrt(dot(sd, sd));
    loc = n > 1e-300 ? v3(s.size.x * sd.x / n, s.size.y * sd.y / n, s.size.z * sd.z / n) : v3(s.size.x, 0.0, 0.0);
  } else if (s.type == CVX_MESH) {
    double best = -INFINITY;
    int bi = -1;
    for (int i = lane; i < s.nvert; i += 64) {
      const double k = s.verts[3 * i] * dl.x + s.verts[3 * i + 1] * dl.y + s.verts[3 * i + 2] * dl.z;
      if (k > best) { best = k; bi = i; }
    }
    cvx_wave_argmax(best, bi);
    bi = bi < 0 ? 0 : bi;
    loc = v3(s.verts[3 * bi], s.verts[3 * bi + 1], s.verts[3 * bi + 2]);
  }
  return s.c + mul(s.R, loc);
}

struct Bary3 {
  double a, b, c;
};

// Barycentric coordinates of the point of triangle abc closest to the origin (Voronoi regions).
MK_CVX Bary3 cvx_closest_triangle(V3 a, V3 b, V3 c) {
  const V3 ab = b - a, ac = c - a;
  const double d1 = -dot(ab, a), d2 = -dot(ac, a);
  if (d1 <= 0.0 && d2 <= 0.0) return Bary3{1.0, 0.0, 0.0};
  const double d3 = -dot(ab, b), d4 = -dot(ac, b);
  if (d3 >= 0.0 && d4 <= d3) return Bary3{0.0, 1.0, 0.0};
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
    const double v = d1 / (d1 - d3);
    return Bary3{1.0 - v, v, 0.0};
  }
  const double d5 = -dot(ab, c), d6 = -dot(ac, c);
  if (d6 >= 0.0 && d5 <= d6) return Bary3{0.0, 0.0, 1.0};
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
    const double w = d2 / (d2 - d6);
    return Bary3{1.0 - w, 0.0, w};
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
    const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    return Bary3{0.0, 1.0 - w, w};
  }
  const double den = 1.0 / (va + vb + vc);
  const double v = vb * den, w = vc * den;
  return Bary3{1.0 - v - w, v, w};
}

struct CvxSimplex {
  V3 w[4], a[4], b[4];
  double lam[4];
  int n;
};

// lam of the simplex point closest to the origin; true when the origin is inside a tetrahedron.
MK_CVX bool cvx_closest_simplex(CvxSimplex& S) {
  if (S.n == 1) {
    S.lam[0] = 1.0;
    return false;
  }
  if (S.n == 2) {
    const V3 ab = S.w[1] - S.w[0];
    const double den = dot(ab, ab);
    const double t = den > 0.0 ? -dot(S.w[0], ab) / den : 0.0;
    if (t <= 0.0) { S.lam[0] = 1.0; S.lam[1] = 0.0; }
    else if (t >= 1.0) { S.lam[0] = 0.0; S.lam[1] = 1.0; }
    else { S.lam[0] = 1.0 - t; S.lam[1] = t; }
    return false;
  }
  if (S.n == 3) {
    const Bary3 l = cvx_closest_triangle(S.w[0], S.w[1], S.w[2]);
    S.lam[0] = l.a; S.lam[1] = l.b; S.lam[2] = l.c;
    return false;
  }
  // tetrahedron: the faces the origin lies outside of, closest one wins (first on ties).
  // Support points of two polytopes are often coplanar (four vertices of one face of A - B): the
  // side tests of such a flat tetrahedron are decided by rounding, and "outside of no face" would
  // be read as an overlap.  A tetrahedron whose fourth vertex lies within 1e-9 (sine of the
  // angle) of a face's plane is therefore never "inside": all its faces are candidates.
  bool flat = false;
#define CVX_FLAT(I, J, K, L)                                                                          \
  {                                                                                                   \
    const V3 fa = S.w[I], fd = S.w[L];                                                                \
    const V3 nrm = cross(S.w[J] - fa, S.w[K] - fa);                                                   \
    const double sd = dot(fd - fa, nrm);                                                              \
    flat |= sd * sd <= 1e-18 * dot(nrm, nrm) * dot(fd - fa, fd - fa);                                  \
  }
  CVX_FLAT(0, 1, 2, 3)
  CVX_FLAT(0, 2, 3, 1)
  CVX_FLAT(0, 3, 1, 2)
  CVX_FLAT(1, 3, 2, 0)
#undef CVX_FLAT
  bool found = false;
  double best = INFINITY;
  double lam[4] = {0.25, 0.25, 0.25, 0.25};
#define CVX_FACE(I, J, K, L)                                                          \
  {                                                                                   \
    const V3 fa = S.w[I], fb = S.w[J], fc = S.w[K], fd = S.w[L];                      \
    const V3 nrm = cross(fb - fa, fc - fa);                                           \
    const double so = -dot(fa, nrm), sd = dot(fd - fa, nrm);                          \
    if (flat || so * sd < 0.0 || sd == 0.0) {                                         \
      const Bary3 l = cvx_closest_triangle(fa, fb, fc);                               \
      const V3 p = l.a * fa + l.b * fb + l.c * fc;                                    \
      const double d2 = dot(p, p);                                                    \
      if (d2 < best) {                                                                \
        best = d2;                                                                    \
        found = true;                                                                 \
        lam[0] = lam[1] = lam[2] = lam[3] = 0.0;                                      \
        lam[I] = l.a; lam[J] = l.b; lam[K] = l.c;                                     \
      }                                                                               \
    }                                                                                 \
  }
  CVX_FACE(0, 1, 2, 3)
  CVX_FACE(0, 2, 3, 1)
  CVX_FACE(0, 3, 1, 2)
  CVX_FACE(1, 3, 2, 0)
#undef CVX_FACE
#pragma unroll
  for (int i = 0; i < 4; ++i) S.lam[i] = lam[i];
  return !found;
}

// Keeps the vertices with lam > 0, in order.
MK_CVX void cvx_reduce(CvxSimplex& S) {
  int j = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < S.n && S.lam[i] > 0.0) {
#pragma unroll
      for (int jj = 0; jj <= i; ++jj) {
        if (jj == j) {
          S.w[jj] = S.w[i]; S.a[jj] = S.a[i]; S.b[jj] = S.b[i]; S.lam[jj] = S.lam[i];
        }
      }
      ++j;
    }
  }
  S.n = j;
}

MK_CVX V3 cvx_combine(const V3 (&p)[4], const double (&lam)[4], int n) {
  V3 r = v3(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < n) r = r + lam[i] * p[i];
  return r;
}

// Distance of the cores; returns true when they overlap (dist, pa, pb then unset).
MK_CVX bool cvx_gjk(int lane, const CvxShape& A, const CvxShape& B, double& dist, V3& pa, V3& pb) {
  V3 v = cvx_centre(A) - cvx_centre(B);
  if (dot(v, v) < 1e-30) v = v3(1.0, 0.0, 0.0);
  const double scale = fmax(fmax(A.ext, B.ext), 1e-12);
  CvxSimplex S;
  S.n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { S.w[i] = S.a[i] = S.b[i] = v3(0, 0, 0); S.lam[i] = 0.0; }
  S.lam[0] = 1.0;
  for (int it = 0; it < CVX_GJK_ITERATIONS; ++it) {
    const V3 a = cvx_support(lane, A, neg(v)), b = cvx_support(lane, B, v);
    const V3 w = a - b;
    const double vv = dot(v, v);
    if (S.n > 0 && vv - dot(v, w) <= CVX_GJK_REL * vv) break;
    bool dup = false;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < S.n && S.w[i].x == w.x && S.w[i].y == w.y && S.w[i].z == w.z) dup = true;
    if (dup) break;
    const CvxSimplex S_old = S;
    const V3 v_old = v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i == S.n) { S.w[i] = w; S.a[i] = a; S.b[i] = b; }
    ++S.n;
    const bool inside = cvx_closest_simplex(S);
    cvx_reduce(S);
    v = cvx_combine(S.w, S.lam, S.n);
    if (inside || dot(v, v) <= (CVX_GJK_TOUCH * scale) * (CVX_GJK_TOUCH * scale)) return true;
    // The new simplex contains the old one, so its closest point cannot be farther.  When the new
    // support point adds nothing but rounding (a needle triangle next to the converged edge) the
    // region tests may say otherwise, and the iteration would alternate between the two states for
    // good: the old state is the answer.
    if (S_old.n > 0 && !(dot(v, v) < vv)) {
      S = S_old;
      v = v_old;
      break;
    }
  }
  pa = cvx_combine(S.a, S.lam, S.n);
  pb = cvx_combine(S.b, S.lam, S.n);
  dist = sqrt(dot(v, v));
  return false;
}

MK_CVX V3 cvx_unit(V3 x) {
  const double n = sqrt(dot(x, x));
  return n > 0.0 ? (1.0 / n) * x : x;
}

// Penetration of overlapping cores by Minkowski portal refinement on A - B: depth, the
// direction from A into B, and a contact position.
MK_CVX void cvx_mpr(int lane, const CvxShape& A, const CvxShape& B, double& depth, V3& n, V3& pos) {
#define CVX_SUP(D, V, PA, PB)              \
  {                                        \
    PA = cvx_support(lane, A, D);          \
    PB = cvx_support(lane, B, neg(D));     \
    V = PA - PB;                           \
  }
  V3 v0 = cvx_centre(A) - cvx_centre(B);
  if (dot(v0, v0) < 1e-24) v0 = v3(1e-5, 0.0, 0.0);
  V3 d = cvx_unit(neg(v0));
  V3 v1, a1, b1, v2, a2, b2, v3_, a3, b3, v4, a4, b4;
  CVX_SUP(d, v1, a1, b1)
  d = cross(v0, v1);
  if (dot(d, d) < 1e-30) {
    n = cvx_unit(neg(v0));
    depth = dot(v1, n);
    pos = 0.5 * (a1 + b1);
    return;
  }
  d = cvx_unit(d);
  CVX_SUP(d, v2, a2, b2)
  d = cvx_unit(cross(v1 - v0, v2 - v0));
  if (dot(d, v0) > 0.0) {
    V3 t;
    t = v1; v1 = v2; v2 = t;
    t = a1; a1 = a2; a2 = t;
    t = b1; b1 = b2; b2 = t;
    d = neg(d);
  }
  v3_ = v2; a3 = a2; b3 = b2;
  for (int it = 0; it < CVX_MPR_ITERATIONS; ++it) {   // a portal the origin ray passes through
    CVX_SUP(d, v3_, a3, b3)
    if (dot(cross(v1, v3_), v0) < 0.0) { v2 = v3_; a2 = a3; b2 = b3; }
    else if (dot(cross(v3_, v2), v0) < 0.0) { v1 = v3_; a1 = a3; b1 = b3; }
    else break;
    d = cvx_unit(cross(v1 - v0, v2 - v0));
  }
  for (int it = 0; it < CVX_MPR_ITERATIONS; ++it) {   // push the portal to the surface
    d = cvx_unit(cross(v2 - v1, v3_ - v1));
    CVX_SUP(d, v4, a4, b4)
    const double s4 = dot(v4, d);
    const double reach = fmin(fmin(s4 - dot(v1, d), s4 - dot(v2, d)), s4 - dot(v3_, d));
    if (reach <= CVX_MPR_TOL * fmax(1.0, fabs(s4))) break;
    const V3 x = cross(v4, v0);
    if (dot(v1, x) > 0.0) {
      if (dot(v2, x) > 0.0) { v1 = v4; a1 = a4; b1 = b4; } else { v3_ = v4; a3 = a4; b3 = b4; }
    } else {
      if (dot(v3_, x) > 0.0) { v2 = v4; a2 = a4; b2 = b4; } else { v1 = v4; a1 = a4; b1 = b4; }
    }
  }
#undef CVX_SUP
  // The portal now lies in the face of A - B that the ray from the interior point through the
  // origin leaves by: depth and direction of that face's plane, position at the foot of the
  // origin on it (properties of the face, not of the triangle the refinement ended with).
  n = cvx_unit(cross(v2 - v1, v3_ - v1));
  depth = dot(v1, n);
  const V3 p = depth * n;
  const V3 e0 = v2 - v1, e1 = v3_ - v1, e2 = p - v1;
  const double d00 = dot(e0, e0), d01 = dot(e0, e1), d11 = dot(e1, e1), d20 = dot(e2, e0), d21 = dot(e2, e1);
  const double den = d00 * d11 - d01 * d01;
  // a portal that has collapsed to a segment or a point (three support points in line) has no
  // barycentric coordinates: the foot of the origin on its longer edge instead (never NaN rows)
  double lv = 0.0, lw = 0.0;
  if (den > 1e-14 * d00 * d11) {
    lv = (d11 * d20 - d01 * d21) / den;
    lw = (d00 * d21 - d01 * d20) / den;
  } else if (d00 >= d11 && d00 > 0.0) {
    lv = fmin(fmax(d20 / d00, 0.0), 1.0);
  } else if (d11 > 0.0) {
    lw = fmin(fmax(d21 / d11, 0.0), 1.0);
  }
  const double lu = 1.0 - lv - lw;
  const V3 pa = lu * a1 + lv * a2 + lw * a3, pb = lu * b1 + lv * b2 + lw * b3;
  pos = 0.5 * (pa + pb);
}

// Signed distance and witness points (on A, on B), in MuJoCo's fromto convention: for a
// penetration the two points are pos -/+ dist/2 * normal.
// (not inlined: the register needs of the iteration stay out of the caller's allocation)
__device__ __attribute__((noinline)) void cvx_distance(int lane, const CvxShape& A, const CvxShape& B, double& dist,
                                                       V3& p1, V3& p2) {
  const double ra = cvx_radius(A), rb = cvx_radius(B);
  double d = 0.0;
  V3 pa = v3(0, 0, 0), pb = v3(0, 0, 0);
  const bool hit = cvx_gjk(lane, A, B, d, pa, pb);
  if (!hit && d > 0.0) {
    const V3 n = (1.0 / d) * (pb - pa);
    dist = d - ra - rb;
    p1 = pa + ra * n;
    p2 = pb - rb * n;
    return;
  }
  double depth;
  V3 n, pos;
  cvx_mpr(lane, A, B, depth, n, pos);
  dist = -depth - ra - rb;
  p1 = pos - (0.5 * dist) * n;
  p2 = pos + (0.5 * dist) * n;
}

// Plane (normal = z axis of its frame, through pc) against a convex shape.
__device__ __attribute__((noinline)) void cvx_plane_distance(int lane, V3 pc, V3 nrm, const CvxShape& S, double& dist,
                                                             V3& p1, V3& p2) {
  const V3 p = cvx_support(lane, S, neg(nrm)) - cvx_radius(S) * nrm;
  dist = dot(p - pc, nrm);
  p1 = p - dist * nrm;
  p2 = p;
}
