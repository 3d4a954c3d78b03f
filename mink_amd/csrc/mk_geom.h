// Closed-form distance pieces of the primitive geom pairs behind CollisionAvoidanceLimit
// (mujoco.mj_geomDistance as called in mink/limits/collision_avoidance_limit.py:214-229),
// shared by the wave-per-problem kernel (candidates spread over the lanes) and the
// lane-per-problem kernels (candidates in a loop).  Both number their candidates the same
// way and take the first strict minimum, so they agree on the witness points.
#pragma once
#include "mk_device.h"

struct Contact {
  double dist;
  V3 p1, p2;
};

MK_DEV V3 box_closest(V3 p, V3 c, const M3& r, V3 h) {
  V3 loc = mulT(r, p - c);
  loc.x = fmin(fmax(loc.x, -h.x), h.x);
  loc.y = fmin(fmax(loc.y, -h.y), h.y);
  loc.z = fmin(fmax(loc.z, -h.z), h.z);
  return c + mul(r, loc);
}

// Closest points of the segments [p1, q1] and [p2, q2].
MK_DEV void seg_seg(V3 p1, V3 q1, V3 p2, V3 q2, V3& c1, V3& c2) {
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

// Column i of a rotation matrix / component i of a vector, by selects (i may be a loop
// variable: indexing a register array with it would move the array to scratch memory).
MK_DEV V3 col(const M3& r, int i) {
  return i == 0 ? v3(r.m[0], r.m[3], r.m[6]) : (i == 1 ? v3(r.m[1], r.m[4], r.m[7]) : v3(r.m[2], r.m[5], r.m[8]));
}
MK_DEV double comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

MK_DEV V3 box_vertex(V3 c, const M3& r, V3 h, int k) {
  // k = 4*bx + 2*by + bz with sign -1 for bit 0, +1 for bit 1
  const V3 s = v3((k & 4) ? h.x : -h.x, (k & 2) ? h.y : -h.y, (k & 1) ? h.z : -h.z);
  return c + mul(r, s);
}
MK_DEV void box_edge(V3 c, const M3& r, V3 h, int e, V3& a0, V3& a1) {
  // e = 4*a + 2*sb + sc over axis a and the signs of the two other axes b=(a+1)%3, c=(a+2)%3
  const int a = e >> 2, b = a == 2 ? 0 : a + 1;
  const double sb = (e & 2) ? 1.0 : -1.0, sc = (e & 1) ? 1.0 : -1.0;
  // component j of the two end points: -h / +h along the edge axis, the fixed signed extent otherwise
  const double fx = (b == 0 ? sb : sc) * h.x, fy = (b == 1 ? sb : sc) * h.y, fz = (b == 2 ? sb : sc) * h.z;
  const V3 lo = v3(a == 0 ? -h.x : fx, a == 1 ? -h.y : fy, a == 2 ? -h.z : fz);
  const V3 hi = v3(a == 0 ? h.x : fx, a == 1 ? h.y : fy, a == 2 ? h.z : fz);
  a0 = c + mul(r, lo);
  a1 = c + mul(r, hi);
}

// Candidate separating axis k of two boxes (0-2: axes of the first, 3-5: of the second,
// 6-14: their cross products): overlap of the projections along it (negative: a gap), and
// the unit axis pointing from box 1 to box 2.  Returns false for a degenerate cross product.
MK_DEV bool box_box_axis(int k, V3 d, const M3& r1, V3 h1, const M3& r2, V3 h2, double& overlap, V3& axis) {
  V3 ax;
  if (k < 3) ax = col(r1, k);
  else if (k < 6) ax = col(r2, k - 3);
  else ax = cross(col(r1, (k - 6) / 3), col(r2, (k - 6) % 3));
  const double n2 = dot(ax, ax);
  if (!(n2 >= 1e-12)) return false;
  ax = (1.0 / sqrt(n2)) * ax;
  const V3 a1 = mulT(r1, ax), a2 = mulT(r2, ax);
  const double ra = fabs(a1.x) * h1.x + fabs(a1.y) * h1.y + fabs(a1.z) * h1.z;
  const double rb = fabs(a2.x) * h2.x + fabs(a2.y) * h2.y + fabs(a2.z) * h2.z;
  const double proj = dot(d, ax);
  overlap = ra + rb - fabs(proj);
  axis = proj >= 0.0 ? ax : neg(ax);
  return true;
}

// Overlapping boxes: penetration = smallest overlap, along axis n (from box 1 to box 2);
// witnesses around the midpoint of the two support features.  A component of n that vanishes
// in a box frame selects the middle of that extent (face / edge centre), not an arbitrary vertex.
MK_DEV Contact box_box_penetration(V3 c1, const M3& r1, V3 h1, V3 c2, const M3& r2, V3 h2, double overlap, V3 n) {
  const V3 a1 = mulT(r1, n), a2 = mulT(r2, n);
  auto sg = [](double a) { return fabs(a) < 1e-9 ? 0.0 : (a > 0.0 ? 1.0 : -1.0); };
  const V3 s1 = c1 + mul(r1, v3(sg(a1.x) * h1.x, sg(a1.y) * h1.y, sg(a1.z) * h1.z));
  const V3 s2 = c2 + mul(r2, v3(-sg(a2.x) * h2.x, -sg(a2.y) * h2.y, -sg(a2.z) * h2.z));
  const V3 pos = 0.5 * (s1 + s2);
  Contact ct;
  ct.dist = -overlap;
  ct.p1 = pos - (0.5 * ct.dist) * n;
  ct.p2 = pos + (0.5 * ct.dist) * n;
  return ct;
}

// Closest points of two separated boxes are searched over 17 candidates: the 8 + 8 vertices of
// one box against the other box (they cover every closest pair that involves a vertex), and one
// edge pair.  A closest pair in the interior of two edges is perpendicular to both, so its
// direction is the cross product of the two edge directions, the gap along that axis equals the
// distance -- the largest of all the separating-axis gaps -- and each edge is the support set of
// its box along that direction.  The one edge pair is therefore the pair of support edges of the
// cross-product axis with the largest gap.
constexpr int MK_BOX_CANDIDATES = 17;

// Support edges along unit axis n (pointing from box 1 to box 2), n = +-(axis i of box 1) x (axis j of box 2).
MK_DEV void box_box_support_edges(int i, int j, V3 n, V3 c1, const M3& r1, V3 h1, V3 c2, const M3& r2, V3 h2,
                                  V3& a0, V3& a1, V3& b0, V3& b1) {
  const V3 l1 = mulT(r1, n), l2 = mulT(r2, n);
  const int ib = i == 2 ? 0 : i + 1, ic = ib == 2 ? 0 : ib + 1;
  const int jb = j == 2 ? 0 : j + 1, jc = jb == 2 ? 0 : jb + 1;
  // box 1: the edge farthest along +n; box 2: the edge farthest along -n
  const int e1 = 4 * i + (comp(l1, ib) >= 0.0 ? 2 : 0) + (comp(l1, ic) >= 0.0 ? 1 : 0);
  const int e2 = 4 * j + (comp(l2, jb) >= 0.0 ? 0 : 2) + (comp(l2, jc) >= 0.0 ? 0 : 1);
  box_edge(c1, r1, h1, e1, a0, a1);
  box_edge(c2, r2, h2, e2, b0, b1);
}

// Candidate cnd: 0-7 vertices of box 1 against box 2, 8-15 vertices of box 2 against box 1,
// 16 the support edges of cross-product axis kx = 6 + 3 i + j with unit direction nx (kx < 0:
// every cross product degenerate, no edge candidate: the pair is left at infinity).
MK_DEV void box_box_candidate(int cnd, int kx, V3 nx, V3 c1, const M3& r1, V3 h1, V3 c2, const M3& r2, V3 h2, V3& u,
                              V3& w) {
  if (cnd < 8) {
    u = box_vertex(c1, r1, h1, cnd);
    w = box_closest(u, c2, r2, h2);
  } else if (cnd < 16) {
    w = box_vertex(c2, r2, h2, cnd - 8);
    u = box_closest(w, c1, r1, h1);
  } else if (kx >= 6) {
    V3 a0, a1, b0, b1;
    box_box_support_edges((kx - 6) / 3, (kx - 6) % 3, nx, c1, r1, h1, c2, r2, h2, a0, a1, b0, b1);
    seg_seg(a0, a1, b0, b1, u, w);
  } else {
    u = v3(INFINITY, INFINITY, INFINITY);
    w = v3(0, 0, 0);
  }
}

// Signed distance and witness points of two oriented boxes by one lane.  Returns false when
// the pair is at least distmax apart (ct is then not meaningful).
MK_DEV bool box_box_lane(V3 c1, const M3& r1, V3 h1, V3 c2, const M3& r2, V3 h2, double distmax, Contact& ct) {
  const V3 d = c2 - c1;
  double least = INFINITY, gap = -INFINITY, xgap = -INFINITY;
  V3 axis = v3(1, 0, 0), nx = v3(1, 0, 0);
  int kx = -1;
  for (int k = 0; k < 15; ++k) {
    double overlap;
    V3 ax;
    if (box_box_axis(k, d, r1, h1, r2, h2, overlap, ax)) {
      if (overlap < least) { least = overlap; axis = ax; }
      gap = fmax(gap, -overlap);
      if (k >= 6 && -overlap > xgap) { xgap = -overlap; kx = k; nx = ax; }
    }
  }
  if (!(gap > 0.0)) {   // no separating axis among the candidates: overlapping
    ct = box_box_penetration(c1, r1, h1, c2, r2, h2, least, axis);
    return ct.dist < distmax;
  }
  if (gap >= distmax) return false;
  double best = INFINITY;
  V3 p1 = c1, p2 = c2;
  for (int cnd = 0; cnd < MK_BOX_CANDIDATES; ++cnd) {
    V3 u, w;
    box_box_candidate(cnd, kx, nx, c1, r1, h1, c2, r2, h2, u, w);
    const double dd = dot(u - w, u - w);
    if (dd < best) { best = dd; p1 = u; p2 = w; }
  }
  ct.dist = sqrt(best);
  ct.p1 = p1;
  ct.p2 = p2;
  return ct.dist < distmax;
}

// Plane (normal = z axis of its frame) against a box: the lowest vertex, first strict minimum.
MK_DEV void plane_box(V3 c1, const M3& r1, V3 c2, const M3& r2, V3 s2, Contact& ct) {
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
}
