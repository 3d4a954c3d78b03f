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
  ax = mk_rsqrt(n2) * ax;          // (seed + two Newton steps: a full-precision division here was a third of the axis test)
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

// ---- the other closed forms (geom types ordered t1 <= t2; MuJoCo sizes: sphere radius = size.x,
// capsule radius = size.x and half length along its z axis = size.y, box half extents) ----
MK_DEV void plane_sphere(V3 c1, const M3& r1, V3 c2, V3 s2, Contact& ct) {
  const V3 n = col(r1, 2);
  ct.dist = dot(c2 - c1, n) - s2.x;
  ct.p2 = c2 - s2.x * n;
  ct.p1 = ct.p2 - ct.dist * n;
}

// Two spheres of radii ra, rb about u and w (the last step of every core + radius pair).
MK_DEV void sphere_pair(V3 u, double ra, V3 w, double rb, Contact& ct) {
  const V3 d = w - u;
  double nrm = sqrt(dot(d, d));
  const V3 ax = nrm < MK_MINVAL ? v3(1, 0, 0) : (1.0 / nrm) * d;
  if (nrm < MK_MINVAL) nrm = 0.0;
  ct.dist = nrm - ra - rb;
  ct.p1 = u + ra * ax;
  ct.p2 = w - rb * ax;
}

MK_DEV void plane_capsule(V3 c1, const M3& r1, V3 c2, const M3& r2, V3 s2, Contact& ct) {
  // the lower of the two end spheres
  const V3 n = col(r1, 2), ax = col(r2, 2);
  const V3 ea = c2 + s2.y * ax, eb = c2 - s2.y * ax;
  const double da = dot(ea - c1, n) - s2.x, db = dot(eb - c1, n) - s2.x;
  const V3 e = da <= db ? ea : eb;
  ct.dist = da <= db ? da : db;
  ct.p2 = e - s2.x * n;
  ct.p1 = ct.p2 - ct.dist * n;
}

// Sphere (capsule = false) or capsule against a capsule: closest points of the axis segments
// (a sphere is a segment of length 0), then two spheres.
MK_DEV void round_capsule(bool capsule, V3 c1, const M3& r1, V3 s1, V3 c2, const M3& r2, V3 s2, Contact& ct) {
  const V3 a2 = col(r2, 2);
  V3 u = c1, w;
  if (!capsule) {
    const double t = fmin(fmax(dot(c1 - c2, a2), -s2.y), s2.y);
    w = c2 + t * a2;
  } else {
    const V3 a1 = col(r1, 2);
    seg_seg(c1 - s1.y * a1, c1 + s1.y * a1, c2 - s2.y * a2, c2 + s2.y * a2, u, w);
  }
  sphere_pair(u, s1.x, w, s2.x, ct);
}

// A point at `loc` (box frame) inside a box leaves through the nearest of the three faces on the
// side of `side` (box frame; the point itself for a sphere, the capsule's centre for the middle
// of the part of its axis inside the box -- that middle sits exactly on a mid-plane whenever the
// axis pierces two opposite faces, so its own side is not defined): contact of a ball of radius
// rad about the point; `centre` is the point in the world frame.
MK_DEV void inside_box(V3 loc, V3 side, V3 centre, double rad, V3 c2, const M3& r2, V3 s2, Contact& ct) {
  const V3 sg = v3(side.x >= 0.0 ? 1.0 : -1.0, side.y >= 0.0 ? 1.0 : -1.0, side.z >= 0.0 ? 1.0 : -1.0);
  const double gx = s2.x - sg.x * loc.x, gy = s2.y - sg.y * loc.y, gz = s2.z - sg.z * loc.z;
  const int k = (gx <= gy && gx <= gz) ? 0 : (gy <= gz ? 1 : 2);
  const double gk = k == 0 ? gx : (k == 1 ? gy : gz);
  const double sgn = comp(sg, k);
  const V3 ax = sgn * col(r2, k);
  V3 face = loc;
  if (k == 0) face.x = sgn * s2.x; else if (k == 1) face.y = sgn * s2.y; else face.z = sgn * s2.z;
  ct.dist = -gk - rad;
  ct.p2 = c2 + mul(r2, face);
  ct.p1 = centre - rad * ax;
}

MK_DEV void sphere_box(V3 c1, V3 s1, V3 c2, const M3& r2, V3 s2, Contact& ct) {
  const V3 loc = mulT(r2, c1 - c2);
  if (fabs(loc.x) > s2.x || fabs(loc.y) > s2.y || fabs(loc.z) > s2.z) {   // centre outside the box
    const V3 cl = v3(fmin(fmax(loc.x, -s2.x), s2.x), fmin(fmax(loc.y, -s2.y), s2.y), fmin(fmax(loc.z, -s2.z), s2.z));
    const V3 pb = c2 + mul(r2, cl);
    const V3 d = c1 - pb;
    double nrm = sqrt(dot(d, d));
    const V3 ax = nrm < MK_MINVAL ? v3(1, 0, 0) : (1.0 / nrm) * d;
    if (nrm < MK_MINVAL) nrm = 0.0;
    ct.dist = nrm - s1.x;
    ct.p1 = c1 - s1.x * ax;
    ct.p2 = pb;
  } else {
    inside_box(loc, loc, c1, s1.x, c2, r2, s2, ct);
  }
}

// Capsule against a box.  Candidate k of the closest points of the axis segment [a, b] and the
// box: 0, 1 the end points against the box, 2-13 the segment against the 12 box edges.
constexpr int MK_CAPSULE_BOX_CANDIDATES = 14;
MK_DEV void capsule_box_candidate(int k, V3 a, V3 b, V3 c2, const M3& r2, V3 s2, V3& u, V3& w) {
  if (k < 2) {
    u = k == 0 ? a : b;
    w = box_closest(u, c2, r2, s2);
  } else {
    V3 b0, b1;
    box_edge(c2, r2, s2, k - 2, b0, b1);
    seg_seg(a, b, b0, b1, u, w);
  }
}
// With the closest points (u on the axis segment, w on the box) found: a segment that passes
// through the box leaves it, from the middle of the part inside, through the nearest face on the
// capsule centre's side; otherwise a sphere about u against the point w.
MK_DEV void capsule_box_finish(V3 a, V3 b, V3 u, V3 w, V3 s1, V3 c2, const M3& r2, V3 s2, Contact& ct) {
  const V3 la = mulT(r2, a - c2), ld = mulT(r2, b - a);
  double t0 = 0.0, t1c = 1.0;
  for (int k = 0; k < 3; ++k) {
    const double lk = comp(la, k), dk = comp(ld, k), hk = comp(s2, k);
    if (fabs(dk) < 1e-300) {
      if (fabs(lk) > hk) { t0 = 1.0; t1c = 0.0; }
    } else {
      const double ta = (-hk - lk) / dk, tb = (hk - lk) / dk;
      t0 = fmax(t0, fmin(ta, tb));
      t1c = fmin(t1c, fmax(ta, tb));
    }
  }
  if (t0 <= t1c) {
    const V3 loc = la + (0.5 * (t0 + t1c)) * ld;
    inside_box(loc, la + 0.5 * ld, c2 + mul(r2, loc), s1.x, c2, r2, s2, ct);
  } else {
    const V3 d = w - u;
    double nrm = sqrt(dot(d, d));
    const V3 n = nrm < MK_MINVAL ? v3(1, 0, 0) : (1.0 / nrm) * d;
    if (nrm < MK_MINVAL) nrm = 0.0;
    ct.dist = nrm - s1.x;
    ct.p1 = u + s1.x * n;
    ct.p2 = w;
  }
}

// mj_geomDistance of a primitive pair (plane 0, sphere 2, capsule 3, box 6; t1 <= t2, not two
// planes) by one lane.  Returns false for "at least distmax apart".
MK_DEV bool geom_distance_lane(int t1, V3 c1, const M3& r1, V3 s1, int t2, V3 c2, const M3& r2, V3 s2, double distmax,
                               Contact& ct) {
  if (t1 == 0 && t2 == 6) {
    plane_box(c1, r1, c2, r2, s2, ct);
  } else if (t1 == 0 && t2 == 2) {
    plane_sphere(c1, r1, c2, s2, ct);
  } else if (t1 == 0 && t2 == 3) {
    plane_capsule(c1, r1, c2, r2, s2, ct);
  } else if (t1 == 2 && t2 == 2) {
    sphere_pair(c1, s1.x, c2, s2.x, ct);
  } else if ((t1 == 2 || t1 == 3) && t2 == 3) {
    round_capsule(t1 == 3, c1, r1, s1, c2, r2, s2, ct);
  } else if (t1 == 2 && t2 == 6) {
    sphere_box(c1, s1, c2, r2, s2, ct);
  } else if (t1 == 3 && t2 == 6) {
    const V3 ax = col(r1, 2);
    const V3 a = c1 - s1.y * ax, b = c1 + s1.y * ax;
    double best = INFINITY;
    V3 u = a, w = c2;
    for (int k = 0; k < MK_CAPSULE_BOX_CANDIDATES; ++k) {
      V3 uk, wk;
      capsule_box_candidate(k, a, b, c2, r2, s2, uk, wk);
      const double dd = dot(uk - wk, uk - wk);
      if (dd < best) { best = dd; u = uk; w = wk; }
    }
    capsule_box_finish(a, b, u, w, s1, c2, r2, s2, ct);
  } else {
    return box_box_lane(c1, r1, s1, c2, r2, s2, distmax, ct);
  }
  return ct.dist < distmax;
}
