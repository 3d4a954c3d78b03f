"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the convex distance / penetration
queries behind mj_geomDistance for geom pairs without a closed form (convex meshes,
cylinders, ellipsoids, and any of them against the primitives).

PARITY UNPINNED.  The reference calls mujoco.mj_geomDistance
(mink/limits/collision_avoidance_limit.py:214-229); MuJoCo 3.2.6 answers such pairs with
libccd on the convex hulls of the meshes (support mapping per geom type, margin inflated
objects, penetration by Minkowski portal refinement, tolerance 1e-6).  Neither MuJoCo nor
libccd is in the image, and the reference holds no fixture of such a distance.  What is
restated here is the geometry itself: the exact Euclidean distance of the two convex sets by
the Gilbert-Johnson-Keerthi iteration (closest point of a simplex by Voronoi regions), and,
when they overlap, depth / direction / position by Minkowski portal refinement (the scheme of
libccd's ccdMPRPenetration), reported for the plane of the face the portal converges to.  Distances of separated pairs agree with any correct
routine to its tolerance; witness points on parallel features and penetration depths may
differ from MuJoCo's.

Shapes are dicts: type (MuJoCo mjtGeom), pos [3], mat [3,3], size [3], verts [V,3] (mesh,
in the geom frame).  Spheres and capsules are handled as a point / segment core plus a
radius, so their distances are exact.
"""
import numpy as np

PLANE, SPHERE, CAPSULE, ELLIPSOID, CYLINDER, BOX, MESH = 0, 2, 3, 4, 5, 6, 7
GJK_ITERATIONS = 128
GJK_REL = 1e-13          # no-progress test, relative to |v|^2
GJK_TOUCH = 1e-13        # |v| below this (relative to the shapes' size) counts as contact
MPR_ITERATIONS = 128
MPR_TOL = 1e-11


def radius(s):
    return float(s["size"][0]) if s["type"] in (SPHERE, CAPSULE) else 0.0


def centre(s):
    """An interior point of the shape (world frame)."""
    if s["type"] == MESH:
        return s["pos"] + s["mat"] @ s["verts"].mean(axis=0)
    return np.asarray(s["pos"], dtype=float)


def extent(s):
    if s["type"] == MESH:
        return float(np.abs(s["verts"]).max())
    return float(np.abs(s["size"]).max())


def support(s, d):
    """Farthest point of the shape's core along d (world frame)."""
    R = s["mat"]
    dl = R.T @ d
    t, sz = s["type"], s["size"]
    if t == SPHERE:
        loc = np.zeros(3)
    elif t == CAPSULE:
        loc = np.array([0.0, 0.0, sz[1] if dl[2] >= 0 else -sz[1]])
    elif t == BOX:
        loc = np.where(dl >= 0, sz, -sz)
    elif t == CYLINDER:
        n = np.hypot(dl[0], dl[1])
        xy = sz[0] * dl[:2] / n if n > 1e-300 else np.zeros(2)
        loc = np.array([xy[0], xy[1], sz[1] if dl[2] >= 0 else -sz[1]])
    elif t == ELLIPSOID:
        n = np.linalg.norm(sz * dl)
        loc = sz * sz * dl / n if n > 1e-300 else np.array([sz[0], 0.0, 0.0])
    elif t == MESH:
        loc = s["verts"][int(np.argmax(s["verts"] @ dl))]      # first maximum
    else:
        raise ValueError(f"no support mapping for geom type {t}")
    return s["pos"] + R @ loc


# ---------------------------------------------------------------- simplex ----
def _closest_segment(a, b):
    ab = b - a
    den = ab @ ab
    t = -(a @ ab) / den if den > 0 else 0.0
    if t <= 0.0:
        return [1.0, 0.0]
    if t >= 1.0:
        return [0.0, 1.0]
    return [1.0 - t, t]


def _closest_triangle(a, b, c):
    """Barycentric coordinates of the point of triangle abc closest to the origin
    (Voronoi-region tests)."""
    ab, ac = b - a, c - a
    d1, d2 = -(ab @ a), -(ac @ a)
    if d1 <= 0.0 and d2 <= 0.0:
        return [1.0, 0.0, 0.0]
    d3, d4 = -(ab @ b), -(ac @ b)
    if d3 >= 0.0 and d4 <= d3:
        return [0.0, 1.0, 0.0]
    vc = d1 * d4 - d3 * d2
    if vc <= 0.0 and d1 >= 0.0 and d3 <= 0.0:
        v = d1 / (d1 - d3)
        return [1.0 - v, v, 0.0]
    d5, d6 = -(ab @ c), -(ac @ c)
    if d6 >= 0.0 and d5 <= d6:
        return [0.0, 0.0, 1.0]
    vb = d5 * d2 - d1 * d6
    if vb <= 0.0 and d2 >= 0.0 and d6 <= 0.0:
        w = d2 / (d2 - d6)
        return [1.0 - w, 0.0, w]
    va = d3 * d6 - d5 * d4
    if va <= 0.0 and (d4 - d3) >= 0.0 and (d5 - d6) >= 0.0:
        w = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        return [0.0, 1.0 - w, w]
    den = 1.0 / (va + vb + vc)
    v, w = vb * den, vc * den
    return [1.0 - v - w, v, w]


_TET_FACES = ((0, 1, 2, 3), (0, 2, 3, 1), (0, 3, 1, 2), (1, 3, 2, 0))


def _closest_simplex(W):
    """(lambda, inside) for the simplex with vertices W (1..4 points) and the origin."""
    n = len(W)
    if n == 1:
        return [1.0], False
    if n == 2:
        return _closest_segment(W[0], W[1]), False
    if n == 3:
        return _closest_triangle(W[0], W[1], W[2]), False
    # a flat tetrahedron (four support points of one face of A - B) has side tests decided by
    # rounding: it is never "inside", all its faces are candidates (as in mk_convex.h)
    flat = False
    for (i, j, k, l) in _TET_FACES:
        a, d = W[i], W[l]
        nrm = np.cross(W[j] - a, W[k] - a)
        sd = (d - a) @ nrm
        flat |= bool(sd * sd <= 1e-18 * (nrm @ nrm) * ((d - a) @ (d - a)))
    best, best_d2 = None, np.inf
    for (i, j, k, l) in _TET_FACES:
        a, b, c, d = W[i], W[j], W[k], W[l]
        nrm = np.cross(b - a, c - a)
        so, sd = -(a @ nrm), (d - a) @ nrm
        if flat or so * sd < 0.0 or sd == 0.0:          # the origin is on the outer side of this face
            lam3 = _closest_triangle(a, b, c)
            p = lam3[0] * a + lam3[1] * b + lam3[2] * c
            d2 = p @ p
            if d2 < best_d2:
                lam = [0.0, 0.0, 0.0, 0.0]
                lam[i], lam[j], lam[k] = lam3
                best, best_d2 = lam, d2
    if best is None:
        return [0.25, 0.25, 0.25, 0.25], True
    return best, False


def gjk(A, B):
    """Distance of the cores of A and B.  Returns (dist, pA, pB, intersecting); when the
    cores overlap dist is 0 and the points are meaningless."""
    v = centre(A) - centre(B)
    if v @ v < 1e-30:
        v = np.array([1.0, 0.0, 0.0])
    scale = max(extent(A), extent(B), 1e-12)
    W, PA, PB = [], [], []
    lam = [1.0]
    for _ in range(GJK_ITERATIONS):
        a, b = support(A, -v), support(B, v)
        w = a - b
        vv = v @ v
        if W and vv - v @ w <= GJK_REL * vv:
            break
        if any(np.array_equal(w, x) for x in W):
            break
        old = (list(W), list(PA), list(PB), list(lam), v)
        W.append(w); PA.append(a); PB.append(b)
        lam, inside = _closest_simplex(W)
        keep = [i for i, l in enumerate(lam) if l > 0.0]
        W, PA, PB = [W[i] for i in keep], [PA[i] for i in keep], [PB[i] for i in keep]
        lam = [lam[i] for i in keep]
        v = sum(l * x for l, x in zip(lam, W))
        if inside or v @ v <= (GJK_TOUCH * scale) ** 2:
            return 0.0, None, None, True
        # the new simplex contains the old one: a closest point that is not nearer is rounding in the
        # region tests of a needle triangle, and the old state is the answer (as in mk_convex.h)
        if old[0] and not (v @ v < vv):
            W, PA, PB, lam, v = old
            break
    pa = sum(l * x for l, x in zip(lam, PA))
    pb = sum(l * x for l, x in zip(lam, PB))
    return float(np.sqrt(v @ v)), pa, pb, False


# -------------------------------------------------------------------- MPR ----
def mpr(A, B):
    """Penetration of the overlapping cores of A and B by Minkowski portal refinement:
    (depth, direction from A into B, position), on the Minkowski difference A - B (the ray
    from its interior point c_A - c_B through the origin leaves it in the direction from A to B)."""
    def sup(d):
        a, b = support(A, d), support(B, -d)
        return a - b, a, b

    def unit(x):
        n = np.linalg.norm(x)
        return x / n if n > 0 else x

    v0 = centre(A) - centre(B)
    if v0 @ v0 < 1e-24:
        v0 = np.array([1e-5, 0.0, 0.0])
    d = unit(-v0)
    v1, a1, b1 = sup(d)
    d = np.cross(v0, v1)
    if d @ d < 1e-30:
        # the origin is on the ray through the centre: the first support point decides
        n = unit(-v0)
        return float(v1 @ n), n, 0.5 * (a1 + b1)
    d = unit(d)
    v2, a2, b2 = sup(d)
    d = unit(np.cross(v1 - v0, v2 - v0))
    if d @ v0 > 0.0:
        v1, v2, a1, a2, b1, b2 = v2, v1, a2, a1, b2, b1
        d = -d
    for _ in range(MPR_ITERATIONS):                 # find a portal the origin ray passes through
        v3, a3, b3 = sup(d)
        if np.cross(v1, v3) @ v0 < 0.0:
            v2, a2, b2 = v3, a3, b3
        elif np.cross(v3, v2) @ v0 < 0.0:
            v1, a1, b1 = v3, a3, b3
        else:
            break
        d = unit(np.cross(v1 - v0, v2 - v0))
    for _ in range(MPR_ITERATIONS):                 # push the portal to the surface
        d = unit(np.cross(v2 - v1, v3 - v1))
        v4, a4, b4 = sup(d)
        reach = min(v4 @ d - v1 @ d, v4 @ d - v2 @ d, v4 @ d - v3 @ d)
        if reach <= MPR_TOL * max(1.0, abs(v4 @ d)):
            break
        x = np.cross(v4, v0)
        if v1 @ x > 0.0:
            if v2 @ x > 0.0:
                v1, a1, b1 = v4, a4, b4
            else:
                v3, a3, b3 = v4, a4, b4
        else:
            if v3 @ x > 0.0:
                v2, a2, b2 = v4, a4, b4
            else:
                v1, a1, b1 = v4, a4, b4
    # The portal now lies in the face of A - B that the ray from the interior point through the
    # origin leaves by.  Depth and direction are those of the face's plane, the position is the
    # foot of the origin on it: all three are properties of that face, not of the particular
    # triangle the refinement ended with.
    n = unit(np.cross(v2 - v1, v3 - v1))
    depth = float(v1 @ n)
    p = depth * n
    e0, e1, e2 = v2 - v1, v3 - v1, p - v1
    d00, d01, d11, d20, d21 = e0 @ e0, e0 @ e1, e1 @ e1, e2 @ e0, e2 @ e1
    den = d00 * d11 - d01 * d01
    lv = lw = 0.0
    if den > 1e-14 * d00 * d11:
        lv = (d11 * d20 - d01 * d21) / den
        lw = (d00 * d21 - d01 * d20) / den
    elif d00 >= d11 and d00 > 0.0:          # collapsed portal: foot on its longer edge
        lv = min(max(d20 / d00, 0.0), 1.0)
    elif d11 > 0.0:
        lw = min(max(d21 / d11, 0.0), 1.0)
    lam = [1.0 - lv - lw, lv, lw]
    pa = lam[0] * a1 + lam[1] * a2 + lam[2] * a3
    pb = lam[0] * b1 + lam[1] * b2 + lam[2] * b3
    return depth, n, 0.5 * (pa + pb)


def distance(A, B):
    """Signed distance of A and B and the witness points (on A, on B) in MuJoCo's fromto
    convention: for a penetration the two points are pos -/+ dist/2 * normal."""
    ra, rb = radius(A), radius(B)
    d, pa, pb, hit = gjk(A, B)
    if not hit and d > 0.0:
        n = (pb - pa) / d
        return d - ra - rb, pa + ra * n, pb - rb * n
    depth, n, pos = mpr(A, B)
    dist = -depth - ra - rb
    return dist, pos - 0.5 * dist * n, pos + 0.5 * dist * n


def plane_distance(P, S):
    """Plane (normal = z axis of its frame) against any supported convex shape."""
    n = P["mat"][:, 2]
    p = support(S, -n) - radius(S) * n
    dist = float((p - P["pos"]) @ n)
    return dist, p - dist * n, p
