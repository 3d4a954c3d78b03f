"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the MuJoCo routines the mink
hot path calls.  Nothing under mink_amd/ may import this module.

MuJoCo (pinned ``mujoco==3.2.6`` in the reference's mac_env.yml:153) is a
third-party dependency that is absent from /root/reference and from this image,
so the routines below restate its *published* kinematics algorithms from the
documentation of ``mj_kinematics``, ``mj_comPos``, ``mj_jac``,
``mj_jacSubtreeCom``, ``mj_integratePos``, ``mj_differentiatePos`` and the
``mju_*`` quaternion helpers.  Call sites in the reference:
mink/configuration.py:63-64,145,225,235; mink/lie/so3.py:83,113,150;
mink/tasks/posture_task.py:107; mink/tasks/com_task.py:96;
mink/limits/configuration_limit.py:100,110;
mink/limits/collision_avoidance_limit.py:49,69,71,219.

PARITY UNPINNED against MuJoCo itself: no MuJoCo binary or golden output exists
here.  These functions are pinned by the reference's own oracle-free checks
instead (task Jacobian == finite difference of task error through
mj_integratePos, tests/test_jacobians.py:41-108; hinge-only integrate == q+v*dt,
tests/test_configuration.py:85-100) and by hand-computed poses.
"""

from __future__ import annotations

import math

import numpy as np

JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_BOX, GEOM_MESH = 0, 2, 3, 6, 7
MINVAL = 1e-15
MAXVAL = 1e10


# ----------------------------------------------------------------- mju_* ----
def mju_mulQuat(a, b):
    """Hamilton product of two (w, x, y, z) quaternions."""
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
    ])


def mju_quat2Mat(q):
    """3x3 rotation matrix of a unit quaternion (no normalisation)."""
    q00, q11, q22, q33 = q[0] * q[0], q[1] * q[1], q[2] * q[2], q[3] * q[3]
    q01, q02, q03 = q[0] * q[1], q[0] * q[2], q[0] * q[3]
    q12, q13, q23 = q[1] * q[2], q[1] * q[3], q[2] * q[3]
    return np.array([
        [q00 + q11 - q22 - q33, 2 * (q12 - q03), 2 * (q13 + q02)],
        [2 * (q12 + q03), q00 - q11 + q22 - q33, 2 * (q23 - q01)],
        [2 * (q13 - q02), 2 * (q23 + q01), q00 - q11 - q22 + q33],
    ])


def mju_normalize4(q):
    n = math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    if n < MINVAL:
        return np.array([1.0, 0.0, 0.0, 0.0])
    return np.asarray(q, dtype=np.float64) / n


def mju_normalize3(v):
    """Returns (unit vector, norm); a vanishing vector maps to (1, 0, 0)."""
    n = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    if n < MINVAL:
        return np.array([1.0, 0.0, 0.0]), 0.0
    return np.asarray(v, dtype=np.float64) / n, n


def mju_mat2Quat(m):
    """Quaternion of a rotation matrix (row-major 9-vector or 3x3); the sign of
    w is not canonicalised."""
    m = np.asarray(m, dtype=np.float64).reshape(9)
    q = np.zeros(4)
    if m[0] + m[4] + m[8] > 0:
        q[0] = 0.5 * math.sqrt(1 + m[0] + m[4] + m[8])
        q[1] = 0.25 * (m[7] - m[5]) / q[0]
        q[2] = 0.25 * (m[2] - m[6]) / q[0]
        q[3] = 0.25 * (m[3] - m[1]) / q[0]
    elif m[0] > m[4] and m[0] > m[8]:
        q[1] = 0.5 * math.sqrt(1 + m[0] - m[4] - m[8])
        q[0] = 0.25 * (m[7] - m[5]) / q[1]
        q[2] = 0.25 * (m[1] + m[3]) / q[1]
        q[3] = 0.25 * (m[2] + m[6]) / q[1]
    elif m[4] > m[8]:
        q[2] = 0.5 * math.sqrt(1 - m[0] + m[4] - m[8])
        q[0] = 0.25 * (m[2] - m[6]) / q[2]
        q[1] = 0.25 * (m[1] + m[3]) / q[2]
        q[3] = 0.25 * (m[5] + m[7]) / q[2]
    else:
        q[3] = 0.5 * math.sqrt(1 - m[0] - m[4] + m[8])
        q[0] = 0.25 * (m[3] - m[1]) / q[3]
        q[1] = 0.25 * (m[2] + m[6]) / q[3]
        q[2] = 0.25 * (m[5] + m[7]) / q[3]
    return mju_normalize4(q)


def mju_axisAngle2Quat(axis, angle):
    if angle == 0.0:
        return np.array([1.0, 0.0, 0.0, 0.0])
    s = math.sin(0.5 * angle)
    return np.array([math.cos(0.5 * angle), s * axis[0], s * axis[1], s * axis[2]])


def mju_rotVecQuat(v, q):
    """Rotate vector ``v`` by unit quaternion ``q``."""
    return mju_quat2Mat(q) @ np.asarray(v, dtype=np.float64)


def mju_quatIntegrate(quat, omega, scale):
    """quat <- normalize(quat) * axisangle(omega/|omega|, scale*|omega|); omega is
    expressed in the rotating (body) frame."""
    axis, n = mju_normalize3(omega)
    qrot = mju_axisAngle2Quat(axis, scale * n)
    return mju_mulQuat(mju_normalize4(quat), qrot)


def mju_quat2Vel(q, dt):
    axis, s = mju_normalize3(q[1:4])
    speed = 2.0 * math.atan2(s, q[0])
    if speed > math.pi:
        speed -= 2.0 * math.pi
    return axis * (speed / dt)


def mju_subQuat(qa, qb):
    """Rotation vector taking qb to qa, in qb's frame: quat2Vel(conj(qb)*qa)."""
    conj = np.array([qb[0], -qb[1], -qb[2], -qb[3]])
    return mju_quat2Vel(mju_mulQuat(conj, qa), 1.0)


# ------------------------------------------------------------------ data ----
class Data:
    """Per-configuration kinematic state; field names mirror ``mujoco.MjData``."""

    def __init__(self, model):
        m = model
        self.model = m
        self.qpos = m.qpos0.copy()
        self.xpos = np.zeros((m.nbody, 3))
        self.xquat = np.tile([1.0, 0.0, 0.0, 0.0], (m.nbody, 1))
        self.xmat = np.tile(np.eye(3).reshape(9), (m.nbody, 1))
        self.xipos = np.zeros((m.nbody, 3))
        self.xanchor = np.zeros((m.njnt, 3))
        self.xaxis = np.zeros((m.njnt, 3))
        self.site_xpos = np.zeros((m.nsite, 3))
        self.site_xmat = np.zeros((m.nsite, 9))
        self.geom_xpos = np.zeros((m.ngeom, 3))
        self.geom_xmat = np.zeros((m.ngeom, 9))
        self.subtree_com = np.zeros((m.nbody, 3))
        nm = max(m.nmocap, 0)
        self.mocap_pos = np.zeros((nm, 3))
        self.mocap_quat = np.tile([1.0, 0.0, 0.0, 0.0], (nm, 1))
        for b in range(m.nbody):
            k = m.body_mocapid[b]
            if k >= 0:
                self.mocap_pos[k] = m.body_pos[b]
                self.mocap_quat[k] = m.body_quat[b]


# ----------------------------------------------------------- kinematics ----
def mj_kinematics(m, d):
    """Forward kinematics of every body, site and geom."""
    d.xpos[0] = 0.0
    d.xquat[0] = [1.0, 0.0, 0.0, 0.0]
    d.xmat[0] = np.eye(3).reshape(9)
    d.xipos[0] = 0.0
    for b in range(1, m.nbody):
        mid = m.body_mocapid[b]
        jadr, jnum = m.body_jntadr[b], m.body_jntnum[b]
        if mid >= 0:
            xpos = d.mocap_pos[mid].copy()
            xquat = mju_normalize4(d.mocap_quat[mid])
        elif jnum == 1 and m.jnt_type[jadr] == JNT_FREE:
            a = m.jnt_qposadr[jadr]
            xpos = d.qpos[a:a + 3].copy()
            xquat = mju_normalize4(d.qpos[a + 3:a + 7])
            d.xanchor[jadr] = xpos
            d.xaxis[jadr] = m.jnt_axis[jadr]
        else:
            p = m.body_parentid[b]
            xpos = d.xmat[p].reshape(3, 3) @ m.body_pos[b] + d.xpos[p]
            xquat = mju_mulQuat(d.xquat[p], m.body_quat[b])
            for j in range(jadr, jadr + jnum):
                a = m.jnt_qposadr[j]
                jt = m.jnt_type[j]
                xaxis = mju_rotVecQuat(m.jnt_axis[j], xquat)
                xanchor = mju_rotVecQuat(m.jnt_pos[j], xquat) + xpos
                d.xaxis[j] = xaxis
                d.xanchor[j] = xanchor
                if jt == JNT_SLIDE:
                    xpos = xpos + xaxis * (d.qpos[a] - m.qpos0[a])
                elif jt in (JNT_BALL, JNT_HINGE):
                    if jt == JNT_BALL:
                        qloc = mju_normalize4(d.qpos[a:a + 4])
                    else:
                        qloc = mju_axisAngle2Quat(m.jnt_axis[j], d.qpos[a] - m.qpos0[a])
                    xquat = mju_mulQuat(xquat, qloc)
                    xpos = xanchor - mju_rotVecQuat(m.jnt_pos[j], xquat)
                else:
                    raise ValueError("free joint must be the only joint of its body")
        xquat = mju_normalize4(xquat)
        d.xpos[b] = xpos
        d.xquat[b] = xquat
        xmat = mju_quat2Mat(xquat)
        d.xmat[b] = xmat.reshape(9)
        d.xipos[b] = xpos + xmat @ m.body_ipos[b]
    for s in range(m.nsite):
        b = m.site_bodyid[s]
        d.site_xpos[s] = d.xmat[b].reshape(3, 3) @ m.site_pos[s] + d.xpos[b]
        d.site_xmat[s] = mju_quat2Mat(mju_mulQuat(d.xquat[b], m.site_quat[s])).reshape(9)
    for g in range(m.ngeom):
        b = m.geom_bodyid[g]
        d.geom_xpos[g] = d.xmat[b].reshape(3, 3) @ m.geom_pos[g] + d.xpos[b]
        d.geom_xmat[g] = mju_quat2Mat(mju_mulQuat(d.xquat[b], m.geom_quat[g])).reshape(9)


def subtree_mass(m):
    tot = m.body_mass.astype(np.float64).copy()
    for b in range(m.nbody - 1, 0, -1):
        tot[m.body_parentid[b]] += tot[b]
    return tot


def mj_comPos(m, d):
    """Subtree centres of mass (mass-weighted xipos accumulated child -> parent)."""
    acc = d.xipos * m.body_mass[:, None]
    for b in range(m.nbody - 1, 0, -1):
        acc[m.body_parentid[b]] += acc[b]
    tot = subtree_mass(m)
    for b in range(m.nbody):
        d.subtree_com[b] = acc[b] / tot[b] if tot[b] >= MINVAL else d.xipos[b]


def mj_jac(m, d, point, body):
    """World-aligned translational and rotational Jacobians (3 x nv each) of a
    world-frame ``point`` rigidly attached to ``body``."""
    jacp = np.zeros((3, m.nv))
    jacr = np.zeros((3, m.nv))
    while body and m.body_dofnum[body] == 0:
        body = m.body_parentid[body]
    if body == 0:
        return jacp, jacr
    i = m.body_dofadr[body] + m.body_dofnum[body] - 1
    point = np.asarray(point, dtype=np.float64)
    while i >= 0:
        j = m.dof_jntid[i]
        jt = m.jnt_type[j]
        k = i - m.jnt_dofadr[j]
        if jt == JNT_SLIDE:
            jacp[:, i] = d.xaxis[j]
        elif jt == JNT_HINGE:
            jacr[:, i] = d.xaxis[j]
            jacp[:, i] = np.cross(d.xaxis[j], point - d.xanchor[j])
        elif jt == JNT_FREE and k < 3:
            jacp[k, i] = 1.0
        else:
            # rotational dofs of free / ball joints act about the body's own axes
            kk = k - 3 if jt == JNT_FREE else k
            axis = d.xmat[m.jnt_bodyid[j]].reshape(3, 3)[:, kk]
            jacr[:, i] = axis
            jacp[:, i] = np.cross(axis, point - d.xanchor[j])
        i = m.dof_parentid[i]
    return jacp, jacr


def mj_jacBody(m, d, body):
    return mj_jac(m, d, d.xpos[body], body)


def mj_jacSite(m, d, site):
    return mj_jac(m, d, d.site_xpos[site], m.site_bodyid[site])


def mj_jacGeom(m, d, geom):
    return mj_jac(m, d, d.geom_xpos[geom], m.geom_bodyid[geom])


def mj_jacSubtreeCom(m, d, body):
    """Translational Jacobian of the centre of mass of the subtree rooted at body."""
    jac = np.zeros((3, m.nv))
    tot = 0.0
    for b in range(body, m.nbody):
        # b is in the subtree iff walking up its parents reaches `body`
        p = b
        while p > body:
            p = m.body_parentid[p]
        if p != body:
            continue
        jp, _ = mj_jac(m, d, d.xipos[b], b)
        jac += m.body_mass[b] * jp
        tot += m.body_mass[b]
    return jac / tot if tot >= MINVAL else jac


def mj_integratePos(m, qpos, qvel, dt):
    """In-place qpos <- qpos (+) dt*qvel on the configuration manifold."""
    for j in range(m.njnt):
        a, v = m.jnt_qposadr[j], m.jnt_dofadr[j]
        jt = m.jnt_type[j]
        if jt == JNT_FREE:
            qpos[a:a + 3] += dt * qvel[v:v + 3]
            qpos[a + 3:a + 7] = mju_quatIntegrate(qpos[a + 3:a + 7], qvel[v + 3:v + 6], dt)
        elif jt == JNT_BALL:
            qpos[a:a + 4] = mju_quatIntegrate(qpos[a:a + 4], qvel[v:v + 3], dt)
        else:
            qpos[a] += dt * qvel[v]


def mj_differentiatePos(m, dt, qpos1, qpos2):
    """Returns qvel = (qpos2 (-) qpos1) / dt in tangent coordinates."""
    qvel = np.zeros(m.nv)
    for j in range(m.njnt):
        a, v = m.jnt_qposadr[j], m.jnt_dofadr[j]
        jt = m.jnt_type[j]
        if jt == JNT_FREE:
            qvel[v:v + 3] = (qpos2[a:a + 3] - qpos1[a:a + 3]) / dt
            qvel[v + 3:v + 6] = mju_subQuat(qpos2[a + 3:a + 7], qpos1[a + 3:a + 7]) / dt
        elif jt == JNT_BALL:
            qvel[v:v + 3] = mju_subQuat(qpos2[a:a + 4], qpos1[a:a + 4]) / dt
        else:
            qvel[v] = (qpos2[a] - qpos1[a]) / dt
    return qvel


# --------------------------------------------------------- geom distance ----
def _box_support_points(center, rot, half):
    signs = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    return center + (signs * half) @ rot.T


def _point_box(p, c, r, h):
    """Closest point on the box (centre c, axes r, half sizes h) to p."""
    loc = r.T @ (p - c)
    return c + r @ np.clip(loc, -h, h)


def _seg_seg(p1, q1, p2, q2):
    """Closest points of segments [p1, q1] and [p2, q2] (clamped parametric solution)."""
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    a, e, f = float(d1 @ d1), float(d2 @ d2), float(d2 @ r)
    eps = 1e-30
    if a <= eps and e <= eps:
        s = t = 0.0
    elif a <= eps:
        s, t = 0.0, min(max(f / e, 0.0), 1.0)
    else:
        c = float(d1 @ r)
        if e <= eps:
            t, s = 0.0, min(max(-c / a, 0.0), 1.0)
        else:
            b = float(d1 @ d2)
            den = a * e - b * b
            s = min(max((b * f - c * e) / den, 0.0), 1.0) if den > eps else 0.0
            t = (b * s + f) / e
            if t < 0.0:
                t, s = 0.0, min(max(-c / a, 0.0), 1.0)
            elif t > 1.0:
                t, s = 1.0, min(max((b - c) / a, 0.0), 1.0)
    return p1 + s * d1, p2 + t * d2


def _box_vertices(c, r, h):
    return [c + r @ (np.array([sx, sy, sz]) * h) for sx in (-1.0, 1.0) for sy in (-1.0, 1.0) for sz in (-1.0, 1.0)]


def _box_edges(c, r, h):
    edges = []
    for a in range(3):
        b, cc = (a + 1) % 3, (a + 2) % 3
        for sb in (-1.0, 1.0):
            for sc in (-1.0, 1.0):
                off = np.zeros(3)
                off[b], off[cc] = sb * h[b], sc * h[cc]
                lo, hi = off.copy(), off.copy()
                lo[a], hi[a] = -h[a], h[a]
                edges.append((c + r @ lo, c + r @ hi))
    return edges


def _closest_box_box(c1, r1, h1, c2, r2, h2):
    """Signed distance and witness points of two oriented boxes.

    Overlapping boxes (no separating axis among the 15 candidates): dist = -(smallest
    overlap), along that axis n (pointing from box 1 to box 2), with the witness points
    pos -/+ dist/2 * n around the midpoint of the two support points.  Separated boxes:
    exact Euclidean closest points, by enumeration of vertex-box and edge-edge pairs
    in a fixed order (first strict minimum wins)."""
    d = c2 - c1
    axes = [r1[:, i] for i in range(3)] + [r2[:, i] for i in range(3)]
    for i in range(3):
        for j in range(3):
            axes.append(np.cross(r1[:, i], r2[:, j]))
    best_overlap, best_axis = np.inf, None
    separated = False
    for ax in axes:
        n2 = float(ax @ ax)
        if n2 < 1e-12:
            continue
        ax = ax / math.sqrt(n2)
        ra = float(np.abs(r1.T @ ax) @ h1)
        rb = float(np.abs(r2.T @ ax) @ h2)
        overlap = ra + rb - abs(float(d @ ax))
        if overlap < 0.0:
            separated = True
            break
        if overlap < best_overlap:
            best_overlap, best_axis = overlap, (ax if float(d @ ax) >= 0.0 else -ax)
    if not separated:
        n = best_axis
        # support features along n: a component of n that vanishes in a box frame selects
        # the middle of that extent (face / edge centre), not an arbitrary vertex
        a1, a2 = r1.T @ n, r2.T @ n
        s1 = c1 + r1 @ (np.where(np.abs(a1) < 1e-9, 0.0, np.sign(a1)) * h1)
        s2 = c2 + r2 @ (np.where(np.abs(a2) < 1e-9, 0.0, -np.sign(a2)) * h2)
        pos = 0.5 * (s1 + s2)
        dist = -best_overlap
        return dist, pos - 0.5 * dist * n, pos + 0.5 * dist * n
    best, p1, p2 = np.inf, None, None
    for v in _box_vertices(c1, r1, h1):
        w = _point_box(v, c2, r2, h2)
        dd = float((v - w) @ (v - w))
        if dd < best:
            best, p1, p2 = dd, v, w
    for v in _box_vertices(c2, r2, h2):
        w = _point_box(v, c1, r1, h1)
        dd = float((v - w) @ (v - w))
        if dd < best:
            best, p1, p2 = dd, w, v
    e2 = _box_edges(c2, r2, h2)
    for (a0, a1) in _box_edges(c1, r1, h1):
        for (b0, b1) in e2:
            u, w = _seg_seg(a0, a1, b0, b1)
            dd = float((u - w) @ (u - w))
            if dd < best:
                best, p1, p2 = dd, u, w
    return math.sqrt(best), p1, p2


def _capsule_box(c1, r1, s1, c2, r2, h2):
    """Signed distance and witness points of a capsule (radius s1[0], half length s1[1]
    along its z axis) and an oriented box.

    Axis segment outside the box: exact closest points of the segment and the box -- the two
    end points against the box, then the segment against the 12 box edges, in this fixed
    order, first strict minimum wins -- and dist = that distance - radius.  Axis segment
    through the box: the point in the middle of the part inside the box leaves through the
    nearest of the three faces on the capsule centre's side (for a sphere centre inside a box
    this is the nearest face), dist = -(gap + radius)."""
    ax = r1[:, 2]
    a, b = c1 - ax * s1[1], c1 + ax * s1[1]
    best, u, w = np.inf, None, None
    for e in (a, b):
        pb = _point_box(e, c2, r2, h2)
        dd = float((e - pb) @ (e - pb))
        if dd < best:
            best, u, w = dd, e, pb
    for (b0, b1) in _box_edges(c2, r2, h2):
        x, y = _seg_seg(a, b, b0, b1)
        dd = float((x - y) @ (x - y))
        if dd < best:
            best, u, w = dd, x, y
    # does the segment pass through the box? clip it against the three slabs
    la, ld = r2.T @ (a - c2), r2.T @ (b - a)
    t0, t1 = 0.0, 1.0
    for k in range(3):
        if abs(ld[k]) < 1e-300:
            if abs(la[k]) > h2[k]:
                t0, t1 = 1.0, 0.0
                break
        else:
            ta, tb = (-h2[k] - la[k]) / ld[k], (h2[k] - la[k]) / ld[k]
            t0, t1 = max(t0, min(ta, tb)), min(t1, max(ta, tb))
    if t0 <= t1:
        # (the side of each axis is the one the capsule's centre lies on: the middle of the part
        # inside sits exactly on a mid-plane whenever the segment pierces two opposite faces)
        loc = la + 0.5 * (t0 + t1) * ld
        side = np.where(la + 0.5 * ld >= 0.0, 1.0, -1.0)
        gap = h2 - side * loc
        k = int(np.argmin(gap))
        sgn = side[k]
        axis = r2[:, k] * sgn
        face = loc.copy()
        face[k] = sgn * h2[k]
        mid = c2 + r2 @ loc
        return -float(gap[k]) - s1[0], mid - s1[0] * axis, c2 + r2 @ face
    axis, nrm = mju_normalize3(w - u)
    return nrm - s1[0], u + s1[0] * axis, w


def _sphere_sphere(c1, r1, c2, r2):
    axis, nrm = mju_normalize3(c2 - c1)
    return nrm - r1 - r2, c1 + r1 * axis, c2 - r2 * axis


def mj_geomDistance(m, d, g1, g2, distmax):
    """Smallest signed distance between two primitive geoms, and the segment
    (fromto) joining the closest points, or (distmax, zeros) when farther than
    distmax.  Supported pairs: plane-{sphere, capsule, box}, sphere-{sphere, capsule, box},
    capsule-{capsule, box}, box-box.
    PARITY UNPINNED: MuJoCo's box-box and capsule-box routines are not restated; the
    Euclidean closest-point distance is used instead."""
    t1, t2 = m.geom_type[g1], m.geom_type[g2]
    fromto = np.zeros(6)
    swap = False
    if t1 > t2:
        g1, g2, t1, t2 = g2, g1, t2, t1
        swap = True
    c1, c2 = d.geom_xpos[g1], d.geom_xpos[g2]
    r1, r2 = d.geom_xmat[g1].reshape(3, 3), d.geom_xmat[g2].reshape(3, 3)
    s1, s2 = m.geom_size[g1], m.geom_size[g2]
    if t1 == GEOM_PLANE and t2 == GEOM_BOX:
        n = r1[:, 2]
        corners = _box_support_points(c2, r2, s2)
        depth = (corners - c1) @ n
        k = int(np.argmin(depth))
        dist = float(depth[k])
        p2 = corners[k]
        p1 = p2 - dist * n
    elif t1 == GEOM_PLANE and t2 == GEOM_SPHERE:
        n = r1[:, 2]
        dist = float((c2 - c1) @ n) - s2[0]
        p2 = c2 - s2[0] * n
        p1 = p2 - dist * n
    elif t1 == GEOM_SPHERE and t2 == GEOM_SPHERE:
        axis, nrm = mju_normalize3(c2 - c1)
        dist = nrm - s1[0] - s2[0]
        p1 = c1 + s1[0] * axis
        p2 = c2 - s2[0] * axis
    elif t1 == GEOM_PLANE and t2 == GEOM_CAPSULE:
        # the lower of the two end spheres (mjc_PlaneCapsule makes one contact per end)
        n, ax = r1[:, 2], r2[:, 2]
        ends = [c2 + ax * s2[1], c2 - ax * s2[1]]
        depth = [float((e - c1) @ n) - s2[0] for e in ends]
        k = 0 if depth[0] <= depth[1] else 1
        dist = depth[k]
        p2 = ends[k] - s2[0] * n
        p1 = p2 - dist * n
    elif t1 == GEOM_SPHERE and t2 == GEOM_CAPSULE:
        ax = r2[:, 2]
        t = float(np.clip((c1 - c2) @ ax, -s2[1], s2[1]))
        dist, p1, p2 = _sphere_sphere(c1, s1[0], c2 + t * ax, s2[0])
    elif t1 == GEOM_CAPSULE and t2 == GEOM_CAPSULE:
        a1, a2 = r1[:, 2], r2[:, 2]
        u, w = _seg_seg(c1 - a1 * s1[1], c1 + a1 * s1[1], c2 - a2 * s2[1], c2 + a2 * s2[1])
        dist, p1, p2 = _sphere_sphere(u, s1[0], w, s2[0])
    elif t1 == GEOM_SPHERE and t2 == GEOM_BOX:
        loc = r2.T @ (c1 - c2)
        clamped = np.clip(loc, -s2, s2)
        if np.any(np.abs(loc) > s2):          # centre outside the box
            pb = c2 + r2 @ clamped
            axis, nrm = mju_normalize3(c1 - pb)
            dist = nrm - s1[0]
            p1, p2 = c1 - s1[0] * axis, pb
        else:                                 # centre inside: leave through the nearest face
            gap = s2 - np.abs(loc)
            k = int(np.argmin(gap))
            sgn = 1.0 if loc[k] >= 0 else -1.0
            axis = r2[:, k] * sgn
            dist = -float(gap[k]) - s1[0]
            face = loc.copy()
            face[k] = sgn * s2[k]
            p2 = c2 + r2 @ face
            p1 = c1 - s1[0] * axis
    elif t1 == GEOM_CAPSULE and t2 == GEOM_BOX:
        dist, p1, p2 = _capsule_box(c1, r1, s1, c2, r2, s2)
    elif t1 == GEOM_BOX and t2 == GEOM_BOX:
        dist, p1, p2 = _closest_box_box(c1, r1, s1, c2, r2, s2)
    else:
        # no closed form: support mappings (oracle/convex.py); meshes by their hull vertices
        from . import convex as cx

        def shape(g, c, r):
            sh = dict(type=int(m.geom_type[g]), pos=np.asarray(c, dtype=float), mat=np.asarray(r, dtype=float),
                      size=np.asarray(m.geom_size[g], dtype=float))
            if sh["type"] == cx.MESH:
                k = int(m.geom_dataid[g])
                if k < 0 or m.mesh_vertnum[k] == 0:
                    raise NotImplementedError(f"mesh geom {g} has no hull vertices in the model table")
                sh["verts"] = m.mesh_vert[m.mesh_vertadr[k]:m.mesh_vertadr[k] + m.mesh_vertnum[k]]
            elif sh["type"] not in (cx.SPHERE, cx.CAPSULE, cx.ELLIPSOID, cx.CYLINDER, cx.BOX):
                raise NotImplementedError(f"geom type {sh['type']}")
            return sh

        if t1 == GEOM_PLANE:
            if t2 == GEOM_PLANE:
                raise NotImplementedError("plane - plane pair")
            dist, p1, p2 = cx.plane_distance(dict(pos=c1, mat=r1), shape(g2, c2, r2))
        else:
            dist, p1, p2 = cx.distance(shape(g1, c1, r1), shape(g2, c2, r2))
    if dist >= distmax:
        return distmax, fromto
    if swap:
        p1, p2 = p2, p1
    fromto[:3], fromto[3:] = p1, p2
    return dist, fromto
