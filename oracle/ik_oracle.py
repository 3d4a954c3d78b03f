"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's differential
IK path, one problem at a time, at the reference's own operation granularity.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it; nothing under mink_amd/ does.

What is restated (reference file:line):
  * SO3 / SE3 exp, log, Jacobians           mink/lie/so3.py:158-226, mink/lie/se3.py:112-249,
                                            mink/lie/base.py:109-156
  * Configuration (FK, frame Jacobian, ...)  mink/configuration.py:53-236
  * Task.compute_qp_objective                mink/tasks/task.py:105-138
  * FrameTask / RelativeFrameTask            mink/tasks/frame_task.py:95-146,
                                            mink/tasks/relative_frame_task.py:106-142
  * PostureTask / DampingTask / ComTask      mink/tasks/posture_task.py:87-142,
                                            mink/tasks/damping_task.py:18-20, mink/tasks/com_task.py:71-97
  * Configuration/Velocity/Collision limits  mink/limits/configuration_limit.py:18-124,
                                            mink/limits/velocity_limit.py:33-101,
                                            mink/limits/collision_avoidance_limit.py:59-278
  * build_ik / solve_ik                      mink/solve_ik.py:13-105
  * the QP backend: qpsolvers 4.4.0 -> quadprog 0.1.13 (third-party, absent here);
    its published algorithm (Goldfarb & Idnani 1983, dual active set for strictly
    convex QPs) is restated in solve_qp().

Pinning: the mink-level functions are checked against golden vectors captured by
running the unmodified reference modules (tests/golden/make_golden.py).  The
MuJoCo kinematics underneath (oracle/mjlite.py) and the QP solver are PARITY
UNPINNED against their third-party originals; they are pinned by finite
differences, KKT residuals and brute-force active-set enumeration instead.
"""

from __future__ import annotations

import logging
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import mjlite as mj

EPS64 = 1e-10  # mink/lie/utils.py:4-8 (float64 entry)


# ====================================================================== lie ==
def skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def so3_exp(w):
    """Rotation vector -> unit quaternion (wxyz). mink/lie/so3.py:158-173."""
    t2 = float(w @ w)
    if t2 < EPS64:
        t4 = t2 * t2
        re = 1.0 - t2 / 8.0 + t4 / 384.0
        im = 0.5 - t2 / 48.0 + t4 / 3840.0
    else:
        t = math.sqrt(t2)
        re = math.cos(0.5 * t)
        im = math.sin(0.5 * t) / t
    return np.array([re, im * w[0], im * w[1], im * w[2]])


def so3_log(q):
    """Unit quaternion -> rotation vector; safe for w < 0. mink/lie/so3.py:176-191."""
    w = q[0]
    n2 = float(q[1:] @ q[1:])
    if n2 < EPS64:
        f = 2.0 / w - 2.0 / 3.0 * n2 / w**3
    else:
        n = math.sqrt(n2)
        if abs(w) < EPS64:
            f = (1.0 if w > 0.0 else -1.0) * math.pi / n
        else:
            f = 2.0 * math.atan2(-n if w < 0 else n, abs(w)) / n
    return f * q[1:]


def so3_inverse(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def so3_apply(q, v):
    """Rotate v by q through the sandwich product. mink/lie/so3.py:143-146."""
    p = np.array([0.0, v[0], v[1], v[2]])
    return mj.mju_mulQuat(mj.mju_mulQuat(q, p), so3_inverse(q))[1:]


def so3_ljac(w):
    """mink/lie/so3.py:200-212 (note: switches on theta, not theta^2)."""
    t = math.sqrt(float(w @ w))
    if t < EPS64:
        t2 = t * t
        A = 0.5 * (1.0 - t2 / 12.0 * (1.0 - t2 / 30.0 * (1.0 - t2 / 56.0)))
        B = (1.0 / 6.0) * (1.0 - t2 / 20.0 * (1.0 - t2 / 42.0 * (1.0 - t2 / 72.0)))
    else:
        A = (1.0 - math.cos(t)) / t**2
        B = (t - math.sin(t)) / t**3
    S = skew(w)
    return np.eye(3) + A * S + B * (S @ S)


def so3_ljacinv(w):
    """mink/lie/so3.py:214-226."""
    t = math.sqrt(float(w @ w))
    if t < EPS64:
        t2 = t * t
        A = (1.0 / 12.0) * (1.0 + t2 / 60.0 * (1.0 + t2 / 42.0 * (1.0 + t2 / 40.0)))
    else:
        A = (1.0 / t**2) * (1.0 - (t * math.sin(t) / (2.0 * (1.0 - math.cos(t)))))
    S = skew(w)
    return np.eye(3) - 0.5 * S + A * (S @ S)


# SE3 elements are 7-vectors (qw, qx, qy, qz, x, y, z); tangents are (v; omega).
def se3_exp(xi):
    """mink/lie/se3.py:112-134."""
    w = xi[3:]
    q = so3_exp(w)
    t2 = float(w @ w)
    if t2 < EPS64:
        V = mj.mju_quat2Mat(q)
    else:
        t = math.sqrt(t2)
        S = skew(w)
        V = np.eye(3) + (1.0 - math.cos(t)) / t2 * S + (t - math.sin(t)) / (t2 * t) * (S @ S)
    return np.concatenate([q, V @ xi[:3]])


def se3_inverse(T):
    qi = so3_inverse(T[:4])
    return np.concatenate([qi, -so3_apply(qi, T[4:])])


def se3_multiply(A, B):
    return np.concatenate([mj.mju_mulQuat(A[:4], B[:4]), so3_apply(A[:4], B[4:]) + A[4:]])


def se3_log(T):
    """mink/lie/se3.py:159-185."""
    w = so3_log(T[:4])
    t2 = float(w @ w)
    S = skew(w)
    S2 = S @ S
    if t2 < EPS64:
        Vinv = np.eye(3) - 0.5 * S + S2 / 12.0
    else:
        t = math.sqrt(t2)
        h = 0.5 * t
        Vinv = np.eye(3) - 0.5 * S + (1.0 - t * math.cos(h) / (2.0 * math.sin(h))) / t2 * S2
    return np.concatenate([Vinv @ T[4:], w])


def se3_adjoint(T):
    """[[R, [t]x R], [0, R]]. mink/lie/se3.py:187-194."""
    R = mj.mju_quat2Mat(T[:4])
    A = np.zeros((6, 6))
    A[:3, :3] = R
    A[:3, 3:] = skew(T[4:]) @ R
    A[3:, 3:] = R
    return A


def se3_getQ(c):
    """mink/lie/se3.py:222-249."""
    t2 = float(c[3:] @ c[3:])
    A = 0.5
    if t2 < EPS64:
        B = 1.0 / 6.0 + (1.0 / 120.0) * t2
        C = -1.0 / 24.0 + (1.0 / 720.0) * t2
        D = -1.0 / 60.0
    else:
        t = math.sqrt(t2)
        s, co = math.sin(t), math.cos(t)
        B = (t - s) / (t2 * t)
        C = (1.0 - t2 / 2.0 - co) / (t2 * t2)
        D = (2.0 * t - 3.0 * s + t * co) / (2.0 * t2 * t2 * t)
    V = skew(c[:3])
    W = skew(c[3:])
    VW = V @ W
    WV = VW.T
    WVW = WV @ W
    VWW = VW @ W
    return A * V + B * (WV + VW + WVW) - C * (VWW - VWW.T - 3.0 * WVW) + D * (WVW @ W + W @ WVW)


def se3_ljac(c):
    w = c[3:]
    if float(w @ w) < EPS64:
        return np.eye(6)
    J = so3_ljac(w)
    out = np.zeros((6, 6))
    out[:3, :3] = J
    out[:3, 3:] = se3_getQ(c)
    out[3:, 3:] = J
    return out


def se3_ljacinv(c):
    """mink/lie/se3.py:210-218 -- exactly I6 below the theta^2 threshold."""
    w = c[3:]
    if float(w @ w) < EPS64:
        return np.eye(6)
    Ji = so3_ljacinv(w)
    out = np.zeros((6, 6))
    out[:3, :3] = Ji
    out[:3, 3:] = -Ji @ se3_getQ(c) @ Ji
    out[3:, 3:] = Ji
    return out


def se3_jlog(T):
    """rjacinv(log T) = ljacinv(-log T). mink/lie/base.py:151-156."""
    return se3_ljacinv(-se3_log(T))


def se3_rminus(A, B):
    """A (-) B = log(B^-1 A). mink/lie/base.py:113-114."""
    return se3_log(se3_multiply(se3_inverse(B), A))


# ============================================================ configuration ==
class NotWithinConfigurationLimits(Exception):
    pass


class Configuration:
    """mink/configuration.py:21-253 on top of oracle/mjlite.py."""

    def __init__(self, model, q=None):
        self.model = model
        self.data = mj.Data(model)
        self.update(q)

    def update(self, q=None):
        if q is not None:
            self.data.qpos = np.array(q, dtype=np.float64)
        mj.mj_kinematics(self.model, self.data)
        mj.mj_comPos(self.model, self.data)

    @property
    def q(self):
        return self.data.qpos.copy()

    def check_limits(self, tol=1e-6, safety_break=True):
        m = self.model
        for j in range(m.njnt):
            if m.jnt_type[j] == mj.JNT_FREE or not m.jnt_limited[j]:
                continue
            qv = self.data.qpos[m.jnt_qposadr[j]]
            lo, hi = m.jnt_range[j]
            if qv < lo - tol or qv > hi + tol:
                if safety_break:
                    raise NotWithinConfigurationLimits(
                        f"Joint {j} ({m.jnt_names[j]}) violates configuration limits "
                        f"{lo} <= {qv} <= {hi}")
                logging.warning(
                    f"Value {qv:.2f} at index {j} is outside of its limits: [{lo:.2f}, {hi:.2f}]")

    def _frame(self, name, ftype):
        m, d = self.model, self.data
        if ftype == "body":
            i = m.body_id(name)
            return i, d.xpos[i], d.xmat[i], m.body_parentid, i
        if ftype == "site":
            i = m.site_id(name)
            return i, d.site_xpos[i], d.site_xmat[i], None, m.site_bodyid[i]
        if ftype == "geom":
            i = m.geom_id(name)
            return i, d.geom_xpos[i], d.geom_xmat[i], None, m.geom_bodyid[i]
        raise ValueError(f"{ftype} is not supported.")

    def get_transform_frame_to_world(self, name, ftype):
        i, pos, mat, _, _ = self._frame(name, ftype)
        if i < 0:
            raise KeyError(name)
        return np.concatenate([mj.mju_mat2Quat(mat), pos])

    def get_transform(self, src, src_type, dst, dst_type):
        Ts = self.get_transform_frame_to_world(src, src_type)
        Td = self.get_transform_frame_to_world(dst, dst_type)
        return se3_multiply(se3_inverse(Td), Ts)

    def get_frame_jacobian(self, name, ftype):
        """Body-frame 6 x nv Jacobian, rows (linear; angular). configuration.py:112-155."""
        i, pos, mat, _, body = self._frame(name, ftype)
        if i < 0:
            raise KeyError(name)
        jp, jr = mj.mj_jac(self.model, self.data, pos, body)
        jac = np.vstack([jp, jr])
        R_wf = mj.mju_mat2Quat(mat)
        A_fw = se3_adjoint(np.concatenate([so3_inverse(R_wf), np.zeros(3)]))
        return A_fw @ jac

    def integrate(self, v, dt):
        q = self.data.qpos.copy()
        mj.mj_integratePos(self.model, q, np.asarray(v, dtype=np.float64), dt)
        return q

    def integrate_inplace(self, v, dt):
        mj.mj_integratePos(self.model, self.data.qpos, np.asarray(v, dtype=np.float64), dt)
        self.update()


def freejoint_dof_ids(model) -> List[int]:
    """mink/utils.py:38-56 (tangent indices only)."""
    ids: List[int] = []
    for j in range(model.njnt):
        if model.jnt_type[j] == mj.JNT_FREE:
            a = model.jnt_dofadr[j]
            ids.extend(range(a, a + 6))
    return ids


# ==================================================================== tasks ==
@dataclass
class FrameTask:
    frame_name: str
    frame_type: str
    position_cost: object = 1.0
    orientation_cost: object = 1.0
    gain: float = 1.0
    lm_damping: float = 0.0
    target: Optional[np.ndarray] = None  # wxyz_xyz, target-to-world

    @property
    def cost(self):
        c = np.zeros(6)
        c[:3] = self.position_cost
        c[3:] = self.orientation_cost
        return c

    def error(self, cfg):
        T_wb = cfg.get_transform_frame_to_world(self.frame_name, self.frame_type)
        return se3_rminus(self.target, T_wb)

    def jacobian(self, cfg):
        jac = cfg.get_frame_jacobian(self.frame_name, self.frame_type)
        T_wb = cfg.get_transform_frame_to_world(self.frame_name, self.frame_type)
        T_tb = se3_multiply(se3_inverse(self.target), T_wb)
        return -se3_jlog(T_tb) @ jac


@dataclass
class RelativeFrameTask:
    frame_name: str
    frame_type: str
    root_name: str
    root_type: str
    position_cost: object = 1.0
    orientation_cost: object = 1.0
    gain: float = 1.0
    lm_damping: float = 0.0
    target: Optional[np.ndarray] = None  # target-to-root

    @property
    def cost(self):
        c = np.zeros(6)
        c[:3] = self.position_cost
        c[3:] = self.orientation_cost
        return c

    def error(self, cfg):
        T_rf = cfg.get_transform(self.frame_name, self.frame_type, self.root_name, self.root_type)
        return se3_rminus(T_rf, self.target)

    def jacobian(self, cfg):
        Jf = cfg.get_frame_jacobian(self.frame_name, self.frame_type)
        Jr = cfg.get_frame_jacobian(self.root_name, self.root_type)
        T_rf = cfg.get_transform(self.frame_name, self.frame_type, self.root_name, self.root_type)
        T_tf = se3_multiply(se3_inverse(self.target), T_rf)
        return se3_jlog(T_tf) @ (Jf - se3_adjoint(se3_inverse(T_rf)) @ Jr)


@dataclass
class PostureTask:
    cost: np.ndarray  # (nv,)
    target_q: np.ndarray  # (nq,)
    gain: float = 1.0
    lm_damping: float = 0.0

    def error(self, cfg):
        e = mj.mj_differentiatePos(cfg.model, 1.0, cfg.q, self.target_q)
        ids = freejoint_dof_ids(cfg.model)
        if ids:
            e[ids] = 0.0
        return e

    def jacobian(self, cfg):
        J = -np.eye(cfg.model.nv)
        ids = freejoint_dof_ids(cfg.model)
        if ids:
            J[:, ids] = 0.0
        return J


def DampingTask(model, cost):
    """PostureTask with gain 0 and target qpos0. mink/tasks/damping_task.py:18-20."""
    c = np.zeros(model.nv)
    c[:] = cost
    return PostureTask(cost=c, target_q=model.qpos0.copy(), gain=0.0, lm_damping=0.0)


@dataclass
class ComTask:
    cost: object
    target_com: np.ndarray
    gain: float = 1.0
    lm_damping: float = 0.0

    def error(self, cfg):
        return cfg.data.subtree_com[1] - self.target_com

    def jacobian(self, cfg):
        return mj.mj_jacSubtreeCom(cfg.model, cfg.data, 1)


def task_qp_objective(task, cfg) -> Tuple[np.ndarray, np.ndarray]:
    """(H, c) of one task. mink/tasks/task.py:105-138."""
    J = task.jacobian(cfg)
    e = -task.gain * task.error(cfg)
    cost = np.zeros(J.shape[0])
    cost[:] = task.cost
    Jw = cost[:, None] * J
    ew = cost * e
    mu = task.lm_damping * float(ew @ ew)
    H = Jw.T @ Jw + mu * np.eye(cfg.model.nv)
    c = -ew @ Jw
    return H, c


# =================================================================== limits ==
@dataclass
class ConfigurationLimit:
    """mink/limits/configuration_limit.py:18-124."""
    model: object
    gain: float = 0.95
    min_distance_from_limits: float = 0.0

    def __post_init__(self):
        m = self.model
        idx: List[int] = []
        lower = np.full(m.nq, -mj.MAXVAL)
        upper = np.full(m.nq, mj.MAXVAL)
        for j in range(m.njnt):
            jt = m.jnt_type[j]
            if jt == mj.JNT_FREE or not m.jnt_limited[j]:
                continue
            a = m.jnt_qposadr[j]
            qw = {0: 7, 1: 4, 2: 1, 3: 1}[int(jt)]
            lower[a:a + qw] = m.jnt_range[j, 0] + self.min_distance_from_limits
            upper[a:a + qw] = m.jnt_range[j, 1] - self.min_distance_from_limits
            vw = {0: 6, 1: 3, 2: 1, 3: 1}[int(jt)]
            idx.extend(range(m.jnt_dofadr[j], m.jnt_dofadr[j] + vw))
        self.indices = np.array(idx, dtype=np.int64)
        self.lower, self.upper = lower, upper

    def inequalities(self, cfg, dt):
        if len(self.indices) == 0:
            return None, None
        m = self.model
        dq_max = mj.mj_differentiatePos(m, 1.0, cfg.q, self.upper)
        dq_min = mj.mj_differentiatePos(m, 1.0, self.lower, cfg.q)
        P = np.eye(m.nv)[self.indices]
        G = np.vstack([P, -P])
        h = np.hstack([self.gain * dq_max[self.indices], self.gain * dq_min[self.indices]])
        return G, h


@dataclass
class VelocityLimit:
    """mink/limits/velocity_limit.py:33-101; ``velocities`` maps joint name -> |v|max."""
    model: object
    velocities: dict = field(default_factory=dict)

    def __post_init__(self):
        m = self.model
        idx: List[int] = []
        lim: List[float] = []
        for name, vmax in self.velocities.items():
            j = m.joint_id(name)
            jt = int(m.jnt_type[j])
            if jt == mj.JNT_FREE:
                raise ValueError(f"Free joint {name} is not supported")
            vw = {1: 3, 2: 1, 3: 1}[jt]
            vmax = np.atleast_1d(vmax).astype(np.float64)
            if vmax.shape != (vw,):
                raise ValueError(f"Joint {name} must have a limit of shape ({vw},). Got: {vmax.shape}")
            idx.extend(range(m.jnt_dofadr[j], m.jnt_dofadr[j] + vw))
            lim.extend(vmax.tolist())
        self.indices = np.array(idx, dtype=np.int64)
        self.limit = np.array(lim)

    def inequalities(self, cfg, dt):
        if len(self.indices) == 0:
            return None, None
        P = np.eye(self.model.nv)[self.indices]
        return np.vstack([P, -P]), np.hstack([dt * self.limit, dt * self.limit])


@dataclass
class CollisionAvoidanceLimit:
    """mink/limits/collision_avoidance_limit.py:118-278; geom_pairs is a sequence of
    (group_a, group_b) with geoms given by name or id."""
    model: object
    geom_pairs: Sequence
    gain: float = 0.85
    minimum_distance_from_collisions: float = 0.005
    collision_detection_distance: float = 0.01
    bound_relaxation: float = 0.0

    def __post_init__(self):
        m = self.model

        def ids(group):
            return list(set(g if isinstance(g, (int, np.integer)) else m.geom_id(g) for g in group))

        pairs = []
        for ga, gb in self.geom_pairs:
            for a in ids(ga):
                for b in ids(gb):
                    if geom_pair_allowed(m, a, b):
                        pairs.append((min(a, b), max(a, b)))
        self.geom_id_pairs = pairs

    def inequalities(self, cfg, dt):
        m, d = self.model, cfg.data
        n = len(self.geom_id_pairs)
        h = np.full(n, np.inf)
        G = np.zeros((n, m.nv))
        for k, (g1, g2) in enumerate(self.geom_id_pairs):
            dist, fromto = mj.mj_geomDistance(m, d, g1, g2, self.collision_detection_distance)
            if dist == self.collision_detection_distance:
                continue
            if dist > self.minimum_distance_from_collisions:
                h[k] = self.gain * (dist - self.minimum_distance_from_collisions) / dt + self.bound_relaxation
            else:
                h[k] = self.bound_relaxation
            normal, _ = mj.mju_normalize3(fromto[3:] - fromto[:3])
            j2, _ = mj.mj_jac(m, d, fromto[3:], m.geom_bodyid[g2])
            j1, _ = mj.mj_jac(m, d, fromto[:3], m.geom_bodyid[g1])
            G[k] = -(normal @ (j2 - j1))
        return G, h


def geom_pair_allowed(m, a, b) -> bool:
    """The three filters of collision_avoidance_limit.py:75-115,269-276."""
    ba, bb = m.geom_bodyid[a], m.geom_bodyid[b]
    wa, wb = m.body_weldid[ba], m.body_weldid[bb]
    if wa == wb:
        return False
    pwa = m.body_weldid[m.body_parentid[wa]]
    pwb = m.body_weldid[m.body_parentid[wb]]
    if wa == pwb or wb == pwa:
        return False
    return bool(m.geom_contype[a] & m.geom_conaffinity[b]) or bool(m.geom_contype[b] & m.geom_conaffinity[a])


# ================================================================= build/solve
def build_ik(cfg, tasks, dt, damping=1e-12, limits=None):
    """(P, q, G, h); G, h are None without active limits. mink/solve_ik.py:13-65."""
    nv = cfg.model.nv
    H = np.eye(nv) * damping
    c = np.zeros(nv)
    for t in tasks:
        Ht, ct = task_qp_objective(t, cfg)
        H += Ht
        c += ct
    if limits is None:
        limits = [ConfigurationLimit(cfg.model)]
    Gs, hs = [], []
    for lim in limits:
        G, h = lim.inequalities(cfg, dt)
        if G is not None:
            Gs.append(G)
            hs.append(h)
    if not Gs:
        return H, c, None, None
    return H, c, np.vstack(Gs), np.hstack(hs)


def solve_qp(P, q, G=None, h=None, max_iter=None):
    """min 1/2 x'Px + q'x  s.t. Gx <= h, P positive definite.

    Dual active-set method of Goldfarb & Idnani (1983), the algorithm behind
    quadprog: start at the unconstrained minimiser, repeatedly pick the most
    violated constraint and move in the (primal, dual) direction that keeps the
    active constraints tight, taking partial steps that drop a blocking
    constraint when a multiplier would turn negative.  The projected operators
    are recomputed from the active set each step (no Givens updates): same
    iterates, simpler code.

    Returns (x, status, active, u): status 0 ok, 1 iteration limit, 2 infeasible.
    """
    n = P.shape[0]
    L = np.linalg.cholesky(P)

    def Hinv(B):
        return np.linalg.solve(L.T, np.linalg.solve(L, B))

    x = -Hinv(q)
    if G is None or G.shape[0] == 0:
        return x, 0, [], np.zeros(0)
    m = G.shape[0]
    active: List[int] = []
    u = np.zeros(0)
    if max_iter is None:
        max_iter = 10 * (m + n)
    rownorm = np.sqrt(np.sum(G * G, axis=1))
    for _ in range(max_iter):
        with np.errstate(invalid="ignore"):
            viol = G @ x - h
        viol[active] = -np.inf
        scaled = np.where(rownorm > 0, viol / np.where(rownorm > 0, rownorm, 1.0), -np.inf)
        p = int(np.argmax(scaled))
        if not scaled[p] > 1e-12 * (1.0 + 0.0):
            return x, 0, active, u
        npl = G[p]
        up = 0.0
        while True:
            HiN = Hinv(npl)
            if active:
                N = G[active].T  # n x k
                HiNa = Hinv(N)
                M = N.T @ HiNa
                r = np.linalg.solve(M, N.T @ HiN)
                z = -(HiN - HiNa @ r)
            else:
                r = np.zeros(0)
                z = -HiN
            nz = -float(npl @ z)  # = n' Pproj n >= 0
            dependent = nz <= 1e-13 * float(npl @ HiN)
            t1, kblock = np.inf, -1
            for k in range(len(active)):
                if r[k] > 0 and u[k] / r[k] < t1:
                    t1, kblock = u[k] / r[k], k
            t2 = np.inf if dependent else float(G[p] @ x - h[p]) / nz
            t = min(t1, t2)
            if not np.isfinite(t):
                return x, 2, active, u
            if dependent:
                u = u - t * r
                up += t
                active.pop(kblock)
                u = np.delete(u, kblock)
                continue
            x = x + t * z
            u = u - t * r
            up += t
            if t2 <= t1:
                active.append(p)
                u = np.append(u, up)
                break
            active.pop(kblock)
            u = np.delete(u, kblock)
    return x, 1, active, u


def solve_ik(cfg, tasks, dt, solver="quadprog", damping=1e-12, safety_break=False, limits=None):
    """Velocity v = dq / dt. mink/solve_ik.py:68-105."""
    cfg.check_limits(safety_break=safety_break)
    P, q, G, h = build_ik(cfg, tasks, dt, damping, limits)
    x, status, _, _ = solve_qp(P, q, G, h)
    assert status == 0, "QP solver failed"
    return x / dt
