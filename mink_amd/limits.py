"""Kinematic limits of the host API (reference: mink/limits/*.py)."""
from __future__ import annotations

from typing import Mapping, NamedTuple, Optional, Sequence

import numpy as np

from . import _lib as L
from .constants import dof_width, qpos_width
from .exceptions import LimitDefinitionError
from .mjcf import JNT_BALL, JNT_FREE, MJ_MAXVAL


class Constraint(NamedTuple):
    r"""Linear inequality :math:`G(q) \Delta q \leq h(q)`; inactive if both are None."""
    G: Optional[object] = None
    h: Optional[object] = None

    @property
    def inactive(self) -> bool:
        return self.G is None and self.h is None


class Limit:
    """Base class; subclasses lower themselves to a ``mk_limit_desc``."""

    def _spec(self, model) -> Optional[dict]:
        raise NotImplementedError

    def _key(self, model):
        """Hashable value of ``_spec`` (None for a limit without rows)."""
        from ._problems import _signature
        return _signature(self._spec(model))

    def _is_custom(self) -> bool:
        """True for a subclass that brings its own compute_qp_inequalities instead of a
        descriptor the kernels evaluate (the Limit seam of mink/limits/limit.py:11-57)."""
        return type(self)._spec is Limit._spec

    def compute_qp_inequalities(self, configuration, dt: float) -> Constraint:
        from ._problems import get_problem
        spec = self._spec(configuration.model)
        if spec is None:
            return Constraint()
        prob, _ = get_problem(configuration, [], [self])
        _, _, G, h = prob.build_qp(configuration.q_tensor, None, dt, 0.0)
        return Constraint(G=configuration._out(G), h=configuration._out(h))


class ConfigurationLimit(Limit):
    """Joint position limits; floating-base joints are ignored
    (mink/limits/configuration_limit.py)."""

    def __init__(self, model, gain: float = 0.95, min_distance_from_limits: float = 0.0):
        if not 0.0 < gain <= 1.0:
            raise LimitDefinitionError(f"{self.__class__.__name__} gain must be in the range (0, 1]")
        index_list: list = []
        lower = np.full(model.nq, -MJ_MAXVAL)
        upper = np.full(model.nq, MJ_MAXVAL)
        for jnt in range(model.njnt):
            jt = int(model.jnt_type[jnt])
            if jt == JNT_FREE or not model.jnt_limited[jnt]:
                continue
            padr = model.jnt_qposadr[jnt]
            qdim = qpos_width(jt)      # a ball joint fills its four quaternion slots
            lower[padr:padr + qdim] = model.jnt_range[jnt, 0] + min_distance_from_limits
            upper[padr:padr + qdim] = model.jnt_range[jnt, 1] - min_distance_from_limits
            vadr = model.jnt_dofadr[jnt]
            index_list.extend(range(vadr, vadr + dof_width(jt)))
        self.indices = np.array(index_list, dtype=np.int64)
        self.indices.setflags(write=False)
        self.projection_matrix = np.eye(model.nv)[self.indices] if len(self.indices) > 0 else None
        self.lower, self.upper = lower, upper
        self.model = model
        self.gain = gain
        self.min_distance_from_limits = min_distance_from_limits
        # the kernels derive the bounds from the model's joint ranges; a caller that
        # overwrites lower / upper afterwards (the reference reads them at every call,
        # mink/limits/configuration_limit.py:69-124) is served by the generic route instead
        self._built_bounds = (lower.copy(), upper.copy())
        self._limited_joints = [j for j in range(model.njnt)
                                if int(model.jnt_type[j]) != JNT_FREE and model.jnt_limited[j]]
        # A limited ball joint has no per-dof range: the reference differences the configuration
        # with `upper` / `lower` read as quaternions (mink/limits/configuration_limit.py:98-119);
        # the kernels do not carry that, the generic route below does.
        self._has_limited_ball = any(int(model.jnt_type[j]) == JNT_BALL for j in self._limited_joints)

    def _overridden(self) -> bool:
        return not (np.array_equal(self.lower, self._built_bounds[0]) and np.array_equal(self.upper, self._built_bounds[1]))

    def _is_custom(self) -> bool:
        return self._overridden() or self._has_limited_ball

    def _difference(self, q, bound):
        """(bound (-) q) on the limited dofs, i.e. mj_differentiatePos(dt = 1, qpos1 = q,
        qpos2 = bound)[indices] for q [B, nq]: plain differences for hinge / slide joints, and for
        a ball joint the rotation vector of conj(q) * bound (mju_subQuat: quaternions as given,
        not normalised; angle wrapped to (-pi, pi])."""
        import torch
        m = self.model
        cols = []
        for j in self._limited_joints:
            a = int(m.jnt_qposadr[j])
            if int(m.jnt_type[j]) != JNT_BALL:
                cols.append((bound[a] - q[:, a])[:, None])
                continue
            qa, qb = bound[a:a + 4][None, :], q[:, a:a + 4] * torch.tensor([1.0, -1.0, -1.0, -1.0], dtype=q.dtype, device=q.device)
            # d = conj(q) * bound
            w = qb[:, 0] * qa[:, 0] - (qb[:, 1:] * qa[:, 1:]).sum(dim=1)
            xyz = qb[:, :1] * qa[:, 1:] + qa[:, :1] * qb[:, 1:] + torch.linalg.cross(qb[:, 1:], qa[:, 1:].expand_as(qb[:, 1:]))
            n = xyz.norm(dim=1)
            axis = torch.where((n < 1e-15)[:, None], torch.tensor([1.0, 0.0, 0.0], dtype=q.dtype, device=q.device), xyz / n.clamp_min(1e-300)[:, None])
            angle = 2.0 * torch.atan2(n, w)
            angle = torch.where(angle > np.pi, angle - 2.0 * np.pi, angle)
            cols.append(axis * angle[:, None])
        return torch.cat(cols, dim=1)

    def compute_qp_inequalities(self, configuration, dt: float) -> Constraint:
        if not self._is_custom():
            return super().compute_qp_inequalities(configuration, dt)
        if self.projection_matrix is None:
            return Constraint()
        import torch
        q = configuration.q_tensor
        lo = torch.as_tensor(np.asarray(self.lower), dtype=torch.float64).to(q.device)
        hi = torch.as_tensor(np.asarray(self.upper), dtype=torch.float64).to(q.device)
        P = torch.as_tensor(self.projection_matrix, dtype=torch.float64).to(q.device)
        B = q.shape[0]
        G = torch.cat([P, -P], dim=0)[None].expand(B, -1, -1).contiguous()
        # h = gain * [(upper (-) q); -(lower (-) q)]   (mink/limits/configuration_limit.py:98-124)
        h = torch.cat([self.gain * self._difference(q, hi), -self.gain * self._difference(q, lo)], dim=1)
        return Constraint(G=configuration._out(G), h=configuration._out(h))

    def _key(self, model):
        if self.projection_matrix is None:
            return None
        return (L.LIMIT_CONFIGURATION, self.gain, self.min_distance_from_limits)

    def _spec(self, model):
        if self.projection_matrix is None:
            return None
        return dict(type=L.LIMIT_CONFIGURATION, gain=self.gain,
                    min_distance_from_limits=self.min_distance_from_limits)


class VelocityLimit(Limit):
    """Joint velocity limits (mink/limits/velocity_limit.py); ``velocities`` maps joint
    name to the maximum magnitude."""

    def __init__(self, model, velocities: Mapping[str, object] = {}):
        limit_list: list = []
        index_list: list = []
        for joint_name, max_vel in velocities.items():
            jid = model.joint_id(joint_name)
            if jid < 0:
                raise LimitDefinitionError(f"Joint {joint_name} does not exist in the model")
            jt = int(model.jnt_type[jid])
            if jt == JNT_FREE:
                raise LimitDefinitionError(f"Free joint {joint_name} is not supported")
            vadr = model.jnt_dofadr[jid]
            vdim = dof_width(jt)
            max_vel = np.atleast_1d(max_vel)
            if max_vel.shape != (vdim,):
                raise LimitDefinitionError(f"Joint {joint_name} must have a limit of shape ({vdim},). "
                                           f"Got: {max_vel.shape}")
            index_list.extend(range(vadr, vadr + vdim))
            limit_list.extend(max_vel.tolist())
        self.indices = np.array(index_list, dtype=np.int64)
        self.indices.setflags(write=False)
        self.limit = np.array(limit_list, dtype=np.float64)
        self.limit.setflags(write=False)
        self.projection_matrix = np.eye(model.nv)[self.indices] if len(self.indices) > 0 else None

    def _key(self, model):
        if self.projection_matrix is None:
            return None
        return (L.LIMIT_VELOCITY, self.indices.tobytes(), self.limit.tobytes())

    def _spec(self, model):
        if self.projection_matrix is None:
            return None
        return dict(type=L.LIMIT_VELOCITY, index=self.indices.astype(np.int32), vmax=np.array(self.limit))


def _allowed(model, a: int, b: int) -> bool:
    """Pair filters of mink/limits/collision_avoidance_limit.py:75-115: not welded
    together, not parent / child, and passing the contype / conaffinity test."""
    ba, bb = model.geom_bodyid[a], model.geom_bodyid[b]
    wa, wb = model.body_weldid[ba], model.body_weldid[bb]
    if wa == wb:
        return False
    pwa = model.body_weldid[model.body_parentid[wa]]
    pwb = model.body_weldid[model.body_parentid[wb]]
    if wa == pwb or wb == pwa:
        return False
    return bool(model.geom_contype[a] & model.geom_conaffinity[b]) or \
        bool(model.geom_contype[b] & model.geom_conaffinity[a])


class CollisionAvoidanceLimit(Limit):
    """Normal-velocity limit between geom pairs (mink/limits/collision_avoidance_limit.py).
    Distances are available between plane, sphere, capsule and box primitives."""

    def __init__(self, model, geom_pairs: Sequence, gain: float = 0.85,
                 minimum_distance_from_collisions: float = 0.005, collision_detection_distance: float = 0.01,
                 bound_relaxation: float = 0.0):
        self.model = model
        self.gain = gain
        self.minimum_distance_from_collisions = minimum_distance_from_collisions
        self.collision_detection_distance = collision_detection_distance
        self.bound_relaxation = bound_relaxation
        self.geom_id_pairs = self._construct_geom_id_pairs(geom_pairs)
        self.max_num_contacts = len(self.geom_id_pairs)

    def _ids(self, group) -> list:
        out = []
        for g in group:
            if isinstance(g, (int, np.integer)):
                out.append(int(g))
            else:
                gid = self.model.geom_id(g)
                if gid < 0:
                    raise LimitDefinitionError(f"geom '{g}' does not exist in the model")
                out.append(gid)
        return list(set(out))

    def _construct_geom_id_pairs(self, geom_pairs):
        pairs = []
        for group_a, group_b in geom_pairs:
            for a in self._ids(group_a):
                for b in self._ids(group_b):
                    if _allowed(self.model, a, b):
                        pairs.append((min(a, b), max(a, b)))
        return pairs

    def _spec(self, model):
        return dict(type=L.LIMIT_COLLISION, index=np.array(self.geom_id_pairs, dtype=np.int32).reshape(-1),
                    gain=self.gain, minimum_distance_from_collisions=self.minimum_distance_from_collisions,
                    collision_detection_distance=self.collision_detection_distance,
                    bound_relaxation=self.bound_relaxation)
