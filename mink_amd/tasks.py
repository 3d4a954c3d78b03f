"""Kinematic tasks of the host API (reference: mink/tasks/*.py).

A task holds its parameters and target; the arithmetic -- error, Jacobian, (H, c) --
is evaluated by the HIP kernels for the whole batch of a ``Configuration``.  Targets
may be one pose / vector shared by the batch or one per configuration.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import numpy as np
import torch

from . import _lib as L
from . import constants as consts
from . import exceptions
from .exceptions import InvalidDamping, InvalidGain, InvalidTarget, TargetNotSet, TaskDefinitionError
from .lie import SE3
from .mjcf import JNT_FREE


class Objective(NamedTuple):
    r"""Quadratic objective :math:`\frac{1}{2} x^T H x + c^T x`."""
    H: object
    c: object

    def value(self, x):
        x = np.asarray(x)
        H, c = np.asarray(self.H), np.asarray(self.c)
        return x.T @ H @ x + c @ x


class Task:
    """Base class: cost / gain / lm_damping validation as in mink/tasks/task.py:54-79."""

    def __init__(self, cost: np.ndarray, gain: float = 1.0, lm_damping: float = 0.0):
        if not 0.0 <= gain <= 1.0:
            raise InvalidGain("`gain` must be in the range [0, 1]")
        if lm_damping < 0.0:
            raise InvalidDamping("`lm_damping` must be >= 0")
        self.cost = cost
        self.gain = gain
        self.lm_damping = lm_damping

    # -- lowering to the C ABI ------------------------------------------------
    def _spec(self, model) -> dict:
        raise NotImplementedError

    def _key(self, model):
        """Hashable value of ``_spec``: problems are cached by it. Subclasses on the
        per-step path override it with something cheaper than building the descriptor."""
        from ._problems import _signature
        return _signature(self._spec(model))

    def _target_width(self) -> int:
        return 0

    def _target_tensor(self, configuration) -> Optional[torch.Tensor]:
        return None

    def _k(self, model) -> int:
        raise NotImplementedError

    def _problem(self, configuration):
        from ._problems import get_problem
        return get_problem(configuration, [self], [])

    def compute_error(self, configuration):
        prob, targets = self._problem(configuration)
        err, _ = prob.task_eval(0, self._k(configuration.model), configuration.q_tensor, targets)
        return configuration._out(err)

    def compute_jacobian(self, configuration):
        prob, targets = self._problem(configuration)
        _, jac = prob.task_eval(0, self._k(configuration.model), configuration.q_tensor, targets)
        return configuration._out(jac)

    def compute_qp_objective(self, configuration) -> Objective:
        if self._is_custom():
            H, c = self._generic_objective(configuration)
            return Objective(configuration._out(H), configuration._out(c))
        prob, targets = self._problem(configuration)
        P, c, _, _ = prob.build_qp(configuration.q_tensor, targets, 1.0, 0.0)
        return Objective(configuration._out(P), configuration._out(c))

    # -- user-defined tasks ----------------------------------------------------
    def _is_custom(self) -> bool:
        """True for a subclass that brings its own compute_error / compute_jacobian instead
        of a descriptor the kernels evaluate (the Task seam of mink/tasks/task.py:81-103)."""
        return type(self)._spec is Task._spec

    def _generic_objective(self, configuration):
        """(H [B, nv, nv], c [B, nv]) device tensors from the subclass's error e [.., k] and
        Jacobian J [.., k, nv] (numpy or tensors, with or without the batch axis), by the
        formula of mink/tasks/task.py:105-138: Jw = diag(cost) J, ew = diag(cost)(-gain e),
        H = Jw'Jw + lm_damping |ew|^2 I, c = -ew'Jw."""
        dev, B, nv = configuration.device, configuration.batch_size, configuration.model.nv

        def dev_t(a, tail):
            t = torch.as_tensor(a, dtype=torch.float64).to(dev)
            if t.ndim == tail:
                t = t[None].expand((B,) + tuple(t.shape))
            return t

        J = dev_t(self.compute_jacobian(configuration), 2)
        e = dev_t(self.compute_error(configuration), 1)
        if J.shape[0] != B or e.shape[0] != B or J.shape[2] != nv or J.shape[1] != e.shape[1]:
            raise TaskDefinitionError(f"{self.__class__.__name__}: compute_error / compute_jacobian must return "
                                      f"[.., k] and [.., k, {nv}] for {B} configuration(s)")
        cost = torch.as_tensor(np.broadcast_to(np.atleast_1d(self.cost), (e.shape[1],)).copy(),
                               dtype=torch.float64).to(dev)
        Jw = cost[None, :, None] * J
        ew = cost[None, :] * (-self.gain * e)
        mu = self.lm_damping * (ew * ew).sum(dim=1)
        H = Jw.transpose(1, 2) @ Jw + mu[:, None, None] * torch.eye(nv, dtype=torch.float64, device=dev)
        c = -torch.einsum("bk,bkn->bn", ew, Jw)
        return H, c


def _pose_target(task, pose: SE3, configuration) -> torch.Tensor:
    """[B, 7] device tensor of a pose target; the upload is kept until the task is given
    another target object (set_target stores a copy, so a new target is a new object)."""
    if pose is None:
        raise TargetNotSet(task.__class__.__name__)
    B = configuration.batch_size
    cached = task.__dict__.get("_target_upload")
    if cached is not None and cached[0] is pose and cached[1] == configuration.device and cached[2] == B:
        return cached[3]
    t = pose.wxyz_xyz.to(device=configuration.device, dtype=torch.float64)
    if t.ndim == 1:
        t = t[None, :].expand(B, -1)
    if tuple(t.shape) != (B, 7):
        raise InvalidTarget(f"Expected {B} target poses but got {tuple(t.shape)}")
    t = t.contiguous()
    if t.data_ptr() == pose.wxyz_xyz.data_ptr():
        return t   # the caller's own device tensor: it may change under us, so no caching
    task.__dict__["_target_upload"] = (pose, configuration.device, B, t)
    return t


class _FrameCosts(Task):
    k = 6

    def set_position_cost(self, position_cost) -> None:
        position_cost = np.atleast_1d(position_cost)
        if position_cost.ndim != 1 or position_cost.shape[0] not in (1, 3):
            raise TaskDefinitionError(
                f"{self.__class__.__name__} position cost should be a vector of shape 1 (aka identical cost for all "
                f"coordinates) or (3,) but got {position_cost.shape}")
        if not np.all(position_cost >= 0.0):
            raise TaskDefinitionError(f"{self.__class__.__name__} position cost should be >= 0")
        self.cost[:3] = position_cost

    def set_orientation_cost(self, orientation_cost) -> None:
        orientation_cost = np.atleast_1d(orientation_cost)
        if orientation_cost.ndim != 1 or orientation_cost.shape[0] not in (1, 3):
            raise TaskDefinitionError(
                f"{self.__class__.__name__} orientation cost should be a vector of shape 1 (aka identical cost for "
                f"all coordinates) or (3,) but got {orientation_cost.shape}")
        if not np.all(orientation_cost >= 0.0):
            raise TaskDefinitionError(f"{self.__class__.__name__} position cost should be >= 0")
        self.cost[3:] = orientation_cost

    def _k(self, model) -> int:
        return 6

    def _target_width(self) -> int:
        return 7


_FRAME_IDS: dict = {}


def _frame_ids_cached(model, name, ftype):
    key = (id(model), name, ftype)
    hit = _FRAME_IDS.get(key)
    if hit is None or hit[0] is not model:
        if len(_FRAME_IDS) > 4096:
            _FRAME_IDS.clear()
        hit = (model,) + _frame_ids(model, name, ftype)
        _FRAME_IDS[key] = hit
    return hit[1], hit[2]


def _frame_ids(model, name, ftype):
    if ftype not in consts.SUPPORTED_FRAMES:
        raise exceptions.UnsupportedFrame(ftype, consts.SUPPORTED_FRAMES)
    fid = {"body": model.body_id, "geom": model.geom_id, "site": model.site_id}[ftype](name)
    if fid == -1:
        raise exceptions.InvalidFrame(frame_name=name, frame_type=ftype, model=model)
    return consts.FRAME_TO_ENUM[ftype], fid


class FrameTask(_FrameCosts):
    """Regulate the pose of a frame in the world frame (mink/tasks/frame_task.py)."""

    def __init__(self, frame_name: str, frame_type: str, position_cost, orientation_cost, gain: float = 1.0,
                 lm_damping: float = 0.0):
        super().__init__(cost=np.zeros((self.k,)), gain=gain, lm_damping=lm_damping)
        self.frame_name = frame_name
        self.frame_type = frame_type
        self.position_cost = position_cost
        self.orientation_cost = orientation_cost
        self.transform_target_to_world: Optional[SE3] = None
        self.set_position_cost(position_cost)
        self.set_orientation_cost(orientation_cost)

    def set_target(self, transform_target_to_world: SE3) -> None:
        self.transform_target_to_world = transform_target_to_world.copy()

    def set_target_from_configuration(self, configuration) -> None:
        self.set_target(configuration.get_transform_frame_to_world(self.frame_name, self.frame_type))

    def _spec(self, model) -> dict:
        ft, fid = _frame_ids(model, self.frame_name, self.frame_type)
        return dict(type=L.TASK_FRAME, frame_type=ft, frame_id=fid, cost=self.cost.copy(), gain=self.gain,
                    lm_damping=self.lm_damping)

    def _key(self, model):
        ft, fid = _frame_ids_cached(model, self.frame_name, self.frame_type)
        return (L.TASK_FRAME, ft, fid, self.cost.tobytes(), self.gain, self.lm_damping)

    def _target_tensor(self, configuration):
        return _pose_target(self, self.transform_target_to_world, configuration)


class RelativeFrameTask(_FrameCosts):
    """Regulate the pose of a frame relative to another frame (mink/tasks/relative_frame_task.py)."""

    def __init__(self, frame_name: str, frame_type: str, root_name: str, root_type: str, position_cost,
                 orientation_cost, gain: float = 1.0, lm_damping: float = 0.0):
        super().__init__(cost=np.zeros((self.k,)), gain=gain, lm_damping=lm_damping)
        self.frame_name, self.frame_type = frame_name, frame_type
        self.root_name, self.root_type = root_name, root_type
        self.position_cost = position_cost
        self.orientation_cost = orientation_cost
        self.transform_target_to_root: Optional[SE3] = None
        self.set_position_cost(position_cost)
        self.set_orientation_cost(orientation_cost)

    def set_target(self, transform_target_to_root: SE3) -> None:
        self.transform_target_to_root = transform_target_to_root.copy()

    def set_target_from_configuration(self, configuration) -> None:
        self.set_target(configuration.get_transform(self.frame_name, self.frame_type, self.root_name, self.root_type))

    def _spec(self, model) -> dict:
        ft, fid = _frame_ids(model, self.frame_name, self.frame_type)
        rt, rid = _frame_ids(model, self.root_name, self.root_type)
        return dict(type=L.TASK_RELATIVE_FRAME, frame_type=ft, frame_id=fid, root_type=rt, root_id=rid,
                    cost=self.cost.copy(), gain=self.gain, lm_damping=self.lm_damping)

    def _key(self, model):
        ft, fid = _frame_ids_cached(model, self.frame_name, self.frame_type)
        rt, rid = _frame_ids_cached(model, self.root_name, self.root_type)
        return (L.TASK_RELATIVE_FRAME, ft, fid, rt, rid, self.cost.tobytes(), self.gain, self.lm_damping)

    def _target_tensor(self, configuration):
        return _pose_target(self, self.transform_target_to_root, configuration)


class PostureTask(Task):
    """Regulate joint angles towards a posture; floating-base dofs are unaffected
    (mink/tasks/posture_task.py)."""

    def __init__(self, model, cost, gain: float = 1.0, lm_damping: float = 0.0):
        super().__init__(cost=np.zeros((model.nv,)), gain=gain, lm_damping=lm_damping)
        self.target_q: Optional[np.ndarray] = None
        self.k = model.nv
        self.nq = model.nq
        self.set_cost(cost)

    def set_cost(self, cost) -> None:
        cost = np.atleast_1d(cost)
        if cost.ndim != 1 or cost.shape[0] not in (1, self.k):
            raise TaskDefinitionError(
                f"{self.__class__.__name__} cost must be a vector of shape (1,) (aka identical cost for all dofs) "
                f"or ({self.k},). Got {cost.shape}")
        if not np.all(cost >= 0.0):
            raise TaskDefinitionError(f"{self.__class__.__name__} cost should be >= 0")
        self.cost[: self.k] = cost

    def set_target(self, target_q) -> None:
        if isinstance(target_q, torch.Tensor):
            target_q = target_q.detach().cpu().numpy()
        target_q = np.atleast_1d(target_q)
        if target_q.ndim != 1 or target_q.shape[0] != (self.nq):
            raise InvalidTarget(f"Expected target posture to have shape ({self.nq},) but got {target_q.shape}")
        self.target_q = target_q.copy()

    def set_target_from_configuration(self, configuration) -> None:
        q = configuration.q
        if configuration.batched:
            if configuration.batch_size != 1:
                raise InvalidTarget("the posture target is shared by the batch: pass a single configuration")
            q = q[0]
        self.set_target(q)

    def _k(self, model) -> int:
        return model.nv

    def _key(self, model):
        if self.target_q is None:
            raise TargetNotSet(self.__class__.__name__)
        # the target is per-call data of the device problem (mk_problem_set_posture_target): a task
        # whose target changes every tick keeps its problem
        return (L.TASK_POSTURE, self.cost.tobytes(), self.gain, self.lm_damping)

    def _posture_target(self) -> np.ndarray:
        return np.ascontiguousarray(self.target_q, dtype=np.float64)

    def _spec(self, model) -> dict:
        if self.target_q is None:
            raise TargetNotSet(self.__class__.__name__)
        return dict(type=L.TASK_POSTURE, cost=self.cost.copy(), target_q=np.asarray(self.target_q, dtype=np.float64),
                    gain=self.gain, lm_damping=self.lm_damping)


class DampingTask(PostureTask):
    """Minimise joint velocities: a posture task with gain 0 and target qpos0
    (mink/tasks/damping_task.py)."""

    def __init__(self, model, cost):
        super().__init__(model=model, cost=cost, gain=0.0, lm_damping=0.0)
        self.target_q = model.qpos0


class ComTask(Task):
    """Regulate the centre of mass of the subtree of body 1 (mink/tasks/com_task.py)."""
    k = 3

    def __init__(self, cost, gain: float = 1.0, lm_damping: float = 0.0):
        super().__init__(cost=np.zeros((self.k,)), gain=gain, lm_damping=lm_damping)
        self.target_com = None
        self.set_cost(cost)

    def set_cost(self, cost) -> None:
        cost = np.atleast_1d(cost)
        if cost.ndim != 1 or cost.shape[0] not in (1, self.k):
            raise TaskDefinitionError(
                f"{self.__class__.__name__} cost must be a vector of shape (1,) (aka identical cost for all "
                f"coordinates) or ({self.k},). Got {cost.shape}")
        if not np.all(cost >= 0.0):
            raise TaskDefinitionError(f"{self.__class__.__name__} cost must be >= 0")
        self.cost[:] = cost

    def set_target(self, target_com) -> None:
        if isinstance(target_com, torch.Tensor):
            target_com = target_com.detach().cpu().numpy()
        target_com = np.atleast_1d(target_com)
        if target_com.shape[-1] != self.k or target_com.ndim > 2:
            raise InvalidTarget(f"Expected target CoM to have shape ({self.k},) but got {target_com.shape}")
        self.target_com = target_com.copy()

    def set_target_from_configuration(self, configuration) -> None:
        saved, self.target_com = self.target_com, np.zeros(3)
        try:
            com = self.compute_error(configuration)   # error with a zero target is the CoM itself
        finally:
            self.target_com = saved
        self.set_target(com.cpu().numpy() if isinstance(com, torch.Tensor) else com)

    def _k(self, model) -> int:
        return 3

    def _target_width(self) -> int:
        return 3

    def _spec(self, model) -> dict:
        return dict(type=L.TASK_COM, cost=self.cost.copy(), gain=self.gain, lm_damping=self.lm_damping)

    def _target_tensor(self, configuration):
        if self.target_com is None:
            raise TargetNotSet(self.__class__.__name__)
        t = torch.tensor(np.asarray(self.target_com, dtype=np.float64), device=configuration.device)
        if t.ndim == 1:
            t = t[None, :].expand(configuration.batch_size, -1)
        return t
