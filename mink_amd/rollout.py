"""Batched waypoint reaching on top of the iterated IK kernel: the kinematic part of the
reference's ``move_to`` (envs/mj_env.py:193-251) with its waypoint generator
``LinearWaypointReach`` (envs/robot_utils.py:113-160).

Every outer step reads the end-effector pose, moves the commanded pose a bounded linear
step towards the goal position and half-way (slerp 0.5) towards the goal orientation,
and runs the callers' IK loop (up to ``max_iters`` x solve_ik + integrate_inplace with the
1e-4 / 1e-4 exit test, envs/mj_env.py:266-276) as ONE kernel launch for the whole batch.
The reference then hands the joint targets to the MuJoCo simulation; physics is outside
this package, so the configuration is taken to track its IK solution exactly.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from .lie import SE3, SO3
from .solve_ik import solve_ik_iterate


@dataclass
class LinearWaypointReachConfig:
    """envs/robot_utils.py:113-118."""
    pos_threshold: float = 0.01
    pos_step_size: float = 0.1
    rot_threshold: float = 0.02
    rot_step_size: float = 0.1


def _euler_xyz(rot: SO3) -> torch.Tensor:
    """Extrinsic x-y-z Euler angles [.., 3] (scipy's ``as_euler("xyz")``)."""
    return torch.stack(rot.as_rpy_radians(), dim=-1)


class LinearWaypointReach:
    """Waypoint generator towards goal poses ``target_pos`` [B, 3] and ``target_rot``
    (SO3 with batch B); ``step`` is the batched form of envs/robot_utils.py:131-160."""

    def __init__(self, target_pos, target_rot: SO3, cfg: LinearWaypointReachConfig):
        self.target_pos = torch.as_tensor(target_pos, dtype=torch.float64)
        self.target_rot = target_rot
        self.cfg = cfg

    @classmethod
    def from_euler(cls, target_pos, target_euler, cfg: LinearWaypointReachConfig):
        e = torch.as_tensor(target_euler, dtype=torch.float64)
        return cls(target_pos, SO3.from_rpy_radians(e[..., 0], e[..., 1], e[..., 2]), cfg)

    def step(self, curr_pos: torch.Tensor, curr_rot: SO3):
        """-> (commanded position [B, 3], commanded rotation SO3, reached [B] bool)."""
        delta = self.target_pos.to(curr_pos.device) - curr_pos
        dist = torch.linalg.norm(delta, dim=-1, keepdim=True)
        pos_reached = dist[..., 0] < self.cfg.pos_threshold
        step = torch.clamp(dist, max=self.cfg.pos_step_size)
        moved = curr_pos + delta / torch.clamp(dist, min=1e-300) * step
        abs_pos = torch.where(pos_reached[..., None], self.target_pos.to(curr_pos.device), moved)
        target = SO3(self.target_rot.wxyz.to(curr_rot.wxyz.device))
        # half-way along the geodesic from the current to the goal orientation
        rel = curr_rot.inverse() @ target
        abs_rot = curr_rot @ SO3.exp(0.5 * rel.log())
        rot_err = _euler_xyz(target @ curr_rot.inverse())
        rot_reached = torch.linalg.norm(rot_err, dim=-1) < self.cfg.rot_threshold
        return abs_pos, abs_rot, pos_reached & rot_reached


def move_to(configuration, frame_task, tasks: Sequence, target_pos, target_rot: SO3,
            cfg: Optional[LinearWaypointReachConfig] = None, dt: float = 1.0 / 200.0, solver: str = "quadprog",
            damping: float = 1e-3, max_steps: int = 50, max_iters: int = 20, pos_threshold: float = 1e-4,
            ori_threshold: float = 1e-4, limits=None) -> dict:
    """Drive the frame of ``frame_task`` (which must be one of ``tasks``) of every
    configuration of the batch to its goal pose.  A problem stops after the step in which
    its waypoint generator reports the goal reached, like the ``break`` of the reference's
    loop; the others carry on for at most ``max_steps`` steps.  Updates ``configuration``
    in place and returns ``dict(steps [B], reached [B], solves [B])``."""
    if not configuration.batched:
        raise ValueError("move_to works on a batched Configuration (q of shape [B, nq])")
    cfg = cfg or LinearWaypointReachConfig()
    dev = configuration.device
    B = configuration.batch_size
    if not any(t is frame_task for t in tasks):
        raise ValueError("frame_task must be one of tasks")
    # One launch for the whole rollout when the fused kernels serve the task / limit set and
    # the driven task is their (single) frame task; one library call queueing the passes of
    # every step for other problems; the host loop below remains as the form both are checked
    # against (and for a driven task that is not the problem's first frame task).
    from ._problems import get_problem
    from .solve_ik import _check_solver, _resolve_limits
    from .tasks import FrameTask
    _check_solver(solver)
    goal = SE3.from_rotation_and_translation(SO3(target_rot.wxyz.to(dev)),
                                             torch.as_tensor(target_pos, dtype=torch.float64).to(dev))
    if type(frame_task) is FrameTask:
        saved = frame_task.transform_target_to_world
        frame_task.transform_target_to_world = goal        # any valid pose: only its shape matters here
        try:
            prob, targets = get_problem(configuration, tasks, _resolve_limits(configuration, limits))
        finally:
            frame_task.transform_target_to_world = saved
        goals = goal.wxyz_xyz.to(dtype=torch.float64)
        if goals.ndim == 1:
            goals = goals[None, :].expand(B, -1)
        if prob.fused and prob.target_stride == 7:
            res = prob.move_to_(configuration.q_tensor, goals.contiguous(), cfg, dt, damping, max_steps, max_iters,
                                pos_threshold, ori_threshold)
            configuration._report_status(res["status"])     # the reference's assert on a failed QP
            return dict(steps=res["steps"], reached=res["reached"].bool(), solves=res["solves"].to(torch.int64))
        # Any other problem whose first frame task is the driven one (collision limits, relative
        # frame tasks, other trees): the whole rollout as launches of the general loop's kernels,
        # queued by one library call -- no host round trip per step.
        index = next(i for i, t in enumerate(tasks) if t is frame_task)
        first_frame = next(i for i, t in enumerate(tasks) if type(t) is FrameTask)
        if not prob.fused and index == first_frame:
            if targets.shape[0] != B:
                targets = targets.expand(B, -1)
            res = prob.move_to_passes_(index, configuration.q_tensor, goals.contiguous(), targets.contiguous().clone(), cfg,
                                       dt, damping, max_steps, max_iters, pos_threshold, ori_threshold)
            configuration._report_status(res["status"])
            return dict(steps=res["steps"], reached=res["reached"].bool(), solves=res["solves"].to(torch.int64))
    return _move_to_stepwise(configuration, frame_task, tasks, target_pos, target_rot, cfg, dt, solver, damping,
                             max_steps, max_iters, pos_threshold, ori_threshold, limits)


def _move_to_stepwise(configuration, frame_task, tasks, target_pos, target_rot, cfg, dt, solver, damping, max_steps,
                      max_iters, pos_threshold, ori_threshold, limits) -> dict:
    """move_to as a host loop over outer steps (waypoint generator in torch, one iterated IK
    launch per step): serves any task / limit set, and is the form the fused launch is
    checked against."""
    dev = configuration.device
    B = configuration.batch_size
    wpr = LinearWaypointReach(torch.as_tensor(target_pos, dtype=torch.float64).to(dev),
                              SO3(target_rot.wxyz.to(dev)), cfg)
    finished = torch.zeros(B, dtype=torch.bool, device=dev)
    reached = torch.zeros(B, dtype=torch.bool, device=dev)
    steps = torch.zeros(B, dtype=torch.int32, device=dev)
    solves = torch.zeros(B, dtype=torch.int64, device=dev)
    q = configuration.q_tensor
    for _ in range(max_steps):
        pose = configuration.get_transform_frame_to_world(frame_task.frame_name, frame_task.frame_type)
        pos_cmd, rot_cmd, now = wpr.step(pose.translation(), pose.rotation())
        frame_task.set_target(SE3.from_rotation_and_translation(rot_cmd, pos_cmd))
        q_before = q.clone()
        res = solve_ik_iterate(configuration, tasks, dt, solver, damping, max_iters=max_iters,
                               pos_threshold=pos_threshold, ori_threshold=ori_threshold, limits=limits)
        # problems that had already finished stay where they were
        q.copy_(torch.where(finished[:, None], q_before, q))
        live = ~finished
        steps += live.to(torch.int32)
        solves += torch.where(live, res["iters"].to(torch.int64), torch.zeros_like(solves))
        reached |= now & live
        finished |= now
        if bool(finished.all()):
            break
    return dict(steps=steps, reached=reached, solves=solves)
