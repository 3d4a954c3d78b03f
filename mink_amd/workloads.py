"""Synthetic TidyBot whole-body IK workloads (inputs only; no arithmetic of the path).

Task and limit sets follow the production callers of the reference
(interactive_scripts/teleop_mouse.py:46-69, envs/mj_env.py:92-107): FrameTask on
``pinch_site`` (costs 1, lm_damping 1), PostureTask (cost 1e-3 on dofs >= 3, target =
home keyframe), optional DampingTask (cost 100 on the base dofs), ConfigurationLimit
and a synthetic VelocityLimit.  Used by tests and bench.py.
"""

from __future__ import annotations

import numpy as np

from . import _lib as L
from .mjcf import JNT_FREE

DT = 1.0 / 200.0
DAMPING = 1e-3
VELOCITY_LIMITS = {
    "joint_x": 0.5, "joint_y": 0.5, "joint_th": 1.0,
    "joint_1": 1.39, "joint_2": 1.39, "joint_3": 1.39, "joint_4": 1.39,
    "joint_5": 1.22, "joint_6": 1.22, "joint_7": 1.22,
}


def home_q(model) -> np.ndarray:
    return model.key_qpos[model.key_id("home")].copy()


def posture_cost(model) -> np.ndarray:
    c = np.zeros(model.nv)
    c[3:] = 1e-3
    return c


def base_damping_cost(model) -> np.ndarray:
    c = np.zeros(model.nv)
    c[:3] = 100.0
    return c


def task_specs(model, with_damping: bool, site: str = "pinch_site"):
    specs = [
        dict(type=L.TASK_FRAME, frame_type=L.FRAME_TYPES["site"], frame_id=model.site_id(site),
             cost=np.ones(6), gain=1.0, lm_damping=1.0),
        dict(type=L.TASK_POSTURE, cost=posture_cost(model), target_q=home_q(model), gain=1.0, lm_damping=0.0),
    ]
    if with_damping:
        specs.append(dict(type=L.TASK_POSTURE, cost=base_damping_cost(model), target_q=model.qpos0.copy(),
                          gain=0.0, lm_damping=0.0))
    return specs


def limit_specs(model, with_velocity: bool):
    specs = [dict(type=L.LIMIT_CONFIGURATION, gain=0.95, min_distance_from_limits=0.0)]
    if with_velocity:
        idx = [int(model.jnt_dofadr[model.joint_id(n)]) for n in VELOCITY_LIMITS]
        specs.append(dict(type=L.LIMIT_VELOCITY, index=idx, vmax=list(VELOCITY_LIMITS.values())))
    return specs


def sample_q(model, rng: np.random.Generator, n: int) -> np.ndarray:
    """home (+) perturbation: base x,y ~ U(-1,1) m, theta ~ U(-pi,pi); arm joints
    ~ U(-0.5,0.5) rad around home, clipped 0.05 rad inside their limits; fingers 0."""
    home = home_q(model)
    q = np.tile(home, (n, 1))
    q[:, 0:2] += rng.uniform(-1.0, 1.0, (n, 2))
    q[:, 2] += rng.uniform(-np.pi, np.pi, n)
    q[:, 3:10] += rng.uniform(-0.5, 0.5, (n, 7))
    for j in range(min(model.njnt, 10)):
        if model.jnt_limited[j] and model.jnt_type[j] != JNT_FREE:
            a = model.jnt_qposadr[j]
            q[:, a] = np.clip(q[:, a], model.jnt_range[j, 0] + 0.05, model.jnt_range[j, 1] - 0.05)
    return q


def offset_targets(poses: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """T_wb (+) exp(xi) with xi_v ~ U(-0.1,0.1) m, xi_w ~ U(-0.3,0.3) rad (numpy, host)."""
    n = poses.shape[0]
    xi_v = rng.uniform(-0.1, 0.1, (n, 3))
    xi_w = rng.uniform(-0.3, 0.3, (n, 3))
    th = np.linalg.norm(xi_w, axis=1, keepdims=True)
    dq = np.concatenate([np.cos(0.5 * th), np.sin(0.5 * th) * xi_w / th], axis=1)
    w, x, y, z = poses[:, 0], poses[:, 1], poses[:, 2], poses[:, 3]
    R = np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
        np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
        np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], 1)
    t = poses[:, 4:] + np.einsum("nij,nj->ni", R, xi_v)
    a, b = poses[:, :4], dq
    qn = np.stack([
        a[:, 0] * b[:, 0] - a[:, 1] * b[:, 1] - a[:, 2] * b[:, 2] - a[:, 3] * b[:, 3],
        a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0] + a[:, 2] * b[:, 3] - a[:, 3] * b[:, 2],
        a[:, 0] * b[:, 2] - a[:, 1] * b[:, 3] + a[:, 2] * b[:, 0] + a[:, 3] * b[:, 1],
        a[:, 0] * b[:, 3] + a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1] + a[:, 3] * b[:, 0]], -1)
    return np.concatenate([qn, t], axis=1)


# ---- benchmark config 4: Frame + RelativeFrame + Posture, Configuration + Velocity +
# CollisionAvoidance over the primitive pairs {gripper pads} x {cube, floor} ----
PADS = ("right_pad1", "right_pad2", "left_pad1", "left_pad2")


def config4_specs(model):
    from .limits import CollisionAvoidanceLimit
    rel = dict(type=L.TASK_RELATIVE_FRAME, frame_type=L.FRAME_TYPES["site"], frame_id=model.site_id("pinch_site"),
               root_type=L.FRAME_TYPES["body"], root_id=model.body_id("base_link"), cost=np.ones(6), gain=1.0,
               lm_damping=0.0)
    base = task_specs(model, False)
    coll = CollisionAvoidanceLimit(model, [(list(PADS), ["cube_geom", "floor"])])
    return [base[0], rel, base[1]], limit_specs(model, True) + [coll._spec(model)]


def config4_inputs(model, dmodel, n, seed, near_fraction=0.25):
    """(q [n, nq], targets [n, 14]) on the device: start configurations with the cube placed
    within 2 cm of a gripper pad in `near_fraction` of the problems (so that collision rows
    are active), world and base-relative pose targets from an independent sample."""
    import torch
    rng = np.random.default_rng(seed)
    q = sample_q(model, rng, n)
    qs = sample_q(model, rng, n)
    dev = dmodel.device
    tq = torch.tensor(q, device=dev)
    k = int(n * near_fraction)
    if k:
        pad_ids = np.array([model.geom_id(p) for p in PADS])
        pick = rng.integers(0, 4, k)
        pos = np.empty((k, 3))
        for j, gid in enumerate(pad_ids):
            sel = np.nonzero(pick == j)[0]
            if len(sel):
                pos[sel] = dmodel.frame_pose(tq[sel].contiguous(), L.FRAME_TYPES["geom"], int(gid))[:, 4:].cpu().numpy()
        d = rng.normal(size=(k, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        free_adr = int(model.jnt_qposadr[model.joint_id("interactive_obj_freejoint")])
        q[:k, free_adr:free_adr + 3] = pos + d * rng.uniform(0.03, 0.05, (k, 1))
        qq = rng.normal(size=(k, 4))
        q[:k, free_adr + 3:free_adr + 7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
        tq = torch.tensor(q, device=dev)
    ts = torch.tensor(qs, device=dev)
    site, base = model.site_id("pinch_site"), model.body_id("base_link")
    t_world = dmodel.frame_pose(ts, L.FRAME_TYPES["site"], site)
    t_base = dmodel.frame_pose(ts, L.FRAME_TYPES["body"], base)
    from .lie import SE3
    t_rel = (SE3(t_base).inverse() @ SE3(t_world)).wxyz_xyz
    return tq, torch.cat([t_world, t_rel], dim=1).contiguous()
