"""Batched ``Configuration`` (reference: mink/configuration.py:21-253).

Holds B configurations of one kinematic tree on the GPU.  Constructed from a 1-D ``q``
it behaves like the reference object (numpy in, numpy out, no batch axis); constructed
from a ``[B, nq]`` array or tensor every quantity gains a leading batch axis and stays
in device memory.  Kinematics (frame poses, Jacobians, integration, limit check) run in
the HIP kernels through the C ABI; there is no host fallback.
"""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np
import torch

from . import constants as consts
from . import exceptions
from .engine import DeviceModel
from .lie import SE3, SO3
from .mjcf import JNT_FREE, Model

def device_model(model: Model, device) -> DeviceModel:
    """One uploaded copy of a model per (model object, device).  The copies are kept on the
    model object itself, so they (and their device tables and cached problems) go away with
    the model instead of living as long as the process."""
    cache = model.__dict__.setdefault("_device_models", {})
    dm = cache.get(str(device))
    if dm is None:
        dm = DeviceModel(model, device)
        dm.problem_cache = {}
        cache[str(device)] = dm
    return dm


class MocapData:
    """The part of ``mujoco.MjData`` the reference's callers touch next to the IK path: the
    poses of the mocap bodies that carry teleoperation targets
    (interactive_scripts/teleop_mouse.py:88-99, mink/utils.py:10-35, mink/lie/se3.py:77-91).
    ``mocap_pos`` [.., nmocap, 3] and ``mocap_quat`` [.., nmocap, 4] (wxyz) are numpy arrays
    for a single configuration and device tensors with a leading batch axis otherwise."""

    def __init__(self, configuration: "Configuration"):
        self.configuration = configuration
        self.reset()

    def reset(self) -> None:
        cfg, m = self.configuration, self.configuration.model
        ids = [b for b in range(m.nbody) if m.body_mocapid[b] >= 0]
        ids.sort(key=lambda b: m.body_mocapid[b])
        pos = np.array([m.body_pos[b] for b in ids], dtype=np.float64).reshape(len(ids), 3)
        quat = np.array([m.body_quat[b] for b in ids], dtype=np.float64).reshape(len(ids), 4)
        if cfg.batched:
            B = cfg.batch_size
            self.mocap_pos = torch.tensor(pos, device=cfg.device).repeat(B, 1, 1)
            self.mocap_quat = torch.tensor(quat, device=cfg.device).repeat(B, 1, 1)
        else:
            self.mocap_pos, self.mocap_quat = pos, quat


class Configuration:
    def __init__(self, model: Model, q=None, device="cuda:0"):
        self.model = model
        self.device = torch.device(device)
        self.dmodel = device_model(model, self.device)
        self.batched = False
        self._q = torch.tensor(model.qpos0[None, :], dtype=torch.float64, device=self.device)
        self.update(q)
        self.data = MocapData(self)

    # ---- state -------------------------------------------------------------
    def update(self, q=None) -> None:
        """Set the configuration(s). Kinematics are evaluated on demand by the kernels, so
        unlike the reference there is no FK cache to refresh."""
        if q is None:
            return
        prev = (self.batched, self._q.shape[0])
        if isinstance(q, torch.Tensor):
            t = q.detach().to(device=self.device, dtype=torch.float64)
        else:
            t = torch.tensor(np.asarray(q, dtype=np.float64), device=self.device)
        if t.ndim == 1:
            self.batched = False
            t = t[None, :]
        elif t.ndim == 2:
            self.batched = True
        else:
            raise ValueError(f"q must have shape ({self.model.nq},) or (B, {self.model.nq})")
        if t.shape[1] != self.model.nq:
            raise ValueError(f"q must have {self.model.nq} entries per configuration, got {t.shape[1]}")
        self._q = t.contiguous().clone()
        if hasattr(self, "data") and prev != (self.batched, self._q.shape[0]):
            self.data.reset()   # the mocap arrays follow the batch shape

    def update_from_keyframe(self, key_name: str) -> None:
        key_id = self.model.key_id(key_name)
        if key_id == -1:
            raise exceptions.InvalidKeyframe(key_name, self.model)
        self.update(q=self.model.key_qpos[key_id])

    @property
    def batch_size(self) -> int:
        return self._q.shape[0]

    def _out(self, t: torch.Tensor):
        return t if self.batched else t[0].cpu().numpy()

    @property
    def q(self):
        """Copy of the configuration vector(s)."""
        return self._out(self._q.clone())

    @property
    def q_tensor(self) -> torch.Tensor:
        """The device tensor [B, nq] itself (no copy)."""
        return self._q

    @property
    def nv(self) -> int:
        return self.model.nv

    @property
    def nq(self) -> int:
        return self.model.nq

    # ---- limits ------------------------------------------------------------
    def check_limits(self, tol: float = 1e-6, safety_break: bool = True) -> None:
        """Raise (or warn) when a limited non-free joint is outside [lower - tol, upper + tol].
        For a batch, the first offending configuration is reported."""
        first = self.dmodel.first_limit_violation(self._q, tol)
        bad = torch.nonzero(first >= 0)
        if bad.numel() == 0:
            return
        b = int(bad[0, 0])
        self._report_limit_violation(b, int(first[b]), bad.shape[0], safety_break)

    # ---- reports of the batched calls -----------------------------------------------
    # By default a batched solve_ik / solve_ik_iterate / move_to keeps the reference's contract:
    # the limit warning and the assert on a failed QP happen inside the call (which therefore
    # waits for the device).  Inside `with configuration.deferred_checks():` (or with
    # `configuration.defer_checks = True`) the calls queue their work and return at once; the
    # reports accumulate on the device and `check()` -- called explicitly, or on leaving the
    # `with` block -- delivers them.
    defer_checks = False

    def deferred_checks(self):
        """Context manager: batched calls inside it do not wait for the device; their limit
        warnings and solver failures are reported when the block ends (or by `check()`)."""
        cfg = self

        class _Deferred:
            def __enter__(self_inner):
                self_inner.saved = cfg.defer_checks
                cfg.defer_checks = True
                return cfg

            def __exit__(self_inner, exc_type, exc, tb):
                cfg.defer_checks = self_inner.saved
                if exc_type is None:
                    cfg.check()
                return False

        return _Deferred()

    def _defer_limit_check(self, tol: float = 1e-6) -> None:
        """Queue the limit check on the stream and keep its result on the device; nothing
        waits for it.  Per configuration the earliest violation since the last `check()` is
        kept, which `check()` turns into the reference's warning."""
        first = self.dmodel.first_limit_violation(self._q, tol)
        acc = getattr(self, "_pending_first", None)
        self._pending_first = first if acc is None or acc.shape != first.shape else torch.where(acc >= 0, acc, first)

    def _report_status(self, status: torch.Tensor) -> None:
        """The reference's `assert dq is not None` (mink/solve_ik.py:103) for a batch: now, or
        folded into the deferred reports."""
        if self.defer_checks:
            self._defer_status(status)
            return
        n_bad = int((status != 0).sum())
        assert n_bad == 0, (f"QP solver failed for {n_bad} of {status.shape[0]} problems (infeasible limits or "
                            f"non-finite input); worst status {int(status.max())}")

    def _defer_status(self, status: torch.Tensor) -> None:
        """Fold the solver status of a batched call into a running per-problem maximum on
        the device (one small element-wise kernel, no synchronisation)."""
        acc = getattr(self, "_pending_status", None)
        self._pending_status = status.clone() if acc is None or acc.shape != status.shape \
            else torch.maximum(acc, status)

    def check(self) -> None:
        """Wait for the batched calls issued so far and report what they deferred: a warning
        for the limit checks of `solve_ik(..., safety_break=False)` since the last `check()`
        (the reference's logging.warning, mink/configuration.py:107-110) and an AssertionError
        if any QP since then failed (the reference's assert, mink/solve_ik.py:103)."""
        first = getattr(self, "_pending_first", None)
        self._pending_first = None
        if first is not None and first.shape[0] == self.batch_size:
            bad = torch.nonzero(first >= 0)
            if bad.numel():
                b = int(bad[0, 0])
                self._report_limit_violation(b, int(first[b]), bad.shape[0], False)
        status = getattr(self, "_pending_status", None)
        self._pending_status = None
        if status is not None:
            n_bad = int((status != 0).sum())
            assert n_bad == 0, (f"QP solver failed for {n_bad} of {status.shape[0]} problems (infeasible limits or "
                                f"non-finite input); worst status {int(status.max())}")

    def _report_limit_violation(self, b: int, jnt: int, n_bad: int, safety_break: bool) -> None:
        qval = float(self._q[b, self.model.jnt_qposadr[jnt]])
        qmin, qmax = (float(x) for x in self.model.jnt_range[jnt])
        if safety_break:
            raise exceptions.NotWithinConfigurationLimits(joint_id=jnt, value=qval, lower=qmin, upper=qmax,
                                                          model=self.model)
        extra = f" (configuration {b}; {n_bad} of {self.batch_size} configurations violate limits)" \
            if self.batched else ""
        logging.warning(f"Value {qval:.2f} at index {jnt} is outside of its limits: [{qmin:.2f}, {qmax:.2f}]{extra}")

    # ---- frames ------------------------------------------------------------
    def _frame_id(self, frame_name: str, frame_type: str) -> int:
        if frame_type not in consts.SUPPORTED_FRAMES:
            raise exceptions.UnsupportedFrame(frame_type, consts.SUPPORTED_FRAMES)
        fid = {"body": self.model.body_id, "geom": self.model.geom_id, "site": self.model.site_id}[frame_type](frame_name)
        if fid == -1:
            raise exceptions.InvalidFrame(frame_name=frame_name, frame_type=frame_type, model=self.model)
        return fid

    def _mocap_frame(self, frame_type: str, fid: int):
        """(mocap id, pose of the frame in its body) when the frame sits on a mocap body, else
        None.  Such a body is placed by ``data.mocap_pos / mocap_quat`` (mj_kinematics reads
        them), which live on the host side of this API: its frames are composed here and never
        reach the kernels (a task or limit that names one is refused by the library)."""
        m = self.model
        body = {"body": lambda i: i, "geom": lambda i: int(m.geom_bodyid[i]), "site": lambda i: int(m.site_bodyid[i])}[frame_type](fid)
        mid = int(m.body_mocapid[body])
        if mid < 0:
            return None
        local = np.array([1.0, 0, 0, 0, 0, 0, 0])
        if frame_type == "geom":
            local = np.concatenate([m.geom_quat[fid], m.geom_pos[fid]])
        elif frame_type == "site":
            local = np.concatenate([m.site_quat[fid], m.site_pos[fid]])
        return mid, local

    def get_frame_jacobian(self, frame_name: str, frame_type: str):
        """Body-frame Jacobian [.., 6, nv], rows (linear; angular)."""
        fid = self._frame_id(frame_name, frame_type)
        if self._mocap_frame(frame_type, fid) is not None:      # no joint moves a mocap body
            return self._out(torch.zeros((self.batch_size, 6, self.model.nv), dtype=torch.float64, device=self.device))
        return self._out(self.dmodel.frame_jacobian(self._q, consts.FRAME_TO_ENUM[frame_type], fid))

    def get_transform_frame_to_world(self, frame_name: str, frame_type: str) -> SE3:
        fid = self._frame_id(frame_name, frame_type)
        mocap = self._mocap_frame(frame_type, fid)
        if mocap is not None:
            mid, local = mocap
            body = SE3.from_rotation_and_translation(SO3(torch.as_tensor(self.data.mocap_quat[..., mid, :])),
                                                     torch.as_tensor(self.data.mocap_pos[..., mid, :]))
            return body @ SE3(torch.as_tensor(local).to(body.wxyz_xyz.device))
        pose = self.dmodel.frame_pose(self._q, consts.FRAME_TO_ENUM[frame_type], fid)
        return SE3(pose if self.batched else pose[0].cpu())

    def get_transform(self, source_name: str, source_type: str, dest_name: str, dest_type: str) -> SE3:
        T_src = self.get_transform_frame_to_world(source_name, source_type)
        T_dst = self.get_transform_frame_to_world(dest_name, dest_type)
        return T_dst.inverse() @ T_src

    # ---- integration ---------------------------------------------------------
    def _velocity(self, velocity) -> torch.Tensor:
        if isinstance(velocity, torch.Tensor):
            v = velocity.detach().to(device=self.device, dtype=torch.float64)
        else:
            v = torch.tensor(np.asarray(velocity, dtype=np.float64), device=self.device)
        if v.ndim == 1:
            v = v[None, :].expand(self.batch_size, -1)
        if tuple(v.shape) != (self.batch_size, self.model.nv):
            raise ValueError(f"velocity must have shape (.., {self.model.nv})")
        return v.contiguous()

    def integrate(self, velocity, dt: float):
        """q (+) dt * v without modifying the configuration."""
        q = self._q.clone()
        self.dmodel.integrate_(q, self._velocity(velocity), dt)
        return self._out(q)

    def integrate_inplace(self, velocity, dt: float) -> None:
        self.dmodel.integrate_(self._q, self._velocity(velocity), dt)
