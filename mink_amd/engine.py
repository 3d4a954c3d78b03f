"""Thin object layer over the C ABI: device-resident model and problem handles and
the batched solve / iterate / integrate / FK calls on torch ROCm tensors.

This is plumbing shared by the mink-named host API (configuration.py, tasks.py,
limits.py, solve_ik.py) and by bench.py; all arithmetic happens in the HIP kernels.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib as L


def _stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _check_tensor(t: torch.Tensor, shape, dtype, name: str) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a torch tensor on the GPU")
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous {dtype} tensor of shape {tuple(shape)}, "
                         f"got {t.dtype} {tuple(t.shape)}")


class DeviceModel:
    """A kinematic tree uploaded to the current GPU (``mk_model``)."""

    def __init__(self, model, device: Optional[torch.device] = None):
        self.lib = L.load_library()
        self.model = model
        self.device = torch.device(device if device is not None else "cuda:0")
        desc, keep = L.pack_model(model)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_model_create(C.byref(desc), C.byref(handle)))
        self.handle = handle

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.mk_model_destroy(h)
            self.handle = None

    def integrate_(self, q: torch.Tensor, v: torch.Tensor, dt: float) -> None:
        B = q.shape[0]
        _check_tensor(q, (B, self.model.nq), torch.float64, "q")
        _check_tensor(v, (B, self.model.nv), torch.float64, "v")
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_integrate_batch(self.handle, q.data_ptr(), v.data_ptr(), float(dt), B,
                                                _stream_ptr(self.device)))

    def frame_pose(self, q: torch.Tensor, frame_type: int, frame_id: int) -> torch.Tensor:
        B = q.shape[0]
        _check_tensor(q, (B, self.model.nq), torch.float64, "q")
        out = torch.empty((B, 7), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_fk_batch(self.handle, q.data_ptr(), B, frame_type, frame_id, out.data_ptr(),
                                         _stream_ptr(self.device)))
        return out

    def frame_jacobian(self, q: torch.Tensor, frame_type: int, frame_id: int) -> torch.Tensor:
        B = q.shape[0]
        _check_tensor(q, (B, self.model.nq), torch.float64, "q")
        out = torch.empty((B, 6, self.model.nv), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_frame_jacobian_batch(self.handle, q.data_ptr(), B, frame_type, frame_id,
                                                     out.data_ptr(), _stream_ptr(self.device)))
        return out

    def first_limit_violation(self, q: torch.Tensor, tol: float) -> torch.Tensor:
        B = q.shape[0]
        _check_tensor(q, (B, self.model.nq), torch.float64, "q")
        out = torch.empty((B,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_check_limits_batch(self.handle, q.data_ptr(), float(tol), B, out.data_ptr(),
                                                   _stream_ptr(self.device)))
        return out


class DeviceProblem:
    """A task + limit set bound to a :class:`DeviceModel` (``mk_problem``)."""

    def __init__(self, dmodel: DeviceModel, task_specs: Sequence[dict], limit_specs: Sequence[dict]):
        self.lib = dmodel.lib
        self.dmodel = dmodel
        self.device = dmodel.device
        ta, k1 = L.pack_tasks(task_specs)
        la, k2 = L.pack_limits(limit_specs)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_problem_create(dmodel.handle, ta, len(task_specs), la, len(limit_specs),
                                               C.byref(handle)))
        self.handle = handle
        info = L.ProblemInfo()
        L.check(self.lib.mk_problem_get_info(handle, C.byref(info)))
        self.nq, self.nv = info.nq, info.nv
        self.target_stride = info.target_stride
        self.n_rows = info.n_rows
        self.fused = bool(info.fused)
        self.n_chain = info.n_chain
        self.staged = bool(info.staged)
        self.staged_rows = bool(info.staged_rows)
        # posture targets the device copies currently hold, by task index
        self._posture_targets = {i: np.ascontiguousarray(s["target_q"], dtype=np.float64).tobytes()
                                 for i, s in enumerate(task_specs) if s["type"] == L.TASK_POSTURE}

    def set_posture_target(self, task_index: int, target_q: np.ndarray) -> None:
        """New target of posture task `task_index` (mk_problem_set_posture_target), ordered on the
        current stream; a no-op when the device copy already holds it."""
        target_q = np.ascontiguousarray(target_q, dtype=np.float64)
        raw = target_q.tobytes()
        if self._posture_targets.get(task_index) == raw:
            return
        if target_q.shape != (self.nq,):
            raise ValueError(f"posture target must have shape ({self.nq},)")
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_problem_set_posture_target(self.handle, int(task_index),
                                                           target_q.ctypes.data_as(C.c_void_p), _stream_ptr(self.device)))
        self._posture_targets[task_index] = raw

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.mk_problem_destroy(h)
            self.handle = None

    def _targets_ptr(self, targets: Optional[torch.Tensor], B: int):
        if self.target_stride == 0:
            return None
        _check_tensor(targets, (B, self.target_stride), torch.float64, "targets")
        return targets.data_ptr()

    def solve(self, q: torch.Tensor, targets: Optional[torch.Tensor], dt: float, damping: float,
              v: Optional[torch.Tensor] = None, status: Optional[torch.Tensor] = None):
        B = q.shape[0]
        _check_tensor(q, (B, self.nq), torch.float64, "q")
        if v is None:
            v = torch.empty((B, self.nv), dtype=torch.float64, device=self.device)
        if status is None:
            status = torch.empty((B,), dtype=torch.int32, device=self.device)
        _check_tensor(v, (B, self.nv), torch.float64, "v")
        _check_tensor(status, (B,), torch.int32, "status")
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_solve_ik_batch(self.handle, q.data_ptr(), self._targets_ptr(targets, B), float(dt),
                                               float(damping), B, v.data_ptr(), status.data_ptr(),
                                               _stream_ptr(self.device)))
        return v, status

    def _single_buffer(self, n_f64: int) -> torch.Tensor:
        """Device staging of the single-configuration calls: n_f64 doubles followed by one
        double's worth of int32 (two of them), so that one copy brings everything back."""
        buf = getattr(self, "_single", None)
        if buf is None or buf.numel() != n_f64 + 1:
            buf = torch.zeros((n_f64 + 1,), dtype=torch.float64, device=self.device)
            self._single = buf
        return buf

    def solve_single(self, q: torch.Tensor, targets: Optional[torch.Tensor], dt: float, damping: float, tol: float):
        """Limit check + solve of ONE configuration with a single device-to-host copy (the
        interactive callers' per-step path). Returns (v [nv] numpy, status, first offending
        joint or -1)."""
        _check_tensor(q, (1, self.nq), torch.float64, "q")
        nv = self.nv
        buf = self._single_buffer(nv)
        base = buf.data_ptr()
        with torch.cuda.device(self.device):
            stream = _stream_ptr(self.device)
            L.check(self.lib.mk_check_limits_batch(self.dmodel.handle, q.data_ptr(), float(tol), 1,
                                                   base + 8 * nv + 4, stream))
            L.check(self.lib.mk_solve_ik_batch(self.handle, q.data_ptr(), self._targets_ptr(targets, 1), float(dt),
                                               float(damping), 1, base, base + 8 * nv, stream))
        host = buf.cpu().numpy()
        flags = host[nv:].view(np.int32)
        return host[:nv], int(flags[0]), int(flags[1])

    def iterate_single_(self, q: torch.Tensor, targets: Optional[torch.Tensor], dt: float, damping: float,
                        max_iters: int, pos_tol: float, ori_tol: float) -> dict:
        """`iterate_` for ONE configuration, results as numpy / ints from a single copy."""
        _check_tensor(q, (1, self.nq), torch.float64, "q")
        nv = self.nv
        buf = self._single_buffer(nv + 6)
        base = buf.data_ptr()
        ints = base + 8 * (nv + 6)
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_solve_ik_iterate_batch(
                self.handle, q.data_ptr(), self._targets_ptr(targets, 1), float(dt), float(damping), 1,
                int(max_iters), float(pos_tol), float(ori_tol), base, ints, base + 8 * nv, ints + 4,
                _stream_ptr(self.device)))
        host = buf.cpu().numpy()
        flags = host[nv + 6:].view(np.int32)
        return dict(iters=int(flags[0]), err=host[nv:nv + 6], status=int(flags[1]), v=host[:nv])

    def build_qp(self, q: torch.Tensor, targets: Optional[torch.Tensor], dt: float, damping: float):
        """Dense (P, c, G, h) of every problem; G, h are None without limit rows."""
        B = q.shape[0]
        _check_tensor(q, (B, self.nq), torch.float64, "q")
        n, m = self.nv, self.n_rows
        P = torch.empty((B, n, n), dtype=torch.float64, device=self.device)
        c = torch.empty((B, n), dtype=torch.float64, device=self.device)
        G = torch.empty((B, m, n), dtype=torch.float64, device=self.device) if m else None
        h = torch.empty((B, m), dtype=torch.float64, device=self.device) if m else None
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_build_qp_batch(self.handle, q.data_ptr(), self._targets_ptr(targets, B), float(dt),
                                               float(damping), B, P.data_ptr(), c.data_ptr(),
                                               G.data_ptr() if m else None, h.data_ptr() if m else None,
                                               _stream_ptr(self.device)))
        return P, c, G, h

    def task_eval(self, task: int, k: int, q: torch.Tensor, targets: Optional[torch.Tensor]):
        """(error [B, k], Jacobian [B, k, nv]) of task number `task`."""
        B = q.shape[0]
        _check_tensor(q, (B, self.nq), torch.float64, "q")
        err = torch.empty((B, k), dtype=torch.float64, device=self.device)
        jac = torch.empty((B, k, self.nv), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_task_eval_batch(self.handle, int(task), q.data_ptr(), self._targets_ptr(targets, B),
                                                B, err.data_ptr(), jac.data_ptr(), _stream_ptr(self.device)))
        return err, jac

    def iterate_(self, q: torch.Tensor, targets: Optional[torch.Tensor], dt: float, damping: float, max_iters: int,
                 pos_tol: float = -1.0, ori_tol: float = -1.0, want_v: bool = False, out: Optional[dict] = None):
        """In-place loop on q. Returns dict(iters, err, status[, v])."""
        B = q.shape[0]
        _check_tensor(q, (B, self.nq), torch.float64, "q")
        if out is None:
            out = {}
        iters = out.get("iters")
        if iters is None:
            iters = torch.empty((B,), dtype=torch.int32, device=self.device)
        err = out.get("err")
        if err is None:
            err = torch.empty((B, 6), dtype=torch.float64, device=self.device)
        status = out.get("status")
        if status is None:
            status = torch.empty((B,), dtype=torch.int32, device=self.device)
        v = out.get("v")
        if v is None and (want_v or not self.fused):   # the general path keeps its state in v / iters / err
            v = torch.zeros((B, self.nv), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_solve_ik_iterate_batch(
                self.handle, q.data_ptr(), self._targets_ptr(targets, B), float(dt), float(damping), B,
                int(max_iters), float(pos_tol), float(ori_tol), v.data_ptr() if v is not None else None,
                iters.data_ptr(), err.data_ptr(), status.data_ptr(), _stream_ptr(self.device)))
        res = dict(iters=iters, err=err, status=status)
        if v is not None:
            res["v"] = v
        return res


    def move_to_(self, q: torch.Tensor, goals: torch.Tensor, wp, dt: float, damping: float, max_steps: int,
                 max_iters: int, pos_tol: float, ori_tol: float) -> dict:
        """Waypoint reaching in one launch (``mk_move_to_batch``); q is updated in place.
        ``wp`` has pos_threshold / pos_step_size / rot_threshold / rot_step_size.
        Returns dict(steps, reached, solves, status) of int32 tensors [B]."""
        B = q.shape[0]
        _check_tensor(q, (B, self.nq), torch.float64, "q")
        _check_tensor(goals, (B, 7), torch.float64, "goals")
        out = {k: torch.empty((B,), dtype=torch.int32, device=self.device)
               for k in ("steps", "reached", "solves", "status")}
        cfg = L.WaypointConfig(float(wp.pos_threshold), float(wp.pos_step_size), float(wp.rot_threshold),
                               float(wp.rot_step_size))
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_move_to_batch(
                self.handle, q.data_ptr(), goals.data_ptr(), C.byref(cfg), float(dt), float(damping), B,
                int(max_steps), int(max_iters), float(pos_tol), float(ori_tol), out["steps"].data_ptr(),
                out["reached"].data_ptr(), out["solves"].data_ptr(), out["status"].data_ptr(),
                _stream_ptr(self.device)))
        return out


    def move_to_passes_(self, task_index: int, q: torch.Tensor, goals: torch.Tensor, targets: torch.Tensor, wp, dt: float,
                        damping: float, max_steps: int, max_iters: int, pos_tol: float, ori_tol: float) -> dict:
        """Waypoint reaching for a problem outside the fused path (``mk_move_to_passes_batch``): every
        outer step as passes of the general loop.  ``targets`` [B, target_stride] carries the other
        tasks' targets; the slot of the driven frame task (``task_index``) is working memory.
        Returns dict(steps, reached, solves, status, iters, err, v)."""
        B = q.shape[0]
        _check_tensor(q, (B, self.nq), torch.float64, "q")
        _check_tensor(goals, (B, 7), torch.float64, "goals")
        _check_tensor(targets, (B, self.target_stride), torch.float64, "targets")
        out = {k: torch.empty((B,), dtype=torch.int32, device=self.device)
               for k in ("steps", "reached", "solves", "status", "iters")}
        out["err"] = torch.empty((B, 6), dtype=torch.float64, device=self.device)
        out["v"] = torch.empty((B, self.nv), dtype=torch.float64, device=self.device)
        cfg = L.WaypointConfig(float(wp.pos_threshold), float(wp.pos_step_size), float(wp.rot_threshold),
                               float(wp.rot_step_size))
        with torch.cuda.device(self.device):
            L.check(self.lib.mk_move_to_passes_batch(
                self.handle, int(task_index), q.data_ptr(), goals.data_ptr(), targets.data_ptr(), C.byref(cfg), float(dt),
                float(damping), B, int(max_steps), int(max_iters), float(pos_tol), float(ori_tol), out["steps"].data_ptr(),
                out["reached"].data_ptr(), out["solves"].data_ptr(), out["status"].data_ptr(), out["v"].data_ptr(),
                out["iters"].data_ptr(), out["err"].data_ptr(), _stream_ptr(self.device)))
        return out


def qp_solve(P: torch.Tensor, c: torch.Tensor, G: Optional[torch.Tensor] = None, h: Optional[torch.Tensor] = None):
    """Batched dense QP: min 1/2 x'Px + c'x s.t. Gx <= h. Returns (x [B, n], status [B])."""
    lib = L.load_library()
    B, n = c.shape
    _check_tensor(P, (B, n, n), torch.float64, "P")
    _check_tensor(c, (B, n), torch.float64, "c")
    m = 0 if G is None else G.shape[1]
    if m:
        _check_tensor(G, (B, m, n), torch.float64, "G")
        _check_tensor(h, (B, m), torch.float64, "h")
    x = torch.empty((B, n), dtype=torch.float64, device=P.device)
    status = torch.empty((B,), dtype=torch.int32, device=P.device)
    with torch.cuda.device(P.device):
        L.check(lib.mk_qp_solve_batch(P.data_ptr(), c.data_ptr(), G.data_ptr() if m else None,
                                      h.data_ptr() if m else None, n, m, B, x.data_ptr(), status.data_ptr(),
                                      _stream_ptr(P.device)))
    return x, status
