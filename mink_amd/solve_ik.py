"""``build_ik`` / ``solve_ik`` of the host API (reference: mink/solve_ik.py:43-105) and
the batched form of the callers' loop (interactive_scripts/teleop_mouse.py:105-121)."""
from __future__ import annotations

from typing import NamedTuple, Optional, Sequence

import numpy as np
import torch

from ._problems import get_problem
from .limits import ConfigurationLimit, Limit
from .tasks import Task

SUPPORTED_SOLVERS = ("quadprog", "daqp", "mink_amd")


class Problem(NamedTuple):
    """The stacked QP, like ``qpsolvers.Problem``: min 1/2 x'Px + q'x s.t. Gx <= h."""
    P: object
    q: object
    G: Optional[object]
    h: Optional[object]

    def unpack(self):
        return self.P, self.q, self.G, self.h, None, None, None, None


def _resolve_limits(configuration, limits):
    if limits is None:
        dm = configuration.dmodel
        default = getattr(dm, "default_limits", None)
        if default is None:
            default = dm.default_limits = [ConfigurationLimit(configuration.model)]
        return default
    return limits


def _has_custom(tasks, limits) -> bool:
    return any(t._is_custom() for t in tasks) or any(l._is_custom() for l in limits)


def _assemble_with_custom(configuration, tasks, limits, dt, damping):
    """Dense (P, c, G, h) device tensors when some tasks / limits are user-defined subclasses:
    the built-in ones are assembled by the kernels (mk_build_qp_batch), the others add their
    objective (generic formula on their compute_error / compute_jacobian) and their rows."""
    dev, B = configuration.device, configuration.batch_size
    builtin_t = [t for t in tasks if not t._is_custom()]
    builtin_l = [l for l in limits if not l._is_custom()]
    prob, targets = get_problem(configuration, builtin_t, builtin_l)
    P, c, G, h = prob.build_qp(configuration.q_tensor, targets, dt, damping)
    for t in tasks:
        if t._is_custom():
            H_t, c_t = t._generic_objective(configuration)
            P = P + H_t
            c = c + c_t
    rows_G, rows_h = ([G], [h]) if G is not None else ([], [])
    for l in limits:
        if not l._is_custom():
            continue
        con = l.compute_qp_inequalities(configuration, dt)
        if con.inactive:
            continue
        Gl = torch.as_tensor(con.G, dtype=torch.float64).to(dev)
        hl = torch.as_tensor(con.h, dtype=torch.float64).to(dev)
        if Gl.ndim == 2:
            Gl, hl = Gl[None].expand(B, -1, -1), hl[None].expand(B, -1)
        if Gl.shape[0] != B or Gl.shape[2] != prob.nv or tuple(hl.shape) != tuple(Gl.shape[:2]):
            raise ValueError(f"{l.__class__.__name__}: compute_qp_inequalities must return G [.., m, {prob.nv}] and "
                             f"h [.., m] for {B} configuration(s)")
        rows_G.append(Gl)
        rows_h.append(hl)
    if rows_G:
        G, h = torch.cat(rows_G, dim=1).contiguous(), torch.cat(rows_h, dim=1).contiguous()
    else:
        G = h = None
    return P.contiguous(), c.contiguous(), G, h


def build_ik(configuration, tasks: Sequence[Task], dt: float, damping: float = 1e-12,
             limits: Optional[Sequence[Limit]] = None) -> Problem:
    """Build the QP of the current configuration(s). ``limits=None`` means a default
    ConfigurationLimit, ``limits=[]`` no limits (G and h are None)."""
    limits = _resolve_limits(configuration, limits)
    if _has_custom(tasks, limits):
        P, c, G, h = _assemble_with_custom(configuration, tasks, limits, dt, damping)
    else:
        prob, targets = get_problem(configuration, tasks, limits)
        P, c, G, h = prob.build_qp(configuration.q_tensor, targets, dt, damping)
    o = configuration._out
    return Problem(o(P), o(c), None if G is None else o(G), None if h is None else o(h))


def _limit_check(configuration, safety_break: bool) -> None:
    """mink/solve_ik.py:99.  The answer comes now, as in the reference, unless the caller has
    opted into deferred reports (Configuration.deferred_checks) and does not want the exception."""
    if safety_break or not configuration.batched or not configuration.defer_checks:
        configuration.check_limits(safety_break=safety_break)
    else:
        configuration._defer_limit_check()


def _check_solver(solver: str) -> None:
    if solver not in SUPPORTED_SOLVERS:
        raise ValueError(f"solver '{solver}' is not available; the exact active-set kernels answer to "
                         f"{SUPPORTED_SOLVERS} (the minimiser of a strictly convex QP does not depend on the backend)")


def solve_ik(configuration, tasks: Sequence[Task], dt: float, solver: str, damping: float = 1e-12,
             safety_break: bool = False, limits: Optional[Sequence[Limit]] = None, **kwargs):
    """Velocity ``v`` tangent to the current configuration(s): [nv] numpy for a single
    configuration, [B, nv] device tensor for a batch.

    As in the reference (mink/solve_ik.py:99-103) the limit check and the assert on a failed QP
    happen inside the call, for a batch too -- any failing problem fails the call.  A caller that
    wants a batched call to return without waiting for the device wraps it in
    ``with configuration.deferred_checks():``; the limit warning and the solver's per-problem
    status are then kept on the device and reported when the block ends (``safety_break=True``
    still raises at once)."""
    _check_solver(solver)
    limits = _resolve_limits(configuration, limits)
    if _has_custom(tasks, limits):
        # user-defined Task / Limit subclasses: assemble the dense QP and hand it to the
        # batched QP kernel (mk_qp_solve_batch)
        from .engine import qp_solve
        _limit_check(configuration, safety_break)
        P, c, G, h = _assemble_with_custom(configuration, tasks, limits, dt, damping)
        x, status = qp_solve(P, c, G, h)
        if configuration.batched:
            configuration._report_status(status)
        else:
            assert int((status != 0).sum()) == 0, "QP solver failed (infeasible limits or non-finite input)"
        return configuration._out(x / dt)
    if not configuration.batched:
        # One configuration: the limit check rides along with the solve and one copy brings
        # back velocity, solver status and the check's result.
        prob, targets = get_problem(configuration, tasks, _resolve_limits(configuration, limits))
        v, status, first = prob.solve_single(configuration.q_tensor, targets, dt, damping, tol=1e-6)
        if first >= 0:
            configuration._report_limit_violation(0, first, 1, safety_break)
        assert status == 0, "QP solver failed (infeasible limits or non-finite input)"
        return v
    # A batch: limit check, solve, and the assert on the solver status (deferred when the caller
    # asked for that).
    _limit_check(configuration, safety_break)
    prob, targets = get_problem(configuration, tasks, _resolve_limits(configuration, limits))
    v, status = prob.solve(configuration.q_tensor, targets, dt, damping)
    configuration._report_status(status)
    return v


def solve_ik_iterate(configuration, tasks: Sequence[Task], dt: float, solver: str, damping: float = 1e-12,
                     max_iters: int = 20, pos_threshold: float = 1e-4, ori_threshold: float = 1e-4,
                     limits: Optional[Sequence[Limit]] = None) -> dict:
    """The callers' loop on the GPU: up to ``max_iters`` x (solve_ik, integrate_inplace,
    frame-task error), each configuration leaving the loop when its frame task error is
    below the thresholds.  Updates the configuration in place and returns
    ``dict(iters, err, status, v)``.  A problem the fused kernels serve (one FrameTask,
    posture-like tasks, Configuration / Velocity limits) runs as one launch; for any other task /
    limit set the library issues the same loop as a sequence of launches of the general
    kernels, with the exit test on the first FrameTask."""
    _check_solver(solver)
    if _has_custom(tasks, _resolve_limits(configuration, limits)):
        raise NotImplementedError("solve_ik_iterate runs built-in tasks and limits only; with user-defined Task / "
                                  "Limit subclasses call solve_ik and integrate_inplace step by step")
    prob, targets = get_problem(configuration, tasks, _resolve_limits(configuration, limits))
    if not configuration.batched:
        res = prob.iterate_single_(configuration.q_tensor, targets, dt, damping, max_iters, pos_threshold,
                                   ori_threshold)
        assert res["status"] == 0, "QP solver failed (infeasible limits or non-finite input)"
        return res
    res = prob.iterate_(configuration.q_tensor, targets, dt, damping, max_iters, pos_threshold, ori_threshold,
                        want_v=True)
    configuration._report_status(res["status"])     # also returned: callers may look at it themselves
    return dict(res)

