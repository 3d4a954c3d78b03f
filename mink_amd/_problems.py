"""Lowering of task / limit objects to device problems, shared by tasks.py, limits.py
and solve_ik.py."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from .engine import DeviceProblem


def _signature(spec: dict) -> tuple:
    """Hashable value of a descriptor; the default ``_key`` of tasks and limits."""
    if spec is None:
        return None
    out = []
    for k in sorted(spec):
        v = spec[k]
        out.append((k, np.asarray(v).tobytes() if isinstance(v, (np.ndarray, list, tuple)) else v))
    return tuple(out)


def get_problem(configuration, tasks: Sequence, limits: Sequence):
    """(device problem, targets tensor) for a task / limit set; problems are cached per
    uploaded model by the value of their descriptors."""
    model = configuration.model
    # keys (and with them missing posture targets) and targets first: a missing target
    # is reported before any device work
    key = (tuple(t._key(model) for t in tasks),
           tuple(k for k in (l._key(model) for l in limits) if k is not None))
    parts = [t._target_tensor(configuration) for t in tasks if t._target_width() > 0]
    if not parts:
        targets = None
    elif len(parts) == 1:
        targets = parts[0] if parts[0].is_contiguous() else parts[0].contiguous()
    else:
        targets = torch.cat(parts, dim=1).contiguous()
    cache = configuration.dmodel.problem_cache
    prob = cache.get(key)
    if prob is None:
        if len(cache) > 64:
            cache.clear()
        task_specs = [t._spec(model) for t in tasks]
        limit_specs = [s for s in (l._spec(model) for l in limits) if s is not None]
        prob = DeviceProblem(configuration.dmodel, task_specs, limit_specs)
        cache[key] = prob
    # posture targets are not part of the key: bring the problem's copies up to date
    for i, t in enumerate(tasks):
        if hasattr(t, "_posture_target"):
            prob.set_posture_target(i, t._posture_target())
    return prob, targets
