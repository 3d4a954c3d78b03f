"""Lowering of task / limit objects to device problems, shared by tasks.py, limits.py
and solve_ik.py."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from .engine import DeviceProblem


def _signature(spec: dict) -> tuple:
    out = []
    for k in sorted(spec):
        v = spec[k]
        out.append((k, np.asarray(v).tobytes() if isinstance(v, (np.ndarray, list, tuple)) else v))
    return tuple(out)


def get_problem(configuration, tasks: Sequence, limits: Sequence):
    """(device problem, targets tensor) for a task / limit set; problems are cached per
    uploaded model by the value of their descriptors."""
    model = configuration.model
    task_specs = [t._spec(model) for t in tasks]
    limit_specs = [s for s in (l._spec(model) for l in limits) if s is not None]
    # targets first: a missing target is reported before any device work
    parts = [t._target_tensor(configuration) for t in tasks if t._target_width() > 0]
    targets = torch.cat(parts, dim=1).contiguous() if parts else None
    key = (tuple(_signature(s) for s in task_specs), tuple(_signature(s) for s in limit_specs))
    cache = configuration.dmodel.problem_cache
    prob = cache.get(key)
    if prob is None:
        if len(cache) > 64:
            cache.clear()
        prob = DeviceProblem(configuration.dmodel, task_specs, limit_specs)
        cache[key] = prob
    return prob, targets
