"""One-off check: staged solve against the general kernel alone on large config-4 batches."""
import os, sys
import torch
from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

model = load_builtin("tidybot_cube")
dev = torch.device("cuda:0")
dm = DeviceModel(model, dev)
tasks, limits = W.config4_specs(model)
staged = DeviceProblem(dm, tasks, limits)
with L.options(staged=0):
    general = DeviceProblem(dm, tasks, limits)
n = 262144
for seed in (1, 2, 3):
    for near in (0.25, 1.0):
        q, t = W.config4_inputs(model, dm, n, seed, near_fraction=near)
        vs, ss = staged.solve(q, t, W.DT, W.DAMPING)
        vg, sg = general.solve(q, t, W.DT, W.DAMPING)
        d = (vs - vg).abs().amax(dim=1)
        print(f"seed {seed} near {near}: max |dv| {float(d.max()):.3e}  >1e-9: {int((d > 1e-9).sum())}  >1e-7: {int((d > 1e-7).sum())}  "
              f"status staged {torch.unique(ss).tolist()} general {torch.unique(sg).tolist()}", flush=True)
