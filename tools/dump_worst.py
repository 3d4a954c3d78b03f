"""Debug helper: dump the config-4 problems where staged and general solves differ most."""
import os
import numpy as np, torch
from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin
model = load_builtin("tidybot_cube")
dm = DeviceModel(model, torch.device("cuda:0"))
tasks, limits = W.config4_specs(model)
staged = DeviceProblem(dm, tasks, limits)
with L.options(staged=0):
    general = DeviceProblem(dm, tasks, limits)
q, t = W.config4_inputs(model, dm, 262144, 1, near_fraction=1.0)
vs, _ = staged.solve(q, t, W.DT, W.DAMPING)
vg, _ = general.solve(q, t, W.DT, W.DAMPING)
d = (vs - vg).abs().amax(dim=1)
idx = torch.argsort(d, descending=True)[:24]
np.savez("gpurun_out/worst.npz", q=q[idx].cpu().numpy(), tg=t[idx].cpu().numpy(), vs=vs[idx].cpu().numpy(), vg=vg[idx].cpu().numpy())
print(d[idx].tolist())
