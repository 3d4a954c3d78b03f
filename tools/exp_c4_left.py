"""Experiment: which config-4 problems does the lane-per-problem rows stage hand on to the general kernel?"""
from mink_amd import _lib as L
import torch
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

L.set_option("general_stage", 0)

model = load_builtin("tidybot_cube")
dev = torch.device("cuda:0")
dm = DeviceModel(model, dev)
tasks, limits = W.config4_specs(model)
n = 262144
q, t = W.config4_inputs(model, dm, n, 7)
p = DeviceProblem(dm, tasks, limits)
v, s = p.solve(q, t, W.DT, W.DAMPING)
left = torch.nonzero(s != 0)[:, 0]
print("left pending:", left.numel(), "of", n, "statuses", torch.unique(s).tolist())
if left.numel():
    idx = left[:4096]
    P, c, G, h = p.build_qp(q[idx].contiguous(), t[idx].contiguous(), W.DT, W.DAMPING)
    nfin = torch.isfinite(h[:, -8:]).sum(dim=1)
    print("rows in range of those:", torch.bincount(nfin).tolist())
