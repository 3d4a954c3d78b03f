"""Tuning helper: a few launches of the iterated kernel on the headline workload, for
rocprofv3 --pmc passes (library chosen by MINK_AMD_LIB)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

model = load_builtin("tidybot_cube")
dm = DeviceModel(model, "cuda:0")
prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
rng = np.random.default_rng(4000)
B = 131072
q0 = torch.tensor(W.sample_q(model, rng, B), device="cuda:0")
tg = dm.frame_pose(torch.tensor(W.sample_q(model, rng, B), device="cuda:0"), 2, 0)
for i in range(3):
    prob.iterate_(q0.clone(), tg, W.DT, W.DAMPING, 50)
torch.cuda.synchronize()
print("done")
