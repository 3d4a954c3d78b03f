"""Tuning helper: a few launches of the wave-per-problem kernel on the config-4 workload,
for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

L.set_option("staged", 0)      # the wave-per-problem kernel for every problem
model = load_builtin("tidybot_cube")
dm = DeviceModel(model, "cuda:0")
tasks, limits = W.config4_specs(model)
mode = sys.argv[1] if len(sys.argv) > 1 else "solve"
B = 32768
q, tg = W.config4_inputs(model, dm, B, 3)
prob = DeviceProblem(dm, tasks, limits)
for _ in range(3):
    if mode == "solve":
        prob.solve(q, tg, W.DT, W.DAMPING)
    else:
        prob.task_eval(0, 6, q, tg)
torch.cuda.synchronize()
print("done")
