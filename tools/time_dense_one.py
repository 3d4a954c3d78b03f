"""Tuning helper: config-4 solve time of the library named by MINK_AMD_LIB."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin
model = load_builtin("tidybot_cube")
dm = DeviceModel(model, "cuda:0")
tasks, limits = W.config4_specs(model)
B = 65536
q, tg = W.config4_inputs(model, dm, B, 3)
prob = DeviceProblem(dm, tasks, limits)
v = torch.empty((B, model.nv), dtype=torch.float64, device="cuda:0"); st = torch.empty(B, dtype=torch.int32, device="cuda:0")
for _ in range(2): prob.solve(q, tg, W.DT, W.DAMPING, v=v, status=st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): prob.solve(q, tg, W.DT, W.DAMPING, v=v, status=st)
torch.cuda.synchronize(); print(os.environ.get("MINK_AMD_LIB", "in-tree"), "config4 solve ms:", round((time.perf_counter() - t0) / 5 * 1e3, 3))
