"""Probe for rocprofv3 --kernel-trace: the near-target loop at 65536 problems as the library
runs it (lane kernel with wave hand-off, then the cooperative kernel)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

model = load_builtin("tidybot_cube")
dm = DeviceModel(model, "cuda:0")
rng = np.random.default_rng(4000)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))
q0 = torch.tensor(W.sample_q(model, rng, B), device="cuda:0")
tg = torch.tensor(W.offset_targets(dm.frame_pose(q0, 2, model.site_id("pinch_site")).cpu().numpy(), rng), device="cuda:0")
from mink_amd import _lib as L
if len(sys.argv) > 2:
    L.set_option("handoff", int(sys.argv[2]))
for i in range(5):
    r = prob.iterate_(q0.clone(), tg, W.DT, W.DAMPING, 20, 1e-4, 1e-4)
torch.cuda.synchronize()
print("solves per problem:", torch.bincount(r["iters"].long(), minlength=21).tolist())
