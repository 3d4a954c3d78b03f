"""Probe for rocprofv3 --pmc passes of the cooperative kernels: a few launches of the iterated
loop on 4096 problems (one row each, one wave per SIMD), 20 solves per problem, no exit test.
With "hist" as argument: histogram of the solves per problem of the near-target loop at 65536."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

model = load_builtin("tidybot_cube")
dm = DeviceModel(model, "cuda:0")
rng = np.random.default_rng(4000)
if len(sys.argv) > 1 and sys.argv[1] == "hist":
    B = 65536
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))
    q0 = torch.tensor(W.sample_q(model, rng, B), device="cuda:0")
    tg = torch.tensor(W.offset_targets(dm.frame_pose(q0, 2, model.site_id("pinch_site")).cpu().numpy(), rng), device="cuda:0")
    with L.options(coop=0):
        r = prob.iterate_(q0.clone(), tg, W.DT, W.DAMPING, 20, 1e-4, 1e-4)
    print("solves per problem:", torch.bincount(r["iters"].long(), minlength=21).tolist())
    sys.exit(0)
B = 4096
prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
q0 = torch.tensor(W.sample_q(model, rng, B), device="cuda:0")
tg = dm.frame_pose(torch.tensor(W.sample_q(model, rng, B), device="cuda:0"), 2, 0)
L.set_option("coop", 1)
for i in range(3):
    prob.iterate_(q0.clone(), tg, W.DT, W.DAMPING, 20)
torch.cuda.synchronize()
print("done")
