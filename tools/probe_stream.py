"""Tuning helper (GPU): three launches of the near-target early-exit loop at 2^20 problems
(lane-streaming kernel), for rocprofv3 --pmc passes (tools/pmc_probe.sh with
PROBE=tools/probe_stream.py KERNEL=fused_iterate_stream)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

dev = "cuda:0"
model = load_builtin("tidybot_cube")
dm = DeviceModel(model, dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rng = np.random.default_rng(6)
q0 = torch.tensor(W.sample_q(model, rng, B), device=dev)
poses = dm.frame_pose(q0, L.FRAME_TYPES["site"], model.site_id("pinch_site")).cpu().numpy()
tg = torch.tensor(W.offset_targets(poses, rng), device=dev)
prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))
for _ in range(3):
    r = prob.iterate_(q0.clone(), tg, W.DT, W.DAMPING, 20, 1e-4, 1e-4)
torch.cuda.synchronize()
print("solves per launch", int(r["iters"].sum()))
