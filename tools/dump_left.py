"""Debug helper: dump the gripper-arm problems the lane rows stage hands on (inputs for the host build)."""
import os, sys
sys.path.insert(0, "tests")
import numpy as np, torch
import test_gpu_lane_rows as T
from mink_amd import _lib as L
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_mjcf
model = load_mjcf("tests/models/grip_arm.xml")
dm = DeviceModel(model, torch.device("cuda:0"))
groups = [(["pad_a", "pad_b"], ["brick_geom", "floor", "pillar"]), (["tip_b"], ["ball_geom", "floor"])]
tasks, limits = T.specs(model, groups)
q, tg = T.inputs(model, dm, 6144, 21)
staged = DeviceProblem(dm, tasks, limits)
with L.options(general_stage=0):
    v, st = staged.solve(q, tg, T.DT, T.DAMPING)
left = torch.nonzero(st != 0)[:, 0]
vf, sf = staged.solve(q, tg, T.DT, T.DAMPING)
np.savez("gpurun_out/left.npz", q=q[left].cpu().numpy(), tg=tg[left].cpu().numpy(), v=vf[left].cpu().numpy())
print("left", left.numel())
