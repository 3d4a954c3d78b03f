"""Soak test (GPU box): the staged solve on the gripper-arm test model (every lane pair type: box, sphere,
capsule, plane; bodies behind the chain, finger joints between chain and pad, free objects) against the
general kernel alone, over many seeds.  Uses the set-up of tests/test_gpu_lane_rows.py.

    PYTHONPATH=. python tools/soak_lane_rows.py [seeds] [problems per seed]"""
import os
import sys

from mink_amd import _lib as L
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_lane_rows as T
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_mjcf

SEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
model = load_mjcf(os.path.join(ROOT, "tests", "models", "grip_arm.xml"))
dm = DeviceModel(model, torch.device("cuda:0"))
GROUPS = {"lane pairs": [(["pad_a", "pad_b"], ["brick_geom", "floor", "pillar"]), (["tip_b"], ["ball_geom", "floor"])],
          "crowded": [(list(T.HAND), list(T.OBJECTS) + ["floor", "pillar"])]}
worst = 0.0
for gname, groups in GROUPS.items():
    tasks, limits = T.specs(model, groups)
    staged = DeviceProblem(dm, tasks, limits)
    with L.options(staged=0):
        general = DeviceProblem(dm, tasks, limits)
    assert staged.staged and not general.staged
    for seed in range(100, 100 + SEEDS):
        q, tg = T.inputs(model, dm, B, seed)
        vs, ss = staged.solve(q, tg, T.DT, T.DAMPING)
        vg, sg = general.solve(q, tg, T.DT, T.DAMPING)
        ok = (ss == 0) & (sg == 0)
        d = (vs - vg).abs().amax(dim=1)[ok]
        worst = max(worst, float(d.max()))
        print(f"{gname}, seed {seed}: max |dv| {float(d.max()):.2e}, > 1e-9: {int((d > 1e-9).sum())}, > 1e-7: {int((d > 1e-7).sum())}, "
              f"status differ {int((ss != sg).sum())}, status nonzero (general) {int((sg != 0).sum())}", flush=True)
print(f"worst |dv| {worst:.2e} over {2 * SEEDS * B} problems")
