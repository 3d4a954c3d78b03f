"""Soak test (GPU box): the staged solve (structured kernel + rows stage + general kernel for the
rest) against the general kernel alone, on the config-4 problem (Frame + RelativeFrame + Posture,
Configuration + Velocity + CollisionAvoidance limits) over randomly drawn parameters.

    PYTHONPATH=. python tools/soak_staged.py [sets] [problems per set]"""
import os
import sys

import numpy as np
from mink_amd import _lib as L
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

SETS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
model = load_builtin("tidybot_cube")
dev = torch.device("cuda:0")
dm = DeviceModel(model, dev)

print(f"# soak: staged solve against the general kernel alone, {SETS} random parameter sets x {B} config-4 problems\n")
print("| # | dt | damping | frame lm | rel. task cost / gain | posture cost | coll. gain | min dist | detect dist | near fraction | "
      "max abs dv | > 1e-9 | > 1e-6 | status staged | status general |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
worst, n6 = 0.0, 0
for k in range(SETS):
    rng = np.random.default_rng([77, k])
    tasks, limits = W.config4_specs(model)
    dt = float(rng.choice([1 / 200, 2e-2]))
    damping = float(rng.choice([1e-12, 1e-6, 1e-3]))
    tasks[0]["lm_damping"] = float(rng.choice([1e-2, 1.0]))
    tasks[0]["cost"] = np.concatenate([np.full(3, 10 ** rng.uniform(-0.5, 0.5)), np.full(3, 10 ** rng.uniform(-0.5, 0.5))])
    tasks[1]["cost"] = np.full(6, 10 ** rng.uniform(-1, 0.5))
    tasks[1]["gain"] = float(rng.uniform(0.3, 1.0))
    pc = np.zeros(model.nv)
    pc[3:] = float(rng.choice([1e-3, 1e-2]))
    tasks[2]["cost"] = pc
    coll = limits[-1]
    coll["gain"] = float(rng.uniform(0.3, 1.0))
    coll["minimum_distance_from_collisions"] = float(rng.choice([0.002, 0.005, 0.02]))
    coll["collision_detection_distance"] = float(coll["minimum_distance_from_collisions"] + rng.choice([0.005, 0.03, 0.1]))
    near = float(rng.choice([0.25, 1.0]))
    staged = DeviceProblem(dm, tasks, limits)
    with L.options(staged=0):
        general = DeviceProblem(dm, tasks, limits)
    assert staged.staged and not general.staged
    q, t = W.config4_inputs(model, dm, B, 1000 + k, near_fraction=near)
    vs, ss = staged.solve(q, t, dt, damping)
    vg, sg = general.solve(q, t, dt, damping)
    ok = (ss == 0) & (sg == 0)
    d = (vs - vg).abs().amax(dim=1)[ok]
    mx = float(d.max()) if d.numel() else 0.0
    c9, c6 = int((d > 1e-9).sum()), int((d > 1e-6).sum())
    worst, n6 = max(worst, mx), n6 + c6
    print(f"| {k} | {dt:.3g} | {damping:.0e} | {tasks[0]['lm_damping']:.0e} | {tasks[1]['cost'][0]:.2f} / {tasks[1]['gain']:.2f} | "
          f"{pc[3]:.0e} | {coll['gain']:.2f} | {coll['minimum_distance_from_collisions']} | {coll['collision_detection_distance']:.3f} | "
          f"{near} | {mx:.2e} | {c9} | {c6} | {torch.unique(ss).tolist()} | {torch.unique(sg).tolist()} |", flush=True)
print(f"\nworst |dv| {worst:.3e} rad/s over {SETS * B} problems; {n6} problems differ by more than 1e-6")
