"""Tuning helper (GPU): wall time of mink_amd.move_to for a benchmark-config-4 problem (collision
limits + relative frame task: mk_move_to_passes_batch) against the host loop over outer steps."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import mink_amd as mk
from mink_amd import rollout
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel

dev = "cuda:0"
model = mk.load_builtin("tidybot_cube")
dm = DeviceModel(model, dev)
wcfg = mk.LinearWaypointReachConfig(0.01, 0.08, 0.05, 0.3)
for B in [int(a) for a in sys.argv[1:]] or [16384]:
    q0, _ = W.config4_inputs(model, dm, B, 5)
    frame = mk.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
    rel = mk.RelativeFrameTask("pinch_site", "site", "base_link", "body", 0.3, 0.3)
    rel.set_target_from_configuration(mk.Configuration(model, q0.clone(), device=dev))
    posture = mk.PostureTask(model, cost=W.posture_cost(model))
    posture.set_target(W.home_q(model))
    limits = [mk.ConfigurationLimit(model), mk.VelocityLimit(model, W.VELOCITY_LIMITS),
              mk.CollisionAvoidanceLimit(model, [(list(W.PADS), ["cube_geom", "floor"])])]
    start = mk.Configuration(model, q0.clone(), device=dev).get_transform_frame_to_world("pinch_site", "site")
    g = torch.Generator(device="cpu").manual_seed(4)
    goal_pos = start.translation() + (torch.rand((B, 3), generator=g, dtype=torch.float64) * 0.3 - 0.15).to(dev)
    goal_rot = start.rotation() @ mk.SO3.exp((torch.rand((B, 3), generator=g, dtype=torch.float64) * 0.8 - 0.4).to(dev))
    for name in ("passes", "stepwise"):
        for rep in range(2):
            cfg = mk.Configuration(model, q0.clone(), device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with cfg.deferred_checks():
                if name == "passes":
                    out = mk.move_to(cfg, frame, [frame, rel, posture], goal_pos, goal_rot, wcfg, max_steps=50, limits=limits)
                else:
                    out = rollout._move_to_stepwise(cfg, frame, [frame, rel, posture], goal_pos, goal_rot, wcfg, 1.0 / 200.0,
                                                    "quadprog", 1e-3, 50, 20, 1e-4, 1e-4, limits)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"B={B} {name}: {1e3 * dt:.1f} ms, {B / dt / 1e3:.1f} k rollouts/s, {float(out['solves'].sum()) / dt / 1e6:.1f} M solves/s, "
              f"reached {float(out['reached'].double().mean()):.4f}, steps mean {float(out['steps'].double().mean()):.2f} "
              f"max {int(out['steps'].max())}", flush=True)
