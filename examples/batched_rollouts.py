"""Batched use: many TidyBots at once, the way a policy-rollout or data-generation job
would call the reference's envs/mj_env.py move_to (lines 193-276), kinematics only.

    python examples/batched_rollouts.py [B]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mink_amd as mink  # noqa: E402
from mink_amd import workloads as W  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
model = mink.load_builtin("tidybot_cube")
rng = np.random.default_rng(0)
configuration = mink.Configuration(model, W.sample_q(model, rng, B))       # q of shape [B, nq]: batched mode

frame = mink.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
posture = mink.PostureTask(model, cost=W.posture_cost(model))
posture.set_target(W.home_q(model))
limits = [mink.ConfigurationLimit(model), mink.VelocityLimit(model, W.VELOCITY_LIMITS)]

# one solve_ik step for the whole batch: targets [B, 7], velocities [B, nv] stay on the GPU
start = configuration.get_transform_frame_to_world("pinch_site", "site")
offsets = torch.tensor(rng.uniform(-0.1, 0.1, (B, 3)), device=configuration.device)
frame.set_target(mink.SE3.from_rotation_and_translation(start.rotation(), start.translation() + offsets))
v = mink.solve_ik(configuration, [frame, posture], 1.0 / 200.0, "quadprog", 1e-3, limits=limits)
print("solve_ik:", tuple(v.shape), v.dtype, v.device)

# waypoint reaching to goal poses, every rollout to its own goal, as one launch
goal_pos = start.translation() + torch.tensor(rng.uniform(-0.15, 0.15, (B, 3)), device=configuration.device)
goal_rot = start.rotation() @ mink.SO3.exp(torch.tensor(rng.uniform(-0.4, 0.4, (B, 3)), device=configuration.device))
wp = mink.LinearWaypointReachConfig(pos_threshold=0.01, pos_step_size=0.08, rot_threshold=0.05, rot_step_size=0.3)
for attempt in range(2):                      # the first call also builds and caches the problem
    cfg = mink.Configuration(model, configuration.q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = mink.move_to(cfg, frame, [frame, posture], goal_pos, goal_rot, wp, max_steps=50, limits=limits)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"move_to: {B} rollouts in {1e3 * dt:.2f} ms, {float(out['reached'].double().mean()):.3f} reached, "
      f"{float(out['steps'].double().mean()):.1f} waypoint steps and {float(out['solves'].double().mean()):.1f} IK solves each")
