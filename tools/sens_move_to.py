"""Tuning helper (GPU box): sensitivity of waypoint-reaching rollouts to the last bit of their start
configuration -- set 3 of tools/soak_move_to.py run twice through mk_move_to_batch, the second time with
every coordinate of q0 moved by at most one unit in the last place."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mink_amd as mink
from mink_amd import rollout, workloads as W
from mink_amd.lie import SO3
B=32768; k=3
rng = np.random.default_rng([555, k])
model = mink.load_builtin("mobile_kinova")
q0 = W.sample_q(model, rng, B)
frame = mink.FrameTask("pinch_site", "site", float(10 ** rng.uniform(-0.3, 0.3)), float(10 ** rng.uniform(-0.3, 0.3)), lm_damping=float(rng.choice([1e-2, 1.0])))
posture = mink.PostureTask(model, cost=W.posture_cost(model)); posture.set_target(W.home_q(model))
limits = [mink.ConfigurationLimit(model)]
if rng.integers(2): limits.append(mink.VelocityLimit(model, {n: v for n, v in W.VELOCITY_LIMITS.items() if model.joint_id(n) >= 0}))
wp = mink.LinearWaypointReachConfig(float(rng.choice([0.005, 0.01, 0.03])), float(rng.choice([0.04, 0.08, 0.2])), float(rng.choice([0.03, 0.05, 0.2])), 0.3)
max_steps, max_iters = int(rng.choice([5, 12, 30])), int(rng.choice([5, 20]))
dt, damping = float(rng.choice([1 / 200, 2e-2])), float(rng.choice([1e-6, 1e-3]))
cfg0 = mink.Configuration(model, q0)
start = cfg0.get_transform_frame_to_world("pinch_site", "site")
spread = float(rng.choice([0.1, 0.3]))
goal_pos = start.translation() + torch.tensor(rng.uniform(-spread, spread, (B, 3)), device="cuda:0")
goal_rot = start.rotation() @ SO3.exp(torch.tensor(rng.uniform(-0.6, 0.6, (B, 3)), device="cuda:0"))
q1 = q0 * (1.0 + 2.0 ** -52 * np.random.default_rng(1).choice([-1.0, 0.0, 1.0], q0.shape))   # last-bit perturbation
ca, cb = mink.Configuration(model, q0), mink.Configuration(model, q1)
a = mink.move_to(ca, frame, [frame, posture], goal_pos, goal_rot, wp, dt=dt, damping=damping, max_steps=12, max_iters=max_iters, limits=limits)
b = mink.move_to(cb, frame, [frame, posture], goal_pos, goal_rot, wp, dt=dt, damping=damping, max_steps=12, max_iters=max_iters, limits=limits)
per = (ca.q - cb.q).abs().amax(dim=1)
print(f"one launch, start configurations differing in the last bit: |dq| > 1e-10: {int((per > 1e-10).sum())}, > 1e-8: {int((per > 1e-8).sum())}, max {float(per.max()):.2e}, median {float(per.median()):.1e}")
