"""Soak test (GPU box): waypoint reaching as one launch (mk_move_to_batch) against the same loop
issued step by step from the host (waypoint generator in torch, one mk_solve_ik_iterate_batch per
outer step), over random waypoint / solver parameters.

    PYTHONPATH=. python tools/soak_move_to.py [sets] [rollouts per set]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import mink_amd as mink
from mink_amd import rollout
from mink_amd import workloads as W
from mink_amd.lie import SO3

SETS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
# third argument: "lane" / "coop" pins the kernel form of both sides.  Left to the library, the one
# launch and the step-by-step loop hand different rollouts to the cooperative kernels at different
# moments; the two forms agree to rounding, and a rollout whose own dynamics amplify the last bits
# (tools/sens_move_to.py) then ends elsewhere -- the comparison is only sharp with the form pinned.
FORM = sys.argv[3] if len(sys.argv) > 3 else "default"
if FORM != "default":
    from mink_amd import _lib as L
    L.set_option("coop", 1 if FORM == "coop" else 0)
    L.set_option("handoff", 0)
print(f"# one-launch waypoint reaching against the step-by-step loop, kernel form: {FORM}\n")
bad = 0
worst = 0.0
for k in range(SETS):
    rng = np.random.default_rng([555, k])
    name = ("tidybot_cube", "mobile_kinova")[k % 2]
    model = mink.load_builtin(name)
    q0 = W.sample_q(model, rng, B)
    frame = mink.FrameTask("pinch_site", "site", float(10 ** rng.uniform(-0.3, 0.3)), float(10 ** rng.uniform(-0.3, 0.3)),
                           lm_damping=float(rng.choice([1e-2, 1.0])))
    posture = mink.PostureTask(model, cost=W.posture_cost(model))
    posture.set_target(W.home_q(model))
    limits = [mink.ConfigurationLimit(model)]
    if rng.integers(2):
        limits.append(mink.VelocityLimit(model, {n: v for n, v in W.VELOCITY_LIMITS.items() if model.joint_id(n) >= 0}))
    wp = mink.LinearWaypointReachConfig(float(rng.choice([0.005, 0.01, 0.03])), float(rng.choice([0.04, 0.08, 0.2])),
                                        float(rng.choice([0.03, 0.05, 0.2])), 0.3)
    max_steps, max_iters = int(rng.choice([5, 12, 30])), int(rng.choice([5, 20]))
    dt, damping = float(rng.choice([1 / 200, 2e-2])), float(rng.choice([1e-6, 1e-3]))
    cfg_a, cfg_b = mink.Configuration(model, q0), mink.Configuration(model, q0)
    start = cfg_a.get_transform_frame_to_world("pinch_site", "site")
    spread = float(rng.choice([0.1, 0.3]))
    goal_pos = start.translation() + torch.tensor(rng.uniform(-spread, spread, (B, 3)), device="cuda:0")
    goal_rot = start.rotation() @ SO3.exp(torch.tensor(rng.uniform(-0.6, 0.6, (B, 3)), device="cuda:0"))
    a = mink.move_to(cfg_a, frame, [frame, posture], goal_pos, goal_rot, wp, dt=dt, damping=damping, max_steps=max_steps,
                     max_iters=max_iters, limits=limits)
    b = rollout._move_to_stepwise(cfg_b, frame, [frame, posture], goal_pos, goal_rot, wp, dt, "quadprog", damping, max_steps,
                                  max_iters, 1e-4, 1e-4, limits)
    same = torch.equal(a["steps"], b["steps"]) and torch.equal(a["reached"], b["reached"]) and torch.equal(a["solves"], b["solves"])
    per = (cfg_a.q - cfg_b.q).abs().amax(dim=1)
    dq = float(per.max())
    n_counts = int(((a["steps"] != b["steps"]) | (a["reached"] != b["reached"]) | (a["solves"] != b["solves"])).sum())
    n_q = int((per > 1e-8).sum())
    # rollouts that took the same decisions everywhere (same counts): their configurations must agree
    agree = (a["steps"] == b["steps"]) & (a["solves"] == b["solves"]) & (a["reached"] == b["reached"])
    dq_same = float(per[agree].max())
    worst = max(worst, dq_same)
    bad += 0 if n_counts <= B // 2000 and n_q <= B // 1000 else 1
    print(f"set {k} ({name}, dt {dt:.3g}, damping {damping:.0e}, steps <= {max_steps}, iters <= {max_iters}, wp {wp.pos_threshold}/"
          f"{wp.pos_step_size}/{wp.rot_threshold}, vel limit {len(limits) > 1}): rollouts with other counts {n_counts}, with |dq| > 1e-8 {n_q} (max {dq:.1e}; "
          f"median over all {float(per.median()):.1e}), reached {float(a['reached'].double().mean()):.3f}, mean steps {float(a['steps'].double().mean()):.2f}, "
          f"mean solves {float(a['solves'].double().mean()):.1f}", flush=True)
print(f"{SETS} sets x {B} rollouts: {bad} sets with more than a handful of diverging rollouts")
sys.exit(1 if bad else 0)
