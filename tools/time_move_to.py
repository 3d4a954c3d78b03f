"""Tuning helper (GPU): kernel time of mk_move_to_batch for B rollouts (default 65536 and 2^20)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import mink_amd as mk
from mink_amd import workloads as W
from mink_amd.mjcf import load_builtin

dev = "cuda:0"
model = load_builtin("tidybot_cube")
for B in [int(a) for a in sys.argv[1:]] or [65536, 1 << 20]:
    rng = np.random.default_rng(4)
    q0 = torch.tensor(W.sample_q(model, rng, B), device=dev)
    cfg = mk.Configuration(model, q0.clone(), device=dev)
    start = cfg.get_transform_frame_to_world("pinch_site", "site")
    g = torch.Generator(device="cpu").manual_seed(4)
    goal_pos = start.translation() + (torch.rand((B, 3), generator=g, dtype=torch.float64) * 0.3 - 0.15).to(dev)
    goal_rot = start.rotation() @ mk.SO3.exp((torch.rand((B, 3), generator=g, dtype=torch.float64) * 0.8 - 0.4).to(dev))
    ftask = mk.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
    ptask = mk.PostureTask(model, cost=W.posture_cost(model))
    ptask.set_target(W.home_q(model))
    wcfg = mk.LinearWaypointReachConfig(0.01, 0.08, 0.05, 0.3)
    for rep in range(3):
        cfg = mk.Configuration(model, q0.clone(), device=dev)
        torch.cuda.synchronize()
        t = time.perf_counter()
        ro = mk.move_to(cfg, ftask, [ftask, ptask], goal_pos, goal_rot, wcfg, max_steps=50)
        torch.cuda.synchronize()
        t = time.perf_counter() - t
    solves = float(ro["solves"].sum())
    print(f"B={B}: {t * 1e3:.3f} ms, {B / t / 1e6:.1f} M rollouts/s, {solves / t / 1e9:.3f} G solves/s, "
          f"reached {float(ro['reached'].double().mean()):.4f}, steps mean {float(ro['steps'].double().mean()):.2f} "
          f"max {int(ro['steps'].max())}, solves/rollout {solves / B:.2f} max {int(ro['solves'].max())}", flush=True)

# steady-state throughput: several launches in flight on separate streams, so that the last
# waves of one launch (a rollout that needs hundreds of solves holds one lane for milliseconds)
# overlap with the bulk of the next
from mink_amd._problems import get_problem  # noqa: E402
from mink_amd.solve_ik import _resolve_limits  # noqa: E402
B, NB = 1 << 18, 8
rng = np.random.default_rng(4)
q0 = torch.tensor(W.sample_q(model, rng, B), device=dev)
cfg = mk.Configuration(model, q0.clone(), device=dev)
start = cfg.get_transform_frame_to_world("pinch_site", "site")
g = torch.Generator(device="cpu").manual_seed(4)
goal_pos = start.translation() + (torch.rand((B, 3), generator=g, dtype=torch.float64) * 0.3 - 0.15).to(dev)
goal_rot = start.rotation() @ mk.SO3.exp((torch.rand((B, 3), generator=g, dtype=torch.float64) * 0.8 - 0.4).to(dev))
goals = mk.SE3.from_rotation_and_translation(goal_rot, goal_pos).wxyz_xyz.contiguous()
ftask.set_target(mk.SE3(goals))
prob, _ = get_problem(cfg, [ftask, ptask], _resolve_limits(cfg, None))
streams = [torch.cuda.Stream() for _ in range(4)]
qs = [q0.clone() for _ in range(NB)]
torch.cuda.synchronize()
for rep in range(2):
    for b in qs:
        b.copy_(q0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    outs = []
    for i, b in enumerate(qs):
        with torch.cuda.stream(streams[i % 4]):
            outs.append(prob.move_to_(b, goals, wcfg, 1.0 / 200.0, 1e-3, 50, 20, 1e-4, 1e-4))
    torch.cuda.synchronize()
    t = time.perf_counter() - t
solves = sum(float(o["solves"].sum()) for o in outs)
print(f"{NB} launches of {B} on 4 streams: {t * 1e3:.3f} ms, {NB * B / t / 1e6:.1f} M rollouts/s, {solves / t / 1e9:.3f} G solves/s")
