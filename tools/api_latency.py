"""Tuning helper: per-call latency of the mink-named API for one configuration (teleop path)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import mink_amd as mink
from mink_amd import workloads as W

model = mink.load_builtin("tidybot_cube")
cfg = mink.Configuration(model)
cfg.update_from_keyframe("home")
frame = mink.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
posture = mink.PostureTask(model, cost=W.posture_cost(model))
posture.set_target_from_configuration(cfg)
T = cfg.get_transform_frame_to_world("pinch_site", "site")
frame.set_target(T @ mink.SE3.exp(torch.tensor([0.05, 0.02, -0.03, 0.1, 0.0, 0.1], dtype=torch.float64)))
tasks = [frame, posture]
dt = 1 / 200

def loop(n):
    for _ in range(n):
        v = mink.solve_ik(cfg, tasks, dt, "quadprog", 1e-3)
        cfg.integrate_inplace(v, dt)
        e = frame.compute_error(cfg)

loop(20)
torch.cuda.synchronize(); t = time.perf_counter(); loop(200); torch.cuda.synchronize()
print("per teleop inner step (solve_ik + integrate_inplace + compute_error): %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
t = time.perf_counter()
for _ in range(200): v = mink.solve_ik(cfg, tasks, dt, "quadprog", 1e-3)
print("solve_ik alone: %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
t = time.perf_counter()
for _ in range(200): cfg.integrate_inplace(v, dt)
print("integrate_inplace alone: %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
t = time.perf_counter()
for _ in range(200): e = frame.compute_error(cfg)
print("compute_error alone: %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
cfg.update_from_keyframe("home")
t = time.perf_counter()
for _ in range(50):
    cfg.update_from_keyframe("home")
    r = mink.solve_ik_iterate(cfg, tasks, dt, "quadprog", 1e-3, max_iters=20)
print("solve_ik_iterate (<=20 steps, one launch): %.1f us" % ((time.perf_counter() - t) / 50 * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); loop(100); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
