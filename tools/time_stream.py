"""Tuning helper (GPU): time of the near-target early-exit loop at 2^20 problems for the
block-per-64 kernel and for the lane-streaming kernel at several refill thresholds / numbers of
persistent waves (options "stream_blocks" / "stream_refill")."""
import os
import sys
import time

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
out = dict(iters=torch.empty(B, dtype=torch.int32, device=dev), err=torch.empty((B, 6), dtype=torch.float64, device=dev),
           status=torch.empty(B, dtype=torch.int32, device=dev))
qs = [q0.clone() for _ in range(4)]


def run(env):
    L.set_option("stream_blocks", env.get("stream_blocks", -1))
    L.set_option("stream_refill", env.get("stream_refill", 16))
    L.set_option("coop", 0)
    for b in qs:
        b.copy_(q0)
    prob.iterate_(qs[0], tg, W.DT, W.DAMPING, 20, 1e-4, 1e-4, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(1, 4):
        prob.iterate_(qs[i], tg, W.DT, W.DAMPING, 20, 1e-4, 1e-4, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 3 * 1e3
    solves = float(out["iters"].sum())
    print(f"{env}: {ms:.3f} ms  {solves / ms / 1e6:.2f} G solves/s  mean iters {solves / B:.2f}", flush=True)


run({"stream_blocks": 0})
for blocks in (1024, 2048):
    for refill in (1, 4, 8, 16, 24, 32, 48):
        run({"stream_blocks": blocks, "stream_refill": refill})
