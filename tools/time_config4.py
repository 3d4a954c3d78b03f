"""Timing of benchmark config 4 (Frame + RelativeFrame + Posture, Cfg + Vel + Collision) through the
staged solve.  option "lane_rows" = 0 skips the lane-per-problem rows stage (A/B)."""
import sys, time
import torch
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
model = load_builtin("tidybot_cube")
dev = torch.device("cuda:0")
dm = DeviceModel(model, dev)
tasks, limits = W.config4_specs(model)
q, t = W.config4_inputs(model, dm, n, 7)
p = DeviceProblem(dm, tasks, limits)
v = torch.empty((n, p.nv), dtype=torch.float64, device=dev)
s = torch.empty((n,), dtype=torch.int32, device=dev)
for _ in range(2):
    p.solve(q, t, W.DT, W.DAMPING, v, s)
torch.cuda.synchronize()
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    p.solve(q, t, W.DT, W.DAMPING, v, s)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / reps
print(f"config 4, B={n}: {ms:.3f} ms per solve_ik batch = {n / ms / 1e3:.1f} M solves/s; status nonzero {int((s != 0).sum())}")
