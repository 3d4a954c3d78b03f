"""Tuning helper: time the wave-per-problem kernels on config-4 style workloads."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

model = load_builtin("tidybot_cube")
dm = DeviceModel(model, "cuda:0")
tasks, limits = W.config4_specs(model)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
q, tg = W.config4_inputs(model, dm, B, 3)
variants = {
    "config4 (frame+rel+posture, cfg+vel+collision)": (tasks, limits, tg),
    "no collision (frame+rel+posture, cfg+vel)": (tasks, limits[:2], tg),
    "rel+posture, cfg+vel": (tasks[1:], limits[:2], tg[:, 7:].contiguous()),
    "rel+posture, no limits": (tasks[1:], [], tg[:, 7:].contiguous()),
    "collision only": ([], limits[2:], None),
}
for name, (t, l, targets) in variants.items():
    prob = DeviceProblem(dm, t, l)
    v = torch.empty((B, model.nv), dtype=torch.float64, device="cuda:0")
    st = torch.empty(B, dtype=torch.int32, device="cuda:0")
    for _ in range(2):
        prob.solve(q, targets, W.DT, W.DAMPING, v=v, status=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 4
    for _ in range(n):
        prob.solve(q, targets, W.DT, W.DAMPING, v=v, status=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        prob.build_qp(q[:16384].contiguous(), None if targets is None else targets[:16384].contiguous(), W.DT, W.DAMPING)
    torch.cuda.synchronize(); db = (time.perf_counter() - t0) / n * (B / 16384)
    print(f"{name:48s} fused={prob.fused} rows={prob.n_rows:3d} solve {dt*1e3:8.3f} ms ({B/dt/1e6:7.2f} M/s)  "
          f"build-only ~{db*1e3:8.3f} ms  bad={int((st != 0).sum())}", flush=True)
