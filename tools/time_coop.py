"""Lane-per-problem against cooperative (16 lanes per problem) kernels on the small-batch
workloads: one solve_ik step at several batch sizes and the callers' <= 20-iteration loop
(device time per launch from HIP events, and wall time per call for B = 1)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

dev = "cuda:0"
model = load_builtin("tidybot_cube")
dm = DeviceModel(model, dev)
rng = np.random.default_rng(4)
BMAX = 65536
q0 = torch.tensor(W.sample_q(model, rng, BMAX), device=dev)
site = model.site_id("pinch_site")
far = dm.frame_pose(torch.tensor(W.sample_q(model, rng, BMAX), device=dev), 2, site)
near = torch.tensor(W.offset_targets(dm.frame_pose(q0, 2, site).cpu().numpy(), rng), device=dev)


def ev_time(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n   # us


p2 = DeviceProblem(dm, W.task_specs(model, True), W.limit_specs(model, True))
pn = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))
print("single step, far targets, Frame+Posture+Damping, Cfg+Vel limits: us per launch")
for B in (1, 64, 1024, 4096, 8192, 16384, 32768, 65536):
    q, t = q0[:B].contiguous(), far[:B].contiguous()
    v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    row = []
    for form in (0, 1):
        with L.options(coop=form):
            row.append(ev_time(lambda: p2.solve(q, t, W.DT, W.DAMPING, v=v, status=st)))
    print(f"  B={B:6d}  lane {row[0]:8.2f}  coop {row[1]:8.2f}")

print("callers' loop (<= 20 iterations, 1e-4 exit), near targets: us per launch, mean solves")
for B in (1, 64, 1024, 4096, 16384, 32768, 65536):
    t = near[:B].contiguous()
    out = dict(iters=torch.empty(B, dtype=torch.int32, device=dev), err=torch.empty((B, 6), dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev))
    qs = [q0[:B].clone() for _ in range(60)]
    row = []
    for form in (0, 1, -1):   # lane kernels alone, cooperative kernels alone, the library's choice
        with L.options(coop=form):
            for k, qq in enumerate(qs):
                qq.copy_(q0[:B])
            it = iter(qs)
            us = ev_time(lambda: pn.iterate_(next(it), t, W.DT, W.DAMPING, 20, 1e-4, 1e-4, out=out), n=50, warm=5)
            row.append((us, float(out["iters"].double().mean()), int(out["iters"].max())))
    print(f"  B={B:6d}  lane {row[0][0]:8.2f} (mean {row[0][1]:.2f} max {row[0][2]})  coop {row[1][0]:8.2f}"
          f"  default {row[2][0]:8.2f} (mean {row[2][1]:.2f})  -> default {B * row[2][1] / row[2][0]:.1f} M solves/s")

# wall time per call at B = 1 (launch + sync), as the 200 Hz teleop loop sees it
q1, t1 = q0[:1].clone(), near[:1].contiguous()
o1 = dict(iters=torch.empty(1, dtype=torch.int32, device=dev), err=torch.empty((1, 6), dtype=torch.float64, device=dev),
          status=torch.empty(1, dtype=torch.int32, device=dev))
for form in (0, 1):
    with L.options(coop=form):
        def call():
            q1.copy_(q0[:1])
            pn.iterate_(q1, t1, W.DT, W.DAMPING, 20, 1e-4, 1e-4, out=o1)
            torch.cuda.synchronize()
        for _ in range(20):
            call()
        t0 = time.perf_counter()
        for _ in range(200):
            call()
        print(f"B=1 loop20 wall, coop={form}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us, iters {int(o1['iters'][0])}")
