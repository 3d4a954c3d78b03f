"""The near-target loop (<= 20 iterations, 1e-4 exit) at several batch sizes for different
hand-off thresholds (live problems per wave at which the lane kernel passes them on)."""
import sys
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
BMAX = max([int(a) for a in sys.argv[1:]] + [131072])
q0 = torch.tensor(W.sample_q(model, rng, BMAX), device=dev)
near = torch.tensor(W.offset_targets(dm.frame_pose(q0, 2, model.site_id("pinch_site")).cpu().numpy(), rng), device=dev)
pn = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))


def ev_time(fn, n=30, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for B in [int(a) for a in sys.argv[1:]] or [32768, 65536, 98304]:
    t = near[:B].contiguous()
    out = dict(iters=torch.empty(B, dtype=torch.int32, device=dev), err=torch.empty((B, 6), dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev))
    qs = [q0[:B].clone() for _ in range(34)]
    line = []
    for h in (0, 2, 4, 6, 8, 12, 16, 24, 32):
        with L.options(handoff=h, coop_max_batch=0):
            for qq in qs:
                qq.copy_(q0[:B])
            it = iter(qs)
            us = ev_time(lambda: pn.iterate_(next(it), t, W.DT, W.DAMPING, 20, 1e-4, 1e-4, out=out))
            line.append(f"{h}: {us:.1f}")
    solves = float(out["iters"].double().sum())
    print(f"B={B}: us per launch by hand-off threshold  " + "  ".join(line) + f"   ({solves / B:.2f} solves per problem)")
