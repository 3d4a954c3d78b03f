"""Timing of the wave-per-problem kernel on benchmark config 4 (staging off), with and without
the collision limit; MINK_AMD_LIB selects a tuning build of the library."""
import os, sys, time
from mink_amd import _lib as L
import torch
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

L.set_option("staged", 0)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
model = load_builtin("tidybot_cube")
dev = torch.device("cuda:0")
dm = DeviceModel(model, dev)
tasks, limits = W.config4_specs(model)
q, t = W.config4_inputs(model, dm, n, 7)


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


out = [os.path.basename(os.environ.get("MINK_AMD_LIB", "default"))]
for name, lim in (("coll", limits), ("nocoll", limits[:-1])):
    p = DeviceProblem(dm, tasks, lim)
    v = torch.empty((n, p.nv), dtype=torch.float64, device=dev)
    s = torch.empty((n,), dtype=torch.int32, device=dev)
    ts = timed(lambda: p.solve(q, t, W.DT, W.DAMPING, v, s))
    out.append(f"{name}: {ts:.3f} ms ({ts*1e6/n:.1f} ns/problem)")
near = int(n * 0.25)
p = DeviceProblem(dm, tasks, limits)
for name, sl in (("near", slice(0, near)), ("far", slice(near, n))):
    qq, tt = q[sl].contiguous(), t[sl].contiguous()
    m = qq.shape[0]
    v = torch.empty((m, p.nv), dtype=torch.float64, device=dev)
    s = torch.empty((m,), dtype=torch.int32, device=dev)
    ts = timed(lambda: p.solve(qq, tt, W.DT, W.DAMPING, v, s))
    out.append(f"{name}: {ts*1e6/m:.1f} ns/problem")
print(" | ".join(out))
