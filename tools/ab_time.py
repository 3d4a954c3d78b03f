"""Kernel tuning helper: time the fused kernels of the library named by MINK_AMD_LIB
(or the in-tree build) on the headline workload, interleaving nothing else."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

def main():
    model = load_builtin("tidybot_cube")
    dm = DeviceModel(model, "cuda:0")
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    rng = np.random.default_rng(4000)
    B = 131072
    q0 = torch.tensor(W.sample_q(model, rng, B), device="cuda:0")
    tg = dm.frame_pose(torch.tensor(W.sample_q(model, rng, B), device="cuda:0"), 2, 0)
    near = torch.tensor(W.offset_targets(dm.frame_pose(q0, 2, 0).cpu().numpy(), rng), device="cuda:0")
    res = {}
    for name, targets in (("far", tg), ("near", near)):
        bufs = [q0.clone() for _ in range(8)]
        for i in range(2): prob.iterate_(bufs[i], targets, W.DT, W.DAMPING, 50)
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(2, 8): prob.iterate_(bufs[i], targets, W.DT, W.DAMPING, 50)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 6
        res[f"iter50_{name}_ms"] = round(dt * 1e3, 3)
        res[f"iter50_{name}_Gsolves"] = round(B * 50 / dt / 1e9, 3)
        v = torch.empty((B, model.nv), dtype=torch.float64, device="cuda:0"); st = torch.empty(B, dtype=torch.int32, device="cuda:0")
        for i in range(3): prob.solve(q0, targets, W.DT, W.DAMPING, v=v, status=st)
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(20): prob.solve(q0, targets, W.DT, W.DAMPING, v=v, status=st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        res[f"step_{name}_us"] = round(dt * 1e6, 1)
    q4, t4 = q0[:4096].contiguous(), tg[:4096].contiguous()
    v = torch.empty((4096, model.nv), dtype=torch.float64, device="cuda:0"); st = torch.empty(4096, dtype=torch.int32, device="cuda:0")
    for i in range(10): prob.solve(q4, t4, W.DT, W.DAMPING, v=v, status=st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(200): prob.solve(q4, t4, W.DT, W.DAMPING, v=v, status=st)
    torch.cuda.synchronize(); res["step4096_us"] = round((time.perf_counter() - t) / 200 * 1e6, 1)
    print(os.environ.get("MINK_AMD_LIB", "in-tree"), res, flush=True)

main()
