"""Tuning helper: where the time of mink_amd.move_to goes (outer steps, kernel vs glue)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import mink_amd as mk
from mink_amd import workloads as W

model = mk.load_builtin("tidybot_cube")
Br = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(4000)
q0 = torch.tensor(W.sample_q(model, rng, Br), device="cuda:0")

def run():
    rcfg = mk.Configuration(model, q0.clone())
    start = rcfg.get_transform_frame_to_world("pinch_site", "site")
    g = torch.Generator(device="cpu").manual_seed(4000)
    goal_pos = start.translation() + (torch.rand((Br, 3), generator=g, dtype=torch.float64) * 0.3 - 0.15).to("cuda:0")
    goal_rot = start.rotation() @ mk.SO3.exp((torch.rand((Br, 3), generator=g, dtype=torch.float64) * 0.8 - 0.4).to("cuda:0"))
    ftask = mk.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
    ptask = mk.PostureTask(model, cost=W.posture_cost(model)); ptask.set_target(W.home_q(model))
    wcfg = mk.LinearWaypointReachConfig(0.01, 0.08, 0.05, 0.3)
    torch.cuda.synchronize(); t = time.perf_counter()
    ro = mk.move_to(rcfg, ftask, [ftask, ptask], goal_pos, goal_rot, wcfg, max_steps=50)
    torch.cuda.synchronize()
    return time.perf_counter() - t, ro

run()
dt, ro = run()
print("B=%d  %.1f ms  outer steps max %d mean %.2f  solves total %d" % (Br, dt * 1e3, int(ro["steps"].max()), float(ro["steps"].double().mean()), int(ro["solves"].sum())))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
