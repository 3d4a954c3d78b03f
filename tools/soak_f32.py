"""Sweep (GPU box): the fused kernels in float32 arithmetic (option "f32") against the same
kernels in float64, which the parity tests and tools/soak_parity.py pin to the restatement of
the reference to 1e-9.

    python tools/soak_f32.py [trials] [problems per trial] > profiles/<name>.md

First the bench workload (TidyBot, damping 1e-3, dt 1/200) at several damping values, then the
random parameter sets of tools/soak_parity.py.  Per set: one solve_ik step (velocity compared,
absolute and relative to the size of the problem's own answer) and the callers' loop with the exit
test (frame-task error reached, configuration, solve counts)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

TRIALS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = "cuda:0"


def compare(prob, dm, model, q0, tg, dt, damping, label, max_iters=20):
    dq0, dtg = torch.tensor(q0, device=dev), torch.tensor(tg, device=dev)
    with L.options(coop=0, handoff=0):
        v64, st64 = prob.solve(dq0, dtg, dt, damping)
        q64 = dq0.clone()
        r64 = prob.iterate_(q64, dtg, dt, damping, max_iters, 1e-4, 1e-4)
        with L.options(f32=1):
            v32, st32 = prob.solve(dq0, dtg, dt, damping)
            q32 = dq0.clone()
            r32 = prob.iterate_(q32, dtg, dt, damping, max_iters, 1e-4, 1e-4)
    v64, v32, st64, st32 = v64.cpu().numpy(), v32.cpu().numpy(), st64.cpu().numpy(), st32.cpu().numpy()
    ok = (st64 == 0) & (st32 == 0)
    if not ok.any():
        print(f"| {label} | {dt:.3g} | {damping:.0e} | no problem solved in both precisions: float32 status 3 (pivot lost) on "
              f"{int((st32 == 3).sum())} of {len(st32)} |", flush=True)
        return 0.0, 0.0, 0.0, 0.0
    dv = np.abs(v32 - v64)[ok].max(axis=1)
    vmag = np.abs(v64[ok]).max(axis=1)
    rel = dv / np.maximum(1.0, vmag)
    # the loop: where both precisions end, and how far apart
    e64 = np.linalg.norm(r64["err"].cpu().numpy(), axis=1)
    e32 = np.linalg.norm(r32["err"].cpu().numpy(), axis=1)
    it64, it32 = r64["iters"].cpu().numpy(), r32["iters"].cpu().numpy()
    conv64, conv32 = it64 < max_iters, it32 < max_iters
    both = (r64["status"].cpu().numpy() == 0) & (r32["status"].cpu().numpy() == 0)
    dq = np.abs(q32.cpu().numpy() - q64.cpu().numpy()).max(axis=1)[both] if both.any() else np.zeros(1)
    e64, e32 = (e64[both], e32[both]) if both.any() else (np.zeros(1), np.zeros(1))
    print(f"| {label} | {dt:.3g} | {damping:.0e} | {dv.max():.2e} | {np.percentile(dv, 99):.2e} | {rel.max():.2e} | {vmag.max():.1e} | "
          f"{int((st32 != st64).sum())} | {int((st32 == 3).sum())} | {conv64.mean():.4f} / {conv32.mean():.4f} | "
          f"{int((it32 != it64).sum())} | {dq.max():.2e} | {np.percentile(dq, 99):.2e} | {np.abs(e32 - e64).max():.2e} | "
          f"{int((r32['status'].cpu().numpy() != r64['status'].cpu().numpy()).sum())} |", flush=True)
    return float(dv.max()), float(rel.max()), float(dq.max()), float(np.abs(e32 - e64).max())


def main():
    import soak_parity as sp
    print(f"# float32 arithmetic against float64: fused kernels, {B} problems per row\n")
    print("| set | dt | damping | max abs dv (1 step, rad/s) | 99 % | max rel. dv | max abs v | status differ | f32 status 3 (pivot lost) | "
          "loop converged f64 / f32 | solve counts differ | max abs dq after the loop | 99 % | max abs d(err norm) | loop status differ |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    model = load_builtin("tidybot_cube")
    dm = DeviceModel(model, dev)
    prob = DeviceProblem(dm, W.task_specs(model, True), W.limit_specs(model, True))
    assert prob.fused
    rng = np.random.default_rng(77)
    q0 = W.sample_q(model, rng, B)
    site = model.site_id("pinch_site")
    near = W.offset_targets(dm.frame_pose(torch.tensor(q0, device=dev), 2, site).cpu().numpy(), rng)
    far = dm.frame_pose(torch.tensor(W.sample_q(model, rng, B), device=dev), 2, site).cpu().numpy()
    head = [0.0, 0.0, 0.0, 0.0]
    for damping in (1e-1, 1e-3, 1e-5, 1e-8, 1e-12):
        for name, tg in (("bench workload, near targets", near), ("bench workload, far targets", far)):
            r = compare(prob, dm, model, q0, tg, W.DT, damping, name)
            if damping == 1e-3:
                head = [max(a, b) for a, b in zip(head, r)]
    print(f"\nbench workload at its own damping (1e-3): max abs dv {head[0]:.3e} rad/s, relative {head[1]:.3e}, "
          f"max abs dq after the loop {head[2]:.3e}, max abs difference of the error norm reached {head[3]:.3e}\n")
    print("| set | dt | damping | max abs dv (1 step, rad/s) | 99 % | max rel. dv | max abs v | status differ | f32 status 3 (pivot lost) | "
          "loop converged f64 / f32 | solve counts differ | max abs dq after the loop | 99 % | max abs d(err norm) | loop status differ |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    t0 = time.time()
    worst = {}
    for trial in range(TRIALS):
        rng = np.random.default_rng([2024, trial])
        name = ("tidybot_cube", "mobile_kinova", "tidybot_open")[trial % 3]
        model, dm = sp.models[name], sp.dms[name]
        p = sp.draw(rng, name)
        tasks, limits, _ = sp.specs(model, p)
        prob = DeviceProblem(dm, tasks, limits)
        if not prob.fused:
            continue
        q0 = W.sample_q(model, rng, B)
        dq0 = torch.tensor(q0, device=dev)
        site = model.site_id("pinch_site")
        if p["near"]:
            tg = W.offset_targets(dm.frame_pose(dq0, 2, site).cpu().numpy(), rng)
        else:
            tg = dm.frame_pose(torch.tensor(W.sample_q(model, rng, B), device=dev), 2, site).cpu().numpy()
        label = (f"{trial} {name} gain {p['frame_gain']:.2f} lm {p['frame_lm']:.0e} posture {p['posture_cost'][3]:.0e} "
                 f"{'near' if p['near'] else 'far'}")
        r = compare(prob, dm, model, q0, tg, p["dt"], p["damping"], label, max_iters=10)
        key = ("frame task with lm_damping = 1 and damping >= 1e-3 (what the callers of the reference set)"
               if p["frame_lm"] >= 1.0 and p["damping"] >= 1e-3
               else "other sets (error grows with the condition number of the Hessian)")
        worst[key] = [max(a, b) for a, b in zip(worst.get(key, [0, 0, 0, 0]), r)]
    for k, w in worst.items():
        print(f"\n{k}: max abs dv {w[0]:.3e} rad/s, relative {w[1]:.3e}, max abs dq {w[2]:.3e}, max abs d(err norm) {w[3]:.3e}")
    print(f"\n{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
