"""Tuning helper (GPU box): for one parameter set of tools/soak_parity.py, the KKT residuals of the
GPU's and of the C restatement's answers on the dense QP the library exports -- which of two
answers that differ is the better one."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_parity as S
from mink_amd import workloads as W
from mink_amd.engine import DeviceProblem


def residual(P, c, lo, hi, x):
    g = torch.einsum("bij,bj->bi", P, x) + c
    tol = 1e-12
    at_hi = x >= hi - tol * hi.abs().clamp(min=1.0)
    at_lo = x <= lo + tol * lo.abs().clamp(min=1.0)
    stat = torch.where(at_hi, g.clamp(min=0.0), torch.where(at_lo, (-g).clamp(min=0.0), g.abs()))
    infeas = torch.maximum(x - hi, lo - x).clamp(min=0.0)
    return stat.max(dim=1).values, infeas.max(dim=1).values


def main():
    trial = int(sys.argv[1])
    B = 16384
    rng = np.random.default_rng([2024, trial])
    name = ("tidybot_cube", "mobile_kinova", "tidybot_open")[trial % 3]
    model, dm = S.models[name], S.dms[name]
    p = S.draw(rng, name)
    tasks, limits, orc = S.specs(model, p)
    prob = DeviceProblem(dm, tasks, limits)
    q0 = W.sample_q(model, rng, B)
    dq0 = torch.tensor(q0, device=S.dev)
    site = model.site_id("pinch_site")
    if p["near"]:
        tg = W.offset_targets(dm.frame_pose(dq0, 2, site).cpu().numpy(), rng)
    else:
        tg = dm.frame_pose(torch.tensor(W.sample_q(model, rng, B), device=S.dev), 2, site).cpu().numpy()
    dtg = torch.tensor(tg, device=S.dev)
    v, st = prob.solve(dq0, dtg, p["dt"], p["damping"])
    v_ref, _ = orc.solve_ik(q0, tg, p["dt"], p["damping"], nthreads=S.THREADS)
    P, c, G, h = prob.build_qp(dq0, dtg, p["dt"], p["damping"])
    inf = torch.full_like(h, float("inf"))[:, :, None]
    hi = torch.where(G > 0.5, h[:, :, None], inf).min(dim=1).values
    lo = torch.where(G < -0.5, -h[:, :, None], -inf).max(dim=1).values
    xg, xr = v * p["dt"], torch.tensor(v_ref, device=S.dev) * p["dt"]
    sg, ig = residual(P, c, lo, hi, xg)
    sr, ir = residual(P, c, lo, hi, xr)
    diff = (xg - xr).abs().max(dim=1).values / p["dt"]
    k = int(diff.argmax())
    obj = lambda x: 0.5 * torch.einsum("bi,bij,bj->b", x, P, x) + (c * x).sum(dim=1)
    print(f"trial {trial} ({name}): worst |dv| {float(diff[k]):.3e} at problem {k}")
    print(f"  stationarity residual there: gpu {float(sg[k]):.3e}  c-oracle {float(sr[k]):.3e}   (|c| max {float(c[k].abs().max()):.3e})")
    print(f"  objective gpu - oracle there: {float(obj(xg)[k] - obj(xr)[k]):.3e}")
    print(f"  over the batch: stationarity max gpu {float(sg.max()):.3e}  oracle {float(sr.max()):.3e}; infeasibility gpu {float(ig.max()):.3e} oracle {float(ir.max()):.3e}")


if __name__ == "__main__":
    main()
