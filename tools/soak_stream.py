"""Soak test (GPU box): the lane-streaming loop kernel against the block-per-64 kernel on the
same inputs (bit-identical by construction: a problem's arithmetic does not depend on the lane
or the company it runs in), over random parameter sets at 262 144 problems each.

    PYTHONPATH=. python tools/soak_stream.py [sets]"""
import os
import sys

import numpy as np
from mink_amd import _lib as L
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_parity as S
from mink_amd import workloads as W
from mink_amd.engine import DeviceProblem

SETS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = 1 << 18
bad = 0
for k in range(SETS):
    rng = np.random.default_rng([4242, k])
    name = ("tidybot_cube", "mobile_kinova", "tidybot_open")[k % 3]
    model, dm = S.models[name], S.dms[name]
    p = S.draw(rng, name)
    tasks, limits, _ = S.specs(model, p)
    prob = DeviceProblem(dm, tasks, limits)
    q0 = torch.tensor(W.sample_q(model, rng, B), device=S.dev)
    site = model.site_id("pinch_site")
    tg = torch.tensor(W.offset_targets(dm.frame_pose(q0, 2, site).cpu().numpy(), rng), device=S.dev)
    max_iters = int(rng.choice([5, 20]))
    res = []
    for knob in (-1, 0):
        with L.options(stream_blocks=knob, coop=0):
            q = q0.clone()
            r = prob.iterate_(q, tg, p["dt"], p["damping"], max_iters, 1e-4, 1e-4, want_v=True)
            torch.cuda.synchronize()
        res.append((q, r))
    (qa, ra), (qb, rb) = res
    same = (torch.equal(qa, qb) and torch.equal(ra["iters"], rb["iters"]) and torch.equal(ra["status"], rb["status"])
            and torch.equal(ra["err"], rb["err"]) and torch.equal(ra["v"], rb["v"]))
    bad += 0 if same else 1
    print(f"set {k} ({name}, dt {p['dt']:.3g}, damping {p['damping']:.0e}, max_iters {max_iters}): "
          f"{'bit-identical' if same else 'DIFFERENT'}; mean solves {float(ra['iters'].double().mean()):.2f}, "
          f"status nonzero {int((ra['status'] != 0).sum())}", flush=True)
print(f"{SETS} sets x {B} problems: {bad} sets differ")
sys.exit(1 if bad else 0)
