"""Soak test (GPU box): mk_fk_batch / mk_frame_jacobian_batch / mk_integrate_batch / mk_check_limits_batch on
every body, geom and site frame of the shipped models and the floating-base test tree, at random
configurations (beyond joint limits included), against the numpy restatement of the MuJoCo helpers.

    PYTHONPATH=. python tools/soak_fk.py [configurations per model]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mink_amd import _lib as L
from mink_amd.engine import DeviceModel
from mink_amd.mjcf import load_builtin, load_mjcf
from oracle import ik_oracle as o
from oracle import mjlite as mj

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(2718)
models = {n: load_builtin(n) for n in ("tidybot_cube", "mobile_kinova", "tidybot_open")}
models["float_arm"] = load_mjcf(os.path.join(ROOT, "tests", "models", "float_arm.xml"))
models["convex_arm"] = load_mjcf(os.path.join(ROOT, "tests", "models", "convex_arm.xml"))
for name, model in models.items():
    dm = DeviceModel(model, "cuda:0")
    q = np.tile(model.qpos0, (B, 1))
    for j in range(model.njnt):
        a, t = int(model.jnt_qposadr[j]), int(model.jnt_type[j])
        if t == 0:
            q[:, a:a + 3] += rng.uniform(-2.0, 2.0, (B, 3))
            w = rng.normal(size=(B, 4))
            q[:, a + 3:a + 7] = w / np.linalg.norm(w, axis=1, keepdims=True)
        elif t == 1:
            w = rng.normal(size=(B, 4))
            q[:, a:a + 4] = w / np.linalg.norm(w, axis=1, keepdims=True)
        else:
            q[:, a] = rng.uniform(-7.0, 7.0, B) if t == 3 else rng.uniform(-0.5, 0.5, B)
    dq = torch.tensor(q, device="cuda:0")
    frames = [("body", i) for i in range(model.nbody)] + [("geom", i) for i in range(model.ngeom)] + \
             [("site", i) for i in range(model.nsite)]
    dp = dj = 0.0
    # the restatement's FK is per configuration: a 16-configuration sample per frame
    sample = list(range(0, B, max(1, B // 16)))
    cfgs = {b: o.Configuration(model, q[b]) for b in sample}
    names = {"body": model.body_names, "geom": model.geom_names, "site": model.site_names}
    for ftype, fid in frames:
        fname = names[ftype][fid]
        body = fid if ftype == "body" else int(model.geom_bodyid[fid] if ftype == "geom" else model.site_bodyid[fid])
        if not fname or model.body_mocapid[body] >= 0:    # unnamed frames have no mink name; mocap frames are the host's
            continue
        pose = dm.frame_pose(dq, L.FRAME_TYPES[ftype], fid).cpu().numpy()
        jac = dm.frame_jacobian(dq, L.FRAME_TYPES[ftype], fid).cpu().numpy()
        for b in sample:
            ref = cfgs[b].get_transform_frame_to_world(fname, ftype)
            sgn = np.sign(ref[:4] @ pose[b, :4]) or 1.0
            dp = max(dp, float(np.abs(pose[b, :4] * sgn - ref[:4]).max()), float(np.abs(pose[b, 4:] - ref[4:]).max()))
            dj = max(dj, float(np.abs(jac[b] - cfgs[b].get_frame_jacobian(fname, ftype)).max()))
    # integration and limit check
    v = rng.normal(size=(B, model.nv))
    qi = dq.clone()
    dm.integrate_(qi, torch.tensor(v, device="cuda:0"), 0.05)
    di = 0.0
    for b in sample:
        ref = q[b].copy()
        mj.mj_integratePos(model, ref, v[b], 0.05)
        di = max(di, float(np.abs(qi[b].cpu().numpy() - ref).max()))
    first = dm.first_limit_violation(dq, 1e-6).cpu().numpy()
    nlim = 0
    for b in sample:
        exp = -1
        for j in range(model.njnt):
            if model.jnt_type[j] in (0,) or not model.jnt_limited[j]:
                continue
            a = int(model.jnt_qposadr[j])
            if q[b, a] < model.jnt_range[j, 0] - 1e-6 or q[b, a] > model.jnt_range[j, 1] + 1e-6:
                exp = j
                break
        nlim += int(first[b] != exp)
    print(f"{name}: {len(frames)} frames x {len(sample)} sampled configurations: max pose difference {dp:.1e}, max Jacobian "
          f"difference {dj:.1e}; integration {di:.1e}; first-violated-joint mismatches {nlim}", flush=True)
