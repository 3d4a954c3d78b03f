"""Soak test (GPU box): the general (wave-per-problem) kernels on the floating-base test tree
(free + ball + hinge + slide joints) against the numpy restatement of mink.solve_ik, over randomly
drawn task / limit sets and parameters.

    PYTHONPATH=. python tools/soak_general.py [sets] [problems per set]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mink_amd import _lib as L
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_mjcf
from oracle import ik_oracle as o

SETS = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 96
model = load_mjcf(os.path.join(ROOT, "tests", "models", "float_arm.xml"))
dm = DeviceModel(model, "cuda:0")


def random_q(rng):
    q = model.qpos0.copy()
    for j in range(model.njnt):
        a, t = int(model.jnt_qposadr[j]), int(model.jnt_type[j])
        if t == 0:
            q[a:a + 3] += rng.uniform(-0.2, 0.2, 3)
            w = rng.normal(size=4)
            q[a + 3:a + 7] = w / np.linalg.norm(w)
        elif t == 1:
            w = rng.normal(size=4)
            q[a:a + 4] = w / np.linalg.norm(w)
        else:
            lo, hi = (model.jnt_range[j] if model.jnt_limited[j] else (-1.0, 1.0))
            q[a] = rng.uniform(lo + 0.05, hi - 0.05)
    return q


print(f"# soak: general kernels on the floating-base tree against the numpy restatement, {SETS} random sets x {B} problems\n")
print("| # | dt | damping | frame cost / gain / lm | posture cost | com task | relative task | vel limit | cfg gain | "
      "max abs dv | rel. | status nonzero |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
worst = 0.0
for k in range(SETS):
    rng = np.random.default_rng([99, k])
    dt = float(rng.choice([1 / 200, 1e-2, 5e-2]))
    damping = float(rng.choice([1e-12, 1e-6, 1e-3]))
    fc = (float(10 ** rng.uniform(-0.5, 0.5)), float(10 ** rng.uniform(-0.5, 0.5)))
    fg, flm = float(rng.uniform(0.3, 1.0)), float(rng.choice([0.0, 0.1, 1.0]))
    pcost = float(rng.choice([1e-3, 0.05]))
    with_com, with_rel, with_vel = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
    cg = float(rng.choice([0.5, 0.9, 1.0]))
    site = model.site_id("tool")
    q = np.array([random_q(rng) for _ in range(B)])
    qs = np.array([random_q(rng) for _ in range(B)])
    dq = torch.tensor(q, device="cuda:0")
    dqs = torch.tensor(qs, device="cuda:0")
    tgt = [dm.frame_pose(dqs, 2, site)]
    tasks = [dict(type=L.TASK_FRAME, frame_type=2, frame_id=site, cost=np.array([fc[0]] * 3 + [fc[1]] * 3), gain=fg,
                  lm_damping=flm),
             dict(type=L.TASK_POSTURE, cost=np.full(model.nv, pcost), target_q=model.qpos0.copy(), gain=1.0, lm_damping=0.0)]
    if with_rel:
        tasks.append(dict(type=L.TASK_RELATIVE_FRAME, frame_type=L.FRAME_TYPES["body"], frame_id=model.body_id("fore"),
                          root_type=L.FRAME_TYPES["body"], root_id=model.body_id("torso"), cost=np.full(6, 0.7), gain=0.8,
                          lm_damping=0.0))
        from mink_amd.lie import SE3
        t_f = dm.frame_pose(dqs, L.FRAME_TYPES["body"], model.body_id("fore"))
        t_r = dm.frame_pose(dqs, L.FRAME_TYPES["body"], model.body_id("torso"))
        tgt.append((SE3(t_r).inverse() @ SE3(t_f)).wxyz_xyz)
    if with_com:
        tasks.append(dict(type=L.TASK_COM, cost=np.array([0.3, 0.3, 0.1]), gain=0.5, lm_damping=0.0))
        tgt.append(torch.tensor(rng.normal(size=(B, 3)) * 0.1 + np.array([0, 0, 0.7]), device="cuda:0"))
    vel = {"elbow": 1.0, "extend": 0.2, "hip": 0.5} if with_vel else {}
    limits = [dict(type=L.LIMIT_CONFIGURATION, gain=cg, min_distance_from_limits=0.0)]
    if with_vel:
        limits.append(dict(type=L.LIMIT_VELOCITY, index=[int(model.jnt_dofadr[model.joint_id(n)]) for n in vel],
                           vmax=list(vel.values())))
    prob = DeviceProblem(dm, tasks, limits)
    tg = torch.cat(tgt, dim=1).contiguous()
    v, st = prob.solve(dq, tg, dt, damping)
    v, st, tgn = v.cpu().numpy(), st.cpu().numpy(), tg.cpu().numpy()
    mx = rel = 0.0
    for b in range(B):
        cfg = o.Configuration(model, q[b])
        ot = [o.FrameTask("tool", "site", fc[0], fc[1], gain=fg, lm_damping=flm, target=tgn[b, :7]),
              o.PostureTask(cost=np.full(model.nv, pcost), target_q=model.qpos0.copy())]
        off = 7
        if with_rel:
            ot.append(o.RelativeFrameTask("fore", "body", "torso", "body", 0.7, 0.7, gain=0.8, target=tgn[b, off:off + 7]))
            off += 7
        if with_com:
            ot.append(o.ComTask(cost=np.array([0.3, 0.3, 0.1]), target_com=tgn[b, off:off + 3], gain=0.5))
        ol = [o.ConfigurationLimit(model, gain=cg)] + ([o.VelocityLimit(model, vel)] if with_vel else [])
        ref = o.solve_ik(cfg, ot, dt, damping=damping, limits=ol)
        if st[b] == 0:
            d = float(np.abs(v[b] - ref).max())
            mx, rel = max(mx, d), max(rel, d / max(1.0, float(np.abs(ref).max())))
    worst = max(worst, rel)
    print(f"| {k} | {dt:.3g} | {damping:.0e} | {fc[0]:.2f},{fc[1]:.2f} / {fg:.2f} / {flm:.0e} | {pcost:.0e} | {with_com} | {with_rel} | "
          f"{with_vel} | {cg} | {mx:.2e} | {rel:.1e} | {int((st != 0).sum())} |", flush=True)
print(f"\nworst relative |dv| {worst:.3e} over {SETS * B} problems")
