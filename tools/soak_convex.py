"""Soak test (GPU box): CollisionAvoidanceLimit rows on convex pairs without a closed form (mesh
hulls, cylinders, ellipsoids: support-mapping distance in the kernels) against the numpy
restatement (oracle/convex.py), on many random configurations.

    PYTHONPATH=. python tools/soak_convex.py [configurations per model]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin, load_mjcf
from oracle import ik_oracle as o

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DEV = "cuda:0"


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)


def spec_of(lim):
    return dict(type=L.LIMIT_COLLISION, index=np.array(lim.geom_id_pairs, dtype=np.int32).reshape(-1), gain=lim.gain,
                minimum_distance_from_collisions=lim.minimum_distance_from_collisions,
                collision_detection_distance=lim.collision_detection_distance, bound_relaxation=lim.bound_relaxation)


def certificate_misses(model, lim, q_row):
    """Pairs in range on which the restatement's own answer is not optimal: the gap of the two support
    functions along its normal (a lower bound on the distance) must meet the distance it reports."""
    from oracle import convex as cx
    from oracle import mjlite as mj
    d = o.Configuration(model, q_row).data
    misses = checked = 0
    for g1, g2 in lim.geom_id_pairs:
        if int(model.geom_type[g1]) == 0 or int(model.geom_type[g2]) == 0:
            continue
        fromto = np.zeros(6)
        dist, fromto = mj.mj_geomDistance(model, d, g1, g2, lim.collision_detection_distance)
        if not (0.0 < dist < lim.collision_detection_distance):
            continue
        sh = []
        for g in (g1, g2):
            s = dict(type=int(model.geom_type[g]), pos=np.asarray(d.geom_xpos[g], float),
                     mat=np.asarray(d.geom_xmat[g], float).reshape(3, 3), size=np.asarray(model.geom_size[g], float))
            if s["type"] == cx.MESH:
                k = int(model.geom_dataid[g])
                s["verts"] = model.mesh_vert[model.mesh_vertadr[k]:model.mesh_vertadr[k] + model.mesh_vertnum[k]]
            sh.append(s)
        n = (fromto[3:] - fromto[:3]) / np.linalg.norm(fromto[3:] - fromto[:3])
        gap = cx.support(sh[1], -n) @ n - cx.support(sh[0], n) @ n - cx.radius(sh[0]) - cx.radius(sh[1])
        checked += 1
        misses += int(abs(gap - dist) > 1e-7)
    return checked, misses


def sweep(name, model, lim, q, dt=0.01, tol=1e-6, certify=True):
    bad = []
    n_checked = n_miss = 0
    dm = DeviceModel(model, DEV)
    prob = DeviceProblem(dm, [], [spec_of(lim)])
    P, c, G, h = prob.build_qp(dev(q), None, dt, 1e-3)
    G, h = G.cpu().numpy(), h.cpu().numpy()
    t0 = time.time()
    n_rows = n_active = n_pen = pattern = 0
    dG = dh = 0.0
    worst = None
    for b in range(len(q)):
        Gr, hr = lim.inequalities(o.Configuration(model, q[b]), dt)
        if certify:
            ck, ms = certificate_misses(model, lim, q[b])
            n_checked, n_miss = n_checked + ck, n_miss + ms
        fin = np.isfinite(hr)
        both = fin & np.isfinite(h[b])
        pattern += int((fin != np.isfinite(h[b])).sum())
        n_rows += len(hr)
        n_active += int(fin.sum())
        n_pen += int((hr[fin] == lim.bound_relaxation).sum())
        if both.any():
            gg = np.abs(G[b] - Gr).max(axis=1)
            for r in np.nonzero(both & (gg > tol))[0]:
                to_d = lambda hv: lim.minimum_distance_from_collisions + (hv - lim.bound_relaxation) * dt / lim.gain
                bad.append((b, lim.geom_id_pairs[r], float(gg[r]), float(to_d(h[b][r])), float(to_d(hr[r])),
                            float(np.linalg.norm(G[b][r])), float(np.linalg.norm(Gr[r]))))
            g = np.abs(G[b][both] - Gr[both]).max(axis=1)
            k = int(g.argmax())
            if g[k] > dG:
                dG, worst = float(g[k]), (b, int(np.nonzero(both)[0][k]))
            dh = max(dh, float(np.abs(h[b][both] - hr[both]).max()))
        if b % 32 == 31:
            print(f"  {name}: {b + 1} configurations, {time.time() - t0:.0f} s", flush=True)
    pair = lim.geom_id_pairs[worst[1]] if worst else None
    print(f"{name}: {len(q)} configurations x {len(lim.geom_id_pairs)} pairs = {n_rows} rows, {n_active} in range, {n_pen} at the "
          f"relaxation bound; in-range pattern differs on {pattern} rows; max |dG| {dG:.2e} (configuration {worst[0] if worst else -1}, "
          f"geoms {pair}), max |dh| {dh:.2e}", flush=True)
    if certify:
        print(f"  separated pairs in range whose distance the restatement certifies (support-function gap along its normal "
              f"= distance, to 1e-7): {n_checked - n_miss} of {n_checked}")
    print(f"  rows with |dG| > {tol:g}: {len(bad)}")
    for r in sorted(bad, key=lambda t: -t[2])[:25]:
        print(f"    configuration {r[0]} geoms {r[1]}: |dG| {r[2]:.2e}, distance gpu {r[3]:+.6f} oracle {r[4]:+.6f}, |G| gpu {r[5]:.3f} oracle {r[6]:.3f}")


def main():
    rng = np.random.default_rng(31)
    model = load_builtin("tidybot_cube")
    coll = [g for g in range(model.ngeom) if model.geom_dataid[g] >= 0 and (model.geom_contype[g] or model.geom_conaffinity[g])]
    arm = [g for g in coll if model.geom_bodyid[g] in (6, 7, 8, 9, 10)]
    base = [g for g in coll if model.geom_bodyid[g] in (1, 2)]
    lim = o.CollisionAvoidanceLimit(model, [(arm, base + ["cube_geom", "floor"])], collision_detection_distance=0.25,
                                    minimum_distance_from_collisions=0.02)
    q = W.sample_q(model, rng, B)
    q[:, 4] = rng.uniform(0.8, 2.0, B)
    q[:, 6] = rng.uniform(-2.4, -1.2, B)
    n8 = B // 3
    q[:n8, 18:21] = q[:n8, 0:3] * [1, 1, 0] + rng.uniform(-0.3, 0.3, (n8, 3)) + [0.35, 0, 0.6]
    sweep("TidyBot mesh hulls", model, lim, q)
    model = load_mjcf(os.path.join(ROOT, "tests", "models", "convex_arm.xml"))
    lim = o.CollisionAvoidanceLimit(model, [(["upper_geom", "fore_geom", "hand_box", "hand_ball", "torso_geom"],
                                             ["post", "egg", "ground", "can_geom"])],
                                    collision_detection_distance=0.5, minimum_distance_from_collisions=0.01)
    q = np.tile(model.qpos0, (B, 1))
    q[:, :3] += rng.uniform(-0.25, 0.25, (B, 3))
    qq = rng.normal(size=(B, 4)) * 0.2 + [1, 0, 0, 0]
    q[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    q[:, 7:11] += rng.uniform(-0.6, 0.6, (B, 4))
    q[:, 11:14] += rng.uniform(-0.2, 0.2, (B, 3))
    qq = rng.normal(size=(B, 4))
    q[:, 14:18] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    sweep("cylinders / ellipsoids", model, lim, q, tol=1e-4)
    # closed-form pairs (plane / sphere / capsule / box) on the floating-base tree, the free object next to a geom
    model = load_mjcf(os.path.join(ROOT, "tests", "models", "float_arm.xml"))
    pairs = [("obj_geom", "shin_geom"), ("shin_geom", "bar_geom"), ("ground", "shin_geom"), ("obj_geom", "torso_geom"),
             ("obj_geom", "fore_geom"), ("leg_geom", "obj_geom"), ("ground", "fore_geom"), ("torso_geom", "fore_geom"),
             ("upper_geom", "bar_geom"), ("ground", "obj_geom"), ("shin_geom", "fore_geom"), ("bar_geom", "torso_geom"),
             ("bar_geom", "fore_geom")]
    lim = o.CollisionAvoidanceLimit(model, [([a], [b]) for a, b in pairs], collision_detection_distance=0.3,
                                    minimum_distance_from_collisions=0.01)
    q = np.tile(model.qpos0, (B, 1))
    for j in range(model.njnt):
        a, t = int(model.jnt_qposadr[j]), int(model.jnt_type[j])
        if t == 0:
            q[:, a:a + 3] += rng.uniform(-0.2, 0.2, (B, 3))
            w = rng.normal(size=(B, 4))
            q[:, a + 3:a + 7] = w / np.linalg.norm(w, axis=1, keepdims=True)
        elif t == 1:
            w = rng.normal(size=(B, 4))
            q[:, a:a + 4] = w / np.linalg.norm(w, axis=1, keepdims=True)
        else:
            lo, hi = (model.jnt_range[j] if model.jnt_limited[j] else (-1.0, 1.0))
            q[:, a] = rng.uniform(lo + 0.05, hi - 0.05, B)
    # the free object next to one of the robot's geoms in most configurations
    fa = int(model.jnt_qposadr[model.joint_id("obj_free")])
    names = ["shin_geom", "torso_geom", "fore_geom", "leg_geom"]
    for b in range(B):
        if b % 5 == 4:
            continue
        c = o.Configuration(model, q[b]).data.geom_xpos[model.geom_id(names[b % 4])]
        dvec = rng.normal(size=3)
        q[b, fa:fa + 3] = c + dvec / np.linalg.norm(dvec) * rng.uniform(0.0, 0.25)
    sweep("closed-form primitive pairs", model, lim, q, tol=1e-7, certify=False)


if __name__ == "__main__":
    main()
