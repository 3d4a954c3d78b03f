"""CPU checks of the support-mapping distance restatement (oracle/convex.py) that the GPU
route is compared with: against brute force on hull vertices, against the closed forms where
both apply, and -- through the collision limit's row -- against a finite difference of the
distance along a joint motion (the reference's tests/test_collision_avoidance_limit.py:65-111
compares the same row with MuJoCo's efc_J, which is not available here)."""
import os

import numpy as np
import pytest

from mink_amd.mjcf import load_builtin, load_mjcf
from oracle import convex as cx
from oracle import ik_oracle as o
from oracle import mjlite as mj

HERE = os.path.dirname(os.path.abspath(__file__))


def rand_rot(rng):
    q = rng.normal(size=4)
    return mj.mju_quat2Mat(q / np.linalg.norm(q)).reshape(3, 3)


def test_gjk_matches_brute_force_on_small_hulls():
    from scipy.optimize import minimize
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(12):
        shapes = []
        for off in (0.0, 0.4):
            pts = rng.normal(size=(20, 3)) * 0.08
            shapes.append(dict(type=cx.MESH, pos=rng.normal(size=3) * 0.05 + np.array([off, 0, 0]), mat=rand_rot(rng),
                               size=np.zeros(3), verts=pts[ConvexHull(pts).vertices]))
        A, B = shapes
        d, pa, pb = cx.distance(A, B)
        VA, VB = A["pos"] + A["verts"] @ A["mat"].T, B["pos"] + B["verts"] @ B["mat"].T
        na = len(VA)

        def f(x):
            r = x[:na] @ VA - x[na:] @ VB
            return r @ r

        cons = [{"type": "eq", "fun": lambda x: x[:na].sum() - 1}, {"type": "eq", "fun": lambda x: x[na:].sum() - 1}]
        best = min(minimize(f, np.concatenate([rng.dirichlet(np.ones(na)), rng.dirichlet(np.ones(len(VB)))]),
                            bounds=[(0, 1)] * (na + len(VB)), constraints=cons, method="SLSQP",
                            options={"ftol": 1e-16, "maxiter": 400}).fun for _ in range(2))
        assert d > 0
        worst = max(worst, abs(d - np.sqrt(best)))
        assert abs(np.linalg.norm(pb - pa) - d) < 1e-13
    assert worst < 1e-7        # SLSQP's own accuracy; GJK is exact on polytopes


def test_support_route_agrees_with_closed_forms():
    rng = np.random.default_rng(1)
    for _ in range(100):
        c1, c2 = rng.normal(size=3) * 0.05, rng.normal(size=3) * 0.05 + np.array([0.15, 0, 0])
        r1, r2 = rand_rot(rng), rand_rot(rng)
        h1, h2 = rng.uniform(0.01, 0.04, 3), rng.uniform(0.01, 0.04, 3)
        d, pa, pb = cx.distance(dict(type=cx.BOX, pos=c1, mat=r1, size=h1), dict(type=cx.BOX, pos=c2, mat=r2, size=h2))
        dref, p1, p2 = mj._closest_box_box(c1, r1, h1, c2, r2, h2)
        if dref > 0:
            assert abs(d - dref) < 1e-12
        # sphere / capsule cores: exact radii
        s = dict(type=cx.SPHERE, pos=c1, mat=r1, size=np.array([0.03, 0, 0]))
        cap = dict(type=cx.CAPSULE, pos=c2, mat=r2, size=np.array([0.02, 0.05, 0]))
        d, pa, pb = cx.distance(s, cap)
        t = float(np.clip((c1 - c2) @ r2[:, 2], -0.05, 0.05))
        assert abs(d - (np.linalg.norm(c1 - (c2 + t * r2[:, 2])) - 0.05)) < 1e-12
    # overlapping axis-aligned boxes: depth along the axis of least overlap
    A = dict(type=cx.BOX, pos=np.zeros(3), mat=np.eye(3), size=np.array([0.1, 0.2, 0.3]))
    B = dict(type=cx.BOX, pos=np.array([0.15, 0.01, 0.02]), mat=np.eye(3), size=np.array([0.1, 0.1, 0.1]))
    d, p1, p2 = cx.distance(A, B)
    assert abs(d + 0.05) < 1e-12 and abs((p1 - p2)[0] - 0.05) < 1e-12
    # ellipsoid against a plane: support point
    E = dict(type=cx.ELLIPSOID, pos=np.array([0, 0, 0.5]), mat=np.eye(3), size=np.array([0.1, 0.2, 0.3]))
    d, p1, p2 = cx.plane_distance(dict(pos=np.zeros(3), mat=np.eye(3)), E)
    assert abs(d - 0.2) < 1e-15


@pytest.mark.parametrize("which", ["meshes", "cylinders"])
def test_collision_row_is_the_distance_gradient(which):
    """G row = -d(dist)/dq (mink/limits/collision_avoidance_limit.py:59-72, 199-209)."""
    rng = np.random.default_rng(3)
    if which == "meshes":
        model = load_builtin("tidybot_cube")
        coll = [g for g in range(model.ngeom) if model.geom_dataid[g] >= 0 and model.geom_contype[g]]
        arm = [g for g in coll if model.geom_bodyid[g] in (6, 7, 8, 9, 10)]
        base = [g for g in coll if model.geom_bodyid[g] == 1]
        lim = o.CollisionAvoidanceLimit(model, [(arm, base + ["cube_geom", "floor"])], collision_detection_distance=0.6)
        q0 = model.key_qpos[0].copy()
    else:
        model = load_mjcf(os.path.join(HERE, "models", "convex_arm.xml"))
        lim = o.CollisionAvoidanceLimit(model, [(["upper_geom", "fore_geom", "hand_box", "hand_ball", "torso_geom"],
                                                 ["post", "egg", "ground", "can_geom"])], collision_detection_distance=1.0)
        q0 = model.qpos0.copy()
    checked = 0
    for _ in range(4):
        q = q0.copy()
        for j in range(model.njnt):
            if model.jnt_type[j] >= 2:
                q[model.jnt_qposadr[j]] += rng.uniform(-0.4, 0.4)
        cfg = o.Configuration(model, q)
        G, h = lim.inequalities(cfg, 0.01)
        v = rng.normal(size=model.nv)
        eps = 1e-6
        qp, qm = q.copy(), q.copy()
        mj.mj_integratePos(model, qp, v, eps)
        mj.mj_integratePos(model, qm, v, -eps)
        dp, dm_ = o.Configuration(model, qp).data, o.Configuration(model, qm).data
        for k, (g1, g2) in enumerate(lim.geom_id_pairs):
            if not np.isfinite(h[k]):
                continue
            d0, _ = mj.mj_geomDistance(model, cfg.data, g1, g2, 10.0)
            if d0 <= 1e-4:
                continue                      # penetration: the row uses the portal's direction
            d1, _ = mj.mj_geomDistance(model, dp, g1, g2, 10.0)
            d2, _ = mj.mj_geomDistance(model, dm_, g1, g2, 10.0)
            fd = (d1 - d2) / (2 * eps)
            assert abs(fd + G[k] @ v) < 2e-5 * max(1.0, np.abs(v).max()), (k, g1, g2, fd, -G[k] @ v)
            checked += 1
    assert checked >= 8


@pytest.mark.parametrize("case", ["FLAT_SIMPLEX", "TWO_STATE_CYCLE"])
def test_restatement_on_configurations_that_break_an_unguarded_gjk(case):
    """The numpy twin on the two configurations of tests/convex_cases.py: the distance it reports is
    optimal (support-function gap along its normal = distance)."""
    import convex_cases as cc
    c = getattr(cc, case)
    model = cc.load_case_model(c)
    dist, gap = cc.distance_and_certificate(model, c["q"], *c["geoms"])
    assert abs(dist - c["distance"]) < 1e-9
    assert abs(gap - dist) < 1e-9
