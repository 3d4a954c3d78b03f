"""The lane-per-problem solve with collision rows (middle stage of the staged solve), run on the
CPU through the host build of the device functions, against the numpy restatement of
mink.solve_ik with a CollisionAvoidanceLimit (mink/limits/collision_avoidance_limit.py:187-210,
mink/solve_ik.py:68-105)."""
import numpy as np
import pytest

import hostsim_lib as hs
from mink_amd import workloads as W
from mink_amd.mjcf import load_builtin
from oracle import ik_oracle as o

pytestmark = pytest.mark.skipif(not hs.available(), reason="tests/hostsim/libhostsim.so not built")


def near_inputs(model, rng, n):
    """Configurations with the cube a few centimetres from a gripper pad (the recipe of the
    benchmark's config 4, with the numpy kinematics) and targets from an independent sample."""
    q = W.sample_q(model, rng, n)
    qs = W.sample_q(model, rng, n)
    adr = int(model.jnt_qposadr[model.joint_id("interactive_obj_freejoint")])
    tg = np.zeros((n, 14))
    for i in range(n):
        cfg = o.Configuration(model, q[i])
        pad = cfg.get_transform_frame_to_world(W.PADS[rng.integers(0, 4)], "geom")
        d = rng.normal(size=3)
        q[i, adr:adr + 3] = pad[4:] + d / np.linalg.norm(d) * rng.uniform(0.03, 0.05)
        qq = rng.normal(size=4)
        q[i, adr + 3:adr + 7] = qq / np.linalg.norm(qq)
        c2 = o.Configuration(model, qs[i])
        t_world = c2.get_transform_frame_to_world("pinch_site", "site")
        t_base = c2.get_transform_frame_to_world("base_link", "body")
        tg[i, :7] = t_world
        tg[i, 7:] = o.se3_multiply(o.se3_inverse(t_base), t_world)
    return q, tg


def oracle_solve(model, q, tg, with_collision=True):
    tasks = [o.FrameTask("pinch_site", "site", 1.0, 1.0, 1.0, 1.0, tg[:7]),
             o.RelativeFrameTask("pinch_site", "site", "base_link", "body", 1.0, 1.0, 1.0, 0.0, tg[7:]),
             o.PostureTask(W.posture_cost(model), W.home_q(model))]
    limits = [o.ConfigurationLimit(model, gain=0.95),
              o.VelocityLimit(model, {k: np.atleast_1d(v) for k, v in W.VELOCITY_LIMITS.items()})]
    coll = o.CollisionAvoidanceLimit(model, [(list(W.PADS), ["cube_geom", "floor"])])
    if with_collision:
        limits.append(coll)
    cfg = o.Configuration(model, q)
    v = o.solve_ik(cfg, tasks, W.DT, damping=W.DAMPING, limits=limits)
    G, h = coll.inequalities(cfg, W.DT)
    return v, G, h


def test_lane_rows_against_oracle():
    model = load_builtin("tidybot_cube")
    rng = np.random.default_rng(3)
    n = 48
    q, tg = near_inputs(model, rng, n)
    tasks, limits = W.config4_specs(model)
    v, good, nrows = hs.solve_rows(model, tasks, limits, q, tg, W.DT, W.DAMPING)
    n_binding = 0
    for i in range(n):
        vr, G, h = oracle_solve(model, q[i], tg[i])
        assert nrows[i] == int(np.isfinite(h).sum()), (i, nrows[i], h)
        if not good[i]:
            assert nrows[i] > 4, f"problem {i} with {nrows[i]} rows was handed on"
            continue
        np.testing.assert_allclose(v[i], vr, rtol=0, atol=2e-9, err_msg=f"problem {i}, rows {nrows[i]}")
        v0, _, _ = oracle_solve(model, q[i], tg[i], with_collision=False)
        n_binding += int(np.abs(v0 - vr).max() > 1e-7)
    assert good.sum() >= n - 2
    assert (nrows > 0).sum() >= n // 2 and n_binding >= 5, (nrows, n_binding)


def test_lane_rows_with_geoms_on_inner_links():
    """Capsules on links inside the task chain of a synthetic arm against a loose box, a pillar and
    the floor: the rows' columns stop at the link's joint."""
    import os
    from mink_amd import _lib as L
    from mink_amd.limits import CollisionAvoidanceLimit
    from mink_amd.mjcf import load_mjcf
    model = load_mjcf(os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "grip_arm.xml"))
    groups = [(["l2_geom", "l3_geom"], ["brick_geom", "pillar", "floor"]), (["l1_geom"], ["brick_geom", "pad_a"])]
    vel = {"lift": 0.4, "j1": 1.5, "j2": 1.5, "j3": 2.0, "j4": 2.5, "fa": 0.05, "fb": 1.0}
    cost = np.where(np.arange(model.nv) < 7, 0.05, 0.0)
    site = model.site_id("tool")
    tasks = [dict(type=L.TASK_FRAME, frame_type=L.FRAME_TYPES["site"], frame_id=site, cost=np.ones(6), gain=1.0, lm_damping=0.5),
             dict(type=L.TASK_RELATIVE_FRAME, frame_type=L.FRAME_TYPES["site"], frame_id=site, root_type=L.FRAME_TYPES["body"],
                  root_id=model.body_id("rail"), cost=np.array([0.5, 0.5, 0.5, 0.2, 0.2, 0.2]), gain=0.7, lm_damping=0.0),
             dict(type=L.TASK_POSTURE, cost=cost, target_q=model.qpos0.copy(), gain=1.0, lm_damping=0.0)]
    limits = [dict(type=L.LIMIT_CONFIGURATION, gain=0.9, min_distance_from_limits=0.0),
              dict(type=L.LIMIT_VELOCITY, index=[int(model.jnt_dofadr[model.joint_id(n)]) for n in vel], vmax=list(vel.values())),
              CollisionAvoidanceLimit(model, groups, gain=0.7, minimum_distance_from_collisions=0.004,
                                      collision_detection_distance=0.02, bound_relaxation=1e-3)._spec(model)]
    rng = np.random.default_rng(12)
    n = 40
    q = np.tile(model.qpos0, (n, 1))
    tg = np.zeros((n, 14))
    adr = int(model.jnt_qposadr[model.joint_id("brick_free")])
    links = ("l1_geom", "l2_geom", "l3_geom")
    for i in range(n):
        for j in range(7):
            lo, hi = model.jnt_range[j]
            q[i, j] = rng.uniform(lo + 0.1 * (hi - lo), hi - 0.1 * (hi - lo))
        cfg = o.Configuration(model, q[i])
        at = cfg.get_transform_frame_to_world(links[rng.integers(0, 3)], "geom")
        d = rng.normal(size=3)
        q[i, adr:adr + 3] = at[4:] + d / np.linalg.norm(d) * rng.uniform(0.05, 0.09)
        qq = rng.normal(size=4)
        q[i, adr + 3:adr + 7] = qq / np.linalg.norm(qq)
        qs = q[i].copy()
        qs[:7] += rng.uniform(-0.1, 0.1, 7)
        c2 = o.Configuration(model, qs)
        t_world = c2.get_transform_frame_to_world("tool", "site")
        tg[i, :7] = t_world
        tg[i, 7:] = o.se3_multiply(o.se3_inverse(c2.get_transform_frame_to_world("rail", "body")), t_world)
    v, good, nrows = hs.solve_rows(model, tasks, limits, q, tg, 0.01, 1e-3)
    n_binding = 0
    for i in range(n):
        otasks = [o.FrameTask("tool", "site", 1.0, 1.0, 1.0, 0.5, tg[i, :7]),
                  o.RelativeFrameTask("tool", "site", "rail", "body", 0.5, 0.2, 0.7, 0.0, tg[i, 7:]),
                  o.PostureTask(cost, model.qpos0.copy())]
        coll = o.CollisionAvoidanceLimit(model, groups, 0.7, 0.004, 0.02, 1e-3)
        base = [o.ConfigurationLimit(model, gain=0.9), o.VelocityLimit(model, {k: np.atleast_1d(x) for k, x in vel.items()})]
        cfg = o.Configuration(model, q[i])
        vr = o.solve_ik(cfg, otasks, 0.01, damping=1e-3, limits=base + [coll])
        h = coll.inequalities(cfg, 0.01)[1]
        assert nrows[i] == int(np.isfinite(h).sum()), (i, nrows[i], h)
        if not good[i]:
            continue
        np.testing.assert_allclose(v[i], vr, rtol=0, atol=2e-9, err_msg=f"problem {i}, rows {nrows[i]}")
        v0 = o.solve_ik(cfg, otasks, 0.01, damping=1e-3, limits=base)
        n_binding += int(np.abs(v0 - vr).max() > 1e-7)
    assert good.sum() >= n - 2
    assert (nrows > 0).sum() >= n // 3 and n_binding >= 5, (nrows, n_binding)


def test_box_box_search_against_exhaustive_enumeration():
    """The device searches the closest points of two separated boxes over the vertex-box candidates
    and one edge pair (the support edges of the widest cross-product axis); the numpy restatement
    enumerates all 8 + 8 + 144 candidates.  Same distance and witness points on boxes in general
    position, at every separation, and same signed distance for overlapping ones."""
    from oracle import mjlite

    rng = np.random.default_rng(5)
    n = 4000
    q1 = rng.normal(size=(n, 4)); q1 /= np.linalg.norm(q1, axis=1, keepdims=True)
    q2 = rng.normal(size=(n, 4)); q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    h1 = rng.uniform(0.01, 0.1, (n, 3))
    h2 = rng.uniform(0.01, 0.1, (n, 3))
    c1 = rng.uniform(-0.05, 0.05, (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    c2 = c1 + d * rng.uniform(0.0, 0.35, (n, 1))
    inr, dist, p1, p2 = hs.box_box(c1, q1, h1, c2, q2, h2, 1.0)
    n_edge = n_pen = 0
    for i in range(n):
        r1, r2 = mjlite.mju_quat2Mat(q1[i]).reshape(3, 3), mjlite.mju_quat2Mat(q2[i]).reshape(3, 3)
        dr, u, w = mjlite._closest_box_box(c1[i], r1, h1[i], c2[i], r2, h2[i])
        assert inr[i]
        assert abs(dist[i] - dr) <= 1e-13, (i, dist[i], dr)
        np.testing.assert_allclose(p1[i], u, rtol=0, atol=1e-9)
        np.testing.assert_allclose(p2[i], w, rtol=0, atol=1e-9)
        if dr < 0:
            n_pen += 1
        else:
            # an interior edge-edge pair: no witness is a vertex of its box
            v1 = np.abs(np.abs(r1.T @ (u - c1[i])) - h1[i]) < 1e-12
            v2 = np.abs(np.abs(r2.T @ (w - c2[i])) - h2[i]) < 1e-12
            n_edge += int(v1.sum() == 2 and v2.sum() == 2)
    assert n_edge > 200 and n_pen > 200, (n_edge, n_pen)


@pytest.mark.parametrize("t1,t2", [(0, 2), (0, 3), (0, 6), (2, 2), (2, 3), (2, 6), (3, 3), (3, 6), (6, 6)])
def test_lane_distance_of_every_primitive_pair(t1, t2):
    """geom_distance_lane (the closed forms both kernels share) against the numpy restatement of
    mj_geomDistance for planes (0), spheres (2), capsules (3) and boxes (6), separated and
    penetrating -- including capsule axes that pierce a thin box through two opposite faces, where
    the exit side is the capsule centre's."""
    from types import SimpleNamespace
    from oracle import mjlite

    rng = np.random.default_rng(100 * t1 + t2)
    n = 600
    def quat(n):
        q = rng.normal(size=(n, 4))
        return q / np.linalg.norm(q, axis=1, keepdims=True)
    def size(t):
        s = rng.uniform(0.01, 0.08, (n, 3))
        if t == 6:
            s[: n // 3, 0] = rng.uniform(0.003, 0.008, n // 3)     # thin plates
        return s
    q1, q2, s1, s2 = quat(n), quat(n), size(t1), size(t2)
    c1 = rng.uniform(-0.05, 0.05, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c2 = c1 + d * rng.uniform(0.0, 0.2, (n, 1))
    inr, dist, p1, p2 = hs.geom_distance(t1, c1, q1, s1, t2, c2, q2, s2, 0.05)
    n_in = n_pen = 0
    for i in range(n):
        m = SimpleNamespace(geom_type=np.array([t1, t2]), geom_size=np.stack([s1[i], s2[i]]), geom_dataid=np.array([-1, -1]))
        dd = SimpleNamespace(geom_xpos=np.stack([c1[i], c2[i]]),
                             geom_xmat=np.stack([mjlite.mju_quat2Mat(q1[i]), mjlite.mju_quat2Mat(q2[i])]))
        dr, fromto = mjlite.mj_geomDistance(m, dd, 0, 1, 0.05)
        if dr == 0.05:
            assert not inr[i], (i, dist[i])
            continue
        assert inr[i]
        n_in += 1
        n_pen += dr < 0
        assert abs(dist[i] - dr) <= 1e-12, (i, dist[i], dr)
        np.testing.assert_allclose(p1[i], fromto[:3], rtol=0, atol=1e-9, err_msg=f"pair {i}")
        np.testing.assert_allclose(p2[i], fromto[3:], rtol=0, atol=1e-9, err_msg=f"pair {i}")
    assert n_in > 100 and n_pen > 30, (n_in, n_pen)
