"""CollisionAvoidanceLimit on geoms without a closed-form distance, on the GPU: TidyBot's
collision meshes (their convex hulls, as MuJoCo collides them:
interactive_scripts/stanford_tidybot/tidybot_cube.xml:112-116,125-167), cylinders and
ellipsoids.  The rows G, h of mink.build_ik (mink/limits/collision_avoidance_limit.py:187-229)
from the kernels against the numpy restatement (oracle/convex.py; parity with MuJoCo's libccd
answers is unpinned, see there), and a solve through them."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.mjcf import load_builtin, load_mjcf
from oracle import ik_oracle as o

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=DEV)


def spec_of(lim):
    return dict(type=L.LIMIT_COLLISION, index=np.array(lim.geom_id_pairs, dtype=np.int32).reshape(-1), gain=lim.gain,
                minimum_distance_from_collisions=lim.minimum_distance_from_collisions,
                collision_detection_distance=lim.collision_detection_distance, bound_relaxation=lim.bound_relaxation)


def compare_rows(model, lim, q, dt=0.01, atol=1e-8):
    from mink_amd.engine import DeviceModel, DeviceProblem
    dm = DeviceModel(model, DEV)
    prob = DeviceProblem(dm, [], [spec_of(lim)])
    P, c, G, h = prob.build_qp(dev(q), None, dt, 1e-3)
    G, h = G.cpu().numpy(), h.cpu().numpy()
    n_active = n_pen = 0
    for b in range(len(q)):
        Gr, hr = lim.inequalities(o.Configuration(model, q[b]), dt)
        assert np.array_equal(np.isfinite(h[b]), np.isfinite(hr)), b
        fin = np.isfinite(hr)
        np.testing.assert_allclose(h[b][fin], hr[fin], rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(G[b], Gr, rtol=0, atol=atol)
        n_active += int(fin.sum())
        n_pen += int((hr[fin] == lim.bound_relaxation).sum())
    return n_active, n_pen, (dm, prob)


def test_tidybot_mesh_pairs_rows_and_solve():
    model = load_builtin("tidybot_cube")
    coll = [g for g in range(model.ngeom) if model.geom_dataid[g] >= 0 and (model.geom_contype[g] or model.geom_conaffinity[g])]
    assert len(coll) == 22 and model.mesh_vert.shape[0] > 5000
    arm = [g for g in coll if model.geom_bodyid[g] in (6, 7, 8, 9, 10)]       # forearm ... gripper base
    base = [g for g in coll if model.geom_bodyid[g] in (1, 2)]                 # mobile base, arm base
    lim = o.CollisionAvoidanceLimit(model, [(arm, base + ["cube_geom", "floor"])], collision_detection_distance=0.25,
                                    minimum_distance_from_collisions=0.02)
    assert len(lim.geom_id_pairs) >= 30
    rng = np.random.default_rng(12)
    B = 24
    q = W.sample_q(model, rng, B)
    q[:, 4] = rng.uniform(0.8, 2.0, B)           # fold the arm towards the base
    q[:, 6] = rng.uniform(-2.4, -1.2, B)
    q[:8, 18:21] = q[:8, 0:3] * [1, 1, 0] + rng.uniform(-0.3, 0.3, (8, 3)) + [0.35, 0, 0.6]   # cube near the arm
    n_active, n_pen, (dm, _) = compare_rows(model, lim, q)
    print("mesh pairs: active rows", n_active, "of", B * len(lim.geom_id_pairs), "penetrating / inside d_min", n_pen)
    assert n_active > 100 and n_pen > 3
    # a solve with these rows against the oracle's solve_ik
    from mink_amd.engine import DeviceProblem
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True) + [spec_of(lim)])
    assert not prob.fused
    tg = dm.frame_pose(dev(W.sample_q(model, rng, B)), 2, model.site_id("pinch_site"))
    v, st = prob.solve(dev(q), tg, W.DT, W.DAMPING)
    tgn = tg.cpu().numpy()
    bad = 0
    for b in range(0, B, 3):
        tasks = [o.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0, target=tgn[b]),
                 o.PostureTask(cost=W.posture_cost(model), target_q=W.home_q(model))]
        limits = [o.ConfigurationLimit(model), o.VelocityLimit(model, W.VELOCITY_LIMITS), lim]
        try:
            ref = o.solve_ik(o.Configuration(model, q[b]), tasks, W.DT, damping=W.DAMPING, limits=limits)
        except AssertionError:
            assert int(st[b]) == 2               # infeasible rows (deep penetration): both say so
            bad += 1
            continue
        assert int(st[b]) == 0
        np.testing.assert_allclose(v[b].cpu().numpy(), ref, rtol=0, atol=1e-6)
    assert bad < B // 3


@pytest.mark.parametrize("case", ["FLAT_SIMPLEX", "TWO_STATE_CYCLE"])
def test_configurations_that_break_an_unguarded_gjk(case):
    """Found by the sweep of tools/soak_convex.py (tests/convex_cases.py says what goes wrong on each):
    the kernels' row against the restatement's, whose distance carries its optimality certificate."""
    import convex_cases as cc
    c = getattr(cc, case)
    model = cc.load_case_model(c)
    g1, g2 = c["geoms"]
    dist, gap = cc.distance_and_certificate(model, c["q"], g1, g2)
    assert abs(dist - c["distance"]) < 1e-9 and abs(gap - dist) < 1e-9
    lim = o.CollisionAvoidanceLimit(model, [([g1], [g2])], collision_detection_distance=0.5,
                                    minimum_distance_from_collisions=0.01)
    assert lim.geom_id_pairs == [(g1, g2)]
    n_active, _, _ = compare_rows(model, lim, np.array([c["q"]], dtype=np.float64), atol=1e-8 if case == "FLAT_SIMPLEX" else 2e-5)
    assert n_active == 1


def test_cylinders_and_ellipsoids():
    model = load_mjcf(os.path.join(HERE, "models", "convex_arm.xml"))
    lim = o.CollisionAvoidanceLimit(model, [(["upper_geom", "fore_geom", "hand_box", "hand_ball", "torso_geom"],
                                             ["post", "egg", "ground", "can_geom"])],
                                    collision_detection_distance=0.5, minimum_distance_from_collisions=0.01)
    assert len(lim.geom_id_pairs) == 17     # torso x world geoms fall to the parent-child filter
    rng = np.random.default_rng(13)
    B = 32
    q = np.tile(model.qpos0, (B, 1))
    q[:, :3] += rng.uniform(-0.25, 0.25, (B, 3))
    qq = rng.normal(size=(B, 4)) * 0.2 + [1, 0, 0, 0]
    q[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    q[:, 7:11] += rng.uniform(-0.6, 0.6, (B, 4))
    q[:, 11:14] += rng.uniform(-0.2, 0.2, (B, 3))
    qq = rng.normal(size=(B, 4))
    q[:, 14:18] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    # smooth shapes: the iteration stops at a relative tolerance on the distance (1e-13), not at
    # a vertex; the witness points, hence the rows, are then good to about its square root
    n_active, n_pen, _ = compare_rows(model, lim, q, atol=2e-5)
    print("cylinder / ellipsoid pairs: active rows", n_active, "penetrating / inside d_min", n_pen)
    assert n_active > 150


def test_mesh_without_hull_is_refused():
    from mink_amd.engine import DeviceModel, DeviceProblem
    model = load_builtin("tidybot_cube")
    visual = [g for g in range(model.ngeom) if model.geom_dataid[g] >= 0 and not model.geom_contype[g]
              and model.mesh_vertnum[model.geom_dataid[g]] == 0]
    assert visual
    dm = DeviceModel(model, DEV)
    spec = dict(type=L.LIMIT_COLLISION, index=np.array([visual[0], model.geom_id("cube_geom")], dtype=np.int32), gain=0.85,
                minimum_distance_from_collisions=0.005, collision_detection_distance=0.01, bound_relaxation=0.0)
    with pytest.raises(L.MinkAmdError) as e:
        DeviceProblem(dm, [], [spec])
    assert e.value.code == L.MK_ERR_UNSUPPORTED and "hull" in str(e.value)
