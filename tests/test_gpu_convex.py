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


# ---- certificates on the KERNEL's own answers (no twin implementation involved) -----------------
# The distance kernels and oracle/convex.py share their algorithm, so agreement between them does
# not exclude a fault common to both.  The two checks below use nothing but the kernel's rows and
# elementary geometry written here:
#  (1) support-function gap.  For a pair with one geom on a floating body, the row's columns on
#      that body's three translational dofs are -+n, the kernel's contact normal.  For ANY unit n,
#      gap(n) = min over B of n.b - max over A of n.a is a lower bound on the distance of two convex
#      sets, and it equals the distance only for the optimal n; the function is concave in n, so a
#      local maximum is the global one.  The kernel's distance (recovered from its h) must equal
#      gap(n) at its own n, and gap must not increase for directions around n.
#  (2) row = -grad(distance): finite differences of the kernel's own h along random joint motions.
def _support_value(model, data, g, d):
    """max over the geom g of d . x (world frame), from the geom's definition."""
    t = int(model.geom_type[g])
    pos, R, sz = np.asarray(data.geom_xpos[g], float), np.asarray(data.geom_xmat[g], float).reshape(3, 3), model.geom_size[g]
    dl = R.T @ d
    if t == 2:      # sphere
        ext = sz[0]
    elif t == 3:    # capsule: segment along z plus radius
        ext = sz[1] * abs(dl[2]) + sz[0]
    elif t == 4:    # ellipsoid
        ext = float(np.linalg.norm(sz * dl))
    elif t == 5:    # cylinder: disc radius in the xy plane plus half height
        ext = sz[0] * float(np.hypot(dl[0], dl[1])) + sz[1] * abs(dl[2])
    elif t == 6:    # box
        ext = float(np.abs(dl) @ sz)
    elif t == 7:    # mesh: its hull vertices
        k = int(model.geom_dataid[g])
        ext = float((model.mesh_vert[model.mesh_vertadr[k]:model.mesh_vertadr[k] + model.mesh_vertnum[k]] @ dl).max())
    else:
        raise ValueError(t)
    return float(pos @ d) + ext


def _gap(model, data, g1, g2, n):
    """Separation of the two geoms along the unit direction n (from g1 towards g2); a plane g1
    is the half space below its surface."""
    if int(model.geom_type[g1]) == 0:
        R = np.asarray(data.geom_xmat[g1], float).reshape(3, 3)
        if not np.allclose(R[:, 2], n, atol=1e-9):
            return -np.inf              # a half space is bounded in the direction of its normal only
        top = float(np.asarray(data.geom_xpos[g1], float) @ n)
    else:
        top = _support_value(model, data, g1, n)
    return -_support_value(model, data, g2, -n) - top


def certify_rows(model, lim, q, float_dofs, dt=0.01, tol=1e-7, smooth_tol=None):
    """float_dofs: geom id -> first translational dof of the free joint that carries it (geoms not
    in the map do not move with those dofs).  Returns the number of rows certified."""
    from mink_amd.engine import DeviceModel, DeviceProblem
    dm = DeviceModel(model, DEV)
    prob = DeviceProblem(dm, [], [spec_of(lim)])
    P, c, G, h = prob.build_qp(dev(q), None, dt, 1e-3)
    G, h = G.cpu().numpy(), h.cpu().numpy()
    rng = np.random.default_rng(5)
    # (2) rows against finite differences of the kernel's own distances
    eps = 1e-6
    vdir = rng.normal(size=(len(q), model.nv))
    qp, qm = dev(q), dev(q)
    dm.integrate_(qp, dev(vdir), eps)
    dm.integrate_(qm, dev(vdir), -eps)
    hp = prob.build_qp(qp, None, dt, 1e-3)[3].cpu().numpy()
    hm = prob.build_qp(qm, None, dt, 1e-3)[3].cpu().numpy()
    n_cert = n_fd = 0
    for b in range(len(q)):
        data = o.Configuration(model, q[b]).data
        for k, (g1, g2) in enumerate(lim.geom_id_pairs):
            if not np.isfinite(h[b, k]) or h[b, k] == lim.bound_relaxation:
                continue                # beyond the detection distance, or inside d_min (h carries no distance)
            dist = (h[b, k] - lim.bound_relaxation) * dt / lim.gain + lim.minimum_distance_from_collisions
            if np.isfinite(hp[b, k]) and np.isfinite(hm[b, k]) and hp[b, k] != lim.bound_relaxation and hm[b, k] != lim.bound_relaxation:
                ddist = (hp[b, k] - hm[b, k]) * dt / lim.gain / (2 * eps)
                assert abs(ddist + G[b, k] @ vdir[b]) <= (smooth_tol or 1e-5) * max(1.0, np.abs(vdir[b]).max()), (b, k)
                n_fd += 1
            # (1) the kernel's normal from the floating body's columns
            f1, f2 = float_dofs.get(g1), float_dofs.get(g2)
            if f1 == f2:
                continue                # both geoms ride on the same floating base (or neither floats): no n in the row
            n = G[b, k, f1:f1 + 3] if f1 is not None else -G[b, k, f2:f2 + 3]
            assert abs(np.linalg.norm(n) - 1.0) <= 1e-9, (b, k, n)
            if dist <= 0.0:
                continue                # overlapping sets have no separating direction
            gap = _gap(model, data, g1, g2, n)
            assert abs(gap - dist) <= tol, (b, k, gap, dist)
            for _ in range(12):         # no direction near n separates the sets further
                m = n + rng.normal(size=3) * rng.choice([1e-2, 1e-3])
                m /= np.linalg.norm(m)
                assert _gap(model, data, g1, g2, m) <= gap + 1e-9, (b, k)
            n_cert += 1
    return n_cert, n_fd


def test_kernel_rows_carry_their_own_certificate_smooth_shapes():
    """Cylinders, ellipsoids, capsules, boxes and spheres on the floating arm of convex_arm.xml
    against world geoms and a free can."""
    model = load_mjcf(os.path.join(HERE, "models", "convex_arm.xml"))
    arm = ["upper_geom", "fore_geom", "hand_box", "hand_ball", "torso_geom"]
    lim = o.CollisionAvoidanceLimit(model, [(arm, ["post", "egg", "ground", "can_geom"])],
                                    collision_detection_distance=0.5, minimum_distance_from_collisions=0.01)
    rng = np.random.default_rng(14)
    B = 48
    q = np.tile(model.qpos0, (B, 1))
    q[:, :3] += rng.uniform(-0.25, 0.25, (B, 3))
    qq = rng.normal(size=(B, 4)) * 0.2 + [1, 0, 0, 0]
    q[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    q[:, 7:11] += rng.uniform(-0.6, 0.6, (B, 4))
    q[:, 11:14] += rng.uniform(-0.2, 0.2, (B, 3))
    qq = rng.normal(size=(B, 4))
    q[:, 14:18] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    root = int(model.jnt_dofadr[model.body_jntadr[model.geom_bodyid[model.geom_id("torso_geom")]]])
    can = int(model.jnt_dofadr[model.body_jntadr[model.geom_bodyid[model.geom_id("can_geom")]]])
    assert int(model.jnt_type[model.body_jntadr[model.geom_bodyid[model.geom_id("can_geom")]]]) == 0
    float_dofs = {model.geom_id(n): root for n in arm}
    float_dofs[model.geom_id("can_geom")] = can
    # smooth shapes stop at a relative tolerance on the distance; the normal is then good to about
    # its square root, and the gap, stationary in n at the optimum, to the tolerance itself
    n_cert, n_fd = certify_rows(model, lim, q, float_dofs, tol=1e-7, smooth_tol=1e-4)
    print("certified rows:", n_cert, " finite-difference rows:", n_fd)
    assert n_cert > 150 and n_fd > 150


def test_kernel_rows_carry_their_own_certificate_tidybot_meshes():
    """TidyBot's arm hulls against the free cube: the cube's free joint gives the normal."""
    model = load_builtin("tidybot_cube")
    coll = [g for g in range(model.ngeom) if model.geom_dataid[g] >= 0 and (model.geom_contype[g] or model.geom_conaffinity[g])]
    arm = [g for g in coll if model.geom_bodyid[g] in (5, 6, 7, 8, 9, 10)]
    lim = o.CollisionAvoidanceLimit(model, [(arm, ["cube_geom"])], collision_detection_distance=0.3,
                                    minimum_distance_from_collisions=0.01)
    rng = np.random.default_rng(15)
    B = 48
    q = W.sample_q(model, rng, B)
    site = np.array([o.Configuration(model, x).get_transform_frame_to_world("pinch_site", "site")[4:] for x in q])
    q[:, 18:21] = site + rng.uniform(-0.2, 0.2, (B, 3))          # the cube somewhere around the gripper
    qq = rng.normal(size=(B, 4))
    q[:, 21:25] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    cube = model.geom_id("cube_geom")
    n_cert, n_fd = certify_rows(model, lim, q, {cube: 18}, tol=1e-8, smooth_tol=1e-5)
    print("certified mesh rows:", n_cert, " finite-difference rows:", n_fd)
    assert n_cert > 60 and n_fd > 60
