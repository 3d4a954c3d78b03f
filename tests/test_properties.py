"""Reference-free properties of the parts whose parity cannot be pinned by outputs of the
reference (the MuJoCo helpers and the QP backend are not available offline): finite
differences of the kinematics, integrate / differentiate round trips, distance gradients,
KKT conditions and exhaustive active-set enumeration of the QP.  They run on the oracle
(CPU) and, marked gpu, on the kernels for a floating-base tree with a ball joint."""
import itertools
import os

import numpy as np
import pytest

from mink_amd import _lib as L
from mink_amd.mjcf import load_builtin, load_mjcf
from oracle import ik_oracle as o
from oracle import mjlite as mj

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fmodel():
    return load_mjcf(os.path.join(HERE, "models", "float_arm.xml"))


def random_q(model, rng):
    q = model.qpos0.copy()
    for j in range(model.njnt):
        a, t = model.jnt_qposadr[j], model.jnt_type[j]
        if t == 0:
            q[a:a + 3] += rng.uniform(-0.3, 0.3, 3)
            qq = rng.normal(size=4)
            q[a + 3:a + 7] = qq / np.linalg.norm(qq)
        elif t == 1:
            qq = rng.normal(size=4)
            q[a:a + 4] = qq / np.linalg.norm(qq)
        elif model.jnt_limited[j]:
            q[a] = rng.uniform(*model.jnt_range[j])
        else:
            q[a] = rng.uniform(-1, 1)
    return q


FRAMES = [("site", "tool"), ("body", "fore"), ("geom", "leg_geom"), ("body", "ball_obj")]


@pytest.fixture(scope="module")
def biped():
    """Humanoid-shaped tree (free joint, two six-hinge legs, two arms, fromto capsules)."""
    return load_mjcf(os.path.join(HERE, "models", "biped.xml"))


BIPED_FRAMES = [("site", "l_foot_site"), ("site", "r_hand"), ("geom", "l_fist"), ("body", "r_shin")]


@pytest.mark.parametrize("ftype,fname", BIPED_FRAMES)
def test_biped_frame_jacobian_is_the_derivative_of_the_pose(biped, ftype, fname):
    """The same first-order property on the multi-branch floating-base tree."""
    rng = np.random.default_rng(10)
    eps = 1e-6
    for _ in range(2):
        q = random_q(biped, rng)
        cfg = o.Configuration(biped, q)
        J = cfg.get_frame_jacobian(fname, ftype)
        T0 = cfg.get_transform_frame_to_world(fname, ftype)
        for i in range(biped.nv):
            dv = np.zeros(biped.nv)
            dv[i] = 1.0
            qp, qm = q.copy(), q.copy()
            mj.mj_integratePos(biped, qp, dv, eps)
            mj.mj_integratePos(biped, qm, dv, -eps)
            Tp = o.Configuration(biped, qp).get_transform_frame_to_world(fname, ftype)
            Tm = o.Configuration(biped, qm).get_transform_frame_to_world(fname, ftype)
            fd = (o.se3_log(o.se3_multiply(o.se3_inverse(T0), Tp)) - o.se3_log(o.se3_multiply(o.se3_inverse(T0), Tm))) / (2 * eps)
            np.testing.assert_allclose(J[:, i], fd, rtol=0, atol=2e-8)
    # a dof of the other leg does not move this frame
    if fname == "l_foot_site":
        r = int(biped.jnt_dofadr[biped.joint_id("r_knee")])
        assert np.all(J[:, r] == 0.0)


def test_biped_com_jacobian_is_the_derivative_of_the_com(biped):
    rng = np.random.default_rng(12)
    eps = 1e-6
    task = o.ComTask(cost=np.ones(3), target_com=np.zeros(3))
    q = random_q(biped, rng)
    J = task.jacobian(o.Configuration(biped, q))
    for i in range(biped.nv):
        dv = np.zeros(biped.nv)
        dv[i] = 1.0
        qp, qm = q.copy(), q.copy()
        mj.mj_integratePos(biped, qp, dv, eps)
        mj.mj_integratePos(biped, qm, dv, -eps)
        fd = (task.error(o.Configuration(biped, qp)) - task.error(o.Configuration(biped, qm))) / (2 * eps)
        np.testing.assert_allclose(J[:, i], fd, rtol=0, atol=1e-8)


@pytest.mark.parametrize("ftype,fname", FRAMES)
def test_frame_jacobian_is_the_derivative_of_the_pose(fmodel, ftype, fname):
    """T(q (+) eps e_i) = T(q) exp(eps J_i) to first order (tests/test_jacobians.py of the
    reference checks the same property against MuJoCo)."""
    rng = np.random.default_rng(0)
    eps = 1e-6
    for _ in range(5):
        q = random_q(fmodel, rng)
        cfg = o.Configuration(fmodel, q)
        J = cfg.get_frame_jacobian(fname, ftype)
        T0 = cfg.get_transform_frame_to_world(fname, ftype)
        for i in range(fmodel.nv):
            dv = np.zeros(fmodel.nv)
            dv[i] = 1.0
            qp, qm = q.copy(), q.copy()
            mj.mj_integratePos(fmodel, qp, dv, eps)
            mj.mj_integratePos(fmodel, qm, dv, -eps)
            Tp = o.Configuration(fmodel, qp).get_transform_frame_to_world(fname, ftype)
            Tm = o.Configuration(fmodel, qm).get_transform_frame_to_world(fname, ftype)
            fd = (o.se3_log(o.se3_multiply(o.se3_inverse(T0), Tp)) - o.se3_log(o.se3_multiply(o.se3_inverse(T0), Tm))) / (2 * eps)
            np.testing.assert_allclose(J[:, i], fd, rtol=0, atol=2e-8)


def test_integrate_differentiate_round_trip(fmodel):
    rng = np.random.default_rng(1)
    for _ in range(20):
        q = random_q(fmodel, rng)
        v = rng.normal(size=fmodel.nv)
        q2 = q.copy()
        mj.mj_integratePos(fmodel, q2, v, 0.05)
        back = mj.mj_differentiatePos(fmodel, 0.05, q, q2)
        np.testing.assert_allclose(back, v, rtol=0, atol=1e-10)
        for j in range(fmodel.njnt):   # quaternions stay unit
            a, t = fmodel.jnt_qposadr[j], fmodel.jnt_type[j]
            if t == 0:
                assert abs(np.linalg.norm(q2[a + 3:a + 7]) - 1) < 1e-14
            elif t == 1:
                assert abs(np.linalg.norm(q2[a:a + 4]) - 1) < 1e-14


def test_com_jacobian_is_the_derivative_of_the_com(fmodel):
    rng = np.random.default_rng(2)
    eps = 1e-6
    task = o.ComTask(cost=np.ones(3), target_com=np.zeros(3))
    q = random_q(fmodel, rng)
    cfg = o.Configuration(fmodel, q)
    J = task.jacobian(cfg)
    for i in range(fmodel.nv):
        dv = np.zeros(fmodel.nv)
        dv[i] = 1.0
        qp, qm = q.copy(), q.copy()
        mj.mj_integratePos(fmodel, qp, dv, eps)
        mj.mj_integratePos(fmodel, qm, dv, -eps)
        fd = (task.error(o.Configuration(fmodel, qp)) - task.error(o.Configuration(fmodel, qm))) / (2 * eps)
        np.testing.assert_allclose(J[:, i], fd, rtol=0, atol=1e-8)


def test_collision_row_is_minus_the_distance_gradient():
    """For an active pair, G = -d(dist)/dq (what the reference's own test checks against
    MuJoCo's efc_J, tests/test_collision_avoidance_limit.py:65-111)."""
    model = load_builtin("tidybot_cube")
    g = np.load(os.path.join(HERE, "golden", "collision.npz"))
    pads = ["right_pad1", "right_pad2", "left_pad1", "left_pad2"]
    lim = o.CollisionAvoidanceLimit(model, [(pads, ["cube_geom", "floor"])], collision_detection_distance=0.05)
    eps = 1e-7
    checked = 0
    for q in g["q"][::2]:
        cfg = o.Configuration(model, q)
        G, h = lim.inequalities(cfg, 0.005)
        for row, (g1, g2) in enumerate(lim.geom_id_pairs):
            if not np.isfinite(h[row]):
                continue
            d0, _ = mj.mj_geomDistance(model, cfg.data, g1, g2, 0.05)
            if d0 < 1e-4:      # the overlap branch reports penetration depth, not a smooth distance
                continue
            grad = np.zeros(model.nv)
            smooth = True
            for i in range(model.nv):
                dv = np.zeros(model.nv)
                dv[i] = 1.0
                qp, qm = q.copy(), q.copy()
                mj.mj_integratePos(model, qp, dv, eps)
                mj.mj_integratePos(model, qm, dv, -eps)
                dp, _ = mj.mj_geomDistance(model, o.Configuration(model, qp).data, g1, g2, 0.05)
                dm, _ = mj.mj_geomDistance(model, o.Configuration(model, qm).data, g1, g2, 0.05)
                grad[i] = (dp - dm) / (2 * eps)
            np.testing.assert_allclose(G[row], -grad, rtol=0, atol=5e-6)
            checked += 1
    assert checked >= 4


def brute_force_qp(P, c, G, h):
    """Minimiser by enumerating active sets (tiny problems only)."""
    n, m = len(c), len(h)
    best, best_x = np.inf, None
    for k in range(0, min(n, m) + 1):
        for A in itertools.combinations(range(m), k):
            A = list(A)
            if k:
                K = np.block([[P, G[A].T], [G[A], np.zeros((k, k))]])
                if np.linalg.matrix_rank(K) < n + k:
                    continue
                sol = np.linalg.solve(K, np.concatenate([-c, h[A]]))
                x, lam = sol[:n], sol[n:]
                if (lam < -1e-10).any():
                    continue
            else:
                x = np.linalg.solve(P, -c)
            if (G @ x - h > 1e-9).any():
                continue
            f = 0.5 * x @ P @ x + c @ x
            if f < best:
                best, best_x = f, x
    return best_x


def test_qp_against_active_set_enumeration_and_kkt():
    rng = np.random.default_rng(3)
    for trial in range(30):
        n, m = 4, 6
        A = rng.normal(size=(n, n))
        P = A @ A.T + 0.1 * np.eye(n)
        c = rng.normal(size=n) * 3
        G = rng.normal(size=(m, n))
        h = rng.uniform(0.1, 1.0, m)
        x, status, active, u = o.solve_qp(P, c, G, h)
        assert status == 0
        np.testing.assert_allclose(x, brute_force_qp(P, c, G, h), rtol=0, atol=1e-9)
    for trial in range(10):   # larger: KKT conditions (sufficient: P is positive definite)
        n, m = 24, 50
        A = rng.normal(size=(n, n))
        P = A @ A.T + 1e-3 * np.eye(n)
        c = rng.normal(size=n) * 5
        G = rng.normal(size=(m, n))
        h = rng.uniform(0.05, 0.5, m)
        x, status, active, u = o.solve_qp(P, c, G, h)
        assert status == 0
        lam = np.zeros(m)
        lam[list(active)] = u
        assert (G @ x - h < 1e-9).all() and (lam > -1e-12).all()
        np.testing.assert_allclose(P @ x + c + G.T @ lam, 0, atol=1e-8)
        np.testing.assert_allclose(lam * (G @ x - h), 0, atol=1e-8)


# ---------------------------------------------------------------------- GPU ----
@pytest.mark.gpu
def test_gpu_floating_base_tree_vs_oracle(fmodel):
    """Free and ball joints on the task chain: served by the wave-per-problem kernels."""
    import torch
    from mink_amd.engine import DeviceModel, DeviceProblem
    dm = DeviceModel(fmodel, "cuda:0")
    rng = np.random.default_rng(4)
    B = 48
    q = np.array([random_q(fmodel, rng) for _ in range(B)])
    qs = np.array([random_q(fmodel, rng) for _ in range(B)])
    dq = torch.tensor(q, device="cuda:0")
    for ftype, fname in FRAMES:
        fid = {"site": fmodel.site_id, "geom": fmodel.geom_id, "body": fmodel.body_id}[ftype](fname)
        pose = dm.frame_pose(dq, L.FRAME_TYPES[ftype], fid).cpu().numpy()
        jac = dm.frame_jacobian(dq, L.FRAME_TYPES[ftype], fid).cpu().numpy()
        for b in range(0, B, 6):
            cfg = o.Configuration(fmodel, q[b])
            ref = cfg.get_transform_frame_to_world(fname, ftype)
            s = np.sign(ref[:4] @ pose[b, :4])
            np.testing.assert_allclose(pose[b, :4] * s, ref[:4], rtol=0, atol=1e-13)
            np.testing.assert_allclose(pose[b, 4:], ref[4:], rtol=0, atol=1e-13)
            np.testing.assert_allclose(jac[b], cfg.get_frame_jacobian(fname, ftype), rtol=0, atol=1e-12)
    site = fmodel.site_id("tool")
    targets = dm.frame_pose(torch.tensor(qs, device="cuda:0"), 2, site)
    tasks = [dict(type=L.TASK_FRAME, frame_type=2, frame_id=site, cost=np.array([1, 1, 1, 0.5, 0.5, 0.5]), gain=1.0,
                  lm_damping=0.1),
             dict(type=L.TASK_POSTURE, cost=np.full(fmodel.nv, 0.05), target_q=fmodel.qpos0.copy(), gain=1.0, lm_damping=0.0),
             dict(type=L.TASK_COM, cost=np.array([0.3, 0.3, 0.1]), gain=0.5, lm_damping=0.0)]
    vel = {"elbow": 1.0, "extend": 0.2, "hip": 0.5}
    limits = [dict(type=L.LIMIT_CONFIGURATION, gain=0.9, min_distance_from_limits=0.0),
              dict(type=L.LIMIT_VELOCITY, index=[int(fmodel.jnt_dofadr[fmodel.joint_id(k)]) for k in vel],
                   vmax=list(vel.values()))]
    prob = DeviceProblem(dm, tasks, limits)
    assert not prob.fused
    com_t = rng.normal(size=(B, 3)) * 0.1 + np.array([0, 0, 0.7])
    tg = torch.cat([targets, torch.tensor(com_t, device="cuda:0")], dim=1).contiguous()
    v, st = prob.solve(dq, tg, 0.01, 1e-4)
    assert int(st.abs().sum()) == 0
    v = v.cpu().numpy()
    tgn = tg.cpu().numpy()
    for b in range(0, B, 3):
        cfg = o.Configuration(fmodel, q[b])
        otasks = [o.FrameTask("tool", "site", 1.0, 0.5, lm_damping=0.1, target=tgn[b, :7]),
                  o.PostureTask(cost=np.full(fmodel.nv, 0.05), target_q=fmodel.qpos0.copy()),
                  o.ComTask(cost=np.array([0.3, 0.3, 0.1]), target_com=tgn[b, 7:], gain=0.5)]
        olimits = [o.ConfigurationLimit(fmodel, gain=0.9), o.VelocityLimit(fmodel, vel)]
        ref = o.solve_ik(cfg, otasks, 0.01, damping=1e-4, limits=olimits)
        np.testing.assert_allclose(v[b], ref, rtol=0, atol=1e-8)
    # integration of free / ball joints against the oracle
    vv = rng.normal(size=(B, fmodel.nv))
    qi = dq.clone()
    dm.integrate_(qi, torch.tensor(vv, device="cuda:0"), 0.02)
    for b in range(0, B, 5):
        ref = q[b].copy()
        mj.mj_integratePos(fmodel, ref, vv[b], 0.02)
        np.testing.assert_allclose(qi[b].cpu().numpy(), ref, rtol=0, atol=1e-14)


PRIM_PAIRS = [("obj_geom", "shin_geom"), ("shin_geom", "bar_geom"), ("ground", "shin_geom"), ("obj_geom", "torso_geom"),
              ("obj_geom", "fore_geom"), ("leg_geom", "obj_geom"), ("ground", "fore_geom"), ("torso_geom", "fore_geom"),
              ("upper_geom", "bar_geom"), ("ground", "obj_geom"), ("shin_geom", "fore_geom"), ("bar_geom", "torso_geom"),
              ("bar_geom", "fore_geom")]


CAPSULE_BOX_PAIRS = [("shin_geom", "fore_geom"), ("bar_geom", "torso_geom"), ("bar_geom", "fore_geom")]
CAPSULE_RADIUS = {"shin_geom": 0.03, "bar_geom": 0.02}


def near_configuration(model, rng, pair):
    """A random configuration with the free object moved next to one geom of `pair`."""
    q = random_q(model, rng)
    if "obj_geom" in pair:
        other = pair[0] if pair[1] == "obj_geom" else pair[1]
        cfg = o.Configuration(model, q)
        c = cfg.data.geom_xpos[model.geom_id(other)]
        d = rng.normal(size=3)
        a = model.jnt_qposadr[model.joint_id("obj_free")]
        q[a:a + 3] = c + d / np.linalg.norm(d) * rng.uniform(0.0, 0.25)
    return q


@pytest.mark.parametrize("pair", PRIM_PAIRS)
def test_primitive_distance_gradient_matches_contact_row(fmodel, pair):
    """d(dist)/dq by finite differences equals n'(Jp2 - Jp1) at the witness points: checks
    distance, witness points and the normal of every primitive routine together."""
    rng = np.random.default_rng(100 + PRIM_PAIRS.index(pair))
    g1, g2 = fmodel.geom_id(pair[0]), fmodel.geom_id(pair[1])
    eps, checked = 1e-7, 0
    for _ in range(12):
        q = near_configuration(fmodel, rng, pair)
        cfg = o.Configuration(fmodel, q)
        d0, fromto = mj.mj_geomDistance(fmodel, cfg.data, g1, g2, 10.0)
        if abs(d0) < 1e-3:
            continue
        nrm = fromto[3:] - fromto[:3]
        nrm = nrm / np.linalg.norm(nrm)
        j1 = mj.mj_jac(fmodel, cfg.data, fromto[:3], fmodel.geom_bodyid[g1])[0]
        j2 = mj.mj_jac(fmodel, cfg.data, fromto[3:], fmodel.geom_bodyid[g2])[0]
        row = nrm @ (j2 - j1)
        if d0 < 0:
            row = -row       # fromto points from geom1 to geom2 also when they overlap; the distance then grows the other way
        grad = np.zeros(fmodel.nv)
        for i in range(fmodel.nv):
            dv = np.zeros(fmodel.nv)
            dv[i] = 1.0
            qp, qm = q.copy(), q.copy()
            mj.mj_integratePos(fmodel, qp, dv, eps)
            mj.mj_integratePos(fmodel, qm, dv, -eps)
            dp, _ = mj.mj_geomDistance(fmodel, o.Configuration(fmodel, qp).data, g1, g2, 10.0)
            dm, _ = mj.mj_geomDistance(fmodel, o.Configuration(fmodel, qm).data, g1, g2, 10.0)
            grad[i] = (dp - dm) / (2 * eps)
        if pair == ("torso_geom", "fore_geom") and d0 < 0:
            continue         # overlapping boxes report a penetration depth, not a smooth distance
        if pair in CAPSULE_BOX_PAIRS and d0 < -CAPSULE_RADIUS[pair[0]]:
            continue         # capsule axis through the box: a face-exit depth, not a smooth distance either
        np.testing.assert_allclose(row, grad, rtol=0, atol=2e-5)
        checked += 1
    assert checked >= 3


@pytest.mark.gpu
def test_gpu_primitive_collision_rows_vs_oracle(fmodel):
    import torch
    from mink_amd.engine import DeviceModel, DeviceProblem
    dm = DeviceModel(fmodel, "cuda:0")
    rng = np.random.default_rng(11)
    pairs = [(fmodel.geom_id(a), fmodel.geom_id(b)) for a, b in PRIM_PAIRS]
    spec = dict(type=L.LIMIT_COLLISION, index=np.array(pairs, dtype=np.int32).reshape(-1), gain=0.85,
                minimum_distance_from_collisions=0.005, collision_detection_distance=0.3, bound_relaxation=0.01)
    prob = DeviceProblem(dm, [], [spec])
    qs = np.array([near_configuration(fmodel, rng, PRIM_PAIRS[k % len(PRIM_PAIRS)]) for k in range(60)])
    P, c, G, h = prob.build_qp(torch.tensor(qs, device="cuda:0"), None, 0.01, 1e-3)
    G, h = G.cpu().numpy(), h.cpu().numpy()
    lim = o.CollisionAvoidanceLimit(fmodel, [], gain=0.85, minimum_distance_from_collisions=0.005,
                                    collision_detection_distance=0.3, bound_relaxation=0.01)
    lim.geom_id_pairs = [(min(a, b), max(a, b)) for a, b in pairs]
    active = 0
    for b in range(len(qs)):
        Go, ho = lim.inequalities(o.Configuration(fmodel, qs[b]), 0.01)
        fin = np.isfinite(ho)
        np.testing.assert_array_equal(np.isfinite(h[b]), fin)
        np.testing.assert_allclose(h[b][fin], ho[fin], rtol=0, atol=1e-9)
        np.testing.assert_allclose(G[b], Go, rtol=0, atol=1e-9)
        active += int(fin.sum())
    assert active > 100


@pytest.mark.gpu
def test_gpu_biped_solve_ik_vs_oracle(biped):
    """Whole-body IK of the humanoid-shaped tree on the wave-per-problem kernels: both feet and a
    hand as frame tasks, the centre of mass, a posture task; joint and velocity limits; one step
    and a short loop against the oracle."""
    import torch
    from mink_amd.engine import DeviceModel, DeviceProblem
    m = biped
    dm = DeviceModel(m, "cuda:0")
    rng = np.random.default_rng(21)
    B = 40
    stand = m.key_qpos[0]
    q = np.array([stand.copy() for _ in range(B)])
    for b in range(B):      # around the standing pose, inside the joint ranges
        v0 = rng.normal(size=m.nv) * 0.15
        mj.mj_integratePos(m, q[b], v0, 1.0)
    for j in range(1, m.njnt):
        a = int(m.jnt_qposadr[j])
        q[:, a] = np.clip(q[:, a], m.jnt_range[j, 0] + 0.02, m.jnt_range[j, 1] - 0.02)
    dq = torch.tensor(q, device="cuda:0")
    frames = [("l_foot_site", 1.0, 1.0), ("r_foot_site", 1.0, 1.0), ("l_hand", 0.5, 0.1)]
    qs = np.array([stand.copy() for _ in range(B)])
    for b in range(B):
        mj.mj_integratePos(m, qs[b], rng.normal(size=m.nv) * 0.15, 1.0)
    tgs = [dm.frame_pose(torch.tensor(qs, device="cuda:0"), 2, m.site_id(n)) for n, _, _ in frames]
    com_t = torch.tensor(rng.normal(size=(B, 3)) * 0.03 + np.array([0.0, 0.0, 0.85]), device="cuda:0")
    tasks = [dict(type=L.TASK_FRAME, frame_type=2, frame_id=m.site_id(n), cost=np.array([pc] * 3 + [oc] * 3), gain=1.0,
                  lm_damping=0.5) for n, pc, oc in frames]
    tasks += [dict(type=L.TASK_COM, cost=np.array([1.0, 1.0, 0.2]), gain=0.8, lm_damping=0.0),
              dict(type=L.TASK_POSTURE, cost=np.concatenate([np.zeros(6), np.full(m.nv - 6, 0.05)]), target_q=stand.copy(),
                   gain=1.0, lm_damping=0.0)]
    vel = {m.jnt_names[j]: 3.0 for j in range(1, m.njnt)}
    limits = [dict(type=L.LIMIT_CONFIGURATION, gain=0.9, min_distance_from_limits=0.0),
              dict(type=L.LIMIT_VELOCITY, index=[int(m.jnt_dofadr[m.joint_id(k)]) for k in vel], vmax=list(vel.values()))]
    prob = DeviceProblem(dm, tasks, limits)
    assert not prob.fused
    tg = torch.cat(tgs + [com_t], dim=1).contiguous()
    v, st = prob.solve(dq, tg, 0.01, 1e-6)
    assert int(st.abs().sum()) == 0
    v = v.cpu().numpy()
    tgn = tg.cpu().numpy()

    def oracle_sets(b):
        ot = [o.FrameTask(n, "site", pc, oc, lm_damping=0.5, target=tgn[b, 7 * i:7 * i + 7]) for i, (n, pc, oc) in enumerate(frames)]
        ot += [o.ComTask(cost=np.array([1.0, 1.0, 0.2]), target_com=tgn[b, 21:24], gain=0.8),
               o.PostureTask(cost=np.concatenate([np.zeros(6), np.full(m.nv - 6, 0.05)]), target_q=stand.copy())]
        return ot, [o.ConfigurationLimit(m, gain=0.9), o.VelocityLimit(m, vel)]

    worst = 0.0
    for b in range(0, B, 2):
        ot, ol = oracle_sets(b)
        ref = o.solve_ik(o.Configuration(m, q[b]), ot, 0.01, damping=1e-6, limits=ol)
        worst = max(worst, float(np.abs(v[b] - ref).max()))
    print("biped, one step: max |dv| vs oracle =", worst)
    assert worst <= 1e-7
    # a five-step loop
    qi = dq.clone()
    res = prob.iterate_(qi, tg, 0.01, 1e-6, 5)
    assert int(res["status"].abs().sum()) == 0
    for b in range(0, B, 8):
        ot, ol = oracle_sets(b)
        cfg = o.Configuration(m, q[b])
        for _ in range(5):
            cfg.integrate_inplace(o.solve_ik(cfg, ot, 0.01, damping=1e-6, limits=ol), 0.01)
        np.testing.assert_allclose(qi[b].cpu().numpy(), cfg.q, rtol=0, atol=1e-8)
