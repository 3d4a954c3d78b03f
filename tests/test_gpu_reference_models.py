"""The reference's small test models through the whole stack on the GPU: MJCF front end ->
device tables -> kernels, against the numpy restatement and the assertions of
tests/test_configuration_limit.py, tests/test_velocity_limit.py and tests/test_utils.py."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import mink_amd as mink
import ref_models as R
from mink_amd.mjcf import load_mjcf_string
from oracle import ik_oracle as o

pytestmark = pytest.mark.gpu


def rand_q(model, rng):
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
        else:
            lo, hi = (model.jnt_range[j] if model.jnt_limited[j] else (-1.0, 1.0))
            q[a] = rng.uniform(lo + 0.1 * (hi - lo), hi - 0.1 * (hi - lo))
    return q


@pytest.mark.parametrize("xml", [R.SUBSET_LIMITED, R.FREEJOINT_AND_HINGE, R.BALL_AND_HINGE])
def test_limits_and_solve_on_reference_models(xml):
    model = load_mjcf_string(xml)
    rng = np.random.default_rng(0)
    dt = 1e-3
    vel = {}
    for j in range(model.njnt):
        if model.jnt_type[j] == 1:
            vel[model.jnt_names[j]] = (np.pi, np.pi / 2, np.pi / 4)
        elif model.jnt_type[j] >= 2:
            vel[model.jnt_names[j]] = (0.5,)
    for trial in range(4):
        q = rand_q(model, rng)
        cfg = mink.Configuration(model, q)
        ocfg = o.Configuration(model, q)
        # ConfigurationLimit / VelocityLimit rows: shapes of the reference's tests, values of the restatement
        climit = mink.ConfigurationLimit(model)
        con = climit.compute_qp_inequalities(cfg, dt)
        nb = len(climit.indices)
        assert con.G.shape == (2 * nb, model.nv) and con.h.shape == (2 * nb,)
        Gr, hr = o.ConfigurationLimit(model).inequalities(ocfg, dt)
        np.testing.assert_array_equal(con.G, Gr)
        np.testing.assert_allclose(con.h, hr, rtol=0, atol=1e-13)
        vlimit = mink.VelocityLimit(model, vel)
        con = vlimit.compute_qp_inequalities(cfg, dt)
        nbv = len(vlimit.indices)
        assert con.G.shape == (2 * nbv, model.nv)
        np.testing.assert_allclose(con.h, np.concatenate([dt * vlimit.limit, dt * vlimit.limit]), rtol=0, atol=1e-18)
        # a posture solve under both limits (ball joint: 3 dofs, free joint: no posture rows)
        target = rand_q(model, rng)
        task = mink.PostureTask(model, cost=1.0)
        task.set_target(target)
        v = mink.solve_ik(cfg, [task], dt, "quadprog", 1e-6, limits=[climit, vlimit])
        otask = o.PostureTask(cost=np.ones(model.nv), target_q=target)
        vr = o.solve_ik(ocfg, [otask], dt, damping=1e-6,
                        limits=[o.ConfigurationLimit(model), o.VelocityLimit(model, {k: np.atleast_1d(x) for k, x in vel.items()})])
        np.testing.assert_allclose(v, vr, rtol=0, atol=1e-8)
        assert np.all(np.abs(v[vlimit.indices]) <= vlimit.limit + 1e-9)
        # integrate on the manifold: quaternions stay unit
        cfg.integrate_inplace(v, dt)
        ocfg.integrate_inplace(vr, dt)
        np.testing.assert_allclose(cfg.q, ocfg.q, rtol=0, atol=1e-12)


def test_move_mocap_to_frame():
    """tests/test_utils.py:62-103 of the reference."""
    model = load_mjcf_string(R.MOCAP)
    rng = np.random.default_rng(1)
    q = rand_q(model, rng)
    cfg = mink.Configuration(model, q)
    body = cfg.get_transform_frame_to_world("test", "body")
    mocap = cfg.get_transform_frame_to_world("mocap", "body")
    np.testing.assert_allclose(mocap.wxyz_xyz.numpy(), [np.sqrt(0.5), np.sqrt(0.5), 0, 0, 0.5, 1, 5], atol=1e-15)
    assert not np.allclose(mocap.translation().numpy(), body.translation().numpy())
    mink.move_mocap_to_frame(model, cfg.data, "mocap", "test", "body")
    moved = cfg.get_transform_frame_to_world("mocap", "body")
    np.testing.assert_allclose(moved.wxyz_xyz.numpy(), body.wxyz_xyz.numpy(), rtol=0, atol=1e-15)
    ref = o.Configuration(model, q).get_transform_frame_to_world("test", "body")
    np.testing.assert_allclose(body.wxyz_xyz.numpy(), ref, rtol=0, atol=1e-13)
    np.testing.assert_allclose(mink.SE3.from_mocap_name(model, cfg.data, "mocap").wxyz_xyz.numpy(), ref, rtol=0, atol=1e-13)
    assert float(np.abs(cfg.get_frame_jacobian("mocap", "body")).max()) == 0.0
    # a task on a frame of the mocap body is refused by the library (its pose is not part of q)
    task = mink.FrameTask("mocap", "body", 1.0, 1.0)
    task.set_target(moved)
    with pytest.raises(Exception) as e:
        mink.solve_ik(cfg, [task], 1e-3, "quadprog", 1e-3)
    assert "mocap" in str(e.value)
    # batched: every configuration has its own mocap arrays
    qb = np.stack([rand_q(model, rng) for _ in range(5)])
    bcfg = mink.Configuration(model, qb)
    mink.move_mocap_to_frame(model, bcfg.data, "mocap", "test", "body")
    got = bcfg.get_transform_frame_to_world("mocap", "body").wxyz_xyz.cpu().numpy()
    want = np.array([o.Configuration(model, x).get_transform_frame_to_world("test", "body") for x in qb])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13)


def test_zero_damping_on_a_dof_without_cost_is_reported():
    """damping = 0 with only a FrameTask without Levenberg-Marquardt term: the dofs outside the task's chain have a
    zero Hessian entry; the reference's QP backend fails there (assert in mink/solve_ik.py:103)."""
    from mink_amd.mjcf import load_builtin
    model = load_builtin("tidybot_cube")
    q = np.tile(model.key_qpos[0], (3, 1))
    cfg = mink.Configuration(model, q)
    task = mink.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=0.0)
    task.set_target_from_configuration(cfg)
    # as in the reference, the call itself fails ...
    with pytest.raises(AssertionError):
        mink.solve_ik(cfg, [task], 0.005, "quadprog", 0.0, limits=[])
    # ... unless the caller defers the reports of its batched calls: then nothing waits for the
    # device, and the failure surfaces at the end of the block (or at an explicit check())
    with pytest.raises(AssertionError):
        with cfg.deferred_checks():
            v = mink.solve_ik(cfg, [task], 0.005, "quadprog", 0.0, limits=[])
            assert float(v.abs().max()) == 0.0
            v2 = mink.solve_ik(cfg, [task], 0.005, "quadprog", 1e-12, limits=[])     # a good call does not clear it
    assert float(v2.abs().max()) <= 1e-9
    v = mink.solve_ik(cfg, [task], 0.005, "quadprog", 1e-12, limits=[])
    assert float(v.abs().max()) <= 1e-9
    with cfg.deferred_checks():
        v = mink.solve_ik(cfg, [task], 0.005, "quadprog", 1e-12, limits=[])
    assert cfg.defer_checks is False


def test_posture_target_update_keeps_the_problem():
    """A PostureTask whose target changes from call to call (PostureTask.set_target,
    mink/tasks/posture_task.py:62-85) reuses its device problem: the target is updated in place,
    on the fused and on the general path, and the answers follow it."""
    from mink_amd.mjcf import load_builtin
    model = load_builtin("tidybot_cube")
    rng = np.random.default_rng(4)
    from mink_amd import workloads as W
    q = W.sample_q(model, rng, 4)
    for extra in ([], [mink.ComTask(cost=0.1)]):      # fused path / general path
        cfg = mink.Configuration(model, q)
        ftask = mink.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
        ftask.set_target_from_configuration(cfg)
        ptask = mink.PostureTask(model, cost=W.posture_cost(model))
        for t in extra:
            t.set_target_from_configuration(cfg)
        n_before = None
        for trial in range(3):
            target = W.sample_q(model, rng, 1)[0]
            ptask.set_target(target)
            v = mink.solve_ik(cfg, [ftask, ptask] + extra, 0.005, "quadprog", 1e-3).cpu().numpy()
            n_now = len(cfg.dmodel.problem_cache)
            assert n_before is None or n_now == n_before, "a changed posture target must not create a new problem"
            n_before = n_now
            for b in range(q.shape[0]):
                ocfg = o.Configuration(model, q[b])
                ot = [o.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0,
                                  target=ocfg.get_transform_frame_to_world("pinch_site", "site")),
                      o.PostureTask(cost=W.posture_cost(model), target_q=target)]
                if extra:
                    ot.append(o.ComTask(cost=np.full(3, 0.1), target_com=np.asarray(extra[0].target_com).reshape(-1, 3)[b]))
                vr = o.solve_ik(ocfg, ot, 0.005, damping=1e-3, limits=[o.ConfigurationLimit(model)])
                np.testing.assert_allclose(v[b], vr, rtol=0, atol=1e-8)


def test_limited_ball_joint_rows_and_solve(golden_dir):
    """A LIMITED ball joint: the rows of ConfigurationLimit against the reference's
    (golden limited_ball.npz: mink/limits/configuration_limit.py:69-124 differences the
    configuration with `upper` / `lower` read as quaternions), and solve_ik with that limit
    against the oracle."""
    import os
    g = np.load(os.path.join(golden_dir, "limited_ball.npz"))
    model = mink.load_mjcf_string(R.LIMITED_BALL_AND_HINGE)
    limit = mink.ConfigurationLimit(model, gain=float(g["gain"]), min_distance_from_limits=float(g["min_distance_from_limits"]))
    cfg = mink.Configuration(model, g["q"])
    con = limit.compute_qp_inequalities(cfg, float(g["dt"]))
    np.testing.assert_allclose(torch.as_tensor(con.G).cpu().numpy(), g["G"], rtol=0, atol=0)
    np.testing.assert_allclose(torch.as_tensor(con.h).cpu().numpy(), g["h"], rtol=0, atol=1e-13)
    # solve_ik with the limit (generic route: rows from the host mirror, QP by mk_qp_solve_batch)
    task = mink.FrameTask("tip", "site", 1.0, 1.0, lm_damping=1.0)
    rng = np.random.default_rng(3)
    tg = np.array([np.concatenate([rq / np.linalg.norm(rq), rng.uniform(-0.3, 0.3, 3)])
                   for rq in rng.normal(size=(len(g["q"]), 4))])
    task.set_target(mink.SE3(torch.tensor(tg, device="cuda:0")))
    v = mink.solve_ik(cfg, [task], 0.01, "quadprog", 1e-3, limits=[limit]).cpu().numpy()
    for b in range(len(g["q"])):
        ocfg = o.Configuration(model, g["q"][b])
        ot = [o.FrameTask("tip", "site", 1.0, 1.0, lm_damping=1.0, target=tg[b])]
        ol = o.ConfigurationLimit(model, gain=float(g["gain"]), min_distance_from_limits=float(g["min_distance_from_limits"]))
        P, c, G, h = o.build_ik(ocfg, ot, 0.01, 1e-3, [ol])
        np.testing.assert_allclose(G, g["G"][b], rtol=0, atol=0)
        np.testing.assert_allclose(h, g["h"][b], rtol=0, atol=1e-13)
        x, status, _, _ = o.solve_qp(P, c, G, h)
        if status == 0:
            np.testing.assert_allclose(v[b], x / 0.01, rtol=0, atol=1e-8)
