"""Cases of the reference's own unit tests (tests/test_frame_task.py, test_posture_task.py,
test_com_task.py, test_relative_frame_task.py, test_damping_task.py, test_configuration.py,
test_configuration_limit.py, test_velocity_limit.py) restated on the models shipped here:
TidyBot for the arm cases, the synthetic floating-base tree (free + ball joints) where the
reference uses a humanoid."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

import mink_amd as mink
from mink_amd.mjcf import load_builtin, load_mjcf

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def model():
    return load_builtin("tidybot_cube")


@pytest.fixture(scope="module")
def fmodel():
    return load_mjcf(os.path.join(HERE, "models", "float_arm.xml"))


@pytest.fixture()
def cfg(model):
    c = mink.Configuration(model)
    c.update_from_keyframe("home")
    return c


def random_float_q(fmodel, rng):
    q = fmodel.qpos0.copy()
    for j in range(fmodel.njnt):
        a, t = int(fmodel.jnt_qposadr[j]), int(fmodel.jnt_type[j])
        if t == 0:
            q[a:a + 3] += rng.uniform(-0.2, 0.2, 3)
            w = rng.normal(size=4)
            q[a + 3:a + 7] = w / np.linalg.norm(w)
        elif t == 1:
            w = rng.normal(size=4)
            q[a:a + 4] = w / np.linalg.norm(w)
        else:
            lo, hi = (fmodel.jnt_range[j] if fmodel.jnt_limited[j] else (-1.0, 1.0))
            q[a] = rng.uniform(lo + 0.05, hi - 0.05)
    return q


# ---- test_frame_task.py ----
def test_frame_task_target_is_a_copy_and_zero_error_at_target(cfg):
    task = mink.FrameTask("pinch_site", "site", position_cost=1.0, orientation_cost=1.0)
    target = cfg.get_transform_frame_to_world("pinch_site", "site")
    task.set_target(target)
    assert task.transform_target_to_world is not target
    np.testing.assert_array_equal(task.transform_target_to_world.wxyz_xyz.numpy(), target.wxyz_xyz.numpy())
    np.testing.assert_allclose(task.compute_error(cfg), np.zeros(6), atol=1e-12)
    task.set_target_from_configuration(cfg)
    np.testing.assert_allclose(task.compute_error(cfg), np.zeros(6), atol=1e-12)


def test_frame_task_unit_cost_qp_objective_and_lm_damping(cfg):
    task = mink.FrameTask("pinch_site", "site", position_cost=1.0, orientation_cost=1.0)
    T = cfg.get_transform_frame_to_world("pinch_site", "site")
    task.set_target(T @ mink.SE3.from_translation(np.array([0.0, 0.0, 0.1])))
    J, e = task.compute_jacobian(cfg), task.compute_error(cfg)
    H, c = task.compute_qp_objective(cfg)
    np.testing.assert_allclose(H, J.T @ J, atol=1e-12)
    np.testing.assert_allclose(c, e @ J, atol=1e-12)
    # Levenberg-Marquardt damping has no effect at the target
    task.set_target(T)
    H0, c0 = task.compute_qp_objective(cfg)
    damped = mink.FrameTask("pinch_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1e-3)
    damped.set_target(T)
    H1, c1 = damped.compute_qp_objective(cfg)
    np.testing.assert_allclose(H0, H1, atol=1e-14)
    np.testing.assert_allclose(c0, c1, atol=1e-14)


# ---- test_posture_task.py / test_damping_task.py ----
def test_posture_task_cases(model, cfg):
    task = mink.PostureTask(model, cost=1.0)
    task.set_target_from_configuration(cfg)
    np.testing.assert_allclose(task.compute_error(cfg), np.zeros(model.nv), atol=1e-14)
    target = cfg.q.copy()
    task.set_target(target)
    target[3] += 1.0
    assert task.target_q[3] == cfg.q[3]          # the target is a copy
    # unit cost: H = I off the free joint, c = -e
    q2 = cfg.q.copy()
    q2[3:10] += 0.1
    task.set_target(q2)
    H, c = task.compute_qp_objective(cfg)
    e = task.compute_error(cfg)
    free = np.zeros(model.nv, dtype=bool)
    free[18:24] = True                            # the cube's free joint
    np.testing.assert_allclose(H, np.diag((~free).astype(float)), atol=1e-14)
    np.testing.assert_allclose(c, -e, atol=1e-14)
    np.testing.assert_allclose(e[~free], (q2 - cfg.q)[:18], atol=1e-14)
    assert not e[free].any()
    # zero cost is the same as no task
    zero = mink.PostureTask(model, cost=0.0)
    zero.set_target(q2)
    H, c = zero.compute_qp_objective(cfg)
    assert not H.any() and not c.any()


def test_damping_task_qp_objective(model, cfg):
    H, c = mink.DampingTask(model, cost=1.0).compute_qp_objective(cfg)
    free = np.zeros(model.nv, dtype=bool)
    free[18:24] = True
    np.testing.assert_allclose(H, np.diag((~free).astype(float)), atol=1e-14)
    np.testing.assert_allclose(c, np.zeros(model.nv), atol=1e-14)


# ---- test_com_task.py ----
def test_com_task_cases(fmodel):
    rng = np.random.default_rng(0)
    cfg = mink.Configuration(fmodel, random_float_q(fmodel, rng))
    task = mink.ComTask(cost=1.0)
    with pytest.raises(mink.TargetNotSet):
        task.compute_error(cfg)
    task.set_target_from_configuration(cfg)
    np.testing.assert_allclose(task.compute_error(cfg), np.zeros(3), atol=1e-13)
    t = task.target_com.copy()
    t[0] += 1.0
    assert task.target_com[0] != t[0]
    zero = mink.ComTask(cost=0.0)
    zero.set_target(np.zeros(3))
    H, c = zero.compute_qp_objective(cfg)
    assert not H.any() and not c.any()
    np.testing.assert_allclose(mink.ComTask(cost=[1.0, 2.0, 3.0]).cost, [1.0, 2.0, 3.0])


# ---- test_relative_frame_task.py ----
def test_relative_frame_task_matches_frame_task(fmodel):
    """Relative to the world body the task is the frame task with the opposite sign."""
    rng = np.random.default_rng(1)
    cfg = mink.Configuration(fmodel, random_float_q(fmodel, rng))
    T_wt = mink.SE3.sample_uniform()
    rel = mink.RelativeFrameTask("fore", "body", "world", "body", position_cost=1.0, orientation_cost=5.0)
    rel.set_target(T_wt)
    frame = mink.FrameTask("fore", "body", position_cost=1.0, orientation_cost=5.0)
    frame.set_target(T_wt)
    np.testing.assert_allclose(frame.compute_error(cfg), -rel.compute_error(cfg), atol=1e-12)
    np.testing.assert_allclose(frame.compute_jacobian(cfg), -rel.compute_jacobian(cfg), atol=1e-11)
    # set_target_from_configuration stores the current relative pose
    rel2 = mink.RelativeFrameTask("tool", "site", "torso", "body", 1.0, 1.0)
    rel2.set_target_from_configuration(cfg)
    np.testing.assert_allclose(rel2.compute_error(cfg), np.zeros(6), atol=1e-12)


# ---- test_configuration.py ----
def test_configuration_cases(model, fmodel):
    cfg = mink.Configuration(model)
    assert cfg.nq == model.nq and cfg.nv == model.nv
    cfg.update_from_keyframe("home")
    np.testing.assert_array_equal(cfg.q, model.key_qpos[model.key_id("home")])
    with pytest.raises(mink.InvalidKeyframe):
        cfg.update_from_keyframe("nope")
    with pytest.raises(mink.InvalidFrame):
        cfg.get_transform_frame_to_world("nope", "site")
    with pytest.raises(mink.UnsupportedFrame):
        cfg.get_frame_jacobian("pinch_site", "joint")
    # in-place integration equals integrate()
    v = np.ones(model.nv) * 0.05
    expected = cfg.integrate(v, 1e-2)
    cfg.integrate_inplace(v, 1e-2)
    np.testing.assert_allclose(cfg.q, expected, atol=0)
    # limits: tolerance, exception and warning
    q = model.key_qpos[0].copy()
    j = model.joint_id("joint_2")
    q[model.jnt_qposadr[j]] = model.jnt_range[j, 1] + 5e-7
    mink.Configuration(model, q).check_limits()            # inside the 1e-6 tolerance
    q[model.jnt_qposadr[j]] = model.jnt_range[j, 1] + 1e-3
    bad = mink.Configuration(model, q)
    with pytest.raises(mink.NotWithinConfigurationLimits):
        bad.check_limits()
    bad.check_limits(safety_break=False)                   # warns only
    # a free joint far away is not a limit violation
    fq = fmodel.qpos0.copy()
    fq[0:3] = [100.0, -50.0, 30.0]
    mink.Configuration(fmodel, fq).check_limits()


# ---- test_configuration_limit.py / test_velocity_limit.py ----
def test_configuration_and_velocity_limit_cases(model, fmodel, cfg):
    with pytest.raises(mink.LimitDefinitionError):
        mink.ConfigurationLimit(model, gain=-1.0)
    with pytest.raises(mink.LimitDefinitionError):
        mink.ConfigurationLimit(model, gain=1.1)
    limit = mink.ConfigurationLimit(model)
    nb = int(np.sum(model.jnt_limited[:18] != 0))
    assert limit.projection_matrix.shape == (nb, model.nv) and len(limit.indices) == nb
    G, h = limit.compute_qp_inequalities(cfg, 1e-3)
    assert G.shape == (2 * nb, model.nv) and h.shape == (2 * nb,)
    # far from the limits the velocity bounds are the tighter ones
    vel = mink.VelocityLimit(model, {model.jnt_names[j]: np.pi for j in (4, 6, 8)})
    full = np.zeros(model.nv)
    full[vel.indices] = vel.limit
    rows = (np.abs(G) @ (full > 0)) > 0          # the arm joints' rows (the fingers rest on their limits at home)
    assert rows.sum() == 6
    assert np.max((+G @ full * 1e-3 - h)[rows]) < -1e-10 and np.max((-G @ full * 1e-3 - h)[rows]) < -1e-10
    # free joints are ignored; ball joints take 3-vectors
    flimit = mink.ConfigurationLimit(fmodel)
    free_dofs = set(range(6)) | set(range(fmodel.nv - 6, fmodel.nv))
    assert not (set(flimit.indices.tolist()) & free_dofs)
    ball = mink.VelocityLimit(fmodel, {"shoulder": (1.0, 2.0, 3.0), "elbow": 0.5})
    assert len(ball.indices) == 4 and ball.projection_matrix.shape == (4, fmodel.nv)
    with pytest.raises(mink.LimitDefinitionError):
        mink.VelocityLimit(fmodel, {"shoulder": (1.0, 2.0)})
    with pytest.raises(mink.LimitDefinitionError):
        mink.VelocityLimit(fmodel, {"root": np.ones(6)})
    fcfg = mink.Configuration(fmodel, random_float_q(fmodel, np.random.default_rng(2)))
    G, h = ball.compute_qp_inequalities(fcfg, 1e-2)
    np.testing.assert_allclose(h, 1e-2 * np.array([1.0, 2.0, 3.0, 0.5] * 2), atol=1e-15)
    # a model without limited joints / without velocity limits gives an inactive constraint
    assert mink.VelocityLimit(model).compute_qp_inequalities(cfg, 1e-3).inactive


def test_configuration_limit_repulsion(model, cfg):
    """tests/test_configuration_limit.py:147-160 of the reference: with the bounds overwritten
    to q -/+ slack the rows allow at most the slack, and solve_ik honours them."""
    dt, slack_vel = 1e-3, 5e-4
    limit = mink.ConfigurationLimit(model, gain=0.5)
    limit.lower = cfg.integrate(-slack_vel * np.ones(cfg.nv), dt)
    limit.upper = cfg.integrate(+slack_vel * np.ones(cfg.nv), dt)
    G, h = limit.compute_qp_inequalities(cfg, dt)
    assert np.max(h) < slack_vel * dt + 1e-10 and np.min(h) > -slack_vel * dt - 1e-10
    assert G.shape == (2 * len(limit.indices), model.nv)
    # a task that pulls hard on joint_2 moves it by no more than gain * slack per step
    posture = mink.PostureTask(model, cost=1.0)
    target = cfg.q.copy()
    j2 = int(model.jnt_qposadr[model.joint_id("joint_2")])
    target[j2] += 1.0
    posture.set_target(target)
    v = mink.solve_ik(cfg, [posture], dt, "quadprog", 1e-6, limits=[limit])
    d2 = int(model.jnt_dofadr[model.joint_id("joint_2")])
    assert abs(v[d2] * dt - 0.5 * slack_vel * dt) < 1e-12
    d1 = int(model.jnt_dofadr[model.joint_id("joint_1")])      # an unlimited joint is free to move
    assert abs(v[d1]) < 1e-9 and abs(v[d2]) > 0
