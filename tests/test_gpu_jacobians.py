"""Task Jacobians of the device code against finite differences of the device code's own
errors: the cases of the reference's tests/test_jacobians.py (every task type, J = de/dq along
mj_integratePos steps of 1e-6, same tolerances), on the floating-base tree of the test models
(free + ball + hinge + slide joints), on the humanoid-shaped tree tests/models/biped.xml (a free
joint, two six-hinge legs, two arms: the shape of the reference's G1 model) and on TidyBot.
The nv + 1 configurations of one check are evaluated as one batch by mk_task_eval_batch."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

import mink_amd as mink
from mink_amd.mjcf import load_builtin, load_mjcf

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
H_STEP = 1e-6
N_CONFIGS = 4


def random_configurations(model, rng, n):
    """Uniform within the joint ranges (+-2 pi where a joint has none), unit quaternions for
    free and ball joints: tests/test_jacobians.py:19-39 of the reference, which leaves the
    quaternions unnormalised and lets MuJoCo normalise them."""
    q = np.tile(model.qpos0, (n, 1))
    for j in range(model.njnt):
        a, t = int(model.jnt_qposadr[j]), int(model.jnt_type[j])
        if t == 0:
            q[:, a:a + 3] = rng.uniform(-1.0, 1.0, (n, 3))
            w = rng.normal(size=(n, 4))
            q[:, a + 3:a + 7] = w / np.linalg.norm(w, axis=1, keepdims=True)
        elif t == 1:
            w = rng.normal(size=(n, 4))
            q[:, a:a + 4] = w / np.linalg.norm(w, axis=1, keepdims=True)
        else:
            lo, hi = model.jnt_range[j] if model.jnt_limited[j] else (-2 * np.pi, 2 * np.pi)
            q[:, a] = rng.uniform(lo, hi, n)
    return q


def check_jacobian_finite_diff(model, task, tol, rng, ball_from=None):
    """ball_from: configuration whose ball-joint quaternions the samples take.  The posture
    Jacobian of the reference is -I (mink/tasks/posture_task.py:114-142), which is de/dq of a
    ball joint only at zero error of that joint; the reference's humanoid has no ball joint."""
    nv = model.nv
    for q0 in random_configurations(model, rng, N_CONFIGS):
        if ball_from is not None:
            for j in np.nonzero(model.jnt_type == 1)[0]:
                a = int(model.jnt_qposadr[j])
                q0[a:a + 4] = ball_from[a:a + 4]
        base = mink.Configuration(model, q0)
        J0 = task.compute_jacobian(base)
        # row 0: q0 itself; row 1 + i: q0 (+) h e_i
        steps = torch.zeros((nv + 1, nv), dtype=torch.float64, device=base.device)
        steps[1:] = torch.eye(nv, dtype=torch.float64, device=base.device)
        batch = mink.Configuration(model, base.q_tensor.repeat(nv + 1, 1))
        batch.integrate_inplace(steps, H_STEP)
        e = task.compute_error(batch).cpu().numpy()
        J_finite = ((e[1:] - e[0]) / H_STEP).T
        assert J0.shape == J_finite.shape
        assert np.linalg.norm(J0 - J_finite, ord=np.inf) < tol


@pytest.fixture(scope="module")
def fmodel():
    return load_mjcf(os.path.join(HERE, "models", "float_arm.xml"))


@pytest.fixture(scope="module")
def tidybot():
    return load_builtin("tidybot_cube")


@pytest.fixture(scope="module")
def biped():
    return load_mjcf(os.path.join(HERE, "models", "biped.xml"))


def test_frame_task(fmodel, tidybot, biped):
    rng = np.random.default_rng(42)
    for model, name, ftype in ((fmodel, "tool", "site"), (fmodel, "fore", "body"), (tidybot, "pinch_site", "site"),
                               (biped, "l_foot_site", "site"), (biped, "r_hand", "site"), (biped, "head", "site"),
                               (biped, "r_sole", "geom")):
        task = mink.FrameTask(frame_name=name, frame_type=ftype, position_cost=1.0, orientation_cost=1.0)
        task.set_target(mink.SE3.sample_uniform(generator=torch.Generator().manual_seed(7)))
        check_jacobian_finite_diff(model, task, 1e-5, rng)


def test_relative_frame_task(fmodel, tidybot, biped):
    rng = np.random.default_rng(43)
    cases = ((fmodel, "tool", "site", "torso", "body"), (tidybot, "pinch_site", "site", "base_link", "body"),
             (biped, "r_hand", "site", "l_foot", "body"), (biped, "l_foot_site", "site", "r_foot_site", "site"))
    for model, name, ftype, root, rtype in cases:
        task = mink.RelativeFrameTask(frame_name=name, frame_type=ftype, root_name=root, root_type=rtype,
                                      position_cost=1.0, orientation_cost=1.0)
        task.set_target(mink.SE3.sample_uniform(generator=torch.Generator().manual_seed(7)))
        check_jacobian_finite_diff(model, task, 1e-5, rng)


def test_posture_task(fmodel, tidybot, biped):
    rng = np.random.default_rng(44)
    for model in (fmodel, tidybot, biped):
        task = mink.PostureTask(model=model, cost=1.0)
        target = mink.Configuration(model, model.qpos0).integrate(rng.normal(size=model.nv) * 1e-3, 1.0)
        task.set_target(target)
        check_jacobian_finite_diff(model, task, 1e-6, rng, ball_from=np.asarray(target))


def test_com_task(fmodel, tidybot, biped):
    rng = np.random.default_rng(45)
    for model in (fmodel, tidybot, biped):
        task = mink.ComTask(cost=1.0)
        task.set_target(np.zeros(3))
        check_jacobian_finite_diff(model, task, 1e-6, rng)


def test_damping_task(fmodel, tidybot, biped):
    rng = np.random.default_rng(46)
    for model in (fmodel, tidybot, biped):
        check_jacobian_finite_diff(model, mink.DampingTask(model, cost=1.0), 1e-6, rng, ball_from=model.qpos0)
