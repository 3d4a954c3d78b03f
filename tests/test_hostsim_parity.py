"""CPU checks of the fused device functions through their TEST-ONLY host build
(tests/hostsim): the same templates the HIP kernels instantiate, called lane by lane,
against the reference's golden vectors and the oracle.  Skipped when the host build is
absent (it is produced by __graft_entry__.build())."""
import os

import numpy as np
import pytest

import hostsim_lib as hs
from mink_amd import workloads as W
from mink_amd.mjcf import load_builtin
from oracle import c_oracle as co

pytestmark = pytest.mark.skipif(not hs.available(), reason="tests/hostsim/libhostsim.so not built")

# Every case runs on both forms of the fused device code: "lane" (one problem per lane,
# mk_fused.hip) and "coop" (one problem per 16-lane row, mk_coop.hip, its row operations
# emulated by fibers).
FORMS = pytest.mark.parametrize("form", ["lane", "coop"])


@pytest.fixture(scope="module")
def model():
    return load_builtin("tidybot_cube")


@FORMS
def test_golden_single_step(model, golden_dir, form):
    g = np.load(os.path.join(golden_dir, "tidybot_build_ik.npz"))
    dt = float(g["dt"])
    v, st = hs.solve_ik(model, W.task_specs(model, True), W.limit_specs(model, True), g["q"], g["target"], dt, 1e-3,
                        form=form)
    assert not st.any()
    np.testing.assert_allclose(v, g["v"], rtol=0, atol=1e-10)
    v, st = hs.solve_ik(model, W.task_specs(model, False), W.limit_specs(model, False), g["q"], g["target"], dt, 1e-3,
                        form=form)
    np.testing.assert_allclose(v, g["v_default_limit"], rtol=0, atol=1e-10)
    v, st = hs.solve_ik(model, W.task_specs(model, False), [], g["q"], g["target"], dt, 1e-3, form=form)
    np.testing.assert_allclose(v, g["v_nolimit"], rtol=0, atol=1e-10)
    v, st = hs.solve_ik(model, W.task_specs(model, False), W.limit_specs(model, False), g["q"], g["target"], dt, 1e-12,
                        form=form)
    np.testing.assert_allclose(v, g["v12"], rtol=1e-7, atol=1e-7)


@FORMS
def test_golden_loop(model, golden_dir, form):
    g = np.load(os.path.join(golden_dir, "tidybot_loop.npz"))
    q, v, iters, err, st = hs.iterate(model, W.task_specs(model, False), W.limit_specs(model, False), g["q0"],
                                      g["target"], float(g["dt"]), 1e-3, 20, 1e-4, 1e-4, form=form)
    np.testing.assert_array_equal(iters, g["iters"])
    for k, n in enumerate(iters):
        np.testing.assert_allclose(q[k], g["q"][k, n - 1], rtol=0, atol=1e-11)
        np.testing.assert_allclose(err[k], g["err"][k, n - 1], rtol=0, atol=1e-10)
        np.testing.assert_allclose(v[k], g["v"][k, n - 1], rtol=0, atol=1e-8)


@pytest.mark.skipif(not co.available(), reason="C oracle not built")
@FORMS
@pytest.mark.parametrize("damping_task", [False, True])
def test_random_saturated_steps_vs_c_oracle(model, damping_task, form):
    """Far targets: most velocity bounds are active, which exercises the guessed active
    set, the multi-drop phase and the one-at-a-time additions."""
    from oracle import ik_oracle as o
    rng = np.random.default_rng(21)
    B = 3000 if form == "lane" else 600   # (the fibers of the row emulation are slower than one lane)
    q = W.sample_q(model, rng, B)
    targets = np.array([o.Configuration(model, qq).get_transform_frame_to_world("pinch_site", "site")
                        for qq in W.sample_q(model, rng, 64)])
    targets = targets[rng.integers(0, 64, B)]
    posts = [dict(cost=W.posture_cost(model), target_q=W.home_q(model))]
    if damping_task:
        posts.append(dict(cost=W.base_damping_cost(model), target_q=model.qpos0, gain=0.0))
    orc = co.COracle(model, postures=posts, velocity=W.VELOCITY_LIMITS)
    v_ref, st_ref = orc.solve_ik(q, targets, W.DT, W.DAMPING, nthreads=8)
    v, st = hs.solve_ik(model, W.task_specs(model, damping_task), W.limit_specs(model, True), q, targets, W.DT,
                        W.DAMPING, form=form)
    assert not st.any() and not st_ref.any()
    assert np.abs(v - v_ref).max() < 1e-9
    sat = np.isclose(np.abs(v[:, :10]), np.array(list(W.VELOCITY_LIMITS.values())), rtol=0, atol=1e-9)
    assert sat.sum() > 3 * B  # on average several saturated dofs per problem


@pytest.mark.skipif(not co.available(), reason="C oracle not built")
@FORMS
def test_iterate_warm_start_vs_c_oracle(model, form):
    """50 iterated steps: every step after the first is warm-started from the previous
    active set; the trajectory must still be the oracle's (cold dense solver)."""
    from oracle import ik_oracle as o
    rng = np.random.default_rng(22)
    B = 256 if form == "lane" else 64
    q0 = W.sample_q(model, rng, B)
    far = np.array([o.Configuration(model, qq).get_transform_frame_to_world("pinch_site", "site")
                    for qq in W.sample_q(model, rng, B)])
    near = W.offset_targets(np.array([o.Configuration(model, qq).get_transform_frame_to_world("pinch_site", "site")
                                      for qq in q0]), rng)
    targets = np.where((np.arange(B) % 2 == 0)[:, None], far, near)
    orc = co.COracle(model, postures=[dict(cost=W.posture_cost(model), target_q=W.home_q(model))],
                     velocity=W.VELOCITY_LIMITS)
    q_ref, it_ref, err_ref, st_ref = orc.iterate(q0, targets, W.DT, W.DAMPING, 50, nthreads=8)
    q, v, iters, err, st = hs.iterate(model, W.task_specs(model, False), W.limit_specs(model, True), q0, targets,
                                      W.DT, W.DAMPING, 50, -1.0, -1.0, form=form)
    assert not st.any() and not st_ref.any()
    assert np.abs(q - q_ref).max() < 1e-10
    assert np.abs(err - err_ref).max() < 1e-10
    # early exit variant
    q_ref, it_ref, err_ref, _ = orc.iterate(q0, targets, W.DT, W.DAMPING, 50, 1e-4, 1e-4, nthreads=8)
    q, v, iters, err, st = hs.iterate(model, W.task_specs(model, False), W.limit_specs(model, True), q0, targets,
                                      W.DT, W.DAMPING, 50, 1e-4, 1e-4, form=form)
    np.testing.assert_array_equal(iters, it_ref)
    assert np.abs(q - q_ref).max() < 1e-10
    assert (iters[1::2] < 50).mean() > 0.3   # many of the nearby targets converge within 50 velocity-limited steps


@FORMS
def test_padding_and_other_model(golden_dir, form):
    """mobile_kinova: nv = nq = 10, the whole model is the chain."""
    model = load_builtin("mobile_kinova")
    g = np.load(os.path.join(golden_dir, "mobile_kinova_build_ik.npz"))
    v, st = hs.solve_ik(model, W.task_specs(model, True), W.limit_specs(model, True), g["q"], g["target"],
                        float(g["dt"]), 1e-3, form=form)
    assert not st.any()
    np.testing.assert_allclose(v, g["v"], rtol=0, atol=1e-10)
    # a frame half way up the arm: 7-dof chain -> 7-slot kernel, 3 leftover arm dofs are clamps
    from oracle import ik_oracle as o
    from mink_amd import _lib as L
    rng = np.random.default_rng(5)
    q = W.sample_q(model, rng, 40)
    body = model.body_id("forearm_link")
    targets = np.array([o.Configuration(model, qq).get_transform_frame_to_world("forearm_link", "body")
                        for qq in W.sample_q(model, rng, 40)])
    specs = W.task_specs(model, False)
    specs[0].update(frame_type=L.FRAME_TYPES["body"], frame_id=body)
    v, st = hs.solve_ik(model, specs, W.limit_specs(model, True), q, targets, W.DT, W.DAMPING, form=form)
    for b in range(40):
        cfg = o.Configuration(model, q[b])
        tasks = [o.FrameTask("forearm_link", "body", 1.0, 1.0, lm_damping=1.0, target=targets[b]),
                 o.PostureTask(cost=W.posture_cost(model), target_q=W.home_q(model))]
        ref = o.solve_ik(cfg, tasks, W.DT, damping=W.DAMPING,
                         limits=[o.ConfigurationLimit(model), o.VelocityLimit(model, W.VELOCITY_LIMITS)])
        np.testing.assert_allclose(v[b], ref, rtol=0, atol=1e-9)


@FORMS
def test_cabinet_scene_model(form):
    """tidybot_open (the reference's open.xml: nq = nv = 20, two limited cabinet door hinges
    beside the robot, no free joint): fused path of the host build against the oracle."""
    from oracle import ik_oracle as o
    model = load_builtin("tidybot_open")
    assert model.nq == 20 and model.nv == 20
    rng = np.random.default_rng(9)
    B = 30
    q, qs = W.sample_q(model, rng, B), W.sample_q(model, rng, B)
    q[:, 18:] = rng.uniform(0.0, 0.5, (B, 2))      # doors ajar
    targets = np.array([o.Configuration(model, x).get_transform_frame_to_world("pinch_site", "site") for x in qs])
    v, st = hs.solve_ik(model, W.task_specs(model, True), W.limit_specs(model, True), q, targets, W.DT, W.DAMPING,
                        form=form)
    assert not st.any()
    for b in range(B):
        tasks = [o.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0, target=targets[b]),
                 o.PostureTask(cost=W.posture_cost(model), target_q=W.home_q(model)),
                 o.DampingTask(model, W.base_damping_cost(model))]
        ref = o.solve_ik(o.Configuration(model, q[b]), tasks, W.DT, damping=W.DAMPING,
                         limits=[o.ConfigurationLimit(model), o.VelocityLimit(model, W.VELOCITY_LIMITS)])
        np.testing.assert_allclose(v[b], ref, rtol=0, atol=1e-9)
