"""GPU parity tests: the HIP path through the C ABI against the golden vectors of the
reference and against the numpy oracle on seeded inputs.  Tolerances are absolute, in
rad/s (or m/s) on v: the stated bound is 1e-6; we assert 1e-9 except where noted."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.mjcf import load_builtin

pytestmark = pytest.mark.gpu

V_TOL = 1e-9


@pytest.fixture(autouse=True, params=["lane", "coop"])
def kernel_form(request):
    """Every test of this module runs twice: with the problems of the fused path on the
    lane-per-problem kernels (mk_fused.hip) and on the cooperative ones (mk_coop.hip)."""
    with L.options(coop=1 if request.param == "coop" else 0):
        yield request.param


@pytest.fixture(scope="module")
def env():
    from mink_amd.engine import DeviceModel
    assert torch.cuda.is_available()
    model = load_builtin("tidybot_cube")
    dm = DeviceModel(model, "cuda:0")
    return model, dm


def dev(a, dtype=torch.float64):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda:0")


def oracle_tasks_limits(model, targets_b, damping_task, velocity):
    from oracle import ik_oracle as o
    tasks = [o.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0, target=targets_b),
             o.PostureTask(cost=W.posture_cost(model), target_q=W.home_q(model))]
    if damping_task:
        tasks.append(o.DampingTask(model, W.base_damping_cost(model)))
    limits = [o.ConfigurationLimit(model)]
    if velocity:
        limits.append(o.VelocityLimit(model, W.VELOCITY_LIMITS))
    return tasks, limits


def test_golden_single_step(env, golden_dir):
    from mink_amd.engine import DeviceProblem
    model, dm = env
    g = np.load(os.path.join(golden_dir, "tidybot_build_ik.npz"))
    q, tg = dev(g["q"]), dev(g["target"])
    dt = float(g["dt"])
    p2 = DeviceProblem(dm, W.task_specs(model, True), W.limit_specs(model, True))
    assert p2.fused and p2.n_chain == 10 and p2.target_stride == 7 and p2.n_rows == 22 + 20
    v, st = p2.solve(q, tg, dt, 1e-3)
    assert int(st.abs().sum()) == 0
    np.testing.assert_allclose(v.cpu().numpy(), g["v"], rtol=0, atol=V_TOL)
    p1 = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))
    v, st = p1.solve(q, tg, dt, 1e-3)
    np.testing.assert_allclose(v.cpu().numpy(), g["v_default_limit"], rtol=0, atol=V_TOL)
    p0 = DeviceProblem(dm, W.task_specs(model, False), [])
    assert p0.n_rows == 0
    v, st = p0.solve(q, tg, dt, 1e-3)
    np.testing.assert_allclose(v.cpu().numpy(), g["v_nolimit"], rtol=0, atol=V_TOL)


def test_golden_default_damping_both_kernels(env, golden_dir):
    """solve_ik at the API's default damping = 1e-12 (mink/solve_ik.py:73) against the
    reference's v on the 24 golden samples, as a measured ABSOLUTE bound in rad/s, on the
    lane-per-problem kernel and on the wave-per-problem kernel (the same problem made
    ineligible for the structured path by a ComTask of zero cost), and through the dense
    mk_build_qp_batch + mk_qp_solve_batch route.  |v| reaches ~1e2 here and the Hessian of
    a fulfilled task has entries down to 1e-12."""
    from mink_amd.engine import DeviceProblem, qp_solve
    model, dm = env
    g = np.load(os.path.join(golden_dir, "tidybot_build_ik.npz"))
    q, tg = dev(g["q"]), dev(g["target"])
    dt = float(g["dt"])
    ref = g["v12"]
    p1 = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))
    assert p1.fused
    v_fused, st = p1.solve(q, tg, dt, 1e-12)
    assert int(st.abs().sum()) == 0
    com = dict(type=L.TASK_COM, cost=np.zeros(3), gain=1.0, lm_damping=0.0)
    pd = DeviceProblem(dm, W.task_specs(model, False) + [com], W.limit_specs(model, False))
    assert not pd.fused
    tg_d = torch.cat([tg, torch.zeros((tg.shape[0], 3), dtype=torch.float64, device=tg.device)], dim=1).contiguous()
    v_dense, st = pd.solve(q, tg_d, dt, 1e-12)
    assert int(st.abs().sum()) == 0
    P, c, G, h = p1.build_qp(q, tg, dt, 1e-12)
    x, st = qp_solve(P, c, G, h)
    assert int(st.abs().sum()) == 0
    worst = {}
    for name, v in (("fused", v_fused), ("dense", v_dense), ("build+qp", x / dt)):
        d = np.abs(v.cpu().numpy() - ref)
        worst[name] = float(d.max())
        print(f"damping 1e-12, {name}: max |v - v_ref| = {d.max():.3e} rad/s at |v|max = {np.abs(ref).max():.3e}")
    assert max(worst.values()) <= 1e-6, worst


def test_golden_mobile_kinova(golden_dir):
    from mink_amd.engine import DeviceModel, DeviceProblem
    model = load_builtin("mobile_kinova")
    dm = DeviceModel(model, "cuda:0")
    g = np.load(os.path.join(golden_dir, "mobile_kinova_build_ik.npz"))
    prob = DeviceProblem(dm, W.task_specs(model, True), W.limit_specs(model, True))
    assert prob.fused
    v, st = prob.solve(dev(g["q"]), dev(g["target"]), float(g["dt"]), 1e-3)
    assert int(st.abs().sum()) == 0
    np.testing.assert_allclose(v.cpu().numpy(), g["v"], rtol=0, atol=V_TOL)


def test_golden_caller_loop(env, golden_dir):
    from mink_amd.engine import DeviceProblem
    model, dm = env
    g = np.load(os.path.join(golden_dir, "tidybot_loop.npz"))
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))
    q = dev(g["q0"])
    res = prob.iterate_(q, dev(g["target"]), float(g["dt"]), 1e-3, 20, 1e-4, 1e-4, want_v=True)
    iters = res["iters"].cpu().numpy()
    np.testing.assert_array_equal(iters, g["iters"])
    for k, n in enumerate(iters):
        np.testing.assert_allclose(q[k].cpu().numpy(), g["q"][k, n - 1], rtol=0, atol=1e-11)
        np.testing.assert_allclose(res["err"][k].cpu().numpy(), g["err"][k, n - 1], rtol=0, atol=1e-10)
        np.testing.assert_allclose(res["v"][k].cpu().numpy(), g["v"][k, n - 1], rtol=0, atol=1e-8)


@pytest.mark.parametrize("damping_task,velocity,seed", [(True, True, 1), (False, False, 0), (False, True, 2)])
def test_random_batch_vs_oracle(env, damping_task, velocity, seed):
    from mink_amd.engine import DeviceProblem
    from oracle import ik_oracle as o
    model, dm = env
    rng = np.random.default_rng(seed)
    B = 200
    q = W.sample_q(model, rng, B)
    qs = W.sample_q(model, rng, B)
    site = model.site_id("pinch_site")
    targets = dm.frame_pose(dev(qs), 2, site).cpu().numpy()
    # second half: nearby targets (few active limits); first half far targets (many)
    near = W.offset_targets(dm.frame_pose(dev(q), 2, site).cpu().numpy(), rng)
    targets[B // 2:] = near[B // 2:]
    prob = DeviceProblem(dm, W.task_specs(model, damping_task), W.limit_specs(model, velocity))
    v, st = prob.solve(dev(q), dev(targets), W.DT, W.DAMPING)
    v = v.cpu().numpy()
    assert int(st.abs().sum()) == 0
    worst = 0.0
    n_active = 0
    for b in range(B):
        tasks, limits = oracle_tasks_limits(model, targets[b], damping_task, velocity)
        cfg = o.Configuration(model, q[b])
        P, c, G, h = o.build_ik(cfg, tasks, W.DT, W.DAMPING, limits)
        x, status, active, _ = o.solve_qp(P, c, G, h)
        assert status == 0
        n_active += len(active)
        worst = max(worst, float(np.abs(x / W.DT - v[b]).max()))
    assert worst < V_TOL, worst
    if velocity:
        assert n_active > B  # the batch really exercises the active-set path


def test_batch_position_independence(env):
    """A problem's answer must not depend on where it sits in the batch (ragged tails)."""
    from mink_amd.engine import DeviceProblem
    model, dm = env
    rng = np.random.default_rng(5)
    B = 4096 + 37
    q = W.sample_q(model, rng, B)
    targets = W.offset_targets(dm.frame_pose(dev(q), 2, 0).cpu().numpy(), rng)
    prob = DeviceProblem(dm, W.task_specs(model, True), W.limit_specs(model, True))
    v_all, st = prob.solve(dev(q), dev(targets), W.DT, W.DAMPING)
    assert int(st.abs().sum()) == 0
    for lo, n in ((0, 1), (5, 63), (64, 65), (4000, 133)):
        v_sub, _ = prob.solve(dev(q[lo:lo + n]), dev(targets[lo:lo + n]), W.DT, W.DAMPING)
        assert torch.equal(v_sub, v_all[lo:lo + n])


def test_iterate_fixed_steps_vs_oracle(env):
    from mink_amd.engine import DeviceProblem
    from oracle import ik_oracle as o
    model, dm = env
    rng = np.random.default_rng(2)
    B, K = 12, 50
    q0 = W.sample_q(model, rng, B)
    targets = dm.frame_pose(dev(W.sample_q(model, rng, B)), 2, 0).cpu().numpy()
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    q = dev(q0)
    res = prob.iterate_(q, dev(targets), W.DT, W.DAMPING, K)
    assert int(res["status"].abs().sum()) == 0
    assert (res["iters"].cpu().numpy() == K).all()
    for b in range(B):
        tasks, limits = oracle_tasks_limits(model, targets[b], False, True)
        cfg = o.Configuration(model, q0[b])
        for _ in range(K):
            cfg.integrate_inplace(o.solve_ik(cfg, tasks, W.DT, damping=W.DAMPING, limits=limits), W.DT)
        np.testing.assert_allclose(q[b].cpu().numpy(), cfg.q, rtol=0, atol=1e-9)
        np.testing.assert_allclose(res["err"][b].cpu().numpy(), tasks[0].error(cfg), rtol=0, atol=1e-9)


def test_iterate_matches_repeated_single_steps(env):
    """Property at a larger size: K fused iterations == K x (solve, integrate) launches."""
    from mink_amd.engine import DeviceProblem
    model, dm = env
    rng = np.random.default_rng(8)
    B, K = 8192, 10
    q0 = W.sample_q(model, rng, B)
    targets = dm.frame_pose(dev(W.sample_q(model, rng, B)), 2, 0)
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    qa = dev(q0)
    prob.iterate_(qa, targets, W.DT, W.DAMPING, K)
    qb = dev(q0)
    for _ in range(K):
        v, _ = prob.solve(qb, targets, W.DT, W.DAMPING)
        dm.integrate_(qb, v, W.DT)
    assert float((qa - qb).abs().max()) < 1e-12


def test_fk_jacobian_integrate_limits(env):
    from oracle import ik_oracle as o
    from oracle import mjlite as mj
    model, dm = env
    rng = np.random.default_rng(3)
    B = 64
    q = W.sample_q(model, rng, B)
    q[:, 10:18] = rng.uniform(0.0, 0.3, (B, 8)) * np.sign(model.jnt_range[10:18, 1] + model.jnt_range[10:18, 0])
    q[:, 18:21] = rng.uniform(-1, 1, (B, 3))
    qq = rng.normal(size=(B, 4))
    q[:, 21:25] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    dq = dev(q)
    frames = [("site", "pinch_site", model.site_id("pinch_site")), ("body", "left_pad", model.body_id("left_pad")),
              ("body", "interactive_obj", model.body_id("interactive_obj")),
              ("geom", "right_pad1", model.geom_id("right_pad1")), ("body", "base_link", 1)]
    for ftype, name, fid in frames:
        pose = dm.frame_pose(dq, L.FRAME_TYPES[ftype], fid).cpu().numpy()
        jac = dm.frame_jacobian(dq, L.FRAME_TYPES[ftype], fid).cpu().numpy()
        for b in range(0, B, 7):
            cfg = o.Configuration(model, q[b])
            ref = cfg.get_transform_frame_to_world(name, ftype)
            s = 1.0 if ref[:4] @ pose[b, :4] > 0 else -1.0
            np.testing.assert_allclose(pose[b, :4] * s, ref[:4], rtol=0, atol=1e-13)
            np.testing.assert_allclose(pose[b, 4:], ref[4:], rtol=0, atol=1e-13)
            np.testing.assert_allclose(jac[b], cfg.get_frame_jacobian(name, ftype), rtol=0, atol=1e-12)
    v = rng.normal(size=(B, model.nv))
    qi = dev(q)
    dm.integrate_(qi, dev(v), 0.01)
    for b in range(0, B, 5):
        ref = q[b].copy()
        mj.mj_integratePos(model, ref, v[b], 0.01)
        np.testing.assert_allclose(qi[b].cpu().numpy(), ref, rtol=0, atol=1e-14)
    qbad = q.copy()
    qbad[3, 4] = 2.24 + 2e-6
    qbad[9, 6] = -2.57 - 1e-3
    qbad[9, 8] = 5.0
    qbad[11, 4] = 2.24 + 5e-7   # inside the 1e-6 tolerance
    first = dm.first_limit_violation(dev(qbad), 1e-6).cpu().numpy()
    expect = np.full(B, -1)
    expect[3], expect[9] = 4, 6
    np.testing.assert_array_equal(first, expect)


def test_infeasible_and_bad_arguments(env):
    from mink_amd.engine import DeviceProblem
    model, dm = env
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    q = np.tile(W.home_q(model), (3, 1))
    q[1, 4] = 2.24 + 0.5      # far outside the joint range: cfg and velocity boxes do not intersect
    tg = dm.frame_pose(dev(q), 2, 0)
    v, st = prob.solve(dev(q), tg, W.DT, W.DAMPING)
    assert st.cpu().tolist() == [0, 2, 0]
    qn = q.copy()
    qn[2, 5] = np.nan                       # non-finite input: reported, velocity zeroed on the chain
    v, st = prob.solve(dev(qn), tg, W.DT, W.DAMPING)
    assert st.cpu().tolist() == [0, 2, 3]
    tn = tg.clone()
    tn[0, 4] = float("inf")
    v, st = prob.solve(dev(q), tn, W.DT, W.DAMPING)
    assert st.cpu().tolist() == [3, 2, 0]
    with pytest.raises(L.MinkAmdError):
        prob.solve(dev(q), tg, -1.0, W.DAMPING)
    with pytest.raises(ValueError):
        prob.solve(dev(q)[:, :10].contiguous(), tg, W.DT, W.DAMPING)
    v, st = prob.solve(dev(q)[:0], tg[:0], W.DT, W.DAMPING)   # empty batch
    assert v.shape == (0, model.nv)


def test_cabinet_scene_model():
    """tidybot_open (the reference's open.xml, nq = nv = 20): single step and the iterated
    loop on the GPU against the oracle."""
    from mink_amd.engine import DeviceModel, DeviceProblem
    from oracle import ik_oracle as o
    model = load_builtin("tidybot_open")
    dm = DeviceModel(model, "cuda:0")
    rng = np.random.default_rng(9)
    B, K = 70, 4
    q, qs = W.sample_q(model, rng, B), W.sample_q(model, rng, B)
    q[:, 18:] = rng.uniform(0.0, 0.5, (B, 2))
    targets = np.array([o.Configuration(model, x).get_transform_frame_to_world("pinch_site", "site") for x in qs])
    prob = DeviceProblem(dm, W.task_specs(model, True), W.limit_specs(model, True))
    assert prob.fused
    v, st = prob.solve(dev(q), dev(targets), W.DT, W.DAMPING)
    assert int(st.abs().sum()) == 0
    qi = dev(q)
    res = prob.iterate_(qi, dev(targets), W.DT, W.DAMPING, K)
    assert int(res["status"].abs().sum()) == 0
    olimits = [o.ConfigurationLimit(model), o.VelocityLimit(model, W.VELOCITY_LIMITS)]
    for b in range(0, B, 7):
        tasks = [o.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0, target=targets[b]),
                 o.PostureTask(cost=W.posture_cost(model), target_q=W.home_q(model)),
                 o.DampingTask(model, W.base_damping_cost(model))]
        cfg = o.Configuration(model, q[b])
        np.testing.assert_allclose(v[b].cpu().numpy(), o.solve_ik(cfg, tasks, W.DT, damping=W.DAMPING, limits=olimits),
                                   rtol=0, atol=V_TOL)
        for _ in range(K):
            cfg.integrate_inplace(o.solve_ik(cfg, tasks, W.DT, damping=W.DAMPING, limits=olimits), W.DT)
        np.testing.assert_allclose(qi[b].cpu().numpy(), cfg.q, rtol=0, atol=1e-10)
