"""The cooperative kernels (one problem per 16-lane row, mk_coop.hip) against the
lane-per-problem kernels (mk_fused.hip) and the C oracle: both forms of the fused path must
give the reference's solve_ik (mink/solve_ik.py:68-105) and the callers' loop
(interactive_scripts/teleop_mouse.py:105-121).  Tolerance |dv| <= 1e-9 rad/s (stated bound
1e-6); statuses and solve counts equal."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.mjcf import load_builtin

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")


@pytest.fixture(scope="module")
def env():
    from mink_amd.engine import DeviceModel
    model = load_builtin("tidybot_cube")
    return model, DeviceModel(model, "cuda:0")


def both(fn):
    out = []
    for form in (0, 1):
        with L.options(coop=form):
            out.append(fn())
    return out


@pytest.mark.parametrize("B", [1, 2, 3, 5, 63, 64, 65, 1027, 4096])
def test_single_step_any_batch_size(env, B):
    """Rows without a problem (B not a multiple of 4) and single problems."""
    from mink_amd.engine import DeviceProblem
    model, dm = env
    rng = np.random.default_rng(100 + B)
    q = dev(W.sample_q(model, rng, B))
    far = dm.frame_pose(dev(W.sample_q(model, rng, B)), 2, 0)
    prob = DeviceProblem(dm, W.task_specs(model, True), W.limit_specs(model, True))
    (vl, sl), (vc, sc) = both(lambda: prob.solve(q, far, W.DT, W.DAMPING))
    assert torch.equal(sl, sc) and int(sl.abs().sum()) == 0
    assert float((vl - vc).abs().max()) <= 1e-9


def test_saturated_steps_vs_c_oracle(env):
    """Far targets: velocity limits bind on most dofs, the active-set rounds and the dual steps run."""
    from mink_amd.engine import DeviceProblem
    from oracle import c_oracle as co
    model, dm = env
    rng = np.random.default_rng(31)
    B = 8192
    q = W.sample_q(model, rng, B)
    targets = dm.frame_pose(dev(W.sample_q(model, rng, B)), 2, 0)
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    with L.options(coop=1):
        v, st = prob.solve(dev(q), targets, W.DT, W.DAMPING)
    assert int(st.abs().sum()) == 0
    orc = co.COracle(model, postures=[dict(cost=W.posture_cost(model), target_q=W.home_q(model))],
                     velocity=W.VELOCITY_LIMITS)
    v_ref, st_ref = orc.solve_ik(q, targets.cpu().numpy(), W.DT, W.DAMPING, nthreads=8)
    assert not st_ref.any()
    d = np.abs(v.cpu().numpy() - v_ref).max()
    print("coop vs C oracle, saturated steps: max |dv| =", d)
    assert d <= 1e-9


@pytest.mark.parametrize("max_iters,tol", [(20, 1e-4), (50, -1.0), (0, 1e-4), (1, 1e-4)])
def test_loop_matches_lane_kernels(env, max_iters, tol):
    from mink_amd.engine import DeviceProblem
    model, dm = env
    rng = np.random.default_rng(41)
    B = 2051
    q0 = W.sample_q(model, rng, B)
    near = W.offset_targets(dm.frame_pose(dev(q0), 2, 0).cpu().numpy(), rng)
    far = dm.frame_pose(dev(W.sample_q(model, rng, B)), 2, 0).cpu().numpy()
    targets = dev(np.where((np.arange(B) % 3 == 0)[:, None], far, near))
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))

    def run():
        q = dev(q0)
        r = prob.iterate_(q, targets, W.DT, W.DAMPING, max_iters, tol, tol, want_v=True)
        return q, r

    (ql, rl), (qc, rc) = both(run)
    assert torch.equal(rl["iters"], rc["iters"])
    assert torch.equal(rl["status"], rc["status"])
    assert float((ql - qc).abs().max()) <= 1e-10
    assert float((rl["err"] - rc["err"]).abs().max()) <= 1e-10
    if max_iters > 0:
        # (the velocity of the last solve: after 50 steps the task is fulfilled, the Hessian is
        # down to the posture cost and damping, and v = dq / dt carries the conditioning)
        assert float((rl["v"] - rc["v"]).abs().max()) <= 1e-7


def test_loop_vs_c_oracle(env):
    from mink_amd.engine import DeviceProblem
    from oracle import c_oracle as co
    model, dm = env
    rng = np.random.default_rng(43)
    B = 4096
    q0 = W.sample_q(model, rng, B)
    targets = W.offset_targets(dm.frame_pose(dev(q0), 2, 0).cpu().numpy(), rng)
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))
    q = dev(q0)
    with L.options(coop=1):
        r = prob.iterate_(q, dev(targets), W.DT, W.DAMPING, 20, 1e-4, 1e-4)
    orc = co.COracle(model, postures=[dict(cost=W.posture_cost(model), target_q=W.home_q(model))])
    q_ref, it_ref, err_ref, st_ref = orc.iterate(q0, targets, W.DT, W.DAMPING, 20, 1e-4, 1e-4, nthreads=8)
    np.testing.assert_array_equal(r["iters"].cpu().numpy(), it_ref)
    assert np.abs(q.cpu().numpy() - q_ref).max() <= 1e-10
    assert (it_ref < 20).mean() > 0.9


def test_status_codes_match(env):
    from mink_amd.engine import DeviceProblem
    model, dm = env
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    q = np.tile(W.home_q(model), (6, 1))
    q[1, 4] = 2.24 + 0.5           # infeasible box
    q[2, 5] = np.nan               # chain coordinate
    q[3, 12] = np.inf              # a finger joint outside the chain
    q[4, 20] = np.nan              # the cube's free joint
    tg = dm.frame_pose(dev(np.tile(W.home_q(model), (6, 1))), 2, 0)
    tg[5, 2] = float("nan")
    (vl, sl), (vc, sc) = both(lambda: prob.solve(dev(q), tg, W.DT, W.DAMPING))
    assert sl.cpu().tolist() == [0, 2, 3, 3, 3, 3]
    assert torch.equal(sl, sc)
    assert torch.equal(vl[sl == 3], vc[sc == 3])          # zero velocity
    assert float((vl[0] - vc[0]).abs().max()) <= 1e-9


@pytest.mark.parametrize("name,frame,ftype", [("mobile_kinova", "pinch_site", "site"),
                                              ("mobile_kinova", "forearm_link", "body"),
                                              ("tidybot_open", "pinch_site", "site")])
def test_other_models_and_chain_lengths(name, frame, ftype):
    """Whole model = chain (no other dofs), a 7-slot chain with padding and leftover arm dofs,
    and the cabinet scene (two door hinges outside the chain)."""
    from mink_amd.engine import DeviceModel, DeviceProblem
    model = load_builtin(name)
    dm = DeviceModel(model, "cuda:0")
    rng = np.random.default_rng(7)
    B = 777
    q = dev(W.sample_q(model, rng, B))
    fid = model.site_id(frame) if ftype == "site" else model.body_id(frame)
    targets = dm.frame_pose(dev(W.sample_q(model, rng, B)), L.FRAME_TYPES[ftype], fid)
    specs = W.task_specs(model, True)
    specs[0].update(frame_type=L.FRAME_TYPES[ftype], frame_id=fid)
    prob = DeviceProblem(dm, specs, W.limit_specs(model, True))
    assert prob.fused
    (vl, sl), (vc, sc) = both(lambda: prob.solve(q, targets, W.DT, W.DAMPING))
    assert torch.equal(sl, sc) and int(sl.abs().sum()) == 0
    assert float((vl - vc).abs().max()) <= 1e-9

    def run():
        qq = q.clone()
        r = prob.iterate_(qq, targets, W.DT, W.DAMPING, 8, 1e-4, 1e-4)
        return qq, r

    (ql, rl), (qc, rc) = both(run)
    assert torch.equal(rl["iters"], rc["iters"])
    assert float((ql - qc).abs().max()) <= 1e-10


def test_default_is_by_batch_size(env):
    """Without the option the library takes the cooperative kernels up to coop_max_batch and
    the lane kernels beyond: results agree across the switch-over."""
    from mink_amd.engine import DeviceProblem
    model, dm = env
    assert L.get_option("coop") == -1
    nmax = L.get_option("coop_max_batch")
    rng = np.random.default_rng(3)
    B = nmax + 4
    q = dev(W.sample_q(model, rng, B))
    targets = dm.frame_pose(dev(W.sample_q(model, rng, B)), 2, 0)
    prob = DeviceProblem(dm, W.task_specs(model, True), W.limit_specs(model, True))
    v_big, _ = prob.solve(q, targets, W.DT, W.DAMPING)               # lane kernels
    v_small, _ = prob.solve(q[:nmax].contiguous(), targets[:nmax].contiguous(), W.DT, W.DAMPING)   # cooperative
    assert float((v_big[:nmax] - v_small).abs().max()) <= 1e-9
    with L.options(coop=0):
        v_lane, _ = prob.solve(q[:nmax].contiguous(), targets[:nmax].contiguous(), W.DT, W.DAMPING)
    assert torch.equal(v_lane, v_big[:nmax])


@pytest.mark.parametrize("B,handoff,stream", [(65536, -1, 0), (20000, 9, 0), (65536 + 77, 1, 0), (40000, -1, 96),
                                              (300000, -1, -1)])
def test_wave_handoff_to_cooperative_kernel(env, B, handoff, stream):
    """Loops with an exit test on batches beyond the cooperative kernels' size: the lane kernel
    stops a wave when few of its problems are left and the cooperative kernel finishes those.
    Same solve counts and configurations as the lane kernel alone.  stream: the block-per-64
    kernel (0), the lane-streaming kernel with that many persistent waves, or the library's choice
    (-1; streaming at 300000 problems) -- the streaming kernel hands over the tail of each chunk."""
    from mink_amd.engine import DeviceProblem
    model, dm = env
    rng = np.random.default_rng(77)
    q0 = W.sample_q(model, rng, B)
    targets = dev(W.offset_targets(dm.frame_pose(dev(q0), 2, 0).cpu().numpy(), rng))
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))

    def run(**opts):
        with L.options(**opts):
            q = dev(q0)
            r = prob.iterate_(q, targets, W.DT, W.DAMPING, 20, 1e-4, 1e-4, want_v=True)
        return q, r

    ql, rl = run(coop=0, stream_blocks=stream)
    qh, rh = run(handoff=handoff, stream_blocks=stream)
    assert int(rh["iters"].max()) <= 20 and int(rh["iters"].min()) >= 1      # no mark left behind
    assert torch.equal(rl["iters"], rh["iters"])
    assert torch.equal(rl["status"], rh["status"])
    assert float((ql - qh).abs().max()) <= 1e-10
    assert float((rl["err"] - rh["err"]).abs().max()) <= 1e-10
    assert float((rl["v"] - rh["v"]).abs().max()) <= 1e-8
    # and the hand-off did happen: with it switched off the launch is the lane kernel's, bit for bit
    q0_, r0_ = run(handoff=0, stream_blocks=stream)
    assert torch.equal(q0_, ql)
