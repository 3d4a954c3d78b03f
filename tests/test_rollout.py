"""The waypoint-reaching harness against a restatement of the reference's own code
(envs/robot_utils.py:131-160 with scipy's Rotation / Slerp, envs/mj_env.py:209-276) that
drives the numpy oracle one problem at a time."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
from scipy.spatial.transform import Rotation, Slerp

from mink_amd.lie import SO3
from mink_amd.rollout import LinearWaypointReach, LinearWaypointReachConfig

CFG = LinearWaypointReachConfig(pos_threshold=0.01, pos_step_size=0.08, rot_threshold=0.05, rot_step_size=0.3)


def ref_step(target_pos, target_euler, curr_pos, curr_euler, cfg):
    """LinearWaypointReach.step of the reference, verbatim in behaviour."""
    target_rot = Rotation.from_euler("xyz", target_euler)
    delta = target_pos - curr_pos
    pos_reached = np.linalg.norm(delta) < cfg.pos_threshold
    if pos_reached:
        abs_pos = target_pos
    else:
        abs_pos = curr_pos + delta / np.linalg.norm(delta) * min(cfg.pos_step_size, np.linalg.norm(delta))
    curr_rot = Rotation.from_euler("xyz", curr_euler)
    half = Slerp([0, 1], Rotation.concatenate([curr_rot, target_rot]))(0.5)
    rot_reached = np.linalg.norm((target_rot * curr_rot.inv()).as_euler("xyz")) < cfg.rot_threshold
    return abs_pos, half, bool(pos_reached and rot_reached)


def test_waypoint_step_matches_reference_code():
    rng = np.random.default_rng(0)
    B = 60
    tp, te = rng.uniform(-0.5, 0.5, (B, 3)), rng.uniform(-1.5, 1.5, (B, 3))
    cp = tp + rng.uniform(-0.2, 0.2, (B, 3)) * (rng.random((B, 1)) < 0.8)
    ce = te + rng.uniform(-0.6, 0.6, (B, 3)) * (rng.random((B, 1)) < 0.8)
    cp[:5], ce[:5] = tp[:5] + 1e-3, te[:5] + 1e-3
    e = torch.tensor(ce)
    p, r, reached = LinearWaypointReach.from_euler(tp, te, CFG).step(
        torch.tensor(cp), SO3.from_rpy_radians(e[:, 0], e[:, 1], e[:, 2]))
    assert 0 < int(reached.sum()) < B
    for b in range(B):
        ap, half, done = ref_step(tp[b], te[b], cp[b], ce[b], CFG)
        np.testing.assert_allclose(p[b].numpy(), ap, rtol=0, atol=1e-14)
        qs, qm = half.as_quat()[[3, 0, 1, 2]], r.wxyz[b].numpy()
        np.testing.assert_allclose(qm * np.sign(qm @ qs), qs, rtol=0, atol=1e-13)
        assert bool(reached[b]) == done


def test_device_waypoint_step_matches_reference_code():
    """The waypoint generator of the one-launch kernel (host build of the same device
    function) against the reference's step() with scipy's Rotation / Slerp."""
    import hostsim_lib as hs
    if not hs.available():
        pytest.skip("host build of the device code is missing")
    rng = np.random.default_rng(3)
    B = 80
    tp, te = rng.uniform(-0.5, 0.5, (B, 3)), rng.uniform(-1.5, 1.5, (B, 3))
    cp = tp + rng.uniform(-0.2, 0.2, (B, 3)) * (rng.random((B, 1)) < 0.8)
    ce = te + rng.uniform(-0.6, 0.6, (B, 3)) * (rng.random((B, 1)) < 0.8)
    cp[:5], ce[:5] = tp[:5] + 1e-3, te[:5] + 1e-3
    cp[5], ce[5] = tp[5], te[5]                                   # exactly on the goal
    wxyz = lambda e: Rotation.from_euler("xyz", e).as_quat()[:, [3, 0, 1, 2]]
    cmd, reached = hs.waypoint_step(np.hstack([wxyz(ce), cp]), np.hstack([wxyz(te), tp]), CFG.pos_threshold,
                                    CFG.pos_step_size, CFG.rot_threshold)
    assert 0 < int(reached.sum()) < B
    for b in range(B):
        ap, half, done = ref_step(tp[b], te[b], cp[b], ce[b], CFG)
        np.testing.assert_allclose(cmd[b, 4:], ap, rtol=0, atol=1e-14)
        qs = half.as_quat()[[3, 0, 1, 2]]
        np.testing.assert_allclose(cmd[b, :4] * np.sign(cmd[b, :4] @ qs), qs, rtol=0, atol=1e-13)
        assert bool(reached[b]) == done


@pytest.fixture(params=["lane", "coop"])
def kernel_form(request):
    """Both forms of the one-launch kernels: one rollout per lane (mk_fused.hip) and one per
    16-lane row (mk_coop.hip)."""
    from mink_amd import _lib as L
    with L.options(coop=1 if request.param == "coop" else 0):
        yield request.param


@pytest.mark.gpu
def test_move_to_matches_sequential_reference_loop(kernel_form):
    import mink_amd as mink
    from mink_amd import workloads as W
    from oracle import ik_oracle as o
    model = mink.load_builtin("tidybot_cube")
    rng = np.random.default_rng(1)
    B, max_steps = 6, 12
    q0 = W.sample_q(model, rng, B)
    cfg = mink.Configuration(model, q0)
    start = cfg.get_transform_frame_to_world("pinch_site", "site")
    goal_pos = (start.translation() + torch.tensor(rng.uniform(-0.15, 0.15, (B, 3)), device="cuda:0"))
    goal_rot = start.rotation() @ SO3.exp(torch.tensor(rng.uniform(-0.4, 0.4, (B, 3)), device="cuda:0"))
    goal_pos[0] = start.translation()[0] + 0.004          # reached at once
    goal_rot.wxyz[0] = start.rotation().wxyz[0]
    frame = mink.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
    posture = mink.PostureTask(model, cost=W.posture_cost(model))
    posture.set_target(W.home_q(model))
    out = mink.move_to(cfg, frame, [frame, posture], goal_pos, goal_rot, CFG, max_steps=max_steps)
    steps = out["steps"].cpu().numpy()
    assert steps[0] == 1 and bool(out["reached"][0])
    # the same thing one problem at a time: reference waypoint code + numpy oracle IK
    gp, gq = goal_pos.cpu().numpy(), goal_rot.wxyz.cpu().numpy()
    for b in range(B):
        ocfg = o.Configuration(model, q0[b])
        goal_euler = Rotation.from_quat(gq[b][[1, 2, 3, 0]]).as_euler("xyz")
        n_steps, n_solves, reached = 0, 0, False
        for _ in range(max_steps):
            T = ocfg.get_transform_frame_to_world("pinch_site", "site")
            eul = Rotation.from_quat(T[[1, 2, 3, 0]]).as_euler("xyz")
            pos_cmd, half, reached = ref_step(gp[b], goal_euler, T[4:], eul, CFG)
            target = np.concatenate([half.as_quat()[[3, 0, 1, 2]], pos_cmd])
            tasks = [o.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0, target=target),
                     o.PostureTask(cost=W.posture_cost(model), target_q=W.home_q(model))]
            for _ in range(20):
                v = o.solve_ik(ocfg, tasks, 1 / 200, damping=1e-3)
                ocfg.integrate_inplace(v, 1 / 200)
                n_solves += 1
                err = tasks[0].error(ocfg)
                if np.linalg.norm(err[:3]) <= 1e-4 and np.linalg.norm(err[3:]) <= 1e-4:
                    break
            n_steps += 1
            if reached:
                break
        assert steps[b] == n_steps and bool(out["reached"][b]) == reached
        assert int(out["solves"][b]) == n_solves
        np.testing.assert_allclose(cfg.q[b].cpu().numpy(), ocfg.q, rtol=0, atol=1e-8)


@pytest.mark.gpu
def test_move_to_single_launch_matches_stepwise(kernel_form):
    """The one-launch form (mk_move_to_batch) against the host loop over outer steps, on a
    batch large enough to have ragged blocks and every outcome (reached early, late, not)."""
    import mink_amd as mink
    from mink_amd import rollout
    from mink_amd import workloads as W
    model = mink.load_builtin("tidybot_cube")
    rng = np.random.default_rng(7)
    B, max_steps = 1000, 9
    q0 = W.sample_q(model, rng, B)
    frame = mink.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
    posture = mink.PostureTask(model, cost=W.posture_cost(model))
    posture.set_target(W.home_q(model))
    limits = [mink.ConfigurationLimit(model), mink.VelocityLimit(model, W.VELOCITY_LIMITS)]
    cfg_a, cfg_b = mink.Configuration(model, q0), mink.Configuration(model, q0)
    start = cfg_a.get_transform_frame_to_world("pinch_site", "site")
    goal_pos = start.translation() + torch.tensor(rng.uniform(-0.25, 0.25, (B, 3)), device="cuda:0")
    goal_rot = start.rotation() @ SO3.exp(torch.tensor(rng.uniform(-0.6, 0.6, (B, 3)), device="cuda:0"))
    a = mink.move_to(cfg_a, frame, [frame, posture], goal_pos, goal_rot, CFG, max_steps=max_steps, limits=limits)
    b = rollout._move_to_stepwise(cfg_b, frame, [frame, posture], goal_pos, goal_rot, CFG, 1.0 / 200.0, "quadprog",
                                  1e-3, max_steps, 20, 1e-4, 1e-4, limits)
    assert torch.equal(a["steps"], b["steps"]) and torch.equal(a["reached"], b["reached"])
    assert torch.equal(a["solves"], b["solves"])
    r = a["reached"]
    assert 0 < int(r.sum()) < B and int(a["steps"].min()) < max_steps
    np.testing.assert_allclose(cfg_a.q.cpu().numpy(), cfg_b.q.cpu().numpy(), rtol=0, atol=1e-9)


def _config4_rollout_inputs(B, seed):
    """Benchmark config 4 as objects of the host mirror: Frame + RelativeFrame + Posture tasks,
    Configuration + Velocity + CollisionAvoidance limits, the cube next to a gripper pad in a
    quarter of the problems."""
    import mink_amd as mink
    from mink_amd import workloads as W
    from mink_amd.engine import DeviceModel
    model = mink.load_builtin("tidybot_cube")
    dm = DeviceModel(model, "cuda:0")
    q0, _ = W.config4_inputs(model, dm, B, seed)
    q0 = q0.cpu().numpy()
    frame = mink.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
    rel = mink.RelativeFrameTask("pinch_site", "site", "base_link", "body", 0.3, 0.3)
    rel.set_target_from_configuration(mink.Configuration(model, q0))
    posture = mink.PostureTask(model, cost=W.posture_cost(model))
    posture.set_target(W.home_q(model))
    limits = [mink.ConfigurationLimit(model), mink.VelocityLimit(model, W.VELOCITY_LIMITS),
              mink.CollisionAvoidanceLimit(model, [(list(W.PADS), ["cube_geom", "floor"])])]
    return model, q0, frame, rel, posture, limits


@pytest.mark.gpu
def test_move_to_passes_for_a_staged_problem_matches_stepwise():
    """A problem outside the fused path (collision limits, a relative frame task): mink.move_to queues
    the whole rollout through mk_move_to_passes_batch; same counts and configurations as the host
    loop over outer steps, which issues the same kernels one step at a time."""
    import mink_amd as mink
    from mink_amd import rollout
    from mink_amd._problems import get_problem
    B, max_steps = 700, 7
    model, q0, frame, rel, posture, limits = _config4_rollout_inputs(B, 21)
    cfg_a, cfg_b = mink.Configuration(model, q0), mink.Configuration(model, q0)
    rng = np.random.default_rng(22)
    start = cfg_a.get_transform_frame_to_world("pinch_site", "site")
    goal_pos = start.translation() + torch.tensor(rng.uniform(-0.2, 0.2, (B, 3)), device="cuda:0")
    goal_rot = start.rotation() @ SO3.exp(torch.tensor(rng.uniform(-0.5, 0.5, (B, 3)), device="cuda:0"))
    goal_pos[:20] = start.translation()[:20] + 0.003            # reached in the first step
    goal_rot.wxyz[:20] = start.rotation().wxyz[:20]
    frame.set_target(start)
    prob, _ = get_problem(cfg_a, [frame, rel, posture], limits)
    assert prob.staged and not prob.fused
    with cfg_a.deferred_checks(), cfg_b.deferred_checks():
        a = mink.move_to(cfg_a, frame, [frame, rel, posture], goal_pos, goal_rot, CFG, max_steps=max_steps, limits=limits)
        b = rollout._move_to_stepwise(cfg_b, frame, [frame, rel, posture], goal_pos, goal_rot, CFG, 1.0 / 200.0, "quadprog",
                                      1e-3, max_steps, 20, 1e-4, 1e-4, limits)
    assert torch.equal(a["steps"], b["steps"]) and torch.equal(a["reached"], b["reached"])
    assert torch.equal(a["solves"], b["solves"])
    assert int(a["steps"][:20].max()) == 1 and bool(a["reached"][:20].all())
    assert 20 < int(a["reached"].sum()) < B
    # (the waypoint step runs on the device in one, in torch in the other: rounding-level differences)
    np.testing.assert_allclose(cfg_a.q.cpu().numpy(), cfg_b.q.cpu().numpy(), rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_move_to_passes_matches_sequential_reference_loop():
    """The same entry against the reference's own loop (envs/mj_env.py:209-276: waypoint step, then up
    to 20 x solve_ik + integrate with the exit test) driving the numpy oracle one problem at a time,
    collision rows and relative frame task included."""
    import mink_amd as mink
    from mink_amd import workloads as W
    from oracle import ik_oracle as o
    B, max_steps = 8, 4
    model, q0, frame, rel, posture, limits = _config4_rollout_inputs(B, 23)
    cfg = mink.Configuration(model, q0)
    rng = np.random.default_rng(24)
    start = cfg.get_transform_frame_to_world("pinch_site", "site")
    goal_pos = start.translation() + torch.tensor(rng.uniform(-0.1, 0.1, (B, 3)), device="cuda:0")
    goal_rot = start.rotation() @ SO3.exp(torch.tensor(rng.uniform(-0.3, 0.3, (B, 3)), device="cuda:0"))
    rel_targets = rel.transform_target_to_root.wxyz_xyz.cpu().numpy()
    with cfg.deferred_checks():
        out = mink.move_to(cfg, frame, [frame, rel, posture], goal_pos, goal_rot, CFG, max_steps=max_steps, limits=limits)
    gp, gq = goal_pos.cpu().numpy(), goal_rot.wxyz.cpu().numpy()
    olimits = [o.ConfigurationLimit(model), o.VelocityLimit(model, W.VELOCITY_LIMITS),
               o.CollisionAvoidanceLimit(model, [(list(W.PADS), ["cube_geom", "floor"])])]
    n_rows_seen = 0
    for b in range(B):
        ocfg = o.Configuration(model, q0[b])
        goal_euler = Rotation.from_quat(gq[b][[1, 2, 3, 0]]).as_euler("xyz")
        n_steps, n_solves, reached = 0, 0, False
        for _ in range(max_steps):
            T = ocfg.get_transform_frame_to_world("pinch_site", "site")
            eul = Rotation.from_quat(T[[1, 2, 3, 0]]).as_euler("xyz")
            pos_cmd, half, reached = ref_step(gp[b], goal_euler, T[4:], eul, CFG)
            target = np.concatenate([half.as_quat()[[3, 0, 1, 2]], pos_cmd])
            tasks = [o.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0, target=target),
                     o.RelativeFrameTask("pinch_site", "site", "base_link", "body", 0.3, 0.3, target=rel_targets[b]),
                     o.PostureTask(cost=W.posture_cost(model), target_q=W.home_q(model))]
            for _ in range(20):
                n_rows_seen += int(np.isfinite(olimits[2].inequalities(ocfg, 1 / 200)[1]).sum())
                v = o.solve_ik(ocfg, tasks, 1 / 200, damping=1e-3, limits=olimits)
                ocfg.integrate_inplace(v, 1 / 200)
                n_solves += 1
                err = tasks[0].error(ocfg)
                if np.linalg.norm(err[:3]) <= 1e-4 and np.linalg.norm(err[3:]) <= 1e-4:
                    break
            n_steps += 1
            if reached:
                break
        assert int(out["steps"][b]) == n_steps and bool(out["reached"][b]) == reached
        assert int(out["solves"][b]) == n_solves
        np.testing.assert_allclose(cfg.q[b].cpu().numpy(), ocfg.q, rtol=0, atol=1e-7)
    assert n_rows_seen > 0          # collision rows took part


@pytest.mark.gpu
def test_move_to_tail_handoff_matches_lane_kernel():
    """Beyond the cooperative kernels' batch size the lane kernel hands the last rollouts of
    each wave to the cooperative kernel at an outer-step boundary: same step / solve counts and
    end configurations as the lane kernel alone (the two differ in rounding only; a rollout whose
    own dynamics amplify the last bits may take one solve more or less)."""
    import mink_amd as mink
    from mink_amd import _lib as L
    from mink_amd import workloads as W
    model = mink.load_builtin("tidybot_cube")
    rng = np.random.default_rng(11)
    B, max_steps = 16384 + 3000, 30
    q0 = W.sample_q(model, rng, B)
    frame = mink.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
    posture = mink.PostureTask(model, cost=W.posture_cost(model))
    posture.set_target(W.home_q(model))
    cfg_a, cfg_b = mink.Configuration(model, q0), mink.Configuration(model, q0)
    start = cfg_a.get_transform_frame_to_world("pinch_site", "site")
    goal_pos = start.translation() + torch.tensor(rng.uniform(-0.25, 0.25, (B, 3)), device="cuda:0")
    goal_rot = start.rotation() @ SO3.exp(torch.tensor(rng.uniform(-0.6, 0.6, (B, 3)), device="cuda:0"))
    with L.options(coop=0):
        a = mink.move_to(cfg_a, frame, [frame, posture], goal_pos, goal_rot, CFG, max_steps=max_steps)
    b = mink.move_to(cfg_b, frame, [frame, posture], goal_pos, goal_rot, CFG, max_steps=max_steps)
    assert int(b["steps"].max()) <= max_steps and int(b["steps"].min()) >= 1     # no mark left behind
    same = (a["steps"] == b["steps"]) & (a["solves"] == b["solves"]) & (a["reached"] == b["reached"])
    print("rollouts with equal counts:", int(same.sum()), "of", B, "; solves max", int(a["solves"].max()),
          "mean", float(a["solves"].double().mean()))
    assert float(same.double().mean()) >= 0.9995
    dq = (cfg_a.q - cfg_b.q).abs().amax(dim=1)[same]
    assert float(dq.max()) <= 1e-7
    assert int(a["solves"].max()) > 3 * float(a["solves"].double().mean())     # there is a tail to hand off


@pytest.mark.gpu
def test_move_to_passes_degenerate_counts():
    """No outer step, no IK iteration per step, and an empty batch through mk_move_to_passes_batch:
    the counters follow the reference's loops (envs/mj_env.py:212-230: a rollout whose generator
    reports the goal reached stops after that step, the others run out of steps)."""
    import mink_amd as mink
    from mink_amd._problems import get_problem
    from mink_amd.lie import SE3
    B = 40
    model, q0, frame, rel, posture, limits = _config4_rollout_inputs(B, 31)
    cfg = mink.Configuration(model, q0)
    start = cfg.get_transform_frame_to_world("pinch_site", "site")
    rng = np.random.default_rng(32)
    goal_pos = start.translation() + torch.tensor(rng.uniform(-0.2, 0.2, (B, 3)), device="cuda:0")
    goal_pos[:10] = start.translation()[:10] + 0.002
    goal_rot = start.rotation()
    frame.set_target(start)
    prob, targets = get_problem(cfg, [frame, rel, posture], limits)
    goals = SE3.from_rotation_and_translation(goal_rot, goal_pos).wxyz_xyz.contiguous()
    q_before = cfg.q_tensor.clone()
    r = prob.move_to_passes_(0, cfg.q_tensor, goals, targets.clone(), CFG, 1 / 200, 1e-3, 0, 20, 1e-4, 1e-4)
    assert int(r["steps"].abs().sum()) == 0 and int(r["reached"].sum()) == 0 and int(r["solves"].sum()) == 0
    assert torch.equal(cfg.q_tensor, q_before)
    r = prob.move_to_passes_(0, cfg.q_tensor, goals, targets.clone(), CFG, 1 / 200, 1e-3, 5, 0, 1e-4, 1e-4)
    assert torch.equal(cfg.q_tensor, q_before) and int(r["solves"].sum()) == 0
    assert r["steps"][:10].tolist() == [1] * 10 and bool(r["reached"][:10].all())
    assert r["steps"][10:].tolist() == [5] * (B - 10) and not bool(r["reached"][10:].any())
    empty = torch.empty((0, model.nq), dtype=torch.float64, device="cuda:0")
    r = prob.move_to_passes_(0, empty, goals[:0].contiguous(), targets[:0].contiguous().clone(), CFG, 1 / 200, 1e-3, 5, 5,
                             1e-4, 1e-4)
    assert r["steps"].numel() == 0
    with pytest.raises(mink._lib.MinkAmdError, match="frame task"):
        prob.move_to_passes_(2, cfg.q_tensor, goals, targets.clone(), CFG, 1 / 200, 1e-3, 5, 5, 1e-4, 1e-4)
