"""Generate the golden fixtures in this directory by running the UNMODIFIED
reference package (/root/reference/mink) in the build container.

``mujoco`` and ``qpsolvers`` are not installed, so the reference is imported on
top of the facades in oracle/ref_facade/ (kinematics and QP come from this
repository's restatements).  The fixtures therefore pin the reference's own
Python -- mink/lie, mink/tasks, mink/limits, mink/configuration.py,
mink/solve_ik.py -- and NOT MuJoCo or quadprog.

Run from the repository root:  python tests/golden/make_golden.py
Outputs (inputs + expected outputs only, no reference source):
  lie.npz, tidybot_build_ik.npz, tidybot_loop.npz, mobile_kinova_build_ik.npz,
  collision.npz, api_strings.json, limited_ball.npz
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("MINK_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_facade"))
sys.path.insert(1, REF)
sys.path.insert(2, ROOT)

import mujoco  # noqa: E402  (facade)
import mink  # noqa: E402  (the real reference package)

assert mink.__file__.startswith(REF), mink.__file__
SCENES = os.path.join(REF, "interactive_scripts", "stanford_tidybot")


# ------------------------------------------------------------------ lie ----
def random_unit_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def lie_inputs(rng):
    Ts = []
    for _ in range(40):
        Ts.append(np.concatenate([random_unit_quat(rng), rng.uniform(-1, 1, 3)]))
    # identity and near-identity (both sides of the theta^2 < 1e-10 switch)
    Ts.append(np.array([1.0, 0, 0, 0, 0, 0, 0]))
    for ang in (1e-9, 3e-6, 0.9e-5, 1.1e-5, 1e-4, 1e-3):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        q = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        Ts.append(np.concatenate([q, rng.uniform(-1, 1, 3)]))
    # rotations near pi, and quaternions with negative / vanishing w
    for ang in (np.pi - 1e-3, np.pi - 1e-7, np.pi, np.pi + 1e-6, 1.5 * np.pi, 2 * np.pi - 1e-3):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        q = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        Ts.append(np.concatenate([q, rng.uniform(-1, 1, 3)]))
    Ts.append(np.array([0.0, 1.0, 0, 0, 0.3, -0.2, 0.1]))
    Ts.append(np.array([-0.5, 0.5, -0.5, 0.5, 0.1, 0.2, 0.3]))
    return np.array(Ts)


def make_lie():
    rng = np.random.default_rng(100)
    T = lie_inputs(rng)
    n = len(T)
    T2 = np.array([np.concatenate([random_unit_quat(rng), rng.uniform(-1, 1, 3)]) for _ in range(n)])
    xi = rng.uniform(-1.5, 1.5, (n, 6))
    xi[:6, 3:] *= 1e-6  # Taylor branches of exp / ljac / getQ
    xi[6, 3:] = 0.0
    out = dict(T=T, T2=T2, xi=xi)
    SE3, SO3 = mink.SE3, mink.SO3
    out["se3_log"] = np.array([SE3(t).log() for t in T])
    out["se3_jlog"] = np.array([SE3(t).jlog() for t in T])
    out["se3_inverse"] = np.array([SE3(t).inverse().wxyz_xyz for t in T])
    out["se3_adjoint"] = np.array([SE3(t).adjoint() for t in T])
    out["se3_matrix"] = np.array([SE3(t).as_matrix() for t in T])
    out["se3_multiply"] = np.array([(SE3(a) @ SE3(b)).wxyz_xyz for a, b in zip(T, T2)])
    out["se3_rminus"] = np.array([SE3(a).rminus(SE3(b)) for a, b in zip(T, T2)])
    out["se3_lminus"] = np.array([SE3(a).lminus(SE3(b)) for a, b in zip(T, T2)])
    out["se3_exp"] = np.array([SE3.exp(x).wxyz_xyz for x in xi])
    out["se3_rplus"] = np.array([SE3(t).rplus(x).wxyz_xyz for t, x in zip(T, xi)])
    out["se3_lplus"] = np.array([SE3(t).lplus(x).wxyz_xyz for t, x in zip(T, xi)])
    out["se3_ljac"] = np.array([SE3.ljac(x) for x in xi])
    out["se3_ljacinv"] = np.array([SE3.ljacinv(x) for x in xi])
    out["se3_apply"] = np.array([SE3(t).apply(x[:3]) for t, x in zip(T, xi)])
    out["so3_log"] = np.array([SO3(t[:4]).log() for t in T])
    out["so3_exp"] = np.array([SO3.exp(x[3:]).wxyz for x in xi])
    out["so3_ljac"] = np.array([SO3.ljac(x[3:]) for x in xi])
    out["so3_ljacinv"] = np.array([SO3.ljacinv(x[3:]) for x in xi])
    out["so3_matrix"] = np.array([SO3(t[:4]).as_matrix() for t in T])
    out["so3_from_matrix"] = np.array([SO3.from_matrix(SO3(t[:4]).as_matrix()).wxyz for t in T])
    out["so3_apply"] = np.array([SO3(t[:4]).apply(x[:3]) for t, x in zip(T, xi)])
    rpy = rng.uniform(-1.2, 1.2, (n, 3))
    out["rpy"] = rpy
    out["so3_from_rpy"] = np.array([SO3.from_rpy_radians(*r).wxyz for r in rpy])
    out["so3_as_rpy"] = np.array([list(SO3.from_rpy_radians(*r).as_rpy_radians()) for r in rpy])
    np.savez_compressed(os.path.join(HERE, "lie.npz"), **out)
    print("lie.npz", n)


# ---------------------------------------------------------- tidybot IK ----
ARM_VEL = {"joint_x": 0.5, "joint_y": 0.5, "joint_th": 1.0,
           "joint_1": 1.39, "joint_2": 1.39, "joint_3": 1.39, "joint_4": 1.39,
           "joint_5": 1.22, "joint_6": 1.22, "joint_7": 1.22}


def sample_q(model, rng, home):
    """Start configurations: home (+) perturbation of base and arm (fingers 0)."""
    q = home.copy()
    q[0:2] += rng.uniform(-1, 1, 2)
    q[2] += rng.uniform(-np.pi, np.pi)
    q[3:10] += rng.uniform(-0.5, 0.5, 7)
    for j in range(model.njnt):
        if model.jnt_limited[j] and model.jnt_type[j] != 0 and j < 10:
            a = model.jnt_qposadr[j]
            q[a] = np.clip(q[a], model.jnt_range[j, 0] + 0.05, model.jnt_range[j, 1] - 0.05)
    return q


def make_build_ik(scene, out_name, with_extras):
    rng = np.random.default_rng(7 if with_extras else 8)
    model = mujoco.MjModel.from_xml_path(os.path.join(SCENES, scene))
    home = model.key_qpos[model.key("home").id].copy()
    nv = model.nv
    N = 24
    pc = np.zeros(nv)
    pc[3:] = 1e-3
    dc = np.zeros(nv)
    dc[:3] = 100.0
    dt = 1.0 / 200.0
    rec = {k: [] for k in (
        "q", "target", "site_pose", "frame_jac", "frame_err", "frame_task_jac", "H_frame", "c_frame",
        "H_posture", "c_posture", "H_damping", "c_damping", "G_cfg", "h_cfg", "G_vel", "h_vel",
        "P", "c", "G", "h", "v", "q_next", "P12", "c12", "v12", "v_nolimit", "v_default_limit",
        "rel_target", "rel_err", "rel_jac", "H_rel", "c_rel", "v_rel", "base_jac", "base_pose",
        "com_err", "com_jac")}
    for i in range(N):
        q = sample_q(model, rng, home)
        if i == 0:
            q = home.copy()  # fingers exactly on their limits, site at its own pose
        if i == 1:
            q[4] = model.jnt_range[4, 1] - 1e-4  # joint_2 close to its upper limit
        if i == 2:
            q[6] = model.jnt_range[6, 0] + 5e-3  # joint_4 close to its lower limit
        cfg = mink.Configuration(model, q)
        qs = sample_q(model, rng, home)
        tgt_cfg = mink.Configuration(model, qs)
        T_wt = tgt_cfg.get_transform_frame_to_world("pinch_site", "site")
        if i == 0:
            T_wt = cfg.get_transform_frame_to_world("pinch_site", "site")  # fulfilled task
        if i == 3:
            T_wt = cfg.get_transform_frame_to_world("pinch_site", "site") @ mink.SE3.exp(
                np.array([1e-3, -2e-3, 1e-3, 2e-6, 1e-6, -3e-6]))  # below the jlog switch
        if i in (4, 5, 6):
            T_wt = cfg.get_transform_frame_to_world("pinch_site", "site") @ mink.SE3.exp(
                np.concatenate([rng.uniform(-0.1, 0.1, 3), rng.uniform(-0.3, 0.3, 3)]))
        frame = mink.FrameTask("pinch_site", "site", position_cost=1.0, orientation_cost=1.0, lm_damping=1.0)
        frame.set_target(T_wt)
        posture = mink.PostureTask(model, cost=pc)
        posture.set_target(home)
        damping = mink.DampingTask(model, dc)
        cfg_lim = mink.ConfigurationLimit(model)
        vel_lim = mink.VelocityLimit(model, ARM_VEL)
        rec["q"].append(q)
        rec["target"].append(T_wt.wxyz_xyz)
        rec["site_pose"].append(cfg.get_transform_frame_to_world("pinch_site", "site").wxyz_xyz)
        rec["frame_jac"].append(cfg.get_frame_jacobian("pinch_site", "site"))
        rec["frame_err"].append(frame.compute_error(cfg))
        rec["frame_task_jac"].append(frame.compute_jacobian(cfg))
        for name, task in (("frame", frame), ("posture", posture), ("damping", damping)):
            H, c = task.compute_qp_objective(cfg)
            rec["H_" + name].append(H)
            rec["c_" + name].append(c)
        G, h = cfg_lim.compute_qp_inequalities(cfg, dt)
        rec["G_cfg"].append(G)
        rec["h_cfg"].append(h)
        G, h = vel_lim.compute_qp_inequalities(cfg, dt)
        rec["G_vel"].append(G)
        rec["h_vel"].append(h)
        # config-2 style problem: Frame + Posture + Damping, Configuration + Velocity limits
        tasks = [frame, posture, damping]
        limits = [cfg_lim, vel_lim]
        prob = mink.build_ik(cfg, tasks, dt, 1e-3, limits)
        rec["P"].append(prob.P)
        rec["c"].append(prob.q)
        rec["G"].append(prob.G)
        rec["h"].append(prob.h)
        v = mink.solve_ik(cfg, tasks, dt, "quadprog", 1e-3, limits=limits)
        rec["v"].append(v)
        rec["q_next"].append(cfg.integrate(v, dt))
        # config-1 style: Frame + Posture, default limit, default 1e-12 damping too
        prob = mink.build_ik(cfg, [frame, posture], dt)
        rec["P12"].append(prob.P)
        rec["c12"].append(prob.q)
        rec["v12"].append(mink.solve_ik(cfg, [frame, posture], dt, "quadprog"))
        rec["v_default_limit"].append(mink.solve_ik(cfg, [frame, posture], dt, "quadprog", 1e-3))
        rec["v_nolimit"].append(mink.solve_ik(cfg, [frame, posture], dt, "quadprog", 1e-3, limits=[]))
        if with_extras:
            rel = mink.RelativeFrameTask("pinch_site", "site", "base_link", "body",
                                         position_cost=[1.0, 2.0, 0.5], orientation_cost=0.7,
                                         gain=0.8, lm_damping=0.5)
            T_rt = tgt_cfg.get_transform("pinch_site", "site", "base_link", "body")
            rel.set_target(T_rt)
            rec["rel_target"].append(T_rt.wxyz_xyz)
            rec["rel_err"].append(rel.compute_error(cfg))
            rec["rel_jac"].append(rel.compute_jacobian(cfg))
            H, c = rel.compute_qp_objective(cfg)
            rec["H_rel"].append(H)
            rec["c_rel"].append(c)
            rec["v_rel"].append(mink.solve_ik(cfg, [rel, posture], dt, "quadprog", 1e-3, limits=limits))
            rec["base_jac"].append(cfg.get_frame_jacobian("base_link", "body"))
            rec["base_pose"].append(cfg.get_transform_frame_to_world("base_link", "body").wxyz_xyz)
            com = mink.ComTask(cost=[1.0, 1.0, 0.5])
            com.set_target(tgt_cfg.data.subtree_com[1])
            rec["com_err"].append(com.compute_error(cfg))
            rec["com_jac"].append(com.compute_jacobian(cfg))
    out = {k: np.array(v) for k, v in rec.items() if v}
    out.update(dict(dt=dt, posture_cost=pc, damping_cost=dc, home=home,
                    vel_names=np.array(list(ARM_VEL.keys())), vel_limits=np.array(list(ARM_VEL.values()))))
    np.savez_compressed(os.path.join(HERE, out_name), **out)
    print(out_name, N)


def make_loop():
    """The caller's loop (interactive_scripts/teleop_mouse.py:105-121): up to 20 x
    (solve_ik -> integrate_inplace -> error) with the 1e-4 / 1e-4 exit test."""
    rng = np.random.default_rng(9)
    model = mujoco.MjModel.from_xml_path(os.path.join(SCENES, "cube.xml"))
    home = model.key_qpos[model.key("home").id].copy()
    pc = np.zeros(model.nv)
    pc[3:] = 1e-3
    dt = 1.0 / 200.0
    K, MAXIT = 6, 20
    q0s, targets, qs, vs, errs, iters = [], [], [], [], [], []
    for i in range(K):
        q0 = sample_q(model, rng, home) if i else home.copy()
        cfg = mink.Configuration(model, q0)
        T0 = cfg.get_transform_frame_to_world("pinch_site", "site")
        T_wt = T0 @ mink.SE3.exp(np.concatenate([rng.uniform(-0.1, 0.1, 3), rng.uniform(-0.3, 0.3, 3)]))
        frame = mink.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
        frame.set_target(T_wt)
        posture = mink.PostureTask(model, cost=pc)
        posture.set_target(home)
        q_tr = np.full((MAXIT, model.nq), np.nan)
        v_tr = np.full((MAXIT, model.nv), np.nan)
        e_tr = np.full((MAXIT, 6), np.nan)
        n = MAXIT
        for it in range(MAXIT):
            v = mink.solve_ik(cfg, [frame, posture], dt, "quadprog", 1e-3)
            cfg.integrate_inplace(v, dt)
            e = frame.compute_error(cfg)
            q_tr[it], v_tr[it], e_tr[it] = cfg.q, v, e
            if np.linalg.norm(e[:3]) <= 1e-4 and np.linalg.norm(e[3:]) <= 1e-4:
                n = it + 1
                break
        q0s.append(q0); targets.append(T_wt.wxyz_xyz); qs.append(q_tr); vs.append(v_tr)
        errs.append(e_tr); iters.append(n)
    np.savez_compressed(os.path.join(HERE, "tidybot_loop.npz"), q0=np.array(q0s), target=np.array(targets),
                        q=np.array(qs), v=np.array(vs), err=np.array(errs), iters=np.array(iters),
                        dt=dt, posture_cost=pc, home=home)
    print("tidybot_loop.npz", iters)


def make_collision():
    """CollisionAvoidanceLimit over the primitive pairs {pad boxes} x {cube, floor}."""
    rng = np.random.default_rng(11)
    model = mujoco.MjModel.from_xml_path(os.path.join(SCENES, "cube.xml"))
    home = model.key_qpos[model.key("home").id].copy()
    pads = ["right_pad1", "right_pad2", "left_pad1", "left_pad2"]
    lim = mink.CollisionAvoidanceLimit(model, [(pads, ["cube_geom", "floor"])])
    dt = 1.0 / 200.0
    qs, Gs, hs = [], [], []
    for i in range(12):
        q = sample_q(model, rng, home)
        cfg = mink.Configuration(model, q)
        if i % 2 == 0:
            # put the cube within ~1 cm of a pad so that rows become active
            g = model.geom(pads[i % 4]).id
            p = cfg.data.geom_xpos[g] + rng.normal(size=3) * 0.004
            dirn = rng.normal(size=3)
            dirn /= np.linalg.norm(dirn)
            q[18:21] = p + dirn * 0.036
            qq = rng.normal(size=4)
            q[21:25] = qq / np.linalg.norm(qq)
            cfg.update(q)
        G, h = lim.compute_qp_inequalities(cfg, dt)
        qs.append(q); Gs.append(G); hs.append(h)
    np.savez_compressed(os.path.join(HERE, "collision.npz"), q=np.array(qs), G=np.array(Gs), h=np.array(hs),
                        pairs=np.array(lim.geom_id_pairs), dt=dt)
    print("collision.npz pairs", lim.geom_id_pairs, "active rows", int(np.isfinite(np.array(hs)).sum()))


def make_api_strings():
    """Exception classes and messages raised on the path (asserted verbatim by the
    reference's tests/test_posture_task.py:29-62, tests/test_velocity_limit.py etc.)."""
    model = mujoco.MjModel.from_xml_path(os.path.join(SCENES, "cube.xml"))
    cfg = mink.Configuration(model)
    cases = {}

    def record(name, fn):
        try:
            fn()
            cases[name] = None
        except Exception as e:  # noqa: BLE001
            cases[name] = [type(e).__name__, str(e)]

    record("frame_neg_cost", lambda: mink.FrameTask("pinch_site", "site", -1.0, 1.0))
    record("frame_bad_cost_shape", lambda: mink.FrameTask("pinch_site", "site", [1.0, 2.0], 1.0))
    record("frame_bad_ori_shape", lambda: mink.FrameTask("pinch_site", "site", 1.0, [1.0, 2.0]))
    record("frame_no_target", lambda: mink.FrameTask("pinch_site", "site", 1.0, 1.0).compute_error(cfg))
    record("task_bad_gain", lambda: mink.FrameTask("pinch_site", "site", 1.0, 1.0, gain=1.5))
    record("task_bad_damping", lambda: mink.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=-1.0))
    record("posture_bad_cost_shape", lambda: mink.PostureTask(model, cost=[1.0, 2.0]))
    record("posture_neg_cost", lambda: mink.PostureTask(model, cost=-1.0))
    record("posture_bad_target", lambda: mink.PostureTask(model, cost=1.0).set_target(np.zeros(3)))
    record("posture_no_target", lambda: mink.PostureTask(model, cost=1.0).compute_error(cfg))
    record("com_bad_cost_shape", lambda: mink.ComTask(cost=[1.0, 2.0]))
    record("com_neg_cost", lambda: mink.ComTask(cost=-1.0))
    record("com_bad_target", lambda: mink.ComTask(cost=1.0).set_target(np.zeros(2)))
    record("com_no_target", lambda: mink.ComTask(cost=1.0).compute_error(cfg))
    record("rel_no_target", lambda: mink.RelativeFrameTask(
        "pinch_site", "site", "base_link", "body", 1.0, 1.0).compute_error(cfg))
    record("cfg_limit_bad_gain", lambda: mink.ConfigurationLimit(model, gain=0.0))
    record("vel_limit_free", lambda: mink.VelocityLimit(model, {"interactive_obj_freejoint": np.ones(6)}))
    record("vel_limit_bad_shape", lambda: mink.VelocityLimit(model, {"joint_1": [1.0, 2.0]}))
    record("unsupported_frame", lambda: cfg.get_frame_jacobian("pinch_site", "joint"))
    record("invalid_frame", lambda: cfg.get_frame_jacobian("nope", "site"))
    record("invalid_keyframe", lambda: cfg.update_from_keyframe("nope"))
    q_bad = model.key_qpos[0].copy()
    q_bad[4] = 3.0
    record("limits_violated", lambda: mink.Configuration(model, q_bad).check_limits())
    record("limits_violated_within_tol", lambda: mink.Configuration(
        model, np.where(np.arange(model.nq) == 4, 2.24 + 5e-7, model.key_qpos[0])).check_limits())
    with open(os.path.join(HERE, "api_strings.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("api_strings.json", len(cases))


def make_limited_ball():
    """ConfigurationLimit of the reference on a model with a LIMITED ball joint: the constructor's
    arrays (qpos_dim slots per joint, mink/limits/configuration_limit.py:46-53) and the rows at a
    few configurations."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_models
    rng = np.random.default_rng(13)
    model = mujoco.MjModel.from_xml_string(ref_models.LIMITED_BALL_AND_HINGE)
    limit = mink.ConfigurationLimit(model, gain=0.9, min_distance_from_limits=0.05)
    out = dict(lower=limit.lower, upper=limit.upper, indices=np.asarray(limit.indices),
               projection_matrix=limit.projection_matrix, gain=0.9, min_distance_from_limits=0.05, dt=0.01)
    qs, Gs, hs = [], [], []
    for _ in range(12):
        q = np.concatenate([random_unit_quat(rng), rng.uniform(-0.4, 1.4, 1)])
        cfg = mink.Configuration(model, q)
        con = limit.compute_qp_inequalities(cfg, 0.01)
        qs.append(q); Gs.append(con.G); hs.append(con.h)
    out.update(q=np.array(qs), G=np.array(Gs), h=np.array(hs))
    np.savez(os.path.join(HERE, "limited_ball.npz"), **out)
    print("limited_ball.npz", out["h"].shape)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "limited_ball":
        make_limited_ball()
        sys.exit(0)
    make_lie()
    make_build_ik("cube.xml", "tidybot_build_ik.npz", with_extras=True)
    make_build_ik("scene_mobile_kinova.xml", "mobile_kinova_build_ik.npz", with_extras=False)
    make_loop()
    make_collision()
    make_api_strings()
    make_limited_ball()
