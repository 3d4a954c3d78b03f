"""Soak test (GPU box): the fused kernels against the C restatement over randomly drawn task /
limit parameters, time steps and damping values -- a wider sweep than the unit tests hold.

    python tools/soak_parity.py [trials] [problems per trial] > profiles/<name>.md

Per trial: one model, one random parameter set, `B` random problems; a single solve_ik step
(velocity compared) and the callers' <= 10-iteration loop with the exit test (configuration,
solve counts and status compared).  The checker is oracle/ik_oracle.c on the host cores; this
script is test infrastructure, not part of the product."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem, qp_solve
from mink_amd.mjcf import load_builtin
from oracle import c_oracle as co

TRIALS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
# third argument: "lane" / "coop" pins the kernel form of the fused path (default: the library's
# choice, which is the cooperative form up to 16384 problems per call)
FORM = sys.argv[3] if len(sys.argv) > 3 else "default"
if FORM != "default":
    L.set_option("coop", 1 if FORM == "coop" else 0)
    L.set_option("handoff", 0 if FORM == "lane" else -1)
THREADS = min(16, len(os.sched_getaffinity(0)))
dev = "cuda:0"
models = {n: load_builtin(n) for n in ("tidybot_cube", "mobile_kinova", "tidybot_open")}
dms = {n: DeviceModel(m, dev) for n, m in models.items()}


def draw(rng, name):
    model = models[name]
    p = dict(model=name)
    p["dt"] = float(rng.choice([1 / 200, 1e-2, 5e-2]))
    p["damping"] = float(rng.choice([1e-12, 1e-8, 1e-3, 1e-1]))
    p["frame_cost"] = np.concatenate([np.full(3, 10 ** rng.uniform(-1, 1)), np.full(3, 10 ** rng.uniform(-1, 1))])
    p["frame_gain"] = float(rng.uniform(0.2, 1.0))
    p["frame_lm"] = float(rng.choice([0.0, 1e-2, 1.0]))
    pc = np.zeros(model.nv)
    pc[3:] = float(rng.choice([0.0, 1e-3, 1e-1]))
    p["posture_cost"], p["posture_gain"] = pc, float(rng.uniform(0.3, 1.0))
    p["with_damping_task"] = bool(rng.integers(2))
    p["with_velocity"] = bool(rng.integers(2))
    p["cfg_gain"] = float(rng.choice([0.5, 0.95, 1.0]))
    p["near"] = bool(rng.integers(2))
    # Levenberg-Marquardt term on the posture task (rare in callers; its own path in the kernels), drawn
    # from a stream of its own so that the other parameters of a set do not depend on it
    p["posture_lm"] = float(np.random.default_rng([2025, int(rng.integers(1 << 30))]).choice([0.0, 0.0, 1e-2]))
    return p


def specs(model, p):
    tasks = [dict(type=L.TASK_FRAME, frame_type=L.FRAME_TYPES["site"], frame_id=model.site_id("pinch_site"),
                  cost=p["frame_cost"], gain=p["frame_gain"], lm_damping=p["frame_lm"]),
             dict(type=L.TASK_POSTURE, cost=p["posture_cost"], target_q=W.home_q(model), gain=p["posture_gain"],
                  lm_damping=p["posture_lm"])]
    postures = [dict(cost=p["posture_cost"], target_q=W.home_q(model), gain=p["posture_gain"], lm_damping=p["posture_lm"])]
    if p["with_damping_task"]:
        tasks.append(dict(type=L.TASK_POSTURE, cost=W.base_damping_cost(model), target_q=model.qpos0.copy(), gain=0.0,
                          lm_damping=0.0))
        postures.append(dict(cost=W.base_damping_cost(model), target_q=model.qpos0.copy(), gain=0.0))
    limits = [dict(type=L.LIMIT_CONFIGURATION, gain=p["cfg_gain"], min_distance_from_limits=0.0)]
    vel = None
    if p["with_velocity"]:
        vel = {k: v for k, v in W.VELOCITY_LIMITS.items() if model.joint_id(k) >= 0}
        limits.append(dict(type=L.LIMIT_VELOCITY, index=[int(model.jnt_dofadr[model.joint_id(n)]) for n in vel],
                           vmax=list(vel.values())))
    orc = co.COracle(model, frame_cost=p["frame_cost"], frame_gain=p["frame_gain"], frame_lm=p["frame_lm"],
                     postures=postures, cfg_limit=(p["cfg_gain"], 0.0), velocity=vel)
    return tasks, limits, orc


def main():
    assert co.available(), "build the C oracle first (python -c 'import __graft_entry__ as g; g.build()')"
    print(f"# soak: fused kernels ({FORM} kernel form) against the C restatement, {TRIALS} random parameter sets x {B} problems\n")
    print("| # | model | dt | damping | frame gain / lm | posture | damping task | vel limit | cfg gain | targets | "
          "max abs dv (1 step) | rel. | max abs v | max abs dq (<= 10-step loop) | rel. | solve counts differ | status differ | general kernels vs fused: rel. dv | status differ |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    worst_v, worst_q, bad = 0.0, 0.0, 0
    worst_d = 0.0
    worst_well = [0.0, 0.0]   # parameter sets with damping >= 1e-3 or a Levenberg-Marquardt term
    t0 = time.time()
    for trial in range(TRIALS):
        rng = np.random.default_rng([2024, trial])
        name = ("tidybot_cube", "mobile_kinova", "tidybot_open")[trial % 3]
        model, dm = models[name], dms[name]
        p = draw(rng, name)
        tasks, limits, orc = specs(model, p)
        prob = DeviceProblem(dm, tasks, limits)
        if not prob.fused:
            print(f"| {trial} | {name} | not served by the fused path: skipped |")
            continue
        q0 = W.sample_q(model, rng, B)
        dq0 = torch.tensor(q0, device=dev)
        site = model.site_id("pinch_site")
        if p["near"]:
            tg = W.offset_targets(dm.frame_pose(dq0, 2, site).cpu().numpy(), rng)
        else:
            tg = dm.frame_pose(torch.tensor(W.sample_q(model, rng, B), device=dev), 2, site).cpu().numpy()
        dtg = torch.tensor(tg, device=dev)
        v, st = prob.solve(dq0, dtg, p["dt"], p["damping"])
        v_ref, st_ref = orc.solve_ik(q0, tg, p["dt"], p["damping"], nthreads=THREADS)
        ok = (st.cpu().numpy() == 0) & (st_ref == 0)
        dvs = np.abs(v.cpu().numpy() - v_ref)[ok]
        vmag = np.abs(v_ref[ok]).max(axis=1) if ok.any() else np.zeros(0)
        dv = float(dvs.max()) if ok.any() else 0.0
        # relative to the size of the problem's own answer: with a vanishing damping and no
        # Levenberg-Marquardt term the Hessian is singular to working precision and |v| is unbounded
        rv = float((dvs.max(axis=1) / np.maximum(1.0, vmag)).max()) if ok.any() else 0.0
        # the same step through the general kernels: dense build + dense dual-method QP
        P, c, G, h = prob.build_qp(dq0, dtg, p["dt"], p["damping"])
        xd, std = qp_solve(P, c, G, h)
        okd = ok & (std.cpu().numpy() == 0)
        dd = (np.abs(xd.cpu().numpy() / p["dt"] - v.cpu().numpy())[okd].max(axis=1) / np.maximum(1.0, np.abs(v_ref[okd]).max(axis=1))
              if okd.any() else np.zeros(1))
        rd = float(dd.max())
        n_std = int((std.cpu().numpy() != st.cpu().numpy()).sum())
        worst_d = max(worst_d, rd) if (p["damping"] >= 1e-3 or p["frame_lm"] > 0.0) else worst_d
        qg = dq0.clone()
        r = prob.iterate_(qg, dtg, p["dt"], p["damping"], 10, 1e-4, 1e-4)
        q_ref, it_ref, _, st2_ref = orc.iterate(q0, tg, p["dt"], p["damping"], 10, 1e-4, 1e-4, nthreads=THREADS)
        it = r["iters"].cpu().numpy()
        same = it == it_ref
        dqs = np.abs(qg.cpu().numpy() - q_ref)[same]
        moved = np.abs(q_ref - q0)[same].max(axis=1) if same.any() else np.zeros(0)
        dq = float(dqs.max()) if same.any() else 0.0
        rq = float((dqs.max(axis=1) / np.maximum(1.0, moved)).max()) if same.any() else 0.0
        n_it = int((~same).sum())
        n_st = int((st.cpu().numpy() != st_ref).sum() + (r["status"].cpu().numpy() != st2_ref).sum())
        worst_v, worst_q, bad = max(worst_v, rv), max(worst_q, rq), bad + n_st
        if p["damping"] >= 1e-3 or p["frame_lm"] > 0.0:
            worst_well = [max(worst_well[0], dv), max(worst_well[1], dq)]
        print(f"| {trial} | {name} | {p['dt']:.3g} | {p['damping']:.0e} | {p['frame_gain']:.2f} / {p['frame_lm']:.0e} | "
              f"{p['posture_cost'][3]:.0e} x {p['posture_gain']:.2f} lm {p['posture_lm']:.0e} | {p['with_damping_task']} | {p['with_velocity']} | "
              f"{p['cfg_gain']} | {'near' if p['near'] else 'far'} | {dv:.2e} | {rv:.1e} | {float(vmag.max()) if ok.any() else 0.0:.1e} | "
              f"{dq:.2e} | {rq:.1e} | {n_it} | {n_st} | {rd:.1e} | {n_std} |", flush=True)
    print(f"\nworst relative dv {worst_v:.3e}, worst relative dq {worst_q:.3e} (relative to max(1, |v|) of the problem / to "
          f"max(1, distance moved)); status mismatches {bad}; {TRIALS * B} problems in {time.time() - t0:.0f} s")
    print(f"\nparameter sets with damping >= 1e-3 or a Levenberg-Marquardt term on the frame task (what the callers use): "
          f"worst absolute |dv| {worst_well[0]:.3e} rad/s, worst absolute |dq| {worst_well[1]:.3e}; general kernels "
          f"(mk_build_qp_batch + mk_qp_solve_batch) against the fused kernel on those sets: worst relative |dv| {worst_d:.3e}")


if __name__ == "__main__":
    main()
