"""Headline benchmark: TidyBot whole-body IK QP solves/sec (fp64) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one pass of the hot path over the whole batch: every rank launches the
iterated IK kernel (50 x [solve_ik -> integrate_inplace -> error]) once over its
contiguous share of ONE fixed global batch of 1 048 576 problems (seed 4), so the total
work is the same at every N (strong scaling) and no data-path collective is needed.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Ranks: started by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment), or -- when `--gpus N` is given without them -- by this script itself, which
then only starts N fresh child processes and waits for them (the parent never touches a
GPU).  With fewer visible devices than ranks (a rehearsal on a one-GPU box) the ranks share
devices and the host-side collectives run over gloo instead of RCCL.

At N=1, rank 0 additionally times the CPU restatement of the same path (oracle/, kind
"port") on the host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TOTAL = 1 << 20         # problems in the global batch (benchmark config 5)
ITERS = 50              # iterated solve_ik steps per problem (benchmark config 3/5)
SEED = 4
# Algorithmic work per solve for TidyBot cube.xml (nv = 24, 10 chain dofs), dense-H
# convention: 14.8 kflop and 648 B (read q 200 + target 56, write q' 200 + v 192).
FLOP_PER_SOLVE = 14.8e3
BYTES_PER_SOLVE = 648.0
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector = 1/2 of the 157.3 TF fp32 vector peak
HBM_PEAK_GBS = 8000.0
ONE_WAVE_FMA_TFLOPS = 62.0       # measured: pure fp64 FMA stream at one wave per SIMD (75 at four waves per SIMD)
KERNEL_SOURCES = ("mink_amd/csrc/mk_fused.hip", "mink_amd/csrc/mk_device.h")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--total", type=int, default=TOTAL, help="problems in the global batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads")
    return ap.parse_args()


def spawn_ranks(n):
    """Start n fresh copies of this script, one rank each, and wait.  Runs before anything in
    this process has touched a GPU (no torch import yet), and the children are new processes,
    not a re-exec of this one."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    return max((abs(c) for c in codes), default=0)


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def time_launches(fn, steps, warmup, barrier):
    import torch
    for i in range(warmup):
        fn(i)
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        fn(warmup + i)
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    return wall, ev0.elapsed_time(ev1) * 1e-3


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline(model, q0_np, targets_np, gpu_q_final):
    """The CPU restatement of the path on the host cores, bounded samples of the GPU workload
    (same rows, same 50 iterated steps): the C port on all cores and on one thread, each the
    median of 5 repeats after a warm-up run, and the single-thread numpy restatement at the
    reference's op granularity through the callers' <= 20-iteration loop."""
    import numpy as np
    from mink_amd import workloads as W
    from oracle import c_oracle as co
    if not co.available():
        return None
    cores = min(16, len(os.sched_getaffinity(0)))
    orc = co.COracle(model, postures=[dict(cost=W.posture_cost(model), target_q=W.home_q(model))],
                     velocity=W.VELOCITY_LIMITS)
    # timed as the algorithm the reference runs: the dual iteration without the checker's polish step
    co.lib().or_set_polish(0)

    def run(n, threads):
        t = time.perf_counter()
        res = orc.iterate(q0_np[:n], targets_np[:n], W.DT, W.DAMPING, ITERS, nthreads=threads)
        return time.perf_counter() - t, res

    def leg(threads, seconds_per_repeat):
        t, _ = run(64 * threads, threads)          # warm-up and calibration
        rate = 64 * threads * ITERS / max(t, 1e-6)
        n = int(min(len(q0_np), max(64 * threads, seconds_per_repeat * rate / ITERS)))
        n -= n % 64
        times = []
        for _ in range(5):
            t, res = run(n, threads)
            times.append(t)
        return n, times, res

    t_all = time.perf_counter()
    n_all, times_all, res_all = leg(cores, 1.5)
    n_one, times_one, _ = leg(1, 1.0)
    diff = float(np.abs(res_all[0] - gpu_q_final[:n_all]).max())
    out = {"value": n_all * ITERS / _median(times_all), "unit": "solves/s", "cores": cores, "kind": "port",
           "sample": f"rows [0, {n_all}) of the global batch x {ITERS} iterated steps, C restatement "
                     f"(oracle/ik_oracle.c, OpenMP over problems, {cores} threads), median of 5 repeats of "
                     f"{_median(times_all):.2f} s after a warm-up run",
           "single_thread_value": n_one * ITERS / _median(times_one),
           "single_thread_sample": f"rows [0, {n_one}) x {ITERS} steps, 1 thread, median of 5 repeats of "
                                   f"{_median(times_one):.2f} s",
           "max_abs_q_diff_vs_gpu": diff}
    # the numpy restatement at the reference's op granularity (one problem at a time, one small
    # numpy op at a time), through the callers' loop: <= 20 x (solve_ik, integrate_inplace,
    # frame-task error) with the 1e-4 / 1e-4 exit test; about 8 s of it
    try:
        from oracle import ik_oracle as o
        t0 = time.perf_counter()
        solves = problems = 0
        while time.perf_counter() - t0 < 8.0 and problems < len(q0_np):
            b = problems
            cfg = o.Configuration(model, q0_np[b])
            ft = o.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0, target=targets_np[b])
            tasks = [ft, o.PostureTask(cost=W.posture_cost(model), target_q=W.home_q(model))]
            limits = [o.ConfigurationLimit(model), o.VelocityLimit(model, W.VELOCITY_LIMITS)]
            for _ in range(20):
                cfg.integrate_inplace(o.solve_ik(cfg, tasks, W.DT, damping=W.DAMPING, limits=limits), W.DT)
                solves += 1
                e = ft.error(cfg)
                if np.linalg.norm(e[:3]) <= 1e-4 and np.linalg.norm(e[3:]) <= 1e-4:
                    break
            problems += 1
        wall = time.perf_counter() - t0
        out["python_numpy_value"] = solves / wall
        out["python_numpy_sample"] = (f"{problems} problems (rows [0, {problems})) through the <= 20-iteration loop = "
                                      f"{solves} solves in {wall:.1f} s, 1 thread, numpy restatement of mink at its "
                                      f"own op granularity (oracle/ik_oracle.py); a bounded sample, not the 10 000 "
                                      f"problems of the full protocol, which would take about "
                                      f"{10000 * 20 / max(solves / wall, 1e-9) / 60:.0f} min")
    except Exception as e:  # noqa: BLE001
        out["python_numpy_sample"] = f"not measured: {e}"
    out["cpu_seconds_total"] = time.perf_counter() - t_all
    return out


def kkt_residual(prob, q, targets, x, W):
    """Worst KKT residual of the solutions x (= dq) over a batch whose inequality rows are all
    +-e_i (Configuration / Velocity limits): primal infeasibility, and stationarity with the
    sign conditions of the multipliers (g = P x + c must vanish on interior dofs, be <= 0 at
    an upper and >= 0 at a lower bound).  Evaluated on the dense (P, c, G, h) the library
    itself exports, in torch; it is a report, not part of any timed region."""
    import torch
    P, c, G, h = prob.build_qp(q, targets, W.DT, W.DAMPING)
    inf = torch.full_like(h, float("inf"))[:, :, None]
    hi = torch.where(G > 0.5, h[:, :, None], inf).min(dim=1).values
    lo = torch.where(G < -0.5, -h[:, :, None], -inf).max(dim=1).values
    g = torch.einsum("bij,bj->bi", P, x) + c
    tol = 1e-12
    at_hi = x >= hi - tol * hi.abs().clamp(min=1.0)
    at_lo = x <= lo + tol * lo.abs().clamp(min=1.0)
    stat = torch.where(at_hi, g.clamp(min=0.0), torch.where(at_lo, (-g).clamp(min=0.0), g.abs()))
    infeas = torch.maximum(x - hi, lo - x).clamp(min=0.0)
    return {"stationarity_max": float(stat.max().item()), "infeasibility_max": float(infeas.max().item()),
            "gradient_scale": float(c.abs().max().item()),
            "active_bounds_mean": float((at_hi | at_lo).double().sum(dim=1).mean().item())}


def other_workloads(model, dm, prob, q0, targets, out, device, barrier):
    """Secondary workloads of the N=1 run (benchmark configs 1-4, the near-target loop, the
    arm + base model, waypoint rollouts)."""
    import numpy as np
    import torch
    from mink_amd import _lib as L
    from mink_amd import workloads as W
    from mink_amd.engine import DeviceModel, DeviceProblem
    from mink_amd.mjcf import load_builtin
    extra = {}
    # benchmark config 1: one problem, Frame+Posture with the default ConfigurationLimit: latency of one
    # solve_ik step and of the callers' <=20-step loop, launch + completion, inputs on the device
    p1 = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))
    q1, t1 = q0[:1].clone(), targets[:1].contiguous()
    v1 = torch.empty((1, model.nv), dtype=torch.float64, device=device)
    s1 = torch.empty(1, dtype=torch.int32, device=device)

    def lat(fn, n=200):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    o1 = dict(iters=torch.empty(1, dtype=torch.int32, device=device),
              err=torch.empty((1, 6), dtype=torch.float64, device=device),
              status=torch.empty(1, dtype=torch.int32, device=device))
    extra["config1_B1_latency"] = {
        "solve_ik_step_us": 1e6 * lat(lambda: p1.solve(q1, t1, W.DT, W.DAMPING, v=v1, status=s1)),
        "loop20_us": 1e6 * lat(lambda: (q1.copy_(q0[:1]), p1.iterate_(q1, t1, W.DT, W.DAMPING, 20, 1e-4, 1e-4, out=o1))),
        "note": "wall time per call incl. launch and synchronisation; the loop is one launch of up to 20 solves"}
    # benchmark config 2: 4096 problems, Frame+Posture+Damping, Cfg+Vel limits, one solve_ik step
    p2 = DeviceProblem(dm, W.task_specs(model, True), W.limit_specs(model, True))
    v = torch.empty((4096, model.nv), dtype=torch.float64, device=device)
    st = torch.empty(4096, dtype=torch.int32, device=device)
    q2, t2 = q0[:4096].contiguous(), targets[:4096].contiguous()
    w2, e2 = time_launches(lambda i: p2.solve(q2, t2, W.DT, W.DAMPING, v=v, status=st), 200, 20, barrier)
    extra["config2_B4096_single_step"] = {"solves_per_s": 4096 * 200 / w2, "us_per_launch": 1e6 * e2 / 200,
                                          "kkt": kkt_residual(p2, q2, t2, v * W.DT, W),
                                          "kernel": "coop_step_kernel<10> (16 lanes per problem; the library's "
                                                    "choice up to 16384 problems)"}
    with L.options(coop=0):
        w2l, e2l = time_launches(lambda i: p2.solve(q2, t2, W.DT, W.DAMPING, v=v, status=st), 200, 20, barrier)
    extra["config2_B4096_single_step"]["lane_kernel_us_per_launch"] = 1e6 * e2l / 200
    # benchmark config 3: 65536 problems x 50 iterated steps (far targets, fixed work), then the same with
    # the early exit
    q3 = [q0[:65536].clone() for _ in range(6)]
    t3 = targets[:65536].contiguous()
    o3 = dict(iters=out["iters"][:65536], err=out["err"][:65536], status=out["status"][:65536])
    w3, e3 = time_launches(lambda i: prob.iterate_(q3[i], t3, W.DT, W.DAMPING, ITERS, out=o3), 5, 1, barrier)
    extra["config3_B65536_x50"] = {"solves_per_s": 65536 * ITERS * 5 / w3, "ms_per_launch": 1e3 * e3 / 5}

    def loop_stats(r):
        it = r["iters"].double()
        conv = (r["err"][:, :3].norm(dim=1) <= 1e-4) & (r["err"][:, 3:].norm(dim=1) <= 1e-4)
        return {"fraction_converged": float(conv.double().mean().item()), "mean_solves": float(it.mean().item()),
                "max_solves": int(it.max().item()),
                "final_err_max": {"pos": float(r["err"][:, :3].norm(dim=1).max().item()),
                                  "ori": float(r["err"][:, 3:].norm(dim=1).max().item())}}

    qe = q0[:65536].clone()
    r = prob.iterate_(qe, t3, W.DT, W.DAMPING, ITERS, 1e-4, 1e-4)
    torch.cuda.synchronize()
    extra["config3_early_exit_far_targets"] = dict(
        loop_stats(r), note="random reachable targets, typically metres away: at 0.5 m/s and dt = 1/200 most "
                            "need far more than 50 steps")
    # near targets (offset mode: T_wb(q0) (+) exp(xi), xi_v ~ U(-0.1, 0.1) m, xi_w ~ U(-0.3, 0.3) rad), the
    # callers' loop as they run it: <= 20 iterations, 1e-4 / 1e-4 exit, default ConfigurationLimit only
    pn = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))
    Bn = 65536
    poses = dm.frame_pose(q0[:Bn].contiguous(), L.FRAME_TYPES["site"], model.site_id("pinch_site")).cpu().numpy()
    tn = torch.tensor(W.offset_targets(poses, np.random.default_rng(SEED + 1)), device=device)
    NSTEP_N = 20       # launches of 0.16 ms: enough of them that the fixed cost of the timing bracket does not show
    qn = [q0[:Bn].clone() for _ in range(NSTEP_N + 2)]
    on = dict(iters=torch.empty(Bn, dtype=torch.int32, device=device),
              err=torch.empty((Bn, 6), dtype=torch.float64, device=device),
              status=torch.empty(Bn, dtype=torch.int32, device=device))
    wn, en = time_launches(lambda i: pn.iterate_(qn[i], tn, W.DT, W.DAMPING, 20, 1e-4, 1e-4, out=on), NSTEP_N, 2, barrier)
    stats_n = loop_stats(on)
    for b in qn:
        b.copy_(q0[:Bn])
    with L.options(coop=0):
        wl_, el_ = time_launches(lambda i: pn.iterate_(qn[i], tn, W.DT, W.DAMPING, 20, 1e-4, 1e-4, out=on), NSTEP_N, 2, barrier)
    extra["near_target_loop_B65536"] = dict(
        stats_n, ms_per_launch=1e3 * en / NSTEP_N, solves_per_s=Bn * stats_n["mean_solves"] * NSTEP_N / wn,
        solves_per_s_by_launch_duration=Bn * stats_n["mean_solves"] * NSTEP_N / en,
        problems_per_s=Bn * NSTEP_N / wn, lane_kernel_alone_ms_per_launch=1e3 * el_ / NSTEP_N,
        kernels="fused_iterate_kernel<10, hand-off>, one launch: a wave leaves the lane-per-problem loop when 4 or "
                "fewer of its problems are left and finishes those at 16 lanes per problem",
        note="offset targets (<= 0.17 m, <= 0.52 rad), FrameTask + PostureTask, default ConfigurationLimit, "
             "<= 20 iterations with the 1e-4 / 1e-4 exit test (interactive_scripts/teleop_mouse.py:105-121); "
             "solves counted as executed")
    del qn, tn
    # the same loop at 2^20 problems, where throughput rather than the slowest wave counts: lanes
    # stream through their chunk's problems (the library's choice at this size), and, for scale,
    # the same launch with one block per 64 problems
    Bl = 1 << 20
    ql0 = q0[:Bl] if q0.shape[0] >= Bl else q0.repeat((Bl + q0.shape[0] - 1) // q0.shape[0], 1)[:Bl].contiguous()
    poses = dm.frame_pose(ql0, L.FRAME_TYPES["site"], model.site_id("pinch_site")).cpu().numpy()
    tl = torch.tensor(W.offset_targets(poses, np.random.default_rng(SEED + 2)), device=device)
    ql = [ql0.clone() for _ in range(4)]
    ol = dict(iters=torch.empty(Bl, dtype=torch.int32, device=device),
              err=torch.empty((Bl, 6), dtype=torch.float64, device=device),
              status=torch.empty(Bl, dtype=torch.int32, device=device))
    big = {}
    for name, knob in (("streaming", -1), ("block_per_64", 0)):
        for b in ql:
            b.copy_(ql0)
        with L.options(stream_blocks=knob):
            wl, el = time_launches(lambda i: pn.iterate_(ql[i], tl, W.DT, W.DAMPING, 20, 1e-4, 1e-4, out=ol), 3, 1, barrier)
        st = loop_stats(ol)
        big[name] = dict(ms_per_launch=1e3 * el / 3, solves_per_s=Bl * st["mean_solves"] * 3 / wl,
                         problems_per_s=Bl * 3 / wl)
        big.update(fraction_converged=st["fraction_converged"], mean_solves=st["mean_solves"],
                   max_solves=st["max_solves"])
    extra["near_target_loop_B1048576"] = big
    del ql, tl, ql0
    # the arm + base only model (mobile_kinova.xml, nq = nv = 10), same tasks and limits
    mk_model = load_builtin("mobile_kinova")
    mk_dm = DeviceModel(mk_model, device)
    mk_prob = DeviceProblem(mk_dm, W.task_specs(mk_model, False), W.limit_specs(mk_model, True))
    rng_k = np.random.default_rng(SEED + 7)
    Bk = 65536
    qk0 = torch.tensor(W.sample_q(mk_model, rng_k, Bk), device=device)
    tk = mk_dm.frame_pose(torch.tensor(W.sample_q(mk_model, rng_k, Bk), device=device),
                          L.FRAME_TYPES["site"], mk_model.site_id("pinch_site"))
    qk = [qk0.clone() for _ in range(6)]
    wk, ek = time_launches(lambda i: mk_prob.iterate_(qk[i], tk, W.DT, W.DAMPING, ITERS), 5, 1, barrier)
    extra["mobile_kinova_B65536_x50"] = {"solves_per_s": Bk * ITERS * 5 / wk, "ms_per_launch": 1e3 * ek / 5,
                                         "fused": bool(mk_prob.fused), "nv": mk_model.nv}
    del qk, tk, qk0
    # benchmark config 4: Frame + RelativeFrame + Posture, Cfg + Vel + CollisionAvoidance over the
    # primitive pairs {pads} x {cube, floor}
    t4, l4 = W.config4_specs(model)
    p4 = DeviceProblem(dm, t4, l4)
    B4 = 262144
    q4, tg4 = W.config4_inputs(model, dm, B4, 3)
    v4 = torch.empty((B4, model.nv), dtype=torch.float64, device=device)
    s4 = torch.empty(B4, dtype=torch.int32, device=device)
    w4, e4 = time_launches(lambda i: p4.solve(q4, tg4, W.DT, W.DAMPING, v=v4, status=s4), 3, 1, barrier)
    extra["config4_B262144_collision_relative_frame"] = {
        "solves_per_s": B4 * 3 / w4, "ms_per_launch": 1e3 * e4 / 3,
        "status_nonzero": int((s4 != 0).sum().item())}
    # the callers' loop on the same problem (20 passes, no exit test): passes of the staged kernels, and the
    # same loop as passes of the general kernel alone
    Bl = 65536
    out4 = dict(iters=torch.empty(Bl, dtype=torch.int32, device=device), err=torch.empty((Bl, 6), dtype=torch.float64, device=device),
                status=torch.empty(Bl, dtype=torch.int32, device=device), v=torch.empty((Bl, model.nv), dtype=torch.float64, device=device))
    ql = [q4[:Bl].clone() for _ in range(3)]
    tl = tg4[:Bl].contiguous()
    wl, el = time_launches(lambda i: p4.iterate_(ql[i], tl, W.DT, W.DAMPING, 20, -1.0, -1.0, out=out4), 2, 1, barrier)
    loop4 = {"staged_passes": {"solves_per_s": Bl * 20 * 2 / wl, "ms_per_loop": 1e3 * el / 2},
             "status_nonzero": int((out4["status"] != 0).sum().item())}
    with L.options(staged_loop=0):
        qg = [q4[:Bl].clone() for _ in range(2)]
        wg, eg = time_launches(lambda i: p4.iterate_(qg[i], tl, W.DT, W.DAMPING, 20, -1.0, -1.0, out=out4), 1, 1, barrier)
    loop4["general_kernel_passes"] = {"solves_per_s": Bl * 20 / wg, "ms_per_loop": 1e3 * eg}
    loop4["max_abs_q_diff"] = float((ql[2] - qg[1]).abs().max().item())
    loop4["note"] = ("rows [0, 65536) of the config-4 batch are its near-contact quarter: every problem of this loop has a "
                     "collision pair in range in every pass; staged_passes_mixed runs the same loop on a batch with the "
                     "single-step workload's mix (a quarter near contact)")
    qm4, tm4 = W.config4_inputs(model, dm, Bl, SEED + 1)
    qm = [qm4.clone() for _ in range(3)]
    wm, em = time_launches(lambda i: p4.iterate_(qm[i], tm4, W.DT, W.DAMPING, 20, -1.0, -1.0, out=out4), 2, 1, barrier)
    loop4["staged_passes_mixed"] = {"solves_per_s": Bl * 20 * 2 / wm, "ms_per_loop": 1e3 * em / 2,
                                    "status_nonzero": int((out4["status"] != 0).sum().item())}
    del qm4, tm4, qm
    extra["config4_loop_B65536_x20"] = loop4
    del q4, tg4, v4, s4, ql, qg
    # kinematic move_to rollouts (waypoint generator + the callers' IK loop per waypoint step) as one launch
    import mink_amd as mk
    ftask = mk.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
    ptask = mk.PostureTask(model, cost=W.posture_cost(model))
    ptask.set_target(W.home_q(model))
    wcfg = mk.LinearWaypointReachConfig(0.01, 0.08, 0.05, 0.3)
    for Br in (65536, 262144):
        rq = q0[:Br].clone()
        rcfg = mk.Configuration(model, rq, device=device)
        start = rcfg.get_transform_frame_to_world("pinch_site", "site")
        g = torch.Generator(device="cpu").manual_seed(SEED)
        goal_pos = start.translation() + (torch.rand((Br, 3), generator=g, dtype=torch.float64) * 0.3 - 0.15).to(device)
        goal_rot = start.rotation() @ mk.SO3.exp((torch.rand((Br, 3), generator=g, dtype=torch.float64) * 0.8 - 0.4).to(device))
        mk.move_to(mk.Configuration(model, rq[:4096].clone(), device=device), ftask, [ftask, ptask],
                   goal_pos[:4096], mk.SO3(goal_rot.wxyz[:4096]), wcfg, max_steps=50)   # warm-up
        torch.cuda.synchronize()
        tr = time.perf_counter()
        ro = mk.move_to(rcfg, ftask, [ftask, ptask], goal_pos, goal_rot, wcfg, max_steps=50)
        torch.cuda.synchronize()
        tr = time.perf_counter() - tr
        extra[f"move_to_rollouts_B{Br}"] = {
            "rollouts_per_s": Br / tr, "seconds": tr, "fraction_reached": float(ro["reached"].double().mean().item()),
            "mean_waypoint_steps": float(ro["steps"].double().mean().item()),
            "ik_solves_per_s": float(ro["solves"].sum().item()) / tr,
            "mean_solves_per_rollout": float(ro["solves"].double().mean().item()),
            "max_solves_per_rollout": int(ro["solves"].max().item()),
            "note": "kinematic part of the reference's move_to (waypoint generator + <=20-iteration IK loop per "
                    "step, default ConfigurationLimit) as one launch (mk_move_to_batch), physics excluded; the "
                    "launch lasts as long as its slowest rollout (max_solves_per_rollout sequential solves)"}
    # the headline workload in float32 arithmetic (option "f32"; q, targets and v stay float64 in HBM): a
    # secondary figure with its deviation from the float64 run, never the headline
    Bf = q0.shape[0]
    qf = [q0.clone() for _ in range(3)]
    with L.options(f32=1):
        wf, ef = time_launches(lambda i: prob.iterate_(qf[i], targets, W.DT, W.DAMPING, ITERS, -1.0, -1.0, out=out), 2, 1, barrier)
    e32 = out["err"].clone()
    st32 = int((out["status"] != 0).sum().item())
    q64 = q0.clone()
    with L.options(coop=0):
        prob.iterate_(q64, targets, W.DT, W.DAMPING, ITERS, -1.0, -1.0, out=out)
    extra["f32_mode_headline_workload"] = {
        "dtype": "f32", "solves_per_s": Bf * ITERS * 2 / wf, "ms_per_launch": 1e3 * ef / 2, "problems": Bf, "iterations": ITERS,
        "solver_status_nonzero": st32,
        "max_abs_dq_vs_f64_after_the_loop": float((qf[-1] - q64).abs().max().item()),
        "max_abs_d_err_norm_vs_f64": float((e32.norm(dim=1) - out["err"].norm(dim=1)).abs().max().item()),
        "stated_bound": "single step: |v_f32 - v_f64| <= 1e-3 max(1, |v|_inf) rad/s for problems with a Levenberg-Marquardt "
                        "term on the frame task at damping >= 1e-3 (every caller of the reference sets lm_damping = 1, damping 1e-3); measured 1.6e-4 on this "
                        "workload (tools/soak_f32.py, tests/test_gpu_f32.py)"}
    return extra


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import numpy as np  # noqa: F401
    import torch

    from mink_amd import sharding as S
    from mink_amd import workloads as W
    from mink_amd.engine import DeviceModel, DeviceProblem
    from mink_amd.mjcf import load_builtin

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE {world}", file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()
    if ndev < 1:
        print("bench.py needs a GPU", file=sys.stderr)
        sys.exit(2)
    # RCCL needs one device per rank; with fewer devices than ranks (rehearsal) the ranks share
    # devices and the host-side collectives run over gloo
    backend = os.environ.get("MINK_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
    dev_index = local_rank % ndev
    device = torch.device(f"cuda:{dev_index}")
    torch.cuda.set_device(dev_index)
    red_device = device if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        barrier = dist.barrier
    else:
        dist = None

        def barrier():
            return None

    model = load_builtin("tidybot_cube")
    dm = DeviceModel(model, device)
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    assert prob.fused, "headline workload must run on the fused kernel"
    lo, hi = S.shard_bounds(args.total, rank, world)
    B = hi - lo
    q0_np, qs_np = S.global_rows(model, SEED, lo, hi)      # a row's inputs do not depend on the rank count
    site = model.site_id("pinch_site")
    targets = dm.frame_pose(torch.tensor(qs_np, device=device), 2, site).contiguous()   # reachable SE3 targets
    q0 = torch.tensor(q0_np, device=device)
    del qs_np
    nbuf = args.steps + args.warmup
    qbufs = [q0.clone() for _ in range(nbuf)]
    out = dict(iters=torch.empty(B, dtype=torch.int32, device=device),
               err=torch.empty((B, 6), dtype=torch.float64, device=device),
               status=torch.empty(B, dtype=torch.int32, device=device))

    def step(i):
        prob.iterate_(qbufs[i], targets, W.DT, W.DAMPING, ITERS, -1.0, -1.0, out=out)

    wall, ev = time_launches(step, args.steps, args.warmup, barrier)
    t = torch.tensor([wall], dtype=torch.float64, device=red_device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())
    # after the timed region: run statistics (one small all-reduce each for sums and maxima) and the gather
    # of the final configurations, timed separately
    stats = S.reduce_stats(
        {"bad": int((out["status"] != 0).sum().item()), "solves": float(out["iters"].sum().item()), "rows": B},
        {"pos": float(out["err"][:, :3].norm(dim=1).max().item()),
         "ori": float(out["err"][:, 3:].norm(dim=1).max().item())}, dist, red_device)
    gather_ms = None
    if dist is not None:
        src = qbufs[-1] if backend == "nccl" else qbufs[-1].cpu()
        S.gather_rows(src[: min(B, 1024)].contiguous(), min(B, 1024) * world, dist)    # warm-up of the collective
        torch.cuda.synchronize()
        barrier()
        tg0 = time.perf_counter()
        allq = S.gather_rows(src, args.total, dist)
        torch.cuda.synchronize()
        barrier()
        gather_ms = 1e3 * (time.perf_counter() - tg0)
        assert allq.shape[0] == args.total
        del allq
    assert int(stats["rows"]) == args.total
    value = args.total * ITERS * args.steps / wall_max
    kernel_s = ev / args.steps                 # average launch duration on this rank (HIP events)
    kernel_rate = B * ITERS / kernel_s

    if rank == 0:
        res = {
            "metric": "IK QP solves/sec (TidyBot whole-body, fp64)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall_max / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"benchmark config 5: {args.total} TidyBot (cube.xml, nv=24) problems, seed {SEED}, "
                                   f"identical at every GPU count, FrameTask(pinch_site)+PostureTask, "
                                   f"ConfigurationLimit+VelocityLimit, {ITERS} iterated solve_ik+integrate steps, "
                                   f"random reachable SE3 targets, dt=1/200, damping=1e-3",
                       "problems_total": args.total, "problems_per_gpu": B, "iterations": ITERS,
                       "solves_per_step": args.total * ITERS,
                       "parallelism": f"problem-sharded x{world}, no data-path collective",
                       "collective_backend": None if dist is None else ("rccl" if backend == "nccl" else backend)},
            "roofline": {"bound": "fp64_valu", "achieved": kernel_rate * FLOP_PER_SOLVE / 1e12,
                         "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": kernel_rate * FLOP_PER_SOLVE / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                         "traffic": None, "kernel": "fused_iterate_kernel<10>",
                         "kernel_ms": 1e3 * kernel_s, "flop_per_solve": FLOP_PER_SOLVE,
                         # what the occupancy of this kernel (one wave per SIMD) allows: a stream of nothing
                         # but fp64 FMAs measured on this machine (tools/ubench/fma_latency.hip)
                         "ceiling_one_wave_tflops": ONE_WAVE_FMA_TFLOPS,
                         # HBM side, from the measured traffic below when there is one; the per-solve
                         # figure of the I/O convention (q in, target in, v and q' out) is kept beside it
                         "hbm": {"achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                                 "algorithmic_bytes_per_solve": BYTES_PER_SOLVE,
                                 "algorithmic_rate_GBs": kernel_rate * BYTES_PER_SOLVE / 1e9}},
            "solver_status_nonzero": int(stats["bad"]),
            "solves_executed_per_step": int(stats["solves"]),
            "final_err_max": {"pos": stats["pos"], "ori": stats["ori"]},
        }
        if gather_ms is not None:
            res["gather_q_final_ms"] = gather_ms
        # counters of the same kernel from the committed profile, if it was taken from these sources
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            with open(tfile) as f:
                tj = json.load(f)
            rl = res["roofline"]
            if tj.get("kernel_source_sha256") == kernel_source_hash() and tj.get("problems_per_launch"):
                scale = B / tj["problems_per_launch"]      # traffic and flops are per problem
                rl["traffic"] = tj["hbm_bytes_per_launch"] * scale
                rl["traffic_note"] = tj.get("note")
                rl["hbm"]["achieved"] = rl["traffic"] / kernel_s / 1e9
                rl["hbm"]["frac"] = rl["hbm"]["achieved"] / HBM_PEAK_GBS
                if tj.get("executed_flop_per_solve"):
                    rl["executed_flop_per_solve"] = tj["executed_flop_per_solve"]
                    rl["executed_achieved"] = kernel_rate * tj["executed_flop_per_solve"] / 1e12
                    rl["executed_frac"] = rl["executed_achieved"] / FP64_VECTOR_PEAK_TFLOPS
            else:
                rl["traffic_note"] = "profiles/traffic.json was taken from other kernel sources than the ones timed: ignored"
        if world == 1 and not args.no_extra:
            res["other_workloads"] = other_workloads(model, dm, prob, q0, targets, out, device, barrier)
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            n_cpu = min(B, 32768)
            cb = cpu_baseline(model, q0_np[:n_cpu], targets[:n_cpu].cpu().numpy(), qbufs[-1][:n_cpu].cpu().numpy())
            if cb is not None:
                res["cpu_baseline"] = cb
                res["vs_cpu_baseline"] = value / cb["value"]
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
