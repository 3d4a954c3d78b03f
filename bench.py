"""Headline benchmark: TidyBot whole-body IK QP solves/sec (fp64) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one pass of the hot path over one batch: a single launch of the
iterated IK kernel (50 x [solve_ik -> integrate_inplace -> error]) over this rank's
shard of 131072 problems -- the per-GPU share of the headline benchmark's 1M-problem / 8-GPU
configuration, so the batch grows with N (weak scaling) and no data-path collective
is needed.  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE
JSON line.  At N=1, rank 0 additionally times the C restatement of the same path
(oracle/, kind "port") on the host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mink_amd import _lib as L  # noqa: E402
from mink_amd import sharding as S  # noqa: E402
from mink_amd import workloads as W  # noqa: E402
from mink_amd.engine import DeviceModel, DeviceProblem  # noqa: E402
from mink_amd.mjcf import load_builtin  # noqa: E402

SHARD = 131072          # problems per GPU (1M / 8)
ITERS = 50              # iterated solve_ik steps per problem (benchmark config 3/5)
SEED = 4
# Algorithmic work per solve for TidyBot cube.xml (nv = 24, 10 chain dofs), dense-H
# convention: 14.8 kflop and 648 B (read q 200 + target 56, write q' 200 + v 192).
FLOP_PER_SOLVE = 14.8e3
BYTES_PER_SOLVE = 648.0
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector = 1/2 of the 157.3 TF fp32 vector peak
HBM_PEAK_GBS = 8000.0


def make_inputs(model, dm, lo, hi, seed, device):
    """Rows [lo, hi) of the global batch; a row's inputs do not depend on the rank count."""
    q0, qs = S.global_rows(model, seed, lo, hi)
    site = model.site_id("pinch_site")
    targets = dm.frame_pose(torch.tensor(qs, device=device), 2, site)   # reachable SE3 targets
    return torch.tensor(q0, device=device), targets.contiguous(), q0


def time_launches(fn, steps, warmup, barrier):
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


def cpu_baseline(model, q0_np, targets_np, gpu_q_final):
    """C restatement (oracle/ik_oracle.c) on the host cores over a bounded sample."""
    from oracle import c_oracle as co
    if not co.available():
        return None
    cores = min(16, len(os.sched_getaffinity(0)))
    orc = co.COracle(model, postures=[dict(cost=W.posture_cost(model), target_q=W.home_q(model))],
                     velocity=W.VELOCITY_LIMITS)
    # calibrate, then size the sample for roughly 10-20 s of CPU work
    t = time.perf_counter()
    orc.iterate(q0_np[:256], targets_np[:256], W.DT, W.DAMPING, ITERS, nthreads=cores)
    rate = 256 * ITERS / max(time.perf_counter() - t, 1e-6)
    n = int(min(len(q0_np), max(1024, 12.0 * rate / ITERS)))
    n -= n % 64
    # roughly 12 s of CPU work: the sample (at most the whole shard) repeated as often as needed
    reps = int(max(1, min(8, round(12.0 * rate / (n * ITERS)))))
    t = time.perf_counter()
    for _ in range(reps):
        qf, iters, err, st = orc.iterate(q0_np[:n], targets_np[:n], W.DT, W.DAMPING, ITERS, nthreads=cores)
    dt = time.perf_counter() - t
    diff = float(np.abs(qf - gpu_q_final[:n]).max())
    out = {"value": reps * n * ITERS / dt, "unit": "solves/s", "cores": cores, "kind": "port",
           "sample": f"{n} of the {len(q0_np)} problems x {ITERS} iterated steps x {reps} repeats, C restatement "
                     f"(oracle/ik_oracle.c, OpenMP over problems), {dt:.1f} s",
           "max_abs_q_diff_vs_gpu": diff}
    # the single-thread numpy restatement at the reference's op granularity, for scale
    try:
        from oracle import ik_oracle as o
        k = 40
        t = time.perf_counter()
        for b in range(k):
            cfg = o.Configuration(model, q0_np[b])
            tasks = [o.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0, target=targets_np[b]),
                     o.PostureTask(cost=W.posture_cost(model), target_q=W.home_q(model))]
            limits = [o.ConfigurationLimit(model), o.VelocityLimit(model, W.VELOCITY_LIMITS)]
            cfg.integrate_inplace(o.solve_ik(cfg, tasks, W.DT, damping=W.DAMPING, limits=limits), W.DT)
        out["python_numpy_value"] = k / (time.perf_counter() - t)
        out["python_numpy_note"] = "1 thread, numpy restatement of mink at its own op granularity (oracle/ik_oracle.py)"
    except Exception as e:  # noqa: BLE001
        out["python_numpy_note"] = f"not measured: {e}"
    return out


def kkt_residual(prob, q, targets, x):
    """Worst KKT residual of the solutions x (= dq) over a batch whose inequality rows are all
    +-e_i (Configuration / Velocity limits): primal infeasibility, and stationarity with the
    sign conditions of the multipliers (g = P x + c must vanish on interior dofs, be <= 0 at
    an upper and >= 0 at a lower bound).  Evaluated on the dense (P, c, G, h) the library
    itself exports, in torch; it is a report, not part of any timed region."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shard", type=int, default=SHARD)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MINK_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: the
    # ranks share the visible devices and the (host-side) collectives run over gloo
    backend = os.environ.get("MINK_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if world > 1 else 0
    device = torch.device(f"cuda:{dev_index}")
    red_device = device if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        barrier = dist.barrier
    else:
        dist = None
        torch.cuda.set_device(0)

        def barrier():
            return None
    n_gpus = world

    model = load_builtin("tidybot_cube")
    dm = DeviceModel(model, device)
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    assert prob.fused, "headline workload must run on the fused kernel"
    B = args.shard
    q0, targets, q0_np = make_inputs(model, dm, rank * B, (rank + 1) * B, SEED, device)
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
    solves_per_rank = B * ITERS * args.steps
    stats = S.reduce_stats(
        {"bad": int((out["status"] != 0).sum().item())},
        {"pos": float(out["err"][:, :3].norm(dim=1).max().item()),
         "ori": float(out["err"][:, 3:].norm(dim=1).max().item())}, dist, red_device)
    value = solves_per_rank * n_gpus / wall_max
    kernel_s = ev / args.steps                 # average launch duration on this rank (HIP events)
    kernel_rate = B * ITERS / kernel_s

    if rank == 0:
        res = {
            "metric": "IK QP solves/sec (TidyBot whole-body, fp64)",
            "value": value, "unit": "solves/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"benchmark config 5 shard: {B} TidyBot (cube.xml, nv=24) problems per GPU "
                                   f"({B * n_gpus} total), FrameTask(pinch_site)+PostureTask, "
                                   f"ConfigurationLimit+VelocityLimit, {ITERS} iterated solve_ik+integrate steps, "
                                   f"random reachable SE3 targets, dt=1/200, damping=1e-3",
                       "problems_per_gpu": B, "iterations": ITERS, "solves_per_step": B * ITERS * n_gpus,
                       "parallelism": f"problem-sharded x{n_gpus}, no data-path collective"},
            "roofline": {"bound": "fp64_valu", "achieved": kernel_rate * FLOP_PER_SOLVE / 1e12,
                         "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": kernel_rate * FLOP_PER_SOLVE / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                         "traffic": None, "kernel": "fused_iterate_kernel<10>",
                         "kernel_ms": 1e3 * kernel_s, "flop_per_solve": FLOP_PER_SOLVE,
                         "hbm": {"achieved": kernel_rate * BYTES_PER_SOLVE / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": kernel_rate * BYTES_PER_SOLVE / 1e9 / HBM_PEAK_GBS,
                                 "bytes_per_solve": BYTES_PER_SOLVE}},
            "solver_status_nonzero": int(stats["bad"]),
            "final_err_max": {"pos": stats["pos"], "ori": stats["ori"]},
        }
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            with open(tfile) as f:
                tj = json.load(f)
            res["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
            res["roofline"]["traffic_note"] = tj.get("note")
        if n_gpus == 1 and not args.no_extra:
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
                                                  "kkt": kkt_residual(p2, q2, t2, v * W.DT)}
            # benchmark config 3: 65536 problems x 50 iterated steps, then the early-exit run
            q3 = [q0[:65536].clone() for _ in range(6)]
            t3 = targets[:65536].contiguous()
            o3 = dict(iters=out["iters"][:65536], err=out["err"][:65536], status=out["status"][:65536])
            w3, e3 = time_launches(lambda i: prob.iterate_(q3[i], t3, W.DT, W.DAMPING, ITERS, out=o3), 5, 1, barrier)
            extra["config3_B65536_x50"] = {"solves_per_s": 65536 * ITERS * 5 / w3, "ms_per_launch": 1e3 * e3 / 5}
            qe = q0[:65536].clone()
            r = prob.iterate_(qe, t3, W.DT, W.DAMPING, ITERS, 1e-4, 1e-4)
            torch.cuda.synchronize()
            it = r["iters"].double()
            conv = (r["err"][:, :3].norm(dim=1) <= 1e-4) & (r["err"][:, 3:].norm(dim=1) <= 1e-4)
            extra["config3_early_exit"] = {"fraction_converged": float(conv.double().mean().item()),
                                           "mean_solves": float(it.mean().item()),
                                           "max_solves": int(it.max().item()),
                                           "final_err_max": {"pos": float(r["err"][:, :3].norm(dim=1).max().item()),
                                                             "ori": float(r["err"][:, 3:].norm(dim=1).max().item())},
                                           "note": "far targets at 0.5 m/s, dt = 1/200: most need more than 50 steps"}
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
            # primitive pairs {pads} x {cube, floor}; served by the wave-per-problem kernels
            t4, l4 = W.config4_specs(model)
            p4 = DeviceProblem(dm, t4, l4)
            B4 = 262144
            q4, tg4 = W.config4_inputs(model, dm, B4, 3)
            v4 = torch.empty((B4, model.nv), dtype=torch.float64, device=device)
            s4 = torch.empty(B4, dtype=torch.int32, device=device)
            w4, e4 = time_launches(lambda i: p4.solve(q4, tg4, W.DT, W.DAMPING, v=v4, status=s4), 3, 1, barrier)
            extra["config4_B262144_collision_relative_frame"] = {
                "solves_per_s": B4 * 3 / w4, "ms_per_launch": 1e3 * e4 / 3, "kernel": "dense_kernel (one problem per wave)",
                "status_nonzero": int((s4 != 0).sum().item())}
            del q4, tg4, v4, s4
            # kinematic move_to rollouts (waypoint generator + the callers' IK loop per waypoint step)
            import mink_amd as mk
            Br = 65536
            rq = q0[:Br].clone()
            rcfg = mk.Configuration(model, rq, device=device)
            start = rcfg.get_transform_frame_to_world("pinch_site", "site")
            g = torch.Generator(device="cpu").manual_seed(SEED)
            goal_pos = start.translation() + (torch.rand((Br, 3), generator=g, dtype=torch.float64) * 0.3 - 0.15).to(device)
            goal_rot = start.rotation() @ mk.SO3.exp((torch.rand((Br, 3), generator=g, dtype=torch.float64) * 0.8 - 0.4).to(device))
            ftask = mk.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=1.0)
            ptask = mk.PostureTask(model, cost=W.posture_cost(model))
            ptask.set_target(W.home_q(model))
            wcfg = mk.LinearWaypointReachConfig(0.01, 0.08, 0.05, 0.3)
            mk.move_to(mk.Configuration(model, rq[:4096].clone(), device=device), ftask, [ftask, ptask],
                       goal_pos[:4096], mk.SO3(goal_rot.wxyz[:4096]), wcfg, max_steps=50)   # warm-up
            torch.cuda.synchronize()
            tr = time.perf_counter()
            ro = mk.move_to(rcfg, ftask, [ftask, ptask], goal_pos, goal_rot, wcfg, max_steps=50)
            torch.cuda.synchronize()
            tr = time.perf_counter() - tr
            extra["move_to_rollouts_B65536"] = {
                "rollouts_per_s": Br / tr, "seconds": tr, "fraction_reached": float(ro["reached"].double().mean().item()),
                "mean_waypoint_steps": float(ro["steps"].double().mean().item()),
                "ik_solves_per_s": float(ro["solves"].sum().item()) / tr,
                "note": "kinematic part of the reference's move_to (waypoint generator + <=20-iteration IK loop per "
                        "step, default ConfigurationLimit) as one launch (mk_move_to_batch), physics excluded"}
            res["other_workloads"] = extra
        if n_gpus == 1 and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            cb = cpu_baseline(model, q0_np, targets.cpu().numpy(), qbufs[-1].cpu().numpy())
            if cb is not None:
                res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
