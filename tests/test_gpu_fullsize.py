"""GPU tests at the benchmark's full batch sizes, where the CPU oracle is too slow to be the
checker: size-independent properties of the path.

* batch invariance: a problem's result does not depend on how many problems are solved
  with it or where it sits in the batch (bit-identical), which ties every row of the big
  launches to the small batches the oracle does check;
* KKT conditions of the unique minimiser on a strided sample of the big batch, evaluated on
  the dense (P, c, G, h) the library exports (mink/solve_ik.py:43-65);
* the limits hold after the iterated loop (ConfigurationLimit with gain <= 1 keeps a
  feasible joint inside its range; VelocityLimit bounds every step);
* a configuration that already satisfies its tasks does not move (the reference's
  tests/test_solve_ik.py:74-93).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.mjcf import JNT_FREE, load_builtin

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def env():
    from mink_amd.engine import DeviceModel
    model = load_builtin("tidybot_cube")
    return model, DeviceModel(model, DEV)


def inputs(model, dm, n, seed):
    rng = np.random.default_rng(seed)
    q = torch.tensor(W.sample_q(model, rng, n), device=DEV)
    tg = dm.frame_pose(torch.tensor(W.sample_q(model, rng, n), device=DEV), L.FRAME_TYPES["site"],
                       model.site_id("pinch_site"))
    return q, tg


def kkt(P, c, G, h, x):
    """(stationarity residual, most negative multiplier, primal infeasibility) of x for
    min 1/2 x'Px + c'x s.t. Gx <= h, with the multipliers of the active rows fitted by least
    squares (numpy, one problem at a time)."""
    P, c, G, h, x = (t.cpu().numpy() for t in (P, c, G, h, x))
    stat = np.zeros(len(x)); neg = np.zeros(len(x)); infeas = np.zeros(len(x))
    for b in range(len(x)):
        g = P[b] @ x[b] + c[b]
        s = h[b] - G[b] @ x[b]
        infeas[b] = max(0.0, -s.min())
        act = s <= 1e-11 * np.maximum(np.abs(h[b]), 1.0)
        if act.any():
            A = G[b][act].T
            lam = np.linalg.lstsq(A, -g, rcond=None)[0]
            stat[b] = np.abs(g + A @ lam).max()
            neg[b] = max(0.0, -lam.min())
        else:
            stat[b] = np.abs(g).max()
    return stat, neg, infeas


def test_config3_size_batch_invariance_kkt_and_limits(env):
    """65536 problems x 50 iterated steps (benchmark config 3) and its single step."""
    from mink_amd.engine import DeviceProblem
    model, dm = env
    B, K = 65536, 50
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    assert prob.fused
    q0, tg = inputs(model, dm, B, 2)
    # single step on everything, and on a slice in the middle of the batch
    v, st = prob.solve(q0, tg, W.DT, W.DAMPING)
    assert int(st.abs().sum()) == 0
    lo, n = 20011, 4099            # odd offset and size: not aligned with the 64-problem blocks
    # (a batch of this size alone would go to the cooperative kernels: bit-equality is a property
    # of one kernel form, equality to rounding holds across the two)
    with L.options(coop=0):
        v_s, st_s = prob.solve(q0[lo:lo + n].contiguous(), tg[lo:lo + n].contiguous(), W.DT, W.DAMPING)
    assert torch.equal(v_s, v[lo:lo + n])
    v_c, st_c = prob.solve(q0[lo:lo + n].contiguous(), tg[lo:lo + n].contiguous(), W.DT, W.DAMPING)
    assert float((v_c - v_s).abs().max()) <= 1e-9 and int(st_c.abs().sum()) == 0
    # KKT of the minimiser on a strided sample
    idx = torch.arange(0, B, 37, device=DEV)[:1536]
    P, c, G, h = prob.build_qp(q0[idx].contiguous(), tg[idx].contiguous(), W.DT, W.DAMPING)
    stat, neg, infeas = kkt(P, c, G, h, v[idx] * W.DT)
    scale = max(float(c.abs().max()), 1.0)
    assert stat.max() <= 1e-9 * scale and neg.max() <= 1e-9 * scale, (stat.max(), neg.max())
    assert infeas.max() <= 1e-12
    # the iterated loop: batch invariance, limits and progress
    q = q0.clone()
    res = prob.iterate_(q, tg, W.DT, W.DAMPING, K, want_v=True)
    assert int(res["status"].abs().sum()) == 0 and int((res["iters"] != K).sum()) == 0
    qs = q0[lo:lo + n].clone()
    with L.options(coop=0):
        res_s = prob.iterate_(qs, tg[lo:lo + n].contiguous(), W.DT, W.DAMPING, K, want_v=True)
    assert torch.equal(qs, q[lo:lo + n]) and torch.equal(res_s["err"], res["err"][lo:lo + n])
    qc = q0[lo:lo + n].clone()
    prob.iterate_(qc, tg[lo:lo + n].contiguous(), W.DT, W.DAMPING, K)      # cooperative kernels
    assert float((qc - qs).abs().max()) <= 1e-9
    rng_lo = torch.tensor(model.jnt_range[:, 0], device=DEV)
    rng_hi = torch.tensor(model.jnt_range[:, 1], device=DEV)
    for j in range(model.njnt):
        if model.jnt_type[j] == JNT_FREE or not model.jnt_limited[j]:
            continue
        a = int(model.jnt_qposadr[j])
        assert float((q[:, a] - rng_hi[j]).max()) <= 1e-12 and float((rng_lo[j] - q[:, a]).max()) <= 1e-12
    vmax = torch.full((model.nv,), float("inf"), dtype=torch.float64, device=DEV)
    for name, lim in W.VELOCITY_LIMITS.items():
        vmax[int(model.jnt_dofadr[model.joint_id(name)])] = lim
    assert float((res["v"].abs() - vmax).max()) <= 1e-9
    e0 = prob.task_eval(0, 6, q0, tg)[0].norm(dim=1)
    e1 = res["err"].norm(dim=1)
    frac = float((e1 <= e0 + 1e-9).double().mean())
    print("fraction of problems closer to the target after the loop:", frac)
    assert frac > 0.99   # the loop approaches its targets


def test_config4_size_batch_invariance_and_kkt(env):
    """262144 problems with RelativeFrameTask and CollisionAvoidanceLimit (benchmark config 4):
    the wave-per-problem kernels."""
    from mink_amd.engine import DeviceProblem
    model, dm = env
    B = 262144
    tasks, limits = W.config4_specs(model)
    prob = DeviceProblem(dm, tasks, limits)
    assert not prob.fused
    q, tg = W.config4_inputs(model, dm, B, 3)
    v, st = prob.solve(q, tg, W.DT, W.DAMPING)
    assert int(st.abs().sum()) == 0
    lo, n = 1003, 2051             # inside the part of the batch with the cube next to a pad
    v_s, _ = prob.solve(q[lo:lo + n].contiguous(), tg[lo:lo + n].contiguous(), W.DT, W.DAMPING)
    assert torch.equal(v_s, v[lo:lo + n])
    idx = torch.cat([torch.arange(0, 65536, 61, device=DEV), torch.arange(65536, B, 331, device=DEV)])[:1536]
    P, c, G, h = prob.build_qp(q[idx].contiguous(), tg[idx].contiguous(), W.DT, W.DAMPING)
    finite = torch.isfinite(h)
    assert int((finite[:, -8:]).sum()) > 50          # collision rows are active in the sample
    hh = torch.where(finite, h, torch.full_like(h, 1e30))
    stat, neg, infeas = kkt(P, c, G, hh, v[idx] * W.DT)
    scale = max(float(c.abs().max()), 1.0)
    assert stat.max() <= 1e-8 * scale and neg.max() <= 1e-8 * scale, (stat.max(), neg.max())
    assert infeas.max() <= 1e-10


def test_config5_shard_size_fixed_point_and_invariance(env):
    """131072 problems (one GPU's share of the 2^20 batch): a configuration whose frame is on
    its target (posture cost zero) stays where it is, for every row, and the iterated launch
    of the headline problem is batch invariant at this size too."""
    from mink_amd.engine import DeviceProblem
    model, dm = env
    B = 131072
    rng = np.random.default_rng(11)
    qn = W.sample_q(model, rng, B)
    q = torch.tensor(qn, device=DEV)
    tg = dm.frame_pose(q, L.FRAME_TYPES["site"], model.site_id("pinch_site"))
    # posture task towards each row's own q is not expressible (shared target): give it zero cost
    specs = W.task_specs(model, False)
    specs[1] = dict(specs[1], cost=np.zeros(model.nv))
    prob = DeviceProblem(dm, specs, W.limit_specs(model, True))
    q1 = q.clone()
    res = prob.iterate_(q1, tg, W.DT, W.DAMPING, 3, want_v=True)
    assert int(res["status"].abs().sum()) == 0
    assert float(res["v"].abs().max()) <= 1e-9
    assert float((q1 - q).abs().max()) <= 1e-12
    # headline problem at the headline size
    prob5 = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    q0, tg0 = inputs(model, dm, B, 4000)
    qa = q0.clone()
    ra = prob5.iterate_(qa, tg0, W.DT, W.DAMPING, 50)
    lo, n = 77777, 1000
    qb = q0[lo:lo + n].clone()
    with L.options(coop=0):
        prob5.iterate_(qb, tg0[lo:lo + n].contiguous(), W.DT, W.DAMPING, 50)
    assert int(ra["status"].abs().sum()) == 0
    assert torch.equal(qb, qa[lo:lo + n])
    qc = q0[lo:lo + n].clone()
    prob5.iterate_(qc, tg0[lo:lo + n].contiguous(), W.DT, W.DAMPING, 50)   # cooperative kernels
    assert float((qc - qb).abs().max()) <= 1e-9


def test_config5_full_batch_one_launch_and_shard_invariance(env):
    """The whole 1 048 576-problem batch of benchmark config 5 (seed 4, the rows bench.py
    uses) x 50 iterated steps as ONE launch on one GPU; the rows that rank 5 of 8 would own
    (sharding.shard_bounds), solved alone from inputs generated for that shard only, are
    bit-identical -- which is what makes the N-GPU runs the same job as the 1-GPU run."""
    from mink_amd import sharding as S
    from mink_amd.engine import DeviceProblem
    model, dm = env
    total, K, seed = 1 << 20, 50, 4
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    assert prob.fused
    site = model.site_id("pinch_site")
    q0, qs = S.global_rows(model, seed, 0, total)
    q = torch.tensor(q0, device=DEV)
    tg = dm.frame_pose(torch.tensor(qs, device=DEV), L.FRAME_TYPES["site"], site)
    res = prob.iterate_(q, tg, W.DT, W.DAMPING, K)
    assert int(res["status"].abs().sum()) == 0 and int((res["iters"] != K).sum()) == 0
    assert bool(torch.isfinite(q).all())
    lo, hi = S.shard_bounds(total, 5, 8)
    assert (lo, hi) == (5 * 131072, 6 * 131072)
    q0s, qss = S.global_rows(model, seed, lo, hi)
    np.testing.assert_array_equal(q0s, q0[lo:hi])
    qsh = torch.tensor(q0s, device=DEV)
    tgs = dm.frame_pose(torch.tensor(qss, device=DEV), L.FRAME_TYPES["site"], site)
    rs = prob.iterate_(qsh, tgs, W.DT, W.DAMPING, K)
    assert torch.equal(qsh, q[lo:hi]) and torch.equal(rs["err"], res["err"][lo:hi])
    # ragged split (world 3) covers every row exactly once
    b3 = [S.shard_bounds(total, r, 3) for r in range(3)]
    assert b3[0][0] == 0 and b3[-1][1] == total and all(b3[i][1] == b3[i + 1][0] for i in range(2))
