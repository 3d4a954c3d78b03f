"""Edge cases of the batch entry points: empty batch, one problem, sizes around the
64-problem block, non-finite inputs and infeasible limits reported per problem
(status 3 / 2 with a zero velocity) without disturbing their neighbours."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from mink_amd import _lib as L
from mink_amd import workloads as W
from mink_amd.mjcf import load_builtin

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def env():
    from mink_amd.engine import DeviceModel, DeviceProblem
    model = load_builtin("tidybot_cube")
    dm = DeviceModel(model, DEV)
    fused = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, True))
    t4, l4 = W.config4_specs(model)
    dense = DeviceProblem(dm, t4, l4)
    return model, dm, fused, dense


def sample(model, dm, n, seed):
    rng = np.random.default_rng(seed)
    q = torch.tensor(W.sample_q(model, rng, n), device=DEV).reshape(n, model.nq)
    tg = dm.frame_pose(torch.tensor(W.sample_q(model, rng, max(n, 1)), device=DEV)[:n].reshape(n, model.nq).contiguous(),
                       L.FRAME_TYPES["site"], model.site_id("pinch_site")) if n else \
        torch.empty((0, 7), dtype=torch.float64, device=DEV)
    return q, tg


def test_empty_batch_everywhere(env):
    model, dm, fused, dense = env
    q = torch.empty((0, model.nq), dtype=torch.float64, device=DEV)
    tg = torch.empty((0, 7), dtype=torch.float64, device=DEV)
    v, st = fused.solve(q, tg, W.DT, W.DAMPING)
    assert v.shape == (0, model.nv) and st.shape == (0,)
    res = fused.iterate_(q, tg, W.DT, W.DAMPING, 5)
    assert res["iters"].shape == (0,)
    v, st = dense.solve(q, torch.empty((0, 14), dtype=torch.float64, device=DEV), W.DT, W.DAMPING)
    assert v.shape == (0, model.nv)
    assert dm.frame_pose(q, L.FRAME_TYPES["site"], 0).shape == (0, 7)
    assert dm.frame_jacobian(q, L.FRAME_TYPES["site"], 0).shape == (0, 6, model.nv)
    dm.integrate_(q, torch.empty((0, model.nv), dtype=torch.float64, device=DEV), W.DT)
    assert dm.first_limit_violation(q, 1e-6).shape == (0,)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 129])
def test_sizes_around_the_block(env, n):
    """Every size gives the rows of the big batch, bit for bit (fused and dense paths)."""
    model, dm, fused, dense = env
    q, tg = sample(model, dm, 200, 5)
    v_all, st_all = fused.solve(q, tg, W.DT, W.DAMPING)
    v, st = fused.solve(q[:n].contiguous(), tg[:n].contiguous(), W.DT, W.DAMPING)
    assert torch.equal(v, v_all[:n]) and int(st.abs().sum()) == 0
    tg2 = torch.cat([tg, tg], dim=1).contiguous()
    d_all, _ = dense.solve(q, tg2, W.DT, W.DAMPING)
    d, st = dense.solve(q[:n].contiguous(), tg2[:n].contiguous(), W.DT, W.DAMPING)
    assert torch.equal(d, d_all[:n]) and int(st.abs().sum()) == 0


def test_non_finite_and_infeasible_rows_are_flagged_alone(env):
    model, dm, fused, dense = env
    q, tg = sample(model, dm, 130, 6)
    v_ref, _ = fused.solve(q, tg, W.DT, W.DAMPING)
    qb, tgb = q.clone(), tg.clone()
    qb[3, 5] = float("nan")                  # a chain coordinate
    tgb[70, 4] = float("inf")                # a target position
    qb[40, 12] = float("nan")                # a finger joint: outside the chain
    a2 = int(model.jnt_qposadr[model.joint_id("joint_2")])
    qb[100, a2] = model.jnt_range[model.joint_id("joint_2"), 1] + 2.0   # far outside its limit: lower > upper bound
    v, st = fused.solve(qb, tgb, W.DT, W.DAMPING)
    st = st.cpu().numpy()
    assert st[3] == 3 and st[70] == 3 and st[40] == 3 and st[100] == 2
    good = np.ones(130, dtype=bool)
    good[[3, 40, 70, 100]] = False
    assert not st[good].any()
    assert torch.equal(v[torch.tensor(good, device=DEV)], v_ref[torch.tensor(good, device=DEV)])
    assert float(v[3].abs().max()) == 0.0 and float(v[70].abs().max()) == 0.0 and float(v[40].abs().max()) == 0.0
    assert bool(torch.isfinite(v).all())
    # the loop keeps flagged problems where they are and leaves the others untouched
    qi = qb.clone()
    res = fused.iterate_(qi, tgb, W.DT, W.DAMPING, 3)
    sti = res["status"].cpu().numpy()
    assert sti[3] == 3 and sti[70] == 3 and sti[40] == 3 and sti[100] == 2 and not sti[good].any()
    assert torch.equal(qi[70], qb[70])      # a flagged problem with finite q stays where it was
    qr = q.clone()
    fused.iterate_(qr, tg, W.DT, W.DAMPING, 3)
    assert torch.equal(qi[torch.tensor(good, device=DEV)], qr[torch.tensor(good, device=DEV)])


def test_dense_path_flags_non_finite_rows(env):
    model, dm, fused, dense = env
    q, tg = sample(model, dm, 70, 8)
    tg2 = torch.cat([tg, tg], dim=1).contiguous()
    v_ref, _ = dense.solve(q, tg2, W.DT, W.DAMPING)
    qb, tgb = q.clone(), tg2.clone()
    qb[5, 4] = float("nan")
    tgb[33, 9] = float("inf")
    qb[50, 20] = float("nan")                # the cube's free joint
    v, st = dense.solve(qb, tgb, W.DT, W.DAMPING)
    st = st.cpu().numpy()
    good = np.ones(70, dtype=bool)
    good[[5, 33, 50]] = False
    assert (st[~good] == 3).all() and not st[good].any()
    assert bool(torch.isfinite(v).all()) and float(v[~torch.tensor(good, device=DEV)].abs().max()) == 0.0
    assert torch.equal(v[torch.tensor(good, device=DEV)], v_ref[torch.tensor(good, device=DEV)])
