"""A tree whose q row is long (a 12-dof chain to the task frame plus 44 joints beside it):
the fused kernels then need more than the default 64 KB of dynamic LDS per block.  GPU
result against the numpy oracle."""
import os
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from mink_amd import _lib as L
from mink_amd.mjcf import load_mjcf


def long_tree_xml(n_chain=12, n_side=44):
    rng = np.random.default_rng(0)
    out = ['<mujoco model="long_rows"><compiler angle="radian"/><worldbody>']
    for k in range(n_chain):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        out.append(f'<body name="c{k}" pos="0.03 {0.02 * (k % 3)} 0.11"><joint name="jc{k}" type="hinge" '
                   f'axis="{ax[0]:.6f} {ax[1]:.6f} {ax[2]:.6f}" range="-2.5 2.5"/>'
                   f'<geom type="sphere" size="0.02" mass="0.3"/>')
        if k == n_chain // 2:
            # side branches hang off the middle of the chain: their dofs are outside it
            for j in range(n_side):
                out.append(f'<body name="s{j}" pos="0.05 {0.01 * j} 0"><joint name="js{j}" type="hinge" axis="0 1 0" '
                           f'range="-1 1"/><geom type="sphere" size="0.01" mass="0.05"/></body>')
    out.append('<site name="tip" pos="0.02 0.01 0.08" quat="0.9 0.1 0.3 0.2"/>')
    out.append("</body>" * n_chain)
    out.append("</worldbody></mujoco>")
    return "\n".join(out)


@pytest.mark.gpu
def test_fused_path_with_more_than_64kb_of_lds():
    from mink_amd.engine import DeviceModel, DeviceProblem
    from oracle import ik_oracle as o
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "long_rows.xml")
        with open(path, "w") as f:
            f.write(long_tree_xml())
        model = load_mjcf(path)
    assert model.nq == 56 and model.nv == 56
    dm = DeviceModel(model, "cuda:0")
    cost = np.full(model.nv, 1e-2)
    tasks = [dict(type=L.TASK_FRAME, frame_type=L.FRAME_TYPES["site"], frame_id=model.site_id("tip"),
                  cost=np.ones(6), gain=1.0, lm_damping=0.5),
             dict(type=L.TASK_POSTURE, cost=cost, target_q=np.zeros(model.nq), gain=1.0, lm_damping=0.0)]
    limits = [dict(type=L.LIMIT_CONFIGURATION, gain=0.95, min_distance_from_limits=0.0)]
    prob = DeviceProblem(dm, tasks, limits)
    assert prob.fused and prob.n_chain == 12
    assert 64 * ((model.nq | 1) + 78) * 8 > 64 * 1024
    rng = np.random.default_rng(1)
    B, dt, damping = 130, 0.01, 1e-3
    q = rng.uniform(-0.9, 0.9, (B, model.nq))
    qs = rng.uniform(-0.9, 0.9, (B, model.nq))
    tg = np.array([o.Configuration(model, x).get_transform_frame_to_world("tip", "site") for x in qs])
    v, st = prob.solve(torch.tensor(q, device="cuda:0"), torch.tensor(tg, device="cuda:0"), dt, damping)
    assert int(st.abs().sum()) == 0
    for b in range(0, B, 13):
        otasks = [o.FrameTask("tip", "site", 1.0, 1.0, lm_damping=0.5, target=tg[b]),
                  o.PostureTask(cost=cost, target_q=np.zeros(model.nq))]
        ref = o.solve_ik(o.Configuration(model, q[b]), otasks, dt, damping=damping, limits=[o.ConfigurationLimit(model)])
        np.testing.assert_allclose(v[b].cpu().numpy(), ref, rtol=0, atol=1e-9)
    # the iterated launch uses the same tile
    qi = torch.tensor(q, device="cuda:0")
    res = prob.iterate_(qi, torch.tensor(tg, device="cuda:0"), dt, damping, 5)
    assert int(res["status"].abs().sum()) == 0 and int((res["iters"] != 5).sum()) == 0
