"""The lane-streaming form of the callers' loop (persistent waves whose lanes take the next
problem of their chunk as soon as their own has left the loop) against the one-block-per-64-
problems form: the same problems must come out bit-identical, whatever the number of
persistent waves, the chunk size and the company a problem runs in."""
import os

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
    from mink_amd.engine import DeviceModel
    model = load_builtin("tidybot_cube")
    return model, DeviceModel(model, DEV)


def near_inputs(model, dm, n, seed):
    rng = np.random.default_rng(seed)
    q = W.sample_q(model, rng, n)
    poses = dm.frame_pose(torch.tensor(q, device=DEV), L.FRAME_TYPES["site"], model.site_id("pinch_site")).cpu().numpy()
    return torch.tensor(q, device=DEV), torch.tensor(W.offset_targets(poses, rng), device=DEV)


def run(prob, q0, tg, max_iters, knob):
    # (coop = 0: this module compares the two lane-per-problem kernels with each other)
    with L.options(stream_blocks=int(knob), coop=0):
        q = q0.clone()
        res = prob.iterate_(q, tg, W.DT, W.DAMPING, max_iters, 1e-4, 1e-4, want_v=True)
        torch.cuda.synchronize()
    return q, res


@pytest.mark.parametrize("velocity", [False, True])
def test_stream_kernel_is_bit_identical(env, velocity):
    from mink_amd.engine import DeviceProblem
    model, dm = env
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, velocity))
    assert prob.fused
    n = 20011
    q0, tg = near_inputs(model, dm, n, 21)
    q0[7] = float("nan")                        # one non-finite row: status 3, left untouched
    q_ref, ref = run(prob, q0, tg, 20, 0)
    assert int(ref["iters"].max()) <= 20 and int(ref["iters"].min()) >= 0
    assert int(ref["status"][7]) == 3 and int((ref["status"] != 0).sum()) == 1
    for blocks in (1, 37, 313, 1024):
        if blocks == 1:
            m = 300                              # one wave through 300 problems
            q_s, res = run(prob, q0[:m].contiguous(), tg[:m].contiguous(), 20, blocks)
            sl = slice(0, m)
        else:
            q_s, res = run(prob, q0, tg, 20, blocks)
            sl = slice(0, n)
        ok = torch.ones(q_s.shape[0], dtype=torch.bool, device=DEV)
        ok[7] = False                            # NaN != NaN
        assert torch.equal(q_s[ok], q_ref[sl][ok]), blocks
        for k in ("iters", "status"):
            assert torch.equal(res[k], ref[k][sl]), (blocks, k)
        assert torch.equal(res["err"][ok], ref["err"][sl][ok]), blocks
        assert torch.equal(res["v"][ok], ref["v"][sl][ok]), blocks
    # more persistent waves than problems: most chunks are empty
    q_s, res = run(prob, q0[:5].contiguous(), tg[:5].contiguous(), 20, 64)
    assert torch.equal(q_s, q_ref[:5]) and torch.equal(res["iters"], ref["iters"][:5])


def test_stream_path_is_taken_for_large_batches_and_matches(env):
    """Above twice the machine's lanes the library chooses the streaming kernel by itself; a
    slice solved by the block kernel is the same."""
    from mink_amd.engine import DeviceProblem
    model, dm = env
    prob = DeviceProblem(dm, W.task_specs(model, False), W.limit_specs(model, False))
    n = 1 << 18
    q0, tg = near_inputs(model, dm, n, 22)
    q = q0.clone()
    with L.options(coop=0):   # lane kernels alone (no hand-off of the tails to the cooperative kernel)
        res = prob.iterate_(q, tg, W.DT, W.DAMPING, 20, 1e-4, 1e-4)
    lo, m = 100003, 4099
    q_b, res_b = run(prob, q0[lo:lo + m].contiguous(), tg[lo:lo + m].contiguous(), 20, 0)
    assert torch.equal(q_b, q[lo:lo + m]) and torch.equal(res_b["iters"], res["iters"][lo:lo + m])
    conv = (res["err"][:, :3].norm(dim=1) <= 1e-4) & (res["err"][:, 3:].norm(dim=1) <= 1e-4)
    assert float(conv.double().mean()) > 0.999 and int(res["status"].abs().sum()) == 0
