"""The multi-process layout on CPU (gloo, world_size 2): shard bounds, world-size
independent inputs, statistics reduction and result gather.  The per-shard solve is
done by the host build of the device code (tests/hostsim), so the two-rank answer can
be compared with the one-process answer bit for bit."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

from mink_amd import sharding as S
from mink_amd import workloads as W
from mink_amd.mjcf import load_builtin

TOTAL = 301      # odd on purpose: ragged shards
SEED = 7


def test_shard_bounds_partition():
    for total in (0, 1, 7, 64, 301, 1 << 20):
        for world in (1, 2, 3, 8):
            b = [S.shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        S.shard_bounds(10, 2, 2)


def test_rows_do_not_depend_on_sharding():
    model = load_builtin("tidybot_cube")
    q0, qs = S.global_rows(model, SEED, 0, 9000)
    for lo, hi in ((0, 1), (4000, 4200), (4096, 8192), (8999, 9000), (5, 5)):
        a, b = S.global_rows(model, SEED, lo, hi)
        np.testing.assert_array_equal(a, q0[lo:hi])
        np.testing.assert_array_equal(b, qs[lo:hi])
    assert not np.array_equal(q0[:10], S.global_rows(model, SEED + 1, 0, 10)[0])


def _solve_shard(model, lo, hi):
    import hostsim_lib as H
    from oracle import ik_oracle as o
    q0, qs = S.global_rows(model, SEED, lo, hi)
    targets = np.array([o.Configuration(model, q).get_transform_frame_to_world("pinch_site", "site") for q in qs])
    return H.solve_ik(model, W.task_specs(model, False), W.limit_specs(model, True), q0, targets.reshape(-1, 7),
                      W.DT, W.DAMPING)


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = load_builtin("tidybot_cube")
        lo, hi = S.shard_bounds(TOTAL, rank, world)
        v, status = _solve_shard(model, lo, hi)
        stats = S.reduce_stats({"solves": hi - lo, "bad": int((status != 0).sum())},
                               {"vmax": float(np.abs(v).max()), "rank_max": rank}, dist)
        allv = S.gather_rows(torch.tensor(v), TOTAL, dist)
        if rank == 0:
            np.savez(os.path.join(out_dir, "out.npz"), v=allv.numpy(), **stats)
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_one_process(tmp_path):
    import hostsim_lib as H
    if not H.available():
        pytest.skip("host build of the device code not built")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(tmp_path, "out.npz"))
    model = load_builtin("tidybot_cube")
    v, status = _solve_shard(model, 0, TOTAL)
    np.testing.assert_array_equal(got["v"], v)
    assert int(got["solves"]) == TOTAL and int(got["bad"]) == 0 and int(got["rank_max"]) == 1
    assert float(got["vmax"]) == float(np.abs(v).max())
