"""The N-rank path of bench.py as the driver launches it: `--gpus 2` started as a fresh
child process (never an exec of this one), which spawns its two ranks itself; the ranks shard
the fixed global batch by problem (mink_amd/sharding.py), reduce their statistics and gather the
final configurations.  On a one-GPU box the two ranks share the device and the host-side
collectives run over gloo; on a multi-GPU box each rank owns a device and they run over RCCL.
The job must be the same job at either rank count: same totals, same solve counts, same errors."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def run_bench(gpus):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--total", "131072", "--steps", "2",
           "--warmup", "1", "--no-extra", "--no-cpu-baseline"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)        # not under a launcher: bench.py starts its ranks itself
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_ranks_are_the_same_job_as_one():
    one = run_bench(1)
    two = run_bench(2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "strong"
    for r in (one, two):
        assert r["config"]["problems_total"] == 131072
        assert r["solver_status_nonzero"] == 0
        assert r["steps"] == 2 and r["warmup"] == 1
    assert two["config"]["problems_per_gpu"] == 65536
    assert two["solves_executed_per_step"] == one["solves_executed_per_step"] == 131072 * 50
    assert two["final_err_max"] == one["final_err_max"]         # max over the same rows, bit for bit
    assert two["config"]["collective_backend"] in ("rccl", "gloo")
    assert "gather_q_final_ms" in two and two["gather_q_final_ms"] > 0.0
    assert "gather_q_final_ms" not in one
    assert two["value"] > 0.0 and one["value"] > 0.0
