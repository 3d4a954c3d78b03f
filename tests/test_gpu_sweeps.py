"""Small editions of the parameter sweeps under tools/ (the full ones are run by hand on the GPU box
and kept as records): random task / limit parameters against the C restatement, collision rows of
convex pairs against the numpy restatement, random QPs with KKT residuals computed on the host.  Each
runs its tool in a child process and reads the summary it prints."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def run_tool(name, *args):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", name), *map(str, args)], env=env, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    return r.stdout


def rows(text):
    return [[c.strip() for c in l.strip().strip("|").split("|")] for l in text.splitlines() if re.match(r"\| \d+ \|", l)]


def test_random_parameter_sets_against_the_c_restatement():
    out = run_tool("soak_parity.py", 18, 2048)
    table = rows(out)
    assert len(table) == 18
    for c in table:
        damping, lm = float(c[3]), float(c[4].split("/")[1])
        assert int(c[15]) == 0 and int(c[16]) == 0, c          # solve counts and status agree
        if damping >= 1e-3 or lm > 0.0:
            # one step (rad/s), the <= 10-step loop (relative to the distance moved), general kernels vs fused
            assert float(c[10]) < 1e-7 and float(c[14]) < 1e-6 and float(c[17]) < 1e-7 and int(c[18]) == 0, c


def test_collision_rows_of_convex_and_primitive_pairs():
    out = run_tool("soak_convex.py", 96)
    s = {m.group(1): (int(m.group(2)), float(m.group(3))) for m in re.finditer(
        r"^(.*?): \d+ configurations .*? in-range pattern differs on (\d+) rows; max \|dG\| ([0-9.e+-]+)", out, re.M)}
    assert set(s) == {"TidyBot mesh hulls", "cylinders / ellipsoids", "closed-form primitive pairs"}, out[-1500:]
    assert s["TidyBot mesh hulls"][0] == 0 and s["TidyBot mesh hulls"][1] < 1e-9
    assert s["cylinders / ellipsoids"][0] == 0 and s["cylinders / ellipsoids"][1] < 1e-3
    assert s["closed-form primitive pairs"][0] == 0 and s["closed-form primitive pairs"][1] < 1e-9
    cert = re.findall(r"distance\): (\d+) of (\d+)", out) or re.findall(r"to 1e-7\): (\d+) of (\d+)", out)
    assert cert and all(int(b) - int(a) <= max(1, int(b) // 500) for a, b in cert), cert


def test_random_qps_with_host_side_kkt_residuals():
    out = run_tool("soak_qp.py", 512)
    table = [[c.strip() for c in l.strip().strip("|").split("|")] for l in out.splitlines() if re.match(r"\| \d+ \| \d+ \|", l)]
    assert len(table) == 45
    for c in table:
        n, m, cond, bad, infeas, stat, comp = int(c[0]), int(c[1]), float(c[2]), int(c[3]), float(c[4]), float(c[5]), float(c[6])
        assert infeas < 1e-6 and stat < 1e-9 and comp < 1e-9, c       # whatever is handed out as solved is a KKT point
        if cond <= 1e6 or m <= n:
            assert bad == 0, c                                          # and these are all solved
