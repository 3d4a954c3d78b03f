"""The C ABI used from plain C (tests/cabi/cabi_check.c: HIP runtime for memory, no Python
objects, no torch): fused solve against the dense build + generic QP route, the iterated
loop, error returns.  The binary is built by __graft_entry__.build()."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cabi", "cabi_check")


def test_c_consumer_of_the_abi():
    if not os.path.exists(BIN):
        pytest.skip("tests/cabi/cabi_check is missing: run __graft_entry__.build()")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi_check ok" in r.stdout
