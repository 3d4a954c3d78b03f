"""Host-side behaviour of the mink-named API that needs no GPU: constructor validation
with the reference's exception classes and messages (captured from the reference into
tests/golden/api_strings.json), limit index tables, collision pair construction, and the
exported C ABI."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import mink_amd
from mink_amd import _lib as L
from mink_amd.mjcf import load_builtin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model():
    return load_builtin("tidybot_cube")


@pytest.fixture(scope="module")
def strings(golden_dir):
    with open(os.path.join(golden_dir, "api_strings.json")) as f:
        return json.load(f)


def _raised(fn):
    try:
        fn()
    except Exception as e:  # noqa: BLE001
        return [type(e).__name__, str(e)]
    return None


def test_constructor_validation_strings(model, strings):
    m = mink_amd
    cases = {
        "frame_neg_cost": lambda: m.FrameTask("pinch_site", "site", -1.0, 1.0),
        "frame_bad_cost_shape": lambda: m.FrameTask("pinch_site", "site", [1.0, 2.0], 1.0),
        "frame_bad_ori_shape": lambda: m.FrameTask("pinch_site", "site", 1.0, [1.0, 2.0]),
        "task_bad_gain": lambda: m.FrameTask("pinch_site", "site", 1.0, 1.0, gain=1.5),
        "task_bad_damping": lambda: m.FrameTask("pinch_site", "site", 1.0, 1.0, lm_damping=-1.0),
        "posture_bad_cost_shape": lambda: m.PostureTask(model, cost=[1.0, 2.0]),
        "posture_neg_cost": lambda: m.PostureTask(model, cost=-1.0),
        "posture_bad_target": lambda: m.PostureTask(model, cost=1.0).set_target(np.zeros(3)),
        "com_bad_cost_shape": lambda: m.ComTask(cost=[1.0, 2.0]),
        "com_neg_cost": lambda: m.ComTask(cost=-1.0),
        "com_bad_target": lambda: m.ComTask(cost=1.0).set_target(np.zeros(2)),
        "cfg_limit_bad_gain": lambda: m.ConfigurationLimit(model, gain=0.0),
        "vel_limit_free": lambda: m.VelocityLimit(model, {"interactive_obj_freejoint": np.ones(6)}),
        "vel_limit_bad_shape": lambda: m.VelocityLimit(model, {"joint_1": [1.0, 2.0]}),
    }
    for name, fn in cases.items():
        assert _raised(fn) == strings[name], name


def test_limit_tables(model, golden_dir):
    g = np.load(os.path.join(golden_dir, "tidybot_build_ik.npz"))
    cfg = mink_amd.ConfigurationLimit(model)
    # rows of G_cfg are [+I; -I] restricted to the limited dofs
    assert 2 * len(cfg.indices) == g["G_cfg"].shape[1]
    np.testing.assert_array_equal(g["G_cfg"][0][: len(cfg.indices)], cfg.projection_matrix)
    vel = mink_amd.VelocityLimit(model, dict(zip(g["vel_names"].tolist(), g["vel_limits"].tolist())))
    np.testing.assert_array_equal(g["G_vel"][0][: len(vel.indices)], vel.projection_matrix)
    np.testing.assert_allclose(g["h_vel"][0][: len(vel.indices)], float(g["dt"]) * vel.limit)
    assert mink_amd.VelocityLimit(model)._spec(model) is None


def test_collision_pairs_match_reference(model, golden_dir):
    g = np.load(os.path.join(golden_dir, "collision.npz"))
    pads = ["right_pad1", "right_pad2", "left_pad1", "left_pad2"]
    lim = mink_amd.CollisionAvoidanceLimit(model, [(pads, ["cube_geom", "floor"])])
    np.testing.assert_array_equal(np.array(lim.geom_id_pairs), g["pairs"])
    # pads against themselves: same weld group -> filtered out
    assert mink_amd.CollisionAvoidanceLimit(model, [(pads[:2], pads[:2])]).geom_id_pairs == []


def test_utils(model):
    q_ids, v_ids = mink_amd.get_freejoint_dims(model)
    assert q_ids == list(range(18, 25)) and v_ids == list(range(18, 24))
    q = mink_amd.custom_configuration_vector(model, "home", joint_1=0.3)
    assert q[model.jnt_qposadr[model.joint_id("joint_1")]] == 0.3
    with pytest.raises(mink_amd.InvalidKeyframe):
        mink_amd.custom_configuration_vector(model, "nope")
    sub = mink_amd.get_subtree_body_ids(model, model.body_id("base_link"))
    assert model.body_id("left_pad") in sub and model.body_id("interactive_obj") not in sub


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly what the header declares."""
    header = open(os.path.join(ROOT, "include", "mink_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mk_[a-z_]+)\s*\(", header))
    assert declared == set(L.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.mk_last_error.restype = ctypes.c_char_p
    assert lib.mk_last_error() == b""
    # argument validation happens before any device work
    lib.mk_model_create.restype = ctypes.c_int
    assert lib.mk_model_create(None, None) == -1
    assert b"null" in lib.mk_last_error().lower()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(L, "_lib", None)
    with pytest.raises(ImportError, match="no CPU fallback"):
        L.load_library()
