"""The reference's small test models through the MJCF front end and the host-side limit /
helper classes (no GPU): tests/test_configuration_limit.py:57-121, tests/test_velocity_limit.py:
67-140, tests/test_utils.py:104-160 of the reference."""
import numpy as np
import pytest

import ref_models as R
from mink_amd import utils
from mink_amd.exceptions import LimitDefinitionError
from mink_amd.limits import ConfigurationLimit, VelocityLimit
from mink_amd.mjcf import JNT_BALL, JNT_FREE, JNT_HINGE, load_mjcf_string

MJ_MAXVAL = 1e10


def test_subset_of_joints_limited():
    model = load_mjcf_string(R.SUBSET_LIMITED)
    assert (model.nq, model.nv, model.njnt) == (2, 2, 2)
    limit = ConfigurationLimit(model)
    assert limit.projection_matrix.shape == (1, model.nv) and len(limit.indices) == 1
    np.testing.assert_allclose(limit.lower, [-MJ_MAXVAL, 0])
    np.testing.assert_allclose(limit.upper, [MJ_MAXVAL, 1.57])


def test_freejoint_ignored_by_configuration_limit():
    model = load_mjcf_string(R.FREEJOINT_AND_HINGE)
    assert (model.nq, model.nv) == (8, 7) and list(model.jnt_type) == [JNT_FREE, JNT_HINGE]
    limit = ConfigurationLimit(model)
    assert limit.projection_matrix.shape == (1, model.nv) and list(limit.indices) == [6]
    np.testing.assert_allclose(limit.lower, [-MJ_MAXVAL] * 7 + [0])
    np.testing.assert_allclose(limit.upper, [MJ_MAXVAL] * 7 + [1.57])
    q_ids, v_ids = utils.get_freejoint_dims(model)
    assert q_ids == list(range(7)) and v_ids == list(range(6))


def test_ball_joint_velocity_limit_and_messages():
    model = load_mjcf_string(R.BALL_AND_HINGE)
    assert list(model.jnt_type) == [JNT_BALL, JNT_HINGE] and (model.nq, model.nv) == (5, 4)
    # no <compiler angle>: the range is in degrees, and autolimits makes the hinge limited
    assert model.jnt_limited[1] and abs(model.jnt_range[1, 1] - np.deg2rad(1.57)) < 1e-15
    limit = VelocityLimit(model, {"ball": (np.pi, np.pi / 2, np.pi / 4), "hinge": (0.5,)})
    assert len(limit.indices) == 4 and limit.projection_matrix.shape == (4, model.nv)
    np.testing.assert_allclose(limit.limit, [np.pi, np.pi / 2, np.pi / 4, 0.5])
    with pytest.raises(LimitDefinitionError) as e:
        VelocityLimit(model, {"ball": (np.pi, np.pi / 2)})
    assert str(e.value) == "Joint ball must have a limit of shape (3,). Got: (2,)"
    fmodel = load_mjcf_string(R.FREEJOINT_AND_HINGE)
    with pytest.raises(LimitDefinitionError) as e:
        VelocityLimit(fmodel, {"floating": np.pi, "hinge": np.pi})
    assert str(e.value) == "Free joint floating is not supported"


def test_subtree_helpers():
    model = load_mjcf_string(R.SUBTREES)
    gid = model.geom_id
    assert set(utils.get_subtree_geom_ids(model, model.body_id("b1"))) == {gid("b1/g1"), gid("b1/g2"), gid("b2/g1")}
    assert set(utils.get_subtree_geom_ids(model, model.body_id("b3"))) == {gid("b3/g1"), gid("b4/g1")}
    assert utils.get_subtree_geom_ids(model, model.body_id("geomless")) == []
    assert set(utils.get_body_geom_ids(model, model.body_id("b1"))) == {gid("b1/g1"), gid("b1/g2")}
    assert set(utils.get_subtree_body_ids(model, model.body_id("b1"))) == {model.body_id("b1"), model.body_id("b2")}
    assert utils.get_body_body_ids(model, model.body_id("b3")) == [model.body_id("b4")]
    assert model.body_mass[model.body_id("geomless")] == pytest.approx(0.1)


def test_mocap_model_tables():
    model = load_mjcf_string(R.MOCAP)
    assert model.nmocap == 1 and model.body_mocapid[model.body_id("mocap")] == 0
    np.testing.assert_allclose(model.body_quat[model.body_id("mocap")], [np.sqrt(0.5), np.sqrt(0.5), 0, 0])
    np.testing.assert_allclose(model.qpos0[:7], [0.1, -0.1, 0, 1, 0, 0, 0])


def test_limited_ball_joint_fills_every_quaternion_slot(golden_dir):
    """ConfigurationLimit on a LIMITED ball joint against the reference's own constructor
    (golden from mink/limits/configuration_limit.py:41-62): lower / upper carry the range in all
    four qpos slots of the joint, and the three dofs are limited."""
    import os
    g = np.load(os.path.join(golden_dir, "limited_ball.npz"))
    model = load_mjcf_string(R.LIMITED_BALL_AND_HINGE)
    assert model.nq == 5 and model.nv == 4 and int(model.jnt_type[0]) == JNT_BALL
    limit = ConfigurationLimit(model, gain=float(g["gain"]), min_distance_from_limits=float(g["min_distance_from_limits"]))
    np.testing.assert_array_equal(limit.lower, g["lower"])
    np.testing.assert_array_equal(limit.upper, g["upper"])
    np.testing.assert_array_equal(limit.indices, g["indices"])
    np.testing.assert_array_equal(limit.projection_matrix, g["projection_matrix"])
