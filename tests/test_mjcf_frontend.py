"""The MJCF front end (mink_amd/mjcf.py: what mujoco.MjModel.from_xml_path gives the reference,
mink/configuration.py:37-52) on the orientation / geom attributes of humanoid-style models, with
expectations worked out by hand, and a pin of the compiled TidyBot table (the kinematic tree the
kernels run on) against literals read off the reference's MJCF
(interactive_scripts/stanford_tidybot/tidybot_cube.xml:36-47,101-265,305-310 and cube.xml:21-32),
independent of the compiler, the oracle and the golden generator."""
import os

import numpy as np
import pytest

from mink_amd.mjcf import (GEOM_BOX, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_ELLIPSOID, JNT_FREE, JNT_HINGE, JNT_SLIDE,
                           load_builtin, load_mjcf, load_mjcf_string)

HERE = os.path.dirname(os.path.abspath(__file__))
S = np.sqrt(0.5)


def quat_close(a, b, atol=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    return np.allclose(a, b, rtol=0, atol=atol) or np.allclose(a, -b, rtol=0, atol=atol)


def rotate(q, v):
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return R @ np.asarray(v, dtype=float)


ORIENT = """
<mujoco>
  <compiler angle="radian"/>
  <worldbody>
    <body name="a" xyaxes="0 1 0 -1 0 0"><geom type="sphere" size=".1"/></body>
    <body name="b" xyaxes="1 0 0 1 1 0"><geom type="sphere" size=".1"/></body>
    <body name="c" xyaxes="0 0 2 0 3 0"><geom type="sphere" size=".1"/></body>
    <body name="d" zaxis="0 1 0"><geom type="sphere" size=".1"/></body>
    <body name="e" zaxis="0 0 -2"><geom type="sphere" size=".1"/></body>
    <body name="f" zaxis="0 0 5"><geom type="sphere" size=".1"/>
      <site name="s" zaxis="1 0 0"/>
    </body>
  </worldbody>
</mujoco>
"""


def test_xyaxes_and_zaxis_hand_computed():
    m = load_mjcf_string(ORIENT)
    q = {n: m.body_quat[m.body_id(n)] for n in "abcdef"}
    # x -> +y, y -> -x: a quarter turn about z
    assert quat_close(q["a"], [S, 0, 0, S])
    # y is made orthogonal to x: (1, 1, 0) - (1, 0, 0) -> +y: the identity
    assert quat_close(q["b"], [1, 0, 0, 0])
    # x -> +z, y -> +y, so z -> x cross y = z cross y = -x: a quarter turn about y taking x to z, i.e. by -90 deg
    assert quat_close(q["c"], [S, 0, -S, 0])
    np.testing.assert_allclose(rotate(q["c"], [1, 0, 0]), [0, 0, 1], atol=1e-15)
    np.testing.assert_allclose(rotate(q["c"], [0, 0, 1]), [-1, 0, 0], atol=1e-15)
    # z -> +y by the smallest rotation: -90 deg about x
    assert quat_close(q["d"], [S, -S, 0, 0])
    np.testing.assert_allclose(rotate(q["d"], [0, 0, 1]), [0, 1, 0], atol=1e-15)
    # z -> -z: half a turn about x (the compiler's choice of axis for the antiparallel case)
    assert quat_close(q["e"], [0, 1, 0, 0])
    assert quat_close(q["f"], [1, 0, 0, 0])
    # z -> +x: +90 deg about y
    assert quat_close(m.site_quat[m.site_id("s")], [S, 0, S, 0])


FROMTO = """
<mujoco>
  <worldbody>
    <body name="b">
      <geom name="up" type="capsule" fromto="0 0 0 0 0 1" size="0.05"/>
      <geom name="along_x" type="cylinder" fromto="0 0 0 1 0 0" size="0.02"/>
      <geom name="diag" type="capsule" fromto="1 2 3 1 5 7" size="0.1"/>
      <geom name="slab" type="box" fromto="0 0 -0.2 0 0 0.2" size="0.3 0.1 0"/>
      <geom name="egg" type="ellipsoid" fromto="0 -0.5 0 0 0.5 0" size="0.1 0.2 0"/>
    </body>
  </worldbody>
</mujoco>
"""


def test_fromto_hand_computed():
    m = load_mjcf_string(FROMTO)
    g = {n: m.geom_id(n) for n in ("up", "along_x", "diag", "slab", "egg")}
    # midpoint, half length in the long-axis slot, z axis along from - to
    np.testing.assert_allclose(m.geom_pos[g["up"]], [0, 0, 0.5])
    np.testing.assert_allclose(m.geom_size[g["up"]], [0.05, 0.5, 0])
    assert quat_close(m.geom_quat[g["up"]], [0, 1, 0, 0])                 # from - to = -z: half a turn about x
    assert int(m.geom_type[g["up"]]) == GEOM_CAPSULE
    np.testing.assert_allclose(m.geom_pos[g["along_x"]], [0.5, 0, 0])
    np.testing.assert_allclose(m.geom_size[g["along_x"]], [0.02, 0.5, 0])
    assert quat_close(m.geom_quat[g["along_x"]], [S, 0, -S, 0])           # z -> -x: -90 deg about y
    assert int(m.geom_type[g["along_x"]]) == GEOM_CYLINDER
    # a 3-4-5 segment: length 5, direction from - to = (0, -0.6, -0.8)
    np.testing.assert_allclose(m.geom_pos[g["diag"]], [1, 3.5, 5])
    np.testing.assert_allclose(m.geom_size[g["diag"]], [0.1, 2.5, 0])
    np.testing.assert_allclose(rotate(m.geom_quat[g["diag"]], [0, 0, 1]), [0, -0.6, -0.8], atol=1e-15)
    # box / ellipsoid: the half length goes to the third size
    np.testing.assert_allclose(m.geom_size[g["slab"]], [0.3, 0.1, 0.2])
    np.testing.assert_allclose(m.geom_pos[g["slab"]], [0, 0, 0])
    assert int(m.geom_type[g["slab"]]) == GEOM_BOX
    np.testing.assert_allclose(m.geom_size[g["egg"]], [0.1, 0.2, 0.5])
    np.testing.assert_allclose(rotate(m.geom_quat[g["egg"]], [0, 0, 1]), [0, -1, 0], atol=1e-15)
    assert int(m.geom_type[g["egg"]]) == GEOM_ELLIPSOID
    with pytest.raises(ValueError):
        load_mjcf_string(FROMTO.replace('type="capsule" fromto="0 0 0 0 0 1"', 'type="sphere" fromto="0 0 0 0 0 1"'))


def test_biped_model_tables():
    """The humanoid-shaped test model: counts, the frames given by xyaxes / zaxis, degrees,
    default classes, and capsule masses from the fromto lengths."""
    m = load_mjcf(os.path.join(HERE, "models", "biped.xml"))
    assert (m.nq, m.nv, m.nbody, m.njnt, m.nsite) == (27, 26, 12, 21, 5)
    assert int(m.jnt_type[0]) == JNT_FREE and all(int(t) == JNT_HINGE for t in m.jnt_type[1:])
    # dof chains: each leg and arm hangs off the free joint's last dof
    assert list(m.dof_parentid[:6]) == [-1, 0, 1, 2, 3, 4]
    l_hip_yaw = int(m.jnt_dofadr[m.joint_id("l_hip_yaw")])
    r_hip_yaw = int(m.jnt_dofadr[m.joint_id("r_hip_yaw")])
    l_sh = int(m.jnt_dofadr[m.joint_id("l_shoulder_pitch")])
    assert m.dof_parentid[l_hip_yaw] == 5 and m.dof_parentid[r_hip_yaw] == 5 and m.dof_parentid[l_sh] == 5
    assert m.dof_parentid[int(m.jnt_dofadr[m.joint_id("l_ankle_roll")])] == int(m.jnt_dofadr[m.joint_id("l_ankle_pitch")])
    np.testing.assert_allclose(m.jnt_range[m.joint_id("l_knee")], np.deg2rad([0, 140]))
    np.testing.assert_allclose(m.jnt_range[m.joint_id("r_shoulder_roll")], np.deg2rad([-160, 10]))
    # foot frame: x tilted down by atan(0.1) about y
    foot = m.body_quat[m.body_id("l_foot")]
    np.testing.assert_allclose(rotate(foot, [1, 0, 0]), np.array([1, 0, -0.1]) / np.hypot(1, 0.1), atol=1e-15)
    np.testing.assert_allclose(rotate(foot, [0, 1, 0]), [0, 1, 0], atol=1e-15)
    np.testing.assert_allclose(rotate(m.site_quat[m.site_id("head")], [0, 0, 1]), np.array([0.2, 0, 1]) / np.hypot(0.2, 1), atol=1e-15)
    assert quat_close(m.site_quat[m.site_id("r_hand")], [S, 0, 0, S])
    # thigh capsule: radius 0.04 (default class), half length from the segment, density 900
    g = m.geom_id("l_thigh")
    r, hl = 0.04, 0.5 * np.sqrt(0.01 ** 2 + 0.38 ** 2)
    np.testing.assert_allclose(m.geom_size[g], [r, hl, 0])
    shin_r, shin_hl = 0.035, 0.18
    vol = lambda rr, hh: np.pi * rr * rr * 2 * hh + 4.0 / 3.0 * np.pi * rr ** 3
    np.testing.assert_allclose(m.body_mass[m.body_id("l_hip")], 900 * vol(r, hl), rtol=1e-12)
    np.testing.assert_allclose(m.body_mass[m.body_id("l_shin")], 900 * vol(shin_r, shin_hl), rtol=1e-12)
    # arm class: radius 0.03
    assert m.geom_size[m.geom_id("r_upper_arm_geom")][0] == 0.03
    assert m.key_qpos.shape == (1, 27)


# ---- the compiled TidyBot table, against literals read off the MJCF ---------------------
TIDYBOT_BODIES = [   # name, parent, pos, quat as written in the file (the compiler normalises it)
    ("world", 0, (0, 0, 0), (1, 0, 0, 0)),
    ("base_link", 0, (0, 0, 0), (1, 0, 0, 0)),
    ("gen3/base_link", 1, (0, 0, 0.335), (1, 0, 0, 0)),
    ("gen3/shoulder_link", 2, (0, 0, 0.15643), (0, 1, 0, 0)),
    ("gen3/half_arm_1_link", 3, (0, 0.005375, -0.12838), (1, 1, 0, 0)),
    ("gen3/half_arm_2_link", 4, (0, -0.21038, -0.006375), (1, -1, 0, 0)),
    ("gen3/forearm_link", 5, (0, 0.006375, -0.21038), (1, 1, 0, 0)),
    ("gen3/spherical_wrist_1_link", 6, (0, -0.20843, -0.006375), (1, -1, 0, 0)),
    ("gen3/spherical_wrist_2_link", 7, (0, 0.00017505, -0.10593), (1, 1, 0, 0)),
    ("gen3/bracelet_link", 8, (0, -0.10593, -0.00017505), (1, -1, 0, 0)),
    ("base", 9, (0, 0, -0.06149039), (0, -1, 1, 0)),
    ("right_driver", 10, (0, 0.0306011, 0.054904), (1, 0, 0, 0)),
    ("right_coupler", 11, (0, 0.0315, -0.0041), (1, 0, 0, 0)),
    ("right_spring_link", 10, (0, 0.0132, 0.0609), (1, 0, 0, 0)),
    ("right_follower", 13, (0, 0.055, 0.0375), (1, 0, 0, 0)),
    ("right_pad", 14, (0, -0.0189, 0.01352), (1, 0, 0, 0)),
    ("right_silicone_pad", 15, (0, 0, 0), (1, 0, 0, 0)),
    ("left_driver", 10, (0, -0.0306011, 0.054904), (0, 0, 0, 1)),
    ("left_coupler", 17, (0, 0.0315, -0.0041), (1, 0, 0, 0)),
    ("left_spring_link", 10, (0, -0.0132, 0.0609), (0, 0, 0, 1)),
    ("left_follower", 19, (0, 0.055, 0.0375), (1, 0, 0, 0)),
    ("left_pad", 20, (0, -0.0189, 0.01352), (1, 0, 0, 0)),
    ("left_silicone_pad", 21, (0, 0, 0), (1, 0, 0, 0)),
    ("pinch_site_target", 0, (0.5, 0, 0.5), (0, 1, 0, 0)),
    ("interactive_obj", 0, (0, 0, 0.1), (1, 0, 0, 0)),
]
X, Y, Z = (1, 0, 0), (0, 1, 0), (0, 0, 1)
TIDYBOT_JOINTS = [   # name, body, type, axis, range or None (range => limited, autolimits)
    ("joint_x", 1, JNT_SLIDE, X, None), ("joint_y", 1, JNT_SLIDE, Y, None), ("joint_th", 1, JNT_HINGE, Z, None),
    ("joint_1", 3, JNT_HINGE, Z, None), ("joint_2", 4, JNT_HINGE, Z, (-2.24, 2.24)),
    ("joint_3", 5, JNT_HINGE, Z, None), ("joint_4", 6, JNT_HINGE, Z, (-2.57, 2.57)),
    ("joint_5", 7, JNT_HINGE, Z, None), ("joint_6", 8, JNT_HINGE, Z, (-2.09, 2.09)),
    ("joint_7", 9, JNT_HINGE, Z, None),
    # the 2F-85 fingers: childclass "2f85" makes the default axis (1, 0, 0)
    ("right_driver_joint", 11, JNT_HINGE, X, (0, 0.8)), ("right_coupler_joint", 12, JNT_HINGE, X, (-1.57, 0)),
    ("right_spring_link_joint", 13, JNT_HINGE, X, (-0.29670597283, 0.8)),
    ("right_follower_joint", 14, JNT_HINGE, X, (-0.872664, 0.872664)),
    ("left_driver_joint", 17, JNT_HINGE, X, (0, 0.8)), ("left_coupler_joint", 18, JNT_HINGE, X, (-1.57, 0)),
    ("left_spring_link_joint", 19, JNT_HINGE, X, (-0.29670597283, 0.8)),
    ("left_follower_joint", 20, JNT_HINGE, X, (-0.872664, 0.872664)),
    ("interactive_obj_freejoint", 24, JNT_FREE, Z, None),
]
TIDYBOT_DOF_PARENT = [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 9, 12, 9, 14, 9, 16, -1, 18, 19, 20, 21, 22]
TIDYBOT_HOME = [0, 0, 0, 0, 0.26179939, 3.14159265, -2.26892803, 0, 0.95993109, 1.57079633] + [0] * 8 + \
    [0.6, 0, 0] + [1, 0, 0, 0]      # the key's all-zero cube quaternion compiles to the identity


@pytest.mark.parametrize("source", ["builtin", "mjcf"])
def test_tidybot_table_matches_the_mjcf_literals(source):
    if source == "builtin":
        m = load_builtin("tidybot_cube")          # the committed table mink_amd/data/tidybot_cube.json
    else:
        path = "/root/reference/interactive_scripts/stanford_tidybot/cube.xml"
        if not os.path.exists(path):
            pytest.skip("the reference's MJCF is not on this machine")
        m = load_mjcf(path)
    assert (m.nbody, m.njnt, m.nq, m.nv) == (25, 19, 25, 24)
    assert [n.split("/")[-1] for n in m.body_names] == [b[0].split("/")[-1] for b in TIDYBOT_BODIES]
    for i, (name, parent, pos, quat) in enumerate(TIDYBOT_BODIES):
        assert int(m.body_parentid[i]) == parent, name
        np.testing.assert_allclose(m.body_pos[i], pos, rtol=0, atol=1e-15, err_msg=name)
        q = np.array(quat, dtype=float)
        np.testing.assert_allclose(m.body_quat[i], q / np.linalg.norm(q), rtol=0, atol=1e-15, err_msg=name)
    assert int(m.body_mocapid[23]) == 0 and all(int(x) == -1 for i, x in enumerate(m.body_mocapid) if i != 23)
    assert [n.split("/")[-1] for n in m.jnt_names] == [j[0] for j in TIDYBOT_JOINTS]
    qadr = dadr = 0
    for j, (name, body, jtype, axis, rng) in enumerate(TIDYBOT_JOINTS):
        assert int(m.jnt_bodyid[j]) == body and int(m.jnt_type[j]) == jtype, name
        assert int(m.jnt_qposadr[j]) == qadr and int(m.jnt_dofadr[j]) == dadr, name
        if jtype != JNT_FREE:
            np.testing.assert_allclose(m.jnt_axis[j], axis, rtol=0, atol=0, err_msg=name)
        assert bool(m.jnt_limited[j]) == (rng is not None), name
        if rng is not None:
            np.testing.assert_allclose(m.jnt_range[j], rng, rtol=0, atol=0, err_msg=name)
        qadr += 7 if jtype == JNT_FREE else 1
        dadr += 6 if jtype == JNT_FREE else 1
    # the one off-centre joint anchor (tidybot_cube.xml:41), everything else at the body origin
    fol = [m.joint_id("right_follower_joint"), m.joint_id("left_follower_joint")]
    for j in range(m.njnt):
        np.testing.assert_allclose(m.jnt_pos[j], (0, -0.018, 0.0065) if j in fol else (0, 0, 0), rtol=0, atol=0)
    np.testing.assert_array_equal(m.dof_parentid, TIDYBOT_DOF_PARENT)
    np.testing.assert_array_equal(m.qpos0[:18], np.zeros(18))
    np.testing.assert_allclose(m.qpos0[18:], [0, 0, 0.1, 1, 0, 0, 0], rtol=0, atol=0)
    home = m.key_qpos[m.key_names.index("home")]
    np.testing.assert_allclose(home, TIDYBOT_HOME, rtol=0, atol=0)
    # the task frame: site pinch_site on the bracelet link
    s = m.site_id("pinch_site")
    assert int(m.site_bodyid[s]) == 9
    np.testing.assert_allclose(m.site_pos[s], (0, 0, -0.181525), rtol=0, atol=0)
    np.testing.assert_allclose(m.site_quat[s], (0, 1, 0, 0), rtol=0, atol=0)
    # the collision primitives of benchmark config 4
    for name, size in (("right_pad1", None), ("left_pad2", None), ("cube_geom", (0.02, 0.02, 0.02))):
        g = m.geom_id(name)
        assert g >= 0 and int(m.geom_type[g]) == GEOM_BOX
        if size is not None:
            np.testing.assert_allclose(m.geom_size[g], size, rtol=0, atol=0)


def test_tidybot_home_pose_by_hand():
    """pinch_site at the home key, composed here link by link from the literals above (plain
    rotation-matrix products, no library code): the known answer (0.57667, 0.00135, 0.76872),
    quaternion (1/2, 1/2, 1/2, 1/2)."""
    def rot(q):
        q = np.array(q, dtype=float)
        w, x, y, z = q / np.linalg.norm(q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    def rotz(a):
        return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])

    R, p = np.eye(3), np.zeros(3)
    angles = {3: TIDYBOT_HOME[3], 4: TIDYBOT_HOME[4], 5: TIDYBOT_HOME[5], 6: TIDYBOT_HOME[6], 7: TIDYBOT_HOME[7],
              8: TIDYBOT_HOME[8], 9: TIDYBOT_HOME[9]}
    for i in range(2, 10):          # gen3/base_link ... bracelet_link (the base joints are at zero)
        _, _, pos, quat = TIDYBOT_BODIES[i]
        p = p + R @ np.array(pos)
        R = R @ rot(quat)
        if i in angles:
            R = R @ rotz(angles[i])
    p = p + R @ np.array([0, 0, -0.181525])
    R = R @ rot((0, 1, 0, 0))
    np.testing.assert_allclose(p, [0.57667, 0.00135, 0.76872], rtol=0, atol=1e-5)
    np.testing.assert_allclose(R, rot((0.5, 0.5, 0.5, 0.5)), rtol=0, atol=1e-8)
    # and the same from the package's kinematics
    from oracle import ik_oracle as o
    m = load_builtin("tidybot_cube")
    T = o.Configuration(m, m.key_qpos[m.key_names.index("home")]).get_transform_frame_to_world("pinch_site", "site")
    np.testing.assert_allclose(T[4:], p, rtol=0, atol=1e-12)
    assert quat_close(T[:4], [0.5, 0.5, 0.5, 0.5], atol=1e-8)
