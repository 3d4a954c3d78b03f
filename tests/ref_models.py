"""The small inline models of the reference's own unit tests, re-typed as fixtures (the MJCF
text is input data of those tests: tests/test_configuration_limit.py:58-99,
tests/test_velocity_limit.py:67-132, tests/test_utils.py:62-137)."""

SUBSET_LIMITED = """
<mujoco>
  <compiler angle="radian"/>
  <worldbody>
    <body>
      <joint type="hinge" name="hinge_unlimited"/>
      <geom type="sphere" size=".1" mass=".1"/>
      <body>
        <joint type="hinge" name="hinge_limited" limited="true" range="0 1.57"/>
        <geom type="sphere" size=".1" mass=".1"/>
      </body>
    </body>
  </worldbody>
</mujoco>
"""

FREEJOINT_AND_HINGE = """
<mujoco>
  <compiler angle="radian"/>
  <worldbody>
    <body>
      <joint type="free" name="floating"/>
      <geom type="sphere" size=".1" mass=".1"/>
      <body>
        <joint type="hinge" name="hinge" range="0 1.57" limited="true"/>
        <geom type="sphere" size=".1" mass=".1"/>
      </body>
    </body>
  </worldbody>
</mujoco>
"""

BALL_AND_HINGE = """
<mujoco>
  <worldbody>
    <body>
      <joint type="ball" name="ball"/>
      <geom type="sphere" size=".1" mass=".1"/>
      <body>
        <joint type="hinge" name="hinge" range="0 1.57"/>
        <geom type="sphere" size=".1" mass=".1"/>
      </body>
    </body>
  </worldbody>
</mujoco>
"""

MOCAP = """
<mujoco>
  <worldbody>
    <body pos=".1 -.1 0">
      <joint type="free" name="floating"/>
      <geom type="sphere" size=".1" mass=".1"/>
      <body name="test">
        <joint type="hinge" name="hinge" range="0 1.57" limited="true"/>
        <geom type="sphere" size=".1" mass=".1"/>
      </body>
    </body>
    <body name="mocap" mocap="true" pos=".5 1 5" quat="1 1 0 0">
      <geom type="sphere" size=".1" mass=".1"/>
    </body>
  </worldbody>
</mujoco>
"""

SUBTREES = """
<mujoco>
  <worldbody>
    <body name="b1" pos=".1 -.1 0">
      <joint type="free"/>
      <geom name="b1/g1" type="sphere" size=".1" mass=".1"/>
      <geom name="b1/g2" type="sphere" size=".1" mass=".1" pos="0 0 .5"/>
      <body name="b2">
        <joint type="hinge" range="0 1.57" limited="true"/>
        <geom name="b2/g1" type="sphere" size=".1" mass=".1"/>
      </body>
    </body>
    <body name="b3" pos="1 1 1">
      <joint type="free"/>
      <geom name="b3/g1" type="sphere" size=".1" mass=".1"/>
      <body name="b4">
        <joint type="hinge" range="0 1.57" limited="true"/>
        <geom name="b4/g1" type="sphere" size=".1" mass=".1"/>
      </body>
    </body>
    <body name="geomless">
      <inertial pos="0 0 0" mass=".1" diaginertia="1 1 1"/>
    </body>
  </worldbody>
</mujoco>
"""

# Not from the reference's tests: a LIMITED ball joint (and a limited hinge behind it), the one
# joint kind for which ConfigurationLimit fills more than one qpos slot per joint
# (mink/limits/configuration_limit.py:46-53).
LIMITED_BALL_AND_HINGE = """
<mujoco>
  <compiler angle="radian"/>
  <worldbody>
    <body>
      <joint type="ball" name="ball" limited="true" range="0 0.9"/>
      <geom type="sphere" size=".1" mass=".1"/>
      <body pos="0 0 0.3">
        <joint type="hinge" name="hinge" axis="0 1 0" range="-0.5 1.57" limited="true"/>
        <geom type="sphere" size=".1" mass=".1"/>
        <site name="tip" pos="0 0 0.2"/>
      </body>
    </body>
  </worldbody>
</mujoco>
"""
