"""Frame-type tables and joint widths (reference: mink/constants.py:3-34)."""
from ._lib import FRAME_TYPES

SUPPORTED_FRAMES = ("body", "geom", "site")
FRAME_TO_ENUM = dict(FRAME_TYPES)


def dof_width(joint_type: int) -> int:
    """Dimension of the joint in the tangent space."""
    return {0: 6, 1: 3, 2: 1, 3: 1}[int(joint_type)]


def qpos_width(joint_type: int) -> int:
    """Dimension of the joint in the configuration vector."""
    return {0: 7, 1: 4, 2: 1, 3: 1}[int(joint_type)]
