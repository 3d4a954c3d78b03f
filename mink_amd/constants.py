"""Frame-type tables and joint widths (reference: mink/constants.py:3-34)."""
from ._lib import FRAME_TYPES

SUPPORTED_FRAMES = ("body", "geom", "site")
FRAME_TO_ENUM = dict(FRAME_TYPES)
# Names of the MuJoCo data fields the reference reads frame poses from; kept for callers
# that index by them (here poses come from Configuration.get_transform_frame_to_world).
FRAME_TO_POS_ATTR = {"body": "xpos", "geom": "geom_xpos", "site": "site_xpos"}
FRAME_TO_XMAT_ATTR = {"body": "xmat", "geom": "geom_xmat", "site": "site_xmat"}


def _world_jacobian(frame_type: str):
    def jac(configuration, frame_id: int):
        """World-aligned (jacp, jacr), each [.., 3, nv], of the frame: what mj_jacBody /
        mj_jacGeom / mj_jacSite fill in (mink/constants.py:10-14), from the kernels'
        body-frame Jacobian rotated back by the frame's orientation."""
        from .lie import SO3
        dm, q = configuration.dmodel, configuration.q_tensor
        pose = dm.frame_pose(q, FRAME_TO_ENUM[frame_type], int(frame_id))
        R = SO3(pose[:, :4]).as_matrix()
        Jb = dm.frame_jacobian(q, FRAME_TO_ENUM[frame_type], int(frame_id))
        return configuration._out(R @ Jb[:, :3]), configuration._out(R @ Jb[:, 3:])
    jac.__name__ = {"body": "mj_jacBody", "geom": "mj_jacGeom", "site": "mj_jacSite"}[frame_type]
    return jac


FRAME_TO_JAC_FUNC = {k: _world_jacobian(k) for k in SUPPORTED_FRAMES}


def dof_width(joint_type: int) -> int:
    """Dimension of the joint in the tangent space."""
    return {0: 6, 1: 3, 2: 1, 3: 1}[int(joint_type)]


def qpos_width(joint_type: int) -> int:
    """Dimension of the joint in the configuration vector."""
    return {0: 7, 1: 4, 2: 1, 3: 1}[int(joint_type)]
