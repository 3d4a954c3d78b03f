"""Model helpers of the host API (reference: mink/utils.py:38-174)."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import constants as consts
from .exceptions import InvalidKeyframe, InvalidMocapBody
from .mjcf import JNT_FREE


def _mocap_id(model, mocap_name: str) -> int:
    bid = model.body_id(mocap_name)
    mocap_id = int(model.body_mocapid[bid]) if bid >= 0 else -1
    if mocap_id == -1:
        raise InvalidMocapBody(mocap_name, model)
    return mocap_id


def move_mocap_to_frame(model, data, mocap_name: str, frame_name: str, frame_type: str) -> None:
    """Put a mocap body at the pose of a frame (mink/utils.py:10-35).  ``data`` is the
    ``.data`` of a :class:`Configuration` (or the configuration itself), which is where the
    frame pose comes from."""
    mocap_id = _mocap_id(model, mocap_name)
    data = getattr(data, "data", data)
    T = data.configuration.get_transform_frame_to_world(frame_name, frame_type)
    if data.configuration.batched:
        data.mocap_pos[:, mocap_id] = T.translation()
        data.mocap_quat[:, mocap_id] = T.rotation().wxyz
    else:
        data.mocap_pos[mocap_id] = T.translation().cpu().numpy()
        data.mocap_quat[mocap_id] = T.rotation().wxyz.cpu().numpy()


def get_freejoint_dims(model):
    """(q_ids, v_ids) of all floating joints."""
    q_ids, v_ids = [], []
    for j in range(model.njnt):
        if model.jnt_type[j] == JNT_FREE:
            qadr, vadr = model.jnt_qposadr[j], model.jnt_dofadr[j]
            q_ids.extend(range(qadr, qadr + 7))
            v_ids.extend(range(vadr, vadr + 6))
    return q_ids, v_ids


def custom_configuration_vector(model, key_name: Optional[str] = None, **kwargs) -> np.ndarray:
    """Configuration where named joints take the given values, others their keyframe /
    neutral value."""
    if key_name is not None:
        key_id = model.key_id(key_name)
        if key_id == -1:
            raise InvalidKeyframe(key_name, model)
        q = model.key_qpos[key_id].copy()
    else:
        q = model.qpos0.copy()
    for name, value in kwargs.items():
        jid = model.joint_id(name)
        if jid < 0:
            raise KeyError(name)
        dim = consts.qpos_width(model.jnt_type[jid])
        qid = model.jnt_qposadr[jid]
        value = np.atleast_1d(value)
        if value.shape != (dim,):
            raise ValueError(f"Joint {name} should have a qpos value of {dim,} but got {value.shape}")
        q[qid:qid + dim] = value
    return q


def get_body_body_ids(model, body_id: int):
    return [i for i in range(model.nbody) if model.body_parentid[i] == body_id and body_id != i]


def get_subtree_body_ids(model, body_id: int):
    out, stack = [], [body_id]
    while stack:
        b = stack.pop()
        out.append(b)
        stack += get_body_body_ids(model, b)
    return out


def get_body_geom_ids(model, body_id: int):
    start = model.body_geomadr[body_id]
    return list(range(start, start + model.body_geomnum[body_id])) if start >= 0 else []


def get_subtree_geom_ids(model, body_id: int):
    out, stack = [], [body_id]
    while stack:
        b = stack.pop()
        out.extend(get_body_geom_ids(model, b))
        stack += get_body_body_ids(model, b)
    return out
