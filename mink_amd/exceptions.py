"""Exception hierarchy of the host API; classes and messages follow the reference
(mink/exceptions.py:8-98, mink/tasks/exceptions.py, mink/limits/exceptions.py), whose
tests assert several of the strings verbatim."""
from typing import Sequence


class MinkError(Exception):
    """Base class of the package's exceptions."""


class UnsupportedFrame(MinkError):
    def __init__(self, frame_type: str, supported_types: Sequence[str]):
        super().__init__(f"{frame_type} is not supported.Supported frame types are: {supported_types}")


class InvalidFrame(MinkError):
    def __init__(self, frame_name: str, frame_type: str, model):
        names = {"body": model.body_names, "site": model.site_names, "geom": model.geom_names}[frame_type]
        super().__init__(f"{frame_type} '{frame_name}' does not exist in the model. "
                         f"Available {frame_type} names: {list(names)}")


class InvalidKeyframe(MinkError):
    def __init__(self, keyframe_name: str, model):
        super().__init__(f"Keyframe {keyframe_name} does not exist in the model. "
                         f"Available keyframe names: {list(model.key_names)}")


class InvalidMocapBody(MinkError):
    def __init__(self, mocap_name: str, model):
        names = [n for n, k in zip(model.body_names, model.body_mocapid) if k != -1]
        super().__init__(f"Body '{mocap_name}' is not a mocap body. Available mocap bodies: {names}")


class NotWithinConfigurationLimits(MinkError):
    def __init__(self, joint_id: int, value: float, lower: float, upper: float, model):
        super().__init__(f"Joint {joint_id} ({model.jnt_names[joint_id]}) violates configuration limits "
                         f"{lower} <= {value} <= {upper}")


class TaskDefinitionError(MinkError):
    """A task definition is ill-formed."""


class TargetNotSet(MinkError):
    def __init__(self, cls_name: str):
        super().__init__(f"No target set for {cls_name}")


class InvalidTarget(MinkError):
    """The target is invalid."""


class InvalidGain(MinkError):
    """The gain is outside the valid range."""


class InvalidDamping(MinkError):
    """The damping is outside the valid range."""


class LimitDefinitionError(MinkError):
    """A limit definition is ill-formed."""
