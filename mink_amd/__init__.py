"""mink_amd: MI355X-native batched differential inverse kinematics behind the ``mink``
API (solve_ik / build_ik, Configuration, tasks, limits, SO3 / SE3).

The kinematics, task and limit assembly and the QP solve run in hand-written HIP
kernels (mink_amd/csrc) reached through the C ABI of include/mink_amd.h; importing the
GPU-facing names requires the built extension and PyTorch-ROCm.
"""
from .constants import (FRAME_TO_ENUM, FRAME_TO_JAC_FUNC, FRAME_TO_POS_ATTR, FRAME_TO_XMAT_ATTR,
                        SUPPORTED_FRAMES)
from .exceptions import (InvalidDamping, InvalidFrame, InvalidGain, InvalidKeyframe, InvalidMocapBody, InvalidTarget,
                         LimitDefinitionError, MinkError, NotWithinConfigurationLimits, TargetNotSet,
                         TaskDefinitionError, UnsupportedFrame)
from .mjcf import Model, load_builtin, load_mjcf, load_mjcf_string

_GPU_NAMES = {
    "Configuration": "configuration", "SE3": "lie", "SO3": "lie", "MatrixLieGroup": "lie",
    "ComTask": "tasks", "DampingTask": "tasks", "FrameTask": "tasks", "Objective": "tasks", "PostureTask": "tasks",
    "RelativeFrameTask": "tasks", "Task": "tasks",
    "CollisionAvoidanceLimit": "limits", "ConfigurationLimit": "limits", "Constraint": "limits", "Limit": "limits",
    "VelocityLimit": "limits",
    "build_ik": "solve_ik", "solve_ik": "solve_ik", "solve_ik_iterate": "solve_ik", "Problem": "solve_ik",
    "custom_configuration_vector": "utils", "get_freejoint_dims": "utils", "get_body_geom_ids": "utils",
    "get_subtree_body_ids": "utils", "get_subtree_geom_ids": "utils", "get_body_body_ids": "utils",
    "move_mocap_to_frame": "utils",
    "move_to": "rollout", "LinearWaypointReach": "rollout", "LinearWaypointReachConfig": "rollout",
}


def __getattr__(name):
    # torch-dependent modules are imported on first use so that the model compiler and
    # the ctypes layer stay importable in minimal environments
    mod = _GPU_NAMES.get(name)
    if mod is None:
        raise AttributeError(f"module 'mink_amd' has no attribute '{name}'")
    import importlib
    module = importlib.import_module(f".{mod}", __name__)
    # bind every public name of that module on the package: the import above has just set
    # the attribute `solve_ik` to the submodule, which must not shadow the function
    for public, owner in _GPU_NAMES.items():
        if owner == mod:
            globals()[public] = getattr(module, public)
    return globals()[name]


__all__ = tuple(_GPU_NAMES) + (
    "MinkError", "UnsupportedFrame", "InvalidFrame", "InvalidKeyframe", "NotWithinConfigurationLimits",
    "TargetNotSet", "InvalidMocapBody", "InvalidTarget", "InvalidGain", "InvalidDamping", "TaskDefinitionError",
    "LimitDefinitionError", "SUPPORTED_FRAMES", "FRAME_TO_ENUM", "FRAME_TO_JAC_FUNC", "FRAME_TO_POS_ATTR",
    "FRAME_TO_XMAT_ATTR", "Model", "load_mjcf", "load_mjcf_string", "load_builtin")
