"""ctypes binding of the C ABI in include/mink_amd.h.

The structures below mirror ``mk_model_desc`` / ``mk_task_desc`` / ``mk_limit_desc``
field by field.  The shared library is built in-tree by ``__graft_entry__.build()``;
importing it fails loudly when it is missing -- there is no CPU fallback.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MINK_AMD_LIB lets kernel-tuning scripts load an experimental build of the same library.
LIB_PATH = os.environ.get("MINK_AMD_LIB") or os.path.join(_HERE, "libmink_amd.so")

MK_OK, MK_ERR_INVALID, MK_ERR_UNSUPPORTED, MK_ERR_HIP = 0, -1, -2, -3
FRAME_TYPES = {"body": 0, "geom": 1, "site": 2}
TASK_FRAME, TASK_RELATIVE_FRAME, TASK_POSTURE, TASK_COM = 0, 1, 2, 3
LIMIT_CONFIGURATION, LIMIT_VELOCITY, LIMIT_COLLISION = 0, 1, 2

_pi32 = C.POINTER(C.c_int32)
_pf64 = C.POINTER(C.c_double)


class ModelDesc(C.Structure):
    _fields_ = [
        ("nq", C.c_int32), ("nv", C.c_int32), ("nbody", C.c_int32), ("njnt", C.c_int32),
        ("ngeom", C.c_int32), ("nsite", C.c_int32),
        ("body_parentid", _pi32), ("body_jntadr", _pi32), ("body_jntnum", _pi32), ("body_mocapid", _pi32),
        ("body_pos", _pf64), ("body_quat", _pf64), ("body_mass", _pf64), ("body_ipos", _pf64),
        ("jnt_type", _pi32), ("jnt_qposadr", _pi32), ("jnt_dofadr", _pi32), ("jnt_bodyid", _pi32),
        ("jnt_limited", _pi32),
        ("jnt_pos", _pf64), ("jnt_axis", _pf64), ("jnt_range", _pf64), ("qpos0", _pf64),
        ("dof_parentid", _pi32), ("dof_jntid", _pi32),
        ("site_bodyid", _pi32), ("site_pos", _pf64), ("site_quat", _pf64),
        ("geom_bodyid", _pi32), ("geom_type", _pi32), ("geom_size", _pf64), ("geom_pos", _pf64),
        ("geom_quat", _pf64),
        ("nmesh", C.c_int32), ("nmeshvert", C.c_int32),
        ("geom_dataid", _pi32), ("mesh_vertadr", _pi32), ("mesh_vertnum", _pi32),
        ("mesh_vert", _pf64), ("mesh_center", _pf64), ("mesh_radius", _pf64),
    ]


class TaskDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("frame_type", C.c_int32), ("frame_id", C.c_int32),
        ("root_type", C.c_int32), ("root_id", C.c_int32), ("ncost", C.c_int32),
        ("gain", C.c_double), ("lm_damping", C.c_double),
        ("cost", _pf64), ("target_q", _pf64),
    ]


class LimitDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("n", C.c_int32),
        ("gain", C.c_double), ("min_distance_from_limits", C.c_double),
        ("index", _pi32), ("vmax", _pf64),
        ("minimum_distance_from_collisions", C.c_double),
        ("collision_detection_distance", C.c_double),
        ("bound_relaxation", C.c_double),
    ]


class ProblemInfo(C.Structure):
    _fields_ = [("nq", C.c_int32), ("nv", C.c_int32), ("target_stride", C.c_int32),
                ("n_rows", C.c_int32), ("fused", C.c_int32), ("n_chain", C.c_int32), ("staged", C.c_int32),
                ("staged_rows", C.c_int32)]


class WaypointConfig(C.Structure):
    _fields_ = [("pos_threshold", C.c_double), ("pos_step_size", C.c_double), ("rot_threshold", C.c_double),
                ("rot_step_size", C.c_double)]


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.float64)


def pack_model(model) -> Tuple[ModelDesc, list]:
    """mk_model_desc for a :class:`mink_amd.mjcf.Model`; the second value keeps the
    numpy buffers alive for the duration of the call."""
    keep: list = []
    d = ModelDesc()
    for name in ("nq", "nv", "nbody", "njnt", "ngeom", "nsite"):
        setattr(d, name, int(getattr(model, name)))
    d.nmesh = int(getattr(model, "nmesh", 0) or 0)
    mv = getattr(model, "mesh_vert", None)
    d.nmeshvert = 0 if mv is None else int(np.asarray(mv).reshape(-1, 3).shape[0])
    for name, ctype in ModelDesc._fields_[6:]:
        if ctype not in (_pi32, _pf64):
            continue
        arr = getattr(model, name, None)
        if arr is None:
            arr = np.full(model.ngeom, -1, dtype=np.int32) if name == "geom_dataid" else np.zeros(0)
        if name == "jnt_limited":
            arr = np.asarray(arr).astype(np.int32)
        arr = _i32(arr) if ctype is _pi32 else _f64(arr)
        keep.append(arr)
        setattr(d, name, arr.ctypes.data_as(ctype))
    return d, keep


def pack_tasks(specs: Sequence[dict]) -> Tuple[C.Array, list]:
    """specs: dicts with keys type, frame_type, frame_id, root_type, root_id, gain,
    lm_damping, cost (array), target_q (array or None)."""
    keep: list = []
    arr = (TaskDesc * max(len(specs), 1))()
    for i, s in enumerate(specs):
        t = arr[i]
        t.type = s["type"]
        t.frame_type = s.get("frame_type", 0)
        t.frame_id = s.get("frame_id", 0)
        t.root_type = s.get("root_type", 0)
        t.root_id = s.get("root_id", 0)
        cost = _f64(s["cost"]).reshape(-1)
        keep.append(cost)
        t.ncost = cost.shape[0]
        t.cost = cost.ctypes.data_as(_pf64)
        t.gain = float(s.get("gain", 1.0))
        t.lm_damping = float(s.get("lm_damping", 0.0))
        if s.get("target_q") is not None:
            tq = _f64(s["target_q"]).reshape(-1)
            keep.append(tq)
            t.target_q = tq.ctypes.data_as(_pf64)
    return arr, keep


def pack_limits(specs: Sequence[dict]) -> Tuple[C.Array, list]:
    keep: list = []
    arr = (LimitDesc * max(len(specs), 1))()
    for i, s in enumerate(specs):
        l = arr[i]
        l.type = s["type"]
        l.gain = float(s.get("gain", 0.0))
        l.min_distance_from_limits = float(s.get("min_distance_from_limits", 0.0))
        l.minimum_distance_from_collisions = float(s.get("minimum_distance_from_collisions", 0.0))
        l.collision_detection_distance = float(s.get("collision_detection_distance", 0.0))
        l.bound_relaxation = float(s.get("bound_relaxation", 0.0))
        idx = _i32(s.get("index", [])).reshape(-1)
        keep.append(idx)
        l.index = idx.ctypes.data_as(_pi32)
        if s["type"] == LIMIT_VELOCITY:
            vm = _f64(s["vmax"]).reshape(-1)
            keep.append(vm)
            l.vmax = vm.ctypes.data_as(_pf64)
            l.n = idx.shape[0]
        elif s["type"] == LIMIT_COLLISION:
            l.n = idx.shape[0] // 2
    return arr, keep


_SIGNATURES = {
    "mk_last_error": (C.c_char_p, []),
    "mk_model_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "mk_model_destroy": (None, [C.c_void_p]),
    "mk_problem_create": (C.c_int, [C.c_void_p, C.POINTER(TaskDesc), C.c_int32, C.POINTER(LimitDesc), C.c_int32,
                                    C.POINTER(C.c_void_p)]),
    "mk_problem_destroy": (None, [C.c_void_p]),
    "mk_problem_get_info": (C.c_int, [C.c_void_p, C.POINTER(ProblemInfo)]),
    "mk_problem_set_posture_target": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mk_solve_ik_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "mk_solve_ik_iterate_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32,
                                            C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "mk_move_to_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "mk_move_to_passes_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                          C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "mk_integrate_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p]),
    "mk_fk_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mk_frame_jacobian_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_void_p]),
    "mk_check_limits_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]),
    "mk_build_qp_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mk_task_eval_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "mk_qp_solve_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "mk_lie_batch": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mk_lie_op_shape": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mk_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "mk_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

# operations of mk_lie_batch (include/mink_amd.h)
LIE_OPS = {
    "so3_exp": 0, "so3_log": 1, "so3_multiply": 2, "so3_inverse": 3, "so3_normalize": 4, "so3_apply": 5,
    "so3_as_matrix": 6, "so3_from_matrix": 7, "so3_ljac": 8, "so3_ljacinv": 9, "so3_rjac": 10, "so3_rjacinv": 11,
    "so3_jlog": 12, "so3_rplus": 13, "so3_lplus": 14, "so3_rminus": 15, "so3_lminus": 16, "so3_as_rpy": 17,
    "so3_from_rpy": 18,
    "se3_exp": 32, "se3_log": 33, "se3_multiply": 34, "se3_inverse": 35, "se3_normalize": 36, "se3_apply": 37,
    "se3_as_matrix": 38, "se3_from_matrix": 39, "se3_adjoint": 40, "se3_ljac": 41, "se3_ljacinv": 42, "se3_rjac": 43,
    "se3_rjacinv": 44, "se3_jlog": 45, "se3_rplus": 46, "se3_lplus": 47, "se3_rminus": 48, "se3_lminus": 49,
}

_lib = None


class MinkAmdError(RuntimeError):
    """A C-ABI call failed; carries the library's error code and message."""

    def __init__(self, code: int, message: str):
        super().__init__(f"mink_amd error {code}: {message}")
        self.code = code


def load_library() -> C.CDLL:
    """Load libmink_amd.so (built by __graft_entry__.build()). Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). mink_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != MK_OK:
        msg = load_library().mk_last_error()
        raise MinkAmdError(code, msg.decode() if msg else "")


def set_option(name: str, value: int) -> None:
    """mk_set_option: a process-wide switch between equivalent kernel paths (include/mink_amd.h)."""
    check(load_library().mk_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    out = C.c_int64(0)
    check(load_library().mk_get_option(name.encode(), C.byref(out)))
    return int(out.value)


class options:
    """Context manager: ``with options(coop=1, stream_blocks=0): ...`` sets the switches and
    restores their previous values on exit (tests and tuning tools)."""

    def __init__(self, **values: int):
        self._values = values
        self._saved: dict = {}

    def __enter__(self):
        for k, v in self._values.items():
            self._saved[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self._saved.items():
            set_option(k, v)
        return False
