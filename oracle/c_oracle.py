"""TEST INFRASTRUCTURE ONLY -- ctypes loader of oracle/ik_oracle.c (built by
oracle/Makefile into oracle/_build/).  Used by tests as a fast checker and by
bench.py's cpu_baseline leg; never imported by mink_amd/."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_build", "libik_oracle.so")
_pi = C.POINTER(C.c_int32)
_pf = C.POINTER(C.c_double)


class OrModel(C.Structure):
    _fields_ = [
        ("nq", C.c_int32), ("nv", C.c_int32), ("nbody", C.c_int32), ("njnt", C.c_int32),
        ("ngeom", C.c_int32), ("nsite", C.c_int32),
        ("body_parentid", _pi), ("body_jntadr", _pi), ("body_jntnum", _pi), ("body_mocapid", _pi),
        ("body_pos", _pf), ("body_quat", _pf), ("body_mass", _pf), ("body_ipos", _pf),
        ("jnt_type", _pi), ("jnt_qposadr", _pi), ("jnt_dofadr", _pi), ("jnt_bodyid", _pi), ("jnt_limited", _pi),
        ("jnt_pos", _pf), ("jnt_axis", _pf), ("jnt_range", _pf), ("qpos0", _pf),
        ("dof_parentid", _pi), ("dof_jntid", _pi),
        ("site_bodyid", _pi), ("site_pos", _pf), ("site_quat", _pf),
        ("geom_bodyid", _pi), ("geom_type", _pi), ("geom_size", _pf), ("geom_pos", _pf), ("geom_quat", _pf),
    ]


class OrProblem(C.Structure):
    _fields_ = [
        ("frame_type", C.c_int32), ("frame_id", C.c_int32),
        ("frame_cost", C.c_double * 6), ("frame_gain", C.c_double), ("frame_lm", C.c_double),
        ("nposture", C.c_int32),
        ("posture_cost", _pf), ("posture_target", _pf), ("posture_gain", _pf), ("posture_lm", _pf),
        ("has_cfg_limit", C.c_int32), ("cfg_gain", C.c_double), ("cfg_min_dist", C.c_double),
        ("nvel", C.c_int32), ("vel_dof", _pi), ("vel_max", _pf),
    ]


def available() -> bool:
    return os.path.exists(PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(PATH)
        _lib.or_build_qp.restype = C.c_int
        _lib.or_solve_qp.restype = C.c_int
        _lib.or_solve_ik_batch.restype = None
        _lib.or_iterate_batch.restype = None
        _lib.or_set_polish.restype = None
        _lib.or_set_polish.argtypes = [C.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class COracle:
    """One model + one task/limit set (single frame task, posture-like tasks,
    ConfigurationLimit, VelocityLimit) evaluated by the C restatement."""

    def __init__(self, model, frame=("site", "pinch_site"), frame_cost=(1,) * 6, frame_gain=1.0, frame_lm=1.0,
                 postures=(), cfg_limit=(0.95, 0.0), velocity=None):
        self.model = model
        self._keep = []
        m = OrModel()
        for name in ("nq", "nv", "nbody", "njnt", "ngeom", "nsite"):
            setattr(m, name, int(getattr(model, name)))
        for name, ctype in OrModel._fields_[6:]:
            arr = getattr(model, name)
            arr = np.ascontiguousarray(np.asarray(arr).astype(np.int32 if ctype is _pi else np.float64))
            self._keep.append(arr)
            setattr(m, name, arr.ctypes.data_as(ctype))
        self.m = m
        p = OrProblem()
        if frame is None:
            p.frame_id = -1
        else:
            ftype, fname = frame
            p.frame_type = {"body": 0, "geom": 1, "site": 2}[ftype]
            p.frame_id = {"body": model.body_id, "geom": model.geom_id, "site": model.site_id}[ftype](fname)
        p.frame_cost = (C.c_double * 6)(*[float(x) for x in frame_cost])
        p.frame_gain, p.frame_lm = float(frame_gain), float(frame_lm)
        pc = np.ascontiguousarray([np.broadcast_to(np.asarray(t["cost"], dtype=np.float64), (model.nv,)) for t in postures],
                                  dtype=np.float64).reshape(-1)
        pt = np.ascontiguousarray([t["target_q"] for t in postures], dtype=np.float64).reshape(-1)
        pg = np.ascontiguousarray([t.get("gain", 1.0) for t in postures], dtype=np.float64)
        pl = np.ascontiguousarray([t.get("lm_damping", 0.0) for t in postures], dtype=np.float64)
        self._keep += [pc, pt, pg, pl]
        p.nposture = len(postures)
        p.posture_cost, p.posture_target = pc.ctypes.data_as(_pf), pt.ctypes.data_as(_pf)
        p.posture_gain, p.posture_lm = pg.ctypes.data_as(_pf), pl.ctypes.data_as(_pf)
        p.has_cfg_limit = 1 if cfg_limit is not None else 0
        if cfg_limit is not None:
            p.cfg_gain, p.cfg_min_dist = float(cfg_limit[0]), float(cfg_limit[1])
        velocity = velocity or {}
        vd = np.ascontiguousarray([model.jnt_dofadr[model.joint_id(n)] for n in velocity], dtype=np.int32)
        vm = np.ascontiguousarray(list(velocity.values()), dtype=np.float64)
        self._keep += [vd, vm]
        p.nvel = len(velocity)
        p.vel_dof, p.vel_max = vd.ctypes.data_as(_pi), vm.ctypes.data_as(_pf)
        self.p = p

    def build_qp(self, q, target, dt, damping):
        nv = self.model.nv
        P = np.zeros((nv, nv)); c = np.zeros(nv); G = np.zeros((256, nv)); h = np.zeros(256); e = np.zeros(6)
        q = np.ascontiguousarray(q, dtype=np.float64); target = np.ascontiguousarray(target, dtype=np.float64)
        rows = lib().or_build_qp(C.byref(self.m), C.byref(self.p), _p(q), _p(target), C.c_double(dt),
                                 C.c_double(damping), _p(P), _p(c), _p(G), _p(h), _p(e))
        return P, c, G[:rows].copy(), h[:rows].copy(), e

    def solve_ik(self, q, targets, dt, damping, nthreads=1):
        q = np.ascontiguousarray(q, dtype=np.float64); targets = np.ascontiguousarray(targets, dtype=np.float64)
        B = q.shape[0]
        v = np.zeros((B, self.model.nv)); st = np.zeros(B, dtype=np.int32)
        lib().or_solve_ik_batch(C.byref(self.m), C.byref(self.p), _p(q), _p(targets), C.c_double(dt),
                                C.c_double(damping), C.c_int(B), _p(v), _p(st), C.c_int(nthreads))
        return v, st

    def iterate(self, q, targets, dt, damping, max_iters, pos_tol=-1.0, ori_tol=-1.0, nthreads=1):
        q = np.array(q, dtype=np.float64, order="C"); targets = np.ascontiguousarray(targets, dtype=np.float64)
        B = q.shape[0]
        iters = np.zeros(B, dtype=np.int32); err = np.zeros((B, 6)); st = np.zeros(B, dtype=np.int32)
        lib().or_iterate_batch(C.byref(self.m), C.byref(self.p), _p(q), _p(targets), C.c_double(dt),
                               C.c_double(damping), C.c_int(B), C.c_int(max_iters), C.c_double(pos_tol),
                               C.c_double(ori_tol), _p(iters), _p(err), _p(st), C.c_int(nthreads))
        return q, iters, err, st


def solve_qp(P, c, G, h):
    P = np.ascontiguousarray(P, dtype=np.float64); c = np.ascontiguousarray(c, dtype=np.float64)
    n = P.shape[0]
    m = 0 if G is None else G.shape[0]
    G = np.ascontiguousarray(G if m else np.zeros((1, n)), dtype=np.float64)
    h = np.ascontiguousarray(h if m else np.zeros(1), dtype=np.float64)
    x = np.zeros(n)
    st = lib().or_solve_qp(_p(P), _p(c), _p(G), _p(h), C.c_int(n), C.c_int(m), _p(x))
    return x, st
