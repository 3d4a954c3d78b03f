"""Loader for the TEST-ONLY host build of the fused device functions
(tests/hostsim/hostsim.hip).  Used by CPU tests to check the structured QP path
lane by lane against the oracle without a GPU."""
import ctypes as C
import os

import numpy as np

from mink_amd import _lib as L

PATH = os.environ.get("HOSTSIM_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "libhostsim.so")


def available():
    return os.path.exists(PATH)


_h = None


def _load():
    global _h
    if _h is None:
        _h = C.CDLL(PATH)
        _h.hostsim_solve_ik.restype = C.c_int
        _h.hostsim_iterate.restype = C.c_int
        _h.hostsim_coop_solve_ik.restype = C.c_int
        _h.hostsim_coop_iterate.restype = C.c_int
        _h.mk_last_error.restype = C.c_char_p
    return _h


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def solve_ik(model, task_specs, limit_specs, q, targets, dt, damping, form="lane", refine=0):
    """form = "lane": the lane-per-problem device functions of mk_fused.hip; "coop": the
    cooperative ones of mk_coop.hip, their 16 lanes emulated by fibers."""
    h = _load()
    md, k1 = L.pack_model(model)
    ta, k2 = L.pack_tasks(task_specs)
    la, k3 = L.pack_limits(limit_specs)
    q = np.ascontiguousarray(q, dtype=np.float64)
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    B = q.shape[0]
    v = np.zeros((B, model.nv))
    status = np.zeros(B, dtype=np.int32)
    if form == "coop":
        rc = h.hostsim_coop_solve_ik(C.byref(md), ta, C.c_int(len(task_specs)), la, C.c_int(len(limit_specs)), _ptr(q),
                                     _ptr(targets), C.c_double(dt), C.c_double(damping), C.c_int(B), _ptr(v),
                                     _ptr(status), C.c_int(refine))
    else:
        rc = h.hostsim_solve_ik(C.byref(md), ta, C.c_int(len(task_specs)), la, C.c_int(len(limit_specs)), _ptr(q),
                                _ptr(targets), C.c_double(dt), C.c_double(damping), C.c_int(B), _ptr(v), _ptr(status))
    if rc != 0:
        raise RuntimeError(f"hostsim rc={rc}: {h.mk_last_error().decode()}")
    return v, status


def iterate(model, task_specs, limit_specs, q, targets, dt, damping, max_iters, pos_tol, ori_tol, form="lane",
            refine=0):
    h = _load()
    md, k1 = L.pack_model(model)
    ta, k2 = L.pack_tasks(task_specs)
    la, k3 = L.pack_limits(limit_specs)
    q = np.array(q, dtype=np.float64, order="C")
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    B = q.shape[0]
    v = np.zeros((B, model.nv))
    iters = np.zeros(B, dtype=np.int32)
    err = np.zeros((B, 6))
    status = np.zeros(B, dtype=np.int32)
    args = (C.byref(md), ta, C.c_int(len(task_specs)), la, C.c_int(len(limit_specs)), _ptr(q),
            _ptr(targets), C.c_double(dt), C.c_double(damping), C.c_int(B), C.c_int(max_iters),
            C.c_double(pos_tol), C.c_double(ori_tol), _ptr(v), _ptr(iters), _ptr(err), _ptr(status))
    rc = h.hostsim_coop_iterate(*args, C.c_int(refine)) if form == "coop" else h.hostsim_iterate(*args)
    if rc != 0:
        raise RuntimeError(f"hostsim rc={rc}: {h.mk_last_error().decode()}")
    return q, v, iters, err, status


def waypoint_step(cur, goal, pos_thr, pos_step, rot_thr):
    """The device waypoint generator on [B, 7] current / goal poses (wxyz_xyz)."""
    h = _load()
    cur = np.ascontiguousarray(cur, dtype=np.float64)
    goal = np.ascontiguousarray(goal, dtype=np.float64)
    B = cur.shape[0]
    cmd = np.zeros((B, 7))
    reached = np.zeros(B, dtype=np.int32)
    h.hostsim_waypoint_step(_ptr(cur), _ptr(goal), C.c_int(B), C.c_double(pos_thr), C.c_double(pos_step),
                            C.c_double(rot_thr), _ptr(cmd), _ptr(reached))
    return cmd, reached.astype(bool)


def solve_rows(model, task_specs, limit_specs, q, targets, dt, damping):
    """The lane form of the staged solve's middle stage (collision rows + QP with them) on
    every problem: (v [B, nv], good [B], rows in range [B])."""
    h = _load()
    h.hostsim_solve_rows.restype = C.c_int
    md, k1 = L.pack_model(model)
    ta, k2 = L.pack_tasks(task_specs)
    la, k3 = L.pack_limits(limit_specs)
    q = np.ascontiguousarray(q, dtype=np.float64)
    targets = np.ascontiguousarray(targets, dtype=np.float64)
    B = q.shape[0]
    v = np.zeros((B, model.nv))
    good = np.zeros(B, dtype=np.int32)
    nrows = np.zeros(B, dtype=np.int32)
    rc = h.hostsim_solve_rows(C.byref(md), ta, C.c_int(len(task_specs)), la, C.c_int(len(limit_specs)), _ptr(q),
                              _ptr(targets), C.c_double(dt), C.c_double(damping), C.c_int(B), _ptr(v), _ptr(good),
                              _ptr(nrows))
    if rc != 0:
        raise RuntimeError(f"hostsim rc={rc}: {h.mk_last_error().decode()}")
    return v, good, nrows


def box_box(c1, q1, h1, c2, q2, h2, distmax):
    """The device's one-lane box-box distance on B pairs: (in range [B], dist [B], p1 [B,3], p2 [B,3])."""
    h = _load()
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (c1, q1, h1, c2, q2, h2)]
    B = arrs[0].shape[0]
    out = np.zeros((B, 8))
    h.hostsim_box_box(*[_ptr(a) for a in arrs], C.c_int(B), C.c_double(distmax), _ptr(out))
    return out[:, 0] > 0, out[:, 1], out[:, 2:5], out[:, 5:8]


def geom_distance(t1, c1, q1, s1, t2, c2, q2, s2, distmax):
    """The device's one-lane distance of primitive pairs of types (t1 <= t2) on B pairs:
    (in range [B], dist [B], p1 [B,3], p2 [B,3])."""
    h = _load()
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (c1, q1, s1, c2, q2, s2)]
    B = arrs[0].shape[0]
    out = np.zeros((B, 8))
    h.hostsim_geom_distance(C.c_int(t1), _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), C.c_int(t2), _ptr(arrs[3]),
                            _ptr(arrs[4]), _ptr(arrs[5]), C.c_int(B), C.c_double(distmax), _ptr(out))
    return out[:, 0] > 0, out[:, 1], out[:, 2:5], out[:, 5:8]
