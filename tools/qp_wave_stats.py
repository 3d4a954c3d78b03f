"""Tuning helper (CPU): active-set work of the fused QP per 64-problem wave, from the host
build of the device code.  Prints mean per-lane and mean per-wave (max over 64 lanes)
phase-1 rounds and phase-2 steps for the bench workload."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostsim_lib as H
from mink_amd import sharding as S
from mink_amd import workloads as W
from mink_amd.mjcf import load_builtin
from oracle import ik_oracle as o

model = load_builtin("tidybot_cube")
B, K = 2048, 50
q0, qs = S.global_rows(model, 4, 0, B)
tg = np.array([o.Configuration(model, q).get_transform_frame_to_world("pinch_site", "site") for q in qs])
if len(sys.argv) > 1 and sys.argv[1] == "near":
    base = np.array([o.Configuration(model, q).get_transform_frame_to_world("pinch_site", "site") for q in q0])
    tg = W.offset_targets(base, np.random.default_rng(1))
limits = W.limit_specs(model, True)
tols = (-1.0, -1.0)
if len(sys.argv) > 1 and sys.argv[1] == "nearloop":
    # the callers' loop as the bench runs it: offset targets, default ConfigurationLimit only,
    # <= 20 iterations with the 1e-4 / 1e-4 exit test (finished problems count as idle lanes)
    base = np.array([o.Configuration(model, q).get_transform_frame_to_world("pinch_site", "site") for q in q0])
    tg = W.offset_targets(base, np.random.default_rng(1))
    limits, tols, K = W.limit_specs(model, False), (1e-4, 1e-4), 20
h = H._load()
trace = np.zeros((B, K, 3), dtype=np.int32)
h.hostsim_set_trace(trace.ctypes.data_as(C.c_void_p), C.c_int(K))
H.iterate(model, W.task_specs(model, False), limits, q0, tg, W.DT, W.DAMPING, K, tols[0], tols[1])
h.hostsim_set_trace(None, C.c_int(0))
r, p, sg = trace[..., 0], trace[..., 1], trace[..., 2]
rw = r.reshape(B // 64, 64, K).max(axis=1)
pw = p.reshape(B // 64, 64, K).max(axis=1)
sw = sg.reshape(B // 64, 64, K).max(axis=1)
# instruction estimate per wave (VALU wave-instructions): assembly + integration ~4400, a phase-1
# round ~850, a phase-2 step ~770; an iteration costs a wave its most expensive lane
busy = rw > 0
cost = (4400 * busy + 850 * rw + 770 * pw).sum()
print("solves %d  estimated wave-instructions per executed solve %.0f" % ((r > 0).sum(), 64.0 * cost / max((r > 0).sum(), 1)))
print("per lane : rounds %.3f  phase2 %.3f  single %.3f" % (r.mean(), p.mean(), sg.mean()))
print("per wave : rounds %.3f  phase2 %.3f  single %.3f" % (rw.mean(), pw.mean(), sw.mean()))
print("rounds per wave by iteration:", np.round(rw.mean(axis=0), 2).tolist())
print("phase2 per wave by iteration:", np.round(pw.mean(axis=0), 2).tolist())
print("hist rounds/lane:", np.bincount(r.ravel())[:10].tolist(), " rounds/wave:", np.bincount(rw.ravel())[:12].tolist())
