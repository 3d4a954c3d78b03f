"""Experiment: in benchmark config 4, how many problems have a collision row in range, how many
have one that binds at the optimum, and how the minimum distance is distributed."""
import numpy as np, torch
from mink_amd import workloads as W
from mink_amd.engine import DeviceModel, DeviceProblem
from mink_amd.mjcf import load_builtin

model = load_builtin("tidybot_cube")
dev = torch.device("cuda:0")
dm = DeviceModel(model, dev)
tasks, limits = W.config4_specs(model)
full = DeviceProblem(dm, tasks, limits)
nocoll = DeviceProblem(dm, tasks, limits[:-1])
n = 32768
q, t = W.config4_inputs(model, dm, n, 7)
print("staged", full.staged, "fused(nocoll)", nocoll.fused, "rows", full.n_rows)
v1, s1 = full.solve(q, t, W.DT, W.DAMPING)
v0, s0 = nocoll.solve(q, t, W.DT, W.DAMPING)
d = (v1 - v0).abs().amax(dim=1)
P, c, G, h = full.build_qp(q, t, W.DT, W.DAMPING)
hc = h[:, -8:]
inrange = torch.isfinite(hc).any(dim=1)
print("in range (exact)", float(inrange.float().mean()))
print("solution differs >1e-9", float((d > 1e-9).float().mean()), " >1e-6", float((d > 1e-6).float().mean()))
gx = torch.einsum("bmn,bn->bm", G[:, -8:], v0 * W.DT)
viol = ((gx > hc + 1e-12) & torch.isfinite(hc)).any(dim=1)
print("box solution violates a collision row", float(viol.float().mean()))
nfin = torch.isfinite(hc).sum(dim=1)
print("rows in range per in-range problem: mean", float(nfin[inrange].float().mean()), "max", int(nfin.max()))
hmin = torch.where(torch.isfinite(hc), hc, torch.full_like(hc, 1e9)).amin(dim=1)
for thr in (0.0, 1e-9, 0.1, 0.5, 1.0):
    print("min h <=", thr, float(((hmin <= thr) & inrange).float().mean()))
print("status nonzero", int((s1 != 0).sum()), int((s0 != 0).sum()))
