"""Soak test (GPU box): mk_qp_solve_batch on random strictly convex QPs (sizes, conditioning, numbers of
general rows, feasible by construction) -- KKT residuals of a sample of its answers computed on the host from the
inputs themselves, and the same sample compared with the C restatement's dense dual method.

    PYTHONPATH=. python tools/soak_qp.py [problems per set]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mink_amd.engine import qp_solve
from oracle import c_oracle as co

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda:0"
print(f"# soak: mk_qp_solve_batch on random QPs, {B} problems per set\n")
print("| n | m | cond(P) | status nonzero | max infeasibility | max stationarity (relative) | max complementarity | "
      "active rows mean | max abs dx vs C restatement (256 sampled) |")
print("|---|---|---|---|---|---|---|---|---|")
k = 0
for n in (4, 10, 24, 40, 64):
    for m in (0, 8, 40):
        for cond in (1e2, 1e6, 1e9):
            k += 1
            rng = np.random.default_rng([808, k])
            Q = np.linalg.qr(rng.normal(size=(B, n, n)))[0]
            ev = np.exp(rng.uniform(0, np.log(cond), size=(B, n)))
            ev[:, 0], ev[:, -1] = 1.0, cond
            P = (Q * ev[:, None, :]) @ Q.transpose(0, 2, 1)
            P = 0.5 * (P + P.transpose(0, 2, 1))
            c = rng.normal(size=(B, n)) * 3 * np.sqrt(cond)
            G = rng.normal(size=(B, max(m, 1), n))
            x_feas = rng.normal(size=(B, n)) * 0.1
            h = np.einsum("bij,bj->bi", G, x_feas) + rng.uniform(0.01, 1.0, size=(B, max(m, 1)))
            tP, tc = torch.tensor(P, device=dev), torch.tensor(c, device=dev)
            tG, th = (torch.tensor(G, device=dev), torch.tensor(h, device=dev)) if m else (None, None)
            try:
                x, st = qp_solve(tP, tc, tG, th)
            except Exception as e:  # noqa: BLE001
                print(f"| {n} | {m} | {cond:.0e} | refused: {str(e)[:80]} |")
                continue
            torch.cuda.synchronize()
            ok = st == 0
            okn = ok.cpu().numpy()
            xs_all = x.cpu().numpy()
            # KKT residuals on a sample, on the host: bound violation, stationarity with multipliers of the rows
            # at their bounds by least squares, sign of those multipliers
            infeas_m = stat_m = comp_m = 0.0
            nact_sum, nsamp = 0, 0
            for b in range(0, B, max(1, B // 256)):
                if not okn[b]:
                    continue
                xb = xs_all[b]
                gb = P[b] @ xb + c[b]
                scale = max(np.abs(c[b]).max() + (np.abs(P[b]) @ np.abs(xb)).max(), 1e-300)
                if m:
                    r = G[b] @ xb - h[b]
                    infeas_m = max(infeas_m, float(np.clip(r, 0, None).max()))
                    act = r > -1e-6 * (1.0 + np.abs(h[b]))
                    lam = np.zeros(m)
                    if act.any():
                        lam[act] = np.linalg.lstsq(G[b][act].T, -gb, rcond=None)[0]
                    stat_m = max(stat_m, float(np.abs(gb + G[b].T @ lam).max() / scale))
                    comp_m = max(comp_m, float(np.clip(-lam, 0, None).max() / scale))
                    nact_sum += int(act.sum())
                else:
                    stat_m = max(stat_m, float(np.abs(gb).max() / scale))
                nsamp += 1
            dx = 0.0
            xs = xs_all
            for b in range(0, B, max(1, B // 256)):
                xr, s = co.solve_qp(P[b], c[b], G[b] if m else None, h[b] if m else None)
                if s == 0 and bool(ok[b]):
                    dx = max(dx, float(np.abs(xs[b] - xr).max() / max(1.0, np.abs(xr).max())))
            print(f"| {n} | {m} | {cond:.0e} | {int((~ok).sum())} | {infeas_m:.1e} | {stat_m:.1e} | {comp_m:.1e} | "
                  f"{nact_sum / max(nsamp, 1):.1f} | {dx:.1e} |", flush=True)
