#!/usr/bin/env python3
"""Probe: how early can the interior-point phase stop if the active-set refinement lands on the vertex anyway?  Runs the headline
batch with looser (tol, tol_x) and reports time, share on the vertex, and the distance to the default run's x."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm

model = fm.KINEMATIC if (len(sys.argv) < 2 or sys.argv[1] == "kin") else fm.DYNAMIC
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
tr = fm.Track.load("fsg2019")
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
q = fm.LtvBatch(model, N, 0.05, tr, B).build_qp(up(x0), up(xr), up(xl), up(ul))
args = [q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")]
ref = None
for tol, tolx in ((1e-8, 1e-7), (1e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1e-4), (1e-4, 1e-3)):
    o = fm.default_opts(tol=tol, tol_x=tolx, tol_loose=max(1e-6, tol))
    out = fm.qp_solve_batch_device(*args, options=o, want_aux=True); ws = out["workspace"]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        out = fm.qp_solve_batch_device(*args, options=o, workspace=ws, want_aux=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    x = out["x"].cpu().numpy(); pol = out["polished"].cpu().numpy() > 0; fl = out["exitflag"].cpu().numpy()
    if ref is None: ref = (x, pol)
    both = pol & ref[1]
    ex = np.abs(x - ref[0]).max(axis=1) / np.maximum(1, np.abs(ref[0]).max(axis=1))
    print("tol %.0e tol_x %.0e: %.3f ms  iters %.2f  flags0 %d  on vertex %.4f  | x vs default run: max on common vertex set %.1e, max overall %.1e, kkt max %.1e"
          % (tol, tolx, 1e3 * dt, out["iter"].double().mean().item(), int((fl == 0).sum()), pol.mean(), ex[both].max(), ex.max(), out["kkt"].max().item()), flush=True)
