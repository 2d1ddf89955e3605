#!/usr/bin/env python3
"""Solves the QPs stored by `tools/cl_failures_check.py --save f.npz` again on the GPU, one at a time, and with the oracle.
usage: tests/harness/solve_saved_qps.py f.npz [key-prefix]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import fsae_mpc_amd as fm
import oracle as orc

z = np.load(sys.argv[1])
pref = sys.argv[2] if len(sys.argv) > 2 else "f"
keys = sorted({k.rsplit("_", 1)[0] for k in z.files if k.endswith("_H") and k.startswith(pref)})
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for key in keys:
    e = {k_: z[key + "_" + k_] for k_ in ("H", "g", "A", "lb", "ub", "lbA", "ubA")}
    if not all(np.isfinite(e[k_]).all() for k_ in ("H", "g", "A")):
        print(key, "non-finite data"); continue
    o = fm.qp_solve_batch_device(*(dev(e[k_][None]) for k_ in ("H", "g", "A", "lb", "ub", "lbA", "ubA")), want_aux=True)
    x, f, fl, it, lam = orc.qp_solve(e["H"].T, e["g"], e["A"].T, e["lb"], e["ub"], e["lbA"], e["ubA"])
    print(key, "gpu flag %d iter %d kkt %.2e | oracle flag %d iter %d | max|x_gpu - x_orc| %.2e" % (
        int(o["exitflag"][0]), int(o["iter"][0]), float(o["kkt"][0]), fl, it, float(np.abs(o["x"][0].cpu().numpy() - x).max())))
