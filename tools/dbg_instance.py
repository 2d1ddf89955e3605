#!/usr/bin/env python3
"""Diagnostic: exit flags over a range of synthetic instance ids, details of the failing ones."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm
lo, hi = int(sys.argv[1]), int(sys.argv[2])
model = fm.DYNAMIC if (len(sys.argv) > 3 and sys.argv[3] == "dyn") else fm.KINEMATIC
N = int(sys.argv[4]) if len(sys.argv) > 4 else 40
tr = fm.Track.load("fsg2019")
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, np.arange(lo, hi))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
q = fm.LtvBatch(model, N, 0.05, tr, hi - lo).build_qp(up(x0), up(xr), up(xl), up(ul))
for pol in (1, 0):
    out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")), options=fm.default_opts(polish=pol)); torch.cuda.synchronize()
    fl = out["exitflag"].cpu().numpy(); it = out["iter"].cpu().numpy()
    bad = np.nonzero(fl != 0)[0]
    print("polish", pol, "flags", dict(zip(*np.unique(fl, return_counts=True))), "bad ids", (bad + lo).tolist(), "iters", it[bad].tolist(), "x0", np.round(x0[bad], 3).tolist())
