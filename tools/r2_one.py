#!/usr/bin/env python3
"""One shape, a few solves (for rocprofv3 counter passes).  usage: r2_one.py [kin|dyn] N B reps"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm
model = fm.KINEMATIC if (len(sys.argv) < 2 or sys.argv[1] == "kin") else fm.DYNAMIC
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
tr = fm.Track.load("fsg2019")
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
q = fm.LtvBatch(model, N, 0.05, tr, B).build_qp(up(x0), up(xr), up(xl), up(ul))
args = [q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")]
ws = None
for _ in range(reps):
    out = fm.qp_solve_batch_device(*args, workspace=ws); ws = out["workspace"]
torch.cuda.synchronize()
print("flags0", int((out["exitflag"] == 0).sum()), "iters", float(out["iter"].double().mean()))
