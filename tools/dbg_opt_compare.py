#!/usr/bin/env python3
"""Diagnostic: solve the same batches with the library named by FSAEMPC_LIB and save x / iterations (to compare an -O1 build
of the same source with the shipped -O3 build: both must compute the same iterates up to round-off)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm
tag = sys.argv[1]
res = {}
tr = fm.Track.load("fsg2019")
# shapes the -O1 guard library carries: T = 5 (kin N=40 border 1, dyn N=40 border 4) and T = 8 (dyn N=60)
for name, model, N, B in (("kin40", fm.KINEMATIC, 40, 512), ("dyn40", fm.DYNAMIC, 40, 96), ("dyn60", fm.DYNAMIC, 60, 24)):
    x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    q = fm.LtvBatch(model, N, 0.05, tr, B).build_qp(up(x0), up(xr), up(xl), up(ul))
    out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA"))); torch.cuda.synchronize()
    res[name + "_x"] = out["x"].cpu().numpy(); res[name + "_it"] = out["iter"].cpu().numpy(); res[name + "_fl"] = out["exitflag"].cpu().numpy()
out_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/optcmp_%s.npz" % tag
np.savez(out_path, **res)
print(tag, {k: (int(v.sum()) if k.endswith("_it") else None) for k, v in res.items() if k.endswith("_it")})
