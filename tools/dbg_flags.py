import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
import torch, fsae_mpc_amd as fm
B = 512
tr = fm.Track.load("fsg2019")
x0, xl, ul, xr = fm.instances(fm.KINEMATIC, 40, 0.05, tr.L, 20190, range(B))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
q = fm.LtvBatch(fm.KINEMATIC, 40, 0.05, tr, B).build_qp(up(x0), up(xr), up(xl), up(ul))
for rep in range(2):
    out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")))
    torch.cuda.synchronize()
    fl = out["exitflag"].cpu().numpy(); it = out["iter"].cpu().numpy()
    print("flags", dict(zip(*np.unique(fl, return_counts=True))), "iters mean", it.mean(), "first its", it[:16])
