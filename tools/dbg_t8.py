import sys, numpy as np
sys.path.insert(0, "/root/repo")
import torch, fsae_mpc_amd as fm
B = 24
tr = fm.Track.load("fsg2019")
x0, xl, ul, xr = fm.instances(fm.DYNAMIC, 60, 0.05, tr.L, 20190, range(B))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
q = fm.LtvBatch(fm.DYNAMIC, 60, 0.05, tr, B).build_qp(up(x0), up(xr), up(xl), up(ul))
out = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA"))); torch.cuda.synchronize()
print("flags", out["exitflag"].cpu().numpy(), "iters", out["iter"].cpu().numpy())
