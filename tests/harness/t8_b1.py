import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch, fsae_mpc_amd as fm
tr = fm.Track.load("fsg2019")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for ids in ([2], [5], [2, 5], [0, 1, 2], [2, 2, 2, 2], list(range(6)), [5, 4, 3, 2, 1, 0]):
    x0, xl, ul, xr = fm.instances(0, 64, 0.05, tr.L, 515, np.array(ids))
    q = fm.LtvBatch(0, 64, 0.05, tr, len(ids)).build_qp(dev(x0), dev(xr), dev(xl), dev(ul))
    o = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")))
    torch.cuda.synchronize()
    print(os.path.basename(fm._lib.LIB_PATH), "ids", ids, "flags", o["exitflag"].cpu().numpy(), "iters", o["iter"].cpu().numpy(), flush=True)
