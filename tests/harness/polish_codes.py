import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch, fsae_mpc_amd as fm
tr = fm.Track.load("fsg2019")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for model, N, B in ((1, 60, 2048), (1, 80, 1024), (1, 40, 2048), (0, 64, 2048)):
    x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, range(B))
    q = fm.LtvBatch(model, N, 0.05, tr, B).build_qp(dev(x0), dev(xr), dev(xl), dev(ul))
    o = fm.qp_solve_batch_device(*(q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")), want_aux=True)
    torch.cuda.synchronize()
    p = o["polished"].cpu().numpy(); k = o["kkt"].cpu().numpy()
    print("model", model, "N", N, "polished codes", dict(zip(*[a.tolist() for a in np.unique(p, return_counts=True)])), "on vertex %.3f" % (p > 0).mean(), "| kkt of rejected: median %.1e max %.1e" % (np.median(k[p <= 0]) if (p <= 0).any() else 0, k[p <= 0].max() if (p <= 0).any() else 0), flush=True)
