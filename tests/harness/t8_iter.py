import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch, fsae_mpc_amd as fm
tr = fm.Track.load("fsg2019")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, xl, ul, xr = fm.instances(0, 64, 0.05, tr.L, 515, np.array([int(sys.argv[2]) if len(sys.argv) > 2 else 2]))
q = fm.LtvBatch(0, 64, 0.05, tr, 1).build_qp(dev(x0), dev(xr), dev(xl), dev(ul))
res = {}
for k in range(0, int(sys.argv[3]) if len(sys.argv) > 3 else 26):
    o = fm.qp_solve_batch_device(*(q[kk] for kk in ("H", "g", "A", "lb", "ub", "lbA", "ubA")), options=fm.default_opts(max_iter=k, polish=0), want_lambda=True, want_aux=True)
    torch.cuda.synchronize()
    res["x%d" % k] = o["x"].cpu().numpy()[0]; res["l%d" % k] = o["lam"].cpu().numpy()[0]; res["f%d" % k] = np.array([o["exitflag"].item(), o["iter"].item(), o["kkt"].item(), o["fval"].item()])
np.savez(sys.argv[1], **res)
