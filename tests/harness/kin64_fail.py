import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch, fsae_mpc_amd as fm
import oracle as orc
tr = fm.Track.load("fsg2019"); otr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
ids = np.array([666, 1495, 2738, 4049, 0, 1])
x0, xl, ul, xr = fm.instances(0, 64, 0.05, tr.L, 20190, ids)
q = fm.LtvBatch(0, 64, 0.05, tr, len(ids)).build_qp(dev(x0), dev(xr), dev(xl), dev(ul))
K = ("H", "g", "A", "lb", "ub", "lbA", "ubA")
for pol in (1, 0):
    o = fm.qp_solve_batch_device(*(q[k] for k in K), options=fm.default_opts(polish=pol), want_aux=True)
    torch.cuda.synchronize()
    print(os.path.basename(fm._lib.LIB_PATH), "polish", pol, "flags", o["exitflag"].cpu().numpy(), "iters", o["iter"].cpu().numpy(), "kkt", np.array2string(o["kkt"].cpu().numpy(), precision=1), "polished", o["polished"].cpu().numpy(), flush=True)
if len(sys.argv) > 1:
    qh = {k: q[k].cpu().numpy() for k in K}
    ref = orc.qp_solve_batch_aux(*(qh[k] for k in K))
    print("oracle flags", ref["exitflag"], "iters", ref["iter"], "polished", ref["polished"])
