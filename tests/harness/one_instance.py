#!/usr/bin/env python3
"""One instance id of the synthetic family: the QP as the construction kernel builds it against the oracle's construction, and the
solve of either through the HIP path and the oracle.  usage: one_instance.py <kin|dyn> <N> <id> [chunk=1]"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
import fsae_mpc_amd as fm
import oracle as orc
model = 1 if sys.argv[1].startswith("dyn") else 0
N, iid = int(sys.argv[2]), int(sys.argv[3])
tr = fm.Track.load("fsg2019"); otr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, [iid])
qg = fm.LtvBatch(model, N, 0.05, tr, 1).build_qp(up(x0), up(xr), up(xl), up(ul))
qo = orc.build_qp_batch(model, otr, N, 0.05, x0, xr, xl, ul)
K = ("H", "g", "A", "lb", "ub", "lbA", "ubA")
for k in K:
    a, b = qg[k].cpu().numpy(), qo[k]
    print("%-4s finite gpu %s | max |gpu - oracle| %.3e (max |.| %.3e)" % (k, bool(np.isfinite(a).all()), np.abs(a - b).max(), np.abs(b).max()))
for name, q in (("GPU-built", {k: qg[k] for k in K}), ("oracle-built", {k: up(qo[k]) for k in K})):
    o = fm.qp_solve_batch_device(*(q[k] for k in K), want_aux=True); torch.cuda.synchronize()
    qq = {k: q[k].cpu().numpy() for k in K}
    r = orc.qp_solve_batch_aux(*(qq[k] for k in K))
    print("%-13s HIP: flag %d iter %d kkt %.2e polished %d x finite %s | oracle: flag %d iter %d kkt %.2e" % (name, o["exitflag"][0].item(), o["iter"][0].item(), o["kkt"][0].item(), o["polished"][0].item(), bool(torch.isfinite(o["x"]).all().item()), r["exitflag"][0], r["iter"][0], r["kkt"][0]))
