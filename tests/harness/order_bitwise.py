#!/usr/bin/env python3
"""Launch order on / off over a whole chunk of instance ids: every output must be bit-identical; prints the ids that
differ, with their exit flags / iteration counts either way.  usage: order_bitwise.py [kin|dyn] [N=40] [lo=4096] [hi=8192] [repeat=2]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import fsae_mpc_amd as fm
model = fm.DYNAMIC if (len(sys.argv) > 1 and sys.argv[1].startswith("dyn")) else fm.KINEMATIC
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
rep = int(sys.argv[5]) if len(sys.argv) > 5 else 2
tr = fm.Track.load("fsg2019")
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
ids = np.arange(lo, hi)
x0, xl, ul, xr = fm.instances(model, N, 0.05, tr.L, 20190, ids)
q = fm.LtvBatch(model, N, 0.05, tr, len(ids)).build_qp(up(x0), up(xr), up(xl), up(ul))
args = [q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")]
runs = []
for tag, env in [("off", "0"), ("on", None)] * rep:
    if env is None: os.environ.pop("FSAEMPC_QP_ORDER", None)
    else: os.environ["FSAEMPC_QP_ORDER"] = env
    o = fm.qp_solve_batch_device(*args, want_lambda=True, want_aux=True); torch.cuda.synchronize()
    runs.append((tag, {k: o[k].cpu().numpy() for k in ("x", "fval", "exitflag", "iter", "lam", "kkt", "polished")}))
ref = runs[0][1]
for tag, r in runs:
    diff = np.zeros(len(ids), bool)
    for k in r: diff |= (r[k].reshape(len(ids), -1) != ref[k].reshape(len(ids), -1)).any(1) & ~(np.isnan(r[k].reshape(len(ids), -1)) & np.isnan(ref[k].reshape(len(ids), -1))).all(1)
    bad = np.where(diff)[0]
    print("order %-3s vs first run: %d instances differ %s | flags %s" % (tag, len(bad), [(int(ids[i]), int(ref["exitflag"][i]), int(ref["iter"][i]), int(r["exitflag"][i]), int(r["iter"][i])) for i in bad[:8]],
                                                                      {int(k): int(v) for k, v in zip(*np.unique(r["exitflag"], return_counts=True))}), flush=True)
