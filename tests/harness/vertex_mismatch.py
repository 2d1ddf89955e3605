#!/usr/bin/env python3
"""Adjudicates instances on which the HIP path and the oracle both report the refined vertex yet disagree in x by more than
X_TOL_VERTEX: working sets of both (from the multipliers), the vertex recomputed from each working set with dense numpy algebra
(tests/kkt_numpy.py: vertex_from_working_set), objective values and feasibility of every candidate.
usage: vertex_mismatch.py [model=1] [N=60] [B=24]"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import fsae_mpc_amd as fm
import oracle as orc
from kkt_numpy import kkt_certificate, vertex_from_working_set, working_set

model = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B = int(sys.argv[3]) if len(sys.argv) > 3 else 24
otr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x0, xl, ul, xr = fm.instances(model, N, 0.05, otr.L, 20190, range(B))
q = orc.build_qp_batch(model, otr, N, 0.05, x0, xr, xl, ul, threads=8)
KEYS = ("H", "g", "A", "lb", "ub", "lbA", "ubA")
o = fm.qp_solve_batch_device(*(dev(q[k]) for k in KEYS), want_lambda=True, want_aux=True)
torch.cuda.synchronize()
out = {k: v.cpu().numpy() for k, v in o.items() if v is not None and k != "workspace"}
ref = orc.qp_solve_batch_aux(*(q[k] for k in KEYS), threads=8)
ex = np.abs(out["x"] - ref["x"]).max(axis=1) / np.maximum(1, np.abs(ref["x"]).max(axis=1))
both = (out["polished"] > 0) & (ref["polished"] > 0)
print("x err where both on the vertex:", np.sort(ex[both])[-5:])
for b in np.where(both & (ex > 1e-6))[0]:
    H, g, A = q["H"][b].T, q["g"][b], q["A"][b].T
    lb, ub, lbA, ubA = q["lb"][b], q["ub"][b], q["lbA"][b], q["ubA"][b]
    def obj(x): return 0.5 * x @ H @ x + g @ x
    def feas(x):
        v = A @ x
        return max(np.max(np.maximum(lb - x, 0) / np.maximum(1, np.abs(lb))), np.max(np.maximum(x - ub, 0) / np.maximum(1, np.abs(ub))),
                   np.max(np.maximum(np.where(lbA > -1e9, lbA - v, 0), 0) / np.maximum(1, np.abs(v))), np.max(np.maximum(np.where(ubA < 1e9, v - ubA, 0), 0) / np.maximum(1, np.abs(v))))
    print("instance", b, "x err %.3e" % ex[b], "iters gpu/oracle", out["iter"][b], ref["iter"][b], "polished", out["polished"][b], ref["polished"][b])
    cand = {}
    for name, sol in (("gpu", out), ("oracle", ref)):
        x, lam = sol["x"][b], sol["lam"][b]
        ws = working_set(lb, ub, lbA, ubA, x, A @ x, lam)
        c = kkt_certificate(*(q[k][b:b + 1] for k in KEYS), x[None], lam[None])
        xv, lamv = vertex_from_working_set(H, g, A, lb, ub, lbA, ubA, ws)[:2]
        cand[name] = (x, ws, xv)
        print("  %-6s obj %.12e  infeas %.2e  cert st %.2e pr %.2e sg %.2e cp %.2e | active bounds %d rows %d | vertex from its working set: obj %.12e infeas %.2e |x - x_ws| %.2e"
              % (name, obj(x), feas(x), c["stationarity"][0], c["primal"][0], c["sign"][0], c["complementarity"][0], int((ws[:len(x)] != 0).sum()), int((ws[len(x):] != 0).sum()),
                 obj(xv), feas(xv), np.abs(x - xv).max() / max(1, np.abs(xv).max())))
    wg, wo = cand["gpu"][1], cand["oracle"][1]
    d = np.where(wg != wo)[0]
    print("  working sets differ at", d, "gpu", wg[d], "oracle", wo[d])
    print("  |x_ws(gpu) - x_ws(oracle)| rel %.3e" % (np.abs(cand["gpu"][2] - cand["oracle"][2]).max() / max(1, np.abs(cand["oracle"][2]).max())))
