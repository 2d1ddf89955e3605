#!/usr/bin/env python3
"""Polish diagnostics on the GPU: accepted fraction, x error against the oracle's LU polish, KKT certificate."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import fsae_mpc_amd as fm
import oracle as orc

def run(model, N, B):
    otr = orc.Track.load(fm.tracks._HERE + "/tracks/fsg2019.json")
    x0, xl, ul, xr = fm.instances(model, N, 0.05, otr.L, 20190, range(B))
    q = orc.build_qp_batch(model, otr, N, 0.05, x0, xr, xl, ul)
    xo, fo, flo, ito, lamo, _ = orc.qp_solve_batch(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    out = fm.qp_solve_batch_device(dev(q["H"]), dev(q["g"]), dev(q["A"]), dev(q["lb"]), dev(q["ub"]), dev(q["lbA"]), dev(q["ubA"]), want_lambda=True, want_aux=True)
    torch.cuda.synchronize()
    pol = out["polished"]
    # refinement codes: 1 accepted; -1 stationarity, -2 primal feasibility, -3 complementarity, -4 multiplier sign,
    # -5 factorisation, -6 CG did not converge in 7 steps, -7 CG breakdown (dependent / inconsistent working set)
    code = pol.cpu().numpy(); pol = code > 0
    print("codes:", dict(zip(*np.unique(code, return_counts=True))))
    x = out["x"].cpu().numpy(); lam = out["lam"].cpu().numpy(); fl = out["exitflag"].cpu().numpy()
    ex = np.abs(x - xo).max(axis=1) / np.maximum(1, np.abs(xo).max(axis=1))
    df = (out["fval"].cpu().numpy() - fo) / np.maximum(1, np.abs(fo))
    print("fval gpu-oracle rel: polished min %.1e max %.1e | unpolished min %.1e max %.1e" % (df[pol].min(), df[pol].max(), df[~pol].min() if (~pol).any() else 0, df[~pol].max() if (~pol).any() else 0))
    kk = np.array([orc.qp_kkt(q["H"][b].T, q["g"][b], q["A"][b].T, q["lb"][b], q["ub"][b], q["lbA"][b], q["ubA"][b], x[b], lam[b])[1] for b in range(B)])
    print("model %d N %d B %d: flags0 %d polished %.3f | x err polished: med %.1e p99 %.1e max %.1e | unpolished: med %.1e max %.1e | kkt max polished %s unpolished %s"
          % (model, N, B, (fl == 0).sum(), pol.mean(), np.median(ex[pol]), np.percentile(ex[pol], 99), ex[pol].max(),
             np.median(ex[~pol]) if (~pol).any() else 0, ex[~pol].max() if (~pol).any() else 0,
             np.array2string(kk[pol].max(axis=0), precision=1), np.array2string(kk[~pol].max(axis=0), precision=1) if (~pol).any() else "-"))

    return q, x, lam, xo, lamo, fo, out["fval"].cpu().numpy(), pol, ex, kk

def worst(model, N, B, k=6):
    q, x, lam, xo, lamo, fo, fg, pol, ex, kk = run(model, N, B)
    idx = np.argsort(-np.where(pol, ex, 0))[:k]
    for b in idx:
        ko = orc.qp_kkt(q["H"][b].T, q["g"][b], q["A"][b].T, q["lb"][b], q["ub"][b], q["lbA"][b], q["ubA"][b], xo[b], lamo[b])[1]
        print("inst %d: x err %.1e  f gpu-oracle %.3e (|f| %.2e)  kkt gpu %s  kkt oracle %s" % (b, ex[b], fg[b] - fo[b], abs(fo[b]), np.array2string(kk[b], precision=1), np.array2string(np.asarray(ko), precision=1)))

if __name__ == "__main__":
    if len(sys.argv) > 1: worst(0, 40, 1024); worst(1, 40, 128); sys.exit(0)
    run(0, 40, 1024); run(1, 40, 128)
