#!/usr/bin/env python3
"""Independent-solver check of the oracle at the dynamic BASELINE sizes (run once, slow: scipy trust-constr takes 10-40 minutes per
instance there; the kinematic headline size is in the CPU suite, tests/test_oracle_cpu.py).
usage: scipy_crosscheck.py out.json [model,N,instance ...]      e.g.  profiles/round3/scipy_crosscheck.json 1,40,1 1,60,0"""
import json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc
from scipy.optimize import Bounds, LinearConstraint, minimize

out_path = sys.argv[1]
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(1, 40, 1)]
otr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
res = []
for model, N, inst in cases:
    x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, otr.L, 20190, [inst])
    q = orc.build_qp_batch(model, otr, N, 0.05, x0, xr, xl, ul)
    H, g, A = q["H"][0].T, q["g"][0], q["A"][0].T
    lb, ub, lbA, ubA = q["lb"][0], q["ub"][0], q["lbA"][0], q["ubA"][0]
    xo, fo, fl, it, lam = orc.qp_solve(H, g, A, lb, ub, lbA, ubA)
    big = 1e9
    t0 = time.time()
    r = minimize(lambda x: 0.5 * x @ H @ x + g @ x, np.clip(np.zeros_like(g), lb, ub), jac=lambda x: H @ x + g, hess=lambda x: H, method="trust-constr",
                 bounds=Bounds(np.where(lb > -big, lb, -np.inf), np.where(ub < big, ub, np.inf)),
                 constraints=[LinearConstraint(A, np.where(lbA > -big, lbA, -np.inf), np.where(ubA < big, ubA, np.inf))],
                 options=dict(gtol=1e-12, xtol=1e-14, barrier_tol=1e-14, maxiter=20000))
    dt = time.time() - t0
    rec = dict(model="dynamic" if model else "kinematic", N=N, instance=inst, nV=int(g.size), nC=int(lbA.size), oracle_flag=int(fl), oracle_iter=int(it),
               scipy_status=int(r.status), scipy_iterations=int(r.nit), scipy_seconds=dt, fval_oracle=float(fo), fval_scipy=float(r.fun),
               fval_rel_diff=float(abs(r.fun - fo) / max(1.0, abs(fo))), x_rel_diff=float(np.abs(r.x - xo).max() / max(1.0, np.abs(xo).max())),
               scipy_constr_violation=float(r.constr_violation))
    print(json.dumps(rec), flush=True)
    res.append(rec)
    json.dump(dict(what="oracle (oracle/ltv_oracle_qp.c) vs scipy.optimize trust-constr on LTV-MPC QPs at BASELINE sizes", cases=res), open(out_path, "w"), indent=1)
