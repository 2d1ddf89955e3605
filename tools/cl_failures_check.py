#!/usr/bin/env python3
"""Closed-loop runs: what ARE the QPs that end with a non-zero exit flag?  Runs the device-resident loop (dynamic model by default),
keeps the dense QP (H, g, A, bounds) of up to --keep failing instances per exit flag, and asks an independent code base about each:
  * feasibility of {lb <= x <= ub, lbA <= A x <= ubA} by scipy.optimize.linprog (HiGHS dual simplex) -- qpOASES answers -2 exactly
    when this set is empty;
  * for the feasible ones, the optimum by scipy trust-constr, compared with what the kernel returned (last iterate).
Prints one JSON line.  usage: tools/cl_failures_check.py [--model dynamic] [--cars 512] [--steps 80] [--keep 24]"""
import argparse, ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm
from fsae_mpc_amd._lib import lib, check


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dynamic", choices=["kinematic", "dynamic"])
    ap.add_argument("--cars", type=int, default=512)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--keep", type=int, default=24)
    ap.add_argument("--seed", type=int, default=20190)
    a = ap.parse_args()
    model = fm.KINEMATIC if a.model == "kinematic" else fm.DYNAMIC
    N = 40
    tr = fm.Track.load("fss2019")
    cart0, s_init = fm.monte_carlo_carts(tr, a.cars, a.seed)
    cl = fm.ClosedLoop(model, N, 0.05, tr, cart0)
    cl.x_opt[:, :, 0] += torch.from_numpy(s_init).to(cl.device)[:, None]
    cl.x_opt[:, :, 3] += torch.from_numpy(cart0[:, 3]).to(cl.device)[:, None]
    kept = {}
    tally = {}
    P = lambda t: C.c_void_p(t.data_ptr())
    for t in range(a.steps):
        cl.pre()
        q = cl.mpc.build_qp(cl.x0, cl.x_ref, cl.x_opt, cl.u_opt)           # the QP the fused step below builds and solves
        out = cl.mpc.step(cl.x0, cl.x_ref, cl.x_opt, cl.u_opt)
        fl = out["exitflag"].cpu().numpy(); drv = (cl.finished == 0).cpu().numpy()
        for f in np.unique(fl[drv]):
            tally[int(f)] = tally.get(int(f), 0) + int(((fl == f) & drv).sum())
        for b in np.where((fl != 0) & drv)[0]:
            lst = kept.setdefault(int(fl[b]), [])
            if len(lst) < a.keep:
                lst.append(dict(step=t, car=int(b), z=torch.cat([out["u_opt"][b], out["slack"][b]]).cpu().numpy(),
                                **{k: q[k][b].cpu().numpy().copy() for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")}))
        check(lib().fsaempc_cl_accept_batch_device(cl.model, cl.N, cl.B, P(out["x_opt"]), P(out["u_opt"]), P(out["exitflag"]), P(cl.x_opt), P(cl.u_opt),
                                                   cl._stream(None)), "accept")
        cl.plant(None)
    torch.cuda.synchronize()
    from scipy.optimize import Bounds, LinearConstraint, linprog, minimize
    res = {}
    for f, lst in sorted(kept.items()):
        n_inf = n_feas = n_match = n_err = 0; worst = 0.0
        for e in lst:
            A = e["A"].T                                                    # stored column-major nC x nV
            lbA = np.where(e["lbA"] < -1e9, -np.inf, e["lbA"]); ubA = np.where(e["ubA"] > 1e9, np.inf, e["ubA"])
            lb = np.where(e["lb"] < -1e9, -np.inf, e["lb"]); ub = np.where(e["ub"] > 1e9, np.inf, e["ub"])
            rows = [A[np.isfinite(ubA)], -A[np.isfinite(lbA)]]; rhs = [ubA[np.isfinite(ubA)], -lbA[np.isfinite(lbA)]]
            lp = linprog(np.zeros(A.shape[1]), A_ub=np.vstack(rows), b_ub=np.concatenate(rhs), bounds=list(zip(lb, ub)), method="highs-ds")
            if lp.status == 2:
                n_inf += 1
                continue
            n_feas += 1
            if n_feas > 4:                                                   # the independent QP solve takes ~a minute per instance
                continue
            H = e["H"].T
            print("flag %d: independent QP solve %d ..." % (f, n_feas), file=sys.stderr, flush=True)
            try:
                    sol = minimize(lambda z: 0.5 * z @ H @ z + e["g"] @ z, lp.x, jac=lambda z: H @ z + e["g"], hess=lambda z: H, method="trust-constr",
                               bounds=Bounds(lb, ub), constraints=[LinearConstraint(A, lbA, ubA)], options=dict(gtol=1e-9, xtol=1e-11, barrier_tol=1e-11, maxiter=1500))
            except Exception as ex:                                          # scipy's own numerics gave up on this instance
                n_err += 1
                continue
            nz = len(e["z"])
            d = float(np.max(np.abs(sol.x[:nz] - e["z"])) / max(1.0, np.max(np.abs(sol.x))))
            worst = max(worst, d); n_match += d <= 1e-3
        res[str(f)] = {"examined": len(lst), "infeasible_by_HiGHS": n_inf, "feasible": n_feas,
                       "feasible_solved_independently": min(n_feas, 4) - n_err, "independent_solver_gave_up": n_err, "of_those_kernel_iterate_within_1e-3_of_independent_optimum": int(n_match), "worst_rel_distance_to_independent_optimum": worst}
    print(json.dumps({"workload": "closed loop, %s N=40, %d cars x %d steps on fss2019, seed %d" % (a.model, a.cars, a.steps, a.seed),
                      "exitflag_tally_driving_cars": tally, "independent_check_of_failing_QPs": res}))


if __name__ == "__main__":
    main()
