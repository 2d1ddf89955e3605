#!/usr/bin/env python3
"""Closed-loop runs: what ARE the QPs that end with a non-zero exit flag?  Runs the device-resident loop (dynamic model by default),
keeps the dense QP (H, g, A, bounds) of up to --keep failing instances per exit flag, and asks an independent code base about each:
  * feasibility of {lb <= x <= ub, lbA <= A x <= ubA} by scipy.optimize.linprog (HiGHS dual simplex) -- qpOASES answers -2 exactly
    when this set is empty;
  * for the feasible ones, the optimum by scipy trust-constr, compared with what the kernel returned (last iterate).
Prints one JSON line.  The feasibility LPs take seconds; the independent QP solves take minutes each on these ill-conditioned
problems (1e8 slack cost), so they are a separate stage: `--save f.npz` stores the kept QPs, `--offline f.npz` solves them (CPU only).
usage: tools/cl_failures_check.py [--model dynamic] [--cars 512] [--steps 80] [--keep 24] [--save f.npz] | --offline f.npz"""
import argparse, ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm
from fsae_mpc_amd._lib import lib, check


def unpack(e):
    A = e["A"].T                                                    # stored column-major nC x nV
    lbA = np.where(e["lbA"] < -1e9, -np.inf, e["lbA"]); ubA = np.where(e["ubA"] > 1e9, np.inf, e["ubA"])
    lb = np.where(e["lb"] < -1e9, -np.inf, e["lb"]); ub = np.where(e["ub"] > 1e9, np.inf, e["ub"])
    return A, lb, ub, lbA, ubA


def offline(a):
    """independent optimum (scipy trust-constr) of the stored FEASIBLE failing QPs vs the iterate the kernel returned"""
    from scipy.optimize import Bounds, LinearConstraint, minimize
    z = np.load(a.offline)
    keys = sorted({k.rsplit("_", 1)[0] for k in z.files if k.endswith("_H")})
    out = {}
    for key in keys:
        e = {k_: z[key + "_" + k_] for k_ in ("H", "g", "A", "lb", "ub", "lbA", "ubA", "z", "infeasible", "x_feas")}
        flag = key.split("_")[0][1:]
        rec = out.setdefault(flag, {"solved_independently": 0, "kernel_iterate_within_1e-3": 0, "worst_rel_distance": 0.0, "solver_gave_up": 0})
        if bool(e["infeasible"]) or rec["solved_independently"] + rec["solver_gave_up"] >= a.max_solves:
            continue
        A, lb, ub, lbA, ubA = unpack(e)
        H = e["H"].T
        print("flag %s: independent QP solve of %s ..." % (flag, key), file=sys.stderr, flush=True)
        try:
            sol = minimize(lambda v: 0.5 * v @ H @ v + e["g"] @ v, e["x_feas"], jac=lambda v: H @ v + e["g"], hess=lambda v: H, method="trust-constr",
                           bounds=Bounds(lb, ub), constraints=[LinearConstraint(A, lbA, ubA)], options=dict(gtol=1e-8, xtol=1e-10, barrier_tol=1e-10, maxiter=1500))
        except Exception:
            rec["solver_gave_up"] += 1
            continue
        nz = len(e["z"])
        d = float(np.max(np.abs(sol.x[:nz] - e["z"])) / max(1.0, np.max(np.abs(sol.x))))
        rec["solved_independently"] += 1; rec["kernel_iterate_within_1e-3"] += int(d <= 1e-3); rec["worst_rel_distance"] = max(rec["worst_rel_distance"], d)
    print(json.dumps({"independent_optimum_of_feasible_failing_QPs": out}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dynamic", choices=["kinematic", "dynamic"])
    ap.add_argument("--cars", type=int, default=512)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--keep", type=int, default=24)
    ap.add_argument("--seed", type=int, default=20190)
    ap.add_argument("--save", default=None)
    ap.add_argument("--offline", default=None)
    ap.add_argument("--max-solves", type=int, default=3)
    a = ap.parse_args()
    if a.offline:
        return offline(a)
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
    from scipy.optimize import linprog
    res = {}
    feas = {}
    for f, lst in sorted(kept.items()):
        n_inf = 0
        for e in lst:
            A, lb, ub, lbA, ubA = unpack(e)
            rows = [A[np.isfinite(ubA)], -A[np.isfinite(lbA)]]; rhs = [ubA[np.isfinite(ubA)], -lbA[np.isfinite(lbA)]]
            lp = linprog(np.zeros(A.shape[1]), A_ub=np.vstack(rows), b_ub=np.concatenate(rhs), bounds=list(zip(lb, ub)), method="highs-ds")
            e["infeasible"] = lp.status == 2
            e["x_feas"] = lp.x if lp.status == 0 else np.zeros(A.shape[1])
            n_inf += lp.status == 2
        res[str(f)] = {"examined": len(lst), "infeasible_by_HiGHS": int(n_inf), "feasible": len(lst) - int(n_inf)}
    if a.save:
        flat = {}
        for f, lst in kept.items():
            for i, e in enumerate(lst):
                for k_, v in e.items():
                    flat["f%d_%d_%s" % (f, i, k_)] = np.asarray(v)
        np.savez_compressed(a.save, **flat)
    print(json.dumps({"workload": "closed loop, %s N=40, %d cars x %d steps on fss2019, seed %d" % (a.model, a.cars, a.steps, a.seed),
                      "exitflag_tally_driving_cars": tally, "independent_check_of_failing_QPs": res}))


if __name__ == "__main__":
    main()
