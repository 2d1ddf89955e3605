#!/usr/bin/env python3
"""Generate tests/golden/*.npz: small LTV-MPC instances (inputs), the QP the construction path builds for
them and the certified solution.  PARITY UNPINNED: the reference has no golden vectors and cannot run here
(MATLAB + Windows MEX), so these are outputs of the oracle (oracle/) recorded once its identities, KKT
certificate and the scipy cross-check (tests/test_oracle_cpu.py) pass; they pin the oracle and the HIP
path against regressions and against each other."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    for tname in ("fsg2019", "fss2019"):
        tr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", tname + ".json"))
        # N in {5, 10, 20, 40} kinematic, {5, 40} dynamic (+ 10); the N = 40 sizes (the BASELINE shapes
        # nV = 81 / 84) on fsg2019 only and with two instances, to keep the fixtures small
        for model, mname, Ns in ((orc.KINEMATIC, "kin", (5, 10, 20, 40)), (orc.DYNAMIC, "dyn", (5, 10, 40))):
            for N in Ns:
                if N == 40 and tname != "fsg2019":
                    continue
                ids = np.arange(2 if N == 40 else 3) + 1000 * N
                x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, tr.L, 20190, ids)
                q = orc.build_qp_batch(model, tr, N, 0.05, x0, xr, xl, ul, keep_prediction=True)
                o = orc.default_opts(tol_x=1e-9)
                sol = orc.qp_solve_batch_aux(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"], o)
                x, f, fl, it, lam, pol = sol["x"], sol["fval"], sol["exitflag"], sol["iter"], sol["lam"], sol["polished"]
                kkt = np.array([orc.qp_kkt(q["H"][b].T, q["g"][b], q["A"][b].T, q["lb"][b], q["ub"][b], q["lbA"][b], q["ubA"][b], x[b], lam[b])[0]
                                for b in range(len(ids))])
                assert (fl == 0).all() and kkt.max() < 1e-7, (tname, mname, N, fl, kkt)
                steps = [orc.ltv_step(model, tr, N, 0.05, x0[b], xr[b].T, xl[b].T, ul[b].T, o) for b in range(len(ids))]
                np.savez_compressed(os.path.join(OUT, "%s_%s_N%d.npz" % (tname, mname, N)),
                                    model=model, N=N, dt=0.05, ids=ids, x0=x0, x_lin=xl, u_lin=ul, x_ref=xr,
                                    H=q["H"], g=q["g"], A=q["A"], lb=q["lb"], ub=q["ub"], lbA=q["lbA"], ubA=q["ubA"],
                                    const=q["const"], x=x, fval=f, exitflag=fl, kkt=kkt, lam=lam, on_vertex=pol,
                                    u_opt=np.array([s[0] for s in steps]), x_opt=np.array([s[1] for s in steps]),
                                    slack=np.array([s[2] for s in steps]), fval_step=np.array([s[3] for s in steps]))
                print(tname, mname, N, "iters", it, "kkt %.1e" % kkt.max(), "on vertex", pol)


if __name__ == "__main__":
    main()
