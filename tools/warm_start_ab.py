#!/usr/bin/env python3
"""Warm start of the interior-point solve, measured: on the QPs of a closed loop -- consecutive MPC
periods of the same cars, the case qpOASES_sequence's hot start is made for -- every QP is solved twice, cold (x = clamp(0, lb, ub))
and started from the previous period's solution shifted by one stage (inputs u_1..u_{N-1}, last stage repeated, slacks kept),
handed in through fsaempc_qp_aux.x_init.  The loop itself advances on the cold solution.  Reported per model: mean / median
interior-point iterations, exit flags, solve time by HIP events, over the cars still driving.
usage: warm_start_ab.py [cars] [steps] > profiles/round3/warm_start_ab.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import fsae_mpc_amd as fm  # noqa: E402
from fsae_mpc_amd.closed_loop import ClosedLoop, monte_carlo_carts  # noqa: E402


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1)


def run(model, N, B, steps, track):
    cl = ClosedLoop(model, N, 0.05, track, monte_carlo_carts(track, B, 20190)[0])
    nV = cl.mpc.nV
    prev = None
    rec = dict(cold_it=[], warm_it=[], cold_ms=[], warm_ms=[], cold_ok=0, warm_ok=0, n=0, dx=[])
    ws = None
    for s in range(steps):
        cl.pre()
        q = cl.mpc.build_qp(cl.x0, cl.x_ref, cl.x_opt, cl.u_opt)
        args = [q[k] for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA")]
        cold, t_c = timed(lambda: fm.qp_solve_batch_device(*args, workspace=ws))
        ws = cold["workspace"]
        driving = (cl.finished == 0)
        if prev is not None:
            xw = prev.clone()
            xw[:, : 2 * (N - 1)] = prev[:, 2: 2 * N]          # inputs of stage k+1 become stage k; the last stage stays
            warm, t_w = timed(lambda: fm.qp_solve_batch_device(*args, workspace=ws, x_init=xw.contiguous()))
            ok = driving & (cold["exitflag"] == 0) & (warm["exitflag"] == 0)
            rec["cold_it"].append(cold["iter"][ok].double().cpu().numpy()); rec["warm_it"].append(warm["iter"][ok].double().cpu().numpy())
            rec["cold_ms"].append(t_c); rec["warm_ms"].append(t_w)
            rec["cold_ok"] += int((driving & (cold["exitflag"] == 0)).sum()); rec["warm_ok"] += int((driving & (warm["exitflag"] == 0)).sum())
            rec["n"] += int(driving.sum())
            d = (warm["x"][ok] - cold["x"][ok]).abs().amax(1) / cold["x"][ok].abs().amax(1).clamp(min=1.0)
            rec["dx"].append(d.cpu().numpy())
        good = (cold["exitflag"] == 0).view(-1, 1)
        prev = torch.where(good, cold["x"], prev if prev is not None else cold["x"])
        cl.step()
    ci, wi = np.concatenate(rec["cold_it"]), np.concatenate(rec["warm_it"])
    dx = np.concatenate(rec["dx"])
    return dict(model="dynamic" if model == fm.DYNAMIC else "kinematic", N=N, cars=B, steps=steps, qps_compared=int(ci.size),
                cold_iter_mean=float(ci.mean()), warm_iter_mean=float(wi.mean()), cold_iter_median=float(np.median(ci)), warm_iter_median=float(np.median(wi)),
                iter_change_pct=float(100.0 * (wi.mean() - ci.mean()) / ci.mean()),
                cold_solved_of_driving=[rec["cold_ok"], rec["n"]], warm_solved_of_driving=[rec["warm_ok"], rec["n"]],
                cold_ms_per_batch=float(np.mean(rec["cold_ms"])), warm_ms_per_batch=float(np.mean(rec["warm_ms"])),
                same_point_max_rel=float(dx.max()), same_point_p99_rel=float(np.quantile(dx, 0.99)))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    track = fm.Track.load("fss2019")
    out = dict(what="interior-point solve started from the previous MPC period's solution shifted by one stage (fsaempc_qp_aux.x_init) vs cold start, "
                    "closed-loop QPs (monte-carlo cars of BASELINE configs[3]); slacks / multipliers start as always",
               runs=[run(fm.KINEMATIC, 40, B, steps, track), run(fm.DYNAMIC, 40, B, steps, track)])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
