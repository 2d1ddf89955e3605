#!/usr/bin/env python3
"""BASELINE configs[3]: closed-loop Monte-Carlo -- B cars on fss2019, every receding-horizon step is
one batch of B LTV-MPC QPs (frame transform + reference -> linearise/condense/solve -> PID + plant), all on the device with
no per-step read-back (fsae_mpc_amd.monte_carlo).  Prints one JSON line: QP solves/s over the QPs of the cars still
driving, the exit-flag tally the way main.m:209,222 reports it ("abnormal exits %"), iterations, progress.
usage: tools/closed_loop_bench.py [--model dynamic|kinematic] [--batch 2048] [--steps 200] [--horizon 40]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dynamic", choices=["kinematic", "dynamic"])
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--seed", type=int, default=20190)
    ap.add_argument("--max-iter", type=int, default=100, help="interior-point iteration limit (bounds the batch tail)")
    ap.add_argument("--no-launch-hint", action="store_true", help="do not hand the previous iteration counts to the solve as its launch-order estimate (A/B)")
    ap.add_argument("--warm", action="store_true", help="start every solve from the previous plan shifted by one stage (ClosedLoop(warm_start=True))")
    a = ap.parse_args()
    model = fm.KINEMATIC if a.model == "kinematic" else fm.DYNAMIC
    tr = fm.Track.load("fss2019")
    fm.monte_carlo(model, a.horizon, tr, min(a.batch, 64), 2, a.seed, warm_start=a.warm)          # warm-up (allocations, code load)
    t0 = time.perf_counter()
    cl, fl, it, ac = fm.monte_carlo(model, a.horizon, tr, a.batch, a.steps, a.seed, options=fm.default_opts(max_iter=a.max_iter), warm_start=a.warm, launch_hint=not a.no_launch_hint)
    dt_wall = time.perf_counter() - t0
    n_act = int(ac.sum())
    hist = {int(k_): int(c_) for k_, c_ in zip(*np.unique(fl[ac], return_counts=True))}
    solved = int(((fl == 0) & ac).sum())
    bad_data = int(((fl == -1) & (it == 0) & ac).sum())    # -1 before the first iteration = non-finite QP data (vehicle state outside
                                                           # the model's domain, e.g. v_x -> 0 in the dynamic model): the reference's
                                                           # MEX gateway rejects such a call ("Argument contains NaN")
    x0 = cl.x0.cpu().numpy()
    fin = cl.finished.cpu().numpy()
    lost_qps = int((~ac[:, fin == 2]).sum())   # steps the lost cars sat out
    print(json.dumps({
        "metric": "QP solves/sec (closed loop, %s N=%d, fp64)" % (a.model, a.horizon), "value": solved / dt_wall, "unit": "QP solves/s",
        "n_gpus": 1, "steps": a.steps, "ms_per_step": 1e3 * dt_wall / a.steps, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] share of one GPU: %d cars on fss2019, %d receding-horizon steps, every step one batch of QPs "
                               "(frame transform + reference + linearise/condense/solve + PID/plant, device-resident loop, no per-step read-back; "
                               "wall time includes the allocation of the run)" % (a.batch, a.steps),
                   "qps_of_driving_cars": n_act, "qps_total_launched": int(a.batch * a.steps),
                   "exitflag_histogram_driving_cars": hist, "minus1_with_nonfinite_qp_data": bad_data,
                   "minus1_on_finite_qp_data_pct": 100.0 * (hist.get(-1, 0) - bad_data) / max(1, n_act), "abnormal_exit_pct": 100.0 * (1.0 - solved / max(1, n_act)),
                   # the same tally with the QPs a lost car (|n| >= 3 m, |v| >= 100 m/s or |state| >= 1e6: `finished` = 2, plant.hip) would still
                   # have launched counted as abnormal: lost cars leave the denominator above, round 1 had no such rule
                   "abnormal_exit_pct_lost_cars_counted": 100.0 * (1.0 - solved / max(1, n_act + lost_qps)), "qps_not_launched_for_lost_cars": lost_qps,
                   "mean_ipm_iterations": float(it[ac].mean()) if n_act else 0.0,
                   "cars_past_end_of_track_parameter": int((cl.finished == 1).sum().item()), "cars_lost": int((cl.finished == 2).sum().item()),
                   "mean_speed_end": float(cl.cart[:, 3].mean().item()),
                   "median_abs_lateral_offset_end": float(np.nanmedian(np.abs(x0[:, 1]))), "seed": a.seed, "max_iter": a.max_iter, "warm_start": bool(a.warm), "launch_hint": not a.no_launch_hint}}))


if __name__ == "__main__":
    main()
