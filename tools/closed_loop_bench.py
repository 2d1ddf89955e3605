#!/usr/bin/env python3
"""BASELINE configs[3]: closed-loop Monte-Carlo -- B cars on fss2019, every receding-horizon step is
one batch of B LTV-MPC QPs (frame transform + reference -> linearise/condense/solve -> PID + plant), all on the device.
Prints one JSON line (QP solves/s over the whole run, flag histogram, iterations, lap progress).
usage: tools/closed_loop_bench.py [--model dynamic|kinematic] [--batch 2048] [--steps 200] [--horizon 40]"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fsae_mpc_amd as fm


def initial_carts(tr, B, seed):
    """s0 = u*L, lateral +-0.5 m, heading +-0.1 rad, speed U[0,15]; numpy PCG64(seed)."""
    rng = np.random.default_rng(seed)
    s = rng.uniform(0, tr.L, B); n = rng.uniform(-0.5, 0.5, B); dth = rng.uniform(-0.1, 0.1, B); v = rng.uniform(0, 15, B)
    def ev(P, t):   # the Bezier table on the host (same formulas as interpolate_spline / interpolate_spline_d); P is M x 4
        r = np.mod(t, tr.dl * tr.M); i = np.minimum(np.floor(r / tr.dl).astype(int), tr.M - 1); u = r / tr.dl - i; w = 1 - u
        val = P[i, 0] * w ** 3 + 3 * P[i, 1] * w * w * u + 3 * P[i, 2] * w * u * u + P[i, 3] * u ** 3
        d = (-3 * w * w * P[i, 0] + 3 * (3 * u * u - 4 * u + 1) * P[i, 1] + 3 * (2 * u - 3 * u * u) * P[i, 2] + 3 * u * u * P[i, 3]) / tr.dl
        return val, d
    x, xd = ev(tr.xP, s); y, yd = ev(tr.yP, s)
    nrm = np.hypot(xd, yd)
    cart = np.zeros((B, 7))
    cart[:, 0] = x - yd / nrm * n; cart[:, 1] = y + xd / nrm * n; cart[:, 2] = np.arctan2(yd, xd) + dth; cart[:, 3] = v
    return cart, s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dynamic", choices=["kinematic", "dynamic"])
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--seed", type=int, default=20190)
    ap.add_argument("--max-iter", type=int, default=100, help="interior-point iteration limit (bounds the batch tail)")
    a = ap.parse_args()
    model = fm.KINEMATIC if a.model == "kinematic" else fm.DYNAMIC
    tr = fm.Track.load("fss2019")
    cart0, s_init = initial_carts(tr, a.batch, a.seed)
    cl = fm.ClosedLoop(model, a.horizon, 0.05, tr, cart0, options=fm.default_opts(max_iter=a.max_iter))
    cl.x_opt[:, :, 0] += torch.from_numpy(s_init).cuda()[:, None]        # start the closest-point search near the car
    cl.x_opt[:, :, 3] += torch.from_numpy(cart0[:, 3]).cuda()[:, None]   # and the first linearisation at its speed
    flags_hist = {}
    iters = 0.0; solved = 0
    cl.step(); torch.cuda.synchronize()                                  # warm-up step (allocations, code load)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = cl.step()
        fl = out["exitflag"]
        solved += int((fl == 0).sum().item())
        iters += float(out["iter"].double().sum().item())
        for k_, c_ in zip(*np.unique(fl.cpu().numpy(), return_counts=True)):
            flags_hist[int(k_)] = flags_hist.get(int(k_), 0) + int(c_)
    torch.cuda.synchronize()
    dt_wall = time.perf_counter() - t0
    x0 = cl.x0.cpu().numpy()
    res = {"metric": "QP solves/sec (closed loop, %s N=%d, fp64)" % (a.model, a.horizon), "value": solved / dt_wall, "unit": "QP solves/s",
           "n_gpus": 1, "steps": a.steps, "ms_per_step": 1e3 * dt_wall / a.steps, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE configs[3] share of one GPU: %d cars on fss2019, %d receding-horizon steps, every step one batch of QPs "
                                  "(frame transform + reference + linearise/condense/solve + PID/plant on the device; host loop with per-step flag readback)" % (a.batch, a.steps),
                      "exitflag_histogram": flags_hist, "mean_ipm_iterations": iters / (a.batch * a.steps),
                      "cars_lap_finished": int((cl.finished == 1).sum().item()), "cars_lost": int((cl.finished == 2).sum().item()), "mean_speed_end": float(cl.cart[:, 3].mean().item()),
                      "median_abs_lateral_offset_end": float(np.nanmedian(np.abs(x0[:, 1]))), "seed": a.seed, "max_iter": a.max_iter}}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
