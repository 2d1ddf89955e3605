#!/usr/bin/env python3
"""Closed-loop Monte-Carlo on the CPU ORACLE (test infrastructure): B cars on fss2019, T receding-horizon steps each
(main.m:91-179 restated by oracle/ltv_oracle_plant.c), collecting every QP whose solve returns a non-zero exit flag.
Used to study the failure modes of the interior-point method outside the GPU budget.
usage: tests/harness/cl_oracle_mc.py [--model dynamic] [--cars 64] [--steps 60] [--out build/cl_fail.npz]"""
import argparse, os, sys, time
import multiprocessing as mp
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)


def initial_carts(tr, B, seed):
    rng = np.random.default_rng(seed)
    s = rng.uniform(0, tr.L, B); n = rng.uniform(-0.5, 0.5, B); dth = rng.uniform(-0.1, 0.1, B); v = rng.uniform(0, 15, B)
    xP, yP = np.asarray(tr.xP), np.asarray(tr.yP)
    def ev(P, t):
        r = np.mod(t, tr.dl * tr.M); i = np.minimum(np.floor(r / tr.dl).astype(int), tr.M - 1); u = r / tr.dl - i; w = 1 - u
        val = P[i, 0] * w ** 3 + 3 * P[i, 1] * w * w * u + 3 * P[i, 2] * w * u * u + P[i, 3] * u ** 3
        d = (-3 * w * w * P[i, 0] + 3 * (3 * u * u - 4 * u + 1) * P[i, 1] + 3 * (2 * u - 3 * u * u) * P[i, 2] + 3 * u * u * P[i, 3]) / tr.dl
        return val, d
    x, xd = ev(xP, s); y, yd = ev(yP, s)
    nrm = np.hypot(xd, yd)
    cart = np.zeros((B, 7))
    cart[:, 0] = x - yd / nrm * n; cart[:, 1] = y + xd / nrm * n; cart[:, 2] = np.arctan2(yd, xd) + dth; cart[:, 3] = v
    return cart, s


def run_car(args):
    model, N, T, cart0, s0, car, polish = args
    import oracle as orc
    tr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fss2019.json"))
    nx = orc.dims(model, N)[0]
    dt = 0.05
    k = np.arange(1, N + 1) * dt
    x_opt = np.zeros((nx, N), order="F"); u_opt = np.zeros((2, N), order="F")
    x_opt[0] = 10 * k ** 2 / 2 + s0; x_opt[3] = 10 * k + cart0[3]; u_opt[0] = 10
    cart = cart0.copy(); pid = np.zeros(4)
    fails, flags, iters = [], [], []
    o = orc.default_opts(polish=polish)
    for t in range(T):
        x0, x_ref, fin = orc.cl_pre(model, N, dt, tr, cart, x_opt[0, 0])
        if fin:
            break
        q = orc.build_qp(model, tr, N, dt, x0, x_ref, x_opt, u_opt)
        z, fv, fl, it, lam = orc.qp_solve(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"], o)
        flags.append(fl); iters.append(it)
        if fl != 0:
            fails.append(dict(car=car, step=t, flag=fl, **{k_: q[k_].copy() for k_ in ("H", "g", "A", "lb", "ub", "lbA", "ubA")}))
        if fl in (0, 1) and np.isfinite(z).all():   # plan taken over (fsaempc_cl_accept_batch_device); else the car drives on its last good plan
            pred = q["A_bar"] @ x0 + q["Bt"] @ z + q["d_bar"]
            x_opt = np.asfortranarray(pred.reshape(N, nx).T); u_opt = np.asfortranarray(z[:2 * N].reshape(N, 2).T)
        if not np.isfinite(cart).all() or np.abs(cart).max() > 1e6:
            break
        cart, pid, _ = orc.plant_step(cart, pid, x_opt[3, 0], x_opt[nx - 1, 0], dt)
    return car, flags, iters, fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dynamic")
    ap.add_argument("--cars", type=int, default=64)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--seed", type=int, default=20190)
    ap.add_argument("--polish", type=int, default=1)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "build", "cl_fail.npz"))
    a = ap.parse_args()
    import oracle as orc
    model = orc.DYNAMIC if a.model == "dynamic" else orc.KINEMATIC
    tr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fss2019.json"))
    cart0, s_init = initial_carts(tr, a.cars, a.seed)
    t0 = time.time()
    with mp.Pool(a.procs) as pool:
        res = pool.map(run_car, [(model, a.horizon, a.steps, cart0[c], s_init[c], c, a.polish) for c in range(a.cars)], chunksize=1)
    hist, its, fails = {}, [], []
    for car, flags, iters, fl in res:
        for f in flags:
            hist[f] = hist.get(f, 0) + 1
        its += iters; fails += fl
    print("cars %d steps %d: flags %s mean iters %.1f, %.0f s" % (a.cars, a.steps, hist, np.mean(its), time.time() - t0))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    keep = fails[:400]
    np.savez_compressed(a.out, car=[f["car"] for f in keep], step=[f["step"] for f in keep], flag=[f["flag"] for f in keep],
                        **{k_: np.stack([f[k_] for f in keep]) if keep else np.zeros(0) for k_ in ("H", "g", "A", "lb", "ub", "lbA", "ubA")})


if __name__ == "__main__":
    main()
