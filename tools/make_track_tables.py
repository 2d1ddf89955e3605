#!/usr/bin/env python3
"""Generate arc-length spline tables (M x 4 Bezier control points per axis) from the
reference's race-line CSVs.  Runs ONLY in the build container (reads /root/reference/data);
its outputs under fsae-mpc_amd/tracks/ are data fixtures that travel to the GPU box.

Restates (setup path):
  util/read_raceline_csv.m:6-19      columns X,Y,...
  spline/make_spline_periodic.m:9-33 cyclic [1 4 1] system for P1, back-substitution for P2
  spline/arclength_reparam.m:15-64   segment lengths (with the x_P(i,1)-for-x_P(i,2) quirk, :20-23),
                                     bisection to 0.01 m (:49, :68-97), refit on M points
MATLAB's adaptive `integral` is replaced by scipy.integrate.quad (epsabs 1e-10, epsrel 1e-6 =
MATLAB defaults); the 0.01 m bisection tolerance makes the midpoint sequence insensitive to that.
"""
import json
import os
import sys

import numpy as np
from scipy.integrate import quad

REF = "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fsae-mpc_amd", "tracks")


def make_spline_periodic(P):
    P = np.asarray(P, dtype=np.float64)
    N = len(P)
    A = np.zeros((N, N))
    for i in range(N):
        A[i, i] = 4.0
        A[i, (i - 1) % N] = 1.0
        A[i, (i + 1) % N] = 1.0
    b = np.zeros(N)
    b[: N - 1] = 4 * P[: N - 1] + 2 * P[1:N]
    b[N - 1] = 4 * P[N - 1] + 2 * P[0]
    P1 = np.linalg.solve(A, b)
    P2 = np.zeros(N)
    P2[: N - 1] = 2 * P[1:N] - P1[1:N]
    P2[N - 1] = 2 * P[0] - P1[0]
    P3 = np.concatenate([P[1:], P[:1]])
    return np.stack([P, P1, P2, P3], axis=1)


def interpolate_spline(t, P, dl):
    M = len(P)
    t = np.mod(t, dl * M)
    i = int(np.floor(t / dl))
    s = t / dl - i
    return P[i, 0] * (1 - s) ** 3 + 3 * P[i, 1] * (1 - s) ** 2 * s + 3 * P[i, 2] * (1 - s) * s**2 + P[i, 3] * s**3


def _speed(xP, yP, i):
    # arclength_reparam.m:20-23 -- note column 1 used where column 2 is expected (quirk C-6)
    def xd(t):
        return -3 * (1 - t) ** 2 * xP[i, 0] + 3 * (3 * t**2 - 4 * t + 1) * xP[i, 0] + 3 * (2 * t - 3 * t**2) * xP[i, 2] + 3 * t**2 * xP[i, 3]

    def yd(t):
        return -3 * (1 - t) ** 2 * yP[i, 0] + 3 * (3 * t**2 - 4 * t + 1) * yP[i, 0] + 3 * (2 * t - 3 * t**2) * yP[i, 2] + 3 * t**2 * yP[i, 3]

    return lambda t: np.sqrt(xd(t) ** 2 + yd(t) ** 2)


def _integral(f, a, b):
    return quad(f, a, b, epsabs=1e-10, epsrel=1e-6, limit=200)[0]


def arclength_reparam(xP, yP, M):
    N = len(xP)
    l = np.array([_integral(_speed(xP, yP, i), 0.0, 1.0) for i in range(N)])
    l_cum = np.concatenate([[0.0], np.cumsum(l)])
    dl = l_cum[N] / M
    Px = np.zeros(M + 1)
    Py = np.zeros(M + 1)
    Px[0], Py[0] = xP[0, 0], yP[0, 0]
    Px[M], Py[M] = xP[N - 1, 3], yP[N - 1, 3]
    for i in range(1, M):
        j = int(np.argmax(l_cum >= i * dl)) - 1  # find(l_cum >= i*dl, 1) - 1, 0-based segment
        sp = _speed(xP, yP, j)
        f = lambda T: _integral(sp, 0.0, T) + l_cum[j] - i * dl
        xl, xu = 0.0, 1.0
        while True:
            t = (xl + xu) / 2
            fx = f(t)
            if abs(fx) <= 0.01:
                break
            if fx < 0:
                xl = t
            else:
                xu = t
        Px[i] = interpolate_spline(t + j, xP, 1.0)
        Py[i] = interpolate_spline(t + j, yP, 1.0)
    return make_spline_periodic(Px[:M]), make_spline_periodic(Py[:M]), float(dl), float(l_cum[-1])


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in ("fsg2019", "fss2019", "fso2020"):
        path = os.path.join(REF, name + ".csv")
        if not os.path.exists(path):
            print("missing", path, file=sys.stderr)
            continue
        raw = np.genfromtxt(path, delimiter=",", skip_header=1)
        x, y = raw[:, 0], raw[:, 1]
        xs, ys = make_spline_periodic(x), make_spline_periodic(y)
        xP, yP, dl, L = arclength_reparam(xs, ys, 100)  # main.m:14-17
        out = {
            "name": name,
            "source": "derived from kerry-he/fsae-mpc data/%s.csv via main.m:11-17" % name,
            "M": 100,
            "dl": dl,
            "L": L,
            "xP": xP.tolist(),  # M x 4, row = segment
            "yP": yP.tolist(),
        }
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(out, f)
        print(name, "dl=%.6f L=%.4f" % (dl, L))


if __name__ == "__main__":
    main()
