#!/usr/bin/env python3
"""Prototype (numpy) of the active-set refinement that follows the interior-point loop: working set from the
multipliers, equality-constrained solve, then single add/drop corrections until the point is a KKT point.
Measures the vertex rate on the synthetic LTV-MPC families.  Test infrastructure (uses the oracle IPM without polish)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import oracle as orc


def eqp(H, g, G, W, side, l, u):
    idx = np.array(sorted(W), dtype=int)
    n = H.shape[0]; na = len(idx)
    Gw = G[idx]
    b = np.array([l[i] if side[i] > 0 else u[i] for i in idx])
    K = np.zeros((n + na, n + na)); K[:n, :n] = H; K[:n, n:] = -Gw.T; K[n:, :n] = Gw
    rhs = np.concatenate([-g, b])
    try:
        sol = np.linalg.solve(K, rhs)
    except np.linalg.LinAlgError:
        sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
    return sol[:n], idx, sol[n:]


def refine(H, g, A, lb, ub, lbA, ubA, x, lam, maxcorr=8, big=1e9, ftol=1e-9, stol=1e-9):
    n = H.shape[0]
    G = np.vstack([np.eye(n), A]); l = np.concatenate([lb, lbA]); u = np.concatenate([ub, ubA])
    hl = l > -big; hu = u < big
    v = G @ x
    side = np.zeros(len(l), dtype=int)
    W = set()
    for i in range(len(l)):
        if lam[i] > 0 and hl[i] and lam[i] > abs(v[i] - l[i]): W.add(i); side[i] = 1
        elif lam[i] < 0 and hu[i] and -lam[i] > abs(u[i] - v[i]): W.add(i); side[i] = -1
    for k in range(maxcorr + 1):
        if len(W) > n: return None, k, "overfull"
        xs, idx, y = eqp(H, g, G, W, side, l, u)
        vs = G @ xs
        sc = np.maximum(1.0, np.abs(vs))
        vl = np.where(hl, (l - vs) / np.maximum(sc, np.abs(np.where(hl, l, 0))), -1)
        vu = np.where(hu, (vs - u) / np.maximum(sc, np.abs(np.where(hu, u, 0))), -1)
        viol = np.maximum(vl, vu); viol[idx] = -1
        ysc = max(1.0, np.abs(y).max()) if len(y) else 1.0
        sg = np.array([(-y[a] if side[i] > 0 else y[a]) for a, i in enumerate(idx)]) / ysc   # > 0 => wrong sign
        wv = int(np.argmax(viol)); ws = int(np.argmax(sg)) if len(sg) else -1
        if viol[wv] <= ftol and (ws < 0 or sg[ws] <= stol):
            lam2 = np.zeros(len(l)); lam2[idx] = y
            return (xs, lam2), k, "ok"
        if k == maxcorr: break
        if viol[wv] > ftol:
            W.add(wv); side[wv] = 1 if vl[wv] > vu[wv] else -1
        else:
            i = int(idx[ws]); W.discard(i); side[i] = 0
    return None, maxcorr, "maxcorr"


def run(model, N, B):
    tr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
    x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, tr.L, 20190, range(B))
    q = orc.build_qp_batch(model, tr, N, 0.05, x0, xr, xl, ul)
    o = orc.default_opts(polish=0)
    x, f, fl, it, lam, _ = orc.qp_solve_batch(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"], o)
    stats = {}; ks = []
    for b in range(B):
        if fl[b] != 0: stats["ipmfail"] = stats.get("ipmfail", 0) + 1; continue
        r, k, why = refine(q["H"][b].T, q["g"][b], q["A"][b].T, q["lb"][b], q["ub"][b], q["lbA"][b], q["ubA"][b], x[b], lam[b])
        stats[why] = stats.get(why, 0) + 1
        if why == "ok": ks.append(k)
    print("model %d N %d B %d: %s corrections histogram %s" % (model, N, B, stats, dict(zip(*np.unique(ks, return_counts=True)))))


if __name__ == "__main__":
    run(0, 40, 512); run(1, 40, 128); run(0, 20, 256)
