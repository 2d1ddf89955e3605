#!/usr/bin/env python3
"""Prototype (numpy): method-of-multipliers solve of the working-set EQP in the solver's equilibrated coordinates,
as the HIP kernel does it (Cholesky of H~ + G_W' R G_W, correction form), for different penalties.  Counts the outer
steps needed to reach the 1e-10 acceptance test.  Test infrastructure."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import oracle as orc


def scale(H, A):
    n = H.shape[0]
    E = np.ones(n)
    for j in range(n):
        if H[j, j] > 1e-12: E[j] = 1 / np.sqrt(H[j, j])
        else:
            cm = np.abs(A[:, j]).max() if A.size else 0
            E[j] = 1 / cm if cm > 1e-12 else 1.0
    AE = A * E[None, :]
    rm = np.abs(AE).max(axis=1)
    F = np.where(rm > 1e-12, 1 / np.maximum(rm, 1e-300), 1.0)
    return E, F


def mom(H, g, A, lb, ub, lbA, ubA, x, lam, rho_r, rho_b, maxit=12, big=1e9):
    n = H.shape[0]; m = A.shape[0]
    E, F = scale(H, A)
    Hs = H * E[:, None] * E[None, :]; gs = g * E; As = A * F[:, None] * E[None, :]
    G = np.vstack([np.eye(n), As])
    l = np.concatenate([np.where(lb > -big, lb / E, -np.inf), np.where(lbA > -big, lbA * F, -np.inf)])
    u = np.concatenate([np.where(ub < big, ub / E, np.inf), np.where(ubA < big, ubA * F, np.inf)])
    xs = x / E
    lams = np.concatenate([lam[:n] * E, lam[n:] / F])
    v = G @ xs
    lo = (lams > 0) & np.isfinite(l) & (lams > np.abs(v - l))
    up = (lams < 0) & np.isfinite(u) & (-lams > np.abs(u - v))
    act = lo | up
    b = np.where(lo, l, np.where(up, u, 0.0))
    sd = np.where(lo, 1.0, np.where(up, -1.0, 0.0))
    R = np.where(act, np.concatenate([np.full(n, rho_b), np.full(m, rho_r)]), 0.0)
    M = Hs + G.T @ (R[:, None] * G)
    try:
        L = np.linalg.cholesky(M)
    except np.linalg.LinAlgError:
        return -1, "chol"
    y = np.where(act, lams, 0.0)
    z = xs.copy()
    for it in range(maxit):
        vz = G @ z
        pen = R * (vz - b)
        yh = y - pen
        Hz = Hs @ z
        gz = G.T @ yh
        r = Hz + gs - gz
        scd = np.maximum(1.0, np.maximum(np.abs(gs), np.maximum(np.abs(Hz), np.abs(gz))))
        m_rd = np.max(np.abs(r) / scd)
        sc = np.maximum(1.0, np.abs(vz)); sc = np.maximum(sc, np.where(np.isfinite(l), np.abs(l), 0)); sc = np.maximum(sc, np.where(np.isfinite(u), np.abs(u), 0))
        viol = np.where(act, np.abs(vz - b), 0.0)
        viol = np.maximum(viol, np.where(np.isfinite(l), l - vz, 0)); viol = np.maximum(viol, np.where(np.isfinite(u), vz - u, 0))
        m_rp = np.max(viol / sc)
        f2 = 0.5 * z @ Hz + gs @ z
        m_cp = np.max(np.where(act, np.abs(yh) * np.abs(vz - b), 0.0))
        m_sg = np.max(np.where(sd > 0, -yh, np.where(sd < 0, yh, 0.0)))
        conv = m_rd <= 1e-10 and m_rp <= 1e-10 and m_cp <= 1e-10 * max(1.0, abs(f2))
        if conv:
            return it, "ok" if m_sg <= 1e-8 else "sign"
        d = -(r + G.T @ pen)
        dz = np.linalg.solve(L.T, np.linalg.solve(L, d))
        z = z + dz
        y = yh
    return maxit, "noconv rd %.1e rp %.1e cp %.1e" % (m_rd, m_rp, m_cp / max(1.0, abs(f2)))


def run(model, N, B, rho_r, rho_b):
    tr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
    x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, tr.L, 20190, range(B))
    q = orc.build_qp_batch(model, tr, N, 0.05, x0, xr, xl, ul)
    o = orc.default_opts(polish=0)
    x, f, fl, it, lam, _ = orc.qp_solve_batch(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"], o)
    res = {}; its = []; bad = []
    for b in range(B):
        if fl[b] != 0: continue
        k, why = mom(q["H"][b].T, q["g"][b], q["A"][b].T, q["lb"][b], q["ub"][b], q["lbA"][b], q["ubA"][b], x[b], lam[b], rho_r, rho_b)
        key = why.split()[0]
        res[key] = res.get(key, 0) + 1
        if key == "ok": its.append(k)
        elif len(bad) < 4: bad.append((b, why))
    print("model %d N %d rho_r %.0e rho_b %.0e: %s steps hist %s %s" % (model, N, rho_r, rho_b, res, dict(zip(*np.unique(its, return_counts=True))), bad))


if __name__ == "__main__":
    for rr, rb in ((1e6, 1e6), (1e6, 1e12), (1e8, 1e12), (1e10, 1e12)):
        run(0, 40, 256, rr, rb); run(1, 40, 96, rr, rb)
