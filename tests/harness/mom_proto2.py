#!/usr/bin/env python3
"""Prototype 2 (numpy): working-set EQP with the active *bounds* eliminated exactly (variables pinned, their multipliers
read off the stationarity residual) and the method of multipliers only on the active general rows.  Test infrastructure."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import oracle as orc
from mom_proto import scale


def mom2(H, g, A, lb, ub, lbA, ubA, x, lam, rho_r, maxit=12, big=1e9, tol=1e-10, pin=1e30):
    n = H.shape[0]; m = A.shape[0]
    E, F = scale(H, A)
    Hs = H * E[:, None] * E[None, :]; gs = g * E; As = A * F[:, None] * E[None, :]
    lv = np.where(lb > -big, lb / E, -np.inf); uv = np.where(ub < big, ub / E, np.inf)
    lr = np.where(lbA > -big, lbA * F, -np.inf); ur = np.where(ubA < big, ubA * F, np.inf)
    xs = x / E; lamv = lam[:n] * E; lamr = lam[n:] / F
    vr = As @ xs
    lo_v = (lamv > 0) & np.isfinite(lv) & (lamv > np.abs(xs - lv)); up_v = (lamv < 0) & np.isfinite(uv) & (-lamv > np.abs(uv - xs))
    lo_r = (lamr > 0) & np.isfinite(lr) & (lamr > np.abs(vr - lr)); up_r = (lamr < 0) & np.isfinite(ur) & (-lamr > np.abs(ur - vr))
    fix = lo_v | up_v; act = lo_r | up_r
    b = np.where(lo_r, lr, np.where(up_r, ur, 0.0)); sd = np.where(lo_r, 1.0, np.where(up_r, -1.0, 0.0))
    R = np.where(act, rho_r, 0.0)
    M = Hs + As.T @ (R[:, None] * As) + np.diag(np.where(fix, pin, 0.0))
    try:
        L = np.linalg.cholesky(M)
    except np.linalg.LinAlgError:
        return -1, "chol"
    z = xs.copy(); z[lo_v] = lv[lo_v]; z[up_v] = uv[up_v]
    y = np.where(act, lamr, 0.0)
    for it in range(maxit):
        vz = As @ z
        pen = R * (vz - b)
        yh = y - pen
        Hz = Hs @ z; gz = As.T @ yh
        r = Hz + gs - gz                       # on pinned variables this IS their bound multiplier
        scd = np.maximum(1.0, np.maximum(np.abs(gs), np.maximum(np.abs(Hz), np.abs(gz))))
        m_rd = np.max(np.where(fix, 0.0, np.abs(r) / scd))
        m_sgv = np.max(np.where(lo_v, -r, np.where(up_v, r, 0.0)) / scd)      # wrong-sign bound multipliers (relative)
        sc = np.maximum(1.0, np.abs(vz)); sc = np.maximum(sc, np.where(np.isfinite(lr), np.abs(lr), 0)); sc = np.maximum(sc, np.where(np.isfinite(ur), np.abs(ur), 0))
        viol = np.where(act, np.abs(vz - b), 0.0)
        viol = np.maximum(viol, np.where(np.isfinite(lr), lr - vz, 0)); viol = np.maximum(viol, np.where(np.isfinite(ur), vz - ur, 0))
        m_rp = np.max(viol / sc) if m else 0.0
        scv = np.maximum(1.0, np.abs(z))
        m_rpv = np.max(np.maximum(np.where(np.isfinite(lv), lv - z, 0), np.where(np.isfinite(uv), z - uv, 0)) / scv)
        f2 = 0.5 * z @ Hz + gs @ z
        m_cp = np.max(np.where(act, np.abs(yh) * np.abs(vz - b), 0.0)) if m else 0.0
        m_sg = np.max(np.where(sd > 0, -yh, np.where(sd < 0, yh, 0.0))) if m else 0.0
        conv = m_rd <= tol and max(m_rp, m_rpv) <= tol and m_cp <= tol * max(1.0, abs(f2))
        if conv:
            ok = m_sg <= 1e-8 and m_sgv <= 1e-9
            return it, "ok" if ok else "sign sg %.1e sgv %.1e" % (m_sg, m_sgv)
        d = -(r + As.T @ pen); d[fix] = 0.0
        dz = np.linalg.solve(L.T, np.linalg.solve(L, d))
        z = z + dz; z[lo_v] = lv[lo_v]; z[up_v] = uv[up_v]
        y = yh
    return maxit, "noconv rd %.1e rp %.1e rpv %.1e cp %.1e" % (m_rd, m_rp, m_rpv, m_cp / max(1.0, abs(f2)))


def run(model, N, B, rho_r, tol=1e-10):
    tr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
    x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, tr.L, 20190, range(B))
    q = orc.build_qp_batch(model, tr, N, 0.05, x0, xr, xl, ul)
    o = orc.default_opts(polish=0)
    x, f, fl, it, lam, _ = orc.qp_solve_batch(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"], o)
    res = {}; its = []; bad = []
    for b in range(B):
        if fl[b] != 0: continue
        k, why = mom2(q["H"][b].T, q["g"][b], q["A"][b].T, q["lb"][b], q["ub"][b], q["lbA"][b], q["ubA"][b], x[b], lam[b], rho_r, tol=tol)
        key = why.split()[0]
        res[key] = res.get(key, 0) + 1
        if key == "ok": its.append(k)
        elif len(bad) < 6: bad.append((b, why))
    print("model %d N %d rho_r %.0e tol %.0e: %s steps hist %s\n    %s" % (model, N, rho_r, tol, res, dict(zip(*np.unique(its, return_counts=True))), bad))


if __name__ == "__main__":
    for rr in (1e4, 1e5, 1e6):
        run(0, 40, 256, rr); run(1, 40, 96, rr)
