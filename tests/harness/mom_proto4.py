#!/usr/bin/env python3
"""Prototype 4 (numpy): working-set EQP solved by conjugate gradients on the dual of the augmented problem
(operator S = A_W (H~ + pin + rho A_W'A_W)^-1 A_W': clustered at 1/rho plus a few small outliers from nearly dependent
active rows -- CG removes the outliers, the fixed-step method of multipliers cannot), + add/drop corrections."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import oracle as orc
from mom_proto import scale

TOLV = 1e-9


def refine(H, g, A, lb, ub, lbA, ubA, x, lam, rho=1e6, maxfac=4, maxcg=12, big=1e9, pin=1e30, verbose=False):
    n = H.shape[0]; m = A.shape[0]
    E, F = scale(H, A)
    Hs = H * E[:, None] * E[None, :]; gs = g * E; As = A * F[:, None] * E[None, :]
    absA = np.abs(As)
    lv = np.where(lb > -big, lb / E, -np.inf); uv = np.where(ub < big, ub / E, np.inf)
    lr = np.where(lbA > -big, lbA * F, -np.inf); ur = np.where(ubA < big, ubA * F, np.inf)
    xs = x / E; lamv = lam[:n] * E; lamr = lam[n:] / F
    vr = As @ xs
    sv = np.where((lamv > 0) & np.isfinite(lv) & (lamv > np.abs(xs - lv)), 1, np.where((lamv < 0) & np.isfinite(uv) & (-lamv > np.abs(uv - xs)), -1, 0))
    sr = np.where((lamr > 0) & np.isfinite(lr) & (lamr > np.abs(vr - lr)), 1, np.where((lamr < 0) & np.isfinite(ur) & (-lamr > np.abs(ur - vr)), -1, 0))
    z = xs.copy(); y = np.where(sr != 0, lamr, 0.0)
    nfac = 0; nsol = 0
    scr = np.maximum(1.0, np.maximum(np.where(np.isfinite(lr), np.abs(lr), 0), np.where(np.isfinite(ur), np.abs(ur), 0)))
    for fac in range(maxfac):
        fix = sv != 0; act = sr != 0
        bv = np.where(sv > 0, lv, np.where(sv < 0, uv, 0.0)); b = np.where(sr > 0, lr, np.where(sr < 0, ur, 0.0))
        R = np.where(act, rho, 0.0)
        M = Hs + As.T @ (R[:, None] * As) + np.diag(np.where(fix, pin, 0.0))
        try:
            L = np.linalg.cholesky(M)
        except np.linalg.LinAlgError:
            return None, nfac, nsol, "chol"
        nfac += 1
        solve = lambda d: np.linalg.solve(L.T, np.linalg.solve(L, np.where(fix, 0.0, d)))
        z = np.where(fix, bv, z); y = np.where(act, y, 0.0)
        # z(y): minimiser of the augmented Lagrangian for the current y (one solve), then dual CG on c(y) = A_W z(y) - b = 0
        def grad(zz, yy):
            vz = As @ zz
            c = np.where(act, vz - b, 0.0)
            Hz = Hs @ zz
            rr = Hz + gs - As.T @ (yy - R * c)      # gradient of L_rho(z, y) in z
            return rr, c, vz, Hz
        rr, c, vz, Hz = grad(z, y)
        dz = solve(-rr); nsol += 1
        z = z + dz
        rr, c, vz, Hz = grad(z, y)
        # CG on S dy = -c with S = A_W M^-1 A_W' (free variables)
        r = -c; p = r.copy(); Atp = As.T @ p; rs = r @ r
        conv = False
        for it in range(maxcg):
            sc = np.maximum(scr, np.abs(vz))
            m_eq = np.max(np.abs(c) / sc) if m else 0.0
            gz = As.T @ y
            scd = np.maximum(1.0, np.maximum(np.abs(gs), np.maximum(np.abs(Hz), absA.T @ np.abs(y))))
            m_rd = np.max(np.where(fix, 0.0, np.abs(Hz + gs - gz) / scd))
            if verbose: print("   fac %d cg %d eq %.2e rd %.2e" % (fac, it, m_eq, m_rd))
            if m_eq <= 1e-11 and m_rd <= TOLV:
                conv = True; break
            w = solve(Atp); nsol += 1
            q = np.where(act, As @ w, 0.0)
            alpha = rs / (p @ q)
            y = y + alpha * p; z = z + alpha * w
            # fresh evaluation (one pass over A + H z): c, and the stationarity residual
            vz = As @ z; c = np.where(act, vz - b, 0.0); Hz = Hs @ z
            r = -c
            rs_new = r @ r
            p = r + (rs_new / rs) * p; rs = rs_new
            Atp = As.T @ p
        if not conv:
            return None, nfac, nsol, "cgstall eq %.1e rd %.1e" % (m_eq, m_rd)
        rfull = Hz + gs - As.T @ y              # on pinned variables: their bound multipliers
        viol_r = np.maximum(np.where(np.isfinite(lr), lr - vz, -np.inf), np.where(np.isfinite(ur), vz - ur, -np.inf)) / sc
        viol_r[act] = -np.inf
        scv = np.maximum(1.0, np.abs(z))
        viol_v = np.maximum(np.where(np.isfinite(lv), lv - z, -np.inf), np.where(np.isfinite(uv), z - uv, -np.inf)) / scv
        viol_v[fix] = -np.inf
        ysc = max(1.0, np.abs(y).max(), np.abs(rfull[fix]).max() if fix.any() else 0.0)
        sg_r = np.where(sr > 0, -y, np.where(sr < 0, y, -np.inf)) / ysc
        sg_v = np.where(sv > 0, -rfull, np.where(sv < 0, rfull, -np.inf)) / ysc
        wv_r = int(np.argmax(viol_r)) if m else 0; wv_v = int(np.argmax(viol_v))
        mv = max(viol_r[wv_r] if m else -np.inf, viol_v[wv_v])
        ws_r = int(np.argmax(sg_r)) if m else 0; ws_v = int(np.argmax(sg_v))
        ms = max(sg_r[ws_r] if m else -np.inf, sg_v[ws_v])
        if verbose: print("   fac %d: viol %.2e sign %.2e" % (fac, mv, ms))
        if mv <= 1e-10 and ms <= 1e-9:
            lam_out = np.concatenate([np.where(fix, rfull, 0.0) / E, y * F])
            return (z * E, lam_out), nfac, nsol, "ok"
        if mv > 1e-10:
            if m and viol_r[wv_r] >= viol_v[wv_v]:
                sr[wv_r] = 1 if (np.isfinite(lr[wv_r]) and lr[wv_r] - vz[wv_r] > 0) else -1
            else:
                sv[wv_v] = 1 if (np.isfinite(lv[wv_v]) and lv[wv_v] - z[wv_v] > 0) else -1
        else:
            if m and sg_r[ws_r] >= sg_v[ws_v]: sr[ws_r] = 0
            else: sv[ws_v] = 0
    return None, nfac, nsol, "maxfac mv %.1e ms %.1e" % (mv, ms)


def run(model, N, B, **kw):
    tr = orc.Track.load(os.path.join(ROOT, "fsae-mpc_amd", "tracks", "fsg2019.json"))
    x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, tr.L, 20190, range(B))
    q = orc.build_qp_batch(model, tr, N, 0.05, x0, xr, xl, ul)
    o = orc.default_opts(polish=0)
    x, f, fl, it, lam, _ = orc.qp_solve_batch(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"], o)
    res = {}; nf = []; nm = []; bad = []; kk = []
    for b in range(B):
        if fl[b] != 0: continue
        args = (q["H"][b].T, q["g"][b], q["A"][b].T, q["lb"][b], q["ub"][b], q["lbA"][b], q["ubA"][b])
        out, nfac, nsol, why = refine(*args, x[b], lam[b], **kw)
        key = why.split()[0]
        res[key] = res.get(key, 0) + 1
        nf.append(nfac); nm.append(nsol)
        if key == "ok":
            kk.append(orc.qp_kkt(*args, out[0], out[1])[0])
        elif len(bad) < 6: bad.append((b, why))
    print("model %d N %d %s: %s | factorizations hist %s mean %.2f | solves mean %.2f hist %s | kkt(oracle measure) max %.1e med %.1e\n    %s"
          % (model, N, kw, res, dict(zip(*np.unique(nf, return_counts=True))), np.mean(nf), np.mean(nm), dict(zip(*np.unique(nm, return_counts=True))), max(kk), np.median(kk), bad))


if __name__ == "__main__":
    run(0, 40, 512); run(1, 40, 128); run(0, 20, 256)
