#!/usr/bin/env python3
"""Prototype 3 (numpy): working-set refinement = [pin active bounds exactly, method of multipliers on active rows]
+ single add/drop corrections of the working set, each followed by a re-factorisation.  Test infrastructure."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import oracle as orc
from mom_proto import scale

TOLV = 1e-9


def refine(H, g, A, lb, ub, lbA, ubA, x, lam, rho_r=1e6, maxfac=5, maxmom=8, big=1e9, pin=1e30, multi_add=False):
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
    nfac = 0; nmom = 0
    for fac in range(maxfac):
        fix = sv != 0; act = sr != 0
        bv = np.where(sv > 0, lv, np.where(sv < 0, uv, 0.0)); b = np.where(sr > 0, lr, np.where(sr < 0, ur, 0.0))
        R = np.where(act, rho_r, 0.0)
        M = Hs + As.T @ (R[:, None] * As) + np.diag(np.where(fix, pin, 0.0))
        try:
            L = np.linalg.cholesky(M)
        except np.linalg.LinAlgError:
            return None, nfac, nmom, "chol"
        nfac += 1
        z = np.where(fix, bv, z); y = np.where(act, y, 0.0)
        conv = False
        for it in range(maxmom):
            vz = As @ z
            pen = R * (vz - b); yh = y - pen
            Hz = Hs @ z; gz = As.T @ yh
            r = Hz + gs - gz
            scd = np.maximum(1.0, np.maximum(np.abs(gs), np.maximum(np.abs(Hz), absA.T @ np.abs(yh))))
            m_rd = np.max(np.where(fix, 0.0, np.abs(r) / scd))
            sc = np.maximum(1.0, np.abs(vz)); sc = np.maximum(sc, np.where(np.isfinite(lr), np.abs(lr), 0)); sc = np.maximum(sc, np.where(np.isfinite(ur), np.abs(ur), 0))
            m_eq = np.max(np.where(act, np.abs(vz - b) / sc, 0.0)) if m else 0.0
            nmom += 1
            if m_rd <= TOLV and m_eq <= 1e-12:
                conv = True; break
            d = -(r + As.T @ pen); d[fix] = 0.0
            dz = np.linalg.solve(L.T, np.linalg.solve(L, d)); dz[fix] = 0.0
            z = z + dz; y = yh
        if not conv:
            return None, nfac, nmom, "momstall rd %.1e eq %.1e" % (m_rd, m_eq)
        y = yh
        # full KKT check of (z, y, r on pinned)
        viol_r = np.maximum(np.where(np.isfinite(lr), lr - vz, -np.inf), np.where(np.isfinite(ur), vz - ur, -np.inf)) / sc
        viol_r[act] = -np.inf
        scv = np.maximum(1.0, np.abs(z))
        viol_v = np.maximum(np.where(np.isfinite(lv), lv - z, -np.inf), np.where(np.isfinite(uv), z - uv, -np.inf)) / scv
        viol_v[fix] = -np.inf
        ysc = max(1.0, np.abs(y).max(), np.abs(r[fix]).max() if fix.any() else 0.0)
        sg_r = np.where(sr > 0, -y, np.where(sr < 0, y, -np.inf)) / ysc
        sg_v = np.where(sv > 0, -r, np.where(sv < 0, r, -np.inf)) / ysc
        wv_r = int(np.argmax(viol_r)) if m else 0; wv_v = int(np.argmax(viol_v))
        mv = max(viol_r[wv_r] if m else -np.inf, viol_v[wv_v])
        ws_r = int(np.argmax(sg_r)) if m else 0; ws_v = int(np.argmax(sg_v))
        ms = max(sg_r[ws_r] if m else -np.inf, sg_v[ws_v])
        if mv <= 1e-10 and ms <= 1e-9:
            lam_out = np.concatenate([np.where(fix, r, 0.0) / E, y * F])
            return (z * E, lam_out), nfac, nmom, "ok"
        if mv > 1e-10:
            if multi_add:
                addr = viol_r > max(1e-10, 0.1 * mv); addv = viol_v > max(1e-10, 0.1 * mv)
                sr = np.where(addr, np.where(np.where(np.isfinite(lr), lr - vz, -np.inf) > np.where(np.isfinite(ur), vz - ur, -np.inf), 1, -1), sr)
                sv = np.where(addv, np.where(np.where(np.isfinite(lv), lv - z, -np.inf) > np.where(np.isfinite(uv), z - uv, -np.inf), 1, -1), sv)
            elif m and viol_r[wv_r] >= viol_v[wv_v]:
                sr[wv_r] = 1 if (np.isfinite(lr[wv_r]) and lr[wv_r] - vz[wv_r] > 0) else -1
            else:
                sv[wv_v] = 1 if (np.isfinite(lv[wv_v]) and lv[wv_v] - z[wv_v] > 0) else -1
        else:
            if m and sg_r[ws_r] >= sg_v[ws_v]: sr[ws_r] = 0
            else: sv[ws_v] = 0
    return None, nfac, nmom, "maxfac mv %.1e ms %.1e" % (mv, ms)


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
        out, nfac, nmom, why = refine(*args, x[b], lam[b], **kw)
        key = why.split()[0]
        res[key] = res.get(key, 0) + 1
        nf.append(nfac); nm.append(nmom)
        if key == "ok":
            kk.append(orc.qp_kkt(*args, out[0], out[1])[0])
        elif len(bad) < 6: bad.append((b, why))
    print("model %d N %d %s: %s | factorizations hist %s mean %.2f | mom steps mean %.2f max %d | kkt(oracle measure) max %.1e med %.1e\n    %s"
          % (model, N, kw, res, dict(zip(*np.unique(nf, return_counts=True))), np.mean(nf), np.mean(nm), max(nm), max(kk), np.median(kk), bad))


if __name__ == "__main__":
    run(0, 40, 512); run(1, 40, 128); run(0, 20, 256)
    run(0, 40, 512, multi_add=True); run(1, 40, 128, multi_add=True)
