"""numpy prototype of the IPM (tuning only; not shipped in any product path)."""
import numpy as np

def ipm(H, g, A, lb, ub, lbA, ubA, tol=1e-9, max_iter=60, inf=1e9, corrector=True, scale=True, verbose=False,
        init_t=1.0, init_z=1.0, sig_pow=3.0, init_mode='plain', corr_guard=0.0, mu_floor_fac=0.1, corr_pow=0.0, step_heur=False, tol_x=None, mu_abs=None):
    last_step = np.inf
    n = len(g); m = A.shape[0]; mt = n + m
    E = np.ones(n); F = np.ones(m)
    if scale:
        d = np.diag(H).copy()
        cm = np.abs(A).max(axis=0) if m else np.ones(n)
        E = np.where(d > 1e-12, 1/np.sqrt(np.maximum(d, 1e-300)), np.where(cm > 1e-12, 1/np.maximum(cm,1e-300), 1.0))
        rm = np.abs(A * E[None, :]).max(axis=1) if m else np.ones(0)
        F = np.where(rm > 1e-12, 1/np.maximum(rm,1e-300), 1.0)
    Hs = H * E[:, None] * E[None, :]; gs = g * E
    As = A * F[:, None] * E[None, :]
    G = np.vstack([np.eye(n), As])
    l = np.concatenate([lb / E, lbA * F]); u = np.concatenate([ub / E, ubA * F])
    hl = np.concatenate([lb, lbA]) > -inf; hu = np.concatenate([ub, ubA]) < inf
    l = np.where(hl, l, -np.inf); u = np.where(hu, u, np.inf)
    cnt = hl.sum() + hu.sum()
    x = np.clip(np.zeros(n), np.where(hl[:n], l[:n], -np.inf), np.where(hu[:n], u[:n], np.inf))
    v = G @ x
    tl = np.where(hl, np.maximum(v - l, init_t), 1.0); tu = np.where(hu, np.maximum(u - v, init_t), 1.0)
    zl = np.where(hl, init_z, 0.0); zu = np.where(hu, init_z, 0.0)
    if init_mode == 'dualfix':
        r = Hs @ x + gs - G[n:].T @ (zl[n:] - zu[n:])
        zl[:n] = np.where(hl[:n], np.maximum(r, 0) + init_z, 0.0)
        zu[:n] = np.where(hu[:n], np.maximum(-r, 0) + init_z, 0.0)
    elif init_mode == 'ooqp':
        pass
    hist = []
    for it in range(max_iter + 1):
        v = G @ x
        rpl = np.where(hl, v - l - tl, 0.0); rpu = np.where(hu, u - v - tu, 0.0)
        gap = (tl * zl)[hl].sum() + (tu * zu)[hu].sum(); mu = gap / cnt
        Hx = Hs @ x; Gz = G.T @ (zl - zu)
        rd = Hx + gs - Gz
        fval = 0.5 * x @ Hx + gs @ x
        sc = np.maximum(1.0, np.maximum(np.abs(gs), np.maximum(np.abs(Hx), np.abs(Gz))))
        rd_rel = np.max(np.abs(rd) / sc)
        scp = np.maximum(1.0, np.abs(v))
        rp_rel = max(np.max(np.where(hl, np.abs(rpl) / np.maximum(scp, np.abs(np.where(hl, l, 0))), 0)),
                     np.max(np.where(hu, np.abs(rpu) / np.maximum(scp, np.abs(np.where(hu, u, 0))), 0)))
        gap_rel = gap / max(1.0, abs(fval))
        if verbose: print("it %2d mu %.3e rd %.3e rp %.3e gap %.3e f %.8e" % (it, mu, rd_rel, rp_rel, gap_rel, fval))
        hist.append((mu, rd_rel, rp_rel, gap_rel))
        okx = True
        if tol_x is not None:
            # Newton-decrement style test: predicted remaining move (affine direction) in unscaled coordinates
            D_ = np.where(hl, zl / tl, 0) + np.where(hu, zu / tu, 0)
            M_ = Hs + G.T @ (D_[:, None] * G)
            w_ = np.where(hl, -(zl / tl) * rpl, 0) + np.where(hu, (zu / tu) * rpu, 0)
            dxa_ = np.linalg.solve(M_, -(Hx + gs) + G.T @ w_)
            okx = np.abs(dxa_ * E).max() <= tol_x * max(1.0, np.abs(x * E).max())
        if mu_abs is not None:
            okx = okx and (mu <= mu_abs)
        if rd_rel <= tol and rp_rel <= tol and gap_rel <= tol and okx:
            return x * E, it, 0, (zl - zu) * np.concatenate([1 / E, F]), hist
        if it == max_iter: break
        D = np.where(hl, zl / tl, 0) + np.where(hu, zu / tu, 0)
        M = Hs + G.T @ (D[:, None] * G)
        try:
            L = np.linalg.cholesky(M)
        except np.linalg.LinAlgError:
            L = np.linalg.cholesky(M + 1e-12 * np.trace(M) / n * np.eye(n))
        solve = lambda r: np.linalg.solve(L.T, np.linalg.solve(L, r))
        w_aff = np.where(hl, -(zl / tl) * rpl, 0) + np.where(hu, (zu / tu) * rpu, 0)
        rhs = -(Hx + gs) + G.T @ w_aff
        dxa = solve(rhs); dv = G @ dxa
        dtl = dv + rpl; dtu = -dv + rpu
        dzl = -zl - (zl / tl) * dtl; dzu = -zu - (zu / tu) * dtu
        def maxstep(t, dt, mask):
            neg = mask & (dt < 0)
            return np.min(-t[neg] / dt[neg]) if neg.any() else np.inf
        a_aff = min(1.0, maxstep(tl, dtl, hl), maxstep(tu, dtu, hu), maxstep(zl, dzl, hl), maxstep(zu, dzu, hu))
        mu_aff = (((tl + a_aff * dtl) * (zl + a_aff * dzl))[hl].sum() + ((tu + a_aff * dtu) * (zu + a_aff * dzu))[hu].sum()) / cnt
        sigma = min(1.0, (mu_aff / mu) ** sig_pow)
        mu_floor = mu_floor_fac * tol * max(1.0, abs(fval)) / cnt
        if mu > 0: sigma = min(1.0, max(sigma, mu_floor / mu))
        cw = (a_aff ** corr_pow) if a_aff >= corr_guard else 0.0
        cl = np.where(hl, sigma * mu - (cw * dtl * dzl if corrector else 0), 0)
        cu = np.where(hu, sigma * mu - (cw * dtu * dzu if corrector else 0), 0)
        w2 = cl / tl - cu / tu
        dx = solve(rhs + G.T @ w2); dv = G @ dx
        dtl = dv + rpl; dtu = -dv + rpu
        dzl = -zl + cl / tl - (zl / tl) * dtl; dzu = -zu + cu / tu - (zu / tu) * dtu
        a = min(maxstep(tl, dtl, hl), maxstep(tu, dtu, hu), maxstep(zl, dzl, hl), maxstep(zu, dzu, hu))
        tau = min(max(0.995, 1 - mu), 0.99999)
        if step_heur:
            amax = min(a, 1e300)
            # OOQP-style Mehrotra step heuristic on the blocking pair
            cands = []
            for (t_, dt_, z_, dz_, msk) in ((tl, dtl, zl, dzl, hl), (tu, dtu, zu, dzu, hu)):
                for (p, dp, d_, dd_, primal) in ((t_, dt_, z_, dz_, True), (z_, dz_, t_, dt_, False)):
                    neg = msk & (dp < 0)
                    if neg.any():
                        r = np.where(neg, -p / np.where(neg, dp, -1.0), np.inf)
                        i = int(np.argmin(r)); cands.append((r[i], p[i], dp[i], d_[i], dd_[i]))
            if cands and amax < 1e299:
                r, p, dp, d_, dd_ = min(cands, key=lambda c: c[0])
                gamma_f = 0.99; gamma_a = 1.0 / (1.0 - gamma_f)
                mufull = (((tl + amax * dtl) * (zl + amax * dzl))[hl].sum() + ((tu + amax * dtu) * (zu + amax * dzu))[hu].sum()) / cnt / gamma_a
                a_h = (-p + mufull / (d_ + amax * dd_)) / dp
                a = min(1.0, max(a_h, gamma_f * amax))
            else:
                a = 1.0
        else:
            a = min(1.0, tau * a)
        if verbose: print("      a_aff %.3e sigma %.3e a %.3e  |dx| %.3e res %.2e" % (a_aff, sigma, a, np.abs(dx).max(), np.abs(M@dx-(rhs+G.T@w2)).max()))
        last_step = a * np.abs(dx).max()
        x = x + a * dx; tl = tl + a * dtl; tu = tu + a * dtu; zl = zl + a * dzl; zu = zu + a * dzu
    return x * E, it, 1, (zl - zu) * np.concatenate([1 / E, F]), hist
