"""Oracle-independent KKT certificate (numpy only), written straight from the problem form of the reference's solver call

    min 1/2 x'Hx + x'g   s.t.  lb <= x <= ub,  lbA <= A x <= ubA            (optimizers/matlab/qpOASES/qpOASES.m:16-19)

with the multiplier convention of its fifth output (`lambda`, nV + nC entries, bounds first; qpOASES.m:49; sign per the
qpOASES manual: >= 0 lower side active, <= 0 upper side active).  For a convex QP a point that passes is THE minimiser in
x (H is positive definite on the input block, the slacks are pinned by their linear cost), whatever produced it -- so this
is the solver-independent statement of "same result as the reference's qpOASES call".  Shares no code with oracle/ (which
has its own orc_qp_kkt in C) nor with the HIP library; tests use it on both.

Batch layout = the device layout of the C ABI: H (B, nV, nV), A (B, nV, nC) (= per-instance column-major nC x nV),
vectors (B, *).
"""
import numpy as np


def kkt_certificate(H, g, A, lb, ub, lbA, ubA, x, lam, inf_bound=1e9):
    """Returns a dict of per-instance relative residuals: stationarity, primal, sign, complementarity and their max."""
    H, g, A, lb, ub, lbA, ubA, x, lam = (np.asarray(a, dtype=np.float64) for a in (H, g, A, lb, ub, lbA, ubA, x, lam))
    B, n = g.shape
    m = lbA.shape[1]
    lam_b, lam_A = lam[:, :n], lam[:, n:]
    Hx = np.einsum("bji,bj->bi", H, x)               # H symmetric; memory (b, col, row)
    Ax = np.einsum("bjr,bj->br", A, x)               # A[b, j, r] = A_{r j}
    Atl = np.einsum("bjr,br->bj", A, lam_A)
    fval = 0.5 * np.einsum("bi,bi->b", x, Hx) + np.einsum("bi,bi->b", g, x)
    # stationarity  H x + g - lam_b - A' lam_A = 0, component-wise relative to the largest term of the sum
    r = Hx + g - lam_b - Atl
    sc = np.maximum(1.0, np.maximum(np.abs(g), np.maximum(np.abs(Hx), np.abs(lam_b + Atl))))
    stat = np.max(np.abs(r) / sc, axis=1)
    # primal feasibility of all n + m two-sided rows
    v = np.concatenate([x, Ax], axis=1)
    lo = np.concatenate([lb, lbA], axis=1)
    hi = np.concatenate([ub, ubA], axis=1)
    has_lo, has_hi = lo > -inf_bound, hi < inf_bound
    scp = np.maximum(1.0, np.abs(v))
    scp = np.maximum(scp, np.where(has_lo, np.abs(lo), 0.0))
    scp = np.maximum(scp, np.where(has_hi, np.abs(hi), 0.0))
    viol = np.maximum(np.where(np.isfinite(lo), lo - v, -np.inf), np.where(np.isfinite(hi), v - hi, -np.inf))
    prim = np.max(np.maximum(viol, 0.0) / scp, axis=1)
    # multiplier signs: a positive multiplier needs a finite lower bound, a negative one a finite upper bound
    fs = np.maximum(1.0, np.abs(fval))[:, None]
    lam_all = np.concatenate([lam_b, lam_A], axis=1)
    pos, neg = lam_all > 0, lam_all < 0
    sign = np.max(np.where(pos & ~has_lo, lam_all, 0.0) / fs + np.where(neg & ~has_hi, -lam_all, 0.0) / fs, axis=1)
    # complementarity: multiplier times distance to ITS bound, relative to the objective
    gap_lo = np.where(has_lo, np.abs(v - np.where(has_lo, lo, 0.0)), 0.0)
    gap_hi = np.where(has_hi, np.abs(np.where(has_hi, hi, 0.0) - v), 0.0)
    comp = np.max(np.where(pos & has_lo, lam_all * gap_lo, 0.0) / fs + np.where(neg & has_hi, -lam_all * gap_hi, 0.0) / fs, axis=1)
    return dict(stationarity=stat, primal=prim, sign=sign, complementarity=comp, fval=fval,
                max=np.maximum(np.maximum(stat, prim), np.maximum(sign, comp)))


def working_set(lb, ub, lbA, ubA, x, Ax, lam, inf_bound=1e9):
    """Working set a vertex solution implies (qpOASES.m:58-61 encoding: -1 lower, 0 inactive, +1 upper): a side is in it iff
    its multiplier is non-zero; used to compare vertices."""
    lam = np.asarray(lam)
    return np.where(lam > 0, -1, np.where(lam < 0, 1, 0))


def vertex_from_working_set(H, g, A, lb, ub, lbA, ubA, ws):
    """The point an active-set method stops at, recomputed from a working set alone with dense numpy linear algebra, for ONE
    QP (H (n,n), A (m,n) as mathematical matrices).  ws: n+m entries in the encoding of qpOASES.m:58-61 (-1 lower, 0 inactive,
    +1 upper).  Variables with an active bound are fixed at it; the free ones and the multipliers y of the active rows solve

        [ H_FF  -A_WF' ] [x_F]   [ -g_F - H_FB x_B ]
        [ A_WF    0    ] [ y ] = [ b_W - A_WB x_B  ]

    (rows equilibrated; least squares, since the working set of a degenerate vertex may hold dependent rows; two steps of
    refinement), the bound multipliers follow from stationarity.  Returns x, lam (n+m, zero off the working set).  If (x, lam)
    passes kkt_certificate, that working set -- whoever produced it -- is optimal and x is the minimiser: a check that needs
    no solver at all."""
    H, g, A = (np.asarray(a, dtype=np.float64) for a in (H, g, A))
    n, m = H.shape[0], A.shape[0]
    A = A.reshape(m, n)
    ws = np.asarray(ws)
    wb, wc = ws[:n], ws[n:]
    Bv, Fv, W = np.nonzero(wb)[0], np.nonzero(wb == 0)[0], np.nonzero(wc)[0]
    x = np.zeros(n)
    x[Bv] = np.where(wb[Bv] < 0, np.asarray(lb)[Bv], np.asarray(ub)[Bv])
    bw = np.where(wc[W] < 0, np.asarray(lbA)[W], np.asarray(ubA)[W])
    Aw = A[np.ix_(W, Fv)]
    hd = np.diag(H)[Fv]
    cs = np.where(hd > 1e-12, 1.0 / np.sqrt(np.maximum(hd, 1e-300)), 1.0 / np.maximum(np.abs(Aw).max(axis=0) if len(W) else 1.0, 1e-300))   # x_F = cs * xt
    Aw = Aw * cs[None, :]
    rs = 1.0 / np.maximum(np.abs(Aw).max(axis=1), 1e-300) if len(W) else np.zeros(0)
    Aws = Aw * rs[:, None]
    nf, k = len(Fv), len(W)
    K = np.block([[H[np.ix_(Fv, Fv)] * cs[:, None] * cs[None, :], -Aws.T], [Aws, np.zeros((k, k))]])
    rhs = np.concatenate([(-g[Fv] - H[np.ix_(Fv, Bv)] @ x[Bv]) * cs, (bw - A[np.ix_(W, Bv)] @ x[Bv]) * rs])
    sol = np.linalg.lstsq(K, rhs, rcond=1e-13)[0]
    for _ in range(2):
        sol = sol + np.linalg.lstsq(K, rhs - K @ sol, rcond=1e-13)[0]
    x[Fv] = sol[:nf] * cs
    lam = np.zeros(n + m)
    lam[n + W] = sol[nf:] * rs
    lam[Bv] = (H @ x + g - A.T @ lam[n:])[Bv]
    return x, lam
