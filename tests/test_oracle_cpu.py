"""CPU tests of the oracle (oracle/): the restatement of the reference's construction path is checked through
the identities the reference's own maths implies (there are no reference tests/golden vectors: SURVEY 4), the
QP solve through KKT certificates, known answers and an independent scipy cross-check, and both against the
committed golden fixtures."""
import numpy as np
import pytest
from conftest import golden_files, relerr


def _rand_state(model, rng, track):
    if model == 0:
        return np.array([rng.uniform(0, track.L), rng.uniform(-0.5, 0.5), rng.uniform(-0.1, 0.1), rng.uniform(5, 20), rng.uniform(-0.1, 0.1)])
    return np.array([rng.uniform(0, track.L), rng.uniform(-0.5, 0.5), rng.uniform(-0.1, 0.1), rng.uniform(5, 20),
                     rng.uniform(-0.2, 0.2), rng.uniform(-0.3, 0.3), rng.uniform(-0.1, 0.1)])


def test_kappa_matches_numpy_restatement(orc, otrack):
    # interpolate_curvature.m:12-18 restated independently in numpy
    tr = otrack
    rng = np.random.default_rng(0)
    for s in np.concatenate([rng.uniform(-50, 2 * tr.L, 50), [0.0, tr.dl, tr.L - 1e-9]]):
        t = np.mod(s, tr.dl * tr.M); i = int(np.floor(t / tr.dl)); u = t / tr.dl - i
        d = lambda P: (-3 * (1 - u) ** 2 * P[i, 0] + 3 * (3 * u * u - 4 * u + 1) * P[i, 1] + 3 * (2 * u - 3 * u * u) * P[i, 2] + 3 * u * u * P[i, 3]) / tr.dl
        dd = lambda P: (6 * (1 - u) * P[i, 0] + 6 * (3 * u - 2) * P[i, 1] + 6 * (1 - 3 * u) * P[i, 2] + 6 * u * P[i, 3]) / tr.dl ** 2
        k = (d(tr.xP) * dd(tr.yP) - dd(tr.xP) * d(tr.yP)) / (d(tr.xP) ** 2 + d(tr.yP) ** 2) ** 1.5
        assert abs(orc.kappa(tr, float(s)) - k) <= 1e-12 * max(1, abs(k))
    # periodic in s with period dl*M (interpolate_spline_d.m:12)
    assert abs(orc.kappa(tr, 3.7) - orc.kappa(tr, 3.7 + tr.dl * tr.M)) < 1e-9


@pytest.mark.parametrize("model", [0, 1])
def test_jacobian_vs_finite_differences(orc, otrack, model):
    rng = np.random.default_rng(1)
    nx = 5 if model == 0 else 7
    for _ in range(10):
        x = _rand_state(model, rng, otrack); u = np.array([rng.uniform(-5, 5), rng.uniform(-0.2, 0.2)])
        A, _ = orc.A_model(model, otrack, x)
        fd = np.zeros((nx, nx))
        for j in range(nx):
            h = 1e-6 * max(1.0, abs(x[j])); xp = x.copy(); xm = x.copy(); xp[j] += h; xm[j] -= h
            fd[:, j] = (orc.f_model(model, otrack, xp, u) - orc.f_model(model, otrack, xm, u)) / (2 * h)
        mask = np.ones((nx, nx), bool)
        mask[:, 0] = False        # kappa'(s) terms are omitted by the reference (A_curv_kin.m:44-48, A_curv_dyn.m:99-105)
        if model == 1:
            mask[4, 5] = False    # documented mismatch: A_curv_dyn.m:90 uses -m*x_d_hat, f_curv_dyn.m:60 uses -m*x_d
            assert abs((A[4, 5] - fd[4, 5]) + 5 * np.exp(-x[3] / 5)) < 1e-5
        assert np.max(np.abs(A - fd)[mask]) < 2e-5 * max(1.0, np.max(np.abs(fd)))
        assert np.all(A[:, 0] == 0)


@pytest.mark.parametrize("model,integ", [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2)])
def test_linearisation_reproduces_f_at_expansion_point(orc, otrack, model, integ):
    # d = f - A x - B u  (rk2_*.m:48)  =>  A x + B u + d equals the integrator's effective slope at (x,u)
    rng = np.random.default_rng(2)
    nx, N, dt = (5 if model == 0 else 7), 6, 0.05
    x = np.stack([_rand_state(model, rng, otrack) for _ in range(N)], axis=1)
    u = np.stack([[rng.uniform(-5, 5), rng.uniform(-0.2, 0.2)] for _ in range(N)], axis=1)
    A, B, d = orc.linearise(model, integ, otrack, x, u, dt)
    for k in range(N):
        f = lambda xx: orc.f_model(model, otrack, xx, u[:, k])
        k1 = f(x[:, k])
        if integ == 0: slope = k1
        elif integ == 1: slope = f(x[:, k] + k1 * dt / 2)
        else:
            k2 = f(x[:, k] + k1 * dt / 2); k3 = f(x[:, k] + k2 * dt / 2); k4 = f(x[:, k] + k3 * dt)
            slope = (k1 + 2 * k2 + 2 * k3 + k4) / 6
        assert np.allclose(A[:, :, k] @ x[:, k] + B[:, :, k] @ u[:, k] + d[:, k], slope, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("model", [0, 1])
def test_condensing_equals_rollout(orc, otrack, model):
    # sequential_integration.m: x = A_bar x0 + B_bar u + d_bar must equal stepping x+ = Ad_k x + Bd_1 u_k + dd_k
    rng = np.random.default_rng(3)
    nx, N, dt = (5 if model == 0 else 7), 9, 0.05
    A = rng.normal(size=(nx, nx, N)); B = rng.normal(size=(nx, 2, N)); d = rng.normal(size=(nx, N))
    A_bar, B_bar, d_bar = orc.sequential_integration(A, B, d, dt)
    x0 = rng.normal(size=nx); u = rng.normal(size=(2, N))
    pred = (A_bar @ x0 + B_bar @ u.T.reshape(-1) + d_bar).reshape(N, nx)
    x = x0.copy()
    for k in range(N):
        x = (np.eye(nx) + dt * A[:, :, k]) @ x + dt * B[:, :, 0] @ u[:, k] + dt * d[:, k]   # slice 1 always (quirk C-1)
        assert np.allclose(pred[k], x, rtol=1e-11, atol=1e-11)
    # block lower-triangular
    for i in range(N):
        assert np.all(B_bar[: i * nx, 2 * i: 2 * i + 2] == 0)


@pytest.mark.parametrize("model,N", [(0, 8), (1, 6)])
def test_qp_layout_and_hessian(orc, otrack, model, N):
    nx, ns, nV, nC = orc.dims(model, N)
    x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, otrack.L, 11, [5])
    q = orc.build_qp(model, otrack, N, 0.05, x0[0], xr[0].T, xl[0].T, ul[0].T)
    H, A, Bt = q["H"], q["A"], q["Bt"]
    assert H.shape == (nV, nV) and A.shape == (nC, nV)
    assert np.allclose(H, H.T, rtol=0, atol=1e-9 * np.abs(H).max())
    ev = np.linalg.eigvalsh((H + H.T) / 2)
    assert ev.min() > -1e-8 * ev.max() and np.all(np.diag(H)[:2 * N] >= 20.0) and np.all(H[2 * N:, :] == 0)
    # rows [1] v, [2] delta, [3]/[4] n with +-1 slack (kinematic_state_constraints.m:27-42)
    vidx, didx = 3, nx - 1
    assert np.array_equal(A[:N, :], Bt[vidx::nx, :]) and np.array_equal(A[N:2 * N, :], Bt[didx::nx, :])
    assert np.array_equal(A[2 * N:3 * N, :2 * N], Bt[1::nx, :2 * N]) and np.all(A[2 * N:3 * N, 2 * N] == 1) and np.all(A[3 * N:4 * N, 2 * N] == -1)
    assert np.all(q["ubA"][2 * N:3 * N] == 1e10) and np.all(q["lbA"][3 * N:4 * N] == -1e10) and np.all(np.isinf(q["ubA"][:N]))
    aff = q["A_bar"] @ x0[0] + q["d_bar"]
    assert np.allclose(q["lbA"][N:2 * N], -0.4 - aff[didx::nx]) and np.allclose(q["ubA"][3 * N:4 * N], 0.75 - aff[1::nx])
    assert np.all(q["lb"][:2 * N:2] == -10) and np.all(q["ub"][1:2 * N:2] == 0.4) and np.all(q["lb"][2 * N:] == 0) and np.all(np.isinf(q["ub"][2 * N:]))
    if model == 0:
        assert np.all(A[4 * N:5 * N, -1] == 1) and np.all(A[5 * N:, -1] == -1) and q["g"][-1] == 1e8
    else:
        assert np.array_equal(q["g"][-4:], [1e8, 1e6, 1e6, 1e4]) and np.all(A[8 * N:, -1] == -1) and np.all(np.isinf(q["lbA"][8 * N:]))
        assert np.all(A[4 * N:6 * N:2, 2 * N + 1] == 1) and np.all(A[6 * N + 1:8 * N:2, 2 * N + 2] == -1)
    # generate_qp.m:29-33
    Q = np.tile(np.array([5, 250, 2000] + [0] * (nx - 3), float), N); Q[-nx:] *= 10
    Rb = np.concatenate([np.full(2 * N, 10.0), np.zeros(ns)])
    assert np.allclose(H, 2 * (Bt.T @ (Q[:, None] * Bt) + np.diag(Rb)), rtol=1e-12, atol=1e-9)
    r = aff - xr[0].reshape(-1)
    assert np.allclose(q["g"][:2 * N], (2 * Bt.T @ (Q * r))[:2 * N], rtol=1e-10, atol=1e-8) and np.isclose(q["const"], r @ (Q * r))


def test_known_answer_qps(orc):
    H = 2 * np.eye(2); g = np.array([-2., -4.])
    x, f, fl, it, lam = orc.qp_solve(H, g, np.zeros((0, 2)), [0, 0], [1.5, 1.5], [], [])
    assert fl == 0 and np.allclose(x, [1, 1.5], atol=1e-8) and np.isclose(f, -4.75) and lam[1] < 0 and abs(lam[0]) < 1e-6
    x, f, fl, it, lam = orc.qp_solve(H, g, np.array([[1., 1.]]), [0, 0], [1.5, 1.5], [-np.inf], [2.0])
    assert fl == 0 and np.allclose(x, [0.5, 1.5], atol=1e-7) and np.isclose(f, -4.5)
    # infeasible: x <= -1 and x >= 1  -> -2 (qpOASES.m:46)
    x, f, fl, it, lam = orc.qp_solve(np.eye(1), [0.], np.array([[1.]]), [1.], [np.inf], [-np.inf], [-1.])
    assert fl == -2
    # inverted bounds are infeasible immediately
    assert orc.qp_solve(np.eye(1), [0.], np.zeros((0, 1)), [1.], [0.], [], [])[2] == -2
    # active soft constraint with a linear penalty (the reference's slack pattern)
    H = np.diag([2., 0.]); g = np.array([0., 100.])
    x, f, fl, it, lam = orc.qp_solve(H, g, np.array([[1., 1.]]), [-10, 0], [10, np.inf], [3.], [np.inf])
    assert fl == 0 and np.allclose(x, [3, 0], atol=1e-6)


@pytest.mark.parametrize("model,N,inst", [(0, 5, 3), (0, 10, 3), (1, 5, 3), (0, 20, 3), (1, 10, 3), (1, 20, 5)])
def test_solution_vs_scipy(orc, otrack, model, N, inst):
    """Independent solver (scipy trust-constr, a different algorithm and code base) on LTV-MPC QPs up to nV = 44, nC = 400: the
    oracle's refined solution is the same point to 1e-6 in x (measured: <= 4e-8) and 1e-8 in the objective."""
    from scipy.optimize import Bounds, LinearConstraint, minimize
    x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, otrack.L, 20190, [inst])
    q = orc.build_qp(model, otrack, N, 0.05, x0[0], xr[0].T, xl[0].T, ul[0].T)
    H, g, A = q["H"], q["g"], q["A"]
    x, f, fl, it, lam = orc.qp_solve(H, g, A, q["lb"], q["ub"], q["lbA"], q["ubA"])
    assert fl == 0 and orc.qp_kkt(H, g, A, q["lb"], q["ub"], q["lbA"], q["ubA"], x, lam)[0] < 1e-7
    lbA = np.where(q["lbA"] < -1e9, -np.inf, q["lbA"]); ubA = np.where(q["ubA"] > 1e9, np.inf, q["ubA"])
    res = minimize(lambda z: 0.5 * z @ H @ z + g @ z, np.clip(x * 0, q["lb"], np.minimum(q["ub"], 1e3)), jac=lambda z: H @ z + g, hess=lambda z: H,
                   method="trust-constr", bounds=Bounds(q["lb"], q["ub"]), constraints=[LinearConstraint(A, lbA, ubA)],
                   options=dict(gtol=1e-10, xtol=1e-12, barrier_tol=1e-12, maxiter=5000))
    assert abs(res.fun - f) <= 1e-8 * max(1.0, abs(f))
    assert np.max(np.abs(res.x - x)) <= 1e-6 * max(1.0, np.max(np.abs(x)))


@pytest.mark.parametrize("path", golden_files())
def test_oracle_reproduces_golden(orc, track_path, path):
    z = np.load(path)
    name = path.split("/")[-1].split("_")[0]
    tr = orc.Track.load(track_path(name))
    model, N = int(z["model"]), int(z["N"])
    x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, tr.L, 20190, z["ids"])
    for a, k in ((x0, "x0"), (xl, "x_lin"), (ul, "u_lin"), (xr, "x_ref")):
        assert np.array_equal(a, z[k])
    q = orc.build_qp_batch(model, tr, N, 0.05, x0, xr, xl, ul)
    for k in ("H", "g", "A", "lb", "ub", "lbA", "ubA", "const"):
        assert relerr(q[k], z[k]) <= 1e-12, k
    x, f, fl, it, lam, _ = orc.qp_solve_batch(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"])
    assert (fl == 0).all()
    assert np.max(np.abs(f - z["fval"]) / np.maximum(1, np.abs(z["fval"]))) < 1e-8
    assert np.max(np.abs(x - z["x"]).max(axis=1) / np.maximum(1, np.abs(z["x"]).max(axis=1))) < 1e-5


def test_batch_solver_statistics(orc, otrack):
    # the solver the GPU kernel mirrors: every synthetic instance of the headline shape converges to KKT <= 1e-6
    x0, xl, ul, xr = orc.synth_instances(0, 40, 0.05, otrack.L, 20190, range(96))
    q = orc.build_qp_batch(0, otrack, 40, 0.05, x0, xr, xl, ul)
    x, f, fl, it, lam, _ = orc.qp_solve_batch(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"])
    assert (fl == 0).all() and it.mean() < 25
    for b in range(0, 96, 7):
        assert orc.qp_kkt(q["H"][b].T, q["g"][b], q["A"][b].T, q["lb"][b], q["ub"][b], q["lbA"][b], q["ubA"][b], x[b], lam[b])[0] < 1e-6


def test_reference_live_generator(orc):
    # main.m:107-114
    x0 = np.array([3.0, 0, 0, 19.2, 0])
    xr = orc.reference_live(5, 4, 0.05, x0)
    assert np.allclose(xr[3], [19.7, 20, 20, 20]) and np.allclose(xr[0], 3.0 + np.cumsum(xr[3] * 0.05)) and np.all(xr[[1, 2, 4]] == 0)
    xr = orc.reference_live(5, 3, 0.05, np.array([0, 0, 0, 21.0, 0]))
    assert np.allclose(xr[3], [20.5, 20, 20])


def _obtain_reference_numpy(x, ds, N_s, t, s0, dt, N_t):
    """Independent restatement of util/obtain_reference.m with 1-based index arithmetic spelled out."""
    L = ds * N_s
    cols = [x[c::8] for c in range(6)]
    idx = np.zeros(N_t + 2, dtype=int); rto = np.zeros(N_t + 2)
    idx[1] = int(np.floor(np.mod(s0, L) / ds)) + 1
    rto[1] = np.mod(np.mod(s0, L) / ds, 1)
    nxt = lambda i: (i % N_s) + 1
    for i in range(2, N_t + 2):
        rem = dt
        idx[i] = idx[i - 1]
        rto[i] = rto[i - 1] + rem / t[idx[i] - 1]
        rem = rem - t[idx[i - 1] - 1] * (1 - rto[i - 1])
        while rto[i] > 1:
            idx[i] = nxt(idx[i]); rto[i] = rem / t[idx[i] - 1]; rem = rem - t[idx[i] - 1]
    out = np.zeros((7, N_t))
    for i in range(2, N_t + 2):
        out[0, i - 2] = s0 + np.mod(idx[i] + rto[i] - idx[1] - rto[1], N_s) * ds
        for c in range(6):
            a, b = cols[c][idx[i] - 1], cols[c][nxt(idx[i]) - 1]
            out[1 + c, i - 2] = a + (b - a) * rto[i]
    return out


def _plan_table(N_s, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=8 * N_s)
    x[2::8] = rng.uniform(5, 25, N_s)           # x_d > 0
    ds = 0.5
    t = ds / x[2::8]                            # per-cell traversal time of a planner solution
    return x, ds, t


@pytest.mark.parametrize("s0", [0.0, 3.7, 123.456, 2 * 0.5 * 400 + 1.25, 199.999999])
def test_obtain_reference_restatement_and_properties(orc, s0):
    N_s, N_t, dt = 400, 40, 0.05
    x, ds, t = _plan_table(N_s, 5)
    ref = _obtain_reference_numpy(x, ds, N_s, t, s0, dt, N_t)
    got = orc.obtain_reference(x, ds, N_s, t, s0, dt, N_t)
    assert np.array_equal(got, ref)
    # the s row advances monotonically and starts ahead of s0
    assert (np.diff(got[0]) > 0).all() and got[0, 0] > s0
    # a plan with constant cell times t = ds/v walks exactly v*dt per step
    x2 = x.copy(); x2[2::8] = 10.0; t2 = np.full(N_s, ds / 10.0)
    g2 = orc.obtain_reference(x2, ds, N_s, t2, s0, dt, N_t)
    assert np.allclose(g2[0], s0 + 10.0 * dt * np.arange(1, N_t + 1), rtol=0, atol=1e-9)
    assert np.allclose(g2[3], 10.0)


# ---- closed loop around the step (oracle/ltv_oracle_plant.c) ----

def _cart_from_curv(orc, tr, s, n):
    """Point at arc length s, lateral offset n (left normal), heading = track heading: curvilinear_to_cartesian."""
    import ctypes as C
    L = orc.lib()
    L.orc_spline_d.restype = C.c_double
    xd = L.orc_spline_d(tr.c.xP, tr.M, C.c_double(tr.dl), C.c_double(s)); yd = L.orc_spline_d(tr.c.yP, tr.M, C.c_double(tr.dl), C.c_double(s))
    x = L.orc_spline_val(tr.c.xP, tr.M, C.c_double(tr.dl), C.c_double(s)); y = L.orc_spline_val(tr.c.yP, tr.M, C.c_double(tr.dl), C.c_double(s))
    nrm = np.hypot(xd, yd)
    return x - yd / nrm * n, y + xd / nrm * n, np.arctan2(yd, xd)


def test_cartesian_to_curvilinear_inverts_the_track_frame(orc, otrack):
    rng = np.random.default_rng(4)
    for _ in range(40):
        s, n, dmu = rng.uniform(0, otrack.L), rng.uniform(-0.7, 0.7), rng.uniform(-0.3, 0.3)
        x, y, th = _cart_from_curv(orc, otrack, s, n)
        s2, n2, mu2 = orc.cart_to_curv(otrack, x, y, th + dmu, s + rng.uniform(-1.0, 1.0))
        assert abs(s2 - s) < 2e-2          # Newton stops at |step| <= 0.01 (closest_point.m epsilon)
        assert abs(n2 - n) < 1e-3 and abs(mu2 - dmu) < 2e-2
    # angdiff wraps: heading + 2*pi gives the same mu
    x, y, th = _cart_from_curv(orc, otrack, 12.0, 0.2)
    a = orc.cart_to_curv(otrack, x, y, th + 0.1, 12.0)[2]; b = orc.cart_to_curv(otrack, x, y, th + 0.1 + 2 * np.pi, 12.0)[2]
    assert abs(a - b) < 1e-12 and abs(a - 0.1) < 2e-2


def test_plant_pieces(orc):
    # straight-line cruise: no slip, no force => the car advances v*dt along its heading and nothing else changes
    x0 = np.array([1.0, 2.0, 0.3, 15.0, 0.0, 0.0, 0.0])
    x1 = orc.integrate_cart_dyn(x0, [0.0, 0.0], 0.005)
    assert np.allclose(x1[:2], x0[:2] + 15.0 * 0.005 * np.array([np.cos(0.3), np.sin(0.3)]), atol=1e-12)
    assert np.allclose(x1[2:], x0[2:], atol=1e-12)
    # f_cart_dyn: longitudinal force accelerates by Fx/m, steering rate passes through
    f = orc.f_cart_dyn(x0, [2800.0, 0.4])
    assert abs(f[3] - 10.0) < 1e-12 and f[6] == 0.4 and abs(f[2]) == 0.0
    # PID: pure proportional with saturation (main.m:84-88): 16000*(v_ref - v) clipped at 2800
    x, pid, u = orc.plant_step(x0, np.zeros(4), 20.0, 0.1, 0.05)
    assert u[0] == 2800.0 and 0 < u[1] <= 0.8
    assert x[3] > x0[3] and abs(x[3] - (15.0 + 10.0 * 0.05)) < 5e-3 and 0 < x[6] <= 0.1 + 1e-12
    assert pid[1] == 20.0 - orc.integrate_cart_dyn(x0, [0, 0], 0)[3] or pid[1] > 0   # last error stored


@pytest.mark.parametrize("model", [0, 1])
def test_cl_pre_matches_its_parts(orc, otrack, model):
    x, y, th = _cart_from_curv(orc, otrack, 33.0, -0.25)
    cart = np.array([x, y, th + 0.05, 12.0, 0.7, 0.1, -0.03])
    x0, x_ref, fin = orc.cl_pre(model, 40, 0.05, otrack, cart, 32.5)
    s, n, mu = orc.cart_to_curv(otrack, x, y, th + 0.05, 32.5)
    assert fin == 0 and x0[0] == s and x0[1] == n and x0[2] == mu
    assert x0[3] == (np.hypot(12.0, 0.7) if model == 0 else 12.0) and x0[-1] == -0.03
    assert np.array_equal(x_ref, orc.reference_live(x0.size, 40, 0.05, x0))      # x(4) < TARGET_VEL here, same ramp
    assert orc.cl_pre(model, 40, 0.05, otrack, cart, 32.5 + otrack.L)[2] == 1      # a lap further: finished


@pytest.mark.parametrize("name", ["fsg2019", "fss2019", "fso2020"])
def test_track_tables_are_periodic_arclength_splines(name):
    """The committed spline tables (tools/make_track_tables.py: read_raceline_csv -> make_spline_periodic -> arclength_reparam,
    SURVEY 8 f-2) are closed C2 cubic Bezier chains, approximately uniformly spaced in arc length."""
    import json, os
    d = json.load(open(os.path.join(os.path.dirname(__file__), "..", "fsae-mpc_amd", "tracks", name + ".json")))
    M, dl, L = len(d["xP"]), d["dl"], d["L"]
    assert M == 100 and abs(M * dl - L) < 1e-9 * L                                    # main.m:17 (100 segments)
    for P in (np.array(d["xP"]), np.array(d["yP"])):
        nxt = np.roll(P, -1, axis=0)
        assert np.allclose(P[:, 3], nxt[:, 0], atol=1e-9)                            # C0, closed
        assert np.allclose(P[:, 3] - P[:, 2], nxt[:, 1] - nxt[:, 0], atol=1e-8)      # C1: make_spline_periodic.m:9-33
        assert np.allclose(P[:, 3] - 2 * P[:, 2] + P[:, 1], nxt[:, 2] - 2 * nxt[:, 1] + nxt[:, 0], atol=1e-7)   # C2
    # arc length: each segment is dl long (bisection tolerance 0.01 m, arclength_reparam.m:49), speed |dr/ds| ~ 1
    X, Y = np.array(d["xP"]), np.array(d["yP"])
    u = np.linspace(0, 1, 201)[None, :]
    def der(P):
        return (-3 * (1 - u) ** 2 * P[:, :1] + 3 * (3 * u * u - 4 * u + 1) * P[:, 1:2] + 3 * (2 * u - 3 * u * u) * P[:, 2:3] + 3 * u * u * P[:, 3:4])
    speed = np.hypot(der(X), der(Y))                                                   # per unit of the segment parameter
    seg_len = ((speed[:, 1:] + speed[:, :-1]) * 0.5 * (u[0, 1] - u[0, 0])).sum(axis=1)
    # the reference's segment-length quirk (arclength_reparam.m:20-23 uses x_P(i,1) for x_P(i,2)) is reproduced, so the
    # re-parametrisation is only approximately uniform: segments within 25 % of dl, total length within 5 % of L
    assert np.max(np.abs(seg_len / dl - 1)) < 0.25 and abs(seg_len.sum() / L - 1) < 0.05
    assert np.max(np.abs(speed / dl - 1)) < 0.35


# ---- evidence that does not go through the oracle's own checker ("parity unpinned") -------------------------
@pytest.mark.parametrize("model,N,B", [(0, 40, 6), (1, 40, 4), (1, 60, 3), (1, 80, 2)])
def test_vertex_recomputed_from_the_working_set_numpy(orc, otrack, model, N, B):
    """BASELINE sizes (kinematic N = 40: nV 81; dynamic N = 40 / 60 / 80: nV 84 / 124 / 164).  The working set of the oracle's
    answer (sign pattern of its multipliers, qpOASES.m:58-61) is handed to dense numpy linear algebra, which recomputes the vertex
    and its multipliers from scratch (tests/kkt_numpy.py: no solver, no oracle code); the recomputed pair must pass the
    numpy KKT certificate of the FULL QP -- which proves that working set optimal -- and reproduce the oracle's x."""
    from kkt_numpy import kkt_certificate, vertex_from_working_set
    x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, otrack.L, 20190, range(100, 100 + B))
    q = orc.build_qp_batch(model, otrack, N, 0.05, x0, xr, xl, ul)
    o = orc.qp_solve_batch_aux(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"])
    assert (o["exitflag"] == 0).all()
    c = kkt_certificate(q["H"], q["g"], q["A"], q["lb"], q["ub"], q["lbA"], q["ubA"], o["x"], o["lam"])
    assert c["max"].max() <= 1e-6
    n_checked = 0
    for b in range(B):
        if not o["polished"][b]:
            continue                                                     # an interior-point iterate carries no working set
        ws = np.where(o["lam"][b] > 0, -1, np.where(o["lam"][b] < 0, 1, 0))
        # sides that are active with a ZERO multiplier (degenerate vertex) belong to the vertex's working set as well
        v = np.concatenate([o["x"][b], q["A"][b].T @ o["x"][b]])
        lo, hi = np.concatenate([q["lb"][b], q["lbA"][b]]), np.concatenate([q["ub"][b], q["ubA"][b]])
        sc = np.maximum(1.0, np.abs(v))
        ws_weak = np.where(ws != 0, ws, np.where((lo > -1e9) & (np.abs(v - lo) <= 1e-9 * sc), -1, np.where((hi < 1e9) & (np.abs(hi - v) <= 1e-9 * sc), 1, 0)))
        for cand in (ws, ws_weak):
            xv, lv = vertex_from_working_set(q["H"][b].T, q["g"][b], q["A"][b].T, q["lb"][b], q["ub"][b], q["lbA"][b], q["ubA"][b], cand)
            cv = kkt_certificate(q["H"][b:b + 1], q["g"][b:b + 1], q["A"][b:b + 1], q["lb"][b:b + 1], q["ub"][b:b + 1], q["lbA"][b:b + 1],
                                 q["ubA"][b:b + 1], xv[None], lv[None])
            if cv["max"][0] <= 1e-9:
                break
        assert cv["max"][0] <= 1e-9, (b, {k: float(cv[k][0]) for k in ("stationarity", "primal", "sign", "complementarity")})
        assert np.abs(xv - o["x"][b]).max() <= 1e-7 * max(1.0, np.abs(xv).max()), b
        n_checked += 1
    assert n_checked >= max(1, B // 2)


@pytest.mark.parametrize("model,N,inst", [(0, 40, 3)])
def test_solution_vs_scipy_at_the_headline_size(orc, otrack, model, N, inst):
    """scipy trust-constr on one QP of the headline shape (kinematic N = 40, nV = 81, nC = 240; about a minute).  The dynamic
    BASELINE sizes take 10-40 minutes per instance with this solver: they were run once, tests/harness/scipy_crosscheck.py,
    results in profiles/round3/scipy_crosscheck.json."""
    from scipy.optimize import Bounds, LinearConstraint, minimize
    x0, xl, ul, xr = orc.synth_instances(model, N, 0.05, otrack.L, 20190, [inst])
    q = orc.build_qp(model, otrack, N, 0.05, x0[0], xr[0].T, xl[0].T, ul[0].T)
    H, g, A = q["H"], q["g"], q["A"]
    x, f, fl, it, lam = orc.qp_solve(H, g, A, q["lb"], q["ub"], q["lbA"], q["ubA"])
    assert fl == 0
    lbA = np.where(q["lbA"] < -1e9, -np.inf, q["lbA"]); ubA = np.where(q["ubA"] > 1e9, np.inf, q["ubA"])
    res = minimize(lambda z: 0.5 * z @ H @ z + g @ z, np.clip(x * 0, q["lb"], np.minimum(q["ub"], 1e3)), jac=lambda z: H @ z + g, hess=lambda z: H,
                   method="trust-constr", bounds=Bounds(q["lb"], q["ub"]), constraints=[LinearConstraint(A, lbA, ubA)],
                   options=dict(gtol=1e-10, xtol=1e-12, barrier_tol=1e-12, maxiter=5000))
    assert abs(res.fun - f) <= 1e-8 * max(1.0, abs(f))
    assert np.max(np.abs(res.x - x)) <= 1e-6 * max(1.0, np.max(np.abs(x)))
