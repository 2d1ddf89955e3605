"""CPU ORACLE -- test infrastructure, NOT the product.

ctypes binding of oracle/libltv_oracle.so (plain-C restatement of the reference's LTV-MPC hot
path, see ltv_oracle.h).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package.  PARITY UNPINNED (no golden vectors in the reference; qpOASES absent).
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KINEMATIC, DYNAMIC = 0, 1
EULER, RK2, RK4 = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class Spline(C.Structure):
    _fields_ = [("M", C.c_int), ("dl", C.c_double), ("xP", _dp), ("yP", _dp)]


class QpOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("tol_loose", C.c_double), ("tol_x", C.c_double), ("max_iter", C.c_int), ("inf_bound", C.c_double),
                ("polish", C.c_int), ("corrector", C.c_int), ("scale", C.c_int), ("verbose", C.c_int)]


def build(force=False):
    if os.environ.get("ORACLE_LIB"):          # `make -C oracle asan-test`: the sanitizer build of the same sources
        return os.environ["ORACLE_LIB"]
    so = os.path.join(_HERE, "libltv_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "clean", "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_kappa.restype = C.c_double
        _LIB.orc_spline_val.restype = C.c_double
        _LIB.orc_qp_kkt.restype = C.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Track:
    """Spline table holder (keeps the column-major buffers alive)."""

    def __init__(self, xP, yP, dl, L, name="track"):
        self.name = name
        self.M = int(np.shape(xP)[0])
        self.dl = float(dl)
        self.L = float(L)
        self.xP = np.asfortranarray(np.asarray(xP, dtype=np.float64))  # M x 4 column-major
        self.yP = np.asfortranarray(np.asarray(yP, dtype=np.float64))
        self.c = Spline(self.M, self.dl, self.xP.ctypes.data_as(_dp), self.yP.ctypes.data_as(_dp))

    @staticmethod
    def load(path):
        with open(path) as f:
            d = json.load(f)
        return Track(d["xP"], d["yP"], d["dl"], d["L"], d.get("name", "track"))


def default_opts(**kw):
    o = QpOpts()
    lib().orc_qp_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def dims(model, N):
    L = lib()
    return L.orc_nx(model), L.orc_ns(model), L.orc_nV(model, N), L.orc_nC(model, N)


def kappa(track, s):
    return lib().orc_kappa(C.byref(track.c), C.c_double(s))


def f_model(model, track, x, u):
    nx = dims(model, 1)[0]
    x, u = _f(x), _f(u)
    f = np.zeros(nx)
    if model == KINEMATIC:
        lib().orc_f_kin(_p(x), _p(u), C.byref(track.c), _p(f))
    else:
        lib().orc_f_dyn(_p(x), _p(u), C.byref(track.c), _p(f), None)
    return f


def A_model(model, track, x):
    nx = dims(model, 1)[0]
    x = _f(x)
    A = np.zeros((nx, nx), order="F")
    byp = np.zeros(8)
    if model == KINEMATIC:
        lib().orc_A_kin(_p(x), C.byref(track.c), _p(A))
    else:
        lib().orc_A_dyn(_p(x), C.byref(track.c), _p(A), _p(byp))
    return A, byp


def linearise(model, integrator, track, x_lin, u_lin, dt):
    """x_lin nx x N, u_lin 2 x N (numpy, any order) -> A (nx,nx,N), B (nx,2,N), d (nx,N), Fortran order."""
    x = np.asfortranarray(x_lin, dtype=np.float64)
    u = np.asfortranarray(u_lin, dtype=np.float64)
    nx, N = x.shape
    A = np.zeros((nx, nx, N), order="F")
    B = np.zeros((nx, 2, N), order="F")
    d = np.zeros((nx, N), order="F")
    lib().orc_linearise(model, integrator, N, _p(x), _p(u), C.byref(track.c), C.c_double(dt), _p(A), _p(B), _p(d))
    return A, B, d


def sequential_integration(A, B, d, dt):
    nx, _, N = A.shape
    A, B, d = (np.asfortranarray(a, dtype=np.float64) for a in (A, B, d))
    A_bar = np.zeros((nx * N, nx), order="F")
    B_bar = np.zeros((nx * N, 2 * N), order="F")
    d_bar = np.zeros(nx * N)
    lib().orc_sequential_integration(nx, N, _p(A), _p(B), _p(d), C.c_double(dt), _p(A_bar), _p(B_bar), _p(d_bar))
    return A_bar, B_bar, d_bar


def build_qp(model, track, N, dt, x0, x_ref, x_lin, u_lin, integrator=-1):
    """One QP of the LTV-MPC step. x_ref/x_lin: nx x N, u_lin: 2 x N. Returns dict of Fortran-ordered arrays."""
    nx, ns, nV, nC = dims(model, N)
    x0 = _f(x0)
    x_ref, x_lin, u_lin = (np.asfortranarray(a, dtype=np.float64) for a in (x_ref, x_lin, u_lin))
    out = dict(H=np.zeros((nV, nV), order="F"), g=np.zeros(nV), A=np.zeros((nC, nV), order="F"),
               lb=np.zeros(nV), ub=np.zeros(nV), lbA=np.zeros(nC), ubA=np.zeros(nC),
               A_bar=np.zeros((nx * N, nx), order="F"), Bt=np.zeros((nx * N, nV), order="F"), d_bar=np.zeros(nx * N))
    qc = C.c_double(0)
    lib().orc_ltv_build_qp(model, integrator, N, C.c_double(dt), C.byref(track.c), _p(x0), _p(x_ref), _p(x_lin), _p(u_lin),
                           _p(out["H"]), _p(out["g"]), _p(out["A"]), _p(out["lb"]), _p(out["ub"]), _p(out["lbA"]),
                           _p(out["ubA"]), _p(out["A_bar"]), _p(out["Bt"]), _p(out["d_bar"]), C.byref(qc))
    out["const"] = qc.value
    return out


def qp_solve(H, g, A, lb, ub, lbA, ubA, opts=None):
    """Oracle twin of qpOASES(H,g,A,lb,ub,lbA,ubA): returns x, fval, exitflag, iter, lambda."""
    H = np.asfortranarray(H, dtype=np.float64)
    nV = H.shape[0]
    A = np.asfortranarray(A, dtype=np.float64).reshape((-1, nV), order="F") if np.size(A) else np.zeros((0, nV), order="F")
    nC = A.shape[0]
    g, lb, ub, lbA, ubA = (_f(a).ravel() for a in (g, lb, ub, lbA, ubA))
    x = np.zeros(nV)
    lam = np.zeros(nV + nC)
    fval = C.c_double(0)
    it = C.c_int(0)
    o = opts if opts is not None else default_opts()
    flag = lib().orc_qp_solve(nV, nC, _p(H), _p(g), _p(A), _p(lb), _p(ub), _p(lbA), _p(ubA), C.byref(o), _p(x),
                              C.byref(fval), C.byref(it), _p(lam))
    return x, fval.value, flag, it.value, lam


def qp_kkt(H, g, A, lb, ub, lbA, ubA, x, lam, inf_bound=1e9):
    H = np.asfortranarray(H, dtype=np.float64)
    nV = H.shape[0]
    A = np.asfortranarray(A, dtype=np.float64).reshape((-1, nV), order="F") if np.size(A) else np.zeros((0, nV), order="F")
    nC = A.shape[0]
    g, lb, ub, lbA, ubA, x, lam = (_f(a).ravel() for a in (g, lb, ub, lbA, ubA, x, lam))
    res = np.zeros(4)
    r = lib().orc_qp_kkt(nV, nC, _p(H), _p(g), _p(A), _p(lb), _p(ub), _p(lbA), _p(ubA), _p(x), _p(lam),
                         C.c_double(inf_bound), _p(res))
    return r, res


def ltv_step(model, track, N, dt, x0, x_ref, x_lin, u_lin, opts=None):
    nx, ns, nV, nC = dims(model, N)
    x0 = _f(x0)
    x_ref, x_lin, u_lin = (np.asfortranarray(a, dtype=np.float64) for a in (x_ref, x_lin, u_lin))
    u_opt = np.zeros(2 * N)
    x_opt = np.zeros(nx * N)
    slack = np.zeros(ns)
    fval = C.c_double(0)
    it = C.c_int(0)
    o = opts if opts is not None else default_opts()
    flag = lib().orc_ltv_step(model, N, C.c_double(dt), C.byref(track.c), _p(x0), _p(x_ref), _p(x_lin), _p(u_lin),
                              C.byref(o), _p(u_opt), _p(x_opt), _p(slack), C.byref(fval), C.byref(it))
    return u_opt, x_opt, slack, fval.value, flag, it.value


def reference_live(nx, N, dt, x0, target_vel=20.0):
    x0 = _f(x0)
    x_ref = np.zeros((nx, N), order="F")
    lib().orc_reference_live(nx, N, C.c_double(dt), C.c_double(target_vel), _p(x0), _p(x_ref))
    return x_ref


def cart_to_curv(track, x, y, theta, s0):
    s, n, mu = C.c_double(0), C.c_double(0), C.c_double(0)
    lib().orc_cart_to_curv(C.byref(track.c), C.c_double(x), C.c_double(y), C.c_double(theta), C.c_double(s0), C.byref(s), C.byref(n), C.byref(mu))
    return s.value, n.value, mu.value


def f_cart_dyn(x, u):
    f = np.zeros(7)
    lib().orc_f_cart_dyn(_p(_f(x)), _p(_f(u)), _p(f))
    return f


def integrate_cart_dyn(x0, u, dt):
    x = np.zeros(7)
    lib().orc_integrate_cart_dyn(_p(_f(x0)), _p(_f(u)), C.c_double(dt), _p(x))
    return x


def plant_step(x, pid, v_ref, delta_ref, dt):
    """main.m:171-175.  Returns new (x, pid, u_last)."""
    x, pid, u = _f(x).copy(), _f(pid).copy(), np.zeros(2)
    lib().orc_plant_step(_p(x), _p(pid), C.c_double(v_ref), C.c_double(delta_ref), C.c_double(dt), _p(u))
    return x, pid, u


def cl_pre(model, N, dt, track, cart, s_guess, target_vel=20.0):
    """main.m:93-114 -> x0 (nx,), x_ref (nx, N), finished."""
    nx = dims(model, N)[0]
    x0 = np.zeros(nx); x_ref = np.zeros((nx, N), order="F")
    fin = lib().orc_cl_pre(model, N, C.c_double(dt), C.c_double(target_vel), C.byref(track.c), C.c_double(track.L), _p(_f(cart)), C.c_double(s_guess),
                           _p(x0), _p(x_ref))
    return x0, x_ref, fin


def obtain_reference(x, ds, N_s, t, s0, dt, N_t):
    """util/obtain_reference.m: planner vector x (8*N_s), per-cell times t (N_s) -> x_ref (7, N_t)."""
    x, t = _f(x).ravel(), _f(t).ravel()
    assert x.size == 8 * N_s and t.size == N_s
    x_ref = np.zeros((7, N_t), order="F")
    lib().orc_obtain_reference(_p(x), C.c_double(ds), N_s, _p(t), C.c_double(s0), C.c_double(dt), N_t, _p(x_ref))
    return x_ref


def synth_instances(model, N, dt, L, seed, ids):
    """SURVEY 8(d) synthetic instances -> x0 (B,nx), x_lin (B,nx,N)F per instance, u_lin, x_ref (batch-major)."""
    nx = dims(model, N)[0]
    ids = np.asarray(ids, dtype=np.uint64)
    B = len(ids)
    x0 = np.zeros((B, nx))
    x_lin = np.zeros((B, N, nx))   # memory = per instance nx x N column-major
    u_lin = np.zeros((B, N, 2))
    x_ref = np.zeros((B, N, nx))
    for b, i in enumerate(ids):
        lib().orc_synth_instance(model, N, C.c_double(dt), C.c_double(L), C.c_ulonglong(seed), C.c_ulonglong(int(i)),
                                 _p(x0[b]), _p(x_lin[b]), _p(u_lin[b]), _p(x_ref[b]))
    return x0, x_lin, u_lin, x_ref


def build_qp_batch(model, track, N, dt, x0, x_ref, x_lin, u_lin, threads=0, keep_prediction=False):
    """Batch-major stacked QPs (each QP column-major).  Inputs as returned by synth_instances."""
    nx, ns, nV, nC = dims(model, N)
    B = x0.shape[0]
    x0, x_ref, x_lin, u_lin = (_f(a) for a in (x0, x_ref, x_lin, u_lin))
    out = dict(H=np.zeros((B, nV, nV)), g=np.zeros((B, nV)), A=np.zeros((B, nV, nC)), lb=np.zeros((B, nV)),
               ub=np.zeros((B, nV)), lbA=np.zeros((B, nC)), ubA=np.zeros((B, nC)), const=np.zeros(B))
    if keep_prediction:
        out.update(A_bar=np.zeros((B, nx, nx * N)), Bt=np.zeros((B, nV, nx * N)), d_bar=np.zeros((B, nx * N)))
    used = lib().orc_ltv_build_qp_batch(model, N, C.c_double(dt), C.byref(track.c), B, _p(x0), _p(x_ref), _p(x_lin), _p(u_lin),
                                        _p(out["H"]), _p(out["g"]), _p(out["A"]), _p(out["lb"]), _p(out["ub"]), _p(out["lbA"]),
                                        _p(out["ubA"]), _p(out.get("A_bar")), _p(out.get("Bt")), _p(out.get("d_bar")),
                                        _p(out["const"]), threads)
    out["threads"] = used
    return out


def qp_solve_batch_aux(H, g, A, lb, ub, lbA, ubA, opts=None, threads=0):
    """As qp_solve_batch, plus the per-instance relative KKT residual the solver measured and the refinement flag."""
    B, nV = g.shape
    nC = lbA.shape[1]
    H, g, A, lb, ub, lbA, ubA = (_f(a) for a in (H, g, A, lb, ub, lbA, ubA))
    x = np.zeros((B, nV)); fval = np.zeros(B); flag = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
    lam = np.zeros((B, nV + nC)); kkt = np.zeros(B); pol = np.zeros(B, dtype=np.int32)
    o = opts if opts is not None else default_opts()
    lib().orc_qp_solve_batch_ex(nV, nC, B, _p(H), _p(g), _p(A), _p(lb), _p(ub), _p(lbA), _p(ubA), C.byref(o), _p(x), _p(fval),
                                flag.ctypes.data_as(_ip), it.ctypes.data_as(_ip), _p(lam), _p(kkt), pol.ctypes.data_as(_ip), threads)
    return dict(x=x, fval=fval, exitflag=flag, iter=it, lam=lam, kkt=kkt, polished=pol)


def qp_solve_batch(H, g, A, lb, ub, lbA, ubA, opts=None, threads=0, want_lambda=True):
    """H (B,nV,nV), A (B,nV,nC) = per-QP column-major nC x nV, vectors (B,*)."""
    B, nV = g.shape
    nC = lbA.shape[1]
    H, g, A, lb, ub, lbA, ubA = (_f(a) for a in (H, g, A, lb, ub, lbA, ubA))
    x = np.zeros((B, nV))
    fval = np.zeros(B)
    flag = np.zeros(B, dtype=np.int32)
    it = np.zeros(B, dtype=np.int32)
    lam = np.zeros((B, nV + nC)) if want_lambda else None
    o = opts if opts is not None else default_opts()
    used = lib().orc_qp_solve_batch(nV, nC, B, _p(H), _p(g), _p(A), _p(lb), _p(ub), _p(lbA), _p(ubA), C.byref(o), _p(x),
                                    _p(fval), flag.ctypes.data_as(_ip), it.ctypes.data_as(_ip), _p(lam), threads)
    return x, fval, flag, it, lam, used
