"""Synthetic LTV-MPC instances of SURVEY 8(d): splitmix64 keyed by seed ^ (id * 0x9E3779B97F4A7C15),
u01 = (next() >> 11) * 2^-53.  Product-side generator (numpy, vectorised over the batch); the oracle has an
independent C copy and tests/ check the two agree bit for bit."""
import numpy as np

KINEMATIC, DYNAMIC = 0, 1
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _next(state):
    with np.errstate(over="ignore"):
        state += _GOLD
        z = state.copy()
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _u01(state):
    return (_next(state) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def reference_live(x0, N, dt, target_vel=20.0):
    """main.m:107-114: velocity ramp +-10 m/s^2 clipped at TARGET_VEL, s_ref = s0 + cumsum(v_ref*dt).
    x0: (B, nx) -> x_ref (B, N, nx) (per-instance memory = nx x N column-major)."""
    B, nx = x0.shape
    k = np.arange(1, N + 1, dtype=np.float64)[None, :]
    v0 = x0[:, 3:4]
    up = np.minimum(v0 + 10 * dt * k, target_vel)
    dn = np.maximum(v0 - 10 * dt * k, target_vel)
    v = np.where(v0 < target_vel, up, dn)
    x_ref = np.zeros((B, N, nx))
    x_ref[:, :, 3] = v
    x_ref[:, :, 0] = x0[:, 0:1] + np.cumsum(v * dt, axis=1)
    return x_ref


def instances(model, N, dt, L, seed, ids):
    """Returns x0 (B,nx), x_lin (B,N,nx), u_lin (B,N,2), x_ref (B,N,nx); per-instance memory is the
    column-major nx x N / 2 x N array the C ABI expects."""
    ids = np.asarray(ids, dtype=np.uint64)
    nx = 5 if model == KINEMATIC else 7
    with np.errstate(over="ignore"):
        st = np.uint64(seed) ^ (ids * _GOLD)
    s0 = _u01(st) * L
    n0 = -0.5 + _u01(st)
    mu0 = -0.1 + 0.2 * _u01(st)
    v0 = 5 + 15 * _u01(st)
    d0 = -0.1 + 0.2 * _u01(st)
    B = len(ids)
    x0 = np.zeros((B, nx))
    if model == KINEMATIC:
        x0[:, 0], x0[:, 1], x0[:, 2], x0[:, 3], x0[:, 4] = s0, n0, mu0, v0, d0
    else:
        yd = -0.2 + 0.4 * _u01(st)
        td = -0.3 + 0.6 * _u01(st)
        x0[:, 0], x0[:, 1], x0[:, 2], x0[:, 3], x0[:, 4], x0[:, 5], x0[:, 6] = s0, n0, mu0, v0, yd, td, d0
    x_lin = np.repeat(x0[:, None, :], N, axis=1).copy()
    x_lin[:, :, 0] = s0[:, None] + v0[:, None] * dt * np.arange(N)[None, :]
    u_lin = np.zeros((B, N, 2))
    x_ref = reference_live(x0, N, dt, 20.0)
    return x0, x_lin, u_lin, x_ref
