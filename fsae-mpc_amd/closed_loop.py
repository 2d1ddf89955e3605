"""Host-side mirror of the reference's closed loop (main.m:91-179) for a batch of independent cars: frame transform +
reference -> LTV-MPC step -> actuator/plant sub-steps, every stage a device kernel behind the C ABI."""
import ctypes as C

import numpy as np

from ._lib import Spline, check, lib
from .ltvmpc import LtvBatch, dims


class ClosedLoop:
    """B cars on one track.  cart0: (B, 7) Cartesian states [x, y, theta, x_d, y_d, theta_d, delta] (main.m:63 starts
    from zeros(7,1)).  step() advances every car by one MPC period dt, all on the device and without reading anything
    back; like the reference (main.m:122-126, 163-175) a car keeps driving after an abnormal solver exit -- on its last good
    plan -- and the exit flags are only tallied.  Cars that completed the lap (s >= L) or left the track keep their state."""

    def __init__(self, model, N, dt, track, cart0, target_vel=20.0, device="cuda:0", options=None, integrator=-1, warm_start=False, launch_hint=True):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.model, self.N, self.dt, self.target_vel = model, N, float(dt), float(target_vel)
        self.nx = dims(model, N)[0]
        cart0 = np.ascontiguousarray(np.asarray(cart0, dtype=np.float64).reshape(-1, 7))
        self.B = cart0.shape[0]
        self.track = track
        self.mpc = LtvBatch(model, N, dt, track, self.B, device=device, options=options, integrator=integrator)
        self.cart = torch.from_numpy(cart0).to(self.device)
        self.pid = torch.zeros((self.B, 4), dtype=torch.float64, device=self.device)
        self.finished = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        # MPC initial guess of main.m:47-55: quadratic arc length, linear velocity, constant acceleration 10
        k = np.arange(1, N + 1, dtype=np.float64) * self.dt
        x_opt = np.zeros((self.B, N, self.nx)); u_opt = np.zeros((self.B, N, 2))
        x_opt[:, :, 0] = 10 * k ** 2 / 2
        x_opt[:, :, 3] = 10 * k
        u_opt[:, :, 0] = 10
        self.x_opt = torch.from_numpy(x_opt).to(self.device)
        self.u_opt = torch.from_numpy(u_opt).to(self.device)
        self.x0 = torch.empty((self.B, self.nx), dtype=torch.float64, device=self.device)
        self.x_ref = torch.empty((self.B, N, self.nx), dtype=torch.float64, device=self.device)
        self.u_last = torch.zeros((self.B, 2), dtype=torch.float64, device=self.device)
        self.steps = 0
        # warm_start: every solve starts from the plan of the previous period shifted by one stage (fsaempc_qp_aux.x_init): the
        # inputs u_1..u_{N-1}, the last stage repeated, the previous slack values.  Off by default -- the reference's loop solves
        # cold (main.m:117-120 passes no initial guess) and the gain is modest (profiles/round3/warm_start_ab.json)
        self.warm_start = bool(warm_start)
        self._slack = None
        # launch_hint: the iteration count of each car's previous QP is handed to the solve as the effort estimate behind its launch
        # order (fsaempc_qp_aux.difficulty: a far better predictor than the library's cold estimate); results do not depend on it
        self.launch_hint = bool(launch_hint)
        self._last_iter = None
        self._x_init = torch.zeros((self.B, self.mpc.nV), dtype=torch.float64, device=self.device) if self.warm_start else None

    def _stream(self, stream):
        return C.c_void_p(stream if stream is not None else self.torch.cuda.current_stream(self.device).cuda_stream)

    def pre(self, stream=None):
        P = lambda t: C.c_void_p(t.data_ptr())
        s_guess = self.x_opt[:, 0, 0].contiguous()
        rc = lib().fsaempc_cl_pre_batch_device(self.model, self.N, C.c_double(self.dt), C.c_double(self.target_vel), C.c_double(self.track.L),
                                               C.byref(self.mpc.sp), P(self.cart), P(s_guess), self.B, P(self.x0), P(self.x_ref),
                                               P(self.finished), self._stream(stream))
        check(rc, "fsaempc_cl_pre_batch_device")

    def plant(self, exitflag, stream=None):
        P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        rc = lib().fsaempc_cl_plant_batch_device(self.model, self.N, C.c_double(self.dt), self.B, P(self.cart), P(self.pid), P(self.x_opt),
                                                 P(self.finished), P(exitflag), P(self.u_last), self._stream(stream))
        check(rc, "fsaempc_cl_plant_batch_device")

    def step(self, stream=None):
        torch = self.torch
        self.pre(stream)
        x_init = None
        if self.warm_start and self.steps > 0:
            N = self.N
            x_init = self._x_init
            x_init[:, : 2 * (N - 1)].view(self.B, N - 1, 2).copy_(self.u_opt[:, 1:, :])   # (no temporaries: strided views on both sides)
            x_init[:, 2 * (N - 1): 2 * N].copy_(self.u_opt[:, N - 1, :])
            if self._slack is not None:
                x_init[:, 2 * N:].copy_(self._slack)
        out = self.mpc.step(self.x0, self.x_ref, self.x_opt, self.u_opt, stream=stream, x_init=x_init,
                            difficulty=self._last_iter if self.launch_hint else None)   # linearised about the previous plan (main.m:121-125)
        if self.warm_start:
            self._slack = out["slack"]
        self._last_iter = out["iter"]
        P = lambda t: C.c_void_p(t.data_ptr())
        check(lib().fsaempc_cl_accept_batch_device(self.model, self.N, self.B, P(out["x_opt"]), P(out["u_opt"]), P(out["exitflag"]), P(self.x_opt), P(self.u_opt),
                                                   self._stream(stream)), "fsaempc_cl_accept_batch_device")
        self.plant(None, stream)
        self.steps += 1
        return out


def monte_carlo_carts(track, B, seed):
    """Initial Cartesian states of BASELINE configs[3]: s0 = u*L, lateral +-0.5 m, heading +-0.1 rad,
    speed U[0,15]; numpy PCG64(seed).  Returns (cart (B,7), s0 (B,))."""
    rng = np.random.default_rng(seed)
    s = rng.uniform(0, track.L, B); n = rng.uniform(-0.5, 0.5, B); dth = rng.uniform(-0.1, 0.1, B); v = rng.uniform(0, 15, B)

    def ev(P, t):   # the Bezier table on the host (same formulas as interpolate_spline / interpolate_spline_d); P is M x 4
        r = np.mod(t, track.dl * track.M); i = np.minimum(np.floor(r / track.dl).astype(int), track.M - 1); u = r / track.dl - i; w = 1 - u
        val = P[i, 0] * w ** 3 + 3 * P[i, 1] * w * w * u + 3 * P[i, 2] * w * u * u + P[i, 3] * u ** 3
        d = (-3 * w * w * P[i, 0] + 3 * (3 * u * u - 4 * u + 1) * P[i, 1] + 3 * (2 * u - 3 * u * u) * P[i, 2] + 3 * u * u * P[i, 3]) / track.dl
        return val, d
    x, xd = ev(track.xP, s); y, yd = ev(track.yP, s)
    nrm = np.hypot(xd, yd)
    cart = np.zeros((B, 7))
    cart[:, 0] = x - yd / nrm * n; cart[:, 1] = y + xd / nrm * n; cart[:, 2] = np.arctan2(yd, xd) + dth; cart[:, 3] = v
    return cart, s


def monte_carlo(model, N, track, B, steps, seed=20190, options=None, device="cuda:0", warm_start=False, launch_hint=True):
    """Closed-loop Monte-Carlo: B cars from random initial states, `steps` receding-horizon steps, device-resident loop.
    Returns the ClosedLoop and the per-step tallies (exit flags, iteration counts, driving mask), read back once at the end --
    the reference reports exactly this tally as "abnormal exits %" (main.m:209,222)."""
    import torch
    cart0, s_init = monte_carlo_carts(track, B, seed)
    cl = ClosedLoop(model, N, 0.05, track, cart0, options=options, device=device, warm_start=warm_start, launch_hint=launch_hint)
    cl.x_opt[:, :, 0] += torch.from_numpy(s_init).to(cl.device)[:, None]        # start the closest-point search near the car
    cl.x_opt[:, :, 3] += torch.from_numpy(cart0[:, 3]).to(cl.device)[:, None]   # and the first linearisation at its speed
    flags = torch.zeros((steps, B), dtype=torch.int32, device=cl.device)
    iters = torch.zeros((steps, B), dtype=torch.int32, device=cl.device)
    active = torch.zeros((steps, B), dtype=torch.bool, device=cl.device)
    for t in range(steps):
        out = cl.step()
        flags[t] = out["exitflag"]; iters[t] = out["iter"]; active[t] = cl.finished == 0
    torch.cuda.synchronize(cl.device)
    return cl, flags.cpu().numpy(), iters.cpu().numpy(), active.cpu().numpy()
