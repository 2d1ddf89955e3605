"""Host-side mirror of the reference's closed loop (main.m:91-179) for a batch of independent cars: frame transform +
reference -> LTV-MPC step -> actuator/plant sub-steps, every stage a device kernel behind the C ABI."""
import ctypes as C

import numpy as np

from ._lib import Spline, check, lib
from .ltvmpc import LtvBatch, dims


class ClosedLoop:
    """B cars on one track.  cart0: (B, 7) Cartesian states [x, y, theta, x_d, y_d, theta_d, delta] (main.m:63 starts
    from zeros(7,1)).  step() advances every car by one MPC period dt; cars that completed the lap (s >= L) or whose
    QP did not solve keep their state for that step."""

    def __init__(self, model, N, dt, track, cart0, target_vel=20.0, device="cuda:0", options=None, integrator=-1):
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
        P = lambda t: C.c_void_p(t.data_ptr())
        rc = lib().fsaempc_cl_plant_batch_device(self.model, self.N, C.c_double(self.dt), self.B, P(self.cart), P(self.pid), P(self.x_opt),
                                                 P(self.finished), P(exitflag), P(self.u_last), self._stream(stream))
        check(rc, "fsaempc_cl_plant_batch_device")

    def step(self, stream=None):
        torch = self.torch
        self.pre(stream)
        out = self.mpc.step(self.x0, self.x_ref, self.x_opt, self.u_opt, stream=stream)   # linearised about the previous plan (main.m:121-125)
        ok = (out["exitflag"] == 0).view(-1, 1, 1)
        self.x_opt = torch.where(ok, out["x_opt"].view(self.B, self.N, self.nx), self.x_opt).contiguous()
        self.u_opt = torch.where(ok, out["u_opt"].view(self.B, self.N, 2), self.u_opt).contiguous()
        self.plant(out["exitflag"], stream)
        self.steps += 1
        return out
