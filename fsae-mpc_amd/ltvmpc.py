"""Host-side mirror of the reference's LTV-MPC step drivers
    [u_opt,x_opt,QP,exitflag,fval,slack_opt] = ltvmpc_kinetmatic_curvilinear(x0,x_ref,kappa,dt,x_lin,u_lin,QP)
    (mpc/ltv/kinematic/ltvmpc_kinetmatic_curvilinear.m:1) and ltvmpc_dynamic_curvilinear
    (mpc/ltv/dynamic/ltvmpc_dynamic_curvilinear.m:1),
with the `kappa` closure replaced by the Track table it closes over (main.m:18).  All compute runs on the
MI355X through libfsaempc.so."""
import ctypes as C

import numpy as np

from ._lib import LtvDesc, QpAux, QpDesc, Spline, check, default_opts, lib
from .synthetic import DYNAMIC, KINEMATIC


def dims(model, N):
    L = lib()
    return L.fsaempc_ltv_nx(model), (1 if model == KINEMATIC else 4), L.fsaempc_ltv_nV(model, N), L.fsaempc_ltv_nC(model, N)


class LtvBatch:
    """Device-resident batched LTV-MPC step (one call = linearise + condense + solve + post-solve for
    `batch` independent instances).  Inputs/outputs are torch tensors on the GPU."""

    def __init__(self, model, N, dt, track, batch, device="cuda:0", options=None, integrator=-1):
        import torch
        self.torch = torch
        self.model, self.N, self.dt, self.batch = model, N, float(dt), batch
        self.device = torch.device(device)
        self.nx, self.ns, self.nV, self.nC = dims(model, N)
        self.track = track
        self.xP, self.yP = track.device(self.device)
        self.sp = Spline(track.M, track.dl, C.c_void_p(self.xP.data_ptr()), C.c_void_p(self.yP.data_ptr()))
        self.desc = LtvDesc(model, N, batch, self.dt, integrator)   # integrator: -1 reference default, 0 Euler, 1 RK2, 2 RK4
        self.opts = options if options is not None else default_opts()
        self._ws = None
        self._qp = None

    def _f64(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.float64, device=self.device)

    def _stream(self, stream):
        return C.c_void_p(stream if stream is not None else self.torch.cuda.current_stream(self.device).cuda_stream)

    def build_qp(self, x0, x_ref, x_lin, u_lin, stream=None):
        """QP tensors of ltvmpc_*.m:38-41 (H,g,A,lb,ub,lbA,ubA, pred = A_bar*x0+d_bar, Bt, const)."""
        B, nx, N, nV, nC = self.batch, self.nx, self.N, self.nV, self.nC
        q = dict(H=self._f64(B, nV, nV), g=self._f64(B, nV), A=self._f64(B, nV, nC), lb=self._f64(B, nV), ub=self._f64(B, nV),
                 lbA=self._f64(B, nC), ubA=self._f64(B, nC), pred=self._f64(B, N * nx), Bt=self._f64(B, nV, N * nx),
                 const=self._f64(B))
        P = lambda t: C.c_void_p(t.data_ptr())
        rc = lib().fsaempc_ltv_build_qp_batch_device(C.byref(self.desc), C.byref(self.sp), P(x0), P(x_ref), P(x_lin), P(u_lin),
                                                     P(q["H"]), P(q["g"]), P(q["A"]), P(q["lb"]), P(q["ub"]), P(q["lbA"]), P(q["ubA"]),
                                                     P(q["pred"]), P(q["Bt"]), P(q["const"]), self._stream(stream))
        check(rc, "fsaempc_ltv_build_qp_batch_device")
        return q

    def step(self, x0, x_ref, x_lin, u_lin, stream=None, want_aux=False, x_init=None, difficulty=None):
        """Fused step.  Returns dict(u_opt (B,2N), x_opt (B,nx*N), slack (B,ns), fval, exitflag, iter); want_aux adds the solve's
        per-instance diagnostics `kkt` (achieved relative KKT residual) and `polished` (> 0: the returned point is the vertex); x_init
        (batch, nV): optional starting point of the interior-point solve (fsaempc_qp_aux.x_init); difficulty (batch,) int32: optional
        effort estimate per instance for the launch order (fsaempc_qp_aux.difficulty)."""
        torch = self.torch
        B = self.batch
        need = lib().fsaempc_ltv_workspace_bytes(C.byref(self.desc))
        if need < 0:
            check(int(need), "fsaempc_ltv_workspace_bytes")
        if self._ws is None or self._ws.numel() * 8 < need:
            self._ws = torch.empty((need + 7) // 8, dtype=torch.float64, device=self.device)
        out = dict(u_opt=self._f64(B, 2 * self.N), x_opt=self._f64(B, self.nx * self.N), slack=self._f64(B, self.ns), fval=self._f64(B),
                   exitflag=torch.empty(B, dtype=torch.int32, device=self.device), iter=torch.empty(B, dtype=torch.int32, device=self.device))
        P = lambda t: C.c_void_p(t.data_ptr())
        if want_aux:
            out["kkt"] = self._f64(B)
            out["polished"] = torch.empty(B, dtype=torch.int32, device=self.device)
        if x_init is not None and (x_init.dtype != torch.float64 or not x_init.is_contiguous() or tuple(x_init.shape) != (B, self.nV)):
            raise ValueError("x_init must be a contiguous float64 (batch, nV) tensor on the GPU")
        xi = P(x_init) if x_init is not None else None   # starting point of the solve in the QP's variables [u (2N); slacks]
        if difficulty is not None and (difficulty.dtype != torch.int32 or not difficulty.is_contiguous() or tuple(difficulty.shape) != (B,)):
            raise ValueError("difficulty must be a contiguous int32 (batch,) tensor on the GPU")
        df = P(difficulty) if difficulty is not None else None
        aux = QpAux(P(out["kkt"]), P(out["polished"]), xi, df) if want_aux else QpAux(None, None, xi, df)
        rc = lib().fsaempc_ltv_step_batch_device_aux(C.byref(self.desc), C.byref(self.sp), P(x0), P(x_ref), P(x_lin), P(u_lin), C.byref(self.opts),
                                                     P(out["u_opt"]), P(out["x_opt"]), P(out["slack"]), P(out["fval"]), P(out["exitflag"]), P(out["iter"]),
                                                     C.byref(aux), P(self._ws), C.c_longlong(self._ws.numel() * 8), self._stream(stream))
        check(rc, "fsaempc_ltv_step_batch_device_aux")
        return out


    def sqp(self, x0, x_ref, x_lin, u_lin, sweeps=3, stream=None, step=1.0):
        """Re-linearisation (SQP) sweeps of SURVEY 8 f-3: the step is solved, its plan becomes the next linearisation point
        (what main.m:121-125 does from one MPC period to the next, here within one period), `sweeps` times.  `step` < 1 damps
        the move of the linearisation point, (x_lin, u_lin) += step * (plan - (x_lin, u_lin)): plain re-linearisation (step = 1)
        has no step control and a bang-bang input may keep flipping between its bounds over a long horizon.  Instances
        whose QP fails keep their previous linearisation point.  Returns the last step's outputs plus `du` = list of
        per-sweep (B,) tensors max|u_opt - u_lin| (how far each sweep still moved)."""
        torch = self.torch
        B, N, nx = self.batch, self.N, self.nx
        xl, ul = x_lin.reshape(B, N * nx).clone(), u_lin.reshape(B, 2 * N).clone()
        du, out = [], None
        for _ in range(sweeps):
            out = self.step(x0, x_ref, xl, ul, stream=stream)
            ok = (out["exitflag"] == 0).view(-1, 1)
            du.append(torch.where(ok.view(-1), (out["u_opt"] - ul).abs().amax(1), torch.full((B,), float("nan"), dtype=torch.float64, device=self.device)))
            xl = torch.where(ok, xl + step * (out["x_opt"] - xl), xl).contiguous()
            ul = torch.where(ok, ul + step * (out["u_opt"] - ul), ul).contiguous()
        out["du"] = du
        return out


def _single(model, x0, x_ref, track, dt, x_lin, u_lin, QP, device):
    import torch
    x_ref = np.asarray(x_ref, dtype=np.float64)
    nx, N = x_ref.shape
    stepper = LtvBatch(model, N, dt, track, 1, device=device)
    dev = stepper.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).T.reshape(1, -1))).to(dev)
    out = stepper.step(torch.from_numpy(np.asarray(x0, dtype=np.float64).reshape(1, -1)).to(dev), t(x_ref), t(x_lin), t(u_lin))
    torch.cuda.synchronize(dev)
    u_opt = out["u_opt"][0].cpu().numpy()
    x_opt = out["x_opt"][0].cpu().numpy()
    return u_opt, x_opt, QP, int(out["exitflag"][0]), float(out["fval"][0]), out["slack"][0].cpu().numpy()


def ltvmpc_kinetmatic_curvilinear(x0, x_ref, kappa, dt, x_lin, u_lin, QP=0, device="cuda:0"):
    """[u_opt, x_opt, QP, exitflag, fval, slack_opt] -- same outputs/order as the reference driver; `kappa` is
    the Track whose table the reference's closure interpolates.  x_ref/x_lin: nx x N, u_lin: 2 x N."""
    return _single(KINEMATIC, x0, x_ref, kappa, dt, x_lin, u_lin, QP, device)


def ltvmpc_dynamic_curvilinear(x0, x_ref, kappa, dt, x_lin, u_lin, QP=0, device="cuda:0"):
    return _single(DYNAMIC, x0, x_ref, kappa, dt, x_lin, u_lin, QP, device)
