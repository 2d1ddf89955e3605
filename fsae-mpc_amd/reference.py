"""Host-side mirror of the reference-trajectory functions of the LTV-MPC step (device kernels behind the C ABI):
`obtain_reference` (util/obtain_reference.m) and the live generator of main.m:107-114, batched over instances."""
import ctypes as C

import numpy as np

from ._lib import check, lib


def _stream(torch, device, stream):
    return C.c_void_p(stream if stream is not None else torch.cuda.current_stream(device).cuda_stream)


def obtain_reference_batch_device(plan, ds, N_s, t, s0, dt, N_t, stream=None):
    """plan: (8*N_s,) planner vector, t: (N_s,), s0: (B,) -- float64 CUDA tensors.  Returns x_ref (B, N_t, 7): per
    instance the 7 x N_t column-major matrix of obtain_reference.m."""
    import torch
    assert plan.is_cuda and t.is_cuda and s0.is_cuda and plan.dtype == t.dtype == s0.dtype == torch.float64
    assert plan.numel() == 8 * N_s and t.numel() == N_s
    B = s0.numel()
    x_ref = torch.empty((B, N_t, 7), dtype=torch.float64, device=s0.device)
    rc = lib().fsaempc_obtain_reference_batch_device(C.c_void_p(plan.data_ptr()), C.c_double(ds), int(N_s), C.c_void_p(t.data_ptr()),
                                                     C.c_void_p(s0.data_ptr()), C.c_double(dt), int(N_t), int(B),
                                                     C.c_void_p(x_ref.data_ptr()), _stream(torch, s0.device, stream))
    check(rc, "fsaempc_obtain_reference_batch_device")
    return x_ref


def obtain_reference(x, ds, N_s, t, s0, dt, N_t, device="cuda:0"):
    """Same signature as the reference's obtain_reference(x, ds, N_s, t, s0, dt, N_t); returns x_ref (7, N_t)."""
    import torch
    dev = torch.device(device)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel())).to(dev)
    out = obtain_reference_batch_device(up(x), float(ds), int(N_s), up(t), up([s0]), float(dt), int(N_t))
    torch.cuda.synchronize(dev)
    return out[0].cpu().numpy().T.copy()


def reference_live_batch_device(x0, N, dt, target_vel=20.0, stream=None):
    """main.m:107-114 on the device: x0 (B, nx) float64 CUDA tensor -> x_ref (B, N, nx)."""
    import torch
    assert x0.is_cuda and x0.dtype == torch.float64 and x0.is_contiguous()
    B, nx = x0.shape
    x_ref = torch.empty((B, N, nx), dtype=torch.float64, device=x0.device)
    rc = lib().fsaempc_reference_live_batch_device(int(nx), int(N), C.c_double(dt), C.c_double(target_vel), int(B),
                                                   C.c_void_p(x0.data_ptr()), C.c_void_p(x_ref.data_ptr()), _stream(torch, x0.device, stream))
    check(rc, "fsaempc_reference_live_batch_device")
    return x_ref
