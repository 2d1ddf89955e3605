// reference.hip -- reference trajectories of the LTV-MPC step, batched on the device (one thread per instance:
// both are short sequential walks, the batch is the parallel axis).
//
//   obtain_reference_kernel   util/obtain_reference.m:5-48  time-resampling of an s-domain plan (8 values per cell:
//                             n, mu, x_d, y_d, theta_d, delta, a, delta_d; per-cell traversal times t) into
//                             x_ref (7 x N_t) for a car at arc length s0
//   reference_live_kernel     main.m:107-114  velocity ramp +-10 m/s^2 clipped at TARGET_VEL, s_ref = s0 + cumsum(v dt)
#include <hip/hip_runtime.h>
#include <math.h>
#include "reference.h"

// plain IEEE operations in source order (no FMA contraction): these index walks are compared bit for bit with the oracle
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ double mod_floor(double a, double b) { return a - floor(a / b) * b; }   // MATLAB mod for b > 0

__global__ void obtain_reference_kernel(RefParams P) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.batch) return;
  const int Ns = P.N_s, Nt = P.N_t;
  const double ds = P.ds, L = ds * Ns, dt = P.dt;
  const double s0 = P.s0[b];
  double* xr = P.x_ref + (size_t)b * 7 * Nt;
  // idx is kept 0-based here (MATLAB idx-1); rto as in the reference
  const double pos = mod_floor(s0, L) / ds;                      // obtain_reference.m:21-22
  int idx = (pos >= 0 && pos < (double)Ns) ? (int)floor(pos) : 0;   // a non-finite s0 must not index the plan
  if (idx >= Ns) idx = Ns - 1;
  double rto = mod_floor(pos, 1.0);
  const int idx1 = idx; const double rto1 = rto;
  for (int i = 0; i < Nt; ++i) {                                  // obtain_reference.m:24-35
    double t_rem = dt;
    const int idx_prev = idx; const double rto_prev = rto;
    rto = rto_prev + t_rem / P.t[idx];
    t_rem -= P.t[idx_prev] * (1.0 - rto_prev);
    while (rto > 1.0) {
      idx = (idx + 1) % Ns;                                       // nxt()
      rto = t_rem / P.t[idx];
      t_rem -= P.t[idx];
    }
    const int nx_ = (idx + 1) % Ns;
    // obtain_reference.m:41: mod(idx(i) + rto(i) - idx(1) - rto(1), N_s) * ds
    xr[i * 7 + 0] = s0 + mod_floor((double)(idx + 1) + rto - (double)(idx1 + 1) - rto1, (double)Ns) * ds;   // same operands and order as the 1-based original
#pragma unroll
    for (int c = 0; c < 6; ++c) {                                 // n, mu, x_d, y_d, theta_d, delta  (:42-47)
      const double a0 = P.plan[(size_t)idx * 8 + c], a1 = P.plan[(size_t)nx_ * 8 + c];
      xr[i * 7 + 1 + c] = a0 + (a1 - a0) * rto;
    }
  }
}

__global__ void reference_live_kernel(int nx, int N, double dt, double target_vel, int batch, const double* x0, double* x_ref) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const double s0 = x0[(size_t)b * nx + 0], v0 = x0[(size_t)b * nx + 3];
  double* xr = x_ref + (size_t)b * nx * N;
  double cum = 0.0;
  for (int k = 0; k < N; ++k) {
    for (int j = 0; j < nx; ++j) xr[k * nx + j] = 0.0;
    double v;
    if (v0 < target_vel) { v = v0 + 10 * dt * (k + 1); if (v > target_vel) v = target_vel; }   // main.m:108-111
    else                 { v = v0 - 10 * dt * (k + 1); if (v < target_vel) v = target_vel; }
    xr[k * nx + 3] = v;
    cum += v * dt;                                                                               // main.m:113 cumsum
    xr[k * nx + 0] = s0 + cum;
  }
}

}  // namespace

hipError_t obtain_reference_launch(const RefParams& P, hipStream_t st) {
  if (P.batch == 0) return hipSuccess;
  hipLaunchKernelGGL(obtain_reference_kernel, dim3((P.batch + 63) / 64), dim3(64), 0, st, P);
  return hipGetLastError();
}
hipError_t reference_live_launch(int nx, int N, double dt, double target_vel, int batch, const double* x0, double* x_ref, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  hipLaunchKernelGGL(reference_live_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, nx, N, dt, target_vel, batch, x0, x_ref);
  return hipGetLastError();
}
