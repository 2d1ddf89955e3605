// reference.h -- internal interface between the C ABI and the reference-trajectory kernels
#pragma once
#include <hip/hip_runtime.h>

struct RefParams {
  const double* plan;   // 8 x N_s planner vector (n, mu, x_d, y_d, theta_d, delta, a, delta_d per cell), device
  const double* t;      // N_s per-cell traversal times, device
  const double* s0;     // batch arc-length positions, device
  double* x_ref;        // batch x (7 x N_t column-major), device
  double ds, dt;
  int N_s, N_t, batch;
};
hipError_t obtain_reference_launch(const RefParams& P, hipStream_t st);
hipError_t reference_live_launch(int nx, int N, double dt, double target_vel, int batch, const double* x0, double* x_ref, hipStream_t st);
