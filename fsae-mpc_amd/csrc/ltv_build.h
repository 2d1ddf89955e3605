// ltv_build.h -- internal interface between the C ABI and the LTV-MPC construction kernels
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct LtvParams {
  int nx, N, integ;   // integ: 0 Euler, 1 RK2 (midpoint), 2 RK4
  double dt;
  int spM; double spdl; const double* xP; const double* yP;   // spline table (device)
  const double *x0, *x_ref, *x_lin, *u_lin;
  double *H, *g, *A, *lb, *ub, *lbA, *ubA;
  double *pred, *Bt, *qconst;   // Bt is required (internal operand); pred/qconst optional
};

hipError_t ltv_build_launch(const LtvParams& P, int batch, hipStream_t st);
hipError_t ltv_post_launch(int nx, int N, int ns, int batch, const double* z, const double* pred, const double* Bt,
                           const double* qconst, double* u_opt, double* x_opt, double* slack, double* fval, hipStream_t st);
size_t ltv_build_lds_bytes(int nx, int N, int threads);
