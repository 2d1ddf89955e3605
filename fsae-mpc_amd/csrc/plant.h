// plant.h -- internal interface between the C ABI and the closed-loop kernels (plant.hip)
#pragma once
#include <hip/hip_runtime.h>

struct ClPreParams {
  int nx, N, batch;
  double dt, target_vel, L;
  int spM; double spdl; const double* xP; const double* yP;   // spline table (device)
  const double* cart;      // batch x 7  Cartesian state [x, y, theta, x_d, y_d, theta_d, delta]
  const double* s_guess;   // batch      start of the closest-point search (first predicted s)
  double* x0;              // batch x nx
  double* x_ref;           // batch x (nx x N)
  int* finished;           // batch      set to 1 when s >= L
};
struct ClPlantParams {
  int nx, N, batch;
  double dt;
  double* cart;            // batch x 7, in/out
  double* pid;             // batch x 4, in/out: [vel integral, vel last error, steer integral, steer last error]
  const double* x_opt;     // batch x (nx x N): predicted states of this step's plan
  const int* finished;     // optional
  const int* exitflag;     // optional: cars whose step did not solve keep their state
  double* u_last;          // optional batch x 2: actuator rates of the last sub-step
};
hipError_t cl_pre_launch(const ClPreParams& P, hipStream_t st);
hipError_t cl_plant_launch(const ClPlantParams& P, hipStream_t st);
hipError_t cl_accept_launch(int len_x, int len_u, int batch, const double* x_new, const double* u_new, const int* exitflag, double* x_keep, double* u_keep, hipStream_t st);
