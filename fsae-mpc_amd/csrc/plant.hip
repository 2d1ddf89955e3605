// plant.hip -- the closed loop around the LTV-MPC step, batched on the device (one thread per car:
// every piece is a short scalar recurrence, the batch of independent cars is the parallel axis).
//
//   cl_pre_kernel    main.m:93-114  Cartesian -> curvilinear frame (vehicle_models/cartesian_to_curvilinear.m:17-26 with
//                    the Newton search of spline/closest_point.m:15-32, epsilon 0.01, started from the first predicted
//                    state), x0 assembly for the kinematic / dynamic model, lap check (s >= L), live reference
//   cl_plant_kernel  main.m:163-175 first predicted state -> set points; ten sub-steps of the two PID actuator loops
//                    (vehicle_models/pid_controller.m:5-18, gains main.m:84-88) driving the Cartesian dynamic bicycle
//                    (cartesian_dynamic/f_cart_dyn.m:13-54) through the 6-stage scheme of integrate_cart_dyn.m:12-22
//                    (stage formulas exactly as written there, including the doubled k2 term of k5)
#include <hip/hip_runtime.h>
#include <math.h>
#include "plant.h"

namespace {

#define DEVINL __device__ __forceinline__
struct Spl { int M; double dl; const double* xP; const double* yP; };

DEVINL double mmod(double a, double b) { return a - floor(a / b) * b; }
DEVINL void seg_lookup(int M, double dl, double t, int& seg, double& tau) {
  const double per = dl * (double)M;
  double r = t - floor(t / per) * per;  // MATLAB mod()
  if (r < 0) r += per;
  if (r >= per) r -= per;
  int i = 0;
  if (r >= 0 && r < per) i = (int)floor(r / dl);   // a non-finite arc length (a car whose state blew up) must not index the table
  if (i >= M) i = M - 1;
  if (i < 0) i = 0;
  seg = i; tau = r / dl - (double)i;
}
// value, first and second derivative of one Bezier spline at t (interpolate_spline{,_d,_dd}.m)
DEVINL void spline3(const double* P, int M, double dl, double t, double& v, double& d, double& dd) {
  int i; double u;
  seg_lookup(M, dl, t, i, u);
  const double p0 = P[i], p1 = P[i + M], p2 = P[i + 2 * M], p3 = P[i + 3 * M];
  const double w = 1 - u;
  v = p0 * (w * w * w) + 3 * p1 * (w * w) * u + 3 * p2 * w * (u * u) + p3 * (u * u * u);
  d = (-3 * w * w * p0 + 3 * (3 * u * u - 4 * u + 1) * p1 + 3 * (2 * u - 3 * u * u) * p2 + 3 * u * u * p3) / dl;
  dd = (6 * w * p0 + 6 * (3 * u - 2) * p1 + 6 * (1 - 3 * u) * p2 + 6 * u * p3) / (dl * dl);
}

DEVINL double closest_point(const Spl& sp, double x0, double y0, double s, double epsilon) {
  double delta = epsilon * 2;
  int guard = 0;
  while (fabs(delta) > epsilon && guard++ < 1000) {   // bounded: every thread leaves the loop (the reference spins on NaN)
    double X, Xd, Xdd, Y, Yd, Ydd;
    spline3(sp.xP, sp.M, sp.dl, s, X, Xd, Xdd);
    spline3(sp.yP, sp.M, sp.dl, s, Y, Yd, Ydd);
    const double dist_d = 2 * (X - x0) * Xd + 2 * (Y - y0) * Yd;
    const double dist_dd = 2 * (X - x0) * Xdd + 2 * Xd * Xd + 2 * (Y - y0) * Ydd + 2 * Yd * Yd;
    delta = dist_d / dist_dd;
    s = s - delta;
  }
  return s;
}

DEVINL double angdiff(double alpha, double beta) {   // MATLAB angdiff: beta - alpha wrapped to [-pi, pi]
  const double d = beta - alpha;
  double w = mmod(d + M_PI, 2 * M_PI) - M_PI;
  if (w == -M_PI && d > 0) w = M_PI;
  return w;
}

__global__ void cl_pre_kernel(ClPreParams P) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.batch) return;
  const Spl sp{P.spM, P.spdl, P.xP, P.yP};
  const int nx = P.nx, N = P.N;
  const double* c = P.cart + (size_t)b * 7;
  double* x0 = P.x0 + (size_t)b * nx;
  double* xr = P.x_ref + (size_t)b * nx * N;
  const double s = closest_point(sp, c[0], c[1], P.s_guess[b], 0.01);
  double X, Xd, Xdd, Y, Yd, Ydd;
  spline3(sp.xP, sp.M, sp.dl, s, X, Xd, Xdd);
  spline3(sp.yP, sp.M, sp.dl, s, Y, Yd, Ydd);
  double tx = -Yd, ty = Xd;
  const double nrm = sqrt(tx * tx + ty * ty);
  tx /= nrm; ty /= nrm;
  const double n = (c[0] - X) * tx + (c[1] - Y) * ty;
  const double mu = angdiff(atan2(Yd, Xd), c[2]);
  x0[0] = s; x0[1] = n; x0[2] = mu;
  if (nx == 5) { x0[3] = sqrt(c[3] * c[3] + c[4] * c[4]); x0[4] = c[6]; }   // main.m:95
  else { x0[3] = c[3]; x0[4] = c[4]; x0[5] = c[5]; x0[6] = c[6]; }           // main.m:97
  if (s >= P.L) P.finished[b] = 1;                                           // main.m:101-104
  {   // the car is out when the frame transform lost the track (Newton diverged) or its state left every physical range
    bool okc = fabs(s) < INFINITY && fabs(n) < 3.0 && fabs(c[3]) < 100.0 && fabs(c[4]) < 100.0;   // 3 m off a 1.5 m wide track: out of the race
    for (int j = 0; j < 7; ++j) okc = okc && fabs(c[j]) < 1e6;
    if (!okc) {
      P.finished[b] = 2;
      for (int j = 0; j < nx; ++j) x0[j] = 0.0;   // finite placeholder data for the (ignored) QP of this car
    }
  }
  double cum = 0.0;
  for (int k = 0; k < N; ++k) {                                              // main.m:107-114
    for (int j = 0; j < nx; ++j) xr[k * nx + j] = 0.0;
    double v;
    if (c[3] < P.target_vel) { v = x0[3] + 10 * P.dt * (k + 1); if (v > P.target_vel) v = P.target_vel; }
    else                     { v = x0[3] - 10 * P.dt * (k + 1); if (v < P.target_vel) v = P.target_vel; }
    xr[k * nx + 3] = v;
    cum += v * P.dt;
    xr[k * nx + 0] = x0[0] + cum;
  }
}

DEVINL void f_cart_dyn(const double* x, const double* u, double* f) {
  const double m = 280, I = 200, lr = 0.6183, lf = 0.8672, g = 9.81;
  const double theta = x[2], x_d = x[3], y_d = x[4], theta_d = x[5], delta = x[6];
  const double alpha_f = delta - atan((y_d + lf * theta_d) / (x_d + 0.01));
  const double alpha_r = -atan((y_d - lr * theta_d) / (x_d + 0.01));
  const double Fzf = m * g * lr / (lr + lf), Fzr = m * g * lf / (lr + lf);
  const double B = 12.56, C = 1.38, D = 1.60, E = -0.58;
  const double Fcf = Fzf * D * sin(C * atan(B * alpha_f - E * (B * alpha_f - atan(B * alpha_f))));
  const double Fcr = Fzr * D * sin(C * atan(B * alpha_r - E * (B * alpha_r - atan(B * alpha_r))));
  f[0] = x_d * cos(theta) - y_d * sin(theta);
  f[1] = x_d * sin(theta) + y_d * cos(theta);
  f[2] = theta_d;
  f[3] = (u[0] - Fcf * sin(delta) + m * y_d * theta_d) / m;
  f[4] = (Fcr + Fcf * cos(delta) - m * x_d * theta_d) / m;
  f[5] = (lf * Fcf * cos(delta) - lr * Fcr) / I;
  f[6] = u[1];
}

DEVINL void integrate_cart_dyn(double* x, const double* u, double dt) {
  double k1[7], k2[7], k3[7], k4[7], k5[7], k6[7], xs[7];
  f_cart_dyn(x, u, k1);
  for (int i = 0; i < 7; ++i) xs[i] = x[i] + k1[i] * dt / 2;
  f_cart_dyn(xs, u, k2);
  for (int i = 0; i < 7; ++i) xs[i] = x[i] + k1[i] * dt / 4 + k2[i] * dt / 8;
  f_cart_dyn(xs, u, k3);
  for (int i = 0; i < 7; ++i) xs[i] = x[i] - k2[i] * dt + 2 * k3[i] * dt;
  f_cart_dyn(xs, u, k4);
  for (int i = 0; i < 7; ++i) xs[i] = x[i] + 7.0 / 27 * k2[i] * dt + 10.0 / 27 * k2[i] * dt + k4[i] * dt / 27;
  f_cart_dyn(xs, u, k5);
  for (int i = 0; i < 7; ++i)
    xs[i] = x[i] + 28.0 / 625 * k1[i] * dt - k2[i] * dt / 5 + 546.0 / 625 * k3[i] * dt + 54.0 / 625 * k4[i] * dt - 378.0 / 625 * k5[i] * dt;
  f_cart_dyn(xs, u, k6);
  for (int i = 0; i < 7; ++i) x[i] = x[i] + dt * (k1[i] / 24 + 5.0 / 48 * k4[i] + 27.0 / 56 * k5[i] + 125.0 / 336 * k6[i]);
}

DEVINL double pid(double target, double current, double kp, double ki, double kd, double max_output, double* status) {
  const double error = target - current;
  const double integral_error = status[0] + error;
  const double derivative_error = error - status[1];
  double output = kp * error + ki * integral_error + kd * derivative_error;
  output = fmax(fmin(output, max_output), -max_output);
  status[0] = integral_error; status[1] = error;
  return output;
}

__global__ void cl_plant_kernel(ClPlantParams P) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.batch) return;
  if (P.finished && P.finished[b]) return;        // the reference leaves the loop when the lap is complete
  // The reference keeps driving on whatever plan the solver returned, whatever its exit flag (main.m:163-175); `exitflag`
  // (optional) only holds a car when the caller asks for it with a flag < -100 (not a solver outcome).
  if (P.exitflag && P.exitflag[b] < -100) return;
  double x[7], st[4], u[2] = {0.0, 0.0};
  for (int i = 0; i < 7; ++i) x[i] = P.cart[(size_t)b * 7 + i];
  for (int i = 0; i < 4; ++i) st[i] = P.pid[(size_t)b * 4 + i];
  const double* xo = P.x_opt + (size_t)b * P.nx * P.N;
  const double v_ref = xo[3], delta_ref = xo[P.nx - 1];     // main.m:167-168 (x_opt(4), x_opt(N_x))
  if (!(fabs(v_ref) < INFINITY) || !(fabs(delta_ref) < INFINITY)) return;   // no finite plan at all: hold the car
  for (int j = 0; j < 10; ++j) {                             // main.m:171-175
    u[0] = pid(v_ref, x[3], 16000.0, 0.0, 0.0, 2800.0, st);
    u[1] = pid(delta_ref, x[6], 80.0, 0.0, 0.0, 0.8, st + 2);
    integrate_cart_dyn(x, u, P.dt / 10);
  }
  for (int i = 0; i < 7; ++i) P.cart[(size_t)b * 7 + i] = x[i];
  for (int i = 0; i < 4; ++i) P.pid[(size_t)b * 4 + i] = st[i];
  if (P.u_last) { P.u_last[(size_t)b * 2] = u[0]; P.u_last[(size_t)b * 2 + 1] = u[1]; }
}

// main.m:122-126: the plan of this step becomes the linearisation point (and the source of the set points) of the next
// one.  The reference takes over whatever qpOASES returned; this build takes a plan over when the solve ended with exit
// flag 0 or 1 and every entry is finite -- after an abnormal exit (-1, -2) the returned point is the last interior-point
// iterate, which unlike an active-set iterate need not respect the actuator bounds, so the car keeps driving on its last
// good plan instead (documented deviation).
__global__ void cl_accept_kernel(int len_x, int len_u, int batch, const double* x_new, const double* u_new, const int* exitflag, double* x_keep, double* u_keep) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  bool ok = !exitflag || exitflag[b] == 0 || exitflag[b] == 1;
  for (int i = 0; i < len_x; ++i) ok = ok && fabs(x_new[(size_t)b * len_x + i]) < INFINITY;
  for (int i = 0; i < len_u; ++i) ok = ok && fabs(u_new[(size_t)b * len_u + i]) < INFINITY;
  if (!ok) return;
  for (int i = 0; i < len_x; ++i) x_keep[(size_t)b * len_x + i] = x_new[(size_t)b * len_x + i];
  for (int i = 0; i < len_u; ++i) u_keep[(size_t)b * len_u + i] = u_new[(size_t)b * len_u + i];
}

}  // namespace

hipError_t cl_accept_launch(int len_x, int len_u, int batch, const double* x_new, const double* u_new, const int* exitflag, double* x_keep, double* u_keep, hipStream_t st) {
  if (batch == 0) return hipSuccess;
  hipLaunchKernelGGL(cl_accept_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, len_x, len_u, batch, x_new, u_new, exitflag, x_keep, u_keep);
  return hipGetLastError();
}
hipError_t cl_pre_launch(const ClPreParams& P, hipStream_t st) {
  if (P.batch == 0) return hipSuccess;
  hipLaunchKernelGGL(cl_pre_kernel, dim3((P.batch + 63) / 64), dim3(64), 0, st, P);
  return hipGetLastError();
}
hipError_t cl_plant_launch(const ClPlantParams& P, hipStream_t st) {
  if (P.batch == 0) return hipSuccess;
  hipLaunchKernelGGL(cl_plant_kernel, dim3((P.batch + 63) / 64), dim3(64), 0, st, P);
  return hipGetLastError();
}
