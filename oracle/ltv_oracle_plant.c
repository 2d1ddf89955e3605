/*
 * ltv_oracle_plant.c -- CPU restatement of the closed-loop pieces around the LTV-MPC step.
 * TEST INFRASTRUCTURE ONLY (see ltv_oracle.h); parity unpinned (no reference vectors exist).
 * Follows, line by line:
 *   spline/closest_point.m:15-32, vehicle_models/cartesian_to_curvilinear.m:17-26 (+ interpolate_spline.m,
 *   interpolate_angle.m), vehicle_models/pid_controller.m:5-18,
 *   vehicle_models/cartesian_dynamic/f_cart_dyn.m:13-54, integrate_cart_dyn.m:12-22 (its stage formulas as written,
 *   including the doubled k2 term of k5), main.m:93-98 (x0 assembly), :107-114 (reference), :171-175 (actuator loop).
 */
#include "ltv_oracle.h"
#include <math.h>
#include <string.h>

static double mmod(double a, double b) { return a - floor(a / b) * b; }

double orc_closest_point(const orc_spline* sp, double x0, double y0, double s, double epsilon) {
  double delta = epsilon * 2;                                              /* closest_point.m:15 */
  int guard = 0;
  while (fabs(delta) > epsilon && guard++ < 1000) {                        /* :17 (guard: the reference loops forever on NaN) */
    const double X = orc_spline_val(sp->xP, sp->M, sp->dl, s), Y = orc_spline_val(sp->yP, sp->M, sp->dl, s);
    const double Xd = orc_spline_d(sp->xP, sp->M, sp->dl, s), Yd = orc_spline_d(sp->yP, sp->M, sp->dl, s);
    const double Xdd = orc_spline_dd(sp->xP, sp->M, sp->dl, s), Ydd = orc_spline_dd(sp->yP, sp->M, sp->dl, s);
    const double dist_d = 2 * (X - x0) * Xd + 2 * (Y - y0) * Yd;           /* :26 */
    const double dist_dd = 2 * (X - x0) * Xdd + 2 * Xd * Xd + 2 * (Y - y0) * Ydd + 2 * Yd * Yd;   /* :27 */
    delta = dist_d / dist_dd;                                              /* :30 */
    s = s - delta;
  }
  return s;
}

static double angdiff(double alpha, double beta) {   /* MATLAB angdiff: beta - alpha wrapped to [-pi, pi] */
  const double d = beta - alpha;
  double w = mmod(d + M_PI, 2 * M_PI) - M_PI;
  if (w == -M_PI && d > 0) w = M_PI;
  return w;
}

void orc_cart_to_curv(const orc_spline* sp, double x, double y, double theta, double s0, double* s_out, double* n_out, double* mu_out) {
  const double s = orc_closest_point(sp, x, y, s0, 0.01);                  /* cartesian_to_curvilinear.m:17 */
  const double cx = x - orc_spline_val(sp->xP, sp->M, sp->dl, s), cy = y - orc_spline_val(sp->yP, sp->M, sp->dl, s);
  double tx = -orc_spline_d(sp->yP, sp->M, sp->dl, s), ty = orc_spline_d(sp->xP, sp->M, sp->dl, s);   /* :21-22 */
  const double nrm = sqrt(tx * tx + ty * ty);
  tx /= nrm; ty /= nrm;
  *s_out = s;
  *n_out = cx * tx + cy * ty;                                              /* :24 */
  const double ang = atan2(orc_spline_d(sp->yP, sp->M, sp->dl, s), orc_spline_d(sp->xP, sp->M, sp->dl, s));   /* interpolate_angle.m */
  *mu_out = angdiff(ang, theta);                                           /* :26 */
}

void orc_f_cart_dyn(const double* x, const double* u, double* f) {
  const double m = 280, I = 200, lr = 0.6183, lf = 0.8672, g = 9.81;       /* f_cart_dyn.m:13-19 */
  const double theta = x[2], x_d = x[3], y_d = x[4], theta_d = x[5], delta = x[6];
  const double Fx = u[0], delta_d = u[1];
  const double alpha_f = delta - atan((y_d + lf * theta_d) / (x_d + 0.01));   /* :31-32 */
  const double alpha_r = -atan((y_d - lr * theta_d) / (x_d + 0.01));
  const double Fzf = m * g * lr / (lr + lf), Fzr = m * g * lf / (lr + lf);
  const double B = 12.56, C = 1.38, D = 1.60, E = -0.58;
  const double Fcf = Fzf * D * sin(C * atan(B * alpha_f - E * (B * alpha_f - atan(B * alpha_f))));
  const double Fcr = Fzr * D * sin(C * atan(B * alpha_r - E * (B * alpha_r - atan(B * alpha_r))));
  f[0] = x_d * cos(theta) - y_d * sin(theta);                              /* :47-53 */
  f[1] = x_d * sin(theta) + y_d * cos(theta);
  f[2] = theta_d;
  f[3] = (Fx - Fcf * sin(delta) + m * y_d * theta_d) / m;
  f[4] = (Fcr + Fcf * cos(delta) - m * x_d * theta_d) / m;
  f[5] = (lf * Fcf * cos(delta) - lr * Fcr) / I;
  f[6] = delta_d;
}

void orc_integrate_cart_dyn(const double* x0, const double* u, double dt, double* x) {
  double k1[7], k2[7], k3[7], k4[7], k5[7], k6[7], xs[7];
  orc_f_cart_dyn(x0, u, k1);                                               /* integrate_cart_dyn.m:12-17 */
  for (int i = 0; i < 7; ++i) xs[i] = x0[i] + k1[i] * dt / 2;
  orc_f_cart_dyn(xs, u, k2);
  for (int i = 0; i < 7; ++i) xs[i] = x0[i] + k1[i] * dt / 4 + k2[i] * dt / 8;
  orc_f_cart_dyn(xs, u, k3);
  for (int i = 0; i < 7; ++i) xs[i] = x0[i] - k2[i] * dt + 2 * k3[i] * dt;
  orc_f_cart_dyn(xs, u, k4);
  for (int i = 0; i < 7; ++i) xs[i] = x0[i] + 7.0 / 27 * k2[i] * dt + 10.0 / 27 * k2[i] * dt + k4[i] * dt / 27;   /* :16 as written (k2 twice) */
  orc_f_cart_dyn(xs, u, k5);
  for (int i = 0; i < 7; ++i)
    xs[i] = x0[i] + 28.0 / 625 * k1[i] * dt - k2[i] * dt / 5 + 546.0 / 625 * k3[i] * dt + 54.0 / 625 * k4[i] * dt - 378.0 / 625 * k5[i] * dt;
  orc_f_cart_dyn(xs, u, k6);
  for (int i = 0; i < 7; ++i) {
    const double f = k1[i] / 24 + 5.0 / 48 * k4[i] + 27.0 / 56 * k5[i] + 125.0 / 336 * k6[i];   /* :19 */
    x[i] = x0[i] + dt * f;                                                 /* :22 */
  }
}

double orc_pid(double target, double current, const double* settings, double* status) {
  const double kp = settings[0], ki = settings[1], kd = settings[2], max_output = settings[3];   /* pid_controller.m:5-8 */
  const double error = target - current;
  const double integral_error = status[0] + error;
  const double derivative_error = error - status[1];
  double output = kp * error + ki * integral_error + kd * derivative_error;
  output = fmax(fmin(output, max_output), -max_output);                    /* :15 */
  status[0] = integral_error; status[1] = error;
  return output;
}

/* main.m:171-175: ten actuator sub-steps towards (v_ref, delta_ref); pid = [vel_I, vel_e, steer_I, steer_e] */
void orc_plant_step(double* x, double* pid, double v_ref, double delta_ref, double dt, double* u_last) {
  const double vel_set[4] = {16000.0, 0, 0, 2800}, steer_set[4] = {80.0, 0, 0, 0.8};   /* main.m:84-88 */
  double u[2] = {0, 0}, xn[7];
  for (int j = 0; j < 10; ++j) {
    u[0] = orc_pid(v_ref, x[3], vel_set, pid);
    u[1] = orc_pid(delta_ref, x[6], steer_set, pid + 2);
    orc_integrate_cart_dyn(x, u, dt / 10, xn);
    memcpy(x, xn, sizeof(xn));
  }
  u_last[0] = u[0]; u_last[1] = u[1];
}

/* main.m:93-114: frame transform, x0 assembly, lap check, reference.  Returns 1 if the lap is finished (s >= L). */
int orc_cl_pre(int model, int N, double dt, double target_vel, const orc_spline* sp, double L, const double* cart, double s_guess,
               double* x0, double* x_ref) {
  const int nx = orc_nx(model);
  double s, n, mu;
  orc_cart_to_curv(sp, cart[0], cart[1], cart[2], s_guess, &s, &n, &mu);
  if (model == ORC_MODEL_KINEMATIC) { x0[0] = s; x0[1] = n; x0[2] = mu; x0[3] = sqrt(cart[3] * cart[3] + cart[4] * cart[4]); x0[4] = cart[6]; }
  else { x0[0] = s; x0[1] = n; x0[2] = mu; x0[3] = cart[3]; x0[4] = cart[4]; x0[5] = cart[5]; x0[6] = cart[6]; }
  memset(x_ref, 0, sizeof(double) * nx * N);
  double cum = 0;
  for (int k = 0; k < N; ++k) {
    double v;
    if (cart[3] < target_vel) { v = x0[3] + 10 * dt * (k + 1); if (v > target_vel) v = target_vel; }   /* main.m:107 tests x(4), ramps from x0(4) */
    else                      { v = x0[3] - 10 * dt * (k + 1); if (v < target_vel) v = target_vel; }
    x_ref[k * nx + 3] = v;
    cum += v * dt;
    x_ref[k * nx + 0] = x0[0] + cum;
  }
  return s >= L;
}
