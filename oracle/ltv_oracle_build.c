/*
 * ltv_oracle_build.c -- CPU ORACLE (test infrastructure, NOT the product).
 * QP *construction* half of the LTV-MPC hot path; each function cites the
 * reference file:line it restates.  See ltv_oracle.h for the parity statement
 * ("parity unpinned": no golden vectors exist in the reference).
 * Column-major everywhere; index helper IDX(i,j,ld) = i + j*ld (0-based).
 */
#include "ltv_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX(i, j, ld) ((size_t)(i) + (size_t)(j) * (size_t)(ld))

static const double LR = 0.6183, LF = 0.8672;

/* ---- spline (spline/interpolate_spline*.m) -------------------------------- */
static void seg_lookup(int M, double dl, double t, int* seg, double* tau) {
  /* interpolate_spline_d.m:12-14: t = mod(t, dl*length(P)); i = floor(t/dl)+1; t = t/dl-(i-1) */
  double per = dl * (double)M;
  double r = t - floor(t / per) * per; /* MATLAB mod() */
  if (r < 0) r += per;
  if (r >= per) r -= per;
  int i = (int)floor(r / dl);
  if (i >= M) i = M - 1; /* guard against r/dl rounding up to M (MATLAB would index out of range) */
  *seg = i;
  *tau = r / dl - (double)i;
}
double orc_spline_val(const double* P, int M, double dl, double t) {
  int i; double s;
  seg_lookup(M, dl, t, &i, &s);
  /* interpolate_spline.m:17-18 */
  return P[IDX(i, 0, M)] * pow(1 - s, 3) + 3 * P[IDX(i, 1, M)] * (1 - s) * (1 - s) * s
       + 3 * P[IDX(i, 2, M)] * (1 - s) * s * s + P[IDX(i, 3, M)] * s * s * s;
}
double orc_spline_d(const double* P, int M, double dl, double t) {
  int i; double s;
  seg_lookup(M, dl, t, &i, &s);
  /* interpolate_spline_d.m:17-21 */
  double v = -3 * (1 - s) * (1 - s) * P[IDX(i, 0, M)] + 3 * (3 * s * s - 4 * s + 1) * P[IDX(i, 1, M)]
           + 3 * (2 * s - 3 * s * s) * P[IDX(i, 2, M)] + 3 * s * s * P[IDX(i, 3, M)];
  return v / dl;
}
double orc_spline_dd(const double* P, int M, double dl, double t) {
  int i; double s;
  seg_lookup(M, dl, t, &i, &s);
  /* interpolate_spline_dd.m:17-21 */
  double v = 6 * (1 - s) * P[IDX(i, 0, M)] + 6 * (3 * s - 2) * P[IDX(i, 1, M)]
           + 6 * (1 - 3 * s) * P[IDX(i, 2, M)] + 6 * s * P[IDX(i, 3, M)];
  return v / (dl * dl);
}
double orc_kappa(const orc_spline* sp, double s) {
  /* interpolate_curvature.m:12-18 */
  double Xd = orc_spline_d(sp->xP, sp->M, sp->dl, s);
  double Yd = orc_spline_d(sp->yP, sp->M, sp->dl, s);
  double Xdd = orc_spline_dd(sp->xP, sp->M, sp->dl, s);
  double Ydd = orc_spline_dd(sp->yP, sp->M, sp->dl, s);
  return (Xd * Ydd - Xdd * Yd) / pow(Xd * Xd + Yd * Yd, 1.5);
}

/* ---- kinematic model ------------------------------------------------------ */
void orc_f_kin(const double* x, const double* u, const orc_spline* sp, double* f) {
  /* f_curv_kin.m:13-29 */
  double lr_ratio = LR / (LR + LF);
  double k = orc_kappa(sp, x[0]);
  double beta = atan(lr_ratio * tan(x[4]));
  double s_mb = sin(x[2] + beta), c_mb = cos(x[2] + beta);
  double denom_nk = 1.0 / (1.0 - x[1] * k);
  f[0] = x[3] * c_mb * denom_nk;
  f[1] = x[3] * s_mb;
  f[2] = x[3] * sin(beta) / LR - x[3] * c_mb * denom_nk * k;
  f[3] = u[0];
  f[4] = u[1];
}
void orc_A_kin(const double* x, const orc_spline* sp, double* A) {
  /* A_curv_kin.m:15-55 ; kappa_d never supplied => s_s = mu_s = 0 (A_curv_kin.m:44-48 dead) */
  double lr_ratio = LR / (LR + LF);
  double k = orc_kappa(sp, x[0]);
  double td = tan(x[4]);
  double beta = atan(lr_ratio * td);
  double s_mb = sin(x[2] + beta), c_mb = cos(x[2] + beta);
  double sec = 1.0 / cos(x[4]);
  double beta_d = lr_ratio * sec * sec / (1 + (lr_ratio * td) * (lr_ratio * td));
  double denom_nk = 1.0 / (1.0 - x[1] * k);
  double s_n = x[3] * c_mb * denom_nk * denom_nk * k;
  double s_mu = -x[3] * s_mb * denom_nk;
  double s_v = c_mb * denom_nk;
  double s_delta = -x[3] * s_mb * denom_nk * beta_d;
  double n_mu = x[3] * c_mb, n_v = s_mb, n_delta = x[3] * c_mb * beta_d;
  double mu_n = -s_n * k, mu_mu = -s_mu * k;
  double mu_v = sin(beta) / LR - s_v * k;
  double mu_delta = x[3] * cos(beta) * beta_d / LR - s_delta * k;
  memset(A, 0, 25 * sizeof(double));
  A[IDX(0, 1, 5)] = s_n;  A[IDX(0, 2, 5)] = s_mu;  A[IDX(0, 3, 5)] = s_v;  A[IDX(0, 4, 5)] = s_delta;
  A[IDX(1, 2, 5)] = n_mu; A[IDX(1, 3, 5)] = n_v;   A[IDX(1, 4, 5)] = n_delta;
  A[IDX(2, 1, 5)] = mu_n; A[IDX(2, 2, 5)] = mu_mu; A[IDX(2, 3, 5)] = mu_v; A[IDX(2, 4, 5)] = mu_delta;
}

/* ---- dynamic model -------------------------------------------------------- */
static const double VM = 280, VI = 200, GRAV = 9.81;
static const double PB = 12.56, PC = 1.38, PD = 1.60, PE = -0.58;

void orc_f_dyn(const double* x, const double* u, const orc_spline* sp, double* f, double* Fcr_out) {
  /* f_curv_dyn.m:13-62 */
  double n = x[1], mu = x[2], x_d = x[3], y_d = x[4], th_d = x[5], delta = x[6];
  double Fx = u[0] * VM, delta_d = u[1];
  double x_d_hat = x_d + 5 * exp(-x_d / 5);
  double k = orc_kappa(sp, x[0]);
  double denom_nk = 1.0 / (1.0 - n * k);
  double alpha_f = delta - atan((y_d + LF * th_d) / x_d_hat);
  double alpha_r = -atan((y_d - LR * th_d) / x_d_hat);
  double Fzf = VM * GRAV * LR / (LR + LF), Fzr = VM * GRAV * LF / (LR + LF);
  double Fcf = Fzf * PD * sin(PC * atan(PB * alpha_f - PE * (PB * alpha_f - atan(PB * alpha_f))));
  double Fcr = Fzr * PD * sin(PC * atan(PB * alpha_r - PE * (PB * alpha_r - atan(PB * alpha_r))));
  f[0] = (x_d * cos(mu) - y_d * sin(mu)) * denom_nk;
  f[1] = x_d * sin(mu) + y_d * cos(mu);
  f[2] = th_d - (x_d * cos(mu) - y_d * sin(mu)) * denom_nk * k;
  f[3] = (Fx - Fcf * sin(delta) + VM * y_d * th_d) / VM;
  f[4] = (Fcr + Fcf * cos(delta) - VM * x_d * th_d) / VM;
  f[5] = (LF * Fcf * cos(delta) - LR * Fcr) / VI;
  f[6] = delta_d;
  if (Fcr_out) *Fcr_out = Fcr;
}
void orc_A_dyn(const double* x, const orc_spline* sp, double* A, double* byp) {
  /* A_curv_dyn.m:15-106 (treated as the definition, incl. yd_thetad's -m*x_d_hat, A_curv_dyn.m:90) */
  double n = x[1], mu = x[2], x_d = x[3], y_d = x[4], th_d = x[5], delta = x[6];
  double m = VM, I = VI;
  double x_d_hat = x_d + 5 * exp(-x_d / 5);
  double x_d_hat_d = 1 - exp(-x_d / 5);
  double alpha_f = delta - atan((y_d + LF * th_d) / x_d_hat);
  double alpha_r = -atan((y_d - LR * th_d) / x_d_hat);
  double Fzf = m * GRAV * LR / (LR + LF), Fzr = m * GRAV * LF / (LR + LF);
  double af_arg = PB * alpha_f - PE * (PB * alpha_f - atan(PB * alpha_f));
  double ar_arg = PB * alpha_r - PE * (PB * alpha_r - atan(PB * alpha_r));
  double Fcf = Fzf * PD * sin(PC * atan(af_arg));
  double Fcr = Fzr * PD * sin(PC * atan(ar_arg));
  double Fcf_d = Fzf * PD * cos(PC * atan(af_arg)) * PC / (1 + af_arg * af_arg)
               * (PB - PE * (PB - PB / (1 + PB * PB * alpha_f * alpha_f)));
  double Fcr_d = Fzr * PD * cos(PC * atan(ar_arg)) * PC / (1 + ar_arg * ar_arg)
               * (PB - PE * (PB - PB / (1 + PB * PB * alpha_r * alpha_r)));
  double k = orc_kappa(sp, x[0]);
  double denom_nk = 1.0 / (1.0 - n * k);
  double vf = (y_d + LF * th_d) / x_d_hat, vr = (y_d - LR * th_d) / x_d_hat;
  double denom_vf2 = 1.0 / (1 + vf * vf), denom_vr2 = 1.0 / (1 + vr * vr);
  double cm = cos(mu), sm = sin(mu), cd = cos(delta), sd = sin(delta);

  double s_n = (x_d * cm - y_d * sm) * denom_nk * denom_nk * k;
  double s_mu = (-x_d * sm - y_d * cm) * denom_nk;
  double s_xd = cm * denom_nk, s_yd = -sm * denom_nk;
  double n_mu = x_d * cm - y_d * sm, n_xd = sm, n_yd = cm;
  double mu_n = -s_n * k, mu_mu = -s_mu * k, mu_xd = -s_xd * k, mu_yd = -s_yd * k, mu_thetad = 1;
  double xd_xd = -Fcf_d * denom_vf2 * vf * sd * x_d_hat_d / (m * x_d_hat);
  double xd_yd = (Fcf_d * denom_vf2 * sd / x_d_hat + m * th_d) / m;
  double xd_thetad = (Fcf_d * denom_vf2 * LF * sd / x_d_hat + m * y_d) / m;
  double xd_delta = (-Fcf * cd - Fcf_d * sd) / m;
  double yd_xd = (Fcr_d * denom_vr2 * vr * x_d_hat_d / x_d_hat + Fcf_d * denom_vf2 * vf * cd * x_d_hat_d / x_d_hat - m * th_d) / m;
  double yd_yd = (-Fcr_d * denom_vr2 / x_d_hat - Fcf_d * denom_vf2 / x_d_hat * cd) / m;
  double yd_thetad = (Fcr_d * denom_vr2 * LR / x_d_hat - Fcf_d * denom_vf2 * LF / x_d_hat * cd - m * x_d_hat) / m;
  double yd_delta = (-Fcf * sd + Fcf_d * cd) / m;
  double t_xd = (LF * Fcf_d * denom_vf2 * vf * cd * x_d_hat_d / x_d_hat - LR * Fcr_d * denom_vr2 * vr * x_d_hat_d / x_d_hat) / I;
  double t_yd = (-LF * Fcf_d * denom_vf2 * cd / x_d_hat + LR * Fcr_d * denom_vr2 / x_d_hat) / I;
  double t_thetad = (-LF * Fcf_d * denom_vf2 * LF * cd / x_d_hat - LR * Fcr_d * denom_vr2 * LR / x_d_hat) / I;
  double t_delta = (-LF * Fcf * sd + LF * Fcf_d * cd) / I;

  if (A) {
    memset(A, 0, 49 * sizeof(double));
    A[IDX(0, 1, 7)] = s_n;  A[IDX(0, 2, 7)] = s_mu;  A[IDX(0, 3, 7)] = s_xd;  A[IDX(0, 4, 7)] = s_yd;
    A[IDX(1, 2, 7)] = n_mu; A[IDX(1, 3, 7)] = n_xd;  A[IDX(1, 4, 7)] = n_yd;
    A[IDX(2, 1, 7)] = mu_n; A[IDX(2, 2, 7)] = mu_mu; A[IDX(2, 3, 7)] = mu_xd; A[IDX(2, 4, 7)] = mu_yd; A[IDX(2, 5, 7)] = mu_thetad;
    A[IDX(3, 3, 7)] = xd_xd; A[IDX(3, 4, 7)] = xd_yd; A[IDX(3, 5, 7)] = xd_thetad; A[IDX(3, 6, 7)] = xd_delta;
    A[IDX(4, 3, 7)] = yd_xd; A[IDX(4, 4, 7)] = yd_yd; A[IDX(4, 5, 7)] = yd_thetad; A[IDX(4, 6, 7)] = yd_delta;
    A[IDX(5, 3, 7)] = t_xd;  A[IDX(5, 4, 7)] = t_yd;  A[IDX(5, 5, 7)] = t_thetad;  A[IDX(5, 6, 7)] = t_delta;
  }
  if (byp) {
    byp[0] = Fcr; byp[1] = Fcr_d; byp[2] = vr; byp[3] = denom_vr2;
    byp[4] = x_d_hat; byp[5] = x_d_hat_d; byp[6] = vf; byp[7] = denom_vf2;
  }
}

/* ---- small dense helpers (column-major) ----------------------------------- */
static void mm(int m, int k, int n, const double* A, const double* B, double* C) { /* C = A(mxk) B(kxn) */
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) {
      double s = 0;
      for (int p = 0; p < k; ++p) s += A[IDX(i, p, m)] * B[IDX(p, j, k)];
      C[IDX(i, j, m)] = s;
    }
}

int orc_nx(int model) { return model == ORC_MODEL_KINEMATIC ? 5 : 7; }
int orc_ns(int model) { return model == ORC_MODEL_KINEMATIC ? 1 : 4; }
int orc_nV(int model, int N) { return 2 * N + orc_ns(model); }
int orc_nC(int model, int N) { return (model == ORC_MODEL_KINEMATIC ? 6 : 20) * N; }

static void model_f(int model, const double* x, const double* u, const orc_spline* sp, double* f) {
  if (model == ORC_MODEL_KINEMATIC) orc_f_kin(x, u, sp, f); else orc_f_dyn(x, u, sp, f, 0);
}
static void model_A(int model, const double* x, const orc_spline* sp, double* A) {
  if (model == ORC_MODEL_KINEMATIC) orc_A_kin(x, sp, A); else orc_A_dyn(x, sp, A, 0);
}
static void model_B(int model, double* B) {
  /* B_curv_kin.m:12-16 ; B_curv_dyn.m:12-18 (constant) */
  int nx = orc_nx(model);
  memset(B, 0, (size_t)nx * 2 * sizeof(double));
  B[IDX(3, 0, nx)] = 1;
  B[IDX(nx - 1, 1, nx)] = 1;
}

/* ---- linearisers ---------------------------------------------------------- */
void orc_linearise(int model, int integrator, int N, const double* x, const double* u,
                   const orc_spline* sp, double dt, double* A, double* B, double* d) {
  int nx = orc_nx(model), nn = nx * nx;
  double k1[7], k2[7], k3[7], k4[7], xs[7], f[7];
  double F1[49], F2[49], F3[49], F4[49], K1[49], K2[49], K3[49], K4[49], T[49], T2[49];
  double Bc[14], U1[14], U2[14], U3[14], U4[14], TB[14];
  model_B(model, Bc);
  for (int i = 0; i < N; ++i) {
    const double* xi = x + (size_t)i * nx;
    const double* ui = u + (size_t)i * 2;
    double* Ai = A + (size_t)i * nn;
    double* Bi = B + (size_t)i * nx * 2;
    double* di = d + (size_t)i * nx;
    if (integrator == ORC_INT_EULER) {
      /* euler_*_curvilinear.m:24-30 : A=dfdx, B=dfdu, d=f-Ax-Bu */
      model_f(model, xi, ui, sp, f);
      model_A(model, xi, sp, Ai);
      memcpy(Bi, Bc, sizeof(double) * nx * 2);
    } else if (integrator == ORC_INT_RK2) {
      /* rk2_*_curvilinear.m:25-50 (midpoint rule; doc-comment says RK4) */
      model_f(model, xi, ui, sp, k1);
      for (int j = 0; j < nx; ++j) xs[j] = xi[j] + k1[j] * dt / 2;
      model_f(model, xs, ui, sp, k2);
      memcpy(f, k2, sizeof(double) * nx);
      model_A(model, xi, sp, F1);
      model_A(model, xs, sp, F2);
      /* dkdx2 = dfdx2*(I + dkdx1*dt/2) */
      for (int j = 0; j < nn; ++j) T[j] = F1[j] * dt / 2;
      for (int j = 0; j < nx; ++j) T[IDX(j, j, nx)] += 1;
      mm(nx, nx, nx, F2, T, Ai);
      /* dkdu2 = B + dfdx2*dkdu1*dt/2 */
      mm(nx, nx, 2, F2, Bc, TB);
      for (int j = 0; j < nx * 2; ++j) Bi[j] = Bc[j] + TB[j] * dt / 2;
    } else {
      /* rk4_*_curvilinear.m:25-59 */
      model_f(model, xi, ui, sp, k1);
      for (int j = 0; j < nx; ++j) xs[j] = xi[j] + k1[j] * dt / 2;
      model_f(model, xs, ui, sp, k2);
      model_A(model, xs, sp, F2);
      for (int j = 0; j < nx; ++j) xs[j] = xi[j] + k2[j] * dt / 2;
      model_f(model, xs, ui, sp, k3);
      model_A(model, xs, sp, F3);
      for (int j = 0; j < nx; ++j) xs[j] = xi[j] + k3[j] * dt;
      model_f(model, xs, ui, sp, k4);
      model_A(model, xs, sp, F4);
      model_A(model, xi, sp, F1);
      for (int j = 0; j < nx; ++j) f[j] = (k1[j] + 2 * k2[j] + 2 * k3[j] + k4[j]) / 6;
      memcpy(K1, F1, sizeof(double) * nn);
      for (int j = 0; j < nn; ++j) T[j] = K1[j] * dt / 2;
      for (int j = 0; j < nx; ++j) T[IDX(j, j, nx)] += 1;
      mm(nx, nx, nx, F2, T, K2);
      for (int j = 0; j < nn; ++j) T[j] = K2[j] * dt / 2;
      for (int j = 0; j < nx; ++j) T[IDX(j, j, nx)] += 1;
      mm(nx, nx, nx, F3, T, K3);
      for (int j = 0; j < nn; ++j) T[j] = K3[j] * dt;
      for (int j = 0; j < nx; ++j) T[IDX(j, j, nx)] += 1;
      mm(nx, nx, nx, F4, T, K4);
      (void)T2;
      memcpy(U1, Bc, sizeof(double) * nx * 2);
      mm(nx, nx, 2, F2, U1, TB);
      for (int j = 0; j < nx * 2; ++j) U2[j] = Bc[j] + TB[j] * dt / 2;
      mm(nx, nx, 2, F3, U2, TB);
      for (int j = 0; j < nx * 2; ++j) U3[j] = Bc[j] + TB[j] * dt / 2;
      mm(nx, nx, 2, F4, U3, TB);
      for (int j = 0; j < nx * 2; ++j) U4[j] = Bc[j] + TB[j] * dt / 2; /* rk4_*.m:52 uses dt/2 (quirk C-3) */
      for (int j = 0; j < nn; ++j) Ai[j] = (K1[j] + 2 * K2[j] + 2 * K3[j] + K4[j]) / 6;
      for (int j = 0; j < nx * 2; ++j) Bi[j] = (U1[j] + 2 * U2[j] + 2 * U3[j] + U4[j]) / 6;
    }
    /* d = f - A x - B u */
    for (int r = 0; r < nx; ++r) {
      double s = f[r];
      for (int c = 0; c < nx; ++c) s -= Ai[IDX(r, c, nx)] * xi[c];
      for (int c = 0; c < 2; ++c) s -= Bi[IDX(r, c, nx)] * ui[c];
      di[r] = s;
    }
  }
}

/* ---- sequential_integration.m:16-47 --------------------------------------- */
void orc_sequential_integration(int nx, int N, const double* A, const double* B, const double* d,
                                double dt, double* A_bar, double* B_bar, double* d_bar) {
  int nn = nx * nx, R = nx * N, C = 2 * N;
  double* Ad = (double*)malloc(sizeof(double) * nn * N);
  double* Bd = (double*)malloc(sizeof(double) * nx * 2 * N);
  double* dd = (double*)malloc(sizeof(double) * nx * N);
  double* D = (double*)calloc((size_t)R * R, sizeof(double));
  double blk[49], prev[49];
  /* :16-18 Euler integration */
  for (int k = 0; k < N; ++k) {
    for (int j = 0; j < nn; ++j) Ad[k * nn + j] = A[k * nn + j] * dt;
    for (int j = 0; j < nx; ++j) Ad[k * nn + IDX(j, j, nx)] += 1;
    for (int j = 0; j < nx * 2; ++j) Bd[k * nx * 2 + j] = B[k * nx * 2 + j] * dt;
    for (int j = 0; j < nx; ++j) dd[k * nx + j] = d[k * nx + j] * dt;
  }
  /* :21-26 A_bar */
  memset(A_bar, 0, sizeof(double) * R * nx);
  for (int c = 0; c < nx; ++c)
    for (int r = 0; r < nx; ++r) A_bar[IDX(r, c, R)] = Ad[IDX(r, c, nx)];
  for (int i = 1; i < N; ++i) {
    for (int c = 0; c < nx; ++c)
      for (int r = 0; r < nx; ++r) prev[IDX(r, c, nx)] = A_bar[IDX((i - 1) * nx + r, c, R)];
    mm(nx, nx, nx, Ad + (size_t)i * nn, prev, blk);
    for (int c = 0; c < nx; ++c)
      for (int r = 0; r < nx; ++r) A_bar[IDX(i * nx + r, c, R)] = blk[IDX(r, c, nx)];
  }
  /* :28-36 B_bar ; diagonal block is ALWAYS B(:,:,1) (quirk C-1) */
  memset(B_bar, 0, sizeof(double) * (size_t)R * C);
  for (int i = 0; i < N; ++i) {
    for (int c = 0; c < 2; ++c)
      for (int r = 0; r < nx; ++r) B_bar[IDX(i * nx + r, i * 2 + c, R)] = Bd[IDX(r, c, nx)];
    for (int j = i + 1; j < N; ++j) {
      double pb[14], nb[14];
      for (int c = 0; c < 2; ++c)
        for (int r = 0; r < nx; ++r) pb[IDX(r, c, nx)] = B_bar[IDX((j - 1) * nx + r, i * 2 + c, R)];
      mm(nx, nx, 2, Ad + (size_t)j * nn, pb, nb);
      for (int c = 0; c < 2; ++c)
        for (int r = 0; r < nx; ++r) B_bar[IDX(j * nx + r, i * 2 + c, R)] = nb[IDX(r, c, nx)];
    }
  }
  /* :38-47 D and d_bar = D*d(:) */
  for (int i = 0; i < N; ++i) {
    for (int r = 0; r < nx; ++r) D[IDX(i * nx + r, i * nx + r, R)] = 1;
    for (int j = i + 1; j < N; ++j) {
      for (int c = 0; c < nx; ++c)
        for (int r = 0; r < nx; ++r) prev[IDX(r, c, nx)] = D[IDX((j - 1) * nx + r, i * nx + c, R)];
      mm(nx, nx, nx, Ad + (size_t)j * nn, prev, blk);
      for (int c = 0; c < nx; ++c)
        for (int r = 0; r < nx; ++r) D[IDX(j * nx + r, i * nx + c, R)] = blk[IDX(r, c, nx)];
    }
  }
  for (int r = 0; r < R; ++r) {
    double s = 0;
    for (int c = 0; c < R; ++c) s += D[IDX(r, c, R)] * dd[c];
    d_bar[r] = s;
  }
  free(Ad); free(Bd); free(dd); free(D);
}

/* ---- constraints + generate_qp -------------------------------------------- */
/* rows of Bt (R x nV) selected by state index `idx` (0-based) for k=0..N-1 -> A rows [row0 .. row0+N) */
static void gather_rows(int nx, int N, int nV, int nC, const double* Bt, int idx, double* A, int row0) {
  int R = nx * N;
  for (int c = 0; c < nV; ++c)
    for (int k = 0; k < N; ++k) A[IDX(row0 + k, c, nC)] = Bt[IDX(k * nx + idx, c, R)];
}

void orc_ltv_build_qp(int model, int integrator, int N, double dt, const orc_spline* sp,
                      const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                      double* H, double* g, double* A, double* lb, double* ub, double* lbA, double* ubA,
                      double* A_bar, double* Bt, double* d_bar, double* qconst) {
  const int nx = orc_nx(model), ns = orc_ns(model), nV = orc_nV(model, N), nC = orc_nC(model, N);
  const int R = nx * N, nu = 2;
  const double INF = INFINITY;
  if (integrator < 0) integrator = (model == ORC_MODEL_KINEMATIC) ? ORC_INT_RK2 : ORC_INT_RK4; /* ltvmpc_*.m:38 */
  double* Al = (double*)malloc(sizeof(double) * nx * nx * N);
  double* Bl = (double*)malloc(sizeof(double) * nx * 2 * N);
  double* dl_ = (double*)malloc(sizeof(double) * nx * N);
  double* Bb = (double*)malloc(sizeof(double) * (size_t)R * 2 * N);
  double* aff = (double*)malloc(sizeof(double) * R); /* A_bar*x0 + d_bar */
  orc_linearise(model, integrator, N, x_lin, u_lin, sp, dt, Al, Bl, dl_);
  orc_sequential_integration(nx, N, Al, Bl, dl_, dt, A_bar, Bb, d_bar);
  /* *_state_constraints.m:11 : append slack columns */
  memset(Bt, 0, sizeof(double) * (size_t)R * nV);
  memcpy(Bt, Bb, sizeof(double) * (size_t)R * 2 * N);
  for (int r = 0; r < R; ++r) {
    double s = d_bar[r];
    for (int c = 0; c < nx; ++c) s += A_bar[IDX(r, c, R)] * x0[c];
    aff[r] = s;
  }
  /* variable bounds: ltvmpc_*.m:28-29 */
  for (int k = 0; k < N; ++k) {
    lb[2 * k] = -10; ub[2 * k] = 10; lb[2 * k + 1] = -0.4; ub[2 * k + 1] = 0.4;
  }
  for (int s = 0; s < ns; ++s) { lb[2 * N + s] = 0; ub[2 * N + s] = INF; }

  memset(A, 0, sizeof(double) * (size_t)nC * nV);
  /* state_idx = [4, nx] (1-based), soft_idx = 2 : ltvmpc_*.m:20-24 ; *_state_constraints.m:14-42 */
  const int vidx = 3, didx = nx - 1, nidx = 1;
  const int scol = 2 * N; /* first slack column: kin `end`, dyn `end-3` */
  gather_rows(nx, N, nV, nC, Bt, vidx, A, 0);
  gather_rows(nx, N, nV, nC, Bt, didx, A, N);
  gather_rows(nx, N, nV, nC, Bt, nidx, A, 2 * N);
  gather_rows(nx, N, nV, nC, Bt, nidx, A, 3 * N);
  for (int k = 0; k < N; ++k) {
    double cv = aff[k * nx + vidx], cd = aff[k * nx + didx], cn = aff[k * nx + nidx];
    lbA[k] = 0 - cv;            ubA[k] = INF;             /* v in [0, inf) */
    lbA[N + k] = -0.4 - cd;     ubA[N + k] = 0.4 - cd;    /* delta */
    lbA[2 * N + k] = -0.75 - cn; ubA[2 * N + k] = 1e10;   /* n + s >= -0.75 */
    lbA[3 * N + k] = -1e10;     ubA[3 * N + k] = 0.75 - cn; /* n - s <= 0.75 */
    A[IDX(2 * N + k, scol, nC)] = 1;
    A[IDX(3 * N + k, scol, nC)] = -1;
  }
  if (model == ORC_MODEL_KINEMATIC) {
    /* kinematic_tyre_linearise_constraints.m:18-32 ; kinematic_state_constraints.m:44-48 */
    for (int k = 0; k < N; ++k) {
      const double* xl = x_lin + (size_t)k * nx;
      double g0 = xl[3] * xl[3] * xl[4] / (LR + LF);
      double C3 = 2 * xl[3] * xl[4] / (LF + LR), C4 = xl[3] * xl[3] / (LF + LR);
      double cst = g0 + C3 * (aff[k * nx + 3] - xl[3]) + C4 * (aff[k * nx + 4] - xl[4]);
      for (int c = 0; c < nV; ++c) {
        double v = C3 * Bt[IDX(k * nx + 3, c, R)] + C4 * Bt[IDX(k * nx + 4, c, R)];
        A[IDX(4 * N + k, c, nC)] = v;
        A[IDX(5 * N + k, c, nC)] = v;
      }
      lbA[4 * N + k] = -5.0 - cst; ubA[4 * N + k] = INF;
      lbA[5 * N + k] = -INF;       ubA[5 * N + k] = 5.0 - cst;
      A[IDX(4 * N + k, scol, nC)] = 1;  /* shared slack (quirk C-7) */
      A[IDX(5 * N + k, scol, nC)] = -1;
    }
  } else {
    /* dynamic_slip_linearise_constraints.m:20-44 ; dynamic_state_constraints.m:46-50 */
    const int r5 = 4 * N, r6 = 6 * N, r7 = 8 * N;
    for (int k = 0; k < N; ++k) {
      const double* xl = x_lin + (size_t)k * nx;
      const double* ul = u_lin + (size_t)k * 2;
      double byp[8];
      orc_A_dyn(xl, sp, 0, byp);
      double Fcr = byp[0], Fcr_d = byp[1], vr = byp[2], dvr2 = byp[3], xh = byp[4], xhd = byp[5], vf = byp[6], dvf2 = byp[7];
      double g0[2] = {-atan(vr), xl[6] - atan(vf)};
      double Cs[2][7] = {{0, 0, 0, dvr2 * vr * xhd / xh, -dvr2 / xh, dvr2 * LR / xh, 0},
                         {0, 0, 0, dvf2 * vf * xhd / xh, -dvf2 / xh, -dvf2 * LF / xh, 1}};
      for (int q = 0; q < 2; ++q) {
        double cst = g0[q];
        for (int j = 0; j < nx; ++j) cst += Cs[q][j] * (aff[k * nx + j] - xl[j]);
        for (int c = 0; c < nV; ++c) {
          double v = 0;
          for (int j = 3; j < nx; ++j) v += Cs[q][j] * Bt[IDX(k * nx + j, c, R)];
          A[IDX(r5 + 2 * k + q, c, nC)] = v;
          A[IDX(r6 + 2 * k + q, c, nC)] = v;
        }
        lbA[r5 + 2 * k + q] = -0.1 - cst; ubA[r5 + 2 * k + q] = INF;
        lbA[r6 + 2 * k + q] = -INF;       ubA[r6 + 2 * k + q] = 0.1 - cst;
        A[IDX(r5 + 2 * k + q, scol + 1 + q, nC)] = 1;   /* end-2:end-1 */
        A[IDX(r6 + 2 * k + q, scol + 1 + q, nC)] = -1;
      }
      /* dynamic_tyre_linearise_constraints.m:18-61 ; dynamic_state_constraints.m:53-57 */
      const int NP = 12;
      const double ac_max = 9.163, al_max = 10.0;
      for (int j = 0; j < NP; ++j) {
        /* linspace(0,2*pi,13) */
        double th0 = 2 * M_PI * (double)j / NP, th1 = (j + 1 == NP) ? 2 * M_PI : 2 * M_PI * (double)(j + 1) / NP;
        double ac0 = ac_max * sin(th0), ac1 = ac_max * sin(th1);
        double al0 = al_max * cos(th0), al1 = al_max * cos(th1);
        double dac = ac1 - ac0, dal = al1 - al0;
        double g0t = (ul[0] - al0) * dac - (Fcr / 280 - ac0) * dal;
        double Ct[7] = {0, 0, 0, -dal * Fcr_d * dvr2 * vr * xhd / xh / 280, dal * Fcr_d * dvr2 / xh / 280,
                        -dal * Fcr_d * dvr2 * LR / xh / 280, 0};
        double cst = g0t;
        for (int jj = 3; jj < nx; ++jj) cst += Ct[jj] * (aff[k * nx + jj] - xl[jj]);
        cst -= dac * ul[0]; /* - D_bar*u_lin(:) */
        int row = r7 + NP * k + j;
        for (int c = 0; c < nV; ++c) {
          double v = 0;
          for (int jj = 3; jj < nx; ++jj) v += Ct[jj] * Bt[IDX(k * nx + jj, c, R)];
          A[IDX(row, c, nC)] = v;
        }
        A[IDX(row, 2 * k, nC)] += dac;   /* + [D_bar, 0] */
        A[IDX(row, scol + 3, nC)] = -1;  /* `end` slack */
        lbA[row] = -INF; ubA[row] = 0 - cst;
      }
    }
  }
  /* generate_qp.m:23-33 ; weights ltvmpc_*.m:32-35 */
  double Q[7] = {5, 250, 2000, 0, 0, 0, 0};
  double Rw[2] = {10, 10};
  double Rs_kin[1] = {1e8}, Rs_dyn[4] = {1e8, 1e6, 1e6, 1e4};
  const double* Rsoft = (model == ORC_MODEL_KINEMATIC) ? Rs_kin : Rs_dyn;
  double* qb = (double*)malloc(sizeof(double) * R);
  double* rr = (double*)malloc(sizeof(double) * R);
  for (int k = 0; k < N; ++k)
    for (int j = 0; j < nx; ++j) {
      qb[k * nx + j] = (k == N - 1) ? Q[j] * 10 : Q[j];
      rr[k * nx + j] = aff[k * nx + j] - x_ref[k * nx + j];
    }
  for (int j = 0; j < nV; ++j)
    for (int i = 0; i < nV; ++i) {
      double s = 0;
      for (int r = 0; r < R; ++r) s += Bt[IDX(r, i, R)] * qb[r] * Bt[IDX(r, j, R)];
      if (i == j && i < nu * N) s += Rw[i % 2];
      H[IDX(i, j, nV)] = 2 * s;
    }
  for (int i = 0; i < nV; ++i) {
    double s = 0;
    for (int r = 0; r < R; ++r) s += Bt[IDX(r, i, R)] * qb[r] * rr[r];
    g[i] = 2 * s;
  }
  for (int s = 0; s < ns; ++s) g[2 * N + s] = Rsoft[s];
  double cst = 0;
  for (int r = 0; r < R; ++r) cst += rr[r] * qb[r] * rr[r];
  *qconst = cst;
  free(Al); free(Bl); free(dl_); free(Bb); free(aff); free(qb); free(rr);
}

/* ---- main.m:107-114 ------------------------------------------------------- */
void orc_reference_live(int nx, int N, double dt, double target_vel, const double* x0, double* x_ref) {
  memset(x_ref, 0, sizeof(double) * nx * N);
  double cum = 0;
  for (int k = 0; k < N; ++k) {
    double v;
    if (x0[3] < target_vel) { v = x0[3] + 10 * dt * (k + 1); if (v > target_vel) v = target_vel; }
    else                    { v = x0[3] - 10 * dt * (k + 1); if (v < target_vel) v = target_vel; }
    x_ref[k * nx + 3] = v;
    cum += v * dt;
    x_ref[k * nx + 0] = x0[0] + cum;
  }
}

/* ---- util/obtain_reference.m:5-48 ------------------------------------------ */
static int nxt1(int i, int N) { return (i % N) + 1; }                       /* nxt(): mod(i, N) + 1, 1-based (:55-57) */
static double mmod(double a, double b) { return a - floor(a / b) * b; }     /* MATLAB mod, b > 0 */
void orc_obtain_reference(const double* x, double ds, int N_s, const double* t, double s0, double dt, int N_t, double* x_ref) {
  const double L = ds * N_s;                                                /* :5 */
  int* idx = (int*)malloc(sizeof(int) * (N_t + 2));
  double* rto = (double*)malloc(sizeof(double) * (N_t + 2));
  idx[1] = (int)floor(mmod(s0, L) / ds) + 1;                                /* :21 */
  rto[1] = mmod(mmod(s0, L) / ds, 1.0);                                     /* :22 */
  for (int i = 2; i <= N_t + 1; ++i) {                                      /* :24-35 */
    double t_remaining = dt;
    idx[i] = idx[i - 1];
    rto[i] = rto[i - 1] + t_remaining / t[idx[i] - 1];
    t_remaining = t_remaining - t[idx[i - 1] - 1] * (1 - rto[i - 1]);
    while (rto[i] > 1) {
      idx[i] = nxt1(idx[i], N_s);
      rto[i] = t_remaining / t[idx[i] - 1];
      t_remaining = t_remaining - t[idx[i] - 1];
    }
  }
  for (int i = 2; i <= N_t + 1; ++i) {                                      /* :40-48 */
    double* col = x_ref + (size_t)(i - 2) * 7;
    const int a = idx[i] - 1, b = nxt1(idx[i], N_s) - 1;
    col[0] = s0 + mmod(idx[i] + rto[i] - idx[1] - rto[1], (double)N_s) * ds;
    for (int c = 0; c < 6; ++c) col[1 + c] = x[8 * a + c] + (x[8 * b + c] - x[8 * a + c]) * rto[i];
  }
  free(idx); free(rto);
}

/* ---- SURVEY 8(d) synthetic instances -------------------------------------- */
static unsigned long long sm64_next(unsigned long long* s) {
  unsigned long long z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static double sm64_u01(unsigned long long* s) { return (double)(sm64_next(s) >> 11) * (1.0 / 9007199254740992.0); }

void orc_synth_instance(int model, int N, double dt, double L, unsigned long long seed, unsigned long long id,
                        double* x0, double* x_lin, double* u_lin, double* x_ref) {
  int nx = orc_nx(model);
  unsigned long long st = seed ^ (id * 0x9E3779B97F4A7C15ULL);
  double s0 = sm64_u01(&st) * L;
  double n0 = -0.5 + sm64_u01(&st);
  double mu0 = -0.1 + 0.2 * sm64_u01(&st);
  double v0 = 5 + 15 * sm64_u01(&st);
  double d0 = -0.1 + 0.2 * sm64_u01(&st);
  if (model == ORC_MODEL_KINEMATIC) {
    x0[0] = s0; x0[1] = n0; x0[2] = mu0; x0[3] = v0; x0[4] = d0;
  } else {
    double yd = -0.2 + 0.4 * sm64_u01(&st);
    double td = -0.3 + 0.6 * sm64_u01(&st);
    x0[0] = s0; x0[1] = n0; x0[2] = mu0; x0[3] = v0; x0[4] = yd; x0[5] = td; x0[6] = d0;
  }
  for (int k = 0; k < N; ++k) {
    for (int j = 0; j < nx; ++j) x_lin[k * nx + j] = x0[j];
    x_lin[k * nx + 0] = s0 + v0 * dt * k;
    u_lin[2 * k] = 0; u_lin[2 * k + 1] = 0;
  }
  orc_reference_live(nx, N, dt, 20.0, x0, x_ref);
}

int orc_ltv_build_qp_batch(int model, int N, double dt, const orc_spline* sp, int batch,
                           const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                           double* H, double* g, double* A, double* lb, double* ub, double* lbA, double* ubA,
                           double* A_bar, double* Bt, double* d_bar, double* qconst, int threads) {
  const int nx = orc_nx(model), nV = orc_nV(model, N), nC = orc_nC(model, N), R = nx * N;
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
  used = omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < batch; ++b) {
    double* Ab = A_bar ? A_bar + (size_t)b * R * nx : (double*)malloc(sizeof(double) * R * nx);
    double* Btb = Bt ? Bt + (size_t)b * R * nV : (double*)malloc(sizeof(double) * (size_t)R * nV);
    double* db = d_bar ? d_bar + (size_t)b * R : (double*)malloc(sizeof(double) * R);
    double qc;
    orc_ltv_build_qp(model, -1, N, dt, sp, x0 + (size_t)b * nx, x_ref + (size_t)b * R, x_lin + (size_t)b * R,
                     u_lin + (size_t)b * 2 * N, H + (size_t)b * nV * nV, g + (size_t)b * nV, A + (size_t)b * nC * nV,
                     lb + (size_t)b * nV, ub + (size_t)b * nV, lbA + (size_t)b * nC, ubA + (size_t)b * nC, Ab, Btb, db, &qc);
    if (qconst) qconst[b] = qc;
    if (!A_bar) free(Ab);
    if (!Bt) free(Btb);
    if (!d_bar) free(db);
  }
  return used;
}
