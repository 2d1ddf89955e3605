// ltv_build.hip -- QP construction of one LTV-MPC step, batched, on MI355X (gfx950).
//
// Replaces (reference file:line):
//   spline/interpolate_spline_d.m:11-21, interpolate_spline_dd.m:11-21, interpolate_curvature.m:12-18   (kappa)
//   vehicle_models/curvilinear_kinematic/{f,A,B}_curv_kin.m, vehicle_models/curvilinear_dynamic/{f,A,B}_curv_dyn.m
//   mpc/ltv/kinematic/rk2_kinematic_curvilinear.m:25-50, mpc/ltv/dynamic/rk4_dynamic_curvilinear.m:25-59
//   mpc/ltv/sequential_integration.m:16-47 (condensing; the prediction offset A_bar*x0+d_bar is produced by
//        the equivalent one-step recursion instead of materialising the dense D matrix of :38-47)
//   mpc/ltv/kinematic/kinematic_state_constraints.m:11-48, kinematic_tyre_linearise_constraints.m:18-32
//   mpc/ltv/dynamic/dynamic_state_constraints.m:11-57, dynamic_slip_linearise_constraints.m:20-44,
//        dynamic_tyre_linearise_constraints.m:18-61
//   mpc/ltv/generate_qp.m:23-33 with the weights/limits of ltvmpc_*_curvilinear.m:20-35
// One 256-thread workgroup per instance.  Reference quirks are preserved: the diagonal block
// of B_bar is always slice 1, RK4's dkdu4 uses dt/2, A_curv_dyn is the Jacobian definition, shared kinematic slack.
#include <hip/hip_runtime.h>
#include <math.h>
#include "ltv_build.h"

namespace {

#define DEVINL __device__ __forceinline__
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr double LR = 0.6183, LF = 0.8672;
constexpr double VM = 280, VI = 200, GRAV = 9.81;
constexpr double PB = 12.56, PC = 1.38, PD = 1.60, PE = -0.58;

struct Spl { int M; double dl; const double* xP; const double* yP; };

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
DEVINL double kappa(const Spl& sp, double s) {
  int i; double u;
  seg_lookup(sp.M, sp.dl, s, i, u);
  const int M = sp.M;
  const double x0 = sp.xP[i], x1 = sp.xP[i + M], x2 = sp.xP[i + 2 * M], x3 = sp.xP[i + 3 * M];
  const double y0 = sp.yP[i], y1 = sp.yP[i + M], y2 = sp.yP[i + 2 * M], y3 = sp.yP[i + 3 * M];
  const double b0 = -3 * (1 - u) * (1 - u), b1 = 3 * (3 * u * u - 4 * u + 1), b2 = 3 * (2 * u - 3 * u * u), b3 = 3 * u * u;
  const double c0 = 6 * (1 - u), c1 = 6 * (3 * u - 2), c2 = 6 * (1 - 3 * u), c3 = 6 * u;
  const double Xd = (b0 * x0 + b1 * x1 + b2 * x2 + b3 * x3) / sp.dl, Yd = (b0 * y0 + b1 * y1 + b2 * y2 + b3 * y3) / sp.dl;
  const double Xdd = (c0 * x0 + c1 * x1 + c2 * x2 + c3 * x3) / (sp.dl * sp.dl), Ydd = (c0 * y0 + c1 * y1 + c2 * y2 + c3 * y3) / (sp.dl * sp.dl);
  return (Xd * Ydd - Xdd * Yd) / pow(Xd * Xd + Yd * Yd, 1.5);
}

// ---- kinematic model (f_curv_kin.m:13-29, A_curv_kin.m:15-55) ----
DEVINL void f_kin(const double* x, const double* u, const Spl& sp, double* f) {
  const double lr_ratio = LR / (LR + LF);
  const double k = kappa(sp, x[0]);
  const double beta = atan(lr_ratio * tan(x[4]));
  const double s_mb = sin(x[2] + beta), c_mb = cos(x[2] + beta);
  const double denom_nk = 1.0 / (1.0 - x[1] * k);
  f[0] = x[3] * c_mb * denom_nk;
  f[1] = x[3] * s_mb;
  f[2] = x[3] * sin(beta) / LR - x[3] * c_mb * denom_nk * k;
  f[3] = u[0];
  f[4] = u[1];
}
DEVINL void A_kin(const double* x, const Spl& sp, double* A) {  // 5x5 column-major
  const double lr_ratio = LR / (LR + LF);
  const double k = kappa(sp, x[0]);
  const double td = tan(x[4]);
  const double beta = atan(lr_ratio * td);
  const double s_mb = sin(x[2] + beta), c_mb = cos(x[2] + beta);
  const double sec = 1.0 / cos(x[4]);
  const double beta_d = lr_ratio * sec * sec / (1 + (lr_ratio * td) * (lr_ratio * td));
  const double denom_nk = 1.0 / (1.0 - x[1] * k);
  const double s_n = x[3] * c_mb * denom_nk * denom_nk * k;
  const double s_mu = -x[3] * s_mb * denom_nk;
  const double s_v = c_mb * denom_nk;
  const double s_delta = -x[3] * s_mb * denom_nk * beta_d;
  for (int i = 0; i < 25; ++i) A[i] = 0.0;
  A[0 + 1 * 5] = s_n; A[0 + 2 * 5] = s_mu; A[0 + 3 * 5] = s_v; A[0 + 4 * 5] = s_delta;
  A[1 + 2 * 5] = x[3] * c_mb; A[1 + 3 * 5] = s_mb; A[1 + 4 * 5] = x[3] * c_mb * beta_d;
  A[2 + 1 * 5] = -s_n * k; A[2 + 2 * 5] = -s_mu * k; A[2 + 3 * 5] = sin(beta) / LR - s_v * k;
  A[2 + 4 * 5] = x[3] * cos(beta) * beta_d / LR - s_delta * k;
}

// ---- dynamic model (f_curv_dyn.m:13-62, A_curv_dyn.m:15-106) ----
DEVINL void f_dyn(const double* x, const double* u, const Spl& sp, double* f) {
  const double n = x[1], mu = x[2], x_d = x[3], y_d = x[4], th_d = x[5], delta = x[6];
  const double Fx = u[0] * VM;
  const double x_d_hat = x_d + 5 * exp(-x_d / 5);
  const double k = kappa(sp, x[0]);
  const double denom_nk = 1.0 / (1.0 - n * k);
  const double alpha_f = delta - atan((y_d + LF * th_d) / x_d_hat);
  const double alpha_r = -atan((y_d - LR * th_d) / x_d_hat);
  const double Fzf = VM * GRAV * LR / (LR + LF), Fzr = VM * GRAV * LF / (LR + LF);
  const double Fcf = Fzf * PD * sin(PC * atan(PB * alpha_f - PE * (PB * alpha_f - atan(PB * alpha_f))));
  const double Fcr = Fzr * PD * sin(PC * atan(PB * alpha_r - PE * (PB * alpha_r - atan(PB * alpha_r))));
  f[0] = (x_d * cos(mu) - y_d * sin(mu)) * denom_nk;
  f[1] = x_d * sin(mu) + y_d * cos(mu);
  f[2] = th_d - (x_d * cos(mu) - y_d * sin(mu)) * denom_nk * k;
  f[3] = (Fx - Fcf * sin(delta) + VM * y_d * th_d) / VM;
  f[4] = (Fcr + Fcf * cos(delta) - VM * x_d * th_d) / VM;
  f[5] = (LF * Fcf * cos(delta) - LR * Fcr) / VI;
  f[6] = u[1];
}
// byp = {Fcr, Fcr_d, vr, denom_vr2, x_d_hat, x_d_hat_d, vf, denom_vf2}; A may be null
DEVINL void A_dyn(const double* x, const Spl& sp, double* A, double* byp) {
  const double n = x[1], mu = x[2], x_d = x[3], y_d = x[4], th_d = x[5], delta = x[6];
  const double m = VM, I = VI;
  const double x_d_hat = x_d + 5 * exp(-x_d / 5);
  const double x_d_hat_d = 1 - exp(-x_d / 5);
  const double alpha_f = delta - atan((y_d + LF * th_d) / x_d_hat);
  const double alpha_r = -atan((y_d - LR * th_d) / x_d_hat);
  const double Fzf = m * GRAV * LR / (LR + LF), Fzr = m * GRAV * LF / (LR + LF);
  const double af_arg = PB * alpha_f - PE * (PB * alpha_f - atan(PB * alpha_f));
  const double ar_arg = PB * alpha_r - PE * (PB * alpha_r - atan(PB * alpha_r));
  const double Fcf = Fzf * PD * sin(PC * atan(af_arg));
  const double Fcr = Fzr * PD * sin(PC * atan(ar_arg));
  const double Fcf_d = Fzf * PD * cos(PC * atan(af_arg)) * PC / (1 + af_arg * af_arg) * (PB - PE * (PB - PB / (1 + PB * PB * alpha_f * alpha_f)));
  const double Fcr_d = Fzr * PD * cos(PC * atan(ar_arg)) * PC / (1 + ar_arg * ar_arg) * (PB - PE * (PB - PB / (1 + PB * PB * alpha_r * alpha_r)));
  const double vf = (y_d + LF * th_d) / x_d_hat, vr = (y_d - LR * th_d) / x_d_hat;
  const double denom_vf2 = 1.0 / (1 + vf * vf), denom_vr2 = 1.0 / (1 + vr * vr);
  if (A) {
    const double k = kappa(sp, x[0]);
    const double denom_nk = 1.0 / (1.0 - n * k);
    const double cm = cos(mu), sm = sin(mu), cd = cos(delta), sd = sin(delta);
    const double s_n = (x_d * cm - y_d * sm) * denom_nk * denom_nk * k;
    const double s_mu = (-x_d * sm - y_d * cm) * denom_nk;
    const double s_xd = cm * denom_nk, s_yd = -sm * denom_nk;
    for (int i = 0; i < 49; ++i) A[i] = 0.0;
    A[0 + 1 * 7] = s_n; A[0 + 2 * 7] = s_mu; A[0 + 3 * 7] = s_xd; A[0 + 4 * 7] = s_yd;
    A[1 + 2 * 7] = x_d * cm - y_d * sm; A[1 + 3 * 7] = sm; A[1 + 4 * 7] = cm;
    A[2 + 1 * 7] = -s_n * k; A[2 + 2 * 7] = -s_mu * k; A[2 + 3 * 7] = -s_xd * k; A[2 + 4 * 7] = -s_yd * k; A[2 + 5 * 7] = 1;
    A[3 + 3 * 7] = -Fcf_d * denom_vf2 * vf * sd * x_d_hat_d / (m * x_d_hat);
    A[3 + 4 * 7] = (Fcf_d * denom_vf2 * sd / x_d_hat + m * th_d) / m;
    A[3 + 5 * 7] = (Fcf_d * denom_vf2 * LF * sd / x_d_hat + m * y_d) / m;
    A[3 + 6 * 7] = (-Fcf * cd - Fcf_d * sd) / m;
    A[4 + 3 * 7] = (Fcr_d * denom_vr2 * vr * x_d_hat_d / x_d_hat + Fcf_d * denom_vf2 * vf * cd * x_d_hat_d / x_d_hat - m * th_d) / m;
    A[4 + 4 * 7] = (-Fcr_d * denom_vr2 / x_d_hat - Fcf_d * denom_vf2 / x_d_hat * cd) / m;
    A[4 + 5 * 7] = (Fcr_d * denom_vr2 * LR / x_d_hat - Fcf_d * denom_vf2 * LF / x_d_hat * cd - m * x_d_hat) / m;
    A[4 + 6 * 7] = (-Fcf * sd + Fcf_d * cd) / m;
    A[5 + 3 * 7] = (LF * Fcf_d * denom_vf2 * vf * cd * x_d_hat_d / x_d_hat - LR * Fcr_d * denom_vr2 * vr * x_d_hat_d / x_d_hat) / I;
    A[5 + 4 * 7] = (-LF * Fcf_d * denom_vf2 * cd / x_d_hat + LR * Fcr_d * denom_vr2 / x_d_hat) / I;
    A[5 + 5 * 7] = (-LF * Fcf_d * denom_vf2 * LF * cd / x_d_hat - LR * Fcr_d * denom_vr2 * LR / x_d_hat) / I;
    A[5 + 6 * 7] = (-LF * Fcf * sd + LF * Fcf_d * cd) / I;
  }
  if (byp) { byp[0] = Fcr; byp[1] = Fcr_d; byp[2] = vr; byp[3] = denom_vr2; byp[4] = x_d_hat; byp[5] = x_d_hat_d; byp[6] = vf; byp[7] = denom_vf2; }
}

template <int NX> DEVINL void mmul(const double* A, const double* B, double* C, int ncol) {  // C = A(NXxNX) B(NX x ncol)
  for (int j = 0; j < ncol; ++j)
    for (int i = 0; i < NX; ++i) {
      double s = 0;
      for (int p = 0; p < NX; ++p) s += A[i + p * NX] * B[p + j * NX];
      C[i + j * NX] = s;
    }
}
template <int NX> DEVINL void model_f(const double* x, const double* u, const Spl& sp, double* f) {
  if (NX == 5) f_kin(x, u, sp, f); else f_dyn(x, u, sp, f);
}
template <int NX> DEVINL void model_A(const double* x, const Spl& sp, double* A) {
  if (NX == 5) A_kin(x, sp, A); else A_dyn(x, sp, A, nullptr);
}

// Linearise step k about (xi, ui): writes Ad = I + dt*A, Bd = dt*B, dd = dt*d   (sequential_integration.m:16-18)
// integ: 0 Euler (euler_*_curvilinear.m:24-30), 1 midpoint rule (rk2_*_curvilinear.m:25-50), 2 classical RK4
// (rk4_*_curvilinear.m:25-59).  The reference drivers use rk2 for the kinematic and rk4 for the dynamic model
// (ltvmpc_kinetmatic_curvilinear.m:38, ltvmpc_dynamic_curvilinear.m:38); the others are the alternates kept beside them.
template <int NX> DEVINL void linearise_step(const double* xi, const double* ui, const Spl& sp, double dt, int integ,
                                             double* Ad, double* Bd, double* dd) {
  constexpr int NN = NX * NX;
  double Bc[NX * 2];
  for (int i = 0; i < NX * 2; ++i) Bc[i] = 0.0;
  Bc[3] = 1.0; Bc[(NX - 1) + NX] = 1.0;  // B_curv_kin.m:12-16 / B_curv_dyn.m:12-18
  double f[NX], Ai[NN], Bi[NX * 2];
  if (integ == 0) {
    model_f<NX>(xi, ui, sp, f);
    model_A<NX>(xi, sp, Ai);
    for (int j = 0; j < NX * 2; ++j) Bi[j] = Bc[j];
  } else if (integ == 1) {
    double k1[NX], xs[NX], F1[NN], F2[NN], Tm[NN], TB[NX * 2];
    model_f<NX>(xi, ui, sp, k1);
    for (int j = 0; j < NX; ++j) xs[j] = xi[j] + k1[j] * dt / 2;
    model_f<NX>(xs, ui, sp, f);
    model_A<NX>(xi, sp, F1);
    model_A<NX>(xs, sp, F2);
    for (int j = 0; j < NN; ++j) Tm[j] = F1[j] * dt / 2;
    for (int j = 0; j < NX; ++j) Tm[j + j * NX] += 1;
    mmul<NX>(F2, Tm, Ai, NX);
    mmul<NX>(F2, Bc, TB, 2);
    for (int j = 0; j < NX * 2; ++j) Bi[j] = Bc[j] + TB[j] * dt / 2;
  } else {
    double k1[NX], k2[NX], k3[NX], k4[NX], xs[NX];
    double F[NN], K[NN], Tm[NN], Ks[NN], U[NX * 2], Us[NX * 2], TB[NX * 2];
    model_f<NX>(xi, ui, sp, k1);
    model_A<NX>(xi, sp, K);                       // dkdx1
    for (int j = 0; j < NN; ++j) Ks[j] = K[j];
    for (int j = 0; j < NX * 2; ++j) { U[j] = Bc[j]; Us[j] = Bc[j]; }
    for (int j = 0; j < NX; ++j) xs[j] = xi[j] + k1[j] * dt / 2;
    model_f<NX>(xs, ui, sp, k2);
    model_A<NX>(xs, sp, F);
    for (int j = 0; j < NN; ++j) Tm[j] = K[j] * dt / 2;
    for (int j = 0; j < NX; ++j) Tm[j + j * NX] += 1;
    mmul<NX>(F, Tm, K, NX);                        // dkdx2
    mmul<NX>(F, U, TB, 2);
    for (int j = 0; j < NX * 2; ++j) U[j] = Bc[j] + TB[j] * dt / 2;   // dkdu2
    for (int j = 0; j < NN; ++j) Ks[j] += 2 * K[j];
    for (int j = 0; j < NX * 2; ++j) Us[j] += 2 * U[j];
    for (int j = 0; j < NX; ++j) xs[j] = xi[j] + k2[j] * dt / 2;
    model_f<NX>(xs, ui, sp, k3);
    model_A<NX>(xs, sp, F);
    for (int j = 0; j < NN; ++j) Tm[j] = K[j] * dt / 2;
    for (int j = 0; j < NX; ++j) Tm[j + j * NX] += 1;
    mmul<NX>(F, Tm, K, NX);                        // dkdx3
    mmul<NX>(F, U, TB, 2);
    for (int j = 0; j < NX * 2; ++j) U[j] = Bc[j] + TB[j] * dt / 2;   // dkdu3
    for (int j = 0; j < NN; ++j) Ks[j] += 2 * K[j];
    for (int j = 0; j < NX * 2; ++j) Us[j] += 2 * U[j];
    for (int j = 0; j < NX; ++j) xs[j] = xi[j] + k3[j] * dt;
    model_f<NX>(xs, ui, sp, k4);
    model_A<NX>(xs, sp, F);
    for (int j = 0; j < NN; ++j) Tm[j] = K[j] * dt;
    for (int j = 0; j < NX; ++j) Tm[j + j * NX] += 1;
    mmul<NX>(F, Tm, K, NX);                        // dkdx4
    mmul<NX>(F, U, TB, 2);
    for (int j = 0; j < NX * 2; ++j) U[j] = Bc[j] + TB[j] * dt / 2;   // dkdu4: dt/2 as in rk4_*.m:52 (quirk C-3)
    for (int j = 0; j < NN; ++j) Ai[j] = (Ks[j] + K[j]) / 6;
    for (int j = 0; j < NX * 2; ++j) Bi[j] = (Us[j] + U[j]) / 6;
    for (int j = 0; j < NX; ++j) f[j] = (k1[j] + 2 * k2[j] + 2 * k3[j] + k4[j]) / 6;
  }
  for (int r = 0; r < NX; ++r) {
    double s = f[r];
    for (int c = 0; c < NX; ++c) s -= Ai[r + c * NX] * xi[c];
    for (int c = 0; c < 2; ++c) s -= Bi[r + c * NX] * ui[c];
    dd[r] = s * dt;
  }
  for (int j = 0; j < NN; ++j) Ad[j] = Ai[j] * dt;
  for (int j = 0; j < NX; ++j) Ad[j + j * NX] += 1;
  for (int j = 0; j < NX * 2; ++j) Bd[j] = Bi[j] * dt;
}

// ---------------------------------------------------------------------------------------------
template <int NX> __global__ __launch_bounds__(256) void ltv_build_kernel(LtvParams P) {
  constexpr int NN = NX * NX, NS = (NX == 5) ? 1 : 4, RPK = (NX == 5) ? 6 : 20;  // rows per step
  const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
  const int N = P.N, R = NX * N, nV = 2 * N + NS, nC = RPK * N;
  const double dt = P.dt;
  Spl sp{P.spM, P.spdl, P.xP, P.yP};
  const double* x0 = P.x0 + (size_t)b * NX;
  const double* x_ref = P.x_ref + (size_t)b * R;
  const double* x_lin = P.x_lin + (size_t)b * R;
  const double* u_lin = P.u_lin + (size_t)b * 2 * N;
  double* H = P.H + (size_t)b * nV * nV;
  double* g = P.g + (size_t)b * nV;
  double* A = P.A + (size_t)b * nC * nV;
  double* lb = P.lb + (size_t)b * nV; double* ub = P.ub + (size_t)b * nV;
  double* lbA = P.lbA + (size_t)b * nC; double* ubA = P.ubA + (size_t)b * nC;
  double* Bt = P.Bt + (size_t)b * R * nV;

  extern __shared__ double sm[];
  double* Ad = sm;                 // N * NN
  double* Bd = Ad + (size_t)N * NN;  // N * NX*2 (only slice 0 is used by the condensing, kept for clarity)
  double* dd = Bd + (size_t)N * NX * 2;  // N * NX
  double* aff = dd + (size_t)N * NX;     // R   : A_bar*x0 + d_bar
  double* cc = aff + R;                  // per-step constraint coefficient scratch: N * CW
  constexpr int CW = (NX == 5) ? 3 : (2 * 4 + 2 + 4 + 2);  // kin: C3,C4,const ; dyn: slip rows (2x4 coef + 2 const), tyre (4 coef K-part) + 2
  double* red = cc + (size_t)N * CW;     // reduction scratch (nth)
  double* ell = red + nth;               // 24: dac[12], dal[12] of the inscribed 12-gon (dynamic_tyre_linearise_constraints.m:33-39)
  double* colst = ell + 24;              // one column of Bt (R doubles) per wavefront: stage of step 4c

  if (tid < 12) {
    const int j = tid;
    const double th0 = 2 * M_PI * (double)j / 12, th1 = (j + 1 == 12) ? 2 * M_PI : 2 * M_PI * (double)(j + 1) / 12;
    ell[j] = 9.163 * sin(th1) - 9.163 * sin(th0); ell[12 + j] = 10.0 * cos(th1) - 10.0 * cos(th0);
  }
  // ---- 1. linearise every step (one thread per step) ----
  for (int k = tid; k < N; k += nth)
    linearise_step<NX>(x_lin + (size_t)k * NX, u_lin + (size_t)k * 2, sp, dt, P.integ, Ad + (size_t)k * NN, Bd + (size_t)k * NX * 2, dd + (size_t)k * NX);
  // zero Bt while the linearisation runs
  for (size_t i = tid; i < (size_t)R * nV; i += nth) Bt[i] = 0.0;
  __syncthreads();

  // ---- 2. prediction offset aff_k = Ad_k aff_{k-1} + dd_k, aff_0 = x0 (== A_bar*x0 + d_bar) ----
  if (tid == 0) {
    double cur[NX], nxt[NX];
    for (int j = 0; j < NX; ++j) cur[j] = x0[j];
    for (int k = 0; k < N; ++k) {
      const double* a = Ad + (size_t)k * NN;
      for (int r = 0; r < NX; ++r) {
        double s = dd[k * NX + r];
        for (int c = 0; c < NX; ++c) s += a[r + c * NX] * cur[c];
        nxt[r] = s;
      }
      for (int r = 0; r < NX; ++r) { cur[r] = nxt[r]; aff[k * NX + r] = nxt[r]; }
    }
  }
  // ---- 3. Phi columns: Phi(i,i) = Bd_1 (always slice 1, quirk C-1), Phi(j,i) = Ad_j Phi(j-1,i) ----
  for (int w = tid; w < 2 * N; w += nth) {
    const int i = w >> 1, col = w & 1;
    double cur[NX], nxt[NX];
    for (int r = 0; r < NX; ++r) cur[r] = Bd[r + col * NX];
    double* dst = Bt + (size_t)w * R;
    for (int r = 0; r < NX; ++r) dst[i * NX + r] = cur[r];
    for (int j = i + 1; j < N; ++j) {
      const double* a = Ad + (size_t)j * NN;
      for (int r = 0; r < NX; ++r) {
        double s = 0;
        for (int c = 0; c < NX; ++c) s += a[r + c * NX] * cur[c];
        nxt[r] = s;
      }
      for (int r = 0; r < NX; ++r) { cur[r] = nxt[r]; dst[j * NX + r] = nxt[r]; }
    }
  }
  // ---- 4a. per-step constraint coefficients ----
  for (int k = tid; k < N; k += nth) {
    const double* xl = x_lin + (size_t)k * NX;
    double* ck = cc + (size_t)k * CW;
    if (NX == 5) {
      // kinematic_tyre_linearise_constraints.m:18-32 ; g = v^2 delta/(lr+lf)
      ck[0] = 2 * xl[3] * xl[4] / (LF + LR);
      ck[1] = xl[3] * xl[3] / (LF + LR);
      ck[2] = xl[3] * xl[3] * xl[4] / (LR + LF);  // g0
    } else {
      double byp[8];
      A_dyn(xl, sp, nullptr, byp);
      const double Fcr = byp[0], Fcr_d = byp[1], vr = byp[2], dvr2 = byp[3], xh = byp[4], xhd = byp[5], vf = byp[6], dvf2 = byp[7];
      // dynamic_slip_linearise_constraints.m:26-30 : rows (alpha_r, alpha_f) coefficients on states 4..7
      ck[0] = dvr2 * vr * xhd / xh; ck[1] = -dvr2 / xh; ck[2] = dvr2 * LR / xh; ck[3] = 0.0;
      ck[4] = dvf2 * vf * xhd / xh; ck[5] = -dvf2 / xh; ck[6] = -dvf2 * LF / xh; ck[7] = 1.0;
      ck[8] = -atan(vr); ck[9] = xl[6] - atan(vf);
      // dynamic_tyre_linearise_constraints.m:41-49 : C_j = dal_j * ck[10..12] on states 4..6
      ck[10] = -Fcr_d * dvr2 * vr * xhd / xh / 280; ck[11] = Fcr_d * dvr2 / xh / 280; ck[12] = -Fcr_d * dvr2 * LR / xh / 280;
      ck[13] = Fcr; ck[14] = 0; ck[15] = 0;
    }
  }
  __syncthreads();

  // ---- 4b. variable bounds (ltvmpc_*.m:28-29) and constraint bounds ----
  for (int i = tid; i < nV; i += nth) {
    if (i < 2 * N) { lb[i] = (i & 1) ? -0.4 : -10.0; ub[i] = (i & 1) ? 0.4 : 10.0; }
    else { lb[i] = 0.0; ub[i] = INFINITY; }
  }
  const int vidx = 3, didx = NX - 1, nidx = 1, scol = 2 * N;
  for (int k = tid; k < N; k += nth) {
    const double cv = aff[k * NX + vidx], cd = aff[k * NX + didx], cn = aff[k * NX + nidx];
    lbA[k] = 0 - cv;              ubA[k] = INFINITY;
    lbA[N + k] = -0.4 - cd;       ubA[N + k] = 0.4 - cd;
    lbA[2 * N + k] = -0.75 - cn;  ubA[2 * N + k] = 1e10;    // *_state_constraints.m:38-39
    lbA[3 * N + k] = -1e10;       ubA[3 * N + k] = 0.75 - cn;
    const double* xl = x_lin + (size_t)k * NX;
    const double* ck = cc + (size_t)k * CW;
    if (NX == 5) {
      const double cst = ck[2] + ck[0] * (aff[k * NX + 3] - xl[3]) + ck[1] * (aff[k * NX + 4] - xl[4]);
      lbA[4 * N + k] = -5.0 - cst;  ubA[4 * N + k] = INFINITY;
      lbA[5 * N + k] = -INFINITY;   ubA[5 * N + k] = 5.0 - cst;
    } else {
      const double* ul = u_lin + (size_t)k * 2;
      for (int q = 0; q < 2; ++q) {
        double cst = ck[8 + q];
        for (int j = 0; j < 4; ++j) cst += ck[4 * q + j] * (aff[k * NX + 3 + j] - xl[3 + j]);
        lbA[4 * N + 2 * k + q] = -0.1 - cst; ubA[4 * N + 2 * k + q] = INFINITY;
        lbA[6 * N + 2 * k + q] = -INFINITY;  ubA[6 * N + 2 * k + q] = 0.1 - cst;
      }
      for (int j = 0; j < 12; ++j) {
        const double th0 = 2 * M_PI * (double)j / 12, th1 = (j + 1 == 12) ? 2 * M_PI : 2 * M_PI * (double)(j + 1) / 12;
        const double ac0 = 9.163 * sin(th0), ac1 = 9.163 * sin(th1), al0 = 10.0 * cos(th0), al1 = 10.0 * cos(th1);
        const double dac = ac1 - ac0, dal = al1 - al0;
        double cst = (ul[0] - al0) * dac - (ck[13] / 280 - ac0) * dal;
        for (int jj = 0; jj < 3; ++jj) cst += dal * ck[10 + jj] * (aff[k * NX + 3 + jj] - xl[3 + jj]);
        cst -= dac * ul[0];
        lbA[8 * N + 12 * k + j] = -INFINITY; ubA[8 * N + 12 * k + j] = 0 - cst;
      }
    }
  }
  // ---- 4c. constraint matrix A (nC x nV, column-major).  One column at a time per wavefront: the column of Bt (R doubles,
  // contiguous) is staged in LDS with coalesced loads, then the nC entries of that column of A are formed from it and stored
  // with coalesced stores.  (Round 2 swept all entries with one thread each and read Bt with stride NX: 4.2 of the 10.3 ms of a
  // dynamic N = 60 batch.)  The 12 half-plane directions of the tyre ellipse come from a small LDS table. ----
  {
    const int lane = tid & 63, wv = tid >> 6, nwv = nth >> 6;
    double* bc = colst + (size_t)wv * R;                     // this wave's column stage
    for (int col = wv; col < nV; col += nwv) {
      const bool inp = col < 2 * N;                          // input column: a column of Bt; slack column: unit entries only
      for (int i = lane; i < R; i += 64) bc[i] = inp ? Bt[(size_t)col * R + i] : 0.0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double* acol = A + (size_t)col * nC;
      for (int row = lane; row < nC; row += 64) {
        double v = 0.0;
        if (row < 4 * N) {
          const int blk = row / N, k = row - blk * N;
          const int idx = blk == 0 ? vidx : (blk == 1 ? didx : nidx);
          v = bc[k * NX + idx];
          if (col == scol && blk == 2) v = 1.0;
          if (col == scol && blk == 3) v = -1.0;
        } else if (NX == 5) {
          const int blk = (row - 4 * N) / N, k = row - 4 * N - blk * N;
          const double* ck = cc + (size_t)k * CW;
          v = ck[0] * bc[k * NX + 3] + ck[1] * bc[k * NX + 4];
          if (col == scol) v = blk == 0 ? 1.0 : -1.0;   // shared slack (quirk C-7)
        } else if (row < 8 * N) {
          const int blk = (row - 4 * N) / (2 * N), rr = row - 4 * N - blk * 2 * N, k = rr >> 1, q = rr & 1;
          const double* ck = cc + (size_t)k * CW;
          for (int j = 0; j < 4; ++j) v += ck[4 * q + j] * bc[k * NX + 3 + j];
          if (col == scol + 1 + q) v = blk == 0 ? 1.0 : -1.0;
        } else {
          const int rr = row - 8 * N, k = rr / 12, j = rr - 12 * k;
          const double* ck = cc + (size_t)k * CW;
          const double dac = ell[j], dal = ell[12 + j];
          for (int jj = 0; jj < 3; ++jj) v += dal * ck[10 + jj] * bc[k * NX + 3 + jj];
          if (col == 2 * k) v += dac;
          if (col == scol + 3) v = -1.0;
        }
        acol[row] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  // ---- 5. H = 2 (Bt' Qbar Bt + Rbar), g = 2 Bt' Qbar r  (generate_qp.m:29-31); only states 1..3 carry weight ----
  // The input block of H is a SYRK over the 3N weighted rows of Bt: on the matrix cores (v_mfma_f64_16x16x4_f64), one 16 x 16 tile
  // pair (I >= J) per wavefront at a time, operands gathered from Bt (column-major, just written by this workgroup: L2), k-steps
  // started at the first block row in which column tile I is non-zero (Bt is block lower-triangular).  (Round 2 computed every
  // entry as a scalar dot product with stride-NX loads: 3.5 of the 10.3 ms of a dynamic N = 60 batch, 0.95 of 2.26 ms on the
  // headline shape.)  Lane (c = l & 15, q = l >> 4): A operand = weight * Bt[row rho][16 I + c], B operand = Bt[row rho][16 J + c],
  // rho = 4 s + q over the weighted rows (k, r) = (rho / 3, rho % 3); result register p holds H[16 I + q + 4 p][16 J + c].
  const double Qw[3] = {5, 250, 2000};   // ltvmpc_*.m:32 ; Q_terminal = 10 Q (:33)
  {
    const int lane = tid & 63, wv = tid >> 6, nwv = nth >> 6, c = lane & 15, q = lane >> 4;
    const int nU = 2 * N, Tu = (nU + 15) >> 4, npairs = Tu * (Tu + 1) / 2, ksteps = (3 * N + 3) >> 2;
    for (int pidx = wv; pidx < npairs; pidx += nwv) {
      int I = 0;
      while ((I + 1) * (I + 2) / 2 <= pidx) ++I;
      const int J = pidx - I * (I + 1) / 2;                      // I >= J
      const int ci = 16 * I + c, cj = 16 * J + c;
      const double* pi_ = Bt + (size_t)(ci < nU ? ci : 0) * R;
      const double* pj_ = Bt + (size_t)(cj < nU ? cj : 0) * R;
      const bool oni = ci < nU, onj = cj < nU;
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      const int s0 = (3 * 8 * I) >> 2;                            // column tile I starts at stage 8 I: weighted row 24 I
      int rho = 4 * s0 + q, k = rho / 3, r = rho - 3 * k;
      constexpr int UN = 8;                                       // k-steps per round: all 2 UN gathers in flight before the first MFMA
      for (int s = s0; s < ksteps; s += UN) {
        double av[UN], bv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const bool on = k < N;                                  // (rows beyond 3N: zero operands; the address stays inside Bt)
          const int off = (on ? k : 0) * NX + r;
          const double wq = ((k == N - 1) ? 10.0 : 1.0) * (r == 0 ? Qw[0] : (r == 1 ? Qw[1] : Qw[2]));
          const double a_ = pi_[off], b_ = pj_[off];
          av[u] = (on && oni) ? wq * a_ : 0.0;
          bv[u] = (on && onj) ? b_ : 0.0;
          rho += 4; ++r; ++k; if (r == 3) { r = 0; ++k; }        // rho + 4 = 3 (k + 1) + (r + 1)
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int i = 16 * I + q + 4 * p, j = cj;
        if (i < nU && j < nU) {
          const double v = 2.0 * (acc[p] + (i == j ? 10.0 : 0.0));   // R = [10,10] (ltvmpc_*.m:34)
          H[(size_t)i + (size_t)j * nV] = v;
          H[(size_t)j + (size_t)i * nV] = v;
        }
      }
    }
    // slack rows / columns of H carry no quadratic cost
    for (int e = tid; e < NS * nV; e += nth) {
      const int sc = 2 * N + e / nV, i = e - (e / nV) * nV;
      H[(size_t)i + (size_t)sc * nV] = 0.0;
      H[(size_t)sc + (size_t)i * nV] = 0.0;
    }
  }
  double qc_local = 0.0;
  for (int i = tid; i < nV; i += nth) {
    double s = 0.0;
    if (i < 2 * N) {
      const double* ci = Bt + (size_t)i * R;
      for (int k = i >> 1; k < N; ++k) {
        const double wq = (k == N - 1) ? 10.0 : 1.0;
        for (int r = 0; r < 3; ++r) s += ci[k * NX + r] * (wq * Qw[r]) * (aff[k * NX + r] - x_ref[k * NX + r]);
      }
      g[i] = 2 * s;
    } else {
      const int sidx = i - 2 * N;
      g[i] = (NX == 5) ? 1e8 : (sidx == 0 ? 1e8 : (sidx == 3 ? 1e4 : 1e6));   // R_soft (ltvmpc_*.m:35)
    }
  }
  for (int e = tid; e < 3 * N; e += nth) {
    const int k = e / 3, r = e - 3 * k;
    const double wq = (k == N - 1) ? 10.0 : 1.0;
    const double rr = aff[k * NX + r] - x_ref[k * NX + r];
    qc_local += rr * (wq * Qw[r]) * rr;
  }
  red[tid] = qc_local;
  __syncthreads();
  if (tid == 0) {
    double s = 0; for (int i = 0; i < nth; ++i) s += red[i];
    if (P.qconst) P.qconst[b] = s;
  }
  if (P.pred) for (int i = tid; i < R; i += nth) P.pred[(size_t)b * R + i] = aff[i];
}

// post-solve: x_opt = aff + Bt z ; u_opt = z(1:2N) ; slack ; fval += const   (ltvmpc_*.m:57-60)
__global__ void ltv_post_kernel(int nx, int N, int ns, const double* z, const double* pred, const double* Bt, const double* qconst,
                                double* u_opt, double* x_opt, double* slack, double* fval) {
  const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
  const int R = nx * N, nV = 2 * N + ns;
  const double* zb = z + (size_t)b * nV;
  const double* Btb = Bt + (size_t)b * R * nV;
  for (int r = tid; r < R; r += nth) {
    double s = pred[(size_t)b * R + r];
    for (int c = 0; c < nV; ++c) s += Btb[r + (size_t)c * R] * zb[c];
    x_opt[(size_t)b * R + r] = s;
  }
  for (int c = tid; c < 2 * N; c += nth) u_opt[(size_t)b * 2 * N + c] = zb[c];
  for (int c = tid; c < ns; c += nth) slack[(size_t)b * ns + c] = zb[2 * N + c];
  if (tid == 0) fval[b] += qconst[b];
}

}  // namespace

size_t ltv_build_lds_bytes(int nx, int N, int threads) {
  const int CW = (nx == 5) ? 3 : 16;
  return ((size_t)N * nx * nx + (size_t)N * nx * 2 + (size_t)N * nx + (size_t)nx * N + (size_t)N * CW + threads + 24 +
          (size_t)(threads / 64) * nx * N) * sizeof(double);   // last term: the per-wavefront column stage of step 4c
}

hipError_t ltv_build_launch(const LtvParams& P, int batch, hipStream_t st) {
  const int threads = 256;
  const size_t lds = ltv_build_lds_bytes(P.nx, P.N, threads);
  if (P.nx == 5) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ltv_build_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ltv_build_kernel<5>, dim3(batch), dim3(threads), lds, st, P);
  } else {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ltv_build_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ltv_build_kernel<7>, dim3(batch), dim3(threads), lds, st, P);
  }
  return hipGetLastError();
}

hipError_t ltv_post_launch(int nx, int N, int ns, int batch, const double* z, const double* pred, const double* Bt, const double* qconst,
                           double* u_opt, double* x_opt, double* slack, double* fval, hipStream_t st) {
  hipLaunchKernelGGL(ltv_post_kernel, dim3(batch), dim3(256), 0, st, nx, N, ns, z, pred, Bt, qconst, u_opt, x_opt, slack, fval);
  return hipGetLastError();
}
