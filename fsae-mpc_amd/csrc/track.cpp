// track.cpp -- the track pipeline of the reference's setup path as a host-side library: race-line CSV ->
// periodic cubic spline in Bezier form -> arc-length reparameterisation -> the M x 4 spline tables the kernels' kappa(s)
// lookup reads (main.m:11-17).  Plain host code (no device work), part of libfsaempc.so; C ABI in include/fsaempc.h.
//
// Restates
//   util/read_raceline_csv.m:6-19      readmatrix of the CSV (one header line), columns 1, 2 = X, Y of the race line
//   spline/make_spline_periodic.m:9-33 P0 = P, P3 = P shifted; cyclic [1 4 1] system for P1 (b_i = 4 P_i + 2 P_{i+1});
//                                      P2_i = 2 P_{i+1} - P1_{i+1}
//   spline/arclength_reparam.m:15-64   segment "lengths" by quadrature of the speed integrand -- which uses the control point
//                                      P0 where the Bezier derivative has P1 (:20-23, :43-46; SURVEY App. C-6, kept) --,
//                                      M+1 evenly spaced stations by bisection to 0.01 (:49, :68-97), evaluation of the OLD
//                                      spline there (spline/interpolate_spline.m:11-21), periodic refit of the first M points
// MATLAB's adaptive `integral` (Gauss-Kronrod 7-15, AbsTol 1e-10, RelTol 1e-6) is restated as an adaptive Gauss-Kronrod 7-15
// with the same tolerances; the integrand is the square root of a quartic, one or two panels resolve it to round-off, and the
// 0.01 bisection tolerance makes the station sequence insensitive to quadrature differences of that size.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "fsaempc.h"

namespace {

// make_spline_periodic.m:9-33.  P: N points -> coeffs N x 4 column-major [P0 | P1 | P2 | P3]
void make_spline_periodic(const std::vector<double>& P, std::vector<double>& C) {
  const int N = (int)P.size();
  C.assign((size_t)4 * N, 0.0);
  std::vector<double> A((size_t)N * N, 0.0), b(N);
  for (int i = 0; i < N; ++i) {
    A[(size_t)i * N + i] = 4.0;
    A[(size_t)i * N + (i + N - 1) % N] += 1.0;      // (N = 2 would fold both neighbours onto one entry, as spdiags + the corner assignments do)
    A[(size_t)i * N + (i + 1) % N] += 1.0;
    b[i] = 4.0 * P[i] + 2.0 * P[(i + 1) % N];
  }
  // dense Gaussian elimination with partial pivoting (the matrix is diagonally dominant; N is a few hundred)
  for (int k = 0; k < N; ++k) {
    int p = k; double best = fabs(A[(size_t)k * N + k]);
    for (int i = k + 1; i < N; ++i) if (fabs(A[(size_t)i * N + k]) > best) { best = fabs(A[(size_t)i * N + k]); p = i; }
    if (p != k) { for (int j = 0; j < N; ++j) { const double t = A[(size_t)k * N + j]; A[(size_t)k * N + j] = A[(size_t)p * N + j]; A[(size_t)p * N + j] = t; } const double t = b[k]; b[k] = b[p]; b[p] = t; }
    const double pv = A[(size_t)k * N + k];
    for (int i = k + 1; i < N; ++i) {
      const double f = A[(size_t)i * N + k] / pv;
      if (f == 0.0) continue;
      for (int j = k; j < N; ++j) A[(size_t)i * N + j] -= f * A[(size_t)k * N + j];
      b[i] -= f * b[k];
    }
  }
  std::vector<double> P1(N);
  for (int i = N - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < N; ++j) s -= A[(size_t)i * N + j] * P1[j];
    P1[i] = s / A[(size_t)i * N + i];
  }
  for (int i = 0; i < N; ++i) {
    C[i] = P[i];
    C[(size_t)N + i] = P1[i];
    C[(size_t)2 * N + i] = 2.0 * P[(i + 1) % N] - P1[(i + 1) % N];
    C[(size_t)3 * N + i] = P[(i + 1) % N];
  }
}

// interpolate_spline.m:11-21 for one t (dl = 1 in the reparameterisation)
double interpolate_spline(double t, const std::vector<double>& C, int N, double dl) {
  t = fmod(t, dl * N); if (t < 0) t += dl * N;
  int i = (int)floor(t / dl);
  if (i >= N) i = N - 1;
  const double s = t / dl - i;
  return C[i] * (1 - s) * (1 - s) * (1 - s) + 3 * C[(size_t)N + i] * (1 - s) * (1 - s) * s + 3 * C[(size_t)2 * N + i] * (1 - s) * s * s + C[(size_t)3 * N + i] * s * s * s;
}

struct Speed {   // arclength_reparam.m:20-23: note x_P(i,1) where x_P(i,2) is expected (quirk C-6, kept)
  double x0, x2, x3, y0, y2, y3;
  double operator()(double t) const {
    const double a = -3 * (1 - t) * (1 - t), b_ = 3 * (3 * t * t - 4 * t + 1), c = 3 * (2 * t - 3 * t * t), d = 3 * t * t;
    const double xd = a * x0 + b_ * x0 + c * x2 + d * x3, yd = a * y0 + b_ * y0 + c * y2 + d * y3;
    return sqrt(xd * xd + yd * yd);
  }
};

// Gauss-Kronrod 7-15 on [a, b]: returns the Kronrod estimate, *err = |K15 - G7|
double gk15(const Speed& f, double a, double b, double* err) {
  static const double xk[8] = {0.991455371120812639206854697526329, 0.949107912342758524526189684047851, 0.864864423359769072789712788640926,
                               0.741531185599394439863864773280788, 0.586087235467691130294144838258730, 0.405845151377397166906606412076961,
                               0.207784955007898467600689403773245, 0.000000000000000000000000000000000};
  static const double wk[8] = {0.022935322010529224963732008058970, 0.063092092629978553290700663189204, 0.104790010322250183839876322541518,
                               0.140653259715525918745189590510238, 0.169004726639267902826583426598550, 0.190350578064785409913256402421014,
                               0.204432940075298892414161999234649, 0.209482141084727828012999174891714};
  static const double wg[4] = {0.129484966168869693270611432679082, 0.279705391489276667901467771423780, 0.381830050505118944950369775488975,
                               0.417959183673469387755102040816327};
  const double c = 0.5 * (a + b), h = 0.5 * (b - a);
  const double fc = f(c);
  double rk = wk[7] * fc, rg = wg[3] * fc;
  for (int j = 0; j < 7; ++j) {
    const double dx = h * xk[j];
    const double s = f(c - dx) + f(c + dx);
    rk += wk[j] * s;
    if (j & 1) rg += wg[j / 2] * s;
  }
  *err = fabs((rk - rg) * h);
  return rk * h;
}
// adaptive: bisect the panel with the largest error until the total error estimate meets max(AbsTol, RelTol |Q|)
double integral(const Speed& f, double a, double b) {
  if (!(b > a)) return 0.0;
  struct Panel { double a, b, q, e; };
  std::vector<Panel> ps;
  { double e; const double q = gk15(f, a, b, &e); ps.push_back({a, b, q, e}); }
  for (int it = 0; it < 200; ++it) {
    double Q = 0, E = 0; size_t worst = 0;
    for (size_t i = 0; i < ps.size(); ++i) { Q += ps[i].q; E += ps[i].e; if (ps[i].e > ps[worst].e) worst = i; }
    if (E <= fmax(1e-10, 1e-6 * fabs(Q))) return Q;
    const Panel p = ps[worst];
    const double mid = 0.5 * (p.a + p.b);
    double e1, e2;
    const double q1 = gk15(f, p.a, mid, &e1), q2 = gk15(f, mid, p.b, &e2);
    ps[worst] = {p.a, mid, q1, e1};
    ps.push_back({mid, p.b, q2, e2});
  }
  double Q = 0; for (const Panel& p : ps) Q += p.q;
  return Q;
}

// arclength_reparam.m:15-64 (periodic branch).  xC, yC: N x 4 column-major coefficient tables
void arclength_reparam(const std::vector<double>& xC, const std::vector<double>& yC, int N, int M,
                       std::vector<double>& xNew, std::vector<double>& yNew, double* dl_out, double* L_out) {
  auto speed = [&](int i) { return Speed{xC[i], xC[(size_t)2 * N + i], xC[(size_t)3 * N + i], yC[i], yC[(size_t)2 * N + i], yC[(size_t)3 * N + i]}; };
  std::vector<double> l_cum(N + 1, 0.0);
  for (int i = 0; i < N; ++i) l_cum[i + 1] = l_cum[i] + integral(speed(i), 0.0, 1.0);
  const double dl = l_cum[N] / M;
  std::vector<double> Px(M + 1), Py(M + 1);
  Px[0] = xC[0]; Py[0] = yC[0];
  Px[M] = xC[(size_t)3 * N + N - 1]; Py[M] = yC[(size_t)3 * N + N - 1];
  for (int i = 1; i < M; ++i) {
    int j1 = 0;                                       // find(l_cum >= i*dl, 1) - 1 (1-based segment) -> 0-based segment j = j1 - 1
    while (j1 <= N && !(l_cum[j1] >= i * dl)) ++j1;
    if (j1 > N) j1 = N;
    const int j = j1 - 1 < 0 ? 0 : j1 - 1;
    const Speed sp = speed(j);
    double xl = 0.0, xu = 1.0, t = 0.5;
    for (int itb = 0; itb < 200; ++itb) {             // bisection(0, 1, f, 0.01), :68-97
      t = 0.5 * (xl + xu);
      const double fx = integral(sp, 0.0, t) + l_cum[j] - i * dl;
      if (fabs(fx) <= 0.01) break;
      if (fx < 0) xl = t; else xu = t;
    }
    Px[i] = interpolate_spline(t + j, xC, N, 1.0);
    Py[i] = interpolate_spline(t + j, yC, N, 1.0);
  }
  make_spline_periodic(std::vector<double>(Px.begin(), Px.begin() + M), xNew);
  make_spline_periodic(std::vector<double>(Py.begin(), Py.begin() + M), yNew);
  *dl_out = dl; *L_out = l_cum[N];
}

thread_local char t_err[256] = "";
int tfail(int code, const char* msg, const char* a = "") { snprintf(t_err, sizeof(t_err), msg, a); return code; }

}  // namespace

extern "C" {

const char* fsaempc_track_last_error(void) { return t_err; }

int fsaempc_track_from_points(const double* x, const double* y, int n, int M, fsaempc_track* out) {
  if (!x || !y || !out || n < 3 || M < 3) return tfail(FSAEMPC_ERR_ARG, "track: need >= 3 points and M >= 3");
  for (int i = 0; i < n; ++i) if (!isfinite(x[i]) || !isfinite(y[i])) return tfail(FSAEMPC_ERR_ARG, "track: non-finite point");
  std::vector<double> xs, ys, xN, yN;
  make_spline_periodic(std::vector<double>(x, x + n), xs);     // main.m:14-15
  make_spline_periodic(std::vector<double>(y, y + n), ys);
  double dl = 0, L = 0;
  arclength_reparam(xs, ys, n, M, xN, yN, &dl, &L);            // main.m:17
  out->M = M; out->dl = dl; out->L = L;
  out->xP = (double*)malloc(sizeof(double) * 4 * M); out->yP = (double*)malloc(sizeof(double) * 4 * M);
  if (!out->xP || !out->yP) { free(out->xP); free(out->yP); out->xP = out->yP = nullptr; return tfail(FSAEMPC_ERR_ARG, "track: out of memory"); }
  memcpy(out->xP, xN.data(), sizeof(double) * 4 * M); memcpy(out->yP, yN.data(), sizeof(double) * 4 * M);
  return 0;
}

int fsaempc_track_from_csv(const char* path, int M, fsaempc_track* out) {
  if (!path || !out) return tfail(FSAEMPC_ERR_ARG, "track: null argument");
  FILE* f = fopen(path, "r");
  if (!f) return tfail(FSAEMPC_ERR_ARG, "track: cannot open %s", path);
  std::vector<double> x, y;
  char line[4096];
  bool first = true;
  while (fgets(line, sizeof(line), f)) {
    char* end = nullptr;
    const double a = strtod(line, &end);
    if (end == line) { if (first) { first = false; continue; } else continue; }   // header line (readmatrix skips it) / blank lines
    first = false;
    while (*end == ',' || *end == ' ' || *end == '\t' || *end == ';') ++end;
    char* end2 = nullptr;
    const double b_ = strtod(end, &end2);
    if (end2 == end) { fclose(f); return tfail(FSAEMPC_ERR_ARG, "track: a row of %s has fewer than two numeric columns", path); }
    x.push_back(a); y.push_back(b_);
  }
  fclose(f);
  if (x.size() < 3) return tfail(FSAEMPC_ERR_ARG, "track: %s holds fewer than three points", path);
  return fsaempc_track_from_points(x.data(), y.data(), (int)x.size(), M, out);
}

void fsaempc_track_free(fsaempc_track* t) { if (t) { free(t->xP); free(t->yP); t->xP = t->yP = nullptr; t->M = 0; } }

// On-disk table format "FSTRK001" (little endian): 8-byte magic, int32 M, int32 reserved (0), double dl, double L,
// then xP (M x 4 doubles, column-major: all P0, then all P1, P2, P3) and yP likewise.
int fsaempc_track_save(const fsaempc_track* t, const char* path) {
  if (!t || !path || !t->xP || !t->yP || t->M <= 0) return tfail(FSAEMPC_ERR_ARG, "track: nothing to save");
  FILE* f = fopen(path, "wb");
  if (!f) return tfail(FSAEMPC_ERR_ARG, "track: cannot write %s", path);
  const char magic[8] = {'F', 'S', 'T', 'R', 'K', '0', '0', '1'};
  const int hdr[2] = {t->M, 0};
  const double sc[2] = {t->dl, t->L};
  const bool ok = fwrite(magic, 1, 8, f) == 8 && fwrite(hdr, sizeof(int), 2, f) == 2 && fwrite(sc, sizeof(double), 2, f) == 2 &&
                  fwrite(t->xP, sizeof(double), (size_t)4 * t->M, f) == (size_t)4 * t->M && fwrite(t->yP, sizeof(double), (size_t)4 * t->M, f) == (size_t)4 * t->M;
  fclose(f);
  return ok ? 0 : tfail(FSAEMPC_ERR_ARG, "track: short write to %s", path);
}

int fsaempc_track_load(const char* path, fsaempc_track* out) {
  if (!path || !out) return tfail(FSAEMPC_ERR_ARG, "track: null argument");
  FILE* f = fopen(path, "rb");
  if (!f) return tfail(FSAEMPC_ERR_ARG, "track: cannot open %s", path);
  char magic[8]; int hdr[2]; double sc[2];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "FSTRK001", 8) != 0 || fread(hdr, sizeof(int), 2, f) != 2 || fread(sc, sizeof(double), 2, f) != 2 || hdr[0] < 3 || hdr[0] > (1 << 20)) {
    fclose(f); return tfail(FSAEMPC_ERR_ARG, "track: %s is not an FSTRK001 table", path);
  }
  const int M = hdr[0];
  out->M = M; out->dl = sc[0]; out->L = sc[1];
  out->xP = (double*)malloc(sizeof(double) * 4 * M); out->yP = (double*)malloc(sizeof(double) * 4 * M);
  const bool ok = out->xP && out->yP && fread(out->xP, sizeof(double), (size_t)4 * M, f) == (size_t)4 * M && fread(out->yP, sizeof(double), (size_t)4 * M, f) == (size_t)4 * M;
  fclose(f);
  if (!ok) { fsaempc_track_free(out); return tfail(FSAEMPC_ERR_ARG, "track: %s is truncated", path); }
  return 0;
}

}  // extern "C"
