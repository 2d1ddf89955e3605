/*
 * ltv_oracle_qp.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Restates the *contract* of the qpOASES MEX call the reference makes at
 * mpc/ltv/kinematic/ltvmpc_kinetmatic_curvilinear.m:52 and
 * mpc/ltv/dynamic/ltvmpc_dynamic_curvilinear.m:52 :
 *     min 1/2 x'Hx + x'g   s.t.  lb <= x <= ub,  lbA <= A x <= ubA
 * (optimizers/matlab/qpOASES/qpOASES.m:16-23, outputs/exit codes :41-62).
 * qpOASES itself (coin-or/qpOASES 3.2.x, online active-set strategy) is a
 * third-party dependency that is NOT in /root/reference (binaries only, version
 * unpinned) => PARITY UNPINNED.  The minimiser of these convex QPs is unique in
 * x, so any exact method must return the same x/fval; this file computes it
 * with a dense fp64 Mehrotra predictor-corrector interior-point method followed
 * by an active-set polish (solves the equality-constrained KKT system of the
 * identified active set, i.e. the vertex-exact point an active-set solver
 * stops at), and certifies it with orc_qp_kkt().
 */
#include "ltv_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX(i, j, ld) ((size_t)(i) + (size_t)(j) * (size_t)(ld))

void orc_qp_default_opts(orc_qp_opts* o) {
  o->tol = 1e-8;
  o->tol_x = 1e-7;
  o->tol_loose = 1e-6;
  o->max_iter = 100;
  o->inf_bound = 1e9;
  o->polish = 1;
  o->corrector = 1;
  o->scale = 1;
  o->verbose = 0;
}

/* in-place Cholesky (lower) of n x n column-major; returns 0 ok, >0 index of failing pivot+1 */
static int chol_lower(int n, double* M, double reg_floor) {
  for (int k = 0; k < n; ++k) {
    double p = M[IDX(k, k, n)];
    if (!(p > reg_floor)) {
      if (!isfinite(p)) return k + 1;
      p = reg_floor > 0 ? reg_floor : 1e-300;
    }
    p = sqrt(p);
    M[IDX(k, k, n)] = p;
    for (int i = k + 1; i < n; ++i) M[IDX(i, k, n)] /= p;
    for (int j = k + 1; j < n; ++j) {
      double ljk = M[IDX(j, k, n)];
      if (ljk == 0) continue;
      for (int i = j; i < n; ++i) M[IDX(i, j, n)] -= M[IDX(i, k, n)] * ljk;
    }
  }
  return 0;
}
static void chol_solve(int n, const double* L, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[IDX(i, k, n)] * b[k];
    b[i] = s / L[IDX(i, i, n)];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[IDX(k, i, n)] * b[k];
    b[i] = s / L[IDX(i, i, n)];
  }
}

/* dense LU with partial pivoting: factor in place (unit lower factors below the diagonal), pivot rows in piv;
 * returns 0 ok, 1 singular */
static int lu_factor(int n, double* K, int* piv) {
  for (int k = 0; k < n; ++k) {
    int p = k; double best = fabs(K[IDX(k, k, n)]);
    for (int i = k + 1; i < n; ++i) if (fabs(K[IDX(i, k, n)]) > best) { best = fabs(K[IDX(i, k, n)]); p = i; }
    if (best < 1e-13) return 1;
    piv[k] = p;
    if (p != k) for (int j = 0; j < n; ++j) { double t = K[IDX(k, j, n)]; K[IDX(k, j, n)] = K[IDX(p, j, n)]; K[IDX(p, j, n)] = t; }
    const double pv = K[IDX(k, k, n)];
    for (int i = k + 1; i < n; ++i) {
      const double f = K[IDX(i, k, n)] / pv;
      K[IDX(i, k, n)] = f;
      if (f == 0) continue;
      for (int j = k + 1; j < n; ++j) K[IDX(i, j, n)] -= f * K[IDX(k, j, n)];
    }
  }
  return 0;
}
static void lu_apply(int n, const double* K, const int* piv, double* r) {
  for (int k = 0; k < n; ++k) if (piv[k] != k) { double t = r[k]; r[k] = r[piv[k]]; r[piv[k]] = t; }   /* whole rows were swapped: P first */
  for (int k = 0; k < n; ++k) {
    const double rk = r[k];
    if (rk != 0) for (int i = k + 1; i < n; ++i) r[i] -= K[IDX(i, k, n)] * rk;
  }
  for (int i = n - 1; i >= 0; --i) {
    double sacc = r[i];
    for (int j = i + 1; j < n; ++j) sacc -= K[IDX(i, j, n)] * r[j];
    r[i] = sacc / K[IDX(i, i, n)];
  }
}
/* solves K y = r (K is destroyed, r overwritten) with two steps of iterative refinement against a copy of K */
static int lu_solve(int n, double* K, double* r) {
  double* K0 = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* r0 = (double*)malloc(sizeof(double) * n);
  double* d = (double*)malloc(sizeof(double) * n);
  int* piv = (int*)malloc(sizeof(int) * n);
  memcpy(K0, K, sizeof(double) * (size_t)n * n); memcpy(r0, r, sizeof(double) * n);
  int sing = lu_factor(n, K, piv);
  if (!sing) {
    lu_apply(n, K, piv, r);
    for (int itr = 0; itr < 2; ++itr) {
      for (int i = 0; i < n; ++i) d[i] = r0[i];
      for (int j = 0; j < n; ++j) { const double yj = r[j]; if (yj != 0) for (int i = 0; i < n; ++i) d[i] -= K0[IDX(i, j, n)] * yj; }
      lu_apply(n, K, piv, d);
      for (int i = 0; i < n; ++i) r[i] += d[i];
    }
  }
  free(K0); free(r0); free(d); free(piv);
  return sing;
}

double orc_qp_kkt(int nV, int nC, const double* H, const double* g, const double* A,
                  const double* lb, const double* ub, const double* lbA, const double* ubA,
                  const double* x, const double* lambda, double inf_bound, double* res) {
  int n = nV, m = nC;
  double r_stat = 0, r_prim = 0, r_sign = 0, r_comp = 0;
  double fval = 0;
  double* Hx = (double*)calloc(n, sizeof(double));
  double* Gz = (double*)calloc(n, sizeof(double));
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) Hx[i] += H[IDX(i, j, n)] * x[j];
  for (int i = 0; i < n; ++i) fval += 0.5 * x[i] * Hx[i] + g[i] * x[i];
  double fs = fmax(1.0, fabs(fval));
  for (int i = 0; i < n; ++i) Gz[i] = lambda[i];
  for (int r = 0; r < m; ++r) {
    double lam = lambda[n + r];
    if (lam != 0) for (int j = 0; j < n; ++j) Gz[j] += A[IDX(r, j, m)] * lam;
  }
  for (int j = 0; j < n; ++j) {
    double sc = fmax(1.0, fmax(fabs(g[j]), fmax(fabs(Hx[j]), fabs(Gz[j]))));
    r_stat = fmax(r_stat, fabs(Hx[j] + g[j] - Gz[j]) / sc);
  }
  for (int i = 0; i < n + m; ++i) {
    double v, l, u;
    if (i < n) { v = x[i]; l = lb[i]; u = ub[i]; }
    else {
      int r = i - n; v = 0;
      for (int j = 0; j < n; ++j) v += A[IDX(r, j, m)] * x[j];
      l = lbA[r]; u = ubA[r];
    }
    int hl = l > -inf_bound, hu = u < inf_bound;
    double sc = fmax(1.0, fabs(v));
    if (hl) { sc = fmax(sc, fabs(l)); }
    if (hu) { sc = fmax(sc, fabs(u)); }
    double viol = 0;
    if (l > -INFINITY && v < l) viol = l - v;
    if (u < INFINITY && v > u) viol = fmax(viol, v - u);
    r_prim = fmax(r_prim, viol / sc);
    double lam = lambda[i];
    if (lam > 0) {
      if (!hl) r_sign = fmax(r_sign, lam / fs);
      else r_comp = fmax(r_comp, lam * fabs(v - l) / fs);
    } else if (lam < 0) {
      if (!hu) r_sign = fmax(r_sign, -lam / fs);
      else r_comp = fmax(r_comp, -lam * fabs(u - v) / fs);
    }
  }
  free(Hx); free(Gz);
  if (res) { res[0] = r_stat; res[1] = r_prim; res[2] = r_sign; res[3] = r_comp; }
  return fmax(fmax(r_stat, r_prim), fmax(r_sign, r_comp));
}

/* ------------------------------------------------------------------------- */
typedef struct {
  int n, m, mt;
  double *H, *g, *A;      /* scaled copies */
  double *l, *u;          /* mt, scaled */
  unsigned char *hl, *hu; /* mt */
  double *E, *F;          /* column / row scaling */
} qp_work;

static void apply_G(const qp_work* w, const double* x, double* v) { /* v = [x; A x] */
  int n = w->n, m = w->m;
  for (int i = 0; i < n; ++i) v[i] = x[i];
  for (int r = 0; r < m; ++r) v[n + r] = 0;
  for (int j = 0; j < n; ++j) {
    double xj = x[j];
    if (xj == 0) continue;
    const double* col = w->A + (size_t)j * m;
    for (int r = 0; r < m; ++r) v[n + r] += col[r] * xj;
  }
}
static void apply_Gt(const qp_work* w, const double* y, double* out) { /* out = y[0:n] + A' y[n:] */
  int n = w->n, m = w->m;
  for (int j = 0; j < n; ++j) {
    const double* col = w->A + (size_t)j * m;
    double s = y[j];
    for (int r = 0; r < m; ++r) s += col[r] * y[n + r];
    out[j] = s;
  }
}

/* Active-set refinement of the interior-point point (test oracle: a dense, exact restatement that shares no
 * machinery with the HIP path).  Working set from the multipliers (side active iff |lambda| dominates its slack), then
 * the equality-constrained KKT system
 *      [ H  -G_W' ] [x]   [ -g ]
 *      [ G_W   0  ] [y] = [ b_W ]
 * by dense LU with partial pivoting, followed by single add / drop corrections of the working set (add the most violated
 * inactive side, else drop the multiplier of the wrong sign) until the point is a KKT point of the full QP -- the vertex
 * an active-set solver such as qpOASES stops at.  Returns 1 and overwrites (x, lambda) on success. */
static int polish(int n, int m, const double* H, const double* g, const double* A,
                  const double* lb, const double* ub, const double* lbA, const double* ubA,
                  double inf_bound, double* x, double* lambda) {
  const int mt = n + m, maxcorr = 8;
  const double ftol = 1e-9, stol = 1e-9;
  int* side = (int*)calloc(mt, sizeof(int));
  int* act = (int*)malloc(sizeof(int) * mt);
  double* v = (double*)calloc(mt, sizeof(double));
  double* xs = (double*)malloc(sizeof(double) * n);
  int ok = 0;
  for (int i = 0; i < n; ++i) v[i] = x[i];
  for (int j = 0; j < n; ++j) for (int r = 0; r < m; ++r) v[n + r] += A[IDX(r, j, m)] * x[j];
  for (int i = 0; i < mt; ++i) {
    double l = i < n ? lb[i] : lbA[i - n], u = i < n ? ub[i] : ubA[i - n];
    double lam = lambda[i];
    if (lam > 0 && l > -inf_bound && lam > fabs(v[i] - l)) side[i] = 1;
    else if (lam < 0 && u < inf_bound && -lam > fabs(u - v[i])) side[i] = -1;
  }
  for (int corr = 0; corr <= maxcorr && !ok; ++corr) {
    int na = 0;
    for (int i = 0; i < mt; ++i) if (side[i]) act[na++] = i;
    if (na > n) break;
    const int K = n + na;
    double* KK = (double*)calloc((size_t)K * K, sizeof(double));
    double* r = (double*)calloc(K, sizeof(double));
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) KK[IDX(i, j, K)] = H[IDX(i, j, n)];
    for (int a = 0; a < na; ++a) {
      const int i = act[a];
      for (int j = 0; j < n; ++j) {
        const double gij = i < n ? (i == j ? 1.0 : 0.0) : A[IDX(i - n, j, m)];
        KK[IDX(n + a, j, K)] = gij;
        KK[IDX(j, n + a, K)] = -gij;
      }
      const double l = i < n ? lb[i] : lbA[i - n], u = i < n ? ub[i] : ubA[i - n];
      r[n + a] = side[i] > 0 ? l : u;
    }
    for (int j = 0; j < n; ++j) r[j] = -g[j];
    const int sing = lu_solve(K, KK, r);
    if (!sing) {
      for (int i = 0; i < n; ++i) { xs[i] = r[i]; v[i] = r[i]; }
      for (int rr = 0; rr < m; ++rr) v[n + rr] = 0;
      for (int j = 0; j < n; ++j) for (int rr = 0; rr < m; ++rr) v[n + rr] += A[IDX(rr, j, m)] * r[j];
      double ymax = 1.0;
      for (int a = 0; a < na; ++a) ymax = fmax(ymax, fabs(r[n + a]));
      double worst_v = 0, worst_s = 0; int iv = -1, sv = 0, is = -1;
      for (int i = 0; i < mt; ++i) {
        if (side[i]) continue;
        const double l = i < n ? lb[i] : lbA[i - n], u = i < n ? ub[i] : ubA[i - n];
        double sc = fmax(1.0, fabs(v[i]));
        if (l > -inf_bound) { const double vl = (l - v[i]) / fmax(sc, fabs(l)); if (vl > worst_v) { worst_v = vl; iv = i; sv = 1; } }
        if (u < inf_bound) { const double vu = (v[i] - u) / fmax(sc, fabs(u)); if (vu > worst_v) { worst_v = vu; iv = i; sv = -1; } }
      }
      for (int a = 0; a < na; ++a) {
        const double y = r[n + a], sg = (side[act[a]] > 0 ? -y : y) / ymax;
        if (sg > worst_s) { worst_s = sg; is = act[a]; }
      }
      if (worst_v <= ftol && worst_s <= stol) {
        for (int i = 0; i < n; ++i) x[i] = xs[i];
        for (int i = 0; i < mt; ++i) lambda[i] = 0;
        for (int a = 0; a < na; ++a) {
          const double y = r[n + a];
          lambda[act[a]] = side[act[a]] > 0 ? fmax(y, 0.0) : fmin(y, 0.0);
        }
        ok = 1;
      } else if (worst_v > ftol && iv >= 0) side[iv] = sv;
      else if (is >= 0) side[is] = 0;
    }
    free(KK); free(r);
    if (sing) break;
  }
  free(side); free(act); free(v); free(xs);
  return ok;
}

int orc_qp_solve(int nV, int nC, const double* H, const double* g, const double* A,
                 const double* lb, const double* ub, const double* lbA, const double* ubA,
                 const orc_qp_opts* opts_in, double* x_out, double* fval_out, int* iter_out, double* lambda_out) {
  return orc_qp_solve_ex(nV, nC, H, g, A, lb, ub, lbA, ubA, opts_in, x_out, fval_out, iter_out, lambda_out, 0, 0);
}

/* Exit semantics (qpOASES.m:43-47 codes): x_out always carries the last (or best saved) iterate -- the reference keeps
 * driving on whatever the solver returned (main.m:163-175) -- NaN only if the data already held one.  -3 (unbounded) is
 * returned only when the objective follows a diverging x to -infinity; a diverging iterate on a bounded problem is an
 * internal failure (-1).  kkt_out: relative KKT residual (max of stationarity, primal, complementarity) of the returned
 * point as the solver measured it; polished_out: 1 if the active-set refinement was accepted. */
int orc_qp_solve_ex(int nV, int nC, const double* H, const double* g, const double* A,
                    const double* lb, const double* ub, const double* lbA, const double* ubA,
                    const orc_qp_opts* opts_in, double* x_out, double* fval_out, int* iter_out, double* lambda_out,
                    double* kkt_out, int* polished_out) {
  orc_qp_opts opts;
  if (opts_in) opts = *opts_in; else orc_qp_default_opts(&opts);
  const int n = nV, m = nC, mt = n + m;
  int flag = 1, it = 0, pol_done = 0;
  double last_merit = INFINITY, fval_it = 0;
  qp_work w;
  w.n = n; w.m = m; w.mt = mt;
  w.H = (double*)malloc(sizeof(double) * n * n);
  w.g = (double*)malloc(sizeof(double) * n);
  w.A = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1) * n);
  w.l = (double*)malloc(sizeof(double) * mt);
  w.u = (double*)malloc(sizeof(double) * mt);
  w.hl = (unsigned char*)malloc(mt);
  w.hu = (unsigned char*)malloc(mt);
  w.E = (double*)malloc(sizeof(double) * n);
  w.F = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
  double *x = (double*)calloc(n, sizeof(double)), *dx = (double*)calloc(n, sizeof(double)),
         *dxa = (double*)calloc(n, sizeof(double));
  double *tl = (double*)calloc(mt, sizeof(double)), *tu = (double*)calloc(mt, sizeof(double)),
         *zl = (double*)calloc(mt, sizeof(double)), *zu = (double*)calloc(mt, sizeof(double));
  double *v = (double*)calloc(mt, sizeof(double)), *dv = (double*)calloc(mt, sizeof(double)),
         *wv = (double*)calloc(mt, sizeof(double)), *D = (double*)calloc(mt, sizeof(double));
  double *rpl = (double*)calloc(mt, sizeof(double)), *rpu = (double*)calloc(mt, sizeof(double));
  double *dtl = (double*)calloc(mt, sizeof(double)), *dtu = (double*)calloc(mt, sizeof(double)),
         *dzl = (double*)calloc(mt, sizeof(double)), *dzu = (double*)calloc(mt, sizeof(double));
  double *cl = (double*)calloc(mt, sizeof(double)), *cu = (double*)calloc(mt, sizeof(double));
  double *Hx = (double*)calloc(n, sizeof(double)), *Gz = (double*)calloc(n, sizeof(double)),
         *rhs = (double*)calloc(n, sizeof(double)), *tmp = (double*)calloc(n, sizeof(double));
  double* M = (double*)malloc(sizeof(double) * n * n);
  int have_saved = 0;
  double saved_merit = INFINITY;
  double* xs = (double*)calloc(n, sizeof(double));
  double* lams = (double*)calloc(mt, sizeof(double));

  /* ---- scaling: x = E xs ; rows of A scaled by F --------------------------- */
  for (int j = 0; j < n; ++j) {
    double e = 1.0;
    if (opts.scale) {
      double hjj = H[IDX(j, j, n)];
      if (hjj > 1e-12) e = 1.0 / sqrt(hjj);
      else {
        double cm = 0;
        for (int r = 0; r < m; ++r) cm = fmax(cm, fabs(A[IDX(r, j, m)]));
        e = cm > 1e-12 ? 1.0 / cm : 1.0;
      }
    }
    w.E[j] = e;
  }
  for (int r = 0; r < m; ++r) {
    double f = 1.0;
    if (opts.scale) {
      double rm = 0;
      for (int j = 0; j < n; ++j) rm = fmax(rm, fabs(A[IDX(r, j, m)] * w.E[j]));
      f = rm > 1e-12 ? 1.0 / rm : 1.0;
    }
    w.F[r] = f;
  }
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i < n; ++i) w.H[IDX(i, j, n)] = H[IDX(i, j, n)] * w.E[i] * w.E[j];
    w.g[j] = g[j] * w.E[j];
    for (int r = 0; r < m; ++r) w.A[IDX(r, j, m)] = A[IDX(r, j, m)] * w.F[r] * w.E[j];
  }
  int infeasible_bounds = 0, cnt = 0;
  for (int i = 0; i < mt; ++i) {
    double l = i < n ? lb[i] : lbA[i - n], u = i < n ? ub[i] : ubA[i - n];
    double sc = i < n ? 1.0 / w.E[i] : w.F[i - n];
    if (isnan(l) || isnan(u)) { infeasible_bounds = 2; }
    w.hl[i] = l > -opts.inf_bound; w.hu[i] = u < opts.inf_bound;
    w.l[i] = w.hl[i] ? l * sc : -INFINITY;
    w.u[i] = w.hu[i] ? u * sc : INFINITY;
    if (w.hl[i] && w.hu[i] && l > u) infeasible_bounds = 1;
    if (w.hl[i] && w.hu[i] && !(u > l)) {
      /* equality row: open a tiny interior so the barrier is defined (documented relaxation) */
      double eps = 1e-9 * fmax(1.0, fabs(w.l[i]));
      w.l[i] -= eps; w.u[i] += eps;
    }
    cnt += w.hl[i] + w.hu[i];
  }
  if (infeasible_bounds) { flag = infeasible_bounds == 1 ? -2 : -1; goto finish; }

  /* ---- initial point: slacks >= T0, multipliers Z0 in the equilibrated problem (scan over the synthetic
   * LTV-MPC families: (10,100) needs 8-16 % fewer iterations than (1,1)) ------------------------------ */
  const double T0 = 10.0, Z0 = 100.0;
  for (int j = 0; j < n; ++j) {
    double xj = 0;
    if (w.hl[j] && xj < w.l[j]) xj = w.l[j];
    if (w.hu[j] && xj > w.u[j]) xj = w.u[j];
    x[j] = xj;
  }
  apply_G(&w, x, v);
  for (int i = 0; i < mt; ++i) {
    if (w.hl[i]) { tl[i] = fmax(v[i] - w.l[i], T0); zl[i] = Z0; }
    if (w.hu[i]) { tu[i] = fmax(w.u[i] - v[i], T0); zu[i] = Z0; }
  }
  /* bound multipliers absorb the initial dual residual (keeps the first Newton step sane when
   * |g| is huge, e.g. the 1e8 slack cost of ltvmpc_*.m:35) */
  for (int i = 0; i < mt; ++i) wv[i] = i < n ? 0.0 : (w.hl[i] ? zl[i] : 0) - (w.hu[i] ? zu[i] : 0);
  apply_Gt(&w, wv, Gz);
  for (int j = 0; j < n; ++j) {
    double hx = 0;
    for (int i = 0; i < n; ++i) hx += w.H[IDX(j, i, n)] * x[i];
    double r = hx + w.g[j] - Gz[j];
    if (w.hl[j]) zl[j] = fmax(r, 0.0) + Z0;
    if (w.hu[j]) zu[j] = fmax(-r, 0.0) + Z0;
  }

  double best_res = INFINITY;
  int stall = 0;
  for (it = 0; it <= opts.max_iter; ++it) {
    /* residuals */
    apply_G(&w, x, v);
    double mu = 0;
    for (int i = 0; i < mt; ++i) {
      rpl[i] = w.hl[i] ? v[i] - w.l[i] - tl[i] : 0;
      rpu[i] = w.hu[i] ? w.u[i] - v[i] - tu[i] : 0;
      wv[i] = (w.hl[i] ? zl[i] : 0) - (w.hu[i] ? zu[i] : 0);
      mu += (w.hl[i] ? tl[i] * zl[i] : 0) + (w.hu[i] ? tu[i] * zu[i] : 0);
    }
    double gap = mu;
    mu = cnt > 0 ? mu / cnt : 0;
    apply_Gt(&w, wv, Gz);
    double fval = 0;
    for (int i = 0; i < n; ++i) Hx[i] = 0;
    for (int j = 0; j < n; ++j) { double xj = x[j]; for (int i = 0; i < n; ++i) Hx[i] += w.H[IDX(i, j, n)] * xj; }
    for (int i = 0; i < n; ++i) fval += 0.5 * x[i] * Hx[i] + w.g[i] * x[i];
    double rd_rel = 0, rp_rel = 0;
    for (int j = 0; j < n; ++j) {
      double sc = fmax(1.0, fmax(fabs(w.g[j]), fmax(fabs(Hx[j]), fabs(Gz[j]))));
      rd_rel = fmax(rd_rel, fabs(Hx[j] + w.g[j] - Gz[j]) / sc);
    }
    for (int i = 0; i < mt; ++i) {
      double sc = fmax(1.0, fabs(v[i]));
      if (w.hl[i]) rp_rel = fmax(rp_rel, fabs(rpl[i]) / fmax(sc, fabs(w.l[i])));
      if (w.hu[i]) rp_rel = fmax(rp_rel, fabs(rpu[i]) / fmax(sc, fabs(w.u[i])));
    }
    double gap_rel = gap / fmax(1.0, fabs(fval));
    if (opts.verbose) printf("it %2d mu %.3e rd %.3e rp %.3e gap %.3e f %.8e\n", it, mu, rd_rel, rp_rel, gap_rel, fval);
    /* two levels: `tol` (strict, default 1e-8) ends the iteration together with the Newton-decrement test;
     * `tol_loose` (default 1e-6 = the KKT tolerance the build is specified to) qualifies an iterate as a
     * fall-back: the normal matrix loses all accuracy once the weights z/t pass ~1e19 (1e8 slack cost), and
     * the best qualified iterate is returned when the next steps turn to numerical garbage */
    const double merit = fmax(rd_rel, fmax(rp_rel, gap_rel));
    const int res_ok = merit <= opts.tol;
    last_merit = merit; fval_it = fval;
    if (merit <= opts.tol_loose && merit < saved_merit) {
      for (int j = 0; j < n; ++j) xs[j] = x[j];
      for (int i = 0; i < mt; ++i) lams[i] = (w.hl[i] ? zl[i] : 0) - (w.hu[i] ? zu[i] : 0);
      have_saved = 1; saved_merit = merit;
    } else if (merit > opts.tol_loose && rp_rel <= opts.tol_loose && gap_rel <= opts.tol_loose) {
      /* only the dual residual is in the way: repair the certificate of a copy of the iterate by moving r_d into the
       * bound multipliers where a finite bound of the right sign exists (same rule as the HIP kernel) */
      double dgap = 0, rd2 = 0;
      for (int j = 0; j < n; ++j) {
        double lam = (w.hl[j] ? zl[j] : 0) - (w.hu[j] ? zu[j] : 0);
        double r = Hx[j] + w.g[j] - Gz[j], lam2 = lam + r;
        int ok = lam2 >= 0 ? w.hl[j] : w.hu[j];
        if (ok) { tmp[j] = lam2; dgap += fabs(r) * fmax(0.0, lam2 >= 0 ? v[j] - w.l[j] : w.u[j] - v[j]); }
        else {
          tmp[j] = lam;
          double sc = fmax(1.0, fmax(fabs(w.g[j]), fmax(fabs(Hx[j]), fabs(Gz[j]))));
          rd2 = fmax(rd2, fabs(r) / sc);
        }
      }
      double merit2 = fmax(rd2, fmax(rp_rel, (gap + dgap) / fmax(1.0, fabs(fval))));
      if (merit2 <= opts.tol_loose && merit2 < saved_merit) {
        for (int j = 0; j < n; ++j) xs[j] = x[j];
        for (int i = 0; i < mt; ++i) lams[i] = i < n ? tmp[i] : (w.hl[i] ? zl[i] : 0) - (w.hu[i] ? zu[i] : 0);
        have_saved = 1; saved_merit = merit2;
      }
      if (have_saved) { flag = 2; break; }
    } else if (have_saved && merit > opts.tol_loose) {
      flag = 2; break;
    }
    double res_now = fmax(rd_rel, fmax(rp_rel, gap_rel));
    if (!isfinite(res_now)) { if (opts.verbose) printf("nonfinite residual\n"); flag = -1; break; }
    if (res_now < 0.9 * best_res) { best_res = res_now; stall = 0; } else ++stall;

    /* normal matrix M = H + diag(D_b) + A' D_A A */
    for (int i = 0; i < mt; ++i) D[i] = (w.hl[i] ? zl[i] / tl[i] : 0) + (w.hu[i] ? zu[i] / tu[i] : 0);
    memcpy(M, w.H, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j) M[IDX(j, j, n)] += D[j];
    for (int j = 0; j < n; ++j) {
      const double* cj = w.A + (size_t)j * m;
      for (int i = j; i < n; ++i) {
        const double* ci = w.A + (size_t)i * m;
        double s = 0;
        for (int r = 0; r < m; ++r) s += ci[r] * D[n + r] * cj[r];
        M[IDX(i, j, n)] += s;
      }
    }
    double dmax = 0;
    for (int j = 0; j < n; ++j) dmax = fmax(dmax, M[IDX(j, j, n)]);
    { int ce = chol_lower(n, M, 1e-30 * dmax); if (ce) { if (opts.verbose) printf("chol fail at %d dmax %g\n", ce, dmax); flag = res_ok ? 0 : (have_saved ? 2 : -1);
        /* the factorisation broke down (weights ~1e24) on an iterate that is primal feasible and complementary to
         * tol_loose; only the dual residual is numerical noise: let the active-set polish certify it (flag 4) */
        if (flag == -1 && opts.polish && rp_rel <= 1e-4 && gap_rel <= 1e-4) flag = 4;   /* nearly feasible and complementary: the refinement gets a try */
        break; } }

    /* predictor */
    for (int i = 0; i < mt; ++i)
      wv[i] = (w.hl[i] ? -(zl[i] / tl[i]) * rpl[i] : 0) + (w.hu[i] ? (zu[i] / tu[i]) * rpu[i] : 0);
    apply_Gt(&w, wv, tmp);
    for (int j = 0; j < n; ++j) { rhs[j] = -(Hx[j] + w.g[j]) + tmp[j]; dxa[j] = rhs[j]; }
    chol_solve(n, M, dxa);
    /* Newton-decrement test: the affine direction is the predicted remaining move to the KKT point;
     * stop when it is below tol_x in the caller's (unscaled) coordinates. |f| can be ~1e9 because of
     * the 1e8 slack cost, so a gap test relative to |f| alone leaves x loose in flat directions. */
    if (res_ok) {
      double dmx = 0, xmx = 1.0;
      for (int j = 0; j < n; ++j) { dmx = fmax(dmx, fabs(dxa[j] * w.E[j])); xmx = fmax(xmx, fabs(x[j] * w.E[j])); }
      if (opts.verbose) printf("      dxaff_rel %.3e\n", dmx / xmx);
      if (dmx <= opts.tol_x * xmx) { flag = 0; break; }
    }
    if (it == opts.max_iter) { flag = 1; break; }
    apply_G(&w, dxa, dv);
    double alpha_aff = 1.0;
    for (int i = 0; i < mt; ++i) {
      if (w.hl[i]) {
        dtl[i] = dv[i] + rpl[i];
        dzl[i] = -zl[i] - (zl[i] / tl[i]) * dtl[i];
        if (dtl[i] < 0) alpha_aff = fmin(alpha_aff, -tl[i] / dtl[i]);
        if (dzl[i] < 0) alpha_aff = fmin(alpha_aff, -zl[i] / dzl[i]);
      }
      if (w.hu[i]) {
        dtu[i] = -dv[i] + rpu[i];
        dzu[i] = -zu[i] - (zu[i] / tu[i]) * dtu[i];
        if (dtu[i] < 0) alpha_aff = fmin(alpha_aff, -tu[i] / dtu[i]);
        if (dzu[i] < 0) alpha_aff = fmin(alpha_aff, -zu[i] / dzu[i]);
      }
    }
    double mu_aff = 0;
    for (int i = 0; i < mt; ++i) {
      if (w.hl[i]) mu_aff += (tl[i] + alpha_aff * dtl[i]) * (zl[i] + alpha_aff * dzl[i]);
      if (w.hu[i]) mu_aff += (tu[i] + alpha_aff * dtu[i]) * (zu[i] + alpha_aff * dzu[i]);
    }
    mu_aff = cnt > 0 ? mu_aff / cnt : 0;
    double sigma = mu > 0 ? pow(mu_aff / mu, 3.0) : 0;
    if (sigma > 1) sigma = 1;
    {
      /* do not drive mu far below what the gap criterion needs (keeps M well conditioned) */
      double mu_floor = 1e-5 * opts.tol * fmax(1.0, fabs(fval)) / (cnt > 0 ? cnt : 1);
      if (mu > 0 && sigma < mu_floor / mu) sigma = fmin(1.0, mu_floor / mu);
    }
    /* corrector */
    for (int i = 0; i < mt; ++i) {
      /* the second-order term is dropped when the affine step is tiny (it then models nothing and makes the
       * iteration cycle on low-speed instances) */
      const double cw = (opts.corrector && alpha_aff >= 0.05) ? 1.0 : 0.0;
      cl[i] = w.hl[i] ? sigma * mu - cw * dtl[i] * dzl[i] : 0;
      cu[i] = w.hu[i] ? sigma * mu - cw * dtu[i] * dzu[i] : 0;
      wv[i] = (w.hl[i] ? cl[i] / tl[i] : 0) - (w.hu[i] ? cu[i] / tu[i] : 0);
    }
    apply_Gt(&w, wv, tmp);
    for (int j = 0; j < n; ++j) dx[j] = rhs[j] + tmp[j];
    chol_solve(n, M, dx);
    apply_G(&w, dx, dv);
    /* step length: Mehrotra's heuristic on the blocking pair (as in OOQP): keeps the pair that blocks
     * the step from collapsing far below the average complementarity, which otherwise jams the method */
    double alpha = 1e300, bp = 0, bdp = 0, bd = 0, bdd = 0;
    for (int i = 0; i < mt; ++i) {
      if (w.hl[i]) {
        dtl[i] = dv[i] + rpl[i];
        dzl[i] = -zl[i] + cl[i] / tl[i] - (zl[i] / tl[i]) * dtl[i];
        if (dtl[i] < 0 && -tl[i] / dtl[i] < alpha) { alpha = -tl[i] / dtl[i]; bp = tl[i]; bdp = dtl[i]; bd = zl[i]; bdd = dzl[i]; }
        if (dzl[i] < 0 && -zl[i] / dzl[i] < alpha) { alpha = -zl[i] / dzl[i]; bp = zl[i]; bdp = dzl[i]; bd = tl[i]; bdd = dtl[i]; }
      }
      if (w.hu[i]) {
        dtu[i] = -dv[i] + rpu[i];
        dzu[i] = -zu[i] + cu[i] / tu[i] - (zu[i] / tu[i]) * dtu[i];
        if (dtu[i] < 0 && -tu[i] / dtu[i] < alpha) { alpha = -tu[i] / dtu[i]; bp = tu[i]; bdp = dtu[i]; bd = zu[i]; bdd = dzu[i]; }
        if (dzu[i] < 0 && -zu[i] / dzu[i] < alpha) { alpha = -zu[i] / dzu[i]; bp = zu[i]; bdp = dzu[i]; bd = tu[i]; bdd = dtu[i]; }
      }
    }
    if (alpha < 1e299) {
      const double gamma_f = 0.99, gamma_a = 1.0 / (1.0 - gamma_f);
      double mufull = 0;
      for (int i = 0; i < mt; ++i) {
        if (w.hl[i]) mufull += (tl[i] + alpha * dtl[i]) * (zl[i] + alpha * dzl[i]);
        if (w.hu[i]) mufull += (tu[i] + alpha * dtu[i]) * (zu[i] + alpha * dzu[i]);
      }
      mufull = mufull / cnt / gamma_a;
      double a_h = (-bp + mufull / (bd + alpha * bdd)) / bdp;
      alpha = fmin(1.0, fmin(0.99999999 * alpha, fmax(a_h, gamma_f * alpha))); /* stay strictly interior */
    } else alpha = 1.0;
    if (opts.verbose) { double sm = 0, xm = 1; for (int j = 0; j < n; ++j) { sm = fmax(sm, fabs(alpha * dx[j] * w.E[j])); xm = fmax(xm, fabs(x[j] * w.E[j])); } printf("      alpha_aff %.3e sigma %.3e alpha %.4f step_rel %.3e\n", alpha_aff, sigma, alpha, sm / xm); }
    for (int j = 0; j < n; ++j) x[j] += alpha * dx[j];
    for (int i = 0; i < mt; ++i) {
      if (w.hl[i]) { tl[i] += alpha * dtl[i]; zl[i] += alpha * dzl[i]; }
      if (w.hu[i]) { tu[i] += alpha * dtu[i]; zu[i] += alpha * dzu[i]; }
    }
    /* divergence heuristics -> qpOASES exit codes (qpOASES.m:43-47) */
    double xn = 0, zn = 0;
    for (int j = 0; j < n; ++j) xn = fmax(xn, fabs(x[j]));
    for (int i = 0; i < mt; ++i) zn = fmax(zn, fmax(w.hl[i] ? zl[i] : 0, w.hu[i] ? zu[i] : 0));
    if (xn > 1e13) { flag = rp_rel > 1e-6 ? -2 : (fval_it < -1e13 ? -3 : -1); break; }
    if (zn > 1e15 && rp_rel > 1e-6) { flag = -2; break; }
    /* once an iterate met tol_loose, a handful of non-improving iterations means the end game has lost its
     * numerical footing: stop early and return the saved iterate (bounds the iteration tail of a batch) */
    if (stall > (have_saved ? 5 : 25)) { flag = rp_rel > 1e-6 && !have_saved ? -2 : ((opts.polish && !have_saved) ? 5 : 1); break; }   /* 5: stalled, refinement may certify */
  }

finish:
  if (flag == 2 || ((flag == 1 || flag == -1 || flag == 5) && have_saved)) {
    /* fall back to the last iterate that met the residual tolerances */
    for (int j = 0; j < n; ++j) x[j] = xs[j];
    for (int i = 0; i < mt; ++i) { zl[i] = lams[i] > 0 ? lams[i] : 0; zu[i] = lams[i] < 0 ? -lams[i] : 0; }
    flag = 0; last_merit = saved_merit;
  }
  /* unscale: the last iterate is returned whatever the exit code */
  {
    for (int j = 0; j < n; ++j) x_out[j] = x[j] * w.E[j];
    if (lambda_out) {
      for (int j = 0; j < n; ++j) lambda_out[j] = ((w.hl[j] ? zl[j] : 0) - (w.hu[j] ? zu[j] : 0)) / w.E[j];
      for (int r = 0; r < m; ++r) lambda_out[n + r] = ((w.hl[n + r] ? zl[n + r] : 0) - (w.hu[n + r] ? zu[n + r] : 0)) * w.F[r];
    }
    if ((flag == 0 || flag == 4 || flag == 5) && opts.polish) {
      double* lam = lambda_out;
      if (!lam) {
        lam = (double*)malloc(sizeof(double) * mt);
        for (int j = 0; j < n; ++j) lam[j] = ((w.hl[j] ? zl[j] : 0) - (w.hu[j] ? zu[j] : 0)) / w.E[j];
        for (int r = 0; r < m; ++r) lam[n + r] = ((w.hl[n + r] ? zl[n + r] : 0) - (w.hu[n + r] ? zu[n + r] : 0)) * w.F[r];
      }
      const int pol_ok = polish(n, m, H, g, A, lb, ub, lbA, ubA, opts.inf_bound, x_out, lam);
      if (flag == 4) flag = pol_ok ? 0 : -1;
      if (flag == 5) flag = pol_ok ? 0 : 1;
      pol_done = pol_ok;
      if (pol_ok) last_merit = orc_qp_kkt(n, m, H, g, A, lb, ub, lbA, ubA, x_out, lam, opts.inf_bound, 0);
      if (!lambda_out) free(lam);
    }
  }
  if (kkt_out) *kkt_out = last_merit;
  if (polished_out) *polished_out = pol_done;
  if (fval_out) {
    double f = 0;
    for (int j = 0; j < n; ++j) {
      double s = 0;
      for (int i = 0; i < n; ++i) s += H[IDX(i, j, n)] * x_out[i];
      f += 0.5 * s * x_out[j] + g[j] * x_out[j];
    }
    *fval_out = f;
  }
  if (iter_out) *iter_out = it;
  free(w.H); free(w.g); free(w.A); free(w.l); free(w.u); free(w.hl); free(w.hu); free(w.E); free(w.F);
  free(x); free(dx); free(dxa); free(tl); free(tu); free(zl); free(zu); free(v); free(dv); free(wv); free(D);
  free(rpl); free(rpu); free(dtl); free(dtu); free(dzl); free(dzu); free(cl); free(cu);
  free(Hx); free(Gz); free(rhs); free(tmp); free(M); free(xs); free(lams);
  return flag;
}

int orc_ltv_step(int model, int N, double dt, const orc_spline* sp,
                 const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                 const orc_qp_opts* opts, double* u_opt, double* x_opt, double* slack, double* fval, int* iter) {
  const int nx = orc_nx(model), ns = orc_ns(model), nV = orc_nV(model, N), nC = orc_nC(model, N), R = nx * N;
  double* H = (double*)malloc(sizeof(double) * nV * nV);
  double* g = (double*)malloc(sizeof(double) * nV);
  double* A = (double*)malloc(sizeof(double) * (size_t)nC * nV);
  double* lb = (double*)malloc(sizeof(double) * nV), *ub = (double*)malloc(sizeof(double) * nV);
  double* lbA = (double*)malloc(sizeof(double) * nC), *ubA = (double*)malloc(sizeof(double) * nC);
  double* A_bar = (double*)malloc(sizeof(double) * R * nx);
  double* Bt = (double*)malloc(sizeof(double) * (size_t)R * nV);
  double* d_bar = (double*)malloc(sizeof(double) * R);
  double* z = (double*)malloc(sizeof(double) * nV);
  double qc, fv;
  orc_ltv_build_qp(model, -1, N, dt, sp, x0, x_ref, x_lin, u_lin, H, g, A, lb, ub, lbA, ubA, A_bar, Bt, d_bar, &qc);
  int flag = orc_qp_solve(nV, nC, H, g, A, lb, ub, lbA, ubA, opts, z, &fv, iter, 0); /* ltvmpc_*.m:52 */
  /* ltvmpc_*.m:57-60 */
  for (int s = 0; s < ns; ++s) slack[s] = z[2 * N + s];
  for (int r = 0; r < R; ++r) {
    double s = d_bar[r];
    for (int c = 0; c < nx; ++c) s += A_bar[IDX(r, c, R)] * x0[c];
    for (int c = 0; c < nV; ++c) s += Bt[IDX(r, c, R)] * z[c];
    x_opt[r] = s;
  }
  for (int c = 0; c < 2 * N; ++c) u_opt[c] = z[c];
  *fval = fv + qc;
  free(H); free(g); free(A); free(lb); free(ub); free(lbA); free(ubA); free(A_bar); free(Bt); free(d_bar); free(z);
  return flag;
}

int orc_qp_solve_batch_ex(int nV, int nC, int batch, const double* H, const double* g, const double* A,
                          const double* lb, const double* ub, const double* lbA, const double* ubA,
                          const orc_qp_opts* opts, double* x, double* fval, int* exitflag, int* iter,
                          double* lambda, double* kkt, int* polished, int threads) {
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
  used = omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < batch; ++b) {
    int it = 0, pol = 0; double fv = 0, kk = 0;
    int fl = orc_qp_solve_ex(nV, nC, H + (size_t)b * nV * nV, g + (size_t)b * nV, A + (size_t)b * nC * nV,
                             lb + (size_t)b * nV, ub + (size_t)b * nV, lbA + (size_t)b * nC, ubA + (size_t)b * nC, opts,
                             x + (size_t)b * nV, &fv, &it, lambda ? lambda + (size_t)b * (nV + nC) : 0, &kk, &pol);
    if (fval) fval[b] = fv;
    if (exitflag) exitflag[b] = fl;
    if (iter) iter[b] = it;
    if (kkt) kkt[b] = kk;
    if (polished) polished[b] = pol;
  }
  return used;
}

int orc_qp_solve_batch(int nV, int nC, int batch, const double* H, const double* g, const double* A,
                       const double* lb, const double* ub, const double* lbA, const double* ubA,
                       const orc_qp_opts* opts, double* x, double* fval, int* exitflag, int* iter,
                       double* lambda, int threads) {
  return orc_qp_solve_batch_ex(nV, nC, batch, H, g, A, lb, ub, lbA, ubA, opts, x, fval, exitflag, iter, lambda, 0, 0, threads);
}
