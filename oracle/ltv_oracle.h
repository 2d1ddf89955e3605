/*
 * ltv_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the batched LTV-MPC QP hot path of kerry-he/fsae-mpc
 * (linearise -> condense -> build QP -> solve -> post-solve).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 *
 * PARITY UNPINNED: the reference holds no golden vectors/tests for this path,
 * MATLAB/Octave are absent, and the QP solve lives in qpOASES (coin-or/qpOASES,
 * LGPL-2.1, 3.2.x era, version NOT pinned; only Windows PE binaries are in the
 * tree).  The QP *construction* follows the reference .m files line by line;
 * the QP *solve* restates the mathematical contract of
 * optimizers/matlab/qpOASES/qpOASES.m:16-62 (unique minimiser of a convex QP)
 * with an interior-point method + active-set polish, certified by KKT residuals
 * and cross-checked against scipy in tests/.
 *
 * All matrices are column-major (MATLAB layout).  Spline tables are M x 4
 * column-major as well: P[i + M*j] = control point j of segment i.
 */
#ifndef LTV_ORACLE_H
#define LTV_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MODEL_KINEMATIC 0
#define ORC_MODEL_DYNAMIC   1
#define ORC_INT_EULER 0
#define ORC_INT_RK2   1
#define ORC_INT_RK4   2

typedef struct {
  int M;            /* number of Bezier segments (main.m:17 -> 100) */
  double dl;        /* segment arc length */
  const double* xP; /* M x 4 column-major */
  const double* yP; /* M x 4 column-major */
} orc_spline;

/* spline/interpolate_spline_d.m:11-21, _dd.m:11-21, interpolate_curvature.m:12-18 */
double orc_spline_d(const double* P, int M, double dl, double t);
double orc_spline_dd(const double* P, int M, double dl, double t);
double orc_spline_val(const double* P, int M, double dl, double t);
double orc_kappa(const orc_spline* sp, double s);

/* vehicle_models/curvilinear_kinematic/{f,A,B}_curv_kin.m */
void orc_f_kin(const double* x, const double* u, const orc_spline* sp, double* f);
void orc_A_kin(const double* x, const orc_spline* sp, double* A /*5x5 colmajor*/);
/* vehicle_models/curvilinear_dynamic/{f,A,B}_curv_dyn.m ; byp = {Fcr,Fcr_d,vr,denom_vr2,x_d_hat,x_d_hat_d,vf,denom_vf2} */
void orc_f_dyn(const double* x, const double* u, const orc_spline* sp, double* f, double* Fcr);
void orc_A_dyn(const double* x, const orc_spline* sp, double* A /*7x7 colmajor*/, double* byp /*8*/);

/* mpc/ltv/{kinematic,dynamic}/{euler,rk2,rk4}_*_curvilinear.m
 * x: nx x N, u: 2 x N  ->  A: nx x nx x N, B: nx x 2 x N, d: nx x N */
void orc_linearise(int model, int integrator, int N, const double* x, const double* u,
                   const orc_spline* sp, double dt, double* A, double* B, double* d);

/* mpc/ltv/sequential_integration.m:16-47 ; B_bar is (nx N) x (2 N) */
void orc_sequential_integration(int nx, int N, const double* A, const double* B, const double* d,
                                double dt, double* A_bar, double* B_bar, double* d_bar);

/* dimensions of the condensed QP */
int orc_nx(int model);
int orc_ns(int model);                /* slack count: 1 kin / 4 dyn */
int orc_nV(int model, int N);         /* 2N + ns */
int orc_nC(int model, int N);         /* 6N kin / 20N dyn */

/* One LTV-MPC QP build (ltvmpc_kinetmatic_curvilinear.m:17-41 / ltvmpc_dynamic_curvilinear.m:17-41).
 * Outputs (all column-major):
 *   H nV x nV, g nV, A nC x nV, lb/ub nV, lbA/ubA nC,
 *   A_bar (nx N) x nx, Bt (nx N) x nV (B_bar with slack columns), d_bar nx N, qconst scalar.
 * integrator < 0 selects the one the reference driver uses (RK2 kin, RK4 dyn). */
void orc_ltv_build_qp(int model, int integrator, int N, double dt, const orc_spline* sp,
                      const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                      double* H, double* g, double* A, double* lb, double* ub, double* lbA, double* ubA,
                      double* A_bar, double* Bt, double* d_bar, double* qconst);

/* Solver options (defaults via orc_qp_default_opts) */
typedef struct {
  double tol;        /* relative KKT tolerance of the IPM */
  double tol_loose;  /* fall-back KKT tolerance (the specified 1e-6), see ltv_oracle_qp.c */
  double tol_x;      /* Newton-decrement test: |dx_aff|_inf <= tol_x * max(1,|x|_inf) */
  int    max_iter;
  double inf_bound;  /* |bound| >= inf_bound => side dropped (documented threshold) */
  int    polish;     /* 1: active-set polish to a vertex-exact solution */
  int    corrector;  /* 1: Mehrotra second-order corrector */
  int    scale;      /* 1: diagonal equilibration */
  int    verbose;
} orc_qp_opts;
void orc_qp_default_opts(orc_qp_opts* o);

/* min 1/2 x'Hx + g'x  s.t. lb<=x<=ub, lbA<=Ax<=ubA   (qpOASES.m:16-23)
 * lambda has nV+nC entries, bounds first, >=0 lower side active, <=0 upper (qpOASES.m:49).
 * returns exitflag: 0 solved, 1 iteration limit, -1 internal, -2 infeasible, -3 unbounded (qpOASES.m:43-47) */
int orc_qp_solve_ex(int nV, int nC, const double* H, const double* g, const double* A,
                    const double* lb, const double* ub, const double* lbA, const double* ubA,
                    const orc_qp_opts* opts, double* x, double* fval, int* iter, double* lambda,
                    double* kkt, int* polished);
int orc_qp_solve(int nV, int nC, const double* H, const double* g, const double* A,
                 const double* lb, const double* ub, const double* lbA, const double* ubA,
                 const orc_qp_opts* opts, double* x, double* fval, int* iter, double* lambda);

/* KKT certificate: returns max of the four scaled residuals; res[4] = {stationarity, primal, dual sign, complementarity} */
double orc_qp_kkt(int nV, int nC, const double* H, const double* g, const double* A,
                  const double* lb, const double* ub, const double* lbA, const double* ubA,
                  const double* x, const double* lambda, double inf_bound, double* res);

/* Full step: ltvmpc_*_curvilinear.m:17-60.  u_opt 2N, x_opt nx N, slack ns. returns exitflag. */
int orc_ltv_step(int model, int N, double dt, const orc_spline* sp,
                 const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                 const orc_qp_opts* opts, double* u_opt, double* x_opt, double* slack, double* fval, int* iter);

/* main.m:107-114 live reference generator (TARGET_VEL ramp); x_ref nx x N zero-filled then rows 1 and 4 set */
void orc_reference_live(int nx, int N, double dt, double target_vel, const double* x0, double* x_ref);
/* ---- closed loop around the step (ltv_oracle_plant.c) ---- */
double orc_closest_point(const orc_spline* sp, double x0, double y0, double s, double epsilon);   /* spline/closest_point.m */
void orc_cart_to_curv(const orc_spline* sp, double x, double y, double theta, double s0, double* s, double* n, double* mu);
void orc_f_cart_dyn(const double* x /*7*/, const double* u /*2*/, double* f /*7*/);             /* cartesian_dynamic/f_cart_dyn.m */
void orc_integrate_cart_dyn(const double* x0, const double* u, double dt, double* x);            /* integrate_cart_dyn.m */
double orc_pid(double target, double current, const double* settings /*4*/, double* status /*2*/);   /* pid_controller.m */
void orc_plant_step(double* x /*7 in/out*/, double* pid /*4 in/out*/, double v_ref, double delta_ref, double dt, double* u_last /*2*/);
int orc_cl_pre(int model, int N, double dt, double target_vel, const orc_spline* sp, double L, const double* cart /*7*/, double s_guess,
               double* x0, double* x_ref);

/* util/obtain_reference.m:5-48: x = planner vector (8 per cell), t = per-cell times; x_ref 7 x N_t column-major */
void orc_obtain_reference(const double* x, double ds, int N_s, const double* t, double s0, double dt, int N_t, double* x_ref);

/* SURVEY 8(d) synthetic instances: splitmix64 keyed by seed ^ (id*0x9E3779B97F4A7C15) */
void orc_synth_instance(int model, int N, double dt, double L, unsigned long long seed, unsigned long long id,
                        double* x0, double* x_lin, double* u_lin, double* x_ref);

/* batched drivers (OpenMP over instances) used by the cpu_baseline leg; return threads used */
int orc_ltv_build_qp_batch(int model, int N, double dt, const orc_spline* sp, int batch,
                           const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                           double* H, double* g, double* A, double* lb, double* ub, double* lbA, double* ubA,
                           double* A_bar, double* Bt, double* d_bar, double* qconst, int threads);
int orc_qp_solve_batch_ex(int nV, int nC, int batch, const double* H, const double* g, const double* A,
                          const double* lb, const double* ub, const double* lbA, const double* ubA,
                          const orc_qp_opts* opts, double* x, double* fval, int* exitflag, int* iter,
                          double* lambda, double* kkt, int* polished, int threads);
int orc_qp_solve_batch(int nV, int nC, int batch, const double* H, const double* g, const double* A,
                       const double* lb, const double* ub, const double* lbA, const double* ubA,
                       const orc_qp_opts* opts, double* x, double* fval, int* exitflag, int* iter,
                       double* lambda, int threads);

#ifdef __cplusplus
}
#endif
#endif
