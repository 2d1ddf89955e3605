/*
 * fsaempc.h -- C ABI of libfsaempc.so: the MI355X (gfx950) batched LTV-MPC QP path.
 *
 * This library is the drop-in for the ONE hot path of kerry-he/fsae-mpc:
 *   linearise -> condense -> build QP -> solve -> post-solve, for many independent instances.
 * Every entry point cites the reference interface it replaces (paths relative to the
 * reference repo).  Plain C types only; device entry points take raw device pointers and a
 * hipStream_t passed as void*.  All matrices are column-major (MATLAB layout); a batch is
 * stacked instance-major (instance b starts at b * <elements per instance>).
 *
 * Return value of every function: 0 on success, <0 = FSAEMPC_ERR_* (argument / runtime
 * errors -- the analogue of the MEX gateway's mexErrMsgTxt, never a solver outcome).
 * Solver outcomes are per-instance exit flags with qpOASES semantics
 * (optimizers/matlab/qpOASES/qpOASES.m:43-47): 0 solved, 1 iteration limit,
 * -1 internal error, -2 infeasible, -3 unbounded.  Whatever the flag, x / fval / lambda carry the last iterate
 * (NaN only where the data already held one): the reference's loop keeps driving on what the solver returned
 * (main.m:163-175).  -3 is returned only when the objective follows a diverging iterate to -infinity.
 */
#ifndef FSAEMPC_H
#define FSAEMPC_H

#ifdef __cplusplus
extern "C" {
#endif

#define FSAEMPC_ERR_ARG      (-1)  /* bad argument (null pointer, negative size, NaN in data ...) */
#define FSAEMPC_ERR_DIM      (-2)  /* unsupported dimension (nV > FSAEMPC_MAX_NV) */
#define FSAEMPC_ERR_HIP      (-3)  /* HIP runtime error (see fsaempc_last_error) */
#define FSAEMPC_ERR_NODEVICE (-4)  /* no gfx950 device / code object not loadable: there is NO CPU fallback */
#define FSAEMPC_ERR_WORKSPACE (-5) /* workspace too small */
#define FSAEMPC_ERR_SOLVER   (-6)  /* a call whose contract is "solve or fail" could not solve (fsaempc_seq_equality) */

#define FSAEMPC_MAX_NV 196   /* 12 column tiles of 16 + up to 4 border columns (config 5: dynamic N = 80, nV = 164) */

#define FSAEMPC_MODEL_KINEMATIC 0  /* mpc/ltv/kinematic/ltvmpc_kinetmatic_curvilinear.m */
#define FSAEMPC_MODEL_DYNAMIC   1  /* mpc/ltv/dynamic/ltvmpc_dynamic_curvilinear.m */

/* Solver options; the reference passes none (=> qpOASES defaults, qpOASES_options.m:180-213).
 * fsaempc_qp_default_opts fills the defaults of this build. */
typedef struct {
  double tol;        /* strict relative KKT tolerance (default 1e-8) */
  double tol_loose;  /* fall-back KKT tolerance = the specified 1e-6 */
  double tol_x;      /* Newton-decrement test on the affine direction (default 1e-7) */
  double inf_bound;  /* |bound| >= inf_bound is treated as infinite (default 1e9; covers the
                        reference's +-1e10 fillers, kinematic_state_constraints.m:38-39) */
  int    max_iter;   /* interior-point iteration limit (default 100) */
  int    polish;     /* 1 (default): active-set polish to the vertex-exact point an active-set solver returns;
                        accepted only if it is a KKT point, otherwise the interior-point iterate is kept */
} fsaempc_qp_opts;

void fsaempc_qp_default_opts(fsaempc_qp_opts* o);

/* Problem descriptor of one batched solve.
 * shared_HA != 0: H and A are given once and shared by all `batch` instances (the reference
 * API's own multi-column form, qpOASES.m:65-67); otherwise H and A are stacked per instance. */
typedef struct {
  int nV;        /* number of variables */
  int nC;        /* number of general constraint rows (0 => bounds-only form, qpOASES.m:34-35) */
  int batch;     /* number of independent QPs */
  int shared_HA;
} fsaempc_qp_desc;

/* Bytes of device workspace fsaempc_qp_solve_batch_device needs for `desc`. */
long long fsaempc_qp_workspace_bytes(const fsaempc_qp_desc* desc);

/*
 * Replaces: [x,fval,exitflag,iter,lambda] = qpOASES(H,g,A,lb,ub,lbA,ubA)
 *           optimizers/matlab/qpOASES/qpOASES.m:22-23 (call sites
 *           mpc/ltv/kinematic/ltvmpc_kinetmatic_curvilinear.m:52,
 *           mpc/ltv/dynamic/ltvmpc_dynamic_curvilinear.m:52), batched.
 * Device pointers, asynchronous on `stream`.  H: nV*nV, g: nV, A: nC*nV (column-major),
 * lb/ub: nV, lbA/ubA: nC per instance; +-inf allowed in bounds.
 * Outputs: x nV, fval 1, exitflag 1 (int), iter 1 (int) per instance; lambda (nV+nC per
 * instance, bounds first, >=0 lower side / <=0 upper side) may be NULL.
 * `workspace` must hold fsaempc_qp_workspace_bytes(desc) bytes.
 */
int fsaempc_qp_solve_batch_device(const fsaempc_qp_desc* desc,
                                  const double* H, const double* g, const double* A,
                                  const double* lb, const double* ub, const double* lbA, const double* ubA,
                                  const fsaempc_qp_opts* opts,
                                  double* x, double* fval, int* exitflag, int* iter, double* lambda,
                                  void* workspace, long long workspace_bytes, void* stream);

/* Optional per-instance diagnostics of a batched solve -- the analogue of qpOASES' sixth output `auxOutput`
 * (optimizers/matlab/qpOASES/qpOASES.m:55-62).  Device arrays of `batch` entries, each may be NULL.
 *   kkt:      relative KKT residual (max of stationarity, primal feasibility, complementarity) of the returned point as
 *             the solver measured it.  Exit flag 0 covers both iterates converged to opts->tol and the fall-back iterate
 *             that met opts->tol_loose; this value tells them apart.
 *   polished: > 0 the active-set refinement was accepted (the returned point is the vertex, value = attempts used),
 *             0 not attempted, < 0 rejected (the interior-point iterate is returned). */
typedef struct {
  double* kkt;
  int* polished;
  const double* x_init;   /* optional INPUT (device, nV per instance; NULL = none): starting point of the interior-point iteration,
                             clamped to the bounds -- the primal part of what qpOASES' auxInput.x0 / a hot start carries.  Slacks and
                             multipliers start as always.  Measured on closed-loop QPs (shifted previous plan vs cold):
                             profiles/round3/warm_start_ab.json */
  const int* difficulty;  /* optional INPUT (device, one int per instance; NULL = none): the caller's estimate of each instance's solve
                             effort, any monotone measure -- e.g. the iteration count of the same car's QP one MPC period earlier.
                             Batches of more than 256 instances are launched hardest-looking first (one wavefront / workgroup per
                             QP is dispatched in order, so a batch ends with its last-started instances); without this array the
                             library ranks by the number of rows and bounds that exclude x = 0.  Only the launch order depends on it,
                             never a result (profiles/round3/launch_order.txt) */
} fsaempc_qp_aux;

int fsaempc_qp_solve_batch_device_aux(const fsaempc_qp_desc* desc,
                                      const double* H, const double* g, const double* A,
                                      const double* lb, const double* ub, const double* lbA, const double* ubA,
                                      const fsaempc_qp_opts* opts,
                                      double* x, double* fval, int* exitflag, int* iter, double* lambda,
                                      const fsaempc_qp_aux* aux,
                                      void* workspace, long long workspace_bytes, void* stream);

/* Same call on host pointers: copies to the device, solves, copies back, synchronises.
 * This is what a MEX gateway calls (mex/qpOASES.cpp); validates like the original gateway
 * (NaN anywhere / Inf in H,g,A => FSAEMPC_ERR_ARG).  The _device entries cannot inspect device data before the launch: an
 * instance with NaN / Inf in H, g, A (or NaN in a bound) returns exitflag -1 with iter = 0 and x = clamp(0, lb, ub); the other
 * instances of the batch are unaffected. */
int fsaempc_qp_solve_batch(const fsaempc_qp_desc* desc,
                           const double* H, const double* g, const double* A,
                           const double* lb, const double* ub, const double* lbA, const double* ubA,
                           const fsaempc_qp_opts* opts,
                           double* x, double* fval, int* exitflag, int* iter, double* lambda);

/* ---- qpOASES_sequence: handle-based solves of a sequence of QPs -------------------------------
 * Replaces optimizers/matlab/qpOASES/qpOASES_sequence.m:23 ('i'), :39 ('h'), :51 ('m'), :76 ('c')
 * (commented call sites ltvmpc_kinetmatic_curvilinear.m:44-50, live cleanup main.m:193).  Host pointers,
 * one QP per call (k columns of g/lb/ub/lbA/ubA => k QPs, as in qpOASES.m:65-67).  The handle owns device copies of H and
 * A (uploaded by 'i' and 'm' only), the solver workspace and the per-call vectors, so a hot start transfers (3 nV + 2 nC) k
 * doubles and the results, nothing else.  Every call is a COLD interior-point solve: same results as a qpOASES hot start, but
 * the previous iterate is not used as a starting point (measured: 3-10 % fewer iterations per QP, profiles/round3/warm_start_ab.json,
 * DESIGN.md 6d; fsaempc_qp_aux.x_init of the batched entries takes a starting point for callers that want one).
 * fsaempc_seq_equality is qpOASES_sequence.m:64 ('e'): the equality-constrained QP fixed by the working set of the handle's
 * last 'i'/'h'/'m' solve (first column; a side is in the set iff its multiplier has the side's sign and exceeds the side's
 * slack -- on a refined vertex: iff the multiplier is non-zero); it returns FSAEMPC_ERR_SOLVER when that QP has no solution and
 * leaves the handle as is.
 * Errors mirror the gateway: unknown handle => FSAEMPC_ERR_ARG "Invalid handle to QP instance!", changed
 * dimensions => FSAEMPC_ERR_ARG "QP dimensions must be constant during a sequence!". */
int fsaempc_seq_init(int nV, int nC, const double* H, const double* g, const double* A,
                     const double* lb, const double* ub, const double* lbA, const double* ubA, int k,
                     const fsaempc_qp_opts* opts, int* handle,
                     double* x, double* fval, int* exitflag, int* iter, double* lambda);      /* 'i' */
int fsaempc_seq_hotstart(int handle, int nV, int nC, const double* g, const double* lb, const double* ub,
                         const double* lbA, const double* ubA, int k, const fsaempc_qp_opts* opts,
                         double* x, double* fval, int* exitflag, int* iter, double* lambda);  /* 'h' */
int fsaempc_seq_hotstart_matrices(int handle, int nV, int nC, const double* H, const double* g, const double* A,
                                  const double* lb, const double* ub, const double* lbA, const double* ubA, int k,
                                  const fsaempc_qp_opts* opts,
                                  double* x, double* fval, int* exitflag, int* iter, double* lambda);  /* 'm' */
int fsaempc_seq_equality(int handle, int nV, int nC, const double* g, const double* lb, const double* ub,
                         const double* lbA, const double* ubA, int k, const fsaempc_qp_opts* opts,
                         double* x, double* lambda, int* workingSetB, int* workingSetC);        /* 'e' */
int fsaempc_seq_cleanup(int handle);                                                         /* 'c' */

/* ---- LTV-MPC step (QP construction + solve + post-solve) ---------------------------------- */

/* Track spline table: the `kappa` closure of main.m:18 as data.  xP,yP: M x 4 column-major. */
typedef struct {
  int M;
  double dl;
  const double* xP;   /* device pointers for *_device entry points, host pointers otherwise */
  const double* yP;
} fsaempc_spline;

#define FSAEMPC_INT_DEFAULT (-1)  /* the one the reference driver calls: RK2 kinematic, RK4 dynamic (ltvmpc_*.m:38) */
#define FSAEMPC_INT_EULER 0       /* mpc/ltv/{kinematic,dynamic}/euler_*_curvilinear.m:24-30 */
#define FSAEMPC_INT_RK2   1       /* rk2_*_curvilinear.m:25-50 (midpoint rule) */
#define FSAEMPC_INT_RK4   2       /* rk4_*_curvilinear.m:25-59 */

typedef struct {
  int model;     /* FSAEMPC_MODEL_* */
  int N;         /* horizon steps (main.m:36) */
  int batch;
  double dt;     /* main.m:37 */
  int integrator; /* FSAEMPC_INT_*: lineariser of the continuous model (the reference keeps all three per model) */
} fsaempc_ltv_desc;

int fsaempc_ltv_nx(int model);            /* 5 / 7 */
int fsaempc_ltv_nV(int model, int N);     /* 2N + slack count */
int fsaempc_ltv_nC(int model, int N);     /* 6N / 20N */

/*
 * Replaces the QP construction of ltvmpc_*_curvilinear.m:38-41 (rk2/rk4 lineariser,
 * sequential_integration.m, *_state_constraints.m, generate_qp.m), batched, on device.
 * Inputs per instance: x0 nx, x_ref nx*N, x_lin nx*N, u_lin 2*N.
 * Outputs per instance: H,g,A,lb,ub,lbA,ubA as for the solver; pred = [A_bar*x0 + d_bar] (nx*N),
 * Bt (nx*N x nV, B_bar with slack columns), qconst 1.  Bt/pred/qconst may be NULL.
 */
int fsaempc_ltv_build_qp_batch_device(const fsaempc_ltv_desc* desc, const fsaempc_spline* sp,
                                      const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                                      double* H, double* g, double* A, double* lb, double* ub, double* lbA, double* ubA,
                                      double* pred, double* Bt, double* qconst, void* stream);

/* Bytes of device workspace for fsaempc_ltv_step_batch_device (QP tensors + solver workspace). */
long long fsaempc_ltv_workspace_bytes(const fsaempc_ltv_desc* desc);

/*
 * Replaces [u_opt,x_opt,QP,exitflag,fval,slack_opt] = ltvmpc_*_curvilinear(x0,x_ref,kappa,dt,x_lin,u_lin,QP)
 * (ltvmpc_kinetmatic_curvilinear.m:1, ltvmpc_dynamic_curvilinear.m:1), batched, on device.
 * Outputs per instance: u_opt 2N, x_opt nx*N, slack ns, fval 1 (incl. the constant, :60), exitflag, iter.
 */
int fsaempc_ltv_step_batch_device(const fsaempc_ltv_desc* desc, const fsaempc_spline* sp,
                                  const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                                  const fsaempc_qp_opts* opts,
                                  double* u_opt, double* x_opt, double* slack, double* fval, int* exitflag, int* iter,
                                  void* workspace, long long workspace_bytes, void* stream);

/* Same step with the per-instance diagnostics of the solve (fsaempc_qp_aux: achieved KKT residual, refinement outcome);
 * aux may be NULL.  The reference's drivers hand back the solver object `QP` at this position of their output list
 * (ltvmpc_*.m:1); a batched build has no such object, the diagnostics take its place. */
int fsaempc_ltv_step_batch_device_aux(const fsaempc_ltv_desc* desc, const fsaempc_spline* sp,
                                      const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                                      const fsaempc_qp_opts* opts,
                                      double* u_opt, double* x_opt, double* slack, double* fval, int* exitflag, int* iter,
                                      const fsaempc_qp_aux* aux,
                                      void* workspace, long long workspace_bytes, void* stream);

/* ---- reference trajectories ---------------------------------------------------------------- */

/*
 * Replaces x_ref = obtain_reference(x, ds, N_s, t, s0, dt, N_t)   (util/obtain_reference.m:1-48; call site
 * main.m:115, commented in the live loop), batched over s0.  plan: the planner vector x (8 values per s-cell:
 * n, mu, x_d, y_d, theta_d, delta, a, delta_d; obtain_reference.m:7-15), t: per-cell traversal times (N_s),
 * s0: `batch` arc-length positions.  Output per instance: x_ref 7 x N_t column-major, row 1 =
 * s0 + mod(idx+rto-idx_1-rto_1, N_s)*ds, rows 2..7 linear interpolation of the six planner states.
 * Device pointers, asynchronous on `stream`.
 */
int fsaempc_obtain_reference_batch_device(const double* plan, double ds, int N_s, const double* t,
                                          const double* s0, double dt, int N_t, int batch,
                                          double* x_ref, void* stream);

/*
 * Replaces the live reference generator of main.m:107-114 (velocity ramp +-10 m/s^2 clipped at target_vel,
 * s_ref = s0 + cumsum(v_ref*dt), all other states 0), batched.  x0: batch x nx, x_ref: batch x (nx x N).
 */
int fsaempc_reference_live_batch_device(int nx, int N, double dt, double target_vel, int batch,
                                        const double* x0, double* x_ref, void* stream);

/* ---- closed loop around the step (main.m:91-179), batched: one car per instance ------------------- */

/*
 * Replaces main.m:93-114 per car: [s,n,mu] = cartesian_to_curvilinear(x(1),x(2),x(3),x_spline,y_spline,dl,x_opt(1))
 * (vehicle_models/cartesian_to_curvilinear.m:17-26, spline/closest_point.m:15-32 with epsilon 0.01), the x0 assembly
 * for the model (:94-98), the lap check s >= L (:101-104, sets finished[b] = 1; 2 = the closest-point search diverged,
 * the car left the track) and the live reference (:107-114).
 * cart: batch x 7 [x,y,theta,x_d,y_d,theta_d,delta]; s_guess: batch (first predicted s of the previous plan).
 * Outputs x0 (batch x nx), x_ref (batch x (nx x N)).
 */
int fsaempc_cl_pre_batch_device(int model, int N, double dt, double target_vel, double L, const fsaempc_spline* sp,
                                const double* cart, const double* s_guess, int batch,
                                double* x0, double* x_ref, int* finished, void* stream);

/*
 * Replaces main.m:163-175 per car: set points v_ref = x_opt(4), delta_ref = x_opt(N_x) from this step's plan, then ten
 * sub-steps of pid_controller (vehicle_models/pid_controller.m; gains main.m:84-88) + integrate_cart_dyn(x, u, dt/10)
 * (vehicle_models/cartesian_dynamic/integrate_cart_dyn.m, f_cart_dyn.m).  cart (batch x 7) and pid (batch x 4:
 * velocity integral / last error, steering integral / last error) are updated in place; cars with finished[b] != 0 or
 * a non-finite set point keep their state; exitflag (optional) holds a car only for values < -100 (a caller's own marker,
 * never a solver outcome: the reference drives on whatever the solver returned, main.m:163-175); u_last (optional,
 * batch x 2) = last actuator rates.
 */
int fsaempc_cl_plant_batch_device(int model, int N, double dt, int batch, double* cart, double* pid, const double* x_opt,
                                  const int* finished, const int* exitflag, double* u_last, void* stream);

/*
 * Replaces the hand-over of main.m:122-126 per car: this step's plan (x_new: nx*N, u_new: 2*N) becomes the linearisation
 * point and set-point source of the next step (x_keep, u_keep) when the solve ended with exit flag 0 or 1 (exitflag may
 * be NULL: every finite plan is taken) and the plan is finite; otherwise the car keeps its last good plan and keeps
 * driving on it.  (The reference takes over whatever qpOASES returned; the last iterate of an interior-point method
 * after an abnormal exit need not respect the actuator bounds, hence the deviation.)
 */
int fsaempc_cl_accept_batch_device(int model, int N, int batch, const double* x_new, const double* u_new, const int* exitflag,
                                   double* x_keep, double* u_keep, void* stream);

/* ---- track pipeline (host side) --------------------------------------------------- */

/* Spline table of a track as main.m:11-17 produces it: M arc-length segments, xP / yP = M x 4 Bezier control points per axis
 * (column-major: all P0, then all P1, P2, P3), dl = segment length, L = total length.  Host memory owned by the library
 * (fsaempc_track_free).  These are the tables fsaempc_spline points at (after a copy to the device). */
typedef struct {
  int M;
  double dl, L;
  double* xP;
  double* yP;
} fsaempc_track;

/* Replaces main.m:11-17: [x,y,...] = read_raceline_csv(file) (util/read_raceline_csv.m:6-19: one header line, columns 1-2 = X, Y),
 * x_spline = make_spline_periodic(x) (spline/make_spline_periodic.m:9-33), likewise y,
 * [x_spline,y_spline,dl,L] = arclength_reparam(x_spline,y_spline,M,true) (spline/arclength_reparam.m:15-64; M = 100 in main.m:17).
 * The reference's quirk in the speed integrand (arclength_reparam.m:20-23) is kept.  Host only, no GPU work. */
int fsaempc_track_from_csv(const char* path, int M, fsaempc_track* out);
int fsaempc_track_from_points(const double* x, const double* y, int n, int M, fsaempc_track* out);
void fsaempc_track_free(fsaempc_track* t);
/* On-disk table format "FSTRK001" (little endian): 8-byte magic, int32 M, int32 0, double dl, double L, xP (4M doubles), yP (4M). */
int fsaempc_track_save(const fsaempc_track* t, const char* path);
int fsaempc_track_load(const char* path, fsaempc_track* out);
const char* fsaempc_track_last_error(void);

/* ---- diagnostics ---------------------------------------------------------------------------- */
const char* fsaempc_last_error(void);
/* Runs the on-device fp64 MFMA layout self-test (v_mfma_f64_16x16x4_f64 operand / accumulator
 * lane maps the kernels rely on).  Returns 0 if the hardware matches, >0 number of mismatches. */
int fsaempc_selftest_mfma(void);
/* Debug hook of the diagnostic builds only (libfsaempc_dbg.so, -DQP_DEBUG_DUMP; the shipped kernels carry no dump
 * branches and ignore it): dumps solver internals of instance 0 after `stage` (see qp_solver.hip) into `out` (device
 * pointer, >= 4*nV*nV+8*(nV+nC) doubles).  Process-global, not thread-safe. */
int fsaempc_debug_set_dump(double* out, int stage);

/* Kernel timing with HIP events recorded on the launch stream of the last fsaempc_qp_solve_batch_device
 * call (prep = scaling/repack kernel, solve = interior-point kernel).  get_timing synchronises on the events.
 * Process-global switch for benchmarks (bench.py); not thread-safe. */
int fsaempc_qp_set_timing(int enable);
int fsaempc_qp_get_timing(double* prep_ms, double* solve_ms);
/* Same switch, for the last fsaempc_ltv_step_batch_device[_aux] call: the four phases of the fused step (construction kernel,
 * prep = scaling / repack / launch order, interior-point solve, post-solve kernel) by HIP events on the launch stream.
 * Every phase is also a roctx range on the calling thread (fsaempc.ltv.step > fsaempc.ltv.build, fsaempc.qp.prep+solve,
 * fsaempc.ltv.post; `rocprofv3 --marker-trace --kernel-trace`), through librocprofiler-sdk-roctx (or roctracer's libroctx64) if the process can load it (FSAEMPC_ROCTX=0
 * turns the ranges off). */
int fsaempc_ltv_get_timing(double* build_ms, double* prep_ms, double* solve_ms, double* post_ms);

#ifdef __cplusplus
}
#endif
#endif
