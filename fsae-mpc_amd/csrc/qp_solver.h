// qp_solver.h -- internal interface between the C ABI (capi.hip) and the QP kernels (qp_solver.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define QP_MAX_T 12
#ifndef QP_WG_W
#define QP_WG_W 8                 // wavefronts per QP of the workgroup kernel
#endif
// an iterate on which the factorisation breaks down gets one try at the active-set refinement if it is this close to primal
// feasibility and complementarity (both kernels and the oracle: ltv_oracle_qp.c)
#define QP_BREAKDOWN_TRY_TOL 1e-4

// owner-layout row arrays of the workspace ([array][slot][64], rowlen doubles each; qp_make_dims reserves R_NARR of them)
enum RowArr { R_L = 0, R_U, R_TL, R_TU, R_ZL, R_ZU, R_V, R_D, R_W1, R_W2, R_W3, R_VA, R_VC, R_RPL, R_RPU, R_CB1, R_CC1, R_CB2, R_CC2, R_NARR };

struct QpDims {
  int n, m;        // variables, general rows
  int T, np;       // 16-wide column tiles of the MFMA core (nc = 16T columns), padded vector length
  int nb, NB, nc;  // border columns (n mod 16 in 1..4 are kept off the matrix cores), template border width (0,1,4), core columns
  int Kq, ntr;     // MFMA k-steps = rows per lane group = ceil(m/4); trips of 4 k-steps
  int prep_tw;     // columns staged per pass of the prep kernel's A transpose
  int J, JB;       // owner-layout slots for rows / for variable bounds
  int nu;          // the CALLER's number of variables.  n above is the solver's: n = nu except when trailing slack columns are kept as
                   // the border although nu mod 16 is not 1..4 (qp_make_dims): then the core is padded to 16 T columns with dummy
                   // variables (unit Hessian diagonal, no bounds, zero columns of A: they stay 0) and n = 16 T + nb.  Solver index i is
                   // real iff i < nu - nb or i >= nc; qp_user_index maps it to the caller's
  int rowlen;      // (J+JB)*64
  size_t off_Aw, off_meta, off_Hw, off_gw, off_E, off_F, off_Ab, off_Hb, off_rows, off_save, off_bad, ws_per_qp;  // in doubles (off_bad: 1.0 if the prep kernel met NaN / Inf in the QP's data)
  size_t lds_solve, lds_prep;                                                  // in bytes
  int W;                  // wavefronts per QP of the workgroup solve kernel (qp_wg.hip)
  int NBk;                // border width of the workgroup kernel's variant: 0 or 4
  size_t wg_ring;         // (reserved; always 1: pass 1 of the workgroup kernel reads the operand stream through its LDS ring for every shape)
  size_t lds_wg;          // total dynamic LDS of the workgroup solve kernel
  size_t off_U;           // workgroup kernel: the Cholesky factor's tiles as register images [tile][4][64] (read by the solving wave), in doubles
};

struct QpParams {
  QpDims d;
  const double *H, *g, *A, *lb, *ub, *lbA, *ubA;
  double* ws;
  double *x, *fval, *lambda;
  int *exitflag, *iter;
  double tol, tol_loose, tol_x, inf_bound;
  int max_iter, shared_HA, polish;
  int reserved0;
  int* polished;   // optional per-instance output: >0 if the active-set refinement was accepted (attempt count), <0 reason of rejection
  double* kkt;     // optional per-instance output: relative KKT residual of the returned point as the kernel measured it
  double* dump; int dump_stage, dump_iter;
  const double* x_init;   // optional starting point, batch x nu in the caller's variables (null: x = clamp(0, lb, ub)); clamped to the bounds.
                          // Slacks and multipliers start as always (profiles/round3/warm_start_ab.json: what that is worth)
  int* score;             // optional, batch ints behind the per-QP workspaces: difficulty estimate written by the prep kernel (rows and bounds
                          // that exclude x = 0: the best cheap predictor of the iteration count found, profiles/round3/launch_order.txt)
  const int* score_in;    // optional: the caller's difficulty estimate (fsaempc_qp_aux.difficulty) instead of the prep kernel's own
  int* order;             // optional, batch ints: workgroup -> instance, hardest-looking first (qp_order_kernel).  One wavefront / workgroup
                          // per QP is dispatched in index order, so a batch ends with its last-started instances: starting the long ones
                          // first shortens that tail.  Null: identity.  Results do not depend on it (every QP is solved on its own)
};
// launch order: batches of more than QP_ORDER_MIN_BATCH instances are solved hardest-looking first
#define QP_ORDER_MIN_BATCH 256
inline size_t qp_order_bytes(int batch) { return batch > QP_ORDER_MIN_BATCH ? (((size_t)2 * batch * sizeof(int)) + 255) & ~(size_t)255 : 0; }

void qp_make_dims(int n, int m, QpDims* d);
// solver index -> caller's variable index (-1: dummy padding variable), and back
__host__ __device__ inline int qp_user_index(const QpDims& d, int i) { const int ncu = d.nu - d.nb; return i < ncu ? i : (i >= d.nc && d.nb > 0 ? ncu + (i - d.nc) : (i < d.nu && d.nb == 0 ? i : -1)); }
__host__ __device__ inline int qp_solver_index(const QpDims& d, int u) { const int ncu = d.nu - d.nb; return (d.nb > 0 && u >= ncu) ? d.nc + (u - ncu) : u; }
bool qp_runs_wavefront_kernel(const QpDims& d);   // kernel selection of qp_launch
hipError_t qp_launch(const QpParams& P, int batch, hipStream_t st, hipEvent_t ev_mid = nullptr);
int qp_selftest_mfma(char* msg, int msglen);
// LDS bytes of the workgroup solve kernel (qp_wg.hip); NBk = border width of the kernel variant (0 or 4).  Mirrors the carve at
// the top of qp_wg_kernel.
#define QP_WG_NVEC_FIXED 14   /* X G HX P1 P2 P3 DX E R1 R2 + DV W1V W2V LV */
inline size_t qp_wg_lds_base_bytes(const QpDims& d, int W, int NBk) {
  const size_t part2 = (size_t)W * d.np;                                        // second set of partial n-vectors ...
  const size_t extra = (size_t)240 * d.T > part2 ? (size_t)240 * d.T : part2;   // ... or the tail of the three-chunk operand ring laid over U_KK^-T tiles + panel buffer + these
  return ((size_t)(QP_WG_NVEC_FIXED + 2 * NBk) * d.np + (size_t)d.T * 272 + (size_t)d.T * 256 + 256 + (size_t)W * 96 +
          (size_t)2 * 8 * W + (size_t)W * 6 * 64 + (size_t)3 * (16 + 256) + 16 + extra + (size_t)(3 * d.ntr / 2 + 2)) * sizeof(double);   // last term: the stream directory (3 ntr + 1 ints)
}

