// qp_solver.h -- internal interface between the C ABI (capi.hip) and the QP kernels (qp_solver.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define QP_MAX_T 12
#define QP_WG_RES_MAX_T 5      // tile counts for which the LDS-resident variant (and the 1-column border) of qp_wg.hip is built (development builds)
#define QP_FLAG_PENDING 99

struct QpDims {
  int n, m;        // variables, general rows
  int T, np;       // 16-wide column tiles of the MFMA core (nc = 16T columns), padded vector length
  int nb, NB, nc;  // border columns (n mod 16 in 1..4 are kept off the matrix cores), template border width (0,1,4), core columns
  int Kq, ntr;     // MFMA k-steps = rows per lane group = ceil(m/4); trips of 4 k-steps
  int prep_tw;     // columns staged per pass of the prep kernel's A transpose
  int J, JB;       // owner-layout slots for rows / for variable bounds
  int ld;          // leading dimension of the LDS normal matrix
  int rowlen;      // (J+JB)*64
  size_t off_Aw, off_meta, off_Hw, off_gw, off_E, off_F, off_Ab, off_Hb, off_rows, off_save, off_bad, ws_per_qp;  // in doubles (off_bad: 1.0 if the prep kernel met NaN / Inf in the QP's data)
  size_t lds_solve, lds_prep;                                                  // in bytes
  int W;                  // wavefronts per QP of the workgroup solve kernel (qp_wg.hip)
  int NBk;                // border width of the kernel variant: 0, 1 (only nb == 1 and T <= QP_WG_RES_MAX_T) or 4
  size_t lds_aw_bytes;    // LDS reserved for the resident operand stream (0: the passes read it from global memory)
  size_t lds_wg;          // total dynamic LDS of the workgroup solve kernel
};

struct QpParams {
  QpDims d;
  const double *H, *g, *A, *lb, *ub, *lbA, *ubA;
  double* ws;
  double *x, *fval, *lambda;
  int *exitflag, *iter;
  double tol, tol_loose, tol_x, inf_bound;
  int max_iter, shared_HA, polish;
  int only_pending;   // workgroup kernel, streaming variant: solve only the instances the resident variant handed over (exit flag QP_FLAG_PENDING)
  int* polished;   // optional per-instance output: >0 if the active-set refinement was accepted (attempt count), <0 reason of rejection
  double* kkt;     // optional per-instance output: relative KKT residual of the returned point as the kernel measured it
  double* dump; int dump_stage, dump_iter;
};

void qp_make_dims(int n, int m, QpDims* d);
bool qp_runs_wavefront_kernel(const QpDims& d);   // kernel selection of qp_launch
hipError_t qp_launch(const QpParams& P, int batch, hipStream_t st, hipEvent_t ev_mid = nullptr);
int qp_selftest_mfma(char* msg, int msglen);
// LDS bytes of the workgroup solve kernel (qp_wg.hip) without the resident operand stream; NBk = border width of the
// kernel variant (0, 1 or 4).  Mirrors the carve at the top of qp_wg_kernel.
#define QP_WG_NVEC_FIXED 14   /* X G HX P1 P2 P3 DX E R1 R2 + DV W1V W2V LV */
inline size_t qp_wg_lds_base_bytes(const QpDims& d, int W, int NBk, bool res) {
  const size_t JS = (size_t)d.J * 64;
  return ((size_t)(QP_WG_NVEC_FIXED + 2 * NBk) * d.np + (size_t)d.T * 272 + (size_t)d.T * 256 + 256 + (size_t)W * 96 +
          (size_t)2 * 8 * W + (6 + (res ? (size_t)NBk : 0)) * JS) * sizeof(double);
}

