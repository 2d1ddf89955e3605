// mex/qpOASES.cpp -- MEX gateway that makes libfsaempc.so a drop-in for the reference's
//     [x,fval,exitflag,iter,lambda,auxOutput] = qpOASES(H,g,A,lb,ub,lbA,ubA{,options{,auxInput}})
// (optimizers/matlab/qpOASES/qpOASES.m:22-23; bounds-only form :34-35; k-column form :65-67).
// Not BUILT in this repo (no MATLAB here); tests/test_abi_cpu.py type-checks it against tests/stub_mex/mex.h and runs it against a
// functional stand-in of the MEX API and a recording stand-in of libfsaempc (tests/stub_mex/run_gateways.cpp).
// Build on a MATLAB host:   mex -I<repo>/include mex/qpOASES.cpp -L<repo>/fsae-mpc_amd/lib -lfsaempc
#include <cmath>
#include <cstring>
#include <vector>
#include "mex.h"
#include "fsaempc.h"

// options struct of qpOASES_options.m -> fsaempc_qp_opts: maxIter (:41, -1 = automatic) bounds the iterations,
// terminationTolerance (:71, relative) is the KKT tolerance; every other field steers the active-set homotopy and has no
// counterpart in an interior-point method (ignored, as documented in INTEGRATION.md)
static void map_options(const mxArray* o, fsaempc_qp_opts* q) {
  fsaempc_qp_default_opts(q);
  if (!o || !mxIsStruct(o)) return;
  const mxArray* f = mxGetField(o, 0, "maxIter");
  if (f && !mxIsEmpty(f) && mxGetScalar(f) > 0) q->max_iter = (int)mxGetScalar(f);
  f = mxGetField(o, 0, "terminationTolerance");
  if (f && !mxIsEmpty(f) && mxGetScalar(f) > 0) { q->tol = mxGetScalar(f) > 1e-12 ? mxGetScalar(f) : 1e-12; if (q->tol_loose < q->tol) q->tol_loose = q->tol; }
}

static void dense(const mxArray* a, std::vector<double>& out) {  // qpOASES.m:30: H and A may be sparse
  const mwSize m = mxGetM(a), n = mxGetN(a);
  out.assign((size_t)m * n, 0.0);
  if (mxIsSparse(a)) {
    const mwIndex *ir = mxGetIr(a), *jc = mxGetJc(a);
    const double* pr = mxGetPr(a);
    for (mwSize j = 0; j < n; ++j)
      for (mwIndex k = jc[j]; k < jc[j + 1]; ++k) out[(size_t)j * m + ir[k]] = pr[k];
  } else {
    std::memcpy(out.data(), mxGetPr(a), sizeof(double) * m * n);
  }
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  if (nrhs < 4) mexErrMsgTxt("ERROR (qpOASES): Invalid number of input arguments!");
  const bool general = nrhs >= 7 && !mxIsStruct(prhs[4]);   // (H,g,A,lb,ub,lbA,ubA{,options}) vs (H,g,lb,ub{,options})
  const int ndata = general ? 7 : 4;                        // the numeric arguments; what follows is the options struct
  for (int i = 0; i < ndata; ++i)
    if (!mxIsEmpty(prhs[i]) && (!mxIsDouble(prhs[i]) || mxIsComplex(prhs[i]))) mexErrMsgTxt("ERROR (qpOASES): All data has to be provided in double precision!");
  fsaempc_qp_opts opts;
  map_options(nrhs > ndata ? prhs[ndata] : nullptr, &opts);
  // auxInput (qpOASES.m:23, qpOASES_auxInput.m: initial guess x0 / working set guess): accepted and ignored -- every solve of
  // this build is a cold interior-point solve followed by an active-set refinement
  if (nrhs > ndata + 1 && !mxIsEmpty(prhs[ndata + 1])) mexWarnMsgTxt("WARNING (qpOASES): auxInput is ignored (every solve is a cold start)");
  const mxArray *H = prhs[0], *g = prhs[1];
  const mxArray *A = general ? prhs[2] : nullptr, *lb = prhs[general ? 3 : 2], *ub = prhs[general ? 4 : 3];
  const mxArray *lbA = general ? prhs[5] : nullptr, *ubA = general ? prhs[6] : nullptr;
  const int nV = (int)mxGetM(H), nC = general ? (int)mxGetM(A) : 0, k = (int)mxGetN(g);
  if ((int)mxGetN(H) != nV || (int)mxGetM(g) != nV) mexErrMsgTxt("ERROR (qpOASES): Input dimension mismatch for argument 2");
  if (general && (int)mxGetN(A) != nV) mexErrMsgTxt("ERROR (qpOASES): Input dimension mismatch for argument 3");
  std::vector<double> Hd, Ad;
  dense(H, Hd);
  if (general) dense(A, Ad);
  auto cols = [&](const mxArray* a, int rows, std::vector<double>& v, double fill) {   // broadcast 1 column to k
    v.assign((size_t)rows * k, fill);
    if (!a || mxIsEmpty(a)) return;
    if ((int)mxGetM(a) != rows) mexErrMsgTxt("ERROR (qpOASES): Input dimension mismatch for argument");
    const int ca = (int)mxGetN(a);
    for (int j = 0; j < k; ++j) std::memcpy(&v[(size_t)j * rows], mxGetPr(a) + (size_t)(ca == k ? j : 0) * rows, sizeof(double) * rows);
  };
  std::vector<double> gv, lbv, ubv, lbAv, ubAv;
  cols(g, nV, gv, 0.0); cols(lb, nV, lbv, -INFINITY); cols(ub, nV, ubv, INFINITY);
  cols(lbA, nC, lbAv, -INFINITY); cols(ubA, nC, ubAv, INFINITY);
  plhs[0] = mxCreateDoubleMatrix(nV, k, mxREAL);
  std::vector<double> fval(k), lam((size_t)(nV + nC) * k);
  std::vector<int> flag(k), iter(k);
  fsaempc_qp_desc d{nV, nC, k, 1};   // k QPs sharing H and A
  const int rc = fsaempc_qp_solve_batch(&d, Hd.data(), gv.data(), general ? Ad.data() : nullptr, lbv.data(), ubv.data(),
                                        general ? lbAv.data() : nullptr, general ? ubAv.data() : nullptr, &opts,
                                        mxGetPr(plhs[0]), fval.data(), flag.data(), iter.data(), lam.data());
  if (rc != 0) mexErrMsgTxt(fsaempc_last_error());   // argument errors are MEX errors, solver outcomes are exit flags
  auto out = [&](int i, int rows, auto&& get) {
    if (nlhs > i) { plhs[i] = mxCreateDoubleMatrix(rows, k, mxREAL); for (size_t e = 0; e < (size_t)rows * k; ++e) mxGetPr(plhs[i])[e] = get(e); }
  };
  out(1, 1, [&](size_t e) { return fval[e]; });
  out(2, 1, [&](size_t e) { return (double)flag[e]; });
  out(3, 1, [&](size_t e) { return (double)iter[e]; });
  out(4, nV + nC, [&](size_t e) { return lam[e]; });
  if (nlhs > 5) {   // auxOutput.workingSetB / workingSetC (qpOASES.m:52-62)
    const char* f[] = {"workingSetB", "workingSetC", "cpuTime"};
    plhs[5] = mxCreateStructMatrix(1, 1, 3, f);
    mxArray* wb = mxCreateDoubleMatrix(nV, k, mxREAL); mxArray* wc = mxCreateDoubleMatrix(nC, k, mxREAL);
    for (int j = 0; j < k; ++j) {
      for (int i = 0; i < nV; ++i) { const double l = lam[(size_t)j * (nV + nC) + i]; mxGetPr(wb)[(size_t)j * nV + i] = l > 0 ? -1 : (l < 0 ? 1 : 0); }
      for (int i = 0; i < nC; ++i) { const double l = lam[(size_t)j * (nV + nC) + nV + i]; mxGetPr(wc)[(size_t)j * nC + i] = l > 0 ? -1 : (l < 0 ? 1 : 0); }
    }
    mxSetField(plhs[5], 0, "workingSetB", wb); mxSetField(plhs[5], 0, "workingSetC", wc);
    mxSetField(plhs[5], 0, "cpuTime", mxCreateDoubleScalar(-1.0));
  }
}
