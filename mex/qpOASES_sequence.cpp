// mex/qpOASES_sequence.cpp -- MEX gateway for the reference's handle-based interface
//   qpOASES_sequence('i',H,g,A,lb,ub,lbA,ubA) / ('h',QP,g,lb,ub,lbA,ubA) / ('m',QP,H,g,A,lb,ub,lbA,ubA) / ('c',QP)
// (optimizers/matlab/qpOASES/qpOASES_sequence.m:23,39,51,76) on top of the fsaempc_seq_* C ABI.
//   [x,lambda,workingSetB,workingSetC] = qpOASES_sequence('e',QP,g,lb,ub,lbA,ubA) (:64)
// Not BUILT in this repo (no MATLAB here); type-checked against tests/stub_mex/mex.h by tests/test_abi_cpu.py.
// Build: mex -Iinclude mex/qpOASES_sequence.cpp -Lfsae-mpc_amd/lib -lfsaempc
#include <cstring>
#include <vector>
#include "mex.h"
#include "fsaempc.h"

static void map_options(const mxArray* o, fsaempc_qp_opts* q) {   // as in mex/qpOASES.cpp
  fsaempc_qp_default_opts(q);
  if (!o || !mxIsStruct(o)) return;
  const mxArray* f = mxGetField(o, 0, "maxIter");
  if (f && !mxIsEmpty(f) && mxGetScalar(f) > 0) q->max_iter = (int)mxGetScalar(f);
  f = mxGetField(o, 0, "terminationTolerance");
  if (f && !mxIsEmpty(f) && mxGetScalar(f) > 0) { q->tol = mxGetScalar(f) > 1e-12 ? mxGetScalar(f) : 1e-12; if (q->tol_loose < q->tol) q->tol_loose = q->tol; }
}

static std::vector<double> dense(const mxArray* a) {
  const mwSize m = mxGetM(a), n = mxGetN(a);
  std::vector<double> out((size_t)m * n, 0.0);
  if (mxIsSparse(a)) {
    const mwIndex *ir = mxGetIr(a), *jc = mxGetJc(a); const double* pr = mxGetPr(a);
    for (mwSize j = 0; j < n; ++j) for (mwIndex k = jc[j]; k < jc[j + 1]; ++k) out[(size_t)j * m + ir[k]] = pr[k];
  } else std::memcpy(out.data(), mxGetPr(a), sizeof(double) * m * n);
  return out;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  if (nrhs < 2 || !mxIsChar(prhs[0])) mexErrMsgTxt("ERROR (qpOASES): Invalid number of input arguments!");
  char cmd[4] = {0}; mxGetString(prhs[0], cmd, 2);
  auto solve_outputs = [&](int first, int nV, int nC, int rc, const std::vector<double>& x, double fval, int flag, int iter, const std::vector<double>& lam) {
    if (rc != 0) mexErrMsgTxt(fsaempc_last_error());
    if (nlhs > first) { plhs[first] = mxCreateDoubleMatrix(nV, 1, mxREAL); std::memcpy(mxGetPr(plhs[first]), x.data(), sizeof(double) * nV); }
    if (nlhs > first + 1) plhs[first + 1] = mxCreateDoubleScalar(fval);
    if (nlhs > first + 2) plhs[first + 2] = mxCreateDoubleScalar((double)flag);
    if (nlhs > first + 3) plhs[first + 3] = mxCreateDoubleScalar((double)iter);
    if (nlhs > first + 4) { plhs[first + 4] = mxCreateDoubleMatrix(nV + nC, 1, mxREAL); std::memcpy(mxGetPr(plhs[first + 4]), lam.data(), sizeof(double) * (nV + nC)); }
  };
  fsaempc_qp_opts opts;
  const int nopt = cmd[0] == 'i' ? 8 : (cmd[0] == 'm' ? 9 : 7);   // position of the optional options struct
  map_options(nrhs > nopt ? prhs[nopt] : nullptr, &opts);
  if (cmd[0] == 'c') { if (fsaempc_seq_cleanup((int)mxGetScalar(prhs[1])) != 0) mexErrMsgTxt(fsaempc_last_error()); return; }
  if (cmd[0] == 'i' && nrhs >= 8) {
    const int nV = (int)mxGetM(prhs[1]), nC = (int)mxGetM(prhs[3]);
    std::vector<double> H = dense(prhs[1]), A = dense(prhs[3]), x(nV), lam(nV + nC); double fval = 0; int flag = 0, iter = 0, handle = 0;
    const int rc = fsaempc_seq_init(nV, nC, H.data(), mxGetPr(prhs[2]), A.data(), mxGetPr(prhs[4]), mxGetPr(prhs[5]), mxGetPr(prhs[6]), mxGetPr(prhs[7]), 1,
                                    &opts, &handle, x.data(), &fval, &flag, &iter, lam.data());
    if (rc == 0) plhs[0] = mxCreateDoubleScalar((double)handle);
    solve_outputs(1, nV, nC, rc, x, fval, flag, iter, lam);
    return;
  }
  if (cmd[0] == 'h' && nrhs >= 7) {
    const int QP = (int)mxGetScalar(prhs[1]), nV = (int)mxGetM(prhs[2]), nC = (int)mxGetM(prhs[5]);
    std::vector<double> x(nV), lam(nV + nC); double fval = 0; int flag = 0, iter = 0;
    const int rc = fsaempc_seq_hotstart(QP, nV, nC, mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetPr(prhs[5]), mxGetPr(prhs[6]), 1, &opts,
                                        x.data(), &fval, &flag, &iter, lam.data());
    solve_outputs(0, nV, nC, rc, x, fval, flag, iter, lam);
    return;
  }
  if (cmd[0] == 'm' && nrhs >= 9) {
    const int QP = (int)mxGetScalar(prhs[1]), nV = (int)mxGetM(prhs[2]), nC = (int)mxGetM(prhs[4]);
    std::vector<double> H = dense(prhs[2]), A = dense(prhs[4]), x(nV), lam(nV + nC); double fval = 0; int flag = 0, iter = 0;
    const int rc = fsaempc_seq_hotstart_matrices(QP, nV, nC, H.data(), mxGetPr(prhs[3]), A.data(), mxGetPr(prhs[5]), mxGetPr(prhs[6]), mxGetPr(prhs[7]),
                                                 mxGetPr(prhs[8]), 1, &opts, x.data(), &fval, &flag, &iter, lam.data());
    solve_outputs(0, nV, nC, rc, x, fval, flag, iter, lam);
    return;
  }
  if (cmd[0] == 'e' && nrhs >= 7) {
    const int QP = (int)mxGetScalar(prhs[1]), nV = (int)mxGetM(prhs[2]), nC = (int)mxGetM(prhs[5]);
    std::vector<double> x(nV), lam(nV + nC); std::vector<int> wb(nV), wc(nC > 0 ? nC : 1);
    if (fsaempc_seq_equality(QP, nV, nC, mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetPr(prhs[5]), mxGetPr(prhs[6]), 1, &opts,
                             x.data(), lam.data(), wb.data(), wc.data()) != 0) mexErrMsgTxt(fsaempc_last_error());
    plhs[0] = mxCreateDoubleMatrix(nV, 1, mxREAL); std::memcpy(mxGetPr(plhs[0]), x.data(), sizeof(double) * nV);
    if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(nV + nC, 1, mxREAL); std::memcpy(mxGetPr(plhs[1]), lam.data(), sizeof(double) * (nV + nC)); }
    if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix(nV, 1, mxREAL); for (int i = 0; i < nV; ++i) mxGetPr(plhs[2])[i] = wb[i]; }
    if (nlhs > 3) { plhs[3] = mxCreateDoubleMatrix(nC, 1, mxREAL); for (int i = 0; i < nC; ++i) mxGetPr(plhs[3])[i] = wc[i]; }
    return;
  }
  mexErrMsgTxt("ERROR (qpOASES): Invalid call of qpOASES_sequence!");
}
