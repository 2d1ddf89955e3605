// mex/qpOASES_sequence.cpp -- MEX gateway for the reference's handle-based interface (optimizers/matlab/qpOASES/qpOASES_sequence.m)
// on top of the fsaempc_seq_* C ABI.  Call forms (line numbers of qpOASES_sequence.m):
//   [QP,x,fval,exitflag,iter,lambda,auxOutput] = qpOASES_sequence('i',H,g,A,lb,ub,lbA,ubA{,options{,auxInput}})     :23-24
//   [QP,x,fval,exitflag,iter,lambda,auxOutput] = qpOASES_sequence('i',H,g,lb,ub{,options{,auxInput}})               :25-26
//   [x,fval,exitflag,iter,lambda,auxOutput]    = qpOASES_sequence('h',QP,g,lb,ub,lbA,ubA{,options})                 :39-40
//   [x,fval,exitflag,iter,lambda,auxOutput]    = qpOASES_sequence('h',QP,g,lb,ub{,options})                         :41-42
//   [x,fval,exitflag,iter,lambda,auxOutput]    = qpOASES_sequence('m',QP,H,g,A,lb,ub,lbA,ubA{,options})             :51-52
//   [x,lambda,workingSetB,workingSetC]         = qpOASES_sequence('e',QP,g,lb,ub,lbA,ubA{,options})   (k columns)   :64-70
//   qpOASES_sequence('c',QP)                                                                                       :76
// g, lb, ub, lbA, ubA may carry k columns in 'h' and 'e' (k QPs sharing the handle's H and A, as in qpOASES.m:65-67).
// auxInput (initial guess / working set of qpOASES_auxInput.m) is accepted and ignored with a warning: every solve of this
// build is a cold interior-point solve followed by an active-set refinement (include/fsaempc.h).
// Not BUILT in this repo (no MATLAB here).  tests/test_abi_cpu.py type-checks it against tests/stub_mex/mex.h and RUNS it against a
// functional stand-in of the MEX API and a recording stand-in of libfsaempc (tests/stub_mex/run_gateways.cpp), so the argument
// positions of every form are tested.  Build: mex -Iinclude mex/qpOASES_sequence.cpp -Lfsae-mpc_amd/lib -lfsaempc
#include <cmath>
#include <cstdio>
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

static std::vector<double> dense(const mxArray* a) {   // qpOASES_sequence.m:35: H and A may be sparse
  const mwSize m = mxGetM(a), n = mxGetN(a);
  std::vector<double> out((size_t)m * n, 0.0);
  if (mxIsSparse(a)) {
    const mwIndex *ir = mxGetIr(a), *jc = mxGetJc(a); const double* pr = mxGetPr(a);
    for (mwSize j = 0; j < n; ++j) for (mwIndex k = jc[j]; k < jc[j + 1]; ++k) out[(size_t)j * m + ir[k]] = pr[k];
  } else if (m > 0 && n > 0) std::memcpy(out.data(), mxGetPr(a), sizeof(double) * m * n);
  return out;
}

namespace {
struct Vecs {   // the vector arguments of one call, k columns each (a single column or an empty argument is broadcast)
  int nV = 0, nC = 0, k = 1;
  std::vector<double> g, lb, ub, lbA, ubA;
};
void col_arg(const mxArray* a, int rows, int k, double fill, std::vector<double>& v, int argno) {
  v.assign((size_t)rows * k, fill);
  if (!a || mxIsEmpty(a)) return;
  if (!mxIsDouble(a) || mxIsComplex(a)) mexErrMsgTxt("ERROR (qpOASES): All data has to be provided in double precision!");
  const int ca = (int)mxGetN(a);
  if ((int)mxGetM(a) != rows || (ca != 1 && ca != k)) {
    char msg[96]; std::snprintf(msg, sizeof(msg), "ERROR (qpOASES): Input dimension mismatch for argument %d", argno);
    mexErrMsgTxt(msg);
  }
  for (int j = 0; j < k; ++j) std::memcpy(&v[(size_t)j * rows], mxGetPr(a) + (size_t)(ca == k ? j : 0) * rows, sizeof(double) * rows);
}
// g at prhs[first]; then lb, ub and (general form) lbA, ubA
void read_vecs(const mxArray* prhs[], int first, bool general, int nV, int nC, Vecs* v) {
  v->nV = nV; v->nC = nC; v->k = (int)mxGetN(prhs[first]) > 0 ? (int)mxGetN(prhs[first]) : 1;
  col_arg(prhs[first], nV, v->k, 0.0, v->g, first + 1);
  col_arg(prhs[first + 1], nV, v->k, -INFINITY, v->lb, first + 2);
  col_arg(prhs[first + 2], nV, v->k, INFINITY, v->ub, first + 3);
  col_arg(general ? prhs[first + 3] : nullptr, nC, v->k, -INFINITY, v->lbA, first + 4);
  col_arg(general ? prhs[first + 4] : nullptr, nC, v->k, INFINITY, v->ubA, first + 5);
}
struct Sol {
  std::vector<double> x, fval, lam; std::vector<int> flag, iter;
  Sol(int nV, int nC, int k) : x((size_t)nV * k), fval(k), lam((size_t)(nV + nC) * k), flag(k), iter(k) {}
};
// outputs [x,fval,exitflag,iter,lambda,auxOutput] starting at plhs[first] (qpOASES_sequence.m:22,38; auxOutput as in qpOASES.m:55-62)
void solve_outputs(int nlhs, mxArray* plhs[], int first, const Vecs& v, const Sol& s) {
  const int nV = v.nV, nC = v.nC, k = v.k;
  auto mat = [&](int i, int rows, auto&& get) {
    if (nlhs > i || i == 0) { plhs[i] = mxCreateDoubleMatrix(rows, k, mxREAL); for (size_t e = 0; e < (size_t)rows * k; ++e) mxGetPr(plhs[i])[e] = get(e); }
  };
  mat(first, nV, [&](size_t e) { return s.x[e]; });
  mat(first + 1, 1, [&](size_t e) { return s.fval[e]; });
  mat(first + 2, 1, [&](size_t e) { return (double)s.flag[e]; });
  mat(first + 3, 1, [&](size_t e) { return (double)s.iter[e]; });
  mat(first + 4, nV + nC, [&](size_t e) { return s.lam[e]; });
  if (nlhs > first + 5) {
    const char* f[] = {"workingSetB", "workingSetC", "cpuTime"};
    plhs[first + 5] = mxCreateStructMatrix(1, 1, 3, f);
    mxArray* wb = mxCreateDoubleMatrix(nV, k, mxREAL); mxArray* wc = mxCreateDoubleMatrix(nC, k, mxREAL);
    for (int j = 0; j < k; ++j) {   // -1 lower / 0 inactive / +1 upper (qpOASES.m:58-61) from the sign of the multipliers
      for (int i = 0; i < nV; ++i) { const double l = s.lam[(size_t)j * (nV + nC) + i]; mxGetPr(wb)[(size_t)j * nV + i] = l > 0 ? -1 : (l < 0 ? 1 : 0); }
      for (int i = 0; i < nC; ++i) { const double l = s.lam[(size_t)j * (nV + nC) + nV + i]; mxGetPr(wc)[(size_t)j * nC + i] = l > 0 ? -1 : (l < 0 ? 1 : 0); }
    }
    mxSetField(plhs[first + 5], 0, "workingSetB", wb); mxSetField(plhs[first + 5], 0, "workingSetC", wc);
    mxSetField(plhs[first + 5], 0, "cpuTime", mxCreateDoubleScalar(-1.0));
  }
}
const double* ptr(const std::vector<double>& v) { return v.empty() ? nullptr : v.data(); }
}  // namespace

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  if (nrhs < 2 || !mxIsChar(prhs[0])) mexErrMsgTxt("ERROR (qpOASES): Invalid number of input arguments!");
  char cmd[4] = {0}; mxGetString(prhs[0], cmd, 2);
  fsaempc_qp_opts opts;
  auto options_at = [&](int pos) {   // optional options struct at prhs[pos], optional auxInput behind it ('i' only)
    map_options(nrhs > pos ? prhs[pos] : nullptr, &opts);
    if (nrhs > pos + 1 && !mxIsEmpty(prhs[pos + 1])) mexWarnMsgTxt("WARNING (qpOASES): auxInput is ignored (every solve is a cold start)");
  };
  if (cmd[0] == 'c') { if (fsaempc_seq_cleanup((int)mxGetScalar(prhs[1])) != 0) mexErrMsgTxt(fsaempc_last_error()); return; }
  if (cmd[0] == 'i') {
    // general form: eight data arguments; bounds-only form: ('i',H,g,lb,ub{,options{,auxInput}}) -- told apart by the argument
    // count and by whether the sixth argument is an options struct
    const bool general = nrhs >= 8 && !mxIsStruct(prhs[5]);
    if (!general && (nrhs < 5 || nrhs > 7)) mexErrMsgTxt("ERROR (qpOASES): Invalid number of input arguments!");
    const int nV = (int)mxGetM(prhs[1]), nC = general ? (int)mxGetM(prhs[3]) : 0;
    if ((int)mxGetN(prhs[1]) != nV) mexErrMsgTxt("ERROR (qpOASES): Input dimension mismatch for argument 2");
    if (general && (int)mxGetN(prhs[3]) != nV) mexErrMsgTxt("ERROR (qpOASES): Input dimension mismatch for argument 4");
    options_at(general ? 8 : 5);
    Vecs v;
    if (general) { const mxArray* a[5] = {prhs[2], prhs[4], prhs[5], prhs[6], prhs[7]}; read_vecs(a, 0, true, nV, nC, &v); }
    else { const mxArray* a[3] = {prhs[2], prhs[3], prhs[4]}; read_vecs(a, 0, false, nV, nC, &v); }
    std::vector<double> H = dense(prhs[1]), A = general ? dense(prhs[3]) : std::vector<double>();
    Sol s(nV, nC, v.k); int handle = 0;
    const int rc = fsaempc_seq_init(nV, nC, H.data(), ptr(v.g), ptr(A), ptr(v.lb), ptr(v.ub), ptr(v.lbA), ptr(v.ubA), v.k, &opts, &handle,
                                    s.x.data(), s.fval.data(), s.flag.data(), s.iter.data(), s.lam.data());
    if (rc != 0) mexErrMsgTxt(fsaempc_last_error());
    plhs[0] = mxCreateDoubleScalar((double)handle);
    if (nlhs > 1) solve_outputs(nlhs, plhs, 1, v, s);
    return;
  }
  if (cmd[0] == 'h' || cmd[0] == 'e') {
    // ('h'|'e',QP,g,lb,ub,lbA,ubA{,options}) or the bounds-only ('h'|'e',QP,g,lb,ub{,options})
    if (nrhs < 5) mexErrMsgTxt("ERROR (qpOASES): Invalid number of input arguments!");
    const bool general = nrhs >= 7 && !mxIsStruct(prhs[5]);
    if (!general && nrhs > 6) mexErrMsgTxt("ERROR (qpOASES): Invalid number of input arguments!");
    const int QP = (int)mxGetScalar(prhs[1]), nV = (int)mxGetM(prhs[2]), nC = general ? (int)mxGetM(prhs[5]) : 0;
    map_options(nrhs > (general ? 7 : 5) ? prhs[general ? 7 : 5] : nullptr, &opts);
    Vecs v;
    read_vecs(prhs, 2, general, nV, nC, &v);
    if (cmd[0] == 'h') {
      Sol s(nV, nC, v.k);
      const int rc = fsaempc_seq_hotstart(QP, nV, nC, ptr(v.g), ptr(v.lb), ptr(v.ub), ptr(v.lbA), ptr(v.ubA), v.k, &opts,
                                          s.x.data(), s.fval.data(), s.flag.data(), s.iter.data(), s.lam.data());
      if (rc != 0) mexErrMsgTxt(fsaempc_last_error());
      solve_outputs(nlhs, plhs, 0, v, s);
    } else {
      std::vector<double> x((size_t)nV * v.k), lam((size_t)(nV + nC) * v.k); std::vector<int> wb(nV > 0 ? nV : 1), wc(nC > 0 ? nC : 1);
      if (fsaempc_seq_equality(QP, nV, nC, ptr(v.g), ptr(v.lb), ptr(v.ub), ptr(v.lbA), ptr(v.ubA), v.k, &opts, x.data(), lam.data(), wb.data(), wc.data()) != 0)
        mexErrMsgTxt(fsaempc_last_error());
      plhs[0] = mxCreateDoubleMatrix(nV, v.k, mxREAL); std::memcpy(mxGetPr(plhs[0]), x.data(), sizeof(double) * nV * v.k);
      if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(nV + nC, v.k, mxREAL); std::memcpy(mxGetPr(plhs[1]), lam.data(), sizeof(double) * (nV + nC) * v.k); }
      if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix(nV, 1, mxREAL); for (int i = 0; i < nV; ++i) mxGetPr(plhs[2])[i] = wb[i]; }
      if (nlhs > 3) { plhs[3] = mxCreateDoubleMatrix(nC, 1, mxREAL); for (int i = 0; i < nC; ++i) mxGetPr(plhs[3])[i] = wc[i]; }
    }
    return;
  }
  if (cmd[0] == 'm') {
    if (nrhs < 9) mexErrMsgTxt("ERROR (qpOASES): Invalid number of input arguments!");
    const int QP = (int)mxGetScalar(prhs[1]), nV = (int)mxGetM(prhs[2]), nC = (int)mxGetM(prhs[4]);
    if ((int)mxGetN(prhs[2]) != nV) mexErrMsgTxt("ERROR (qpOASES): Input dimension mismatch for argument 3");
    if ((int)mxGetN(prhs[4]) != nV) mexErrMsgTxt("ERROR (qpOASES): Input dimension mismatch for argument 5");
    map_options(nrhs > 9 ? prhs[9] : nullptr, &opts);
    Vecs v;
    { const mxArray* a[5] = {prhs[3], prhs[5], prhs[6], prhs[7], prhs[8]}; read_vecs(a, 0, true, nV, nC, &v); }
    std::vector<double> H = dense(prhs[2]), A = dense(prhs[4]);
    Sol s(nV, nC, v.k);
    const int rc = fsaempc_seq_hotstart_matrices(QP, nV, nC, H.data(), ptr(v.g), ptr(A), ptr(v.lb), ptr(v.ub), ptr(v.lbA), ptr(v.ubA), v.k, &opts,
                                                 s.x.data(), s.fval.data(), s.flag.data(), s.iter.data(), s.lam.data());
    if (rc != 0) mexErrMsgTxt(fsaempc_last_error());
    solve_outputs(nlhs, plhs, 0, v, s);
    return;
  }
  mexErrMsgTxt("ERROR (qpOASES): Invalid call of qpOASES_sequence!");
}
