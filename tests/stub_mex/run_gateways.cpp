// Runs mex/qpOASES.cpp and mex/qpOASES_sequence.cpp on the CPU: functional stand-in of the MEX API (mex_fake.cpp) + a RECORDING
// stand-in of libfsaempc defined here.  Every vector argument carries its own sentinel value (g = 2, lb = 3, ub = 4, lbA = 5,
// ubA = 6, H = 7 on the diagonal, A = 8), so an argument that lands in the wrong position of the C ABI is seen at once; the stand-in
// answers x = 10 + column, fval = 20 + column, exitflag = 0, iter = 30 + column, lambda = +1 on variable 0 / -1 on row 0.
// Test infrastructure (tests/test_abi_cpu.py::test_mex_gateways_run); exits 0 and prints "gateways ok" when every form checks out.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "mex_fake.h"
#include "fsaempc.h"

// ---- recording stand-in of the C ABI ----
struct Call { std::string fn; int nV = 0, nC = 0, k = 0, handle = 0, max_iter = 0; bool hasA = false, hasLbA = false; };
static std::vector<Call> g_calls;
static std::string g_err;
static void require(bool ok, const char* what) { if (!ok) throw std::logic_error(std::string("check failed: ") + what); }
static void check_vecs(int nV, int nC, int k, const double* g, const double* lb, const double* ub, const double* lbA, const double* ubA) {
  for (int j = 0; j < k; ++j) {
    require(g && g[(size_t)j * nV] == 2.0 + j, "g sentinel"); require(lb && lb[(size_t)j * nV] == 3.0, "lb sentinel"); require(ub && ub[(size_t)j * nV] == 4.0, "ub sentinel");
    if (nC) { require(lbA && lbA[(size_t)j * nC] == 5.0, "lbA sentinel"); require(ubA && ubA[(size_t)j * nC] == 6.0, "ubA sentinel"); }
  }
}
static void answer(int nV, int nC, int k, double* x, double* fval, int* flag, int* iter, double* lam) {
  for (int j = 0; j < k; ++j) {
    for (int i = 0; i < nV; ++i) x[(size_t)j * nV + i] = 10.0 + j;
    if (fval) fval[j] = 20.0 + j;
    if (flag) flag[j] = 0;
    if (iter) iter[j] = 30 + j;
    if (lam) { for (int i = 0; i < nV + nC; ++i) lam[(size_t)j * (nV + nC) + i] = 0.0; lam[(size_t)j * (nV + nC)] = 1.0; if (nC) lam[(size_t)j * (nV + nC) + nV] = -1.0; }
  }
}
extern "C" {
const char* fsaempc_last_error(void) { return g_err.c_str(); }
void fsaempc_qp_default_opts(fsaempc_qp_opts* o) { o->tol = 1e-8; o->tol_loose = 1e-6; o->tol_x = 1e-7; o->inf_bound = 1e9; o->max_iter = 100; o->polish = 1; }
int fsaempc_qp_solve_batch(const fsaempc_qp_desc* d, const double* H, const double* g, const double* A, const double* lb, const double* ub,
                           const double* lbA, const double* ubA, const fsaempc_qp_opts* opts, double* x, double* fval, int* flag, int* iter, double* lam) {
  Call c; c.fn = "solve_batch"; c.nV = d->nV; c.nC = d->nC; c.k = d->batch; c.max_iter = opts->max_iter; c.hasA = A != nullptr; c.hasLbA = lbA != nullptr;
  require(d->shared_HA == 1, "k columns share H and A"); require(H[0] == 7.0, "H sentinel"); if (d->nC) require(A && A[0] == 8.0, "A sentinel");
  check_vecs(d->nV, d->nC, d->batch, g, lb, ub, lbA, ubA);
  answer(d->nV, d->nC, d->batch, x, fval, flag, iter, lam);
  g_calls.push_back(c); return 0;
}
int fsaempc_seq_init(int nV, int nC, const double* H, const double* g, const double* A, const double* lb, const double* ub, const double* lbA, const double* ubA,
                     int k, const fsaempc_qp_opts* opts, int* handle, double* x, double* fval, int* flag, int* iter, double* lam) {
  Call c; c.fn = "init"; c.nV = nV; c.nC = nC; c.k = k; c.max_iter = opts->max_iter; c.hasA = A != nullptr; c.hasLbA = lbA != nullptr;
  require(H[0] == 7.0, "H sentinel"); if (nC) require(A && A[0] == 8.0, "A sentinel");
  check_vecs(nV, nC, k, g, lb, ub, lbA, ubA);
  *handle = 41; answer(nV, nC, k, x, fval, flag, iter, lam);
  g_calls.push_back(c); return 0;
}
int fsaempc_seq_hotstart(int handle, int nV, int nC, const double* g, const double* lb, const double* ub, const double* lbA, const double* ubA, int k,
                         const fsaempc_qp_opts* opts, double* x, double* fval, int* flag, int* iter, double* lam) {
  Call c; c.fn = "hotstart"; c.nV = nV; c.nC = nC; c.k = k; c.handle = handle; c.max_iter = opts->max_iter; c.hasLbA = lbA != nullptr;
  if (handle != 41) { g_err = "ERROR (qpOASES): Invalid handle to QP instance!"; return FSAEMPC_ERR_ARG; }
  check_vecs(nV, nC, k, g, lb, ub, lbA, ubA);
  answer(nV, nC, k, x, fval, flag, iter, lam);
  g_calls.push_back(c); return 0;
}
int fsaempc_seq_hotstart_matrices(int handle, int nV, int nC, const double* H, const double* g, const double* A, const double* lb, const double* ub,
                                  const double* lbA, const double* ubA, int k, const fsaempc_qp_opts* opts, double* x, double* fval, int* flag, int* iter, double* lam) {
  Call c; c.fn = "hotstart_matrices"; c.nV = nV; c.nC = nC; c.k = k; c.handle = handle; c.max_iter = opts->max_iter; c.hasA = A != nullptr;
  require(H[0] == 7.0 && A && A[0] == 8.0, "H / A sentinels");
  check_vecs(nV, nC, k, g, lb, ub, lbA, ubA);
  answer(nV, nC, k, x, fval, flag, iter, lam);
  g_calls.push_back(c); return 0;
}
int fsaempc_seq_equality(int handle, int nV, int nC, const double* g, const double* lb, const double* ub, const double* lbA, const double* ubA, int k,
                         const fsaempc_qp_opts*, double* x, double* lam, int* wb, int* wc) {
  Call c; c.fn = "equality"; c.nV = nV; c.nC = nC; c.k = k; c.handle = handle;
  check_vecs(nV, nC, k, g, lb, ub, lbA, ubA);
  answer(nV, nC, k, x, nullptr, nullptr, nullptr, lam);
  for (int i = 0; i < nV; ++i) wb[i] = i == 0 ? -1 : 0;
  for (int i = 0; i < nC; ++i) wc[i] = i == 0 ? 1 : 0;
  g_calls.push_back(c); return 0;
}
int fsaempc_seq_cleanup(int handle) { Call c; c.fn = "cleanup"; c.handle = handle; g_calls.push_back(c); if (handle != 41) { g_err = "ERROR (qpOASES): Invalid handle to QP instance!"; return FSAEMPC_ERR_ARG; } return 0; }
}

// ---- the two gateways, each with its own mexFunction ----
#define mexFunction mexFunction_qpOASES
#define map_options map_options_qpOASES
#define dense dense_qpOASES
#include "../../mex/qpOASES.cpp"
#undef mexFunction
#undef map_options
#undef dense
#define mexFunction mexFunction_sequence
#define map_options map_options_sequence
#define dense dense_sequence
#include "../../mex/qpOASES_sequence.cpp"
#undef mexFunction
#undef map_options
#undef dense

static const int nV = 3, nC = 2;
static mxArray* Hm() { std::vector<double> h(nV * nV, 0.0); for (int i = 0; i < nV; ++i) h[i * nV + i] = 7.0; return fake_matrix(nV, nV, h); }
static mxArray* Am() { return fake_fill(nC, nV, 8.0); }
static mxArray* gcols(int k) { std::vector<double> g((size_t)nV * k); for (int j = 0; j < k; ++j) for (int i = 0; i < nV; ++i) g[(size_t)j * nV + i] = 2.0 + j; return fake_matrix(nV, k, g); }
static mxArray* opts_struct(double max_iter) { mxArray* o = fake_struct(); const char* f[] = {"maxIter"}; (void)f; mxSetField(o, 0, "maxIter", fake_fill(1, 1, max_iter)); return o; }
static double at(const mxArray* a, size_t i) { return mxGetPr(a)[i]; }
static bool raises(void (*fn)(int, mxArray**, int, const mxArray**), int nlhs, std::vector<const mxArray*> in, const char* needle) {
  mxArray* out[8] = {nullptr};
  try { fn(nlhs, out, (int)in.size(), in.data()); } catch (const std::runtime_error& e) { return std::strstr(e.what(), needle) != nullptr; }
  return false;
}

int main() {
  try {
    mxArray* out[8];
    auto call = [&](void (*fn)(int, mxArray**, int, const mxArray**), int nlhs, std::vector<const mxArray*> in) { for (auto& o : out) o = nullptr; g_calls.clear(); fn(nlhs, out, (int)in.size(), in.data()); };
    mxArray *lb = fake_fill(nV, 1, 3.0), *ub = fake_fill(nV, 1, 4.0), *lbA = fake_fill(nC, 1, 5.0), *ubA = fake_fill(nC, 1, 6.0);
    // --- qpOASES: general form, 2 columns, options, auxInput (warning), six outputs
    g_mex_warnings.clear();
    call(mexFunction_qpOASES, 6, {Hm(), gcols(2), Am(), lb, ub, lbA, ubA, opts_struct(55), fake_fill(1, 1, 1.0)});
    require(g_calls.size() == 1 && g_calls[0].fn == "solve_batch" && g_calls[0].nV == nV && g_calls[0].nC == nC && g_calls[0].k == 2 && g_calls[0].max_iter == 55, "qpOASES general form");
    require(mxGetM(out[0]) == (size_t)nV && mxGetN(out[0]) == 2 && at(out[0], nV) == 11.0 && at(out[1], 1) == 21.0 && at(out[2], 0) == 0.0 && at(out[3], 1) == 31.0, "qpOASES outputs");
    require(mxGetM(out[4]) == (size_t)(nV + nC) && at(fake_field(out[5], "workingSetB"), 0) == -1.0 && at(fake_field(out[5], "workingSetC"), 0) == 1.0, "lambda / auxOutput");
    require(g_mex_warnings.size() == 1, "auxInput warning");
    // --- qpOASES: bounds-only form with options
    call(mexFunction_qpOASES, 3, {Hm(), gcols(1), lb, ub, opts_struct(9)});
    require(g_calls[0].nC == 0 && !g_calls[0].hasA && !g_calls[0].hasLbA && g_calls[0].max_iter == 9 && at(out[0], 0) == 10.0, "qpOASES bounds-only form");
    require(raises(mexFunction_qpOASES, 1, {Hm(), gcols(1), fake_fill(nC, nV + 1, 8.0), lb, ub, lbA, ubA}, "dimension mismatch"), "A with a wrong column count is rejected");
    // --- sequence 'i': general form with options + auxInput, seven outputs
    g_mex_warnings.clear();
    call(mexFunction_sequence, 7, {fake_string("i"), Hm(), gcols(1), Am(), lb, ub, lbA, ubA, opts_struct(77), fake_fill(1, 1, 1.0)});
    require(g_calls[0].fn == "init" && g_calls[0].nC == nC && g_calls[0].hasA && g_calls[0].max_iter == 77 && at(out[0], 0) == 41.0 && at(out[1], 0) == 10.0 && at(out[2], 0) == 20.0, "'i' general form");
    require(at(out[4], 0) == 30.0 && mxGetM(out[5]) == (size_t)(nV + nC) && at(fake_field(out[6], "workingSetC"), 0) == 1.0 && g_mex_warnings.size() == 1, "'i' outputs / auxOutput / auxInput warning");
    // --- sequence 'i': bounds-only form ('i',H,g,lb,ub{,options})
    call(mexFunction_sequence, 2, {fake_string("i"), Hm(), gcols(1), lb, ub, opts_struct(66)});
    require(g_calls[0].fn == "init" && g_calls[0].nC == 0 && !g_calls[0].hasA && g_calls[0].max_iter == 66 && at(out[0], 0) == 41.0, "'i' bounds-only form");
    call(mexFunction_sequence, 2, {fake_string("i"), Hm(), gcols(1), lb, ub});
    require(g_calls[0].nC == 0 && g_calls[0].max_iter == 100, "'i' bounds-only form without options");
    // --- 'h': general (k = 2 columns, six outputs incl. auxOutput) and bounds-only
    call(mexFunction_sequence, 6, {fake_string("h"), fake_fill(1, 1, 41.0), gcols(2), fake_fill(nV, 2, 3.0), fake_fill(nV, 2, 4.0), fake_fill(nC, 2, 5.0), fake_fill(nC, 2, 6.0)});
    require(g_calls[0].fn == "hotstart" && g_calls[0].handle == 41 && g_calls[0].k == 2 && g_calls[0].nC == nC && mxGetN(out[0]) == 2 && at(out[0], nV) == 11.0 && at(out[3], 1) == 31.0, "'h' general form, two columns");
    require(at(fake_field(out[5], "workingSetB"), 0) == -1.0, "'h' auxOutput");
    call(mexFunction_sequence, 1, {fake_string("h"), fake_fill(1, 1, 41.0), gcols(1), lb, ub, opts_struct(12)});
    require(g_calls[0].fn == "hotstart" && g_calls[0].nC == 0 && !g_calls[0].hasLbA && g_calls[0].max_iter == 12, "'h' bounds-only form with options");
    require(raises(mexFunction_sequence, 1, {fake_string("h"), fake_fill(1, 1, 40.0), gcols(1), lb, ub, lbA, ubA}, "Invalid handle"), "'h' on an unknown handle");
    // --- 'm'
    call(mexFunction_sequence, 5, {fake_string("m"), fake_fill(1, 1, 41.0), Hm(), gcols(1), Am(), lb, ub, lbA, ubA, opts_struct(31)});
    require(g_calls[0].fn == "hotstart_matrices" && g_calls[0].handle == 41 && g_calls[0].hasA && g_calls[0].max_iter == 31 && at(out[1], 0) == 20.0, "'m'");
    // --- 'e': two columns, four outputs
    call(mexFunction_sequence, 4, {fake_string("e"), fake_fill(1, 1, 41.0), gcols(2), fake_fill(nV, 2, 3.0), fake_fill(nV, 2, 4.0), fake_fill(nC, 2, 5.0), fake_fill(nC, 2, 6.0)});
    require(g_calls[0].fn == "equality" && g_calls[0].k == 2 && mxGetN(out[0]) == 2 && mxGetN(out[1]) == 2 && at(out[2], 0) == -1.0 && at(out[3], 0) == 1.0, "'e' with two columns");
    // --- 'c' and the call main.m:193 makes (QP == 0)
    call(mexFunction_sequence, 0, {fake_string("c"), fake_fill(1, 1, 41.0)});
    require(g_calls[0].fn == "cleanup" && g_calls[0].handle == 41, "'c'");
    require(raises(mexFunction_sequence, 0, {fake_string("c"), fake_fill(1, 1, 0.0)}, "Invalid handle"), "'c' with QP == 0 (main.m:193) raises like the original");
    require(raises(mexFunction_sequence, 1, {fake_string("x"), fake_fill(1, 1, 41.0)}, "Invalid call"), "unknown command");
  } catch (const std::exception& e) { std::printf("FAILED: %s\n", e.what()); return 1; }
  std::printf("gateways ok\n");
  return 0;
}
