// capi.hip -- the C ABI of libfsaempc.so (include/fsaempc.h).  Host-side glue only: argument checks,
// workspace carving, kernel launches.  There is no CPU compute path: without a gfx950 device every entry
// point fails with FSAEMPC_ERR_NODEVICE / FSAEMPC_ERR_HIP.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <dlfcn.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "fsaempc.h"
#include "qp_solver.h"
#include "ltv_build.h"
#include "reference.h"
#include "plant.h"

namespace {
thread_local char g_err[512] = "";
// Diagnostics (fsaempc_debug_set_dump, fsaempc_qp_set_timing): process-wide switches meant for ONE measuring thread (bench.py,
// tools/).  Atomics keep a solve on another thread well-defined while they are flipped; the event triple itself is not
// per-thread, so timing figures are only meaningful when a single thread launches solves.
std::atomic<double*> g_dump{nullptr}; std::atomic<int> g_dump_stage{0};
std::atomic<bool> g_timing{false}; hipEvent_t g_ev[3] = {nullptr, nullptr, nullptr};
hipEvent_t g_evf[2] = {nullptr, nullptr};   // fused step: before the construction kernel, after the post-solve kernel

// roctx ranges around the launches of every phase (rocprofv3 --marker-trace shows them next to the kernel trace).  The roctx library is
// looked up at run time: without it (or with FSAEMPC_ROCTX=0) the ranges are no-ops and the library has no dependency on it.
struct Roctx {
  int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
  Roctx() {
    const char* off = getenv("FSAEMPC_ROCTX");
    if (off && off[0] == '0') return;
    void* h = nullptr;   // rocprofv3 listens to the rocprofiler-sdk flavour; libroctx64 is roctracer's (rocprof v1 / v2)
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"})
      if ((h = dlopen(name, RTLD_LAZY | RTLD_GLOBAL))) break;
    if (!h) return;
    push = (int (*)(const char*))dlsym(h, "roctxRangePushA"); pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) { push = nullptr; pop = nullptr; }
  }
};
struct Range {   // RAII: one named range on the calling thread
  static Roctx& api() { static Roctx r; return r; }
  bool on;
  explicit Range(const char* name) : on(api().push != nullptr) { if (on) api().push(name); }
  ~Range() { if (on) api().pop(); }
};

int fail(int code, const char* fmt, const char* a = "") { snprintf(g_err, sizeof(g_err), fmt, a); return code; }
int hipfail(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorNoBinaryForGpu || e == hipErrorInsufficientDriver)
             ? FSAEMPC_ERR_NODEVICE : FSAEMPC_ERR_HIP;
}
size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }
}  // namespace

extern "C" {

const char* fsaempc_last_error(void) { return g_err; }

void fsaempc_qp_default_opts(fsaempc_qp_opts* o) {
  if (!o) return;
  o->tol = 1e-8; o->tol_loose = 1e-6; o->tol_x = 1e-7; o->inf_bound = 1e9; o->max_iter = 100; o->polish = 1;
}

long long fsaempc_qp_workspace_bytes(const fsaempc_qp_desc* desc) {
  if (!desc || desc->nV <= 0 || desc->nC < 0 || desc->batch < 0) return FSAEMPC_ERR_ARG;
  if (desc->nV > FSAEMPC_MAX_NV) return FSAEMPC_ERR_DIM;
  QpDims d; qp_make_dims(desc->nV, desc->nC, &d);
  return (long long)(d.ws_per_qp * sizeof(double) * (size_t)(desc->batch > 0 ? desc->batch : 1) + qp_order_bytes(desc->batch));
}

int fsaempc_qp_solve_batch_device(const fsaempc_qp_desc* desc, const double* H, const double* g, const double* A,
                                  const double* lb, const double* ub, const double* lbA, const double* ubA,
                                  const fsaempc_qp_opts* opts, double* x, double* fval, int* exitflag, int* iter,
                                  double* lambda, void* workspace, long long workspace_bytes, void* stream) {
  return fsaempc_qp_solve_batch_device_aux(desc, H, g, A, lb, ub, lbA, ubA, opts, x, fval, exitflag, iter, lambda, nullptr, workspace, workspace_bytes, stream);
}

int fsaempc_qp_solve_batch_device_aux(const fsaempc_qp_desc* desc, const double* H, const double* g, const double* A,
                                      const double* lb, const double* ub, const double* lbA, const double* ubA,
                                      const fsaempc_qp_opts* opts, double* x, double* fval, int* exitflag, int* iter,
                                      double* lambda, const fsaempc_qp_aux* aux, void* workspace, long long workspace_bytes, void* stream) {
  if (!desc || !H || !g || !lb || !ub || !x || !fval || !exitflag || !iter || !workspace) return fail(FSAEMPC_ERR_ARG, "null argument");
  if (desc->nV <= 0 || desc->nC < 0 || desc->batch < 0) return fail(FSAEMPC_ERR_ARG, "bad dimensions");
  if (desc->nC > 0 && (!A || !lbA || !ubA)) return fail(FSAEMPC_ERR_ARG, "nC > 0 needs A, lbA, ubA");
  if (desc->nV > FSAEMPC_MAX_NV) return fail(FSAEMPC_ERR_DIM, "nV exceeds FSAEMPC_MAX_NV");
  if (desc->batch == 0) return 0;
  fsaempc_qp_opts o; if (opts) o = *opts; else fsaempc_qp_default_opts(&o);
  QpParams P; memset(&P, 0, sizeof(P));
  qp_make_dims(desc->nV, desc->nC, &P.d);
  const size_t ws_qps = P.d.ws_per_qp * sizeof(double) * (size_t)desc->batch;
  if ((long long)(ws_qps + qp_order_bytes(desc->batch)) > workspace_bytes) return fail(FSAEMPC_ERR_WORKSPACE, "workspace too small");
  const char* no_order = getenv("FSAEMPC_QP_ORDER");   // A/B runs only: FSAEMPC_QP_ORDER=0 solves in index order (read per call)
  if (qp_order_bytes(desc->batch) && !(no_order && no_order[0] == '0')) { P.score = (int*)((char*)workspace + ws_qps); P.order = P.score + desc->batch; }
  const bool wg = !qp_runs_wavefront_kernel(P.d);
  if ((wg ? P.d.lds_wg : P.d.lds_solve) > 160 * 1024 || P.d.lds_prep > 160 * 1024) return fail(FSAEMPC_ERR_DIM, "problem exceeds the 160 KiB LDS budget of the kernels");
  P.H = H; P.g = g; P.A = A; P.lb = lb; P.ub = ub; P.lbA = lbA; P.ubA = ubA;
  P.ws = (double*)workspace; P.x = x; P.fval = fval; P.lambda = lambda; P.exitflag = exitflag; P.iter = iter;
  P.tol = o.tol; P.tol_loose = o.tol_loose; P.tol_x = o.tol_x; P.inf_bound = o.inf_bound; P.max_iter = o.max_iter; P.polish = o.polish; P.polished = aux ? aux->polished : nullptr; P.kkt = aux ? aux->kkt : nullptr; P.x_init = aux ? aux->x_init : nullptr; P.score_in = aux ? aux->difficulty : nullptr;
  P.shared_HA = desc->shared_HA;
  { const int st = g_dump_stage.load(); P.dump = g_dump.load(); P.dump_stage = st & 0xff; P.dump_iter = st >> 8; }
  const bool timing = g_timing.load();
  hipError_t e;
  if (timing) { e = hipEventRecord(g_ev[0], (hipStream_t)stream); if (e != hipSuccess) return hipfail(e, "hipEventRecord"); }
  { Range r("fsaempc.qp.prep+solve");   // (prep, order and solve kernels are enqueued by one call; the kernel trace separates them)
    e = qp_launch(P, desc->batch, (hipStream_t)stream, timing ? g_ev[1] : nullptr); }
  if (e != hipSuccess) return hipfail(e, "qp_launch");
  if (timing) { e = hipEventRecord(g_ev[2], (hipStream_t)stream); if (e != hipSuccess) return hipfail(e, "hipEventRecord"); }
  return 0;
}

int fsaempc_qp_solve_batch(const fsaempc_qp_desc* desc, const double* H, const double* g, const double* A,
                           const double* lb, const double* ub, const double* lbA, const double* ubA,
                           const fsaempc_qp_opts* opts, double* x, double* fval, int* exitflag, int* iter, double* lambda) {
  if (!desc || !H || !g || !lb || !ub || !x) return fail(FSAEMPC_ERR_ARG, "null argument");
  if (desc->nV <= 0 || desc->nC < 0 || desc->batch < 0) return fail(FSAEMPC_ERR_ARG, "bad dimensions");
  if (desc->nC > 0 && (!A || !lbA || !ubA)) return fail(FSAEMPC_ERR_ARG, "nC > 0 needs A, lbA, ubA");
  if (desc->nV > FSAEMPC_MAX_NV) return fail(FSAEMPC_ERR_DIM, "nV exceeds FSAEMPC_MAX_NV");
  const size_t n = desc->nV, m = desc->nC, B = desc->batch, BH = desc->shared_HA ? 1 : B;
  if (B == 0) return 0;
  // the original gateway rejects NaN anywhere and Inf in H,g,A ("Argument %d contains 'NaN' !")
  for (size_t i = 0; i < BH * n * n; ++i) if (!isfinite(H[i])) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): Argument 1 contains 'NaN' or 'Inf' !");
  for (size_t i = 0; i < B * n; ++i) if (!isfinite(g[i])) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): Argument 2 contains 'NaN' or 'Inf' !");
  for (size_t i = 0; i < BH * m * n; ++i) if (!isfinite(A[i])) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): Argument 3 contains 'NaN' or 'Inf' !");
  for (size_t i = 0; i < B * n; ++i) if (isnan(lb[i]) || isnan(ub[i])) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): bounds contain 'NaN' !");
  for (size_t i = 0; i < B * m; ++i) if (isnan(lbA[i]) || isnan(ubA[i])) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): constraint bounds contain 'NaN' !");
  long long wsb = fsaempc_qp_workspace_bytes(desc);
  if (wsb < 0) return fail((int)wsb, "workspace size");
  const size_t szH = BH * n * n, szg = B * n, szA = BH * m * n, szm = B * m, szl = B * (n + m);
  const size_t nd = szH + szg + szA + 2 * szg + 2 * szm + szg /*x*/ + B /*fval*/ + szl /*lambda*/;
  char* dev = nullptr;
  hipError_t e = hipMalloc((void**)&dev, nd * sizeof(double) + 2 * B * sizeof(int) + (size_t)wsb + 4096);
  if (e != hipSuccess) return hipfail(e, "hipMalloc");
  double* p = (double*)dev;
  double* dH = p; p += szH; double* dg = p; p += szg; double* dA = p; p += szA;
  double* dlb = p; p += szg; double* dub = p; p += szg; double* dlbA = p; p += szm; double* dubA = p; p += szm;
  double* dx = p; p += szg; double* dfv = p; p += B; double* dlam = p; p += szl;
  int* dfl = (int*)p; int* dit = dfl + B;
  void* dws = (void*)(((uintptr_t)(dit + B) + 255) & ~(uintptr_t)255);
  int rc = 0;
#define CP(dst, src, cnt) do { if (rc == 0 && (cnt) > 0) { e = hipMemcpy(dst, src, (cnt) * sizeof(double), hipMemcpyHostToDevice); if (e != hipSuccess) rc = hipfail(e, "hipMemcpy H2D"); } } while (0)
  CP(dH, H, szH); CP(dg, g, szg); CP(dA, A, szA); CP(dlb, lb, szg); CP(dub, ub, szg); CP(dlbA, lbA, szm); CP(dubA, ubA, szm);
#undef CP
  if (rc == 0) rc = fsaempc_qp_solve_batch_device(desc, dH, dg, m ? dA : nullptr, dlb, dub, m ? dlbA : nullptr, m ? dubA : nullptr, opts,
                                                  dx, dfv, dfl, dit, dlam, dws, wsb, nullptr);
  if (rc == 0) { e = hipDeviceSynchronize(); if (e != hipSuccess) rc = hipfail(e, "solve"); }
#define BK(dst, src, bytes) do { if (rc == 0 && (dst)) { e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = hipfail(e, "hipMemcpy D2H"); } } while (0)
  BK(x, dx, szg * sizeof(double)); BK(fval, dfv, B * sizeof(double)); BK(exitflag, dfl, B * sizeof(int)); BK(iter, dit, B * sizeof(int));
  BK(lambda, dlam, szl * sizeof(double));
#undef BK
  (void)hipFree(dev);
  return rc;
}

// ---- qpOASES_sequence handles (qpOASES_sequence.m:23-78) ----
// A handle owns device copies of H and A (uploaded by 'i' / 'm' only), the solver workspace and the per-call vectors (grown to
// the largest k seen), so a hot start moves (3 nV + 2 nC) k doubles to the device and the results back -- no allocation, no
// matrix upload.  Every solve is a cold interior-point solve; the handle also remembers the working set of its last solve
// (first column) for the 'e' call.
namespace {
struct SeqQP {
  int nV = 0, nC = 0, kcap = 0;
  bool used = false;
  double *dH = nullptr, *dA = nullptr, *dvec = nullptr; void* dws = nullptr; long long wsb = 0;
  std::vector<signed char> wsB, wsC;   // -1 lower / 0 inactive / +1 upper, qpOASES.m:52-62 encoding
  std::vector<double> hA;              // host copy of A (the working set of a solve is derived on the host: it needs A x)
  void release() {
    (void)hipFree(dH); (void)hipFree(dA); (void)hipFree(dvec); (void)hipFree(dws);
    dH = dA = dvec = nullptr; dws = nullptr; kcap = 0; wsb = 0; used = false; wsB.clear(); wsC.clear(); hA.clear();
  }
};
std::mutex g_seq_mu;
std::vector<SeqQP> g_seq;   // handle = index + 1 (the MEX gateway also hands out small integers)
SeqQP* seq_get(int handle) { return (handle >= 1 && handle <= (int)g_seq.size() && g_seq[handle - 1].used) ? &g_seq[handle - 1] : nullptr; }
int seq_upload_matrices(SeqQP* q, const double* H, const double* A) {
  const size_t n = q->nV, m = q->nC;
  for (size_t i = 0; i < n * n; ++i) if (!isfinite(H[i])) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): Argument 1 contains 'NaN' or 'Inf' !");
  for (size_t i = 0; i < m * n; ++i) if (!isfinite(A[i])) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): Argument 3 contains 'NaN' or 'Inf' !");
  hipError_t e;
  if (!q->dH) { e = hipMalloc((void**)&q->dH, n * n * sizeof(double)); if (e != hipSuccess) return hipfail(e, "hipMalloc"); }
  if (m && !q->dA) { e = hipMalloc((void**)&q->dA, m * n * sizeof(double)); if (e != hipSuccess) return hipfail(e, "hipMalloc"); }
  e = hipMemcpy(q->dH, H, n * n * sizeof(double), hipMemcpyHostToDevice); if (e != hipSuccess) return hipfail(e, "hipMemcpy H2D");
  if (m) { e = hipMemcpy(q->dA, A, m * n * sizeof(double), hipMemcpyHostToDevice); if (e != hipSuccess) return hipfail(e, "hipMemcpy H2D"); }
  q->hA.assign(A ? A : nullptr, A ? A + m * n : nullptr);
  return 0;
}
int seq_solve(SeqQP* q, const double* g, const double* lb, const double* ub, const double* lbA, const double* ubA, int k,
              const fsaempc_qp_opts* opts, double* x, double* fval, int* exitflag, int* iter, double* lambda, bool remember) {
  const size_t n = q->nV, m = q->nC, B = k;
  if (!g || !lb || !ub || !x || (m && (!lbA || !ubA))) return fail(FSAEMPC_ERR_ARG, "null argument");
  for (size_t i = 0; i < B * n; ++i) if (!isfinite(g[i])) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): Argument 2 contains 'NaN' or 'Inf' !");
  for (size_t i = 0; i < B * n; ++i) if (isnan(lb[i]) || isnan(ub[i])) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): bounds contain 'NaN' !");
  for (size_t i = 0; i < B * m; ++i) if (isnan(lbA[i]) || isnan(ubA[i])) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): constraint bounds contain 'NaN' !");
  fsaempc_qp_desc d{q->nV, q->nC, k, 1};
  hipError_t e;
  if (k > q->kcap) {   // grow the per-call buffers: g lb ub lbA ubA x fval lambda (doubles), exitflag iter (ints), workspace
    (void)hipFree(q->dvec); (void)hipFree(q->dws); q->dvec = nullptr; q->dws = nullptr; q->kcap = 0;
    const long long wsb = fsaempc_qp_workspace_bytes(&d);
    if (wsb < 0) return fail((int)wsb, "workspace size");
    const size_t nd = B * (4 * n + 2 * m + 1 + (n + m)) + B + 16;
    e = hipMalloc((void**)&q->dvec, nd * sizeof(double)); if (e != hipSuccess) return hipfail(e, "hipMalloc");
    e = hipMalloc(&q->dws, (size_t)wsb); if (e != hipSuccess) return hipfail(e, "hipMalloc");
    q->kcap = k; q->wsb = wsb;
  }
  double* p = q->dvec;
  double* dg = p; p += B * n; double* dlb = p; p += B * n; double* dub = p; p += B * n; double* dlbA = p; p += B * m; double* dubA = p; p += B * m;
  double* dx = p; p += B * n; double* dfv = p; p += B; double* dlam = p; p += B * (n + m);
  int* dfl = (int*)p; int* dit = dfl + B;
  int rc = 0;
#define CP(dst, src, cnt) do { if (rc == 0 && (cnt) > 0) { e = hipMemcpy(dst, src, (cnt) * sizeof(double), hipMemcpyHostToDevice); if (e != hipSuccess) rc = hipfail(e, "hipMemcpy H2D"); } } while (0)
  CP(dg, g, B * n); CP(dlb, lb, B * n); CP(dub, ub, B * n); CP(dlbA, lbA, B * m); CP(dubA, ubA, B * m);
#undef CP
  if (rc == 0) rc = fsaempc_qp_solve_batch_device(&d, q->dH, dg, m ? q->dA : nullptr, dlb, dub, m ? dlbA : nullptr, m ? dubA : nullptr, opts,
                                                  dx, dfv, dfl, dit, dlam, q->dws, q->wsb, nullptr);
  if (rc == 0) { e = hipDeviceSynchronize(); if (e != hipSuccess) rc = hipfail(e, "solve"); }
  std::vector<double> lam0, x0h;
  if (rc == 0 && remember && !lambda) lam0.resize(n + m);
  if (rc == 0 && remember) x0h.resize(n);
#define BK(dst, src, bytes) do { if (rc == 0 && (dst)) { e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = hipfail(e, "hipMemcpy D2H"); } } while (0)
  BK(x, dx, B * n * sizeof(double)); BK(fval, dfv, B * sizeof(double)); BK(exitflag, dfl, B * sizeof(int)); BK(iter, dit, B * sizeof(int));
  BK(lambda, dlam, B * (n + m) * sizeof(double));
  if (!lam0.empty()) BK(lam0.data(), dlam, (n + m) * sizeof(double));
  if (!x0h.empty()) BK(x0h.data(), dx, n * sizeof(double));
#undef BK
  if (rc == 0 && remember) {
    // Working set of the first column (qpOASES.m:52-62 encoding), by the rule the kernels' active-set refinement uses: a side is
    // active iff its multiplier has the side's sign AND exceeds the side's slack.  On a refined vertex that is the same as
    // "multiplier non-zero"; on an interior-point iterate (refinement rejected or switched off, opts->polish = 0) every finite
    // side carries a small non-zero multiplier and the sign alone would put all of them into the set.
    const double* l = lambda ? lambda : lam0.data();
    q->wsB.assign(n, 0); q->wsC.assign(m, 0);
    const double ib = opts ? opts->inf_bound : 1e9;
    for (size_t i = 0; i < n; ++i) {
      const double v = x0h[i];
      if (l[i] > 0 && lb[i] > -ib && l[i] > fabs(v - lb[i])) q->wsB[i] = -1;
      else if (l[i] < 0 && ub[i] < ib && -l[i] > fabs(ub[i] - v)) q->wsB[i] = 1;
    }
    for (size_t r = 0; r < m; ++r) {
      double v = 0.0;
      for (size_t j = 0; j < n; ++j) v += q->hA[j * m + r] * x0h[j];
      const double lr = l[n + r];
      if (lr > 0 && lbA[r] > -ib && lr > fabs(v - lbA[r])) q->wsC[r] = -1;
      else if (lr < 0 && ubA[r] < ib && -lr > fabs(ubA[r] - v)) q->wsC[r] = 1;
    }
  }
  return rc;
}
}  // namespace

int fsaempc_seq_init(int nV, int nC, const double* H, const double* g, const double* A, const double* lb, const double* ub,
                     const double* lbA, const double* ubA, int k, const fsaempc_qp_opts* opts, int* handle,
                     double* x, double* fval, int* exitflag, int* iter, double* lambda) {
  if (!handle || !H || nV <= 0 || nC < 0 || k <= 0 || (nC > 0 && !A)) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): invalid arguments to 'i'");
  if (nV > FSAEMPC_MAX_NV) return fail(FSAEMPC_ERR_DIM, "nV exceeds FSAEMPC_MAX_NV");
  std::lock_guard<std::mutex> lk(g_seq_mu);
  int idx = -1;
  for (size_t i = 0; i < g_seq.size(); ++i) if (!g_seq[i].used) { idx = (int)i; break; }
  if (idx < 0) { g_seq.emplace_back(); idx = (int)g_seq.size() - 1; }
  SeqQP& q = g_seq[idx];
  q.nV = nV; q.nC = nC; q.used = true;
  *handle = idx + 1;
  int rc = seq_upload_matrices(&q, H, A);
  if (rc == 0) rc = seq_solve(&q, g, lb, ub, lbA, ubA, k, opts, x, fval, exitflag, iter, lambda, true);
  if (rc != 0) { q.release(); *handle = 0; }
  return rc;
}
int fsaempc_seq_hotstart(int handle, int nV, int nC, const double* g, const double* lb, const double* ub, const double* lbA,
                         const double* ubA, int k, const fsaempc_qp_opts* opts, double* x, double* fval, int* exitflag, int* iter, double* lambda) {
  std::lock_guard<std::mutex> lk(g_seq_mu);
  SeqQP* q = seq_get(handle);
  if (!q) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): Invalid handle to QP instance!");
  if (nV != q->nV || nC != q->nC) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): QP dimensions must be constant during a sequence!");
  if (k <= 0) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): invalid arguments to 'h'");
  return seq_solve(q, g, lb, ub, lbA, ubA, k, opts, x, fval, exitflag, iter, lambda, true);
}
int fsaempc_seq_hotstart_matrices(int handle, int nV, int nC, const double* H, const double* g, const double* A, const double* lb,
                                  const double* ub, const double* lbA, const double* ubA, int k, const fsaempc_qp_opts* opts,
                                  double* x, double* fval, int* exitflag, int* iter, double* lambda) {
  std::lock_guard<std::mutex> lk(g_seq_mu);
  SeqQP* q = seq_get(handle);
  if (!q) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): Invalid handle to QP instance!");
  if (nV != q->nV || nC != q->nC) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): QP dimensions must be constant during a sequence!");
  if (!H || (nC > 0 && !A) || k <= 0) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): invalid arguments to 'm'");
  int rc = seq_upload_matrices(q, H, A);
  if (rc != 0) return rc;
  return seq_solve(q, g, lb, ub, lbA, ubA, k, opts, x, fval, exitflag, iter, lambda, true);
}
// 'e' (qpOASES_sequence.m:64-72): the equality-constrained QP fixed by the working set of the handle's last solve -- sides in the
// working set hold with equality at the new bound values, every other bound / row is dropped ("might be violated").  Does not
// alter the handle.  Solved by the same kernel with lb = ub on the active sides and no bound elsewhere.
int fsaempc_seq_equality(int handle, int nV, int nC, const double* g, const double* lb, const double* ub, const double* lbA,
                         const double* ubA, int k, const fsaempc_qp_opts* opts, double* x, double* lambda, int* workingSetB, int* workingSetC) {
  std::lock_guard<std::mutex> lk(g_seq_mu);
  SeqQP* q = seq_get(handle);
  if (!q) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): Invalid handle to QP instance!");
  if (nV != q->nV || nC != q->nC) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): QP dimensions must be constant during a sequence!");
  if (k <= 0 || !g || !lb || !ub || !x || (nC && (!lbA || !ubA))) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): invalid arguments to 'e'");
  if ((int)q->wsB.size() != nV) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): no working set yet (solve with 'i' / 'h' / 'm' first)");
  const size_t n = nV, m = nC, B = k;
  std::vector<double> l2(B * n), u2(B * n), lA2(B * m), uA2(B * m);
  for (size_t j = 0; j < B; ++j) {
    for (size_t i = 0; i < n; ++i) {
      const int w = q->wsB[i]; const double v = w < 0 ? lb[j * n + i] : ub[j * n + i];
      l2[j * n + i] = w ? v : -INFINITY; u2[j * n + i] = w ? v : INFINITY;
    }
    for (size_t i = 0; i < m; ++i) {
      const int w = q->wsC[i]; const double v = w < 0 ? lbA[j * m + i] : ubA[j * m + i];
      lA2[j * m + i] = w ? v : -INFINITY; uA2[j * m + i] = w ? v : INFINITY;
    }
  }
  std::vector<int> fl(B), it(B);
  int rc = seq_solve(q, g, l2.data(), u2.data(), m ? lA2.data() : nullptr, m ? uA2.data() : nullptr, k, opts, x, nullptr, fl.data(), it.data(), lambda, false);
  if (rc != 0) return rc;
  if (workingSetB) for (size_t i = 0; i < n; ++i) workingSetB[i] = q->wsB[i];
  if (workingSetC) for (size_t i = 0; i < m; ++i) workingSetC[i] = q->wsC[i];
  for (size_t j = 0; j < B; ++j) if (fl[j] != 0) return fail(FSAEMPC_ERR_SOLVER, "ERROR (qpOASES): equality-constrained QP could not be solved (working set linearly dependent or infeasible)");
  return 0;
}
int fsaempc_seq_cleanup(int handle) {
  std::lock_guard<std::mutex> lk(g_seq_mu);
  SeqQP* q = seq_get(handle);
  if (!q) return fail(FSAEMPC_ERR_ARG, "ERROR (qpOASES): Invalid handle to QP instance!");   // what main.m:193 would hit with QP == 0
  q->release();
  return 0;
}

int fsaempc_ltv_nx(int model) { return model == FSAEMPC_MODEL_KINEMATIC ? 5 : 7; }
static int ltv_ns(int model) { return model == FSAEMPC_MODEL_KINEMATIC ? 1 : 4; }
int fsaempc_ltv_nV(int model, int N) { return 2 * N + ltv_ns(model); }
int fsaempc_ltv_nC(int model, int N) { return (model == FSAEMPC_MODEL_KINEMATIC ? 6 : 20) * N; }

static int ltv_check(const fsaempc_ltv_desc* d, const fsaempc_spline* sp) {
  if (!d || !sp || !sp->xP || !sp->yP) return fail(FSAEMPC_ERR_ARG, "null argument");
  if (d->model != FSAEMPC_MODEL_KINEMATIC && d->model != FSAEMPC_MODEL_DYNAMIC) return fail(FSAEMPC_ERR_ARG, "unknown model");
  if (d->N <= 0 || d->batch < 0 || !(d->dt > 0) || sp->M <= 0 || !(sp->dl > 0)) return fail(FSAEMPC_ERR_ARG, "bad dimensions");
  if (d->integrator < FSAEMPC_INT_DEFAULT || d->integrator > FSAEMPC_INT_RK4) return fail(FSAEMPC_ERR_ARG, "unknown integrator");
  if (ltv_build_lds_bytes(fsaempc_ltv_nx(d->model), d->N, 256) > 160 * 1024) return fail(FSAEMPC_ERR_DIM, "horizon too long for the LDS staging");
  return 0;
}

int fsaempc_ltv_build_qp_batch_device(const fsaempc_ltv_desc* desc, const fsaempc_spline* sp,
                                      const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                                      double* H, double* g, double* A, double* lb, double* ub, double* lbA, double* ubA,
                                      double* pred, double* Bt, double* qconst, void* stream) {
  int rc = ltv_check(desc, sp); if (rc) return rc;
  if (!x0 || !x_ref || !x_lin || !u_lin || !H || !g || !A || !lb || !ub || !lbA || !ubA || !Bt) return fail(FSAEMPC_ERR_ARG, "null argument (Bt is required as scratch)");
  if (desc->batch == 0) return 0;
  LtvParams P; memset(&P, 0, sizeof(P));
  P.nx = fsaempc_ltv_nx(desc->model); P.N = desc->N; P.dt = desc->dt;
  P.integ = desc->integrator >= 0 ? desc->integrator : (desc->model == FSAEMPC_MODEL_KINEMATIC ? FSAEMPC_INT_RK2 : FSAEMPC_INT_RK4);   // ltvmpc_*.m:38
  P.spM = sp->M; P.spdl = sp->dl; P.xP = sp->xP; P.yP = sp->yP;
  P.x0 = x0; P.x_ref = x_ref; P.x_lin = x_lin; P.u_lin = u_lin;
  P.H = H; P.g = g; P.A = A; P.lb = lb; P.ub = ub; P.lbA = lbA; P.ubA = ubA; P.pred = pred; P.Bt = Bt; P.qconst = qconst;
  hipError_t e = ltv_build_launch(P, desc->batch, (hipStream_t)stream);
  if (e != hipSuccess) return hipfail(e, "ltv_build_launch");
  return 0;
}

struct LtvCarve { size_t H, g, A, lb, ub, lbA, ubA, pred, Bt, qc, z, qpws, total; };
static void ltv_carve(const fsaempc_ltv_desc* d, LtvCarve* c) {
  const size_t B = d->batch > 0 ? d->batch : 1, nx = fsaempc_ltv_nx(d->model), N = d->N;
  const size_t nV = fsaempc_ltv_nV(d->model, d->N), nC = fsaempc_ltv_nC(d->model, d->N), R = nx * N;
  size_t off = 0;
  auto take = [&](size_t cnt) { size_t o = off; off = align64(off + cnt * sizeof(double)); return o; };
  c->H = take(B * nV * nV); c->g = take(B * nV); c->A = take(B * nC * nV); c->lb = take(B * nV); c->ub = take(B * nV);
  c->lbA = take(B * nC); c->ubA = take(B * nC); c->pred = take(B * R); c->Bt = take(B * R * nV); c->qc = take(B); c->z = take(B * nV);
  off = (off + 255) & ~(size_t)255;
  c->qpws = off;
  fsaempc_qp_desc q{(int)nV, (int)nC, (int)B, 0};
  long long w = fsaempc_qp_workspace_bytes(&q);
  c->total = off + (w > 0 ? (size_t)w : 0);
}

long long fsaempc_ltv_workspace_bytes(const fsaempc_ltv_desc* desc) {
  if (!desc || desc->N <= 0 || desc->batch < 0) return FSAEMPC_ERR_ARG;
  if (fsaempc_ltv_nV(desc->model, desc->N) > FSAEMPC_MAX_NV) return FSAEMPC_ERR_DIM;
  LtvCarve c; ltv_carve(desc, &c);
  return (long long)c.total;
}

int fsaempc_ltv_step_batch_device(const fsaempc_ltv_desc* desc, const fsaempc_spline* sp,
                                  const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                                  const fsaempc_qp_opts* opts, double* u_opt, double* x_opt, double* slack, double* fval,
                                  int* exitflag, int* iter, void* workspace, long long workspace_bytes, void* stream) {
  return fsaempc_ltv_step_batch_device_aux(desc, sp, x0, x_ref, x_lin, u_lin, opts, u_opt, x_opt, slack, fval, exitflag, iter, nullptr, workspace, workspace_bytes, stream);
}

int fsaempc_ltv_step_batch_device_aux(const fsaempc_ltv_desc* desc, const fsaempc_spline* sp,
                                      const double* x0, const double* x_ref, const double* x_lin, const double* u_lin,
                                      const fsaempc_qp_opts* opts, double* u_opt, double* x_opt, double* slack, double* fval,
                                      int* exitflag, int* iter, const fsaempc_qp_aux* aux, void* workspace, long long workspace_bytes, void* stream) {
  int rc = ltv_check(desc, sp); if (rc) return rc;
  if (!x0 || !x_ref || !x_lin || !u_lin || !u_opt || !x_opt || !slack || !fval || !exitflag || !iter || !workspace) return fail(FSAEMPC_ERR_ARG, "null argument");
  if (fsaempc_ltv_nV(desc->model, desc->N) > FSAEMPC_MAX_NV) return fail(FSAEMPC_ERR_DIM, "nV exceeds FSAEMPC_MAX_NV");
  if (desc->batch == 0) return 0;
  LtvCarve c; ltv_carve(desc, &c);
  if ((long long)c.total > workspace_bytes) return fail(FSAEMPC_ERR_WORKSPACE, "workspace too small");
  char* w = (char*)workspace;
  auto D = [&](size_t off) { return (double*)(w + off); };
  Range whole("fsaempc.ltv.step");
  const bool timing = g_timing.load();
  if (timing) { hipError_t e = hipEventRecord(g_evf[0], (hipStream_t)stream); if (e != hipSuccess) return hipfail(e, "hipEventRecord"); }
  { Range r("fsaempc.ltv.build");
    rc = fsaempc_ltv_build_qp_batch_device(desc, sp, x0, x_ref, x_lin, u_lin, D(c.H), D(c.g), D(c.A), D(c.lb), D(c.ub), D(c.lbA), D(c.ubA),
                                           D(c.pred), D(c.Bt), D(c.qc), stream); }
  if (rc) return rc;
  fsaempc_qp_desc q{fsaempc_ltv_nV(desc->model, desc->N), fsaempc_ltv_nC(desc->model, desc->N), desc->batch, 0};
  rc = fsaempc_qp_solve_batch_device_aux(&q, D(c.H), D(c.g), D(c.A), D(c.lb), D(c.ub), D(c.lbA), D(c.ubA), opts, D(c.z), fval, exitflag, iter,
                                         nullptr, aux, w + c.qpws, (long long)(c.total - c.qpws), stream);
  if (rc) return rc;
  hipError_t e;
  { Range r("fsaempc.ltv.post");
    e = ltv_post_launch(fsaempc_ltv_nx(desc->model), desc->N, ltv_ns(desc->model), desc->batch, D(c.z), D(c.pred), D(c.Bt), D(c.qc),
                        u_opt, x_opt, slack, fval, (hipStream_t)stream); }
  if (e != hipSuccess) return hipfail(e, "ltv_post_launch");
  if (timing) { e = hipEventRecord(g_evf[1], (hipStream_t)stream); if (e != hipSuccess) return hipfail(e, "hipEventRecord"); }
  return 0;
}

int fsaempc_obtain_reference_batch_device(const double* plan, double ds, int N_s, const double* t, const double* s0, double dt,
                                          int N_t, int batch, double* x_ref, void* stream) {
  if (!plan || !t || !s0 || !x_ref) return fail(FSAEMPC_ERR_ARG, "null argument");
  if (N_s <= 0 || N_t <= 0 || batch < 0 || !(ds > 0) || !(dt > 0)) return fail(FSAEMPC_ERR_ARG, "bad dimensions");
  RefParams P; P.plan = plan; P.t = t; P.s0 = s0; P.x_ref = x_ref; P.ds = ds; P.dt = dt; P.N_s = N_s; P.N_t = N_t; P.batch = batch;
  hipError_t e = obtain_reference_launch(P, (hipStream_t)stream);
  if (e != hipSuccess) return hipfail(e, "obtain_reference_launch");
  return 0;
}

int fsaempc_reference_live_batch_device(int nx, int N, double dt, double target_vel, int batch, const double* x0, double* x_ref, void* stream) {
  if (!x0 || !x_ref) return fail(FSAEMPC_ERR_ARG, "null argument");
  if ((nx != 5 && nx != 7) || N <= 0 || batch < 0 || !(dt > 0)) return fail(FSAEMPC_ERR_ARG, "bad dimensions");
  hipError_t e = reference_live_launch(nx, N, dt, target_vel, batch, x0, x_ref, (hipStream_t)stream);
  if (e != hipSuccess) return hipfail(e, "reference_live_launch");
  return 0;
}

int fsaempc_cl_pre_batch_device(int model, int N, double dt, double target_vel, double L, const fsaempc_spline* sp, const double* cart,
                                const double* s_guess, int batch, double* x0, double* x_ref, int* finished, void* stream) {
  if (!sp || !sp->xP || !sp->yP || !cart || !s_guess || !x0 || !x_ref || !finished) return fail(FSAEMPC_ERR_ARG, "null argument");
  if (model != FSAEMPC_MODEL_KINEMATIC && model != FSAEMPC_MODEL_DYNAMIC) return fail(FSAEMPC_ERR_ARG, "unknown model");
  if (N <= 0 || batch < 0 || !(dt > 0) || sp->M <= 0 || !(sp->dl > 0) || !(L > 0)) return fail(FSAEMPC_ERR_ARG, "bad dimensions");
  ClPreParams P; P.nx = fsaempc_ltv_nx(model); P.N = N; P.batch = batch; P.dt = dt; P.target_vel = target_vel; P.L = L;
  P.spM = sp->M; P.spdl = sp->dl; P.xP = sp->xP; P.yP = sp->yP; P.cart = cart; P.s_guess = s_guess; P.x0 = x0; P.x_ref = x_ref; P.finished = finished;
  hipError_t e = cl_pre_launch(P, (hipStream_t)stream);
  if (e != hipSuccess) return hipfail(e, "cl_pre_launch");
  return 0;
}

int fsaempc_cl_plant_batch_device(int model, int N, double dt, int batch, double* cart, double* pid, const double* x_opt,
                                  const int* finished, const int* exitflag, double* u_last, void* stream) {
  if (!cart || !pid || !x_opt) return fail(FSAEMPC_ERR_ARG, "null argument");
  if (model != FSAEMPC_MODEL_KINEMATIC && model != FSAEMPC_MODEL_DYNAMIC) return fail(FSAEMPC_ERR_ARG, "unknown model");
  if (N <= 0 || batch < 0 || !(dt > 0)) return fail(FSAEMPC_ERR_ARG, "bad dimensions");
  ClPlantParams P; P.nx = fsaempc_ltv_nx(model); P.N = N; P.batch = batch; P.dt = dt; P.cart = cart; P.pid = pid; P.x_opt = x_opt;
  P.finished = finished; P.exitflag = exitflag; P.u_last = u_last;
  hipError_t e = cl_plant_launch(P, (hipStream_t)stream);
  if (e != hipSuccess) return hipfail(e, "cl_plant_launch");
  return 0;
}

int fsaempc_cl_accept_batch_device(int model, int N, int batch, const double* x_new, const double* u_new, const int* exitflag, double* x_keep, double* u_keep, void* stream) {
  if (!x_new || !u_new || !x_keep || !u_keep) return fail(FSAEMPC_ERR_ARG, "null argument");
  if (model != FSAEMPC_MODEL_KINEMATIC && model != FSAEMPC_MODEL_DYNAMIC) return fail(FSAEMPC_ERR_ARG, "unknown model");
  if (N <= 0 || batch < 0) return fail(FSAEMPC_ERR_ARG, "bad dimensions");
  hipError_t e = cl_accept_launch(fsaempc_ltv_nx(model) * N, 2 * N, batch, x_new, u_new, exitflag, x_keep, u_keep, (hipStream_t)stream);
  if (e != hipSuccess) return hipfail(e, "cl_accept_launch");
  return 0;
}

int fsaempc_selftest_mfma(void) {
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt == 0) return fail(FSAEMPC_ERR_NODEVICE, "no HIP device");
  char msg[256] = "";
  int bad = qp_selftest_mfma(msg, sizeof(msg));
  if (bad != 0) snprintf(g_err, sizeof(g_err), "%s", msg);
  return bad;
}

int fsaempc_debug_set_dump(double* out, int stage) { g_dump.store(out); g_dump_stage.store(stage); return 0; }

int fsaempc_qp_set_timing(int enable) {
  if (enable && !g_ev[0]) {
    for (int i = 0; i < 3; ++i) { hipError_t e = hipEventCreate(&g_ev[i]); if (e != hipSuccess) return hipfail(e, "hipEventCreate"); }
    for (int i = 0; i < 2; ++i) { hipError_t e = hipEventCreate(&g_evf[i]); if (e != hipSuccess) return hipfail(e, "hipEventCreate"); }
  }
  g_timing.store(enable != 0);
  return 0;
}

int fsaempc_ltv_get_timing(double* build_ms, double* prep_ms, double* solve_ms, double* post_ms) {
  if (!g_evf[0]) return fail(FSAEMPC_ERR_ARG, "timing was never enabled");
  hipError_t e = hipEventSynchronize(g_evf[1]);
  if (e != hipSuccess) return hipfail(e, "hipEventSynchronize (no fused step since timing was enabled?)");
  float t[4] = {0, 0, 0, 0};
  e = hipEventElapsedTime(&t[0], g_evf[0], g_ev[0]); if (e != hipSuccess) return hipfail(e, "hipEventElapsedTime");
  e = hipEventElapsedTime(&t[1], g_ev[0], g_ev[1]); if (e != hipSuccess) return hipfail(e, "hipEventElapsedTime");
  e = hipEventElapsedTime(&t[2], g_ev[1], g_ev[2]); if (e != hipSuccess) return hipfail(e, "hipEventElapsedTime");
  e = hipEventElapsedTime(&t[3], g_ev[2], g_evf[1]); if (e != hipSuccess) return hipfail(e, "hipEventElapsedTime");
  if (build_ms) *build_ms = t[0];
  if (prep_ms) *prep_ms = t[1];
  if (solve_ms) *solve_ms = t[2];
  if (post_ms) *post_ms = t[3];
  return 0;
}

int fsaempc_qp_get_timing(double* prep_ms, double* solve_ms) {
  if (!g_ev[0]) return fail(FSAEMPC_ERR_ARG, "timing was never enabled");
  hipError_t e = hipEventSynchronize(g_ev[2]);
  if (e != hipSuccess) return hipfail(e, "hipEventSynchronize");
  float a = 0, b = 0;
  e = hipEventElapsedTime(&a, g_ev[0], g_ev[1]); if (e != hipSuccess) return hipfail(e, "hipEventElapsedTime");
  e = hipEventElapsedTime(&b, g_ev[1], g_ev[2]); if (e != hipSuccess) return hipfail(e, "hipEventElapsedTime");
  if (prep_ms) *prep_ms = a;
  if (solve_ms) *solve_ms = b;
  return 0;
}

}  // extern "C"
