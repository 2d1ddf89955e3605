// qp_wg.hip -- batched dense convex QP solve on MI355X (gfx950): ONE WORKGROUP of W wavefronts per QP.
//
// Replaces the qpOASES MEX call of the reference
//   (mpc/ltv/kinematic/ltvmpc_kinetmatic_curvilinear.m:52, mpc/ltv/dynamic/ltvmpc_dynamic_curvilinear.m:52,
//    contract optimizers/matlab/qpOASES/qpOASES.m:16-62)
// with a primal-dual interior-point method (Mehrotra predictor-corrector, single step length, OOQP-style step
// heuristic) followed by an active-set refinement to the vertex an active-set solver stops at.
//
// Round-2 shape (the round-1 kernel ran one wavefront per QP with all 512 registers, spilled, and re-streamed A three
// times per iteration from the L2/Infinity Cache):
//   * the upper tiles of the normal matrix M = H + A'DA (and of its Cholesky factor) are dealt round-robin to the W
//     wavefronts, NT/W accumulator tiles each -> no spills up to T = 12 (nV <= 196);
//   * the operand stream of A~ (written once by qp_prep_kernel, see qp_solver.hip) is copied into LDS once per QP when
//     it fits (kinematic shapes: 93 KB for N = 40) and every pass of every iteration reads it from there; otherwise the
//     passes read it from global memory (dynamic shapes);
//   * pass 1 (SYRK on the matrix cores) is split by tiles, the matrix-vector passes by trips (16 sorted rows), the row
//     sweeps by owner-layout slots; wave-uniform scalars are reduced through a small LDS scratch so that every wave
//     takes the same branches;
//   * the blocked Cholesky runs right-looking over the distributed tiles: diagonal tile in one wave (four 4-row panels
//     on the matrix cores), U_KK^-T and the panel row U_K* pass through LDS, two barriers per block step; right-hand
//     sides live in LDS as plain vectors and are updated by the owner of the tile that couples them.
// No hand-placed s_waitcnt, no LDS-DMA: every cross-wave hand-off is a __syncthreads().
//
// Data layout: see qp_solver.hip (qp_prep_kernel).  fp64 MFMA lane maps (cdna_hip_programming.md section 3):
// A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C/D col = l&15, row = (l>>4) + 4*reg.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "qp_solver.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
#define DEVINL __device__ __forceinline__

extern __shared__ __attribute__((aligned(16))) double slds[];

namespace {

DEVINL double rl(double v, int src) {  // wave-uniform broadcast of lane `src`
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int CTRL> DEVINL double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
DEVINL double grp16_sum(double v) {  // sum over the 16 lanes sharing l>>4 (one DPP row); every lane gets the total
  v += dpp_f64<0xB1>(v); v += dpp_f64<0x4E>(v); v += dpp_f64<0x141>(v); v += dpp_f64<0x140>(v);
  return v;
}
DEVINL double grp16_max(double v) {
  v = fmax(v, dpp_f64<0xB1>(v)); v = fmax(v, dpp_f64<0x4E>(v)); v = fmax(v, dpp_f64<0x141>(v)); v = fmax(v, dpp_f64<0x140>(v));
  return v;
}
DEVINL double grp16_min(double v) {
  v = fmin(v, dpp_f64<0xB1>(v)); v = fmin(v, dpp_f64<0x4E>(v)); v = fmin(v, dpp_f64<0x141>(v)); v = fmin(v, dpp_f64<0x140>(v));
  return v;
}
// whole-wave reductions (all 64 lanes active at the call): DPP within the four rows, then four scalar lane reads
DEVINL double wave_sum(double v) { v = grp16_sum(v); return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48)); }
DEVINL double wave_max(double v) { v = grp16_max(v); return fmax(fmax(rl(v, 0), rl(v, 16)), fmax(rl(v, 32), rl(v, 48))); }
DEVINL double wave_min(double v) { v = grp16_min(v); return fmin(fmin(rl(v, 0), rl(v, 16)), fmin(rl(v, 32), rl(v, 48))); }
DEVINL double q_sum(double v) {  // sum over the 4 lane groups (same l&15)
  v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
  return v;
}

enum RowArr { R_L = 0, R_U, R_TL, R_TU, R_ZL, R_ZU, R_V, R_D, R_W1, R_W2, R_W3, R_VA, R_VC, R_RPL, R_RPU, R_CB1, R_CC1, R_CB2, R_CC2, R_NARR };
enum VecArr { V_X = 0, V_G, V_HX, V_P1, V_P2, V_P3, V_DX, V_E, V_R1, V_R2, V_NARR };   // R1, R2 and the 4 border-column vectors MB[b] behind them are contiguous (right-hand-side columns 0..5)

DEVINL void mfma4_sub(const v4d& X, const v4d& Y, v4d& Dst) {  // Dst -= X' Y  (X, Y, Dst in C/D layout)
#pragma unroll
  for (int p = 0; p < 4; ++p) Dst = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[p], Y[p], Dst, 0, 0, 0);
}
DEVINL v4d mfma4_new(const v4d& X, const v4d& Y) {  // X' Y
  v4d Z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int p = 0; p < 4; ++p) Z = __builtin_amdgcn_mfma_f64_16x16x4f64(X[p], Y[p], Z, 0, 0, 0);
  return Z;
}

// Factorise one 16x16 diagonal tile D = U'U in place, four 4-row panels, and apply the same row operations to the
// companion tile Yk (enters as the identity, leaves as U^-T).  Per panel p: the 4x4 diagonal block (10 numbers, read
// with v_readlane) is factorised and inverted redundantly by every lane -- W = R^-T, wave-uniform -- and applied to
// the panel rows of both tiles as one K=4 MFMA each; the rows of the later panels are then updated by one more K=4
// MFMA per tile.  No cross-lane data movement besides the readlanes, no LDS.
DEVINL int diag_factor(int c, int q, v4d& Ud, v4d& Yk, double floor_abs) {
  int bad = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    // D[a][b] = M[4p+a][4p+b] lives in lane (c = 4p+b, q = a), register p
    const double d00 = rl(Ud[p], 4 * p + 0), d01 = rl(Ud[p], 4 * p + 1), d02 = rl(Ud[p], 4 * p + 2), d03 = rl(Ud[p], 4 * p + 3);
    const double d11 = rl(Ud[p], 16 + 4 * p + 1), d12 = rl(Ud[p], 16 + 4 * p + 2), d13 = rl(Ud[p], 16 + 4 * p + 3);
    const double d22 = rl(Ud[p], 32 + 4 * p + 2), d23 = rl(Ud[p], 32 + 4 * p + 3);
    const double d33 = rl(Ud[p], 48 + 4 * p + 3);
    auto piv = [&](double t) __attribute__((always_inline)) { if (!(t > floor_abs)) { if (!(fabs(t) < INFINITY)) bad = 1; t = floor_abs; } return rsqrt(t); };
    const double i0 = piv(d00);
    const double r01 = d01 * i0, r02 = d02 * i0, r03 = d03 * i0;
    const double i1 = piv(fma(-r01, r01, d11));
    const double r12 = fma(-r01, r02, d12) * i1, r13 = fma(-r01, r03, d13) * i1;
    const double i2 = piv(fma(-r12, r12, fma(-r02, r02, d22)));
    const double r23 = fma(-r12, r13, fma(-r02, r03, d23)) * i2;
    const double i3 = piv(fma(-r23, r23, fma(-r13, r13, fma(-r03, r03, d33))));
    const double w10 = -r01 * i0 * i1;
    const double w20 = -fma(r12, w10, r02 * i0) * i2, w21 = -r12 * i1 * i2;
    const double w30 = -fma(r23, w20, fma(r13, w10, r03 * i0)) * i3, w31 = -fma(r23, w21, r13 * i1) * i3, w32 = -r23 * i2 * i3;
    const int a_ = c - 4 * p;
    double wa = 0.0;
    if (q == 0) wa = a_ == 0 ? i0 : (a_ == 1 ? w10 : (a_ == 2 ? w20 : (a_ == 3 ? w30 : 0.0)));
    if (q == 1) wa = a_ == 1 ? i1 : (a_ == 2 ? w21 : (a_ == 3 ? w31 : 0.0));
    if (q == 2) wa = a_ == 2 ? i2 : (a_ == 3 ? w32 : 0.0);
    if (q == 3) wa = a_ == 3 ? i3 : 0.0;
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    const v4d nu = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, Ud[p], z, 0, 0, 0);
    const v4d ny = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, Yk[p], z, 0, 0, 0);
    Ud[p] = nu[p]; Yk[p] = ny[p];
    if (p < 3) {
      const double a = (c > 4 * p + 3) ? -Ud[p] : 0.0;
      Yk = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Yk[p], Yk, 0, 0, 0);
      Ud = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Ud[p], Ud, 0, 0, 0);
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) if (c < q + 4 * p) Ud[p] = 0.0;   // strictly lower part only ever held the symmetric copy
  return bad;
}

#ifndef QP_STAMPS
#define QP_STAMPS 0
#endif
#if QP_STAMPS   // diagnostic build only: cycles of wave 0 per phase (barrier waits included), written to P.dump[b*16 + phase]
#define STAMP_DECL unsigned long long st_acc[16]; for (int i_ = 0; i_ < 16; ++i_) st_acc[i_] = 0; unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#define STAMP(id) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[id] += t_ - st_t0; st_t0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_OUT do { if (P.dump && P.dump_stage == 9 && tid == 0) for (int i_ = 0; i_ < 16; ++i_) P.dump[(size_t)b * 16 + i_] = (double)st_acc[i_]; } while (0)
#else
#define STAMP_DECL
#define STAMP(id) do { } while (0)
#define STAMP_OUT do { } while (0)
#endif

template <int T> struct TileMap {   // column-major order of the upper triangle: index i = J(J+1)/2 + I, I <= J
  static constexpr int NT = T * (T + 1) / 2;
};

// ---------------------------------------------------------------------------------------------
// solve kernel: T column tiles of 16, NB border columns (0 or 4), W wavefronts per QP
// ---------------------------------------------------------------------------------------------
template <int T, int NB, int W, bool RES> __global__ __launch_bounds__(64 * W) void qp_wg_kernel(QpParams P) {
  constexpr int NT = TileMap<T>::NT;
  constexpr int NTW = (NT + W - 1) / W;         // accumulator tiles per wave
  constexpr int CW = (T + W - 1) / W;           // column tiles per wave (A'w products of pass 1, H x)
  constexpr int NBB = NB > 0 ? NB : 1;
  constexpr int NTH = 64 * W;
  const int b = blockIdx.x;
  const QpDims& d = P.d;
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = d.n, m = d.m, Kq = d.Kq, J = d.J, JB = d.JB, JT = d.J + d.JB, np = d.np, nc = d.nc, nb = d.nb, ntr = d.ntr;
  const int rowlen = d.rowlen, JS = d.J * 64;
  double* __restrict__ ws = P.ws + (size_t)b * d.ws_per_qp;
  const double* __restrict__ Awg = ws + d.off_Aw;
  const double* __restrict__ Hw = ws + d.off_Hw;
  const double* __restrict__ Ab = ws + d.off_Ab;
  const double* __restrict__ Hb = ws + d.off_Hb;
  double* __restrict__ rows = ws + d.off_rows;
  const int* __restrict__ perm = reinterpret_cast<const int*>(ws + d.off_meta);
  const int* __restrict__ tcs = perm + (size_t)(d.J > 0 ? d.J : 1) * 64;
  const int* __restrict__ aoff = tcs + d.ntr;
  const double* __restrict__ gw = ws + d.off_gw;
  const double* __restrict__ Es = ws + d.off_E;
  const double* __restrict__ Fs = ws + d.off_F;

  // ---- LDS carve (offsets in doubles; host mirror: qp_wg_lds_bytes) ----
  const int oMB = V_NARR * np;                 // 4 border-column vectors
  const int oYL = oMB + 4 * np;                // T tiles U_KK^-T, row-major, 17-double rows
  const int oPB = oYL + T * 272;               // T panel-row tiles in register image [p][lane]
  const int oWP = oPB + T * 256;               // 2 x W partial n-vectors (cross-wave sums of A'w products)
  const int oRed = oWP + 2 * W * np;           // reduction scratch: 2 buffers x 8 values x W
  const int oAw = oRed + 2 * 8 * W;            // resident operand stream (when it fits)
#define VEC(a) ((a) * np)
#define X_(i) slds[VEC(V_X) + (i)]
#define G_(i) slds[VEC(V_G) + (i)]
#define HX_(i) slds[VEC(V_HX) + (i)]
#define R1_(i) slds[VEC(V_R1) + (i)]
#define R2_(i) slds[VEC(V_R2) + (i)]
#define P1_(i) slds[VEC(V_P1) + (i)]
#define P2_(i) slds[VEC(V_P2) + (i)]
#define P3_(i) slds[VEC(V_P3) + (i)]
#define DX_(i) slds[VEC(V_DX) + (i)]
#define EV_(i) slds[VEC(V_E) + (i)]
#define MB_(e, i) slds[oMB + (e) * np + (i)]
#define ROW(a) (rows + (size_t)(a) * rowlen)
  double* __restrict__ aL = ROW(R_L); double* __restrict__ aU = ROW(R_U);
  double* __restrict__ aTL = ROW(R_TL); double* __restrict__ aTU = ROW(R_TU);
  double* __restrict__ aZL = ROW(R_ZL); double* __restrict__ aZU = ROW(R_ZU);
  double* __restrict__ aV = ROW(R_V); double* __restrict__ aD = ROW(R_D);
  double* __restrict__ aW1 = ROW(R_W1); double* __restrict__ aW2 = ROW(R_W2); double* __restrict__ aW3 = ROW(R_W3);
  double* __restrict__ aVA = ROW(R_VA); double* __restrict__ aVC = ROW(R_VC);
  double* __restrict__ aRPL = ROW(R_RPL); double* __restrict__ aRPU = ROW(R_RPU);
  double* __restrict__ aCB1 = ROW(R_CB1); double* __restrict__ aCC1 = ROW(R_CC1);
  double* __restrict__ aCB2 = ROW(R_CB2); double* __restrict__ aCC2 = ROW(R_CC2);

  // ---- my accumulator tiles: linear index i = w + W t in the column-major upper triangle ----
  int tI[NTW], tJ[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int i = w + W * t;
    int Jc = 0;
    while ((Jc + 1) * (Jc + 2) / 2 <= i) ++Jc;
    tJ[t] = i < NT ? Jc : 1 << 20;             // out-of-range tiles never match a step
    tI[t] = i < NT ? i - Jc * (Jc + 1) / 2 : 1 << 20;
  }

  // ---- operand stream: RES = resident in LDS (copied once), else read from global memory by every pass.  A QP whose
  //      stream does not fit the reserved LDS is handed to the streaming kernel (launched right behind this one) by the
  //      sentinel exit flag QP_FLAG_PENDING; the streaming kernel in `only_pending` mode skips everything else. ----
  const int nrec = aoff[ntr];
  if (RES) {
    if ((size_t)nrec * 1024 > (size_t)d.lds_aw_bytes) { if (tid == 0) P.exitflag[b] = QP_FLAG_PENDING; return; }
    for (int r = w; r < nrec; r += W) {
      const v2d v = *reinterpret_cast<const v2d*>(Awg + (size_t)r * 128 + lane * 2);
      *reinterpret_cast<v2d*>(&slds[oAw + r * 128 + lane * 2]) = v;
    }
  } else if (P.only_pending) {
    if (P.exitflag[b] != QP_FLAG_PENDING) return;
  }
  auto opnd = [&](int rec) __attribute__((always_inline)) -> v2d {   // one 1 KB record: lane (c,q) gets its two k-steps
    if constexpr (RES) return *reinterpret_cast<const v2d*>(&slds[oAw + rec * 128 + lane * 2]);
    else return *reinterpret_cast<const v2d*>(Awg + (size_t)rec * 128 + lane * 2);
  };

  // ---- workgroup reductions of wave-uniform scalars (double-buffered scratch, one barrier each) ----
  int red_buf = 0;
  auto red_put = [&](int slot, double v) __attribute__((always_inline)) { if (lane == 0) slds[oRed + red_buf * 8 * W + slot * W + w] = v; };
  auto red_sync = [&]() __attribute__((always_inline)) { __syncthreads(); };
  auto red_sum = [&](int slot) __attribute__((always_inline)) { double s = 0; for (int i = 0; i < W; ++i) s += slds[oRed + red_buf * 8 * W + slot * W + i]; return s; };
  auto red_max = [&](int slot) __attribute__((always_inline)) { double s = -INFINITY; for (int i = 0; i < W; ++i) s = fmax(s, slds[oRed + red_buf * 8 * W + slot * W + i]); return s; };
  auto red_min = [&](int slot) __attribute__((always_inline)) { double s = INFINITY; for (int i = 0; i < W; ++i) s = fmin(s, slds[oRed + red_buf * 8 * W + slot * W + i]); return s; };
  auto red_next = [&]() __attribute__((always_inline)) { red_buf ^= 1; };

  auto row_valid = [&](int js) __attribute__((always_inline)) -> bool {
    if (js < J) { const int s = 16 * js + c; return s < Kq && 4 * s + q < m; }
    return (js - J) * 64 + lane < n;
  };
  auto ixv = [&](int i) __attribute__((always_inline)) { return (J + (i >> 6)) * 64 + (i & 63); };   // owner-layout index of variable-bound row i

  // y = A~ v for NVEC LDS vectors -> owner-layout row arrays, split by trips.  FUSE 1: second-order weight of the
  // corrector formed row by row and A~'w accumulated in the same pass (partial sums per wave in WP[0]).
  // FUSE 2: polish evaluation (see below).  FUSE 3: A~'w for a row array (initial point).
  auto part = [&](int which, int i) __attribute__((always_inline)) { double s = 0; for (int ww = 0; ww < W; ++ww) s += slds[oWP + (which * W + ww) * np + i]; return s; };

  // ------------------------------------------------------------------------------------------
  // generic streaming pass over the rows of A~, split by trips.  The row callback gets, per k-step pair (two
  // consecutive sorted k-steps of lane group q), the operands b[t] and decides what to do.
  // ------------------------------------------------------------------------------------------
#define TRIP_LOOP_BEGIN                                                                                   \
  for (int tr = w; tr < ntr; tr += W) {                                                                   \
    const int tc = tcs[tr], rbase = aoff[tr];                                                             \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                       \
      v2d bq[T];                                                                                          \
      _Pragma("unroll") for (int t = 0; t < T; ++t) { bq[t] = (v2d){0.0, 0.0}; if (t < tc) bq[t] = opnd(rbase + u * tc + t); } \
      const int s0 = 4 * tr + 2 * u;                                                                      \
      const int rix = (s0 >> 4) * 64 + q * 16 + (s0 & 15);   /* owner-layout index of k-step s0, lane group q (s0 even: 16-B aligned pairs) */
#define TRIP_LOOP_END } }

  // v = A~ x for one LDS vector -> owner-layout row array (A rows only), split by trips
  auto pass_Av = [&](int oV, double* __restrict__ rout) __attribute__((always_inline)) {
    double v[T], vb[NBB];
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = slds[oV + 16 * t + c];
#pragma unroll
    for (int f = 0; f < NBB; ++f) vb[f] = NB ? slds[oV + nc + f] : 0.0;
    TRIP_LOOP_BEGIN
      v2d ab[NBB];
#pragma unroll
      for (int f = 0; f < NB; ++f) ab[f] = *reinterpret_cast<const v2d*>(Ab + (size_t)f * JS + rix);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double dsum = 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) dsum = fma(bq[t][h], v[t], dsum);
        dsum = grp16_sum(dsum);
#pragma unroll
        for (int f = 0; f < NB; ++f) dsum = fma(ab[f][h], vb[f], dsum);
        if (c == ((s0 + h) & 15)) rout[rix + h] = dsum;
      }
    TRIP_LOOP_END
  };

  // Hx = H~ x: wave w takes the column tiles Jt = w, w+W, ... of the full symmetric grid (complete sums, no cross-wave
  // reduction); border columns on the VALU by wave 0 after a barrier.
  auto hx_full = [&](int oXV) __attribute__((always_inline)) {
#pragma unroll
    for (int ci = 0; ci < CW; ++ci) {
      const int Jt = w + W * ci;
      if (Jt < T) {
        double acc_ = 0.0;
        for (int I = 0; I < T; ++I) {
#pragma unroll
          for (int p = 0; p < 4; ++p) acc_ = fma(Hw[((size_t)(I * T + Jt) * 4 + p) * 64 + lane], slds[oXV + 16 * I + q + 4 * p], acc_);
        }
        acc_ = q_sum(acc_);
        if (q == 0) slds[VEC(V_HX) + 16 * Jt + c] = acc_;
      }
    }
    if (NB > 0) {
      __syncthreads();
      if (w == 0) {
        double xb[NBB], sb[NBB];
#pragma unroll
        for (int e = 0; e < NB; ++e) { xb[e] = slds[oXV + nc + e]; sb[e] = 0.0; }
        for (int i = lane; i < n; i += 64) {
          double add = 0.0;
#pragma unroll
          for (int e = 0; e < NB; ++e) { const double hbi = Hb[(size_t)e * np + i]; add = fma(hbi, xb[e], add); sb[e] = fma(hbi, slds[oXV + i], sb[e]); }
          if (i < nc) slds[VEC(V_HX) + i] += add;
        }
#pragma unroll
        for (int e = 0; e < NB; ++e) { const double tot = wave_sum(sb[e]); if (lane == 0) slds[VEC(V_HX) + nc + e] = tot; }
      }
    }
    __syncthreads();
  };

  // ---- load n-vectors, initial x = clamp(0, l, u) (scaled), count finite sides ----
  for (int i = tid; i < np; i += NTH) { G_(i) = gw[i]; EV_(i) = Es[i]; R1_(i) = 0; R2_(i) = 0; DX_(i) = 0; X_(i) = 0; }
  __syncthreads();
  int cnt_local = 0, infeas_l = 0;
  for (int js = w; js < JT; js += W) {
    const int ix = js * 64 + lane;
    const bool valid = row_valid(js);
    double l = aL[ix], u = aU[ix];
    if (valid) {
      if (l > -INFINITY && u < INFINITY) {
        if (l > u) infeas_l = 1;
        if (!(u > l)) {  // equality row: open a tiny interior (documented relaxation)
          const double eps = 1e-9 * fmax(1.0, fabs(l));
          l -= eps; u += eps; aL[ix] = l; aU[ix] = u;
        }
      }
      cnt_local += (l > -INFINITY) + (u < INFINITY);
    }
    if (js >= J) {
      const int i = (js - J) * 64 + lane;
      if (i < np) {
        double xi = 0.0;
        if (valid) { if (l > -INFINITY && xi < l) xi = l; if (u < INFINITY && xi > u) xi = u; }
        X_(i) = xi;
      }
    }
  }
  red_put(0, wave_sum((double)cnt_local)); red_put(1, wave_max((double)infeas_l));
  red_sync();
  const double cnt = fmax(1.0, red_sum(0));
  const int infeas = red_max(1) > 0;
  red_next();

  int flag = 1, it = 0, flag_polished = 0;
  double fval_s = 0.0;
  if (infeas) flag = -2;
  STAMP_DECL

  // ---- v = G x ----
  {
    pass_Av(VEC(V_X), aV);
    for (int js = J + w; js < JT; js += W) { const int i = (js - J) * 64 + lane; aV[js * 64 + lane] = i < n ? X_(i) : 0.0; }
  }
  __syncthreads();
  // ---- initial slacks / multipliers in the equilibrated problem ----
  const double T0 = 10.0, Z0 = 100.0;
  for (int js = w; js < JT; js += W) {
    const int ix = js * 64 + lane;
    const bool valid = row_valid(js);
    const double l = aL[ix], u = aU[ix], v = aV[ix];
    const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
    aTL[ix] = hl ? fmax(v - l, T0) : 1.0;
    aTU[ix] = hu ? fmax(u - v, T0) : 1.0;
    aZL[ix] = hl ? Z0 : 0.0;
    aZU[ix] = hu ? Z0 : 0.0;
    aW3[ix] = (js < J) ? ((hl ? Z0 : 0.0) - (hu ? Z0 : 0.0)) : 0.0;
  }
  __syncthreads();
  // A~'w for a row array (owner layout), split by trips -> partial sums WP[which]
  auto pass_Atw = [&](const double* Wr, int which) __attribute__((always_inline)) {
    double p[T], pbv[NBB];
#pragma unroll
    for (int t = 0; t < T; ++t) p[t] = 0.0;
#pragma unroll
    for (int f = 0; f < NBB; ++f) pbv[f] = 0.0;
    TRIP_LOOP_BEGIN
      const v2d wv = *reinterpret_cast<const v2d*>(Wr + rix);
      v2d ab[NBB];
#pragma unroll
      for (int f = 0; f < NB; ++f) ab[f] = *reinterpret_cast<const v2d*>(Ab + (size_t)f * JS + rix);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int t = 0; t < T; ++t) p[t] = fma(wv[h], bq[t][h], p[t]);
#pragma unroll
        for (int f = 0; f < NB; ++f) pbv[f] = fma(wv[h], ab[f][h], pbv[f]);
      }
    TRIP_LOOP_END
#pragma unroll
    for (int t = 0; t < T; ++t) { const double v = q_sum(p[t]); if (q == 0) slds[oWP + (which * W + w) * np + 16 * t + c] = v; }
#pragma unroll
    for (int f = 0; f < NB; ++f) { const double v = q_sum(pbv[f]); if (lane == 0) slds[oWP + (which * W + w) * np + nc + f] = v; }
    if (NB > 0 && lane >= NB && lane < 16) slds[oWP + (which * W + w) * np + nc + lane] = 0.0;
    __syncthreads();
  };
  // bound multipliers absorb the initial dual residual r = Hx + g - A'(zl - zu)
  {
    hx_full(VEC(V_X));
    pass_Atw(aW3, 0);
    for (int i = tid; i < n; i += NTH) {
      const int ix = ixv(i);
      const double r = HX_(i) + G_(i) - part(0, i);
      if (aL[ix] > -INFINITY) aZL[ix] = fmax(r, 0.0) + Z0;
      if (aU[ix] < INFINITY) aZU[ix] = fmax(-r, 0.0) + Z0;
    }
    __syncthreads();
  }

  // fall-back iterate (best one that met tol_loose)
  double saved_merit = INFINITY, best_res = INFINITY;
  int have_saved = 0, stall = 0;
  double* __restrict__ XS = ws + d.off_save;            // np
  double* __restrict__ LAMS = ws + d.off_save + np;     // rowlen

  auto row1_body = [&](int ix, bool valid, double l, double u, double v, double tl, double tu, double zl, double zu,
                       double& s_gap, double& m_rp) __attribute__((always_inline)) {
    const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
    const double rpl = hl ? v - l - tl : 0.0, rpu = hu ? u - v - tu : 0.0;
    const double dl_ = hl ? zl / tl : 0.0, du_ = hu ? zu / tu : 0.0;
    aRPL[ix] = rpl; aRPU[ix] = rpu;
    aCB1[ix] = dl_; aCC1[ix] = hl ? dl_ / tl : 0.0;
    aCB2[ix] = du_; aCC2[ix] = hu ? du_ / tu : 0.0;
    aD[ix] = dl_ + du_;
    aW1[ix] = -dl_ * rpl + du_ * rpu;                              // affine rhs weight
    aW2[ix] = (hl ? 1.0 / tl : 0.0) - (hu ? 1.0 / tu : 0.0);      // centering weight (times sigma*mu)
    aW3[ix] = (hl ? zl : 0.0) - (hu ? zu : 0.0);                  // current multiplier (for the dual residual)
    s_gap += (hl ? tl * zl : 0.0) + (hu ? tu * zu : 0.0);
    const double sc = fmax(1.0, fabs(v));
    if (hl) m_rp = fmax(m_rp, fabs(rpl) / fmax(sc, fabs(l)));
    if (hu) m_rp = fmax(m_rp, fabs(rpu) / fmax(sc, fabs(u)));
  };
  double gap = 0.0, rp_rel = 0.0;   // carried across iterations (produced by the update sweep)

  v4d acc[NTW];          // my tiles of M, then of its Cholesky factor U
  double Ubb[NBB][NBB];  // Cholesky factor of the border Schur complement (wave-uniform scalars, every wave has them)

  // right-hand-side tile row K of NS LDS vectors (vec offsets vo[e]): B-operand form, column e of the tile = vector e
  auto rhs_load = [&](int K, int vo, int ns) __attribute__((always_inline)) -> v4d {
    v4d r = {0.0, 0.0, 0.0, 0.0};
    if (c < ns) {
      const int o = vo + c * np + 16 * K + q;
#pragma unroll
      for (int p = 0; p < 4; ++p) r[p] = slds[o + 4 * p];
    }
    return r;
  };
  auto rhs_store = [&](int K, int vo, int ns, const v4d& r) __attribute__((always_inline)) {
    if (c < ns) {
      const int o = vo + c * np + 16 * K + q;
#pragma unroll
      for (int p = 0; p < 4; ++p) slds[o + 4 * p] = r[p];
    }
  };
  auto rhs_sub = [&](int K, int vo, int ns, const v4d& r) __attribute__((always_inline)) {   // the only writer of row K at this step
    if (c < ns) {
      const int o = vo + c * np + 16 * K + q;
#pragma unroll
      for (int p = 0; p < 4; ++p) slds[o + 4 * p] -= r[p];
    }
  };
  auto tile_store17 = [&](int o, const v4d& Xt) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) slds[o + (q + 4 * p) * 17 + c] = Xt[p];
  };
  auto tile_load17 = [&](int o) __attribute__((always_inline)) -> v4d {
    v4d Z;
#pragma unroll
    for (int p = 0; p < 4; ++p) Z[p] = slds[o + (q + 4 * p) * 17 + c];
    return Z;
  };
  auto tile_load17_t = [&](int o) __attribute__((always_inline)) -> v4d {
    v4d Z;
#pragma unroll
    for (int p = 0; p < 4; ++p) Z[p] = slds[o + c * 17 + q + 4 * p];
    return Z;
  };
  auto img_store = [&](int o, const v4d& Xt) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) slds[o + p * 64 + lane] = Xt[p];
  };
  auto img_load = [&](int o) __attribute__((always_inline)) -> v4d {
    v4d Z;
#pragma unroll
    for (int p = 0; p < 4; ++p) Z[p] = slds[o + p * 64 + lane];
    return Z;
  };

  // forward solve U'y = b for `ns` LDS vectors with the resident factor (in place)
  auto fwd_solve = [&](int vo, int ns) __attribute__((always_inline)) {
#pragma unroll
    for (int K = 0; K < T; ++K) {
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (tI[t] == K && tJ[t] == K) {
          const v4d rk = rhs_load(K, vo, ns);
          const v4d yk = mfma4_new(tile_load17_t(oYL + K * 272), rk);     // U_KK^-T b_K
          rhs_store(K, vo, ns, yk);
        }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (tI[t] == K && tJ[t] > K && tJ[t] < T) {
          const v4d yk = rhs_load(K, vo, ns);
          const v4d z = mfma4_new(acc[t], yk);                            // U_KJ' y_K
          rhs_sub(tJ[t], vo, ns, z);
        }
      __syncthreads();
    }
  };
  // backward solve U x = y in place: the coupling U_IK x_K is formed on the VALU (no tile transposes)
  auto bwd_solve = [&](int vo, int ns) __attribute__((always_inline)) {
#pragma unroll
    for (int K = T - 1; K >= 0; --K) {
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (tI[t] == K && tJ[t] == K) {
          const v4d rk = rhs_load(K, vo, ns);
          const v4d xk = mfma4_new(tile_load17(oYL + K * 272), rk);       // (U_KK^-T)' y_K = U_KK^-1 y_K
          rhs_store(K, vo, ns, xk);
        }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (tJ[t] == K && tI[t] < K) {
          for (int e = 0; e < ns; ++e) {
            const double xc = slds[vo + e * np + 16 * K + c];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const double sm = grp16_sum(acc[t][p] * xc);                // row q+4p of U_IK times x_K
              if (c == 0) slds[vo + e * np + 16 * tI[t] + q + 4 * p] -= sm;
            }
          }
        }
      __syncthreads();
    }
  };
  // border part of a solve (wave 0 writes; every wave would compute the same numbers): R holds y_c = U^-T b_c (core)
  // and b_b (border); leaves the border solution in R[nc+e] and y_c - sum_e u_e x_e in the core
  auto border_solve = [&](int oR) __attribute__((always_inline)) {
    if (NB > 0) {
      if (w == 0) {
        double yb[NBB], xb[NBB];
#pragma unroll
        for (int e = 0; e < NB; ++e) {
          double dsum = 0.0;
          for (int i = lane; i < nc; i += 64) dsum = fma(MB_(e, i), slds[oR + i], dsum);
          double tt = slds[oR + nc + e] - wave_sum(dsum);
#pragma unroll
          for (int g2 = 0; g2 < e; ++g2) tt -= Ubb[g2][e] * yb[g2];
          yb[e] = tt / Ubb[e][e];
        }
#pragma unroll
        for (int e = NB - 1; e >= 0; --e) {
          double tt = yb[e];
#pragma unroll
          for (int f = e + 1; f < NB; ++f) tt -= Ubb[e][f] * xb[f];
          xb[e] = tt / Ubb[e][e];
        }
        for (int i = lane; i < nc; i += 64) {
          double r = slds[oR + i];
#pragma unroll
          for (int e = 0; e < NB; ++e) r = fma(-MB_(e, i), xb[e], r);
          slds[oR + i] = r;
        }
        if (lane == 0) {
#pragma unroll
          for (int e = 0; e < NB; ++e) slds[oR + nc + e] = xb[e];
        }
      }
      __syncthreads();
    }
  };

  // M = (acc from pass 1) + diag(dadd on the variable rows), border columns = H~ border + A'DA border (MB); blocked
  // right-looking Cholesky over the distributed tiles with the right-hand sides R1, R2 (and the NB border columns)
  // riding along, then the border Schur complement and the backward solve for R1, R2.  Returns 1 on a non-finite pivot.
  // `dsrc`: row array holding the diagonal weight of the variable-bound rows.
  auto factor_solve2 = [&](const double* dsrc) __attribute__((always_inline)) -> int {
    double dmax_l = 0;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
      if (tI[t] == tJ[t] && tI[t] < T) {
        const int i = 16 * tI[t] + c;                     // diagonal element lives on lane c with q = c&3, reg c>>2
        const double dadd = i < n ? dsrc[ixv(i)] : 1.0;   // padded indices get a unit diagonal
        const bool mine = (q == (c & 3));
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (mine && p == (c >> 2)) { acc[t][p] += dadd; dmax_l = fmax(dmax_l, acc[t][p]); }
      }
    if (NB > 0) {
      for (int i = tid; i < np; i += NTH) {
#pragma unroll
        for (int e = 0; e < NB; ++e) {   // border column e of M: H~ column + A'DA column (+ its variable-bound weight on the diagonal)
          double v = i < n ? Hb[(size_t)e * np + i] + MB_(e, i) : 0.0;
          if (i == nc + e) { v += e < nb ? dsrc[ixv(i)] : 1.0; dmax_l = fmax(dmax_l, v); }
          MB_(e, i) = v;
        }
      }
    }
    red_put(0, wave_max(dmax_l));
    red_sync();
    const double dmax = red_max(0);
    red_next();
    const double floor_abs = 1e-30 * dmax;
    STAMP(5);
    const int vo = VEC(V_R1);          // right-hand-side columns: R1, R2, MB[0..NB-1] (contiguous vectors)
    constexpr int NS = 2 + NB;
    int fbad = 0;
#pragma unroll
    for (int K = 0; K < T; ++K) {
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (tI[t] == K && tJ[t] == K) {
          v4d Yk;
#pragma unroll
          for (int p = 0; p < 4; ++p) Yk[p] = (q + 4 * p == c) ? 1.0 : 0.0;
          fbad |= diag_factor(c, q, acc[t], Yk, floor_abs);
          tile_store17(oYL + K * 272, Yk);               // U_KK^-T stays in LDS for the solves of this iteration
        }
      __syncthreads();
      {   // y_K = U_KK^-T b_K by the wave that owns the diagonal tile (after the store above is visible to itself: same wave)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
          if (tI[t] == K && tJ[t] == K) {
            const v4d rk = rhs_load(K, vo, NS);
            const v4d yk = mfma4_new(tile_load17_t(oYL + K * 272), rk);
            rhs_store(K, vo, NS, yk);
          }
#pragma unroll
        for (int t = 0; t < NTW; ++t)
          if (tI[t] == K && tJ[t] > K && tJ[t] < T) {
            const v4d Wk = tile_load17_t(oYL + K * 272);           // U_KK^-1 as the A operand acts as U_KK^-T
            acc[t] = mfma4_new(Wk, acc[t]);                         // U_KJ = U_KK^-T M_KJ
            img_store(oPB + tJ[t] * 256, acc[t]);
          }
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        if (tI[t] == K && tJ[t] > K && tJ[t] < T) {
          const v4d yk = rhs_load(K, vo, NS);
          const v4d z = mfma4_new(acc[t], yk);                     // U_KJ' y_K
          rhs_sub(tJ[t], vo, NS, z);
        }
        if (tI[t] > K && tI[t] < T) {
          const v4d UKI = img_load(oPB + tI[t] * 256);
          const v4d UKJ = img_load(oPB + tJ[t] * 256);
          mfma4_sub(UKI, UKJ, acc[t]);                             // M_IJ -= U_KI' U_KJ
        }
      }
      __syncthreads();
    }
    STAMP(6);
    red_put(0, (double)fbad);
    red_sync();
    fbad = red_max(0) > 0;
    red_next();
    if (NB > 0) {   // bordered factor: u_e = U^-T m_e came out of the forward sweep; S = M_bb - u'u is factorised as scalars
      double S[NBB][NBB];
#pragma unroll
      for (int e = 0; e < NB; ++e)
#pragma unroll
        for (int f = e; f < NB; ++f) {
          double dsum = 0.0;
          for (int i = lane; i < nc; i += 64) dsum = fma(MB_(e, i), MB_(f, i), dsum);
          S[e][f] = MB_(e, nc + f) - wave_sum(dsum);
        }
#pragma unroll
      for (int e = 0; e < NB; ++e) {
        double dd = S[e][e];
#pragma unroll
        for (int g2 = 0; g2 < e; ++g2) dd -= Ubb[g2][e] * Ubb[g2][e];
        if (!(dd > floor_abs)) { if (!(fabs(dd) < INFINITY)) fbad = 1; dd = floor_abs; }
        Ubb[e][e] = sqrt(dd);
#pragma unroll
        for (int f = e + 1; f < NB; ++f) {
          double tt = S[e][f];
#pragma unroll
          for (int g2 = 0; g2 < e; ++g2) tt -= Ubb[g2][e] * Ubb[g2][f];
          Ubb[e][f] = tt / Ubb[e][e];
        }
      }
    }
    if (fbad) return 1;
    border_solve(VEC(V_R1)); border_solve(VEC(V_R2));
    bwd_solve(vo, 2);
    return 0;
  };
  // one more solve with the resident factor: V <- M^-1 V (LDS n-vector, in place)
  auto solve1 = [&](int oV) __attribute__((always_inline)) {
    fwd_solve(oV, 1);
    border_solve(oV);
    bwd_solve(oV, 1);
  };

  // pass 1: acc += A~' D A~ on the matrix cores (my tiles); A~'w1..w3 and the border products for my column tiles on
  // the VALU beside them.  Writes P1..P3 and MB (border column / block of A'DA) directly (one owner per entry).
  auto pass_syrk = [&](const double* Dr, const double* W1r, const double* W2r, const double* W3r) __attribute__((always_inline)) {
    double p1[CW], p2[CW], p3[CW], pb[NBB][CW], sbb[NBB][NBB], pwb[3][NBB];
#pragma unroll
    for (int ci = 0; ci < CW; ++ci) { p1[ci] = p2[ci] = p3[ci] = 0.0; for (int e = 0; e < NBB; ++e) pb[e][ci] = 0.0; }
#pragma unroll
    for (int e = 0; e < NBB; ++e) { for (int f = 0; f < NBB; ++f) sbb[e][f] = 0.0; pwb[0][e] = pwb[1][e] = pwb[2][e] = 0.0; }
    for (int tr = 0; tr < ntr; ++tr) {
      const int tc = tcs[tr], rbase = aoff[tr];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s0 = 4 * tr + 2 * u;
        const int rix = (s0 >> 4) * 64 + q * 16 + (s0 & 15);
        const v2d dd = *reinterpret_cast<const v2d*>(Dr + rix);
        const int rb = rbase + u * tc;
#pragma unroll
        for (int t = 0; t < NTW; ++t)
          if (tJ[t] < tc) {
            const v2d bi = opnd(rb + tI[t]), bj = opnd(rb + tJ[t]);
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(dd[h] * bi[h], bj[h], acc[t], 0, 0, 0);
          }
        // VALU side products for my column tiles (and the border scalars on the wave that owns column tile 0)
        const v2d w1 = *reinterpret_cast<const v2d*>(W1r + rix), w2 = *reinterpret_cast<const v2d*>(W2r + rix), w3 = *reinterpret_cast<const v2d*>(W3r + rix);
        v2d ab[NBB];
#pragma unroll
        for (int e = 0; e < NB; ++e) ab[e] = *reinterpret_cast<const v2d*>(Ab + (size_t)e * JS + rix);
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) {
          const int ct = w + W * ci;
          if (ct < tc) {
            const v2d bc = opnd(rb + ct);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              p1[ci] = fma(w1[h], bc[h], p1[ci]); p2[ci] = fma(w2[h], bc[h], p2[ci]); p3[ci] = fma(w3[h], bc[h], p3[ci]);
#pragma unroll
              for (int e = 0; e < NB; ++e) pb[e][ci] = fma(dd[h] * ab[e][h], bc[h], pb[e][ci]);
            }
          }
        }
        if (NB > 0 && w == 0) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < NB; ++e) {
              const double dab = dd[h] * ab[e][h];
#pragma unroll
              for (int f = e; f < NB; ++f) sbb[e][f] = fma(dab, ab[f][h], sbb[e][f]);
              pwb[0][e] = fma(w1[h], ab[e][h], pwb[0][e]); pwb[1][e] = fma(w2[h], ab[e][h], pwb[1][e]); pwb[2][e] = fma(w3[h], ab[e][h], pwb[2][e]);
            }
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < CW; ++ci) {
      const int ct = w + W * ci;
      if (ct < T) {
        const double v1 = q_sum(p1[ci]), v2 = q_sum(p2[ci]), v3 = q_sum(p3[ci]);
        if (q == 0) { P1_(16 * ct + c) = v1; P2_(16 * ct + c) = v2; P3_(16 * ct + c) = v3; }
#pragma unroll
        for (int e = 0; e < NB; ++e) { const double vb = q_sum(pb[e][ci]); if (q == 0) MB_(e, 16 * ct + c) = vb; }
      }
    }
    if (NB > 0 && w == 0) {
      // border scalars are identical on the 16 lanes of a group: sum the four groups, lane 0 writes
#pragma unroll
      for (int e = 0; e < NB; ++e) {
        const double v1 = q_sum(pwb[0][e]), v2 = q_sum(pwb[1][e]), v3 = q_sum(pwb[2][e]);
        if (lane == 0) { P1_(nc + e) = v1; P2_(nc + e) = v2; P3_(nc + e) = v3; }
#pragma unroll
        for (int f = 0; f < NB; ++f) {
          const double sv = q_sum(f >= e ? sbb[e][f] : sbb[f][e]);
          if (lane == 0) MB_(e, nc + f) = sv;
        }
      }
    }
    __syncthreads();
  };
  auto acc_init = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      v4d h = {0.0, 0.0, 0.0, 0.0};
      if (tJ[t] < T) {
        const double* hp = Hw + ((size_t)(tI[t] * T + tJ[t]) * 4) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 4; ++p) h[p] = hp[p * 64];
      }
      acc[t] = h;
    }
  };

  STAMP(0);
  for (it = 0; flag == 1; ++it) {
    // ================= row phase 1: residuals, weights (only on entry; afterwards fused into the update sweep) =================
    if (it == 0) {
      double s_gap = 0, m_rp = 0;
      for (int js = w; js < JT; js += W) {
        const int ix = js * 64 + lane;
        row1_body(ix, row_valid(js), aL[ix], aU[ix], aV[ix], aTL[ix], aTU[ix], aZL[ix], aZU[ix], s_gap, m_rp);
      }
      red_put(0, wave_sum(s_gap)); red_put(1, wave_max(m_rp));
      red_sync();
      gap = red_sum(0); rp_rel = red_max(1);
      red_next();
    }
    const double mu = gap / cnt;
    STAMP(1);

    // ================= pass 1: M = H + A'DA (MFMA), p1, p2, p3; Hx =================
    hx_full(VEC(V_X));
    STAMP(2);
    acc_init();
    pass_syrk(aD, aW1, aW2, aW3);
    STAMP(3);
    // objective, dual residual (every wave computes the same numbers from LDS)
    double fl = 0, m_rd = 0;
    for (int i = lane; i < n; i += 64) {
      const double gz = P3_(i) + aW3[ixv(i)];
      fl += 0.5 * X_(i) * HX_(i) + G_(i) * X_(i);
      const double sc = fmax(1.0, fmax(fabs(G_(i)), fmax(fabs(HX_(i)), fabs(gz))));
      m_rd = fmax(m_rd, fabs(HX_(i) + G_(i) - gz) / sc);
    }
    const double fval = wave_sum(fl);
    const double rd_rel = wave_max(m_rd);
    const double gap_rel = gap / fmax(1.0, fabs(fval));
    const double merit = fmax(rd_rel, fmax(rp_rel, gap_rel));
    fval_s = fval;
    const bool res_ok = merit <= P.tol;
    if (!(merit < INFINITY)) { flag = have_saved ? 2 : -1; break; }
    if (merit <= P.tol_loose && merit < saved_merit) {
      for (int i = tid; i < np; i += NTH) XS[i] = X_(i);
      for (int js = w; js < JT; js += W) LAMS[js * 64 + lane] = aW3[js * 64 + lane];
      have_saved = 1; saved_merit = merit;
    } else if (merit > P.tol_loose && rp_rel <= P.tol_loose && gap_rel <= P.tol_loose) {
      // Only the dual residual is in the way: repair the certificate of a *copy* of the iterate by moving r_d into the
      // bound multipliers, where a finite bound of the right sign exists.
      double dgap = 0, m_rd2 = 0;
      for (int i = lane; i < n; i += 64) {
        const int ix = ixv(i);
        const double lam = aW3[ix], gz = P3_(i) + lam, r = HX_(i) + G_(i) - gz;
        const double lam2 = lam + r, l = aL[ix], u = aU[ix], v = aV[ix];
        const bool ok = lam2 >= 0 ? l > -INFINITY : u < INFINITY;
        if (ok) dgap += fabs(r) * fmax(0.0, lam2 >= 0 ? v - l : u - v);
        else {
          const double sc = fmax(1.0, fmax(fabs(G_(i)), fmax(fabs(HX_(i)), fabs(gz))));
          m_rd2 = fmax(m_rd2, fabs(r) / sc);
        }
      }
      const double merit2 = fmax(wave_max(m_rd2), fmax(rp_rel, (gap + wave_sum(dgap)) / fmax(1.0, fabs(fval))));
      if (merit2 <= P.tol_loose && merit2 < saved_merit) {
        for (int i = tid; i < np; i += NTH) XS[i] = X_(i);
        for (int js = w; js < J; js += W) LAMS[js * 64 + lane] = aW3[js * 64 + lane];
        for (int i = tid; i < np; i += NTH) {
          double lf = 0.0;
          if (i < n) {
            const int ix = ixv(i);
            const double lam = aW3[ix], r = HX_(i) + G_(i) - (P3_(i) + lam), lam2 = lam + r;
            const bool ok = lam2 >= 0 ? aL[ix] > -INFINITY : aU[ix] < INFINITY;
            lf = ok ? lam2 : lam;
          }
          LAMS[ixv(i)] = lf;
        }
        have_saved = 1; saved_merit = merit2;
      }
      if (have_saved) { flag = 2; break; }
    } else if (have_saved && merit > P.tol_loose) { flag = 2; break; }
    if (merit < 0.9 * best_res) { best_res = merit; stall = 0; } else ++stall;

    // ================= factorise with the affine / centering right-hand sides riding along =================
    for (int i = tid; i < np; i += NTH) {
      const int ix = ixv(i);
      R1_(i) = i < n ? -(HX_(i) + G_(i)) + P1_(i) + aW1[ix] : 0.0;
      R2_(i) = i < n ? P2_(i) + aW2[ix] : 0.0;
    }
    __syncthreads();
#ifdef QP_DEBUG_DUMP
    if (P.dump && b == 0 && P.dump_stage == 1 && it == P.dump_iter) {  // debug: M (before the diagonal add of variable rows), p1, p2, p3, Hx
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (tJ[t] < T)
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int r = 16 * tI[t] + q + 4 * p, cc = 16 * tJ[t] + c;
            double v = acc[t][p];
            if (r == cc && r < n) v += aD[ixv(r)];
            if (r < n && cc < n) { P.dump[r * n + cc] = v; if (tI[t] != tJ[t] || cc >= r) P.dump[cc * n + r] = v; }
          }
      if (w == 0) {
        for (int e = 0; e < nb; ++e)
          for (int i = lane; i < n; i += 64) {
            double v = Hb[(size_t)e * np + i] + MB_(e, i);
            if (i == nc + e) v += aD[ixv(i)];
            P.dump[i * n + nc + e] = v; P.dump[(nc + e) * n + i] = v;
          }
        for (int i = lane; i < n; i += 64) { P.dump[n * n + i] = P1_(i); P.dump[n * n + n + i] = P2_(i); P.dump[n * n + 2 * n + i] = P3_(i); P.dump[n * n + 3 * n + i] = HX_(i); }
      }
      __syncthreads();
    }
#endif
    STAMP(4);
    if (factor_solve2(aD)) {
      flag = (res_ok || have_saved) ? 2 : -1;
      if (flag == -1 && P.polish && rp_rel <= P.tol_loose && gap_rel <= P.tol_loose) flag = 4;
      break;
    }
    STAMP(7);
    if (res_ok) {  // Newton-decrement test in the caller's coordinates
      double dm = 0, xm = 1.0;
      for (int i = lane; i < n; i += 64) { dm = fmax(dm, fabs(R1_(i) * EV_(i))); xm = fmax(xm, fabs(X_(i) * EV_(i))); }
      dm = wave_max(dm); xm = wave_max(xm);
      if (dm <= P.tol_x * xm) { flag = 0; break; }
    }
    if (it >= P.max_iter) { flag = have_saved ? 2 : 1; break; }

    // ================= pass 2: va = G dxa, vc = G dxc; fused: WP[0] = A~' w_cor =================
    {
      double v[2][T], vb[2][NBB], pc[T], pcb[NBB];
#pragma unroll
      for (int t = 0; t < T; ++t) { v[0][t] = R1_(16 * t + c); v[1][t] = R2_(16 * t + c); pc[t] = 0.0; }
#pragma unroll
      for (int f = 0; f < NBB; ++f) { vb[0][f] = NB ? R1_(nc + f) : 0.0; vb[1][f] = NB ? R2_(nc + f) : 0.0; pcb[f] = 0.0; }
      TRIP_LOOP_BEGIN
        const v2d rpl = *reinterpret_cast<const v2d*>(aRPL + rix), cb1 = *reinterpret_cast<const v2d*>(aCB1 + rix), cc1 = *reinterpret_cast<const v2d*>(aCC1 + rix);
        const v2d rpu = *reinterpret_cast<const v2d*>(aRPU + rix), cb2 = *reinterpret_cast<const v2d*>(aCB2 + rix), cc2 = *reinterpret_cast<const v2d*>(aCC2 + rix);
        v2d ab[NBB];
#pragma unroll
        for (int f = 0; f < NB; ++f) ab[f] = *reinterpret_cast<const v2d*>(Ab + (size_t)f * JS + rix);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double ds0 = 0.0, ds1 = 0.0;
#pragma unroll
          for (int t = 0; t < T; ++t) { ds0 = fma(bq[t][h], v[0][t], ds0); ds1 = fma(bq[t][h], v[1][t], ds1); }
          ds0 = grp16_sum(ds0); ds1 = grp16_sum(ds1);
#pragma unroll
          for (int f = 0; f < NB; ++f) { ds0 = fma(ab[f][h], vb[0][f], ds0); ds1 = fma(ab[f][h], vb[1][f], ds1); }
          if (c == ((s0 + h) & 15)) { aVA[rix + h] = ds0; aVC[rix + h] = ds1; }
          const double dl_ = ds0 + rpl[h], du_ = rpu[h] - ds0;
          const double wc = dl_ * fma(cc1[h], dl_, cb1[h]) - du_ * fma(cc2[h], du_, cb2[h]);
#pragma unroll
          for (int t = 0; t < T; ++t) pc[t] = fma(wc, bq[t][h], pc[t]);
#pragma unroll
          for (int f = 0; f < NB; ++f) pcb[f] = fma(wc, ab[f][h], pcb[f]);
        }
      TRIP_LOOP_END
#pragma unroll
      for (int t = 0; t < T; ++t) { const double pv = q_sum(pc[t]); if (q == 0) slds[oWP + w * np + 16 * t + c] = pv; }
#pragma unroll
      for (int f = 0; f < NB; ++f) { const double pv = q_sum(pcb[f]); if (lane == 0) slds[oWP + w * np + nc + f] = pv; }
      for (int js = J + w; js < JT; js += W) {
        const int i = (js - J) * 64 + lane;
        aVA[js * 64 + lane] = i < n ? R1_(i) : 0.0;
        aVC[js * 64 + lane] = i < n ? R2_(i) : 0.0;
      }
    }
    __syncthreads();
    STAMP(8);
    // ================= row phase 2: affine step length, sigma, corrector weights (one sweep) =================
    double a_aff = 1.0, s1 = 0.0, s2 = 0.0;
    for (int js = w; js < JT; js += W) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(js);
      const double l = aL[ix], u = aU[ix], tl = aTL[ix], tu = aTU[ix], zl = aZL[ix], zu = aZU[ix], va = aVA[ix], rpl = aRPL[ix], rpu = aRPU[ix];
      const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
      double wv = 0.0;
      if (hl) {
        const double dt = va + rpl, dz = -zl - (zl / tl) * dt;
        if (dt < 0) a_aff = fmin(a_aff, -tl / dt);
        if (dz < 0) a_aff = fmin(a_aff, -zl / dz);
        s1 += tl * dz + zl * dt; s2 += dt * dz;
        wv -= dt * dz / tl;
      }
      if (hu) {
        const double dt = -va + rpu, dz = -zu - (zu / tu) * dt;
        if (dt < 0) a_aff = fmin(a_aff, -tu / dt);
        if (dz < 0) a_aff = fmin(a_aff, -zu / dz);
        s1 += tu * dz + zu * dt; s2 += dt * dz;
        wv += dt * dz / tu;
      }
      if (js >= J) aW1[ix] = wv;   // second-order weight of the variable-bound rows (A rows: fused in pass 2)
    }
    red_put(0, wave_min(a_aff)); red_put(1, wave_sum(s1)); red_put(2, wave_sum(s2));
    red_sync();
    a_aff = red_min(0); s1 = red_sum(1); s2 = red_sum(2);
    red_next();
    const double mu_aff = fmax(0.0, gap + a_aff * (s1 + a_aff * s2)) / cnt;
    double sigma = mu > 0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
    if (sigma > 1.0) sigma = 1.0;
    {
      const double mu_floor = 1e-5 * P.tol * fmax(1.0, fabs(fval)) / cnt;
      if (mu > 0 && sigma < mu_floor / mu) sigma = fmin(1.0, mu_floor / mu);
    }
    const double smu = sigma * mu;
    const double cw = a_aff >= 0.05 ? 1.0 : 0.0;   // second-order term dropped when the affine step is tiny
    STAMP(9);
    if (cw != 0.0) {
      // ================= corrector: A' w_cor came out of the fused pass 2 =================
      for (int i = tid; i < np; i += NTH) DX_(i) = i < n ? part(0, i) + aW1[ixv(i)] : 0.0;
      __syncthreads();
      solve1(VEC(V_DX));
      STAMP(10);
      // ================= pass 3: G dx_cor =================
      {
        pass_Av(VEC(V_DX), aW2);   // W2 reused for G dx_cor
        for (int js = J + w; js < JT; js += W) { const int i = (js - J) * 64 + lane; aW2[js * 64 + lane] = i < n ? DX_(i) : 0.0; }
      }
    } else {   // no corrector this iteration
      for (int i = tid; i < np; i += NTH) DX_(i) = 0.0;
      for (int js = w; js < JT; js += W) aW2[js * 64 + lane] = 0.0;
    }
    __syncthreads();
    STAMP(11);
    // full direction dx = dxa + smu*dxc + dxcor ; dv likewise
    for (int i = tid; i < n; i += NTH) DX_(i) = R1_(i) + smu * R2_(i) + DX_(i);
    // ================= row phase 3: step length (Mehrotra heuristic on the blocking pair), update =================
    double amax = 1e300, bp = 0, bdp = 0, bd = 0, bdd = 0, q1 = 0.0, q2 = 0.0;
    for (int js = w; js < JT; js += W) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(js);
      const double l = aL[ix], u = aU[ix], tl = aTL[ix], tu = aTU[ix], zl = aZL[ix], zu = aZU[ix], va = aVA[ix], vc = aVC[ix], w2 = aW2[ix], rpl = aRPL[ix], rpu = aRPU[ix];
      const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
      const double dv = va + smu * vc + w2;
      aVC[ix] = dv;  // keep the full G dx for the update
      if (hl) {
        const double dta = va + rpl, dza = -zl - (zl / tl) * dta;
        const double cl = smu - cw * dta * dza;
        const double dt = dv + rpl, dz = -zl + cl / tl - (zl / tl) * dt;
        if (dt < 0 && -tl / dt < amax) { amax = -tl / dt; bp = tl; bdp = dt; bd = zl; bdd = dz; }
        if (dz < 0 && -zl / dz < amax) { amax = -zl / dz; bp = zl; bdp = dz; bd = tl; bdd = dt; }
        q1 += tl * dz + zl * dt; q2 += dt * dz;
      }
      if (hu) {
        const double dta = -va + rpu, dza = -zu - (zu / tu) * dta;
        const double cu = smu - cw * dta * dza;
        const double dt = -dv + rpu, dz = -zu + cu / tu - (zu / tu) * dt;
        if (dt < 0 && -tu / dt < amax) { amax = -tu / dt; bp = tu; bdp = dt; bd = zu; bdd = dz; }
        if (dz < 0 && -zu / dz < amax) { amax = -zu / dz; bp = zu; bdp = dz; bd = tu; bdd = dt; }
        q1 += tu * dz + zu * dt; q2 += dt * dz;
      }
    }
    {
      const double amax_w = wave_min(amax);
      if (amax_w < 1e299) {
        const unsigned long long msk = __ballot(amax == amax_w);
        const int src = __ffsll((long long)msk) - 1;
        bp = rl(bp, src); bdp = rl(bdp, src); bd = rl(bd, src); bdd = rl(bdd, src);
      }
      red_put(0, amax_w); red_put(1, bp); red_put(2, bdp); red_put(3, bd); red_put(4, bdd); red_put(5, wave_sum(q1)); red_put(6, wave_sum(q2));
    }
    red_sync();
    double alpha = 1.0;
    {
      double amax_g = 1e300; int wsel = 0;
      for (int i = 0; i < W; ++i) { const double a_ = slds[oRed + red_buf * 8 * W + 0 * W + i]; if (a_ < amax_g) { amax_g = a_; wsel = i; } }
      q1 = red_sum(5); q2 = red_sum(6);
      if (amax_g < 1e299) {
        bp = slds[oRed + red_buf * 8 * W + 1 * W + wsel]; bdp = slds[oRed + red_buf * 8 * W + 2 * W + wsel];
        bd = slds[oRed + red_buf * 8 * W + 3 * W + wsel]; bdd = slds[oRed + red_buf * 8 * W + 4 * W + wsel];
        const double gamma_f = 0.99, gamma_a = 1.0 / (1.0 - gamma_f);
        const double mufull = fmax(0.0, gap + amax_g * (q1 + amax_g * q2)) / cnt / gamma_a;
        const double a_h = (-bp + mufull / (bd + amax_g * bdd)) / bdp;
        alpha = fmin(1.0, fmin(0.99999999 * amax_g, fmax(a_h, gamma_f * amax_g)));
      }
    }
    red_next();
    STAMP(12);
    // update, fused with the residual / weight phase of the next iteration
    double xn = 0, zn = 0, s_gap = 0, m_rp = 0;
    for (int js = w; js < JT; js += W) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(js);
      const double l = aL[ix], u = aU[ix], va = aVA[ix], dv = aVC[ix], rpl = aRPL[ix], rpu = aRPU[ix];
      double tl = aTL[ix], tu = aTU[ix], zl = aZL[ix], zu = aZU[ix];
      const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
      if (hl) {
        const double dta = va + rpl, dza = -zl - (zl / tl) * dta;
        const double cl = smu - cw * dta * dza;
        const double dt = dv + rpl, dz = -zl + cl / tl - (zl / tl) * dt;
        tl += alpha * dt; zl += alpha * dz;
        aTL[ix] = tl; aZL[ix] = zl;
        zn = fmax(zn, zl);
      }
      if (hu) {
        const double dta = -va + rpu, dza = -zu - (zu / tu) * dta;
        const double cu = smu - cw * dta * dza;
        const double dt = -dv + rpu, dz = -zu + cu / tu - (zu / tu) * dt;
        tu += alpha * dt; zu += alpha * dz;
        aTU[ix] = tu; aZU[ix] = zu;
        zn = fmax(zn, zu);
      }
      const double v = aV[ix] + alpha * dv;
      aV[ix] = v;
      row1_body(ix, valid, l, u, v, tl, tu, zl, zu, s_gap, m_rp);
    }
    for (int i = tid; i < n; i += NTH) { const double xv = X_(i) + alpha * DX_(i); X_(i) = xv; xn = fmax(xn, fabs(xv)); }
    const double rp_prev = rp_rel;
    red_put(0, wave_sum(s_gap)); red_put(1, wave_max(m_rp)); red_put(2, wave_max(xn)); red_put(3, wave_max(zn));
    red_sync();
    gap = red_sum(0); rp_rel = red_max(1); xn = red_max(2); zn = red_max(3);
    red_next();
    STAMP(13);
    // divergence heuristics -> qpOASES exit codes (qpOASES.m:43-47)
    if (xn > 1e13) { flag = -3; break; }
    if (zn > 1e15 && rp_prev > 1e-6) { flag = -2; break; }
    if (stall > (have_saved ? 5 : 25)) { flag = have_saved ? 2 : (rp_prev > 1e-6 ? -2 : 1); break; }
  }
  __syncthreads();

  // ---- outputs ----
  if (flag == 2) {  // restore the best iterate that met tol_loose
    for (int i = tid; i < np; i += NTH) X_(i) = XS[i];
    for (int js = w; js < JT; js += W) aW3[js * 64 + lane] = LAMS[js * 64 + lane];
    flag = 0;
  } else if (flag == 0 || flag == 1 || flag == 4) {
    for (int js = w; js < JT; js += W) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(js);
      const bool hl = valid && aL[ix] > -INFINITY, hu = valid && aU[ix] < INFINITY;
      aW3[ix] = (hl ? aZL[ix] : 0.0) - (hu ? aZU[ix] : 0.0);
    }
  }
  __syncthreads();
  if (flag == 4) flag = -1;   // not certified (polish lives in a later revision of this kernel)
  const bool have_x = flag == 0 || flag == 1;
  double* xo = P.x + (size_t)b * n;
  for (int i = tid; i < n; i += NTH) xo[i] = have_x ? X_(i) * EV_(i) : NAN;
  if (P.lambda) {
    double* lo = P.lambda + (size_t)b * (n + m);
    for (int i = tid; i < n; i += NTH) lo[i] = have_x ? aW3[ixv(i)] / EV_(i) : NAN;
    for (int js = w; js < J; js += W) {
      const int r = perm[js * 64 + lane];   // original row of this sorted position
      if (r >= 0) lo[n + r] = have_x ? aW3[js * 64 + lane] * Fs[js * 64 + lane] : NAN;
    }
  }
  if (have_x) {  // objective at the returned point (H~, g~ scaling is objective preserving)
    hx_full(VEC(V_X));
    double fl = 0;
    for (int i = lane; i < n; i += 64) fl += 0.5 * X_(i) * HX_(i) + G_(i) * X_(i);
    fval_s = wave_sum(fl);
  }
  STAMP(14);
  STAMP_OUT;
  if (tid == 0) {
    P.fval[b] = have_x ? fval_s : NAN;
    P.exitflag[b] = flag;
    P.iter[b] = it;
    if (P.polished) P.polished[b] = flag_polished;
  }
}

}  // namespace

static_assert(V_NARR == QP_WG_NVEC, "qp_wg_lds_base_bytes (qp_solver.h) mirrors the LDS carve of qp_wg_kernel");

template <int T, int NB, int W, bool RES> static hipError_t launch_wg(const QpParams& P, int batch, hipStream_t st) {
  const size_t lds = qp_wg_lds_base_bytes(P.d, W) + (RES ? (size_t)P.d.lds_aw_bytes : 0);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qp_wg_kernel<T, NB, W, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((qp_wg_kernel<T, NB, W, RES>), dim3(batch), dim3(64 * W), lds, st, P);
  return hipGetLastError();
}
// resident kernel (when the host reserved LDS for the stream) followed by the streaming kernel for the leftovers
template <int T, int NB> static hipError_t launch_wg_T(const QpParams& P0, int batch, hipStream_t st) {
  QpParams P = P0;
  if constexpr (T <= QP_WG_RES_MAX_T) {
    if (P.d.lds_aw_bytes > 0) {
      P.only_pending = 0;
      hipError_t e = P.d.W == 8 ? launch_wg<T, NB, 8, true>(P, batch, st) : launch_wg<T, NB, 4, true>(P, batch, st);
      if (e != hipSuccess) return e;
      P.only_pending = 1;
      return launch_wg<T, NB, 4, false>(P, batch, st);
    }
  }
  P.only_pending = 0;
  return launch_wg<T, NB, 4, false>(P, batch, st);
}

#ifndef QP_WG_TLO
#define QP_WG_TLO 1
#define QP_WG_THI 12
#endif
// instantiated tile counts of this translation unit: [QP_WG_TLO, QP_WG_THI]
template <int T> static hipError_t launch_wg_sel(const QpParams& P, int batch, hipStream_t st) {
  if constexpr (T >= QP_WG_TLO && T <= QP_WG_THI) {
    if (P.d.T == T) return P.d.NB == 0 ? launch_wg_T<T, 0>(P, batch, st) : launch_wg_T<T, 4>(P, batch, st);
  }
  if constexpr (T < QP_MAX_T) return launch_wg_sel<T + 1>(P, batch, st);
  return hipErrorInvalidValue;
}
#define QP_WG_CAT2(a, b) a##b
#define QP_WG_CAT(a, b) QP_WG_CAT2(a, b)
#ifdef QP_WG_ONE_TU
hipError_t qp_wg_launch_1(const QpParams& P, int batch, hipStream_t st) { return launch_wg_sel<1>(P, batch, st); }
#else
hipError_t QP_WG_CAT(qp_wg_launch_, QP_WG_TLO)(const QpParams& P, int batch, hipStream_t st) { return launch_wg_sel<1>(P, batch, st); }
#endif
