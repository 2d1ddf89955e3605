// qp_wg.hip -- batched dense convex QP solve on MI355X (gfx950): ONE WORKGROUP of W wavefronts per QP.
//
// Replaces the qpOASES MEX call of the reference
//   (mpc/ltv/kinematic/ltvmpc_kinetmatic_curvilinear.m:52, mpc/ltv/dynamic/ltvmpc_dynamic_curvilinear.m:52,
//    contract optimizers/matlab/qpOASES/qpOASES.m:16-62)
// with a primal-dual interior-point method (Mehrotra predictor-corrector, single step length, OOQP-style step
// heuristic).  Not a port of qpOASES: the algorithm is chosen for the hardware.
//
// Shape of the kernel (round 2; the round-1 kernel ran one wavefront per QP on all 512 registers, spilled, kept every
// per-row array in global memory and re-streamed A three times per iteration through the L2 / Infinity Cache):
//   * nothing on the per-iteration path lives in global memory except the Hessian tiles:
//       - the per-row state (bounds, slacks, multipliers, Gx) sits in REGISTERS of the lane that owns the row
//         (owner layout: slot js = 64 rows, wave w owns slots w, w+W, ...; SW slots per wave at most),
//       - what the passes over A~ need per row (weights, corrector coefficients) goes through six LDS arrays,
//       - the operand stream of A~ (written once by qp_prep_kernel, qp_solver.hip) is copied into LDS once per QP when
//         it fits (RES; kinematic shapes, 93 KB for N = 40), otherwise the passes read it from global memory;
//   * pass 1 (M = H~ + A~'DA~ on the matrix cores) is split by tiles: the upper tiles of M are dealt round-robin to
//     the W wavefronts, NT/W accumulator tiles each (T up to 12, nV <= 196; the accumulators fit, the row state of the large shapes does not:
//     the T = 8 instantiation spills 473 VGPRs at nC = 1200);
//   * the matrix-vector passes are split by slots: the wave that owns a slot also streams its 16 k-steps, so their
//     results land in the registers of the owner lane and never leave the wave;
//   * blocked right-looking Cholesky over the distributed tiles: diagonal tile in one wave (four 4-row panels on the
//     matrix cores), U_KK^-T and the panel row U_K* pass through LDS, two barriers per block step with the next
//     diagonal tile factorised while the others finish their trailing updates; triangular solves: one barrier per block
//     step (every wave that needs y_K = U_KK^-T b_K forms it itself);
//   * wave-uniform scalars are reduced through a small LDS scratch so that every wave takes the same branches.
// No hand-placed s_waitcnt, no LDS-DMA: every cross-wave hand-off is a __syncthreads().
//
// Data layout of the workspace: see qp_solver.hip (qp_prep_kernel).  fp64 MFMA lane maps (cdna_hip_programming.md
// section 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C/D col = l&15, row = (l>>4) + 4*reg.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "qp_solver.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
#define DEVINL __device__ __forceinline__
#define AINL __attribute__((always_inline))

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
// reductions over the 16 lanes sharing l>>4 (one DPP row); every lane gets the result
DEVINL double grp16_sum(double v) { v += dpp_f64<0xB1>(v); v += dpp_f64<0x4E>(v); v += dpp_f64<0x141>(v); v += dpp_f64<0x140>(v); return v; }
DEVINL double grp16_max(double v) { v = fmax(v, dpp_f64<0xB1>(v)); v = fmax(v, dpp_f64<0x4E>(v)); v = fmax(v, dpp_f64<0x141>(v)); v = fmax(v, dpp_f64<0x140>(v)); return v; }
DEVINL double grp16_min(double v) { v = fmin(v, dpp_f64<0xB1>(v)); v = fmin(v, dpp_f64<0x4E>(v)); v = fmin(v, dpp_f64<0x141>(v)); v = fmin(v, dpp_f64<0x140>(v)); return v; }
// whole-wave reductions (all 64 lanes active at the call): DPP within the four rows, then four scalar lane reads
DEVINL double wave_sum(double v) { v = grp16_sum(v); return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48)); }
DEVINL double wave_max(double v) { v = grp16_max(v); return fmax(fmax(rl(v, 0), rl(v, 16)), fmax(rl(v, 32), rl(v, 48))); }
DEVINL double wave_min(double v) { v = grp16_min(v); return fmin(fmin(rl(v, 0), rl(v, 16)), fmin(rl(v, 32), rl(v, 48))); }
DEVINL double q_sum(double v) {  // sum over the 4 lane groups (same l&15)
  v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
  return v;
}

// LDS n-vectors (np doubles each).  R1, R2 and the NB border-column vectors MB are contiguous: they are the
// right-hand-side columns 0..NB+1 of the factorisation.  Behind them: NB border columns of H~ (constant) and four
// vectors of the variable-bound rows (barrier weight, affine / centering rhs weights, multiplier).
enum VecArr { V_X = 0, V_G, V_HX, V_P1, V_P2, V_P3, V_DX, V_E, V_R1, V_R2, V_MB0 };
static_assert(V_MB0 + 4 == QP_WG_NVEC_FIXED, "qp_wg_lds_base_bytes (qp_solver.h) mirrors the LDS carve of qp_wg_kernel");

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

// 1/sqrt(t) for t > 0: hardware estimate + two Newton steps (the library call carries range scaling this path does
// not need: the pivots are floored relative to the largest diagonal entry)
DEVINL double fast_rsqrt(double t) {
  double y = __builtin_amdgcn_rsq(t);
  const double h = 0.5 * t;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
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
    auto piv = [&](double t) AINL { if (!(t > floor_abs)) { if (!(fabs(t) < INFINITY)) bad = 1; t = floor_abs; } return fast_rsqrt(t); };
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

// per-row state in the registers of the owner lane
struct RowState {
  double l, u, tl, tu, zl, zu, v;   // scaled bounds (+-inf: no such side), slacks, multipliers, (G x)_row
  double dl, du;                    // zl/tl, zu/tu of the current iterate
  double va, vc, wc;                // (G dx_aff)_row, (G dx_cen)_row, (G dx_cor)_row  [vc is reused for the full G dx]
};

// ---------------------------------------------------------------------------------------------
// solve kernel: T column tiles of 16, NB border columns (0, 1 or 4), W wavefronts per QP, at most SW owner-layout
// slots per wave, RES = operand stream resident in LDS
// ---------------------------------------------------------------------------------------------
// second launch bound = waves per SIMD the register budget must allow: streaming variants with few waves share a CU
template <int T, int NB, int W, int SW, bool RES> __global__ __launch_bounds__(64 * W, (!RES && W <= 4) ? 2 : (W >= 8 ? 2 : 1)) void qp_wg_kernel(QpParams P) {
  constexpr int NT = T * (T + 1) / 2;
  constexpr int NTW = (NT + W - 1) / W;         // accumulator tiles per wave
  constexpr int CW = (T + W - 1) / W;           // column tiles per wave (A'w products of pass 1)
  constexpr int NBB = NB > 0 ? NB : 1;
  constexpr int NTH = 64 * W;
  constexpr int NS = 2 + NB;                    // right-hand-side columns riding along the factorisation
  constexpr int V_HB0 = V_MB0 + NB, V_DV = V_MB0 + 2 * NB, V_W1V = V_DV + 1, V_W2V = V_DV + 2, V_LV = V_DV + 3, V_NARR = V_DV + 4;
  const int b = blockIdx.x;
  const QpDims& d = P.d;
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int nc = 16 * T, np = nc + (NB > 0 ? 16 : 0);   // core columns, padded vector length (= d.nc, d.np: qp_make_dims)
  const int n = d.n, m = d.m, J = d.J, JT = d.J + d.JB, nb = d.nb, ntr = d.ntr;
  const int JS = d.J * 64;
  double* __restrict__ ws = P.ws + (size_t)b * d.ws_per_qp;
  const double* __restrict__ Awg = ws + d.off_Aw;
  const double* __restrict__ Hw = ws + d.off_Hw;
  const int* __restrict__ perm = reinterpret_cast<const int*>(ws + d.off_meta);
  const int* __restrict__ tcs = perm + (size_t)(d.J > 0 ? d.J : 1) * 64;
  const int* __restrict__ aoff = tcs + d.ntr;

  // ---- LDS carve (offsets in doubles; host mirror: qp_wg_lds_base_bytes in qp_solver.h) ----
  constexpr int oYL = V_NARR * np;             // T tiles U_KK^-T, row-major, 17-double rows
  constexpr int oPB = oYL + T * 272;           // T panel-row tiles in register image [p][lane]; outside the factorisation:
                                               //   W partial n-vectors of H~ z (T*256 >= W*np for every T)
  constexpr int oWP = oPB;                     // W partial n-vectors of A'w products: same region (never live together with the above)
  constexpr int oUK = oPB + T * 256;           // register image of the diagonal factor tile U_KK of the current block step
  constexpr int oScr = oUK + 256;              // per-wave scratch [6][16]
  constexpr int oRed = oScr + W * 96;          // reduction scratch: 2 buffers x 8 values x W
  constexpr int oEx = oRed + 2 * 8 * W;        // per-row exchange: [slot][6 arrays (+ NB border columns of A~ when RES)][64]
  constexpr int EXS = (6 + (RES ? NB : 0)) * 64;   // doubles per slot
  const int oAw = oEx + J * EXS;               // resident operand stream (RES)
  static_assert(T * 256 >= W * (16 * T + 16), "H~ z partials alias the panel buffer");
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
#define MB_(e, i) slds[VEC(V_MB0 + (e)) + (i)]
#define HB_(e, i) slds[VEC(V_HB0 + (e)) + (i)]
#define DV_(i) slds[VEC(V_DV) + (i)]
#define W1V_(i) slds[VEC(V_W1V) + (i)]
#define W2V_(i) slds[VEC(V_W2V) + (i)]
#define LV_(i) slds[VEC(V_LV) + (i)]
#define EX_(a, ix) slds[oEx + ((ix) >> 6) * EXS + (a) * 64 + ((ix) & 63)]
#define ABL_(e, ix) slds[oEx + ((ix) >> 6) * EXS + (6 + (e)) * 64 + ((ix) & 63)]

  // ---- my accumulator tiles: linear index i = w + W t in the column-major upper triangle (i = J(J+1)/2 + I) ----
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
  const double* __restrict__ Abg = ws + d.off_Ab;
  auto AB_ = [&](int e, int ix) AINL -> double {   // border column e of A~ at owner-layout index ix: LDS copy when RES, else global
    if constexpr (RES) return ABL_(e, ix); else return Abg[(size_t)e * JS + ix];
  };
  auto AB2_ = [&](int e, int ix) AINL -> v2d {     // pair (ix even)
    if constexpr (RES) return *reinterpret_cast<const v2d*>(&ABL_(e, ix)); else return *reinterpret_cast<const v2d*>(Abg + (size_t)e * JS + ix);
  };
  auto opnd = [&](int rec) AINL -> v2d {   // one 1 KB record: lane (c,q) gets its two k-steps
    if constexpr (RES) return *reinterpret_cast<const v2d*>(&slds[oAw + rec * 128 + lane * 2]);
    else return *reinterpret_cast<const v2d*>(Awg + (size_t)rec * 128 + lane * 2);
  };

  // ---- workgroup reductions of wave-uniform scalars (double-buffered scratch, one barrier each) ----
  int red_buf = 0;
  auto red_put = [&](int slot, double v) AINL { if (lane == 0) slds[oRed + red_buf * 8 * W + slot * W + w] = v; };
  auto red_get = [&](int slot, int i) AINL { return slds[oRed + red_buf * 8 * W + slot * W + i]; };
  auto red_sum = [&](int slot) AINL { double s = 0; for (int i = 0; i < W; ++i) s += red_get(slot, i); return s; };
  auto red_max = [&](int slot) AINL { double s = -INFINITY; for (int i = 0; i < W; ++i) s = fmax(s, red_get(slot, i)); return s; };
  auto red_min = [&](int slot) AINL { double s = INFINITY; for (int i = 0; i < W; ++i) s = fmin(s, red_get(slot, i)); return s; };
  auto red_next = [&]() AINL { red_buf ^= 1; };
  auto part = [&](int i) AINL { double s = 0; for (int ww = 0; ww < W; ++ww) s += slds[oWP + ww * np + i]; return s; };   // sum of the A'w partials
  auto part_h = [&](int i) AINL { double s = 0; for (int ww = 0; ww < W; ++ww) s += slds[oPB + ww * np + i]; return s; };   // sum of the H~ z partials

  // ---- constant per-QP data into LDS ----
  {
    const double* __restrict__ gw = ws + d.off_gw;
    const double* __restrict__ Es = ws + d.off_E;
    const double* __restrict__ Hb = ws + d.off_Hb;
    const double* __restrict__ Ab = ws + d.off_Ab;
    for (int i = tid; i < np; i += NTH) {
      G_(i) = gw[i]; EV_(i) = Es[i]; R1_(i) = 0; R2_(i) = 0; DX_(i) = 0; X_(i) = 0; DV_(i) = 0; W1V_(i) = 0; W2V_(i) = 0; LV_(i) = 0;
#pragma unroll
      for (int e = 0; e < NB; ++e) { HB_(e, i) = Hb[(size_t)e * np + i]; MB_(e, i) = 0; }
    }
    if (RES) {
      for (int ix = tid; ix < JS; ix += NTH) {
#pragma unroll
        for (int e = 0; e < NB; ++e) ABL_(e, ix) = Ab[(size_t)e * JS + ix];
      }
    }
  }

  // ---- my rows: slots js = w + W*si; bounds from the workspace (scaled by qp_prep_kernel; invalid rows carry
  //      infinite bounds and are inert), initial x = clamp(0, l, u), count finite sides ----
  RowState st[SW];
  int cnt_local = 0, infeas_l = 0;
  {
    const double* __restrict__ Lr = ws + d.off_rows;                       // row array 0: scaled lower bounds
    const double* __restrict__ Ur = ws + d.off_rows + (size_t)d.rowlen;    // row array 1: scaled upper bounds
#pragma unroll
    for (int si = 0; si < SW; ++si) {
      const int js = w + W * si;
      RowState& r = st[si];
      r.l = -INFINITY; r.u = INFINITY; r.tl = r.tu = 1.0; r.zl = r.zu = 0.0; r.v = 0.0; r.dl = r.du = 0.0; r.va = r.vc = r.wc = 0.0;
      if (js < JT) {
        double l = Lr[js * 64 + lane], u = Ur[js * 64 + lane];
        if (l > -INFINITY && u < INFINITY) {
          if (l > u) infeas_l = 1;
          if (!(u > l)) {  // equality row: open a tiny interior (documented relaxation)
            const double eps = 1e-9 * fmax(1.0, fabs(l));
            l -= eps; u += eps;
          }
        }
        cnt_local += (l > -INFINITY) + (u < INFINITY);
        r.l = l; r.u = u;
      }
    }
  }
  __syncthreads();   // LDS constants visible, X = 0 written before the owners of the variable slots set it
#pragma unroll
  for (int si = 0; si < SW; ++si) {
    const int js = w + W * si;
    if (js >= J && js < JT) {
      const int i = (js - J) * 64 + lane;
      if (i < np) {
        double xi = 0.0;
        if (st[si].l > -INFINITY && xi < st[si].l) xi = st[si].l;
        if (st[si].u < INFINITY && xi > st[si].u) xi = st[si].u;
        X_(i) = xi;
      }
    }
  }
  red_put(0, wave_sum((double)cnt_local)); red_put(1, wave_max((double)infeas_l));
  __syncthreads();
  const double cnt = fmax(1.0, red_sum(0));
  const int infeas = red_max(1) > 0;
  red_next();

  int flag = 1, it = 0, flag_polished = 0;
  double fval_s = 0.0, merit_s = INFINITY;   // objective / relative KKT residual of the point that is returned
  if (infeas) flag = -2;
  if (ws[d.off_bad] != 0.0) flag = -1;   // NaN / Inf in this QP's data (found by the prep kernel): -1 after 0 iterations
  STAMP_DECL

  // ------------------------------------------------------------------------------------------
  // streaming pass over the k-steps of MY A-row slots (slot js = 16 k-steps = 4 trips).  Per pair of k-steps the body
  // sees the operands bq[t] (zero beyond the trip's tile count), the first k-step s0 and the owner-layout index rix of
  // (s0, lane group q); the row value of k-step s0+h belongs to the lane with c == ((s0+h) & 15).
  // ------------------------------------------------------------------------------------------
#define SLOT_PASS_BEGIN(si_)                                                                                  \
    { const int js_ = w + W * (si_);                                                                           \
      if (js_ < J) {                                                                                            \
        for (int tr = 4 * js_; tr < 4 * js_ + 4 && tr < ntr; ++tr) {                                            \
          const int tc = tcs[tr], rbase = aoff[tr];                                                             \
          _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                       \
            v2d bq[T];                                                                                          \
            _Pragma("unroll") for (int t = 0; t < T; ++t) { bq[t] = (v2d){0.0, 0.0}; if (t < tc) bq[t] = opnd(rbase + u * tc + t); } \
            const int s0 = 4 * tr + 2 * u;                                                                      \
            const int rix = (s0 >> 4) * 64 + q * 16 + (s0 & 15);
#define SLOT_PASS_END } } } }

  // v = A~ z for one LDS vector, my slots; result into dst(si) of the owner lane; variable slots copy z
  auto rows_Av = [&](int oV, auto dst) AINL {
    double v[T], vb[NBB];
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = slds[oV + 16 * t + c];
#pragma unroll
    for (int f = 0; f < NBB; ++f) vb[f] = NB ? slds[oV + nc + f] : 0.0;
#pragma unroll
    for (int si = 0; si < SW; ++si) {
      SLOT_PASS_BEGIN(si)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double dsum = 0.0;
#pragma unroll
          for (int t = 0; t < T; ++t) dsum = fma(bq[t][h], v[t], dsum);
          dsum = grp16_sum(dsum);
#pragma unroll
          for (int f = 0; f < NB; ++f) dsum = fma(AB_(f, rix + h), vb[f], dsum);
          if (c == ((s0 + h) & 15)) dst(si) = dsum;
        }
      SLOT_PASS_END
      const int js = w + W * si;
      if (js >= J && js < JT) { const int i = (js - J) * 64 + lane; dst(si) = i < n ? slds[oV + i] : 0.0; }
    }
  };
  // A~' y for a per-row value held by the owner lanes (src(si)), my slots -> partial n-vector WP[w].
  // The owner lane's value is passed to its lane group through the exchange array 0 (wave-private use).
  auto rows_Atw = [&](auto src) AINL {
    double p[T], pbv[NBB];
#pragma unroll
    for (int t = 0; t < T; ++t) p[t] = 0.0;
#pragma unroll
    for (int f = 0; f < NBB; ++f) pbv[f] = 0.0;
#pragma unroll
    for (int si = 0; si < SW; ++si) {
      const int js = w + W * si;
      if (js < J) EX_(0, js * 64 + lane) = src(si);
      SLOT_PASS_BEGIN(si)
        const v2d wv = *reinterpret_cast<const v2d*>(&EX_(0, rix));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int t = 0; t < T; ++t) p[t] = fma(wv[h], bq[t][h], p[t]);
#pragma unroll
          for (int f = 0; f < NB; ++f) pbv[f] = fma(wv[h], AB_(f, rix + h), pbv[f]);
        }
      SLOT_PASS_END
    }
#pragma unroll
    for (int t = 0; t < T; ++t) { const double v = q_sum(p[t]); if (q == 0) slds[oWP + w * np + 16 * t + c] = v; }
    {   // border entries: identical on the 16 lanes of a group, sum the four groups; lanes NB..15 write the zero padding
      double vbd = 0.0;
#pragma unroll
      for (int f = 0; f < NB; ++f) { const double s = q_sum(pbv[f]); if (lane == f) vbd = s; }
      if (NB > 0 && lane < 16) slds[oWP + w * np + nc + lane] = vbd;
    }
  };

  v4d acc[NTW];          // my tiles of M, then of its Cholesky factor U
  double Ubb[NBB][NBB];  // Cholesky factor of the border Schur complement (wave-uniform scalars, every wave has them)

  // acc = H~ (my tiles) and, from the same registers, my share of H~ z -> partial n-vector WP[1][w]; then HX = sum of the
  // partials + border columns.  (Tile (I,J), I < J, contributes H_IJ z_J to rows I and H_IJ' z_I to rows J.)
  auto acc_init_hx = [&](int oZ) AINL {
    for (int i = lane; i < np; i += 64) slds[oPB + w * np + i] = 0.0;
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
#pragma unroll
    for (int t = 0; t < NTW; ++t)
      if (tJ[t] < T) {
        const int I = tI[t], Jt = tJ[t];
        const double zc = slds[oZ + 16 * Jt + c];
        double colsum = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const double rs = grp16_sum(acc[t][p] * zc);                       // row 16I + q + 4p of H_IJ z_J
          if (c == 0) slds[oPB + w * np + 16 * I + q + 4 * p] += rs;
          colsum = fma(acc[t][p], slds[oZ + 16 * I + q + 4 * p], colsum);
        }
        if (I != Jt) {
          colsum = q_sum(colsum);                                            // column 16J + c of H_IJ' z_I
          if (q == 0) slds[oPB + w * np + 16 * Jt + c] += colsum;
        }
      }
    __syncthreads();
    for (int i = tid; i < np; i += NTH) {
      double hx = 0.0;
      if (i < nc) {
        hx = part_h(i);
#pragma unroll
        for (int e = 0; e < NB; ++e) hx = fma(HB_(e, i), slds[oZ + nc + e], hx);
        HX_(i) = hx;
      } else if (i >= nc + NB) HX_(i) = 0.0;
    }
    if (NB > 0 && w == W - 1) {   // border rows: full-length dot products with the border columns (entries nc..nc+NB-1)
#pragma unroll
      for (int e = 0; e < NB; ++e) {
        double sb = 0.0;
        for (int i = lane; i < n; i += 64) sb = fma(HB_(e, i), slds[oZ + i], sb);
        sb = wave_sum(sb);
        if (lane == 0) slds[VEC(V_HX) + nc + e] = sb;
      }
    }
    __syncthreads();
  };

  // ---- v = G x ----
  rows_Av(VEC(V_X), [&](int si) AINL -> double& { return st[si].v; });
  // ---- initial slacks / multipliers in the equilibrated problem: t = max(resid, T0), z = Z0 ----
  const double T0 = 10.0, Z0 = 100.0;
#pragma unroll
  for (int si = 0; si < SW; ++si) {
    RowState& r = st[si];
    const bool hl = r.l > -INFINITY, hu = r.u < INFINITY;
    r.tl = hl ? fmax(r.v - r.l, T0) : 1.0;
    r.tu = hu ? fmax(r.u - r.v, T0) : 1.0;
    r.zl = hl ? Z0 : 0.0;
    r.zu = hu ? Z0 : 0.0;
  }
  // bound multipliers absorb the initial dual residual r = Hx + g - A'(zl - zu)
  {
    acc_init_hx(VEC(V_X));
    rows_Atw([&](int si) AINL { return st[si].zl - st[si].zu; });
    __syncthreads();
#pragma unroll
    for (int si = 0; si < SW; ++si) {
      const int js = w + W * si;
      if (js >= J && js < JT) {
        const int i = (js - J) * 64 + lane;
        if (i < n) {
          const double r = HX_(i) + G_(i) - part(i);
          if (st[si].l > -INFINITY) st[si].zl = fmax(r, 0.0) + Z0;
          if (st[si].u < INFINITY) st[si].zu = fmax(-r, 0.0) + Z0;
        }
      }
    }
    __syncthreads();   // the partials are consumed before the next H x reuses the buffer
  }

  // fall-back iterate (best one that met tol_loose): x in the workspace, multipliers per owner lane in the workspace
  double saved_merit = INFINITY, best_res = INFINITY;
  int have_saved = 0, stall = 0;
  double* __restrict__ XS = ws + d.off_save;            // np
  double* __restrict__ LAMS = ws + d.off_save + np;     // rowlen (owner layout; written and read by the owner lanes only)

  // residuals and barrier weights of my rows: everything pass 1 needs goes to the exchange arrays 0..3 (A rows) or
  // the variable-row vectors; returns the complementarity sum and the relative primal residual of my rows
  auto row_weights = [&](double& s_gap, double& m_rp) AINL {
#pragma unroll
    for (int si = 0; si < SW; ++si) {
      const int js = w + W * si;
      RowState& r = st[si];
      const bool hl = r.l > -INFINITY, hu = r.u < INFINITY;
      const double rpl = hl ? r.v - r.l - r.tl : 0.0, rpu = hu ? r.u - r.v - r.tu : 0.0;
      r.dl = hl ? r.zl / r.tl : 0.0; r.du = hu ? r.zu / r.tu : 0.0;
      const double D = r.dl + r.du;
      const double W1 = -r.dl * rpl + r.du * rpu;                           // affine rhs weight
      const double W2 = (hl ? 1.0 / r.tl : 0.0) - (hu ? 1.0 / r.tu : 0.0);  // centering weight (times sigma*mu)
      const double W3 = (hl ? r.zl : 0.0) - (hu ? r.zu : 0.0);              // current multiplier (for the dual residual)
      if (js < J) { const int ix = js * 64 + lane; EX_(0, ix) = D; EX_(1, ix) = W1; EX_(2, ix) = W2; EX_(3, ix) = W3; }
      else if (js < JT) { const int i = (js - J) * 64 + lane; if (i < np) { DV_(i) = D; W1V_(i) = W1; W2V_(i) = W2; LV_(i) = W3; } }
      s_gap += (hl ? r.tl * r.zl : 0.0) + (hu ? r.tu * r.zu : 0.0);
      const double sc = fmax(1.0, fabs(r.v));
      if (hl) m_rp = fmax(m_rp, fabs(rpl) / fmax(sc, fabs(r.l)));
      if (hu) m_rp = fmax(m_rp, fabs(rpu) / fmax(sc, fabs(r.u)));
    }
  };
  double gap = 0.0, rp_rel = 0.0;   // carried across iterations (produced by the update sweep)

  // right-hand-side tile row K of `ns` contiguous LDS vectors starting at vo: B-operand form, column e = vector e
  auto rhs_load = [&](int K, int vo, int ns) AINL -> v4d {
    v4d r = {0.0, 0.0, 0.0, 0.0};
    if (c < ns) {
      const int o = vo + c * np + 16 * K + q;
#pragma unroll
      for (int p = 0; p < 4; ++p) r[p] = slds[o + 4 * p];
    }
    return r;
  };
  auto rhs_store = [&](int K, int vo, int ns, const v4d& r) AINL {
    if (c < ns) {
      const int o = vo + c * np + 16 * K + q;
#pragma unroll
      for (int p = 0; p < 4; ++p) slds[o + 4 * p] = r[p];
    }
  };
  auto rhs_sub = [&](int K, int vo, int ns, const v4d& r) AINL {   // the only writer of row K at this step
    if (c < ns) {
      const int o = vo + c * np + 16 * K + q;
#pragma unroll
      for (int p = 0; p < 4; ++p) slds[o + 4 * p] -= r[p];
    }
  };
  auto tile_store17 = [&](int o, const v4d& Xt) AINL {
#pragma unroll
    for (int p = 0; p < 4; ++p) slds[o + (q + 4 * p) * 17 + c] = Xt[p];
  };
  auto tile_load17 = [&](int o) AINL -> v4d {
    v4d Z;
#pragma unroll
    for (int p = 0; p < 4; ++p) Z[p] = slds[o + (q + 4 * p) * 17 + c];
    return Z;
  };
  auto tile_load17_t = [&](int o) AINL -> v4d {
    v4d Z;
#pragma unroll
    for (int p = 0; p < 4; ++p) Z[p] = slds[o + c * 17 + q + 4 * p];
    return Z;
  };
  auto img_store = [&](int o, const v4d& Xt) AINL {
#pragma unroll
    for (int p = 0; p < 4; ++p) slds[o + p * 64 + lane] = Xt[p];
  };
  auto img_load = [&](int o) AINL -> v4d {
    v4d Z;
#pragma unroll
    for (int p = 0; p < 4; ++p) Z[p] = slds[o + p * 64 + lane];
    return Z;
  };

  // forward solve U'y = b for `ns` LDS vectors with the resident factor, in place.  One barrier per block step: every
  // wave that owns a tile of block row K forms y_K = U_KK^-T b_K itself; the owner of the diagonal tile writes y_K back
  // one step later (nobody reads row K by then).
  auto fwd_solve = [&](int vo, int ns) AINL {
    v4d ydef = {0.0, 0.0, 0.0, 0.0}; int kdef = -1;
#pragma unroll
    for (int K = 0; K < T; ++K) {
      bool need = false;
#pragma unroll
      for (int t = 0; t < NTW; ++t) need = need || (tI[t] == K && tJ[t] < T);
      if (kdef >= 0) { rhs_store(kdef, vo, ns, ydef); kdef = -1; }
      if (need) {
        const v4d yk = mfma4_new(tile_load17_t(oYL + K * 272), rhs_load(K, vo, ns));     // U_KK^-T b_K
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          if (tI[t] == K && tJ[t] == K) { ydef = yk; kdef = K; }
          if (tI[t] == K && tJ[t] > K && tJ[t] < T) rhs_sub(tJ[t], vo, ns, mfma4_new(acc[t], yk));   // b_J -= U_KJ' y_K
        }
      }
      __syncthreads();
    }
    if (kdef >= 0) rhs_store(kdef, vo, ns, ydef);
    __syncthreads();
  };
  // backward solve U x = y in place, one barrier per block step: every wave that owns a tile of block column K forms
  // x_K = U_KK^-1 y_K itself (matrix cores), turns it into the row-indexed form through its private LDS scratch, and
  // applies U_IK x_K on the VALU (no tile transposes)
  auto bwd_solve = [&](int vo, int ns) AINL {
    v4d xdef = {0.0, 0.0, 0.0, 0.0}; int kdef = -1;
#pragma unroll
    for (int K = T - 1; K >= 0; --K) {
      bool need = false;
#pragma unroll
      for (int t = 0; t < NTW; ++t) need = need || (tJ[t] == K);
      if (kdef >= 0) { rhs_store(kdef, vo, ns, xdef); kdef = -1; }
      if (need) {
        const v4d xk = mfma4_new(tile_load17(oYL + K * 272), rhs_load(K, vo, ns));       // (U_KK^-T)' y_K = U_KK^-1 y_K
        if (c < ns) {
#pragma unroll
          for (int p = 0; p < 4; ++p) slds[oScr + w * 96 + c * 16 + q + 4 * p] = xk[p];   // scratch[e][row]
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          if (tJ[t] == K && tI[t] == K) { xdef = xk; kdef = K; }
          if (tJ[t] == K && tI[t] < K) {
            for (int e = 0; e < ns; ++e) {
              const double xc = slds[oScr + w * 96 + e * 16 + c];
#pragma unroll
              for (int p = 0; p < 4; ++p) {
                const double sm = grp16_sum(acc[t][p] * xc);                // row q+4p of U_IK times x_K
                if (c == 0) slds[vo + e * np + 16 * tI[t] + q + 4 * p] -= sm;
              }
            }
          }
        }
      }
      __syncthreads();
    }
    if (kdef >= 0) rhs_store(kdef, vo, ns, xdef);
    __syncthreads();
  };
  // border part of a solve (the last wave writes; every wave holds Ubb): R holds y_c = U^-T b_c (core) and b_b
  // (border); leaves the border solution in R[nc+e] and y_c - sum_e u_e x_e in the core.  Caller syncs.
  auto border_solve = [&](int oR) AINL {
    if (NB > 0 && w == W - 1) {
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
  };

  // M = (acc from pass 1) + diag(DV) on the variable rows, border columns = H~ border + A'DA border (MB); blocked
  // right-looking Cholesky over the distributed tiles with the right-hand sides R1, R2 (and the NB border columns)
  // riding along, then the border Schur complement and the backward solve for R1, R2.  Returns 1 on a non-finite pivot.
  auto factor_solve2 = [&]() AINL -> int {
    double dmax_l = 0;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
      if (tI[t] == tJ[t] && tI[t] < T) {
        const int i = 16 * tI[t] + c;                     // diagonal element lives on lane c with q = c&3, reg c>>2
        const double dadd = i < n ? DV_(i) : 1.0;         // padded indices get a unit diagonal
        const bool mine = (q == (c & 3));
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (mine && p == (c >> 2)) { acc[t][p] += dadd; dmax_l = fmax(dmax_l, acc[t][p]); }
      }
    if (NB > 0) {
      for (int i = tid; i < np; i += NTH) {
#pragma unroll
        for (int e = 0; e < NB; ++e) {   // border column e of M: H~ column + A'DA column (+ its variable-bound weight on the diagonal)
          double v = i < n ? HB_(e, i) + MB_(e, i) : 0.0;
          if (i == nc + e) { v += e < nb ? DV_(i) : 1.0; dmax_l = fmax(dmax_l, v); }
          MB_(e, i) = v;
        }
      }
    }
    red_put(0, wave_max(dmax_l));
    __syncthreads();
    const double dmax = red_max(0);
    red_next();
    const double floor_abs = 1e-30 * dmax;
    STAMP(5);
    const int vo = VEC(V_R1);          // right-hand-side columns: R1, R2, MB[0..NB-1] (contiguous vectors)
    int fbad = 0;
    v4d ydef = {0.0, 0.0, 0.0, 0.0}; int kdef = -1;
#pragma unroll
    for (int K = 0; K < T; ++K) {
      // A: the diagonal tile (its owner arrives here straight from its trailing update of step K-1)
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (tI[t] == K && tJ[t] == K) {
          v4d Yk;
#pragma unroll
          for (int p = 0; p < 4; ++p) Yk[p] = (q + 4 * p == c) ? 1.0 : 0.0;
          fbad |= diag_factor(c, q, acc[t], Yk, floor_abs);
          tile_store17(oYL + K * 272, Yk);               // U_KK^-T stays in LDS for the solves of this iteration
          img_store(oUK, acc[t]);                        // U_KK itself: the block row is refined against it
        }
      __syncthreads();
      // B: block row K: U_KJ = U_KK^-T M_KJ -> panel images; y_K = U_KK^-T b_K formed by every wave of the row; b_J -= U_KJ' y_K
      {
        bool need = false;
#pragma unroll
        for (int t = 0; t < NTW; ++t) need = need || (tI[t] == K && tJ[t] < T);
        if (need) {
          const v4d Wk = tile_load17_t(oYL + K * 272);             // U_KK^-1 as the A operand acts as U_KK^-T
          const v4d Ukk = img_load(oUK);
          const v4d yk = mfma4_new(Wk, rhs_load(K, vo, NS));
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            if (tI[t] == K && tJ[t] == K) { ydef = yk; kdef = K; }
            if (tI[t] == K && tJ[t] > K && tJ[t] < T) {
              // U_KJ = U_KK^-T M_KJ through the explicit inverse + one step of refinement against U_KK (see FactorStep in
              // qp_solver.hip: the unrefined row carries a backward error of cond(U_KK) eps)
              v4d Rr = acc[t];
              acc[t] = mfma4_new(Wk, Rr);
              mfma4_sub(Ukk, acc[t], Rr);
#pragma unroll
              for (int p = 0; p < 4; ++p) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Wk[p], Rr[p], acc[t], 0, 0, 0);
              img_store(oPB + tJ[t] * 256, acc[t]);
              rhs_sub(tJ[t], vo, NS, mfma4_new(acc[t], yk));
            }
          }
        }
      }
      __syncthreads();
      // C: trailing update M_IJ -= U_KI' U_KJ; the owner of the diagonal tile writes y_K back (row K is final now)
      if (kdef >= 0) { rhs_store(kdef, vo, NS, ydef); kdef = -1; }
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (tI[t] > K && tI[t] < T) mfma4_sub(img_load(oPB + tI[t] * 256), img_load(oPB + tJ[t] * 256), acc[t]);
    }
    red_put(0, (double)fbad);
    __syncthreads();
    fbad = red_max(0) > 0;
    red_next();
    STAMP(6);
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
    if (NB > 0) {
      border_solve(VEC(V_R1)); border_solve(VEC(V_R2));
      __syncthreads();
    }
    bwd_solve(vo, 2);
    return 0;
  };
  // one more solve with the resident factor: V <- M^-1 V (LDS n-vector, in place)
  auto solve1 = [&](int oV) AINL {
    fwd_solve(oV, 1);
    if (NB > 0) { border_solve(oV); __syncthreads(); }
    bwd_solve(oV, 1);
  };

  // pass 1: acc += A~' D A~ on the matrix cores (my tiles); A~'w1..w3 and the border products for my column tiles on
  // the VALU beside them.  Per-row weights from the exchange arrays 0..3.  Writes P1..P3 and MB (border column / block
  // of A'DA) directly (one owner per entry).
  auto pass_syrk = [&]() AINL {
    double p1[CW], p2[CW], p3[CW], pb[NBB][CW], sbb[NBB][NBB], pwb[3][NBB];
#pragma unroll
    for (int ci = 0; ci < CW; ++ci) {
      p1[ci] = p2[ci] = p3[ci] = 0.0;
#pragma unroll
      for (int e = 0; e < NBB; ++e) pb[e][ci] = 0.0;
    }
#pragma unroll
    for (int e = 0; e < NBB; ++e) {
#pragma unroll
      for (int f = 0; f < NBB; ++f) sbb[e][f] = 0.0;
      pwb[0][e] = pwb[1][e] = pwb[2][e] = 0.0;
    }
    // software pipeline over the pairs of k-steps: the operands and per-row weights of pair pi+1 are requested before
    // the matrix-core work of pair pi is issued
    struct PairOps { v2d bi[NTW], bj[NTW], bc[CW], dd, w1, w2, w3, ab[NBB]; int tc; };
    auto load_pair = [&](int pi, PairOps& o) AINL {
      const int tr = pi >> 1, u = pi & 1;
      const int tc = tcs[tr], rb = aoff[tr] + u * tc;
      const int s0 = 4 * tr + 2 * u;
      const int rix = (s0 >> 4) * 64 + q * 16 + (s0 & 15);
      o.tc = tc;
      o.dd = *reinterpret_cast<const v2d*>(&EX_(0, rix)); o.w1 = *reinterpret_cast<const v2d*>(&EX_(1, rix));
      o.w2 = *reinterpret_cast<const v2d*>(&EX_(2, rix)); o.w3 = *reinterpret_cast<const v2d*>(&EX_(3, rix));
#pragma unroll
      for (int e = 0; e < NB; ++e) o.ab[e] = AB2_(e, rix);
#pragma unroll
      for (int t = 0; t < NTW; ++t) if (tJ[t] < tc) { o.bi[t] = opnd(rb + tI[t]); o.bj[t] = opnd(rb + tJ[t]); }
#pragma unroll
      for (int ci = 0; ci < CW; ++ci) if (w + W * ci < tc) o.bc[ci] = opnd(rb + w + W * ci);
    };
    auto do_pair = [&](const PairOps& o) AINL {
      const int tc = o.tc;
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (tJ[t] < tc) {
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.dd[h] * o.bi[t][h], o.bj[t][h], acc[t], 0, 0, 0);
        }
      // VALU side products for my column tiles (and the border scalars on the last wave)
#pragma unroll
      for (int ci = 0; ci < CW; ++ci)
        if (w + W * ci < tc) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const double bch = o.bc[ci][h];
            p1[ci] = fma(o.w1[h], bch, p1[ci]); p2[ci] = fma(o.w2[h], bch, p2[ci]); p3[ci] = fma(o.w3[h], bch, p3[ci]);
#pragma unroll
            for (int e = 0; e < NB; ++e) pb[e][ci] = fma(o.dd[h] * o.ab[e][h], bch, pb[e][ci]);
          }
        }
      if (NB > 0 && w == W - 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < NB; ++e) {
            const double dab = o.dd[h] * o.ab[e][h];
#pragma unroll
            for (int f = e; f < NB; ++f) sbb[e][f] = fma(dab, o.ab[f][h], sbb[e][f]);
            pwb[0][e] = fma(o.w1[h], o.ab[e][h], pwb[0][e]); pwb[1][e] = fma(o.w2[h], o.ab[e][h], pwb[1][e]); pwb[2][e] = fma(o.w3[h], o.ab[e][h], pwb[2][e]);
          }
      }
    };
    {
      const int npair = 2 * ntr;
      PairOps cur, nxt;
      if (npair > 0) load_pair(0, cur);
      for (int pi = 0; pi < npair; ++pi) {
        if (pi + 1 < npair) load_pair(pi + 1, nxt);
        do_pair(cur);
        cur = nxt;
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
    if (NB > 0 && w == W - 1) {
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

  STAMP(0);
  for (it = 0; flag == 1; ++it) {
    // ================= row phase 1: residuals, weights (only on entry; afterwards fused into the update sweep) =================
    if (it == 0) {
      double s_gap = 0, m_rp = 0;
      row_weights(s_gap, m_rp);
      red_put(0, wave_sum(s_gap)); red_put(1, wave_max(m_rp));
      __syncthreads();
      gap = red_sum(0); rp_rel = red_max(1);
      red_next();
    }
    const double mu = gap / cnt;
    STAMP(1);
    // ================= H~ x and the accumulator initialisation from the same tile loads =================
    acc_init_hx(VEC(V_X));
    STAMP(2);
    // ================= pass 1: M = H + A'DA (MFMA), p1, p2, p3 =================
    pass_syrk();
    STAMP(3);
    // objective, dual residual (every wave computes the same numbers from LDS)
    double fl = 0, m_rd = 0;
    for (int i = lane; i < n; i += 64) {
      const double gz = P3_(i) + LV_(i);
      fl += 0.5 * X_(i) * HX_(i) + G_(i) * X_(i);
      const double sc = fmax(1.0, fmax(fabs(G_(i)), fmax(fabs(HX_(i)), fabs(gz))));
      m_rd = fmax(m_rd, fabs(HX_(i) + G_(i) - gz) / sc);
    }
    const double fval = wave_sum(fl);
    const double rd_rel = wave_max(m_rd);
    const double gap_rel = gap / fmax(1.0, fabs(fval));
    const double merit = fmax(rd_rel, fmax(rp_rel, gap_rel));
    fval_s = fval; merit_s = merit;
    const bool res_ok = merit <= P.tol;
    if (!(merit < INFINITY)) { flag = have_saved ? 2 : -1; break; }
    if (merit <= P.tol_loose && merit < saved_merit) {
      for (int i = tid; i < np; i += NTH) XS[i] = X_(i);
#pragma unroll
      for (int si = 0; si < SW; ++si) { const int js = w + W * si; if (js < JT) LAMS[js * 64 + lane] = st[si].zl - st[si].zu; }
      have_saved = 1; saved_merit = merit;
    } else if (merit > P.tol_loose && rp_rel <= P.tol_loose && gap_rel <= P.tol_loose) {
      // Only the dual residual is in the way: repair the certificate of a *copy* of the iterate by moving r_d into the
      // bound multipliers, where a finite bound of the right sign exists.
      double dgap_l = 0, m_rd2_l = 0;
#pragma unroll
      for (int si = 0; si < SW; ++si) {
        const int js = w + W * si;
        if (js >= J && js < JT) {
          const int i = (js - J) * 64 + lane;
          if (i < n) {
            const RowState& r = st[si];
            const double lam = r.zl - r.zu, gz = P3_(i) + lam, rr = HX_(i) + G_(i) - gz, lam2 = lam + rr;
            const bool ok = lam2 >= 0 ? r.l > -INFINITY : r.u < INFINITY;
            if (ok) dgap_l += fabs(rr) * fmax(0.0, lam2 >= 0 ? r.v - r.l : r.u - r.v);
            else {
              const double sc = fmax(1.0, fmax(fabs(G_(i)), fmax(fabs(HX_(i)), fabs(gz))));
              m_rd2_l = fmax(m_rd2_l, fabs(rr) / sc);
            }
          }
        }
      }
      red_put(0, wave_sum(dgap_l)); red_put(1, wave_max(m_rd2_l));
      __syncthreads();
      const double merit2 = fmax(red_max(1), fmax(rp_rel, (gap + red_sum(0)) / fmax(1.0, fabs(fval))));
      red_next();
      if (merit2 <= P.tol_loose && merit2 < saved_merit) {
        for (int i = tid; i < np; i += NTH) XS[i] = X_(i);
#pragma unroll
        for (int si = 0; si < SW; ++si) {
          const int js = w + W * si;
          if (js < J) LAMS[js * 64 + lane] = st[si].zl - st[si].zu;
          else if (js < JT) {
            const int i = (js - J) * 64 + lane;
            double lf = 0.0;
            if (i < n) {
              const RowState& r = st[si];
              const double lam = r.zl - r.zu, rr = HX_(i) + G_(i) - (P3_(i) + lam), lam2 = lam + rr;
              const bool ok = lam2 >= 0 ? r.l > -INFINITY : r.u < INFINITY;
              lf = ok ? lam2 : lam;
            }
            LAMS[js * 64 + lane] = lf;
          }
        }
        have_saved = 1; saved_merit = merit2;
      }
      if (have_saved) { flag = 2; break; }
    } else if (have_saved && merit > P.tol_loose) { flag = 2; break; }
    if (merit < 0.9 * best_res) { best_res = merit; stall = 0; } else ++stall;

    // ================= factorise with the affine / centering right-hand sides riding along =================
    for (int i = tid; i < np; i += NTH) {
      R1_(i) = i < n ? -(HX_(i) + G_(i)) + P1_(i) + W1V_(i) : 0.0;
      R2_(i) = i < n ? P2_(i) + W2V_(i) : 0.0;
    }
#ifdef QP_DEBUG_DUMP
    __syncthreads();
    if (P.dump && b == 0 && P.dump_stage == 1 && it == P.dump_iter) {  // debug: M, p1, p2, p3, Hx of this iteration
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (tJ[t] < T)
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int r = 16 * tI[t] + q + 4 * p, cc = 16 * tJ[t] + c;
            double v = acc[t][p];
            if (r == cc && r < n) v += DV_(r);
            if (r < n && cc < n) { P.dump[r * n + cc] = v; if (tI[t] != tJ[t] || cc >= r) P.dump[cc * n + r] = v; }
          }
      if (w == 0) {
        for (int e = 0; e < nb; ++e)
          for (int i = lane; i < n; i += 64) {
            double v = HB_(e, i) + MB_(e, i);
            if (i == nc + e) v += DV_(i);
            P.dump[i * n + nc + e] = v; P.dump[(nc + e) * n + i] = v;
          }
        for (int i = lane; i < n; i += 64) { P.dump[n * n + i] = P1_(i); P.dump[n * n + n + i] = P2_(i); P.dump[n * n + 2 * n + i] = P3_(i); P.dump[n * n + 3 * n + i] = HX_(i); }
      }
    }
#endif
    STAMP(4);
    if (factor_solve2()) {
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

    // ================= pass 2 (my slots): va = G dxa, vc = G dxc; fused: WP[0] = A~' w_cor =================
    {
      // corrector coefficients of my A rows into the exchange arrays (all waves are past pass 1: barriers since)
#pragma unroll
      for (int si = 0; si < SW; ++si) {
        const int js = w + W * si;
        if (js < J) {
          const RowState& r = st[si];
          const bool hl = r.l > -INFINITY, hu = r.u < INFINITY;
          const int ix = js * 64 + lane;
          EX_(0, ix) = hl ? r.v - r.l - r.tl : 0.0; EX_(1, ix) = r.dl; EX_(2, ix) = hl ? r.dl / r.tl : 0.0;
          EX_(3, ix) = hu ? r.u - r.v - r.tu : 0.0; EX_(4, ix) = r.du; EX_(5, ix) = hu ? r.du / r.tu : 0.0;
        }
      }
      double v[2][T], vb[2][NBB], pc[T], pcb[NBB];
#pragma unroll
      for (int t = 0; t < T; ++t) { v[0][t] = R1_(16 * t + c); v[1][t] = R2_(16 * t + c); pc[t] = 0.0; }
#pragma unroll
      for (int f = 0; f < NBB; ++f) { vb[0][f] = NB ? R1_(nc + f) : 0.0; vb[1][f] = NB ? R2_(nc + f) : 0.0; pcb[f] = 0.0; }
#pragma unroll
      for (int si = 0; si < SW; ++si) {
        SLOT_PASS_BEGIN(si)
          const v2d rpl = *reinterpret_cast<const v2d*>(&EX_(0, rix)), cb1 = *reinterpret_cast<const v2d*>(&EX_(1, rix)), cc1 = *reinterpret_cast<const v2d*>(&EX_(2, rix));
          const v2d rpu = *reinterpret_cast<const v2d*>(&EX_(3, rix)), cb2 = *reinterpret_cast<const v2d*>(&EX_(4, rix)), cc2 = *reinterpret_cast<const v2d*>(&EX_(5, rix));
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            double ds0 = 0.0, ds1 = 0.0;
#pragma unroll
            for (int t = 0; t < T; ++t) { ds0 = fma(bq[t][h], v[0][t], ds0); ds1 = fma(bq[t][h], v[1][t], ds1); }
            ds0 = grp16_sum(ds0); ds1 = grp16_sum(ds1);
#pragma unroll
            for (int f = 0; f < NB; ++f) { const double a_ = AB_(f, rix + h); ds0 = fma(a_, vb[0][f], ds0); ds1 = fma(a_, vb[1][f], ds1); }
            if (c == ((s0 + h) & 15)) { st[si].va = ds0; st[si].vc = ds1; }
            const double dl_ = ds0 + rpl[h], du_ = rpu[h] - ds0;
            const double wc = dl_ * fma(cc1[h], dl_, cb1[h]) - du_ * fma(cc2[h], du_, cb2[h]);
#pragma unroll
            for (int t = 0; t < T; ++t) pc[t] = fma(wc, bq[t][h], pc[t]);
#pragma unroll
            for (int f = 0; f < NB; ++f) pcb[f] = fma(wc, AB_(f, rix + h), pcb[f]);
          }
        SLOT_PASS_END
        const int js = w + W * si;
        if (js >= J && js < JT) { const int i = (js - J) * 64 + lane; st[si].va = i < n ? R1_(i) : 0.0; st[si].vc = i < n ? R2_(i) : 0.0; }
      }
#pragma unroll
      for (int t = 0; t < T; ++t) { const double pv = q_sum(pc[t]); if (q == 0) slds[oWP + w * np + 16 * t + c] = pv; }
      {
        double pvb = 0.0;
#pragma unroll
        for (int f = 0; f < NB; ++f) { const double s = q_sum(pcb[f]); if (lane == f) pvb = s; }
        if (NB > 0 && lane < 16) slds[oWP + w * np + nc + lane] = pvb;
      }
    }
    STAMP(8);
    // ================= row phase 2: affine step length, sigma, second-order weights of the variable rows =================
    double a_aff = 1.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int si = 0; si < SW; ++si) {
      const int js = w + W * si;
      const RowState& r = st[si];
      const bool hl = r.l > -INFINITY, hu = r.u < INFINITY;
      double wv = 0.0;
      if (hl) {
        const double dt = r.va + (r.v - r.l - r.tl), dz = -r.zl - r.dl * dt;
        if (dt < 0) a_aff = fmin(a_aff, -r.tl / dt);
        if (dz < 0) a_aff = fmin(a_aff, -r.zl / dz);
        s1 += r.tl * dz + r.zl * dt; s2 += dt * dz;
        wv -= dt * dz / r.tl;
      }
      if (hu) {
        const double dt = -r.va + (r.u - r.v - r.tu), dz = -r.zu - r.du * dt;
        if (dt < 0) a_aff = fmin(a_aff, -r.tu / dt);
        if (dz < 0) a_aff = fmin(a_aff, -r.zu / dz);
        s1 += r.tu * dz + r.zu * dt; s2 += dt * dz;
        wv += dt * dz / r.tu;
      }
      if (js >= J && js < JT) { const int i = (js - J) * 64 + lane; if (i < np) W1V_(i) = wv; }   // (A rows: fused in pass 2)
    }
    red_put(0, wave_min(a_aff)); red_put(1, wave_sum(s1)); red_put(2, wave_sum(s2));
    __syncthreads();
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
      for (int i = tid; i < np; i += NTH) DX_(i) = i < n ? part(i) + W1V_(i) : 0.0;
      __syncthreads();
      solve1(VEC(V_DX));
      STAMP(10);
      // ================= pass 3 (my slots): G dx_cor =================
      rows_Av(VEC(V_DX), [&](int si) AINL -> double& { return st[si].wc; });
    } else {   // no corrector this iteration
      for (int i = tid; i < np; i += NTH) DX_(i) = 0.0;
#pragma unroll
      for (int si = 0; si < SW; ++si) st[si].wc = 0.0;
    }
    STAMP(11);
    // ================= row phase 3: step length (Mehrotra heuristic on the blocking pair) =================
    double amax = 1e300, bp = 0, bdp = 0, bd = 0, bdd = 0, q1 = 0.0, q2 = 0.0;
#pragma unroll
    for (int si = 0; si < SW; ++si) {
      RowState& r = st[si];
      const bool hl = r.l > -INFINITY, hu = r.u < INFINITY;
      const double dv = r.va + smu * r.vc + r.wc;
      r.vc = dv;  // keep the full G dx for the update
      if (hl) {
        const double rpl = r.v - r.l - r.tl;
        const double dta = r.va + rpl, dza = -r.zl - r.dl * dta;
        const double cl = smu - cw * dta * dza;
        const double dt = dv + rpl, dz = -r.zl + cl / r.tl - r.dl * dt;
        if (dt < 0 && -r.tl / dt < amax) { amax = -r.tl / dt; bp = r.tl; bdp = dt; bd = r.zl; bdd = dz; }
        if (dz < 0 && -r.zl / dz < amax) { amax = -r.zl / dz; bp = r.zl; bdp = dz; bd = r.tl; bdd = dt; }
        q1 += r.tl * dz + r.zl * dt; q2 += dt * dz;
      }
      if (hu) {
        const double rpu = r.u - r.v - r.tu;
        const double dta = -r.va + rpu, dza = -r.zu - r.du * dta;
        const double cu = smu - cw * dta * dza;
        const double dt = -dv + rpu, dz = -r.zu + cu / r.tu - r.du * dt;
        if (dt < 0 && -r.tu / dt < amax) { amax = -r.tu / dt; bp = r.tu; bdp = dt; bd = r.zu; bdd = dz; }
        if (dz < 0 && -r.zu / dz < amax) { amax = -r.zu / dz; bp = r.zu; bdp = dz; bd = r.tu; bdd = dt; }
        q1 += r.tu * dz + r.zu * dt; q2 += dt * dz;
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
    __syncthreads();
    double alpha = 1.0;
    {
      double amax_g = 1e300; int wsel = 0;
      for (int i = 0; i < W; ++i) { const double a_ = red_get(0, i); if (a_ < amax_g) { amax_g = a_; wsel = i; } }
      q1 = red_sum(5); q2 = red_sum(6);
      if (amax_g < 1e299) {
        bp = red_get(1, wsel); bdp = red_get(2, wsel); bd = red_get(3, wsel); bdd = red_get(4, wsel);
        const double gamma_f = 0.99, gamma_a = 1.0 / (1.0 - gamma_f);
        const double mufull = fmax(0.0, gap + amax_g * (q1 + amax_g * q2)) / cnt / gamma_a;
        const double a_h = (-bp + mufull / (bd + amax_g * bdd)) / bdp;
        alpha = fmin(1.0, fmin(0.99999999 * amax_g, fmax(a_h, gamma_f * amax_g)));
      }
    }
    red_next();
    STAMP(12);
    // ================= update, fused with the residual / weight phase of the next iteration =================
    double xn = 0, zn = 0, s_gap = 0, m_rp = 0;
#pragma unroll
    for (int si = 0; si < SW; ++si) {
      RowState& r = st[si];
      const bool hl = r.l > -INFINITY, hu = r.u < INFINITY;
      const double dv = r.vc;
      if (hl) {
        const double rpl = r.v - r.l - r.tl;
        const double dta = r.va + rpl, dza = -r.zl - r.dl * dta;
        const double cl = smu - cw * dta * dza;
        const double dt = dv + rpl, dz = -r.zl + cl / r.tl - r.dl * dt;
        r.tl += alpha * dt; r.zl += alpha * dz;
        zn = fmax(zn, r.zl);
      }
      if (hu) {
        const double rpu = r.u - r.v - r.tu;
        const double dta = -r.va + rpu, dza = -r.zu - r.du * dta;
        const double cu = smu - cw * dta * dza;
        const double dt = -dv + rpu, dz = -r.zu + cu / r.tu - r.du * dt;
        r.tu += alpha * dt; r.zu += alpha * dz;
        zn = fmax(zn, r.zu);
      }
      r.v += alpha * dv;
    }
    // full direction dx = dxa + smu*dxc + dxcor (R1, R2, DX are stable since the last barrier)
    for (int i = tid; i < n; i += NTH) { const double xv = X_(i) + alpha * (R1_(i) + smu * R2_(i) + DX_(i)); X_(i) = xv; xn = fmax(xn, fabs(xv)); }
    row_weights(s_gap, m_rp);
    const double rp_prev = rp_rel;
    red_put(0, wave_sum(s_gap)); red_put(1, wave_max(m_rp)); red_put(2, wave_max(xn)); red_put(3, wave_max(zn));
    __syncthreads();
    gap = red_sum(0); rp_rel = red_max(1); xn = red_max(2); zn = red_max(3);
    red_next();
    STAMP(13);
    // divergence heuristics -> qpOASES exit codes (qpOASES.m:43-47)
    // a diverging iterate is 'unbounded' (-3) only if it is primal feasible and the objective follows it to -infinity; with a
    // primal residual it is the signature of an infeasible QP (-2); otherwise an internal failure (-1)
    if (xn > 1e13) { flag = rp_prev > 1e-6 ? -2 : (fval < -1e13 ? -3 : -1); break; }
    if (zn > 1e15 && rp_prev > 1e-6) { flag = -2; break; }
    if (stall > (have_saved ? 5 : 25)) { flag = have_saved ? 2 : (rp_prev > 1e-6 ? -2 : 1); break; }
  }
  __syncthreads();

  // ---- outputs ----
  double lam_out[SW];
#pragma unroll
  for (int si = 0; si < SW; ++si) lam_out[si] = st[si].zl - st[si].zu;
  if (flag == 2) {  // restore the best iterate that met tol_loose
    for (int i = tid; i < np; i += NTH) X_(i) = XS[i];
#pragma unroll
    for (int si = 0; si < SW; ++si) { const int js = w + W * si; if (js < JT) lam_out[si] = LAMS[js * 64 + lane]; }
    flag = 0; merit_s = saved_merit;
  }
  __syncthreads();
  if (flag == 4) flag = -1;   // not certified (this kernel has no active-set refinement yet)
  const bool have_x = true;   // the last iterate is returned whatever the exit code (main.m:163-175 keeps driving on it)
  double* xo = P.x + (size_t)b * n;
  for (int i = tid; i < n; i += NTH) xo[i] = have_x ? X_(i) * EV_(i) : NAN;
  if (P.lambda) {
    double* lo = P.lambda + (size_t)b * (n + m);
    const double* __restrict__ Fs = ws + d.off_F;
#pragma unroll
    for (int si = 0; si < SW; ++si) {
      const int js = w + W * si;
      if (js < J) {
        const int r = perm[js * 64 + lane];   // original row of this sorted position
        if (r >= 0) lo[n + r] = have_x ? lam_out[si] * Fs[js * 64 + lane] : NAN;
      } else if (js < JT) {
        const int i = (js - J) * 64 + lane;
        if (i < n) lo[i] = have_x ? lam_out[si] / EV_(i) : NAN;
      }
    }
  }
  if (have_x) {  // objective at the returned point (H~, g~ scaling is objective preserving)
    acc_init_hx(VEC(V_X));
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
    if (P.kkt) P.kkt[b] = merit_s;
  }
}

}  // namespace

template <int T, int NB, int W, int SW, bool RES> static hipError_t launch_wg(const QpParams& P, int batch, hipStream_t st) {
  const size_t lds = qp_wg_lds_base_bytes(P.d, W, NB, RES) + (RES ? (size_t)P.d.lds_aw_bytes : 0);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qp_wg_kernel<T, NB, W, SW, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((qp_wg_kernel<T, NB, W, SW, RES>), dim3(batch), dim3(64 * W), lds, st, P);
  return hipGetLastError();
}
// streaming variants by the number of owner-layout slots per wave (W = 8): JT <= 16 -> SW = 2, JT <= 32 -> SW = 4
template <int T, int NB> static hipError_t launch_stream(const QpParams& P, int batch, hipStream_t st) {
  const int JT = P.d.J + P.d.JB;
#ifdef QP_WG_EXPERIMENT   // development only: other workgroup shapes, selected by qp_make_dims from FSAEMPC_WG="W,RES"
  if (P.d.W == 4) { if (JT <= 8) return launch_wg<T, NB, 4, 2, false>(P, batch, st); if (JT <= 16) return launch_wg<T, NB, 4, 4, false>(P, batch, st); return hipErrorInvalidValue; }
  if (P.d.W == 2) { if (JT <= 8) return launch_wg<T, NB, 2, 4, false>(P, batch, st); return hipErrorInvalidValue; }
#endif
  if (JT <= 16) return launch_wg<T, NB, 8, 2, false>(P, batch, st);
#ifndef QP_WG_DEV
  if (T >= 8 && JT <= 24) return launch_wg<T, NB, 8, 3, false>(P, batch, st);   // three slots per wave: 12 fewer doubles of row state per lane
  if (JT <= 32) return launch_wg<T, NB, 8, 4, false>(P, batch, st);
#endif
  return hipErrorInvalidValue;
}
// resident kernel (when the host reserved LDS for the stream; JT <= 8) followed by the streaming kernel for the leftovers
template <int T, int NB> static hipError_t launch_wg_T(const QpParams& P0, int batch, hipStream_t st) {
  QpParams P = P0;
#ifdef QP_WG_DEV
  if constexpr (T <= QP_WG_RES_MAX_T) {
    if (P.d.lds_aw_bytes > 0 && P.d.J + P.d.JB <= 8) {
      P.only_pending = 0;
      hipError_t e;
#ifdef QP_WG_EXPERIMENT
      if (P.d.W == 4) e = launch_wg<T, NB, 4, 2, true>(P, batch, st); else
#endif
      e = launch_wg<T, NB, 8, 1, true>(P, batch, st);
      if (e != hipSuccess) return e;
      P.only_pending = 1;
      return launch_stream<T, NB>(P, batch, st);
    }
  }
#endif
  P.only_pending = 0;
  return launch_stream<T, NB>(P, batch, st);
}

#ifndef QP_WG_TLO
#define QP_WG_TLO 1
#define QP_WG_THI 12
#endif
// instantiated tile counts of this translation unit: [QP_WG_TLO, QP_WG_THI]; border width 1 only where the headline
// shapes live (T <= QP_WG_RES_MAX_T), wider borders and 1 elsewhere run the 4-column variant (padded unit columns)
template <int T> static hipError_t launch_wg_sel(const QpParams& P, int batch, hipStream_t st) {
  if constexpr (T >= QP_WG_TLO && T <= QP_WG_THI) {
    if (P.d.T == T) {
#ifdef QP_WG_DEV    // development builds: the bordered headline shapes, every variant
      if constexpr (T <= QP_WG_RES_MAX_T) { if (P.d.NBk == 1) return launch_wg_T<T, 1>(P, batch, st); }
      return launch_wg_T<T, 4>(P, batch, st);
#else               // product: bordered shapes up to T = 5 run on the one-wavefront kernel (qp_solver.hip)
      if (P.d.NBk == 0) return launch_wg_T<T, 0>(P, batch, st);
      if constexpr (T > 5) return launch_wg_T<T, 4>(P, batch, st);
      return hipErrorInvalidValue;
#endif
    }
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
