// qp_wg.hip -- batched dense convex QP solve on MI355X (gfx950): ONE WORKGROUP of W wavefronts per QP (nV up to 196).
//
// Replaces the qpOASES MEX call of the reference
//   (mpc/ltv/kinematic/ltvmpc_kinetmatic_curvilinear.m:52, mpc/ltv/dynamic/ltvmpc_dynamic_curvilinear.m:52,
//    contract optimizers/matlab/qpOASES/qpOASES.m:16-62)
// with the same algorithm as the one-wavefront kernel (qp_solver.hip): a primal-dual interior-point method (Mehrotra
// predictor-corrector, single step length, OOQP-style step heuristic) followed by an active-set refinement that lands on the
// vertex qpOASES stops at (working set from the multipliers, active bounds pinned, conjugate gradients on the dual of the
// augmented problem, add / drop corrections, acceptance by a fresh KKT evaluation).  Not a port of qpOASES.
//
// Shape of the kernel (round 3; the round-2 version kept the per-row state in registers of owner lanes: 473 spilled VGPRs at
// nC = 1200, and it had no refinement):
//   * per-row state (bounds, slacks, multipliers, G x, step products) lives in the workspace's owner-layout row arrays, the
//     same arrays the one-wavefront kernel uses; a wave sweeps the slots js = w, w + W, ... it owns, one slot (64 rows) of
//     coalesced loads at a time.  Nothing per-row is kept in registers across phases.
//   * pass 1 (M = H~ + A~'DA~ on the matrix cores) is split by tiles: the upper tiles of M are dealt round-robin to the W
//     wavefronts (<= 10 accumulator tiles per wave at T = 12).  The operand stream of A~ goes global memory -> LDS ONCE per
//     trip (4 k-steps) by LDS-DMA (`global_load_lds_dwordx4`, every wave issues its share of the trip's records) into a
//     two-chunk ring and is read from there by all waves (one workgroup barrier per trip); the per-row weights of a trip are
//     staged next to it.  (Round 2: every wave fetched the operand columns of its own tiles with plain global loads.)
//   * the matrix-vector passes are split by slots: the wave that owns a slot streams its 16 k-steps (operands of the next
//     pair of k-steps requested before the current pair is consumed), per-row coefficients through a private LDS stage;
//   * blocked right-looking Cholesky over the distributed tiles (diagonal tile in one wave, four 4-row panels on the matrix
//     cores; block rows refined against U_KK; U_KK^-T and the panel row pass through LDS; two barriers per block step);
//     triangular solves: one barrier per block step;
//   * wave-uniform scalars are reduced through a small LDS scratch so that every wave takes the same branches.
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
// Lanes of ONE wave exchanging data through LDS: the hardware keeps a wave's DS operations in order, so no s_barrier is needed, but
// the compiler must be told -- from a single thread's point of view a location another lane writes never changes, and load
// elimination / PRE across a predicated store hands lanes their own stale value (seen: the last block of the single-wave forward
// solve, right in lane group 0 and wrong in groups 1..3).  Every such hand-off gets this fence.
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

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
// exchange between the four 16-lane rows with the gfx950 lane-swap instructions (see qp_solver.hip)
struct RowPair { double a, b; };
DEVINL RowPair rows_xor16(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return {__hiloint2double(h[0], l[0]), __hiloint2double(h[1], l[1])};
}
DEVINL RowPair rows_xor32(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return {__hiloint2double(h[0], l[0]), __hiloint2double(h[1], l[1])};
}
DEVINL double q_sum(double v) {  // sum over the 4 lane groups (same l&15); every lane gets the total
  RowPair r = rows_xor16(v); v = r.a + r.b;
  r = rows_xor32(v); return r.a + r.b;
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
#if QP_STAMPS   // diagnostic build only: cycles of wave 0 per phase (barrier waits included), written to P.dump[b*16 + phase].  The factorisation
                // and the solves are shared with the refinement: its calls land in their phases (read 5..7 as main loop + refinement)
#define STAMP_DECL unsigned long long st_acc[16]; for (int i_ = 0; i_ < 16; ++i_) st_acc[i_] = 0; unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#define STAMP(id) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[id] += t_ - st_t0; st_t0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_OUT do { if (P.dump && P.dump_stage == 9 && tid == 0) for (int i_ = 0; i_ < 16; ++i_) P.dump[(size_t)b * 16 + i_] = (double)st_acc[i_]; } while (0)
#elif defined(QP_PARANOID)   // diagnostic build only: a workgroup barrier at every phase boundary (hunting a missing one)
#define STAMP_DECL
#define STAMP(id) __syncthreads()
#define STAMP_OUT do { } while (0)
#else
#define STAMP_DECL
#define STAMP(id) do { } while (0)
#define STAMP_OUT do { } while (0)
#endif

#ifndef QP_REFINE_ATTEMPTS
#define QP_REFINE_ATTEMPTS 3
#endif

// one owner-layout row as the sweeps see it
struct Row { double l, u, tl, tu, zl, zu, v; };
template <int C> struct IC { static constexpr int value = C; };
struct TagKeep { static constexpr bool value = true; };
struct TagInit { static constexpr bool value = false; };

// ---------------------------------------------------------------------------------------------
// solve kernel: T column tiles of 16, NB border columns (0 or 4), W wavefronts per QP
// ---------------------------------------------------------------------------------------------
template <int T, int NB, int W> __global__ __launch_bounds__(64 * W, W >= 8 ? 2 : 1) void qp_wg_kernel(QpParams P) {
  constexpr int NT = T * (T + 1) / 2;
  constexpr int NTW = (NT + W - 1) / W;         // accumulator tiles per wave
  constexpr int CW = (T + W - 1) / W;           // column tiles per wave (A'w products of pass 1)
  constexpr int NBB = NB > 0 ? NB : 1;
  constexpr int NTH = 64 * W;
  constexpr int NS = 2 + NB;                    // right-hand-side columns riding along the factorisation
  constexpr int V_HB0 = V_MB0 + NB, V_DV = V_MB0 + 2 * NB, V_W1V = V_DV + 1, V_W2V = V_DV + 2, V_LV = V_DV + 3, V_NARR = V_DV + 4;
  const int b = P.order ? P.order[blockIdx.x] : blockIdx.x;   // launch order: hardest-looking instances first
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
  const int* __restrict__ tcs_g = perm + (size_t)(d.J > 0 ? d.J : 1) * 64;
  const int* __restrict__ aoff_g = tcs_g + d.ntr;
  const double* __restrict__ Abg = ws + d.off_Ab;
  auto rowarr = [&](int a) AINL -> double* { return ws + d.off_rows + (size_t)a * d.rowlen; };
  double* aL = rowarr(R_L); double* aU = rowarr(R_U);
  double* aTL = rowarr(R_TL); double* aTU = rowarr(R_TU);
  double* aZL = rowarr(R_ZL); double* aZU = rowarr(R_ZU);
  double* aV = rowarr(R_V);
  double* aD = rowarr(R_D); double* aW1 = rowarr(R_W1); double* aW2 = rowarr(R_W2); double* aW3 = rowarr(R_W3);
  double* aVA = rowarr(R_VA); double* aVC = rowarr(R_VC); double* aWC = rowarr(R_RPL);   // G dx_aff, G dx_cen (then G dx), G dx_cor

  // ---- LDS carve (offsets in doubles; host mirror: qp_wg_lds_base_bytes in qp_solver.h) ----
  constexpr int oUK = V_NARR * np;             // register image of the diagonal factor tile U_KK of the current block step
  constexpr int oScr = oUK + 256;              // per-wave scratch [6][16]
  constexpr int oRed = oScr + W * 96;          // reduction scratch: 2 buffers x 8 values x W
  constexpr int oCw = oRed + 2 * 8 * W;        // per-wave coefficient stage of the slot passes: [wave][6 arrays][64]
  constexpr int oC1 = oCw + W * 6 * 64;        // pass 1: per-trip row weights and side operand, three buffers of C1S
  constexpr int C1S = 16 + 256;                // per buffer: D[16] (row q*4 + k-step), then the side operand S[pair][q][16 columns][2 k-steps]
  constexpr int oUbb = oC1 + 3 * C1S;          // NB x NB factor of the border Schur complement
  constexpr int oYL = oUbb + 16;               // T tiles U_KK^-T, row-major, 17-double rows
  constexpr int oPB = oYL + T * 272;           // T panel-row tiles in register image [p][lane]; outside the factorisation:
                                               //   W partial n-vectors of H~ z (T*256 >= W*np for every T)
  constexpr int oWP = oPB;                     // W partial n-vectors of A'w products: same region (never live together with the above)
  constexpr int oP2 = oPB + T * 256;           // W more partial n-vectors (refinement)
  constexpr int XS_ = 240 * T > W * np ? 240 * T : W * np;
  // pass 1: three chunks of 2T operand records (1 KiB each) = 768 T doubles laid over U_KK^-T tiles, panel buffer and the
  // second partials (272 T + 256 T + >= 240 T, contiguous).  Every pass 1 is followed by a factorisation that rewrites the first
  // two, and the partial vectors are dead across it.
  constexpr int oRing = oYL;
  constexpr int oMeta = oP2 + XS_;             // stream directory (ints): tcs[ntr], aoff[ntr + 1], rend[ntr]
  // The directory of the operand stream in LDS: every pair of k-steps of every pass looks its trip up, and from global memory
  // each lookup was a full round trip in front of the operand loads that depend on it (seen in the ISA of pass 1: a
  // global_load_dword + s_waitcnt vmcnt(0) at the top of every trip).  rend[tr] = end of the run of trips with tr's tile count.
  int* tcs = reinterpret_cast<int*>(&slds[oMeta]);
  int* aoff = tcs + ntr;
  int* rend = aoff + ntr + 1;
  for (int i = tid; i < 2 * ntr + 1; i += NTH) tcs[i] = tcs_g[i];   // (tcs and aoff are contiguous in the workspace too)
  for (int i = tid; i < ntr; i += NTH) { const int t0 = tcs_g[i]; int e = i + 1; while (e < ntr && tcs_g[e] == t0) ++e; rend[i] = e; }
  for (int i = tid; i < 3 * C1S; i += NTH) slds[oC1 + i] = 0.0;     // unused columns of the side operand of pass 1 stay zero
  static_assert(T * 256 >= W * (16 * T + 16), "H~ z partials alias the panel buffer");
  static_assert((oRing % 2) == 0 && (oC1 % 2) == 0 && (oCw % 2) == 0 && (oMeta % 2) == 0, "16-byte aligned LDS arrays");
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
#define CWS_(a, l_) slds[oCw + w * 384 + (a) * 64 + (l_)]

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

  auto AB_ = [&](int e, int ix) AINL -> double { return Abg[(size_t)e * JS + ix]; };   // border column e of A~ at owner-layout index ix
  auto AB2_ = [&](int e, int ix) AINL -> v2d { return *reinterpret_cast<const v2d*>(Abg + (size_t)e * JS + ix); };
  auto opnd = [&](int rec) AINL -> v2d { return *reinterpret_cast<const v2d*>(Awg + (size_t)rec * 128 + lane * 2); };   // one 1 KB record: lane (c,q) gets its two k-steps

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
    for (int i = tid; i < np; i += NTH) {
      G_(i) = gw[i]; EV_(i) = Es[i]; R1_(i) = 0; R2_(i) = 0; DX_(i) = 0; X_(i) = 0; DV_(i) = 0; W1V_(i) = 0; W2V_(i) = 0; LV_(i) = 0;
#pragma unroll
      for (int e = 0; e < NB; ++e) { HB_(e, i) = Hb[(size_t)e * np + i]; MB_(e, i) = 0; }
    }
  }

  // ---- my rows: slots js = w, w + W, ...; bounds from the workspace (scaled by qp_prep_kernel; invalid rows carry
  //      infinite bounds and are inert), initial x = clamp(0, l, u), count finite sides ----
  int cnt_local = 0, infeas_l = 0;
  for (int js = w; js < JT; js += W) {
    const int ix = js * 64 + lane;
    double l = aL[ix], u = aU[ix];
    if (l > -INFINITY && u < INFINITY) {
      if (l > u) infeas_l = 1;
      if (!(u > l)) {  // equality row: open a tiny interior (documented relaxation)
        const double eps = 1e-9 * fmax(1.0, fabs(l));
        l -= eps; u += eps; aL[ix] = l; aU[ix] = u;
      }
    }
    cnt_local += (l > -INFINITY) + (u < INFINITY);
  }
  __syncthreads();   // LDS constants visible, X = 0 written before the owners of the variable slots set it
  for (int js = w; js < JT; js += W) {
    if (js >= J) {
      const int i = (js - J) * 64 + lane, ix = js * 64 + lane;
      if (i < np) {
        const double l = aL[ix], u = aU[ix];
        double xi = 0.0;
        if (P.x_init) { const int ui = i < n ? qp_user_index(d, i) : -1; if (ui >= 0) { const double xs = P.x_init[(size_t)b * d.nu + ui] / EV_(i); if (fabs(xs) < INFINITY) xi = xs; } }
        if (l > -INFINITY && xi < l) xi = l;
        if (u < INFINITY && xi > u) xi = u;
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
  // streaming pass over the 16 k-steps (4 trips, 8 pairs) of ONE A-row slot of mine.  The operands of the next pair are
  // requested before the current pair is handed to the body; the body sees bq[t] (zero beyond the trip's tile count) and the
  // first k-step s0 of the pair; the row value of k-step s0+h belongs to the lane with c == ((s0+h) & 15).
  // ------------------------------------------------------------------------------------------
  auto slot_pairs = [&](int js, auto body) AINL {
    const int p_lo = 8 * js, p_hi = (8 * js + 8 < 2 * ntr) ? 8 * js + 8 : 2 * ntr;
    v2d cur[T], nxt[T];
    // the trip's tile count selects a straight-line variant: C loads, T - C zero fills (no per-record branches)
    auto load_c = [&](auto Cc, int rb, v2d (&o)[T]) AINL {
      constexpr int C = decltype(Cc)::value;
#pragma unroll
      for (int t = 0; t < T; ++t) { if (t < C) o[t] = opnd(rb + t); else o[t] = (v2d){0.0, 0.0}; }
    };
    auto load = [&](int pi, v2d (&o)[T]) AINL {
      const int tr = pi >> 1, tc = tcs[tr], rb = aoff[tr] + (pi & 1) * tc;
      switch (tc) {
        case 1: load_c(IC<1>{}, rb, o); break;
        case 2: if constexpr (T >= 2) load_c(IC<2>{}, rb, o); break;
        case 3: if constexpr (T >= 3) load_c(IC<3>{}, rb, o); break;
        case 4: if constexpr (T >= 4) load_c(IC<4>{}, rb, o); break;
        case 5: if constexpr (T >= 5) load_c(IC<5>{}, rb, o); break;
        case 6: if constexpr (T >= 6) load_c(IC<6>{}, rb, o); break;
        case 7: if constexpr (T >= 7) load_c(IC<7>{}, rb, o); break;
        case 8: if constexpr (T >= 8) load_c(IC<8>{}, rb, o); break;
        case 9: if constexpr (T >= 9) load_c(IC<9>{}, rb, o); break;
        case 10: if constexpr (T >= 10) load_c(IC<10>{}, rb, o); break;
        case 11: if constexpr (T >= 11) load_c(IC<11>{}, rb, o); break;
        default: if constexpr (T >= 12) load_c(IC<12>{}, rb, o); break;
      }
    };
    if (p_lo < p_hi) load(p_lo, cur);
    for (int pi = p_lo; pi < p_hi; ++pi) {
      if (pi + 1 < p_hi) load(pi + 1, nxt);
      body(cur, 2 * pi);
#pragma unroll
      for (int t = 0; t < T; ++t) cur[t] = nxt[t];
    }
  };

  // v = A~ z for one LDS vector, my slots; result into the owner-layout row array `out`; variable slots copy z
  auto rows_Av = [&](int oV, double* out) AINL {
    double v[T], vb[NBB];
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = slds[oV + 16 * t + c];
#pragma unroll
    for (int f = 0; f < NBB; ++f) vb[f] = NB ? slds[oV + nc + f] : 0.0;
    for (int js = w; js < JT; js += W) {
      if (js < J) {
        double keep = 0.0;
        slot_pairs(js, [&](const v2d (&bq)[T], int s0) AINL {
          const int rix = (s0 >> 4) * 64 + q * 16 + (s0 & 15);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            double dsum = 0.0;
#pragma unroll
            for (int t = 0; t < T; ++t) dsum = fma(bq[t][h], v[t], dsum);
            dsum = grp16_sum(dsum);
#pragma unroll
            for (int f = 0; f < NB; ++f) dsum = fma(AB_(f, rix + h), vb[f], dsum);
            if (c == ((s0 + h) & 15)) keep = dsum;
          }
        });
        out[js * 64 + lane] = keep;
      } else { const int i = (js - J) * 64 + lane; out[js * 64 + lane] = i < n ? slds[oV + i] : 0.0; }
    }
  };
  // A~' y for a per-row value of the owner lanes (src(ix)), my slots -> partial n-vector WP[w].
  // The owner lane's value is passed to its lane group through the wave's coefficient stage.
  auto rows_Atw = [&](auto src) AINL {
    double p[T], pbv[NBB];
#pragma unroll
    for (int t = 0; t < T; ++t) p[t] = 0.0;
#pragma unroll
    for (int f = 0; f < NBB; ++f) pbv[f] = 0.0;
    for (int js = w; js < J; js += W) {
      WAVE_SYNC();
      CWS_(0, lane) = src(js * 64 + lane);
      WAVE_SYNC();
      slot_pairs(js, [&](const v2d (&bq)[T], int s0) AINL {
        const int rix = (s0 >> 4) * 64 + q * 16 + (s0 & 15);
        const v2d wv = *reinterpret_cast<const v2d*>(&CWS_(0, q * 16 + (s0 & 15)));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int t = 0; t < T; ++t) p[t] = fma(wv[h], bq[t][h], p[t]);
#pragma unroll
          for (int f = 0; f < NB; ++f) pbv[f] = fma(wv[h], AB_(f, rix + h), pbv[f]);
        }
      });
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
  // Cholesky factor of the border Schur complement: NB x NB scalars in LDS (written and read by the last wave only)
#define UBB_(e, f) slds[oUbb + (e) * NBB + (f)]

  // HX = H~ z from my tiles of H~ (partial n-vector per wave, then the sum + border columns).  KEEP: the accumulators hold the
  // resident factor and stay untouched (refinement); else acc = H~ (my tiles) comes out of the same loads.
  // (Tile (I,J), I < J, contributes H_IJ z_J to rows I and H_IJ' z_I to rows J.)
  auto hx_tiles = [&](int oZ, auto keep_tag) AINL {
    constexpr bool KEEP = decltype(keep_tag)::value;
    for (int i = lane; i < np; i += 64) slds[oPB + w * np + i] = 0.0;
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      v4d h = {0.0, 0.0, 0.0, 0.0};
      if (tJ[t] < T) {
        const double* hp = Hw + ((size_t)(tI[t] * T + tJ[t]) * 4) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 4; ++p) h[p] = hp[p * 64];
        const int I = tI[t], Jt = tJ[t];
        const double zc = slds[oZ + 16 * Jt + c];
        double colsum = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const double rs = grp16_sum(h[p] * zc);                            // row 16I + q + 4p of H_IJ z_J
          if (c == 0) slds[oPB + w * np + 16 * I + q + 4 * p] += rs;
          colsum = fma(h[p], slds[oZ + 16 * I + q + 4 * p], colsum);
        }
        WAVE_SYNC();   // (lanes c == 0 and lanes q == 0 read-modify-write the same partial vector)
        if (I != Jt) {
          colsum = q_sum(colsum);                                            // column 16J + c of H_IJ' z_I
          if (q == 0) slds[oPB + w * np + 16 * Jt + c] += colsum;
        }
        WAVE_SYNC();
      }
      if (!KEEP) acc[t] = h;
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
  auto acc_init_hx = [&](int oZ) AINL { hx_tiles(oZ, TagInit{}); };
  auto hx_keep = [&](int oZ) AINL { hx_tiles(oZ, TagKeep{}); };
  auto acc_init = [&]() AINL {   // acc = H~ (my tiles) alone
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

  // ---- v = G x ----
  rows_Av(VEC(V_X), aV);
  // ---- initial slacks / multipliers in the equilibrated problem: t = max(resid, T0), z = Z0 ----
  const double T0 = 10.0, Z0 = 100.0;
  for (int js = w; js < JT; js += W) {
    const int ix = js * 64 + lane;
    const double l = aL[ix], u = aU[ix], v = aV[ix];
    const bool hl = l > -INFINITY, hu = u < INFINITY;
    aTL[ix] = hl ? fmax(v - l, T0) : 1.0;
    aTU[ix] = hu ? fmax(u - v, T0) : 1.0;
    aZL[ix] = hl ? Z0 : 0.0;
    aZU[ix] = hu ? Z0 : 0.0;
  }
  // bound multipliers absorb the initial dual residual r = Hx + g - A'(zl - zu)
  {
    acc_init_hx(VEC(V_X));
    rows_Atw([&](int ix) AINL { return aZL[ix] - aZU[ix]; });
    __syncthreads();
    for (int js = w; js < JT; js += W) {
      if (js >= J) {
        const int i = (js - J) * 64 + lane, ix = js * 64 + lane;
        if (i < n) {
          const double r = HX_(i) + G_(i) - part(i);
          if (aL[ix] > -INFINITY) aZL[ix] = fmax(r, 0.0) + Z0;
          if (aU[ix] < INFINITY) aZU[ix] = fmax(-r, 0.0) + Z0;
        }
      }
    }
    __syncthreads();   // the partials are consumed before the next H x reuses the buffer
  }

  // fall-back iterate (best one that met tol_loose): x in the workspace, multipliers per owner lane in the workspace
  double saved_merit = INFINITY, best_res = INFINITY;
  int have_saved = 0, stall = 0;
  double* XS = ws + d.off_save;            // np
  double* LAMS = ws + d.off_save + np;     // rowlen (owner layout; written and read by the owner lanes only)

  auto ld_row = [&](int ix) AINL -> Row { Row r; r.l = aL[ix]; r.u = aU[ix]; r.tl = aTL[ix]; r.tu = aTU[ix]; r.zl = aZL[ix]; r.zu = aZU[ix]; r.v = aV[ix]; return r; };
  // residuals and barrier weights of one row: what pass 1 needs goes to the row arrays D, W1, W2, W3 (A rows) or the
  // variable-row vectors; accumulates the complementarity sum and the relative primal residual
  auto row_weights = [&](int js, const Row& r, double& s_gap, double& m_rp) AINL {
    const bool hl = r.l > -INFINITY, hu = r.u < INFINITY;
    const double rpl = hl ? r.v - r.l - r.tl : 0.0, rpu = hu ? r.u - r.v - r.tu : 0.0;
    const double dl = hl ? r.zl / r.tl : 0.0, du = hu ? r.zu / r.tu : 0.0;
    const double D = dl + du;
    const double W1 = -dl * rpl + du * rpu;                               // affine rhs weight
    const double W2 = (hl ? 1.0 / r.tl : 0.0) - (hu ? 1.0 / r.tu : 0.0);  // centering weight (times sigma*mu)
    const double W3 = (hl ? r.zl : 0.0) - (hu ? r.zu : 0.0);              // current multiplier (for the dual residual)
    if (js < J) { const int ix = js * 64 + lane; aD[ix] = D; aW1[ix] = W1; aW2[ix] = W2; aW3[ix] = W3; }
    else { const int i = (js - J) * 64 + lane; if (i < np) { DV_(i) = D; W1V_(i) = W1; W2V_(i) = W2; LV_(i) = W3; } }
    s_gap += (hl ? r.tl * r.zl : 0.0) + (hu ? r.tu * r.zu : 0.0);
    const double sc = fmax(1.0, fabs(r.v));
    if (hl) m_rp = fmax(m_rp, fabs(rpl) / fmax(sc, fabs(r.l)));
    if (hu) m_rp = fmax(m_rp, fabs(rpu) / fmax(sc, fabs(r.u)));
  };
  double gap = 0.0, rp_rel = 0.0;   // carried across iterations (produced by the update sweep)

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

  // ---- triangular solves: ONE wavefront per right-hand side, on the VALU (as vec_forward / vec_backward of qp_solver.hip) ----
  // After a factorisation every wave leaves its tiles of U in the workspace as register images (72 KB at T = 8: they stay in L2);
  // the solving wave streams them from there, U_KK^-T comes from LDS.  A solve is then a few thousand cycles of one wave and no
  // workgroup barrier per block step (round 2 rode the right-hand sides through the factorisation and the solves as 16-wide tile
  // columns on the matrix cores with a barrier per block step: ~35-60 k cycles per solve).  Several right-hand sides are solved by
  // several waves side by side.  Vector layouts: "by column" lane (c, .) holds v[c]; "by row" reg p of lane (., q) holds v[q + 4p].
  double* Ug = ws + d.off_U;
  auto store_factor = [&]() AINL {
#pragma unroll
    for (int t = 0; t < NTW; ++t)
      if (tJ[t] < T) {
        double* up = Ug + (size_t)(w + W * t) * 256 + lane;
#pragma unroll
        for (int p = 0; p < 4; ++p) up[p * 64] = acc[t][p];
      }
  };
  auto utile = [&](int I, int Jt) AINL -> v4d {
    const double* up = Ug + (size_t)(Jt * (Jt + 1) / 2 + I) * 256 + lane;
    v4d z;
#pragma unroll
    for (int p = 0; p < 4; ++p) z[p] = up[p * 64];
    return z;
  };
  // U'y = b in place on the LDS vector at oV (core part), right-looking: y_K = U_KK^-T b_K (+ one step of refinement against U_KK),
  // then b_J -= U_KJ' y_K for J > K.  Called by one wave.  The tiles of block row K+1 (the first PFN of them) are requested before
  // step K's arithmetic: a step is a dependent chain of reductions, and with the loads issued behind the previous step's fence
  // every step waited a full L2 / Infinity-Cache round trip for its tiles (measured: 3.6 k cycles per step, ~0.8 k of it arithmetic).
  constexpr int PFN = T <= 8 ? T : 5;
  auto vec_fwd = [&](int oV) AINL {
    v4d pf[PFN];                                                                       // pf[j] = tile (K, K + j) of the coming step
#pragma unroll
    for (int j = 0; j < PFN; ++j) if (j < T) pf[j] = utile(0, j);
#pragma unroll
    for (int K = 0; K < T; ++K) {
      v4d row[PFN];
#pragma unroll
      for (int j = 0; j < PFN; ++j) row[j] = pf[j];
      if (K + 1 < T) {
#pragma unroll
        for (int j = 0; j < PFN; ++j) if (K + 1 + j < T) pf[j] = utile(K + 1, K + 1 + j);
      }
      const double tk = slds[oV + 16 * K + c];
      const v4d Yt = tile_load17(oYL + K * 272);                                      // U_KK^-T
      const v4d Ukk = row[0];
      double y[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) y[p] = grp16_sum(Yt[p] * tk);                       // y_K[q+4p] = sum_c Y[q+4p][c] t[c]
      {
        double r = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) r = fma(Ukk[p], y[p], r);
        r = tk - q_sum(r);                                                             // t - U_KK' y  (by column)
#pragma unroll
        for (int p = 0; p < 4; ++p) y[p] += grp16_sum(Yt[p] * r);
      }
      WAVE_SYNC();                                                                     // every lane has read its t[c] of block K
      if (c == 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) slds[oV + 16 * K + q + 4 * p] = y[p];
      }
#pragma unroll
      for (int Jt = K + 1; Jt < T; ++Jt) {
        const v4d Ukj = (Jt - K < PFN) ? row[(Jt - K) < PFN ? (Jt - K) : 0] : utile(K, Jt);
        double sj = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) sj = fma(Ukj[p], y[p], sj);                        // this lane group's rows of (U_KJ' y_K)[c]
        sj = q_sum(sj);
        if (q == 0) slds[oV + 16 * Jt + c] -= sj;
      }
      WAVE_SYNC();                                                                     // lane group 0's updates are what block K+1 reads
    }
  };
  // U x = y in place: x_K = U_KK^-1 (y_K - sum_{J>K} U_KJ x_J) (+ one step of refinement).  Called by one wave.  Same prefetch.
  auto vec_bwd = [&](int oV) AINL {
    double x[T];
    v4d pf[PFN];
#pragma unroll
    for (int j = 0; j < PFN; ++j) if (T - 1 + j < T) pf[j] = utile(T - 1, T - 1 + j);
    WAVE_SYNC();
#pragma unroll
    for (int K = T - 1; K >= 0; --K) {
      v4d row[PFN];
#pragma unroll
      for (int j = 0; j < PFN; ++j) row[j] = pf[j];
      if (K > 0) {
#pragma unroll
        for (int j = 0; j < PFN; ++j) if (K - 1 + j < T) pf[j] = utile(K - 1, K - 1 + j);
      }
      const v4d Yt = tile_load17(oYL + K * 272);
      const v4d Ukk = row[0];
      double sp[4] = {0.0, 0.0, 0.0, 0.0}, wv[4];
#pragma unroll
      for (int Jt = K + 1; Jt < T; ++Jt) {
        const v4d Ukj = (Jt - K < PFN) ? row[(Jt - K) < PFN ? (Jt - K) : 0] : utile(K, Jt);
#pragma unroll
        for (int p = 0; p < 4; ++p) sp[p] = fma(Ukj[p], x[Jt], sp[p]);                 // this lane's column of (U_KJ x_J)[q+4p]
      }
      double s2 = 0.0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        wv[p] = slds[oV + 16 * K + q + 4 * p] - (K < T - 1 ? grp16_sum(sp[p]) : 0.0);
        s2 = fma(Yt[p], wv[p], s2);
      }
      x[K] = q_sum(s2);                                                                // (U_KK^-T)' w
      {
        double dcor = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) dcor = fma(Yt[p], wv[p] - grp16_sum(Ukk[p] * x[K]), dcor);
        x[K] += q_sum(dcor);
      }
      WAVE_SYNC();                                                                     // every lane has read its y[q+4p] of block K
      if (q == 0) slds[oV + 16 * K + c] = x[K];
    }
    WAVE_SYNC();
  };
  // border part of a solve (the last wave does it; the factor of the border block is in LDS): R holds y_c = U^-T b_c (core) and b_b
  // (border); leaves the border solution in R[nc+e] and y_c - sum_e u_e x_e in the core.  Caller syncs.
  auto border_solve = [&](int oR) AINL {
    if (NB > 0 && w == W - 1) {
      double yb[NBB], xb[NBB];
      WAVE_SYNC();
#pragma unroll
      for (int e = 0; e < NB; ++e) {
        double dsum = 0.0;
        for (int i = lane; i < nc; i += 64) dsum = fma(MB_(e, i), slds[oR + i], dsum);
        double tt = slds[oR + nc + e] - wave_sum(dsum);
#pragma unroll
        for (int g2 = 0; g2 < e; ++g2) tt -= UBB_(g2, e) * yb[g2];
        yb[e] = tt / UBB_(e, e);
      }
#pragma unroll
      for (int e = NB - 1; e >= 0; --e) {
        double tt = yb[e];
#pragma unroll
        for (int f = e + 1; f < NB; ++f) tt -= UBB_(e, f) * xb[f];
        xb[e] = tt / UBB_(e, e);
      }
      WAVE_SYNC();
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
      WAVE_SYNC();
    }
  };

  // M = (acc from pass 1) + diag(DV) on the variable rows, border columns = H~ border + A'DA border (MB); blocked
  // right-looking Cholesky over the distributed tiles, then (with_rhs) the solves for R1, R2 in place: forward sweeps of R1, R2 and
  // the NB border columns side by side (one wave each), the border Schur complement, backward sweeps.  Returns 1 on a non-finite pivot.
  auto factor_solve2 = [&](bool with_rhs) AINL -> int {
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
    int fbad = 0;
#pragma unroll 1   // (run-time loop over the block steps: same speed as unrolled, an eighth of the code)
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
      // B: block row K: U_KJ = U_KK^-T M_KJ -> panel images
      {
        bool need = false;
#pragma unroll
        for (int t = 0; t < NTW; ++t) need = need || (tI[t] == K && tJ[t] > K && tJ[t] < T);
        if (need) {
          const v4d Wk = tile_load17_t(oYL + K * 272);             // U_KK^-1 as the A operand acts as U_KK^-T
          const v4d Ukk = img_load(oUK);
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            if (tI[t] == K && tJ[t] > K && tJ[t] < T) {
              // U_KJ = U_KK^-T M_KJ through the explicit inverse + one step of refinement against U_KK (see FactorStep in
              // qp_solver.hip: the unrefined row carries a backward error of cond(U_KK) eps)
              v4d Rr = acc[t];
              acc[t] = mfma4_new(Wk, Rr);
              mfma4_sub(Ukk, acc[t], Rr);
#pragma unroll
              for (int p = 0; p < 4; ++p) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Wk[p], Rr[p], acc[t], 0, 0, 0);
              img_store(oPB + tJ[t] * 256, acc[t]);
            }
          }
        }
      }
      __syncthreads();
      // C: trailing update M_IJ -= U_KI' U_KJ
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (tI[t] > K && tI[t] < T) mfma4_sub(img_load(oPB + tI[t] * 256), img_load(oPB + tJ[t] * 256), acc[t]);
    }
    store_factor();
    red_put(0, (double)fbad);
    __syncthreads();                                       // factor tiles in the workspace, U_KK^-T tiles in LDS: visible to the solving waves
    fbad = red_max(0) > 0;
    red_next();
    STAMP(6);
    if (fbad) return 1;
    // forward sweeps, one wave per right-hand side (R1, R2 and the NB border columns are contiguous LDS vectors; right-hand side e
    // goes to wave W-1 - e mod W: R1 to the last wave, R2 to the one before)
    for (int e = W - 1 - w; e < NS; e += W) if (e >= 2 || with_rhs) vec_fwd(VEC(V_R1 + e));
    if constexpr (NB > 0) {
      __syncthreads();
      // bordered factor: u_e = U^-T m_e; S = M_bb - u'u is factorised as scalars by the last wave, which also owns border_solve
      int fb2 = 0;
      if (w == W - 1) {
        double S[NBB][NBB], Ub[NBB][NBB];
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
          for (int g2 = 0; g2 < e; ++g2) dd -= Ub[g2][e] * Ub[g2][e];
          if (!(dd > floor_abs)) { if (!(fabs(dd) < INFINITY)) fb2 = 1; dd = floor_abs; }
          Ub[e][e] = sqrt(dd);
#pragma unroll
          for (int f = e + 1; f < NB; ++f) {
            double tt = S[e][f];
#pragma unroll
            for (int g2 = 0; g2 < e; ++g2) tt -= Ub[g2][e] * Ub[g2][f];
            Ub[e][f] = tt / Ub[e][e];
          }
        }
        if (lane == 0) {
#pragma unroll
          for (int e = 0; e < NB; ++e)
#pragma unroll
            for (int f = e; f < NB; ++f) UBB_(e, f) = Ub[e][f];
        }
        WAVE_SYNC();
        if (with_rhs) { border_solve(VEC(V_R1)); border_solve(VEC(V_R2)); }
      }
      red_put(0, (double)fb2);
      __syncthreads();
      fbad = red_max(0) > 0;
      red_next();
      if (fbad) return 1;
    }
    if (with_rhs) {
      if (NB == 0) __syncthreads();                       // (with a border the barrier above already separates the sweeps)
      if (w >= W - 2) vec_bwd(VEC(V_R1 + (W - 1 - w)));
      __syncthreads();
    }
    return 0;
  };
  // one more solve with the resident factor: V <- M^-1 V (LDS n-vector, in place), by the last wave
  auto solve1 = [&](int oV) AINL {
    if (w == W - 1) {
      vec_fwd(oV);
      if (NB > 0) border_solve(oV);
      vec_bwd(oV);
    }
    __syncthreads();
  };

  // pass 1: acc += A~' D A~ on the matrix cores (my tiles); A~'w1..w3 and the border products for my column tiles on
  // the VALU beside them.  Per-row weights from the row arrays D, W1, W2, W3.  Writes P1..P3 and MB (border column / block
  // of A'DA) directly (one owner per entry).
  auto pass_syrk = [&]() AINL {
    if constexpr (NB > 0) {
      // border block of A'DA and border entries of A'w1..w3: sums over ALL rows of products of per-row numbers only -- a sweep over
      // my slots (owner lanes), one wave reduction per sum, partials into the per-wave scratch (summed behind the pass's barriers)
      constexpr int NSUM = NB * (NB + 1) / 2 + 3 * NB;
      static_assert(NSUM <= 96, "per-wave scratch");
      double sm[NSUM];
#pragma unroll
      for (int k_ = 0; k_ < NSUM; ++k_) sm[k_] = 0.0;
      for (int js = w; js < J; js += W) {
        const int ix = js * 64 + lane;
        const double dd = aD[ix], w1 = aW1[ix], w2 = aW2[ix], w3 = aW3[ix];
        double ab[NB];
#pragma unroll
        for (int e = 0; e < NB; ++e) ab[e] = AB_(e, ix);
        int k_ = 0;
#pragma unroll
        for (int e = 0; e < NB; ++e)
#pragma unroll
          for (int f = e; f < NB; ++f) { sm[k_] = fma(dd * ab[e], ab[f], sm[k_]); ++k_; }
#pragma unroll
        for (int e = 0; e < NB; ++e) { sm[k_] = fma(w1, ab[e], sm[k_]); ++k_; sm[k_] = fma(w2, ab[e], sm[k_]); ++k_; sm[k_] = fma(w3, ab[e], sm[k_]); ++k_; }
      }
#pragma unroll
      for (int k_ = 0; k_ < NSUM; ++k_) { const double t_ = wave_sum(sm[k_]); if (lane == 0) slds[oScr + w * 96 + k_] = t_; }
    }
    {
      // Lane and wave id are re-made at the places that use them (two v_mbcnt / one s_mov, opaque to the compiler), and the stream
      // directory is read into scalar registers: the kernel-wide `lane` and `w` have the longest live ranges of the kernel, so they
      // are what the register allocator spills -- and a scratch reload inside the trip loop is a vector-memory operation BEHIND the
      // DMAs just issued: waiting for it (vmcnt) waits for the DMAs too, which exposes the very latency the ring hides (seen in the
      // ISA: a reload + s_waitcnt vmcnt(0) in front of every global_load_lds of the issue loop).
      auto lane_now = [&]() AINL -> int { int l_; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_)); return l_; };
      int w_r;
      asm volatile("s_mov_b32 %0, %1" : "=s"(w_r) : "s"(w));
      auto sload = [&](const int* p_) AINL -> int { return __builtin_amdgcn_readfirstlane(*p_); };
      auto issue = [&](int tr, int chunk) AINL {
        const int nrec = 2 * sload(&tcs[tr]);
        const char* g0 = reinterpret_cast<const char*>(Awg + (size_t)sload(&aoff[tr]) * 128);
        const int l16 = lane_now() * 16;
        for (int r = w_r; r < nrec; r += W)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g0 + (size_t)r * 1024 + l16),
                                           (__attribute__((address_space(3))) void*)(&slds[oRing + (chunk * 2 * T + r) * 128]), 16, 0, 0);
      };
      // stage of the trip's 16 rows: lane a*16 + e of the staging wave(s) holds array a, entry e (e = lane group q * 4 + k-step of
      // the trip).  Wave W-1: D, W1, W2, W3; wave W-2: D * border column a.  D goes to the broadcast array the matrix tiles scale
      // their A operand with; the others become columns of the SIDE OPERAND S (16 rows x 16 columns, columns 0..2 = w1..w3,
      // 3..3+NB = D * border columns, the rest stays zero), laid out as the B operand of one matrix-core instruction per pair:
      // A~_ct' S gives A~'w1..w3 and the border column of A~'DA~ for column tile ct in one accumulator tile.
      auto stage_load = [&](int tr, double (&reg)[2]) AINL {
        if (w_r < W - 2) return;
        const int ln = lane_now();
        const int s = 4 * tr, js = s >> 4, e = ln & 15;
        const int ix = js * 64 + (e >> 2) * 16 + (s & 15) + (e & 3);
        const int a = ln >> 4;
        reg[0] = reg[1] = 0.0;
        if (w_r == W - 1) reg[0] = (a == 0 ? aD : (a == 1 ? aW1 : (a == 2 ? aW2 : aW3)))[ix];
        if (NB > 0 && w_r == W - 2 && a < NB) reg[1] = aD[ix] * Abg[(size_t)a * JS + ix];
      };
      auto stage_commit = [&](int buf, const double (&reg)[2]) AINL {
        if (w_r < W - 2) return;
        const int ln = lane_now();
        const int e = ln & 15, a = ln >> 4, qq = e >> 2, kk = e & 3;
        const int si = oC1 + buf * C1S + 16 + (((kk >> 1) * 4 + qq) * 16) * 2 + (kk & 1);   // S[pair kk/2][qq][column][kk & 1]
        if (w_r == W - 1) { if (a == 0) slds[oC1 + buf * C1S + e] = reg[0]; else slds[si + (a - 1) * 2] = reg[0]; }
        if (NB > 0 && w_r == W - 2 && a < NB) slds[si + (3 + a) * 2] = reg[1];
      };
      double sreg[2] = {0.0, 0.0};
      v4d sacc[CW];
#pragma unroll
      for (int ci = 0; ci < CW; ++ci) sacc[ci] = (v4d){0.0, 0.0, 0.0, 0.0};
      if (ntr > 0) {
        issue(0, 0); stage_load(0, sreg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stage_commit(0, sreg);
        if (ntr > 1) { issue(1, 1); stage_load(1, sreg); }
      }
      int ch = 0;                       // chunk / stage buffer of the current trip (tr mod 3)
      // the barrier of trip t (see above); afterwards the DMAs and stage loads of trip t+2 are under way
      auto sync_trip = [&](int t, int cht) AINL {
#if QP_STAMPS
        const unsigned long long tw0_ = __builtin_amdgcn_s_memtime();
#endif
        const int c1 = cht == 2 ? 0 : cht + 1, c2 = c1 == 2 ? 0 : c1 + 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t + 1 < ntr) stage_commit(c1, sreg);
        __syncthreads();
#if QP_STAMPS
        st_acc[15] += __builtin_amdgcn_s_memtime() - tw0_;   // diagnostic build: share of pass 1 spent in the per-trip wait + barrier
#endif
        if (t + 2 < ntr) { issue(t + 2, c2); stage_load(t + 2, sreg); }
        asm volatile("" ::: "memory");
      };
      // One run of trips with the same tile count tc.  My active tiles of such a trip are a PREFIX of my tile list (tiles are dealt
      // in column-major order, so tJ[t] grows with t): their count K selects a straight-line body without per-tile branches.
      auto run_k = [&](auto Kc, int tr0, int tr1, int tc) AINL {
        constexpr int K = decltype(Kc)::value;
        constexpr int KK = K > 0 ? K : 1;
#ifdef QP_WG_NOPIPE   // diagnostic / guard builds
        constexpr bool PIPE = false;
#elif defined(QP_WG_NOPIPE_MASK)   // diagnostic builds: bit K set = tile count K takes the unpipelined body
        constexpr bool PIPE = K <= 5 && !((QP_WG_NOPIPE_MASK >> K) & 1);
#else
        constexpr bool PIPE = K <= 5;     // (more tiles: two pairs of operands in registers next to the accumulators would spill)
#endif
        struct POps { v2d dd, sv, bi[KK], bj[KK], bc[CW]; };
        int bcrec[CW];                    // my column tiles' records in the chunk (record 0 stands in for an inactive one)
        bool side_on[CW];
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) { const int ct = w_r + W * ci; side_on[ci] = ct < tc; bcrec[ci] = side_on[ci] ? ct * 128 : 0; }
        auto ld = [&](int chk, int u, POps& o) AINL {
          const int ln = lane_now();
          const int rb = oRing + (chk * 2 * T + u * tc) * 128 + ln * 2;
          o.dd = *reinterpret_cast<const v2d*>(&slds[oC1 + chk * C1S + (ln >> 4) * 4 + 2 * u]);
          o.sv = *reinterpret_cast<const v2d*>(&slds[oC1 + chk * C1S + 16 + u * 128 + ln * 2]);
#pragma unroll
          for (int t = 0; t < K; ++t) { o.bi[t] = *reinterpret_cast<const v2d*>(&slds[rb + tI[t] * 128]); o.bj[t] = *reinterpret_cast<const v2d*>(&slds[rb + tJ[t] * 128]); }
#pragma unroll
          for (int ci = 0; ci < CW; ++ci) o.bc[ci] = *reinterpret_cast<const v2d*>(&slds[rb + bcrec[ci]]);
#ifdef QP_WG_LDWAIT   // diagnostic builds: the reads complete before anything else is issued
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        };
        auto mm = [&](const POps& o) AINL {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < K; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.dd[h] * o.bi[t][h], o.bj[t][h], acc[t], 0, 0, 0);
          // side products of my column tiles: A~_ct' S
#pragma unroll
          for (int ci = 0; ci < CW; ++ci)
            if (side_on[ci]) {
#pragma unroll
              for (int h = 0; h < 2; ++h) sacc[ci] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.bc[ci][h], o.sv[h], sacc[ci], 0, 0, 0);
            }
        };
        POps A, B;
        sync_trip(tr0, ch);
        ld(ch, 0, A);
        for (int t = tr0;;) {
          const int chn = ch == 2 ? 0 : ch + 1;
          const bool more = t + 1 < tr1;
          if constexpr (PIPE) {
            ld(ch, 1, B);
            mm(A);
            if (more) ld(chn, 0, A);     // chunk and stage of trip t+1 are complete since the barrier of trip t
            mm(B);
          } else {
            mm(A);
            ld(ch, 1, A);
            mm(A);
            if (more) ld(chn, 0, A);
          }
          ch = chn;
          if (!more) break;
          ++t;
          sync_trip(t, ch);
        }
      };
      for (int tr = 0; tr < ntr;) {
        const int tc = sload(&tcs[tr]), tr1 = sload(&rend[tr]);
        int nact = 0;
#pragma unroll
        for (int t = 0; t < NTW; ++t) nact += (tJ[t] < tc) ? 1 : 0;
        switch (nact) {
          case 0: run_k(IC<0>{}, tr, tr1, tc); break;
          case 1: run_k(IC<1>{}, tr, tr1, tc); break;
          case 2: if constexpr (NTW >= 2) run_k(IC<2>{}, tr, tr1, tc); break;
          case 3: if constexpr (NTW >= 3) run_k(IC<3>{}, tr, tr1, tc); break;
          case 4: if constexpr (NTW >= 4) run_k(IC<4>{}, tr, tr1, tc); break;
          case 5: if constexpr (NTW >= 5) run_k(IC<5>{}, tr, tr1, tc); break;
          case 6: if constexpr (NTW >= 6) run_k(IC<6>{}, tr, tr1, tc); break;
          case 7: if constexpr (NTW >= 7) run_k(IC<7>{}, tr, tr1, tc); break;
          case 8: if constexpr (NTW >= 8) run_k(IC<8>{}, tr, tr1, tc); break;
          case 9: if constexpr (NTW >= 9) run_k(IC<9>{}, tr, tr1, tc); break;
          default: if constexpr (NTW >= 10) run_k(IC<10>{}, tr, tr1, tc); break;
        }
        tr = tr1;
      }
      const int lane_r = lane_now(), q_r = lane_r >> 4, c_r = lane_r & 15;
      // accumulator tile of column tile ct: lane (q, c), register p = entry 16 ct + q + 4 p of side column c
#pragma unroll
      for (int ci = 0; ci < CW; ++ci) {
        const int ct = w_r + W * ci;
        if (ct < T) {
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int i = 16 * ct + q_r + 4 * p;
            if (c_r == 0) P1_(i) = sacc[ci][p];
            else if (c_r == 1) P2_(i) = sacc[ci][p];
            else if (c_r == 2) P3_(i) = sacc[ci][p];
            else if (c_r < 3 + NB) MB_(c_r - 3, i) = sacc[ci][p];
          }
        }
      }
      __syncthreads();   // the ring region is reused (second set of partial n-vectors) once everybody is through the last chunk
    }
    if constexpr (NB > 0) {
      constexpr int NSUM = NB * (NB + 1) / 2 + 3 * NB;
      if (w == W - 1 && lane < NSUM) {   // the border sums of the sweep at the top (the pass's barriers lie in between): one lane per sum
        double t_ = 0.0;
#pragma unroll
        for (int ww = 0; ww < W; ++ww) t_ += slds[oScr + ww * 96 + lane];
        int k_ = 0;
#pragma unroll
        for (int e = 0; e < NB; ++e)
#pragma unroll
          for (int f = e; f < NB; ++f) { if (k_ == lane) { MB_(e, nc + f) = t_; MB_(f, nc + e) = t_; } ++k_; }
#pragma unroll
        for (int e = 0; e < NB; ++e) {
          if (k_ == lane) P1_(nc + e) = t_;
          if (k_ + 1 == lane) P2_(nc + e) = t_;
          if (k_ + 2 == lane) P3_(nc + e) = t_;
          k_ += 3;
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
      for (int js = w; js < JT; js += W) row_weights(js, ld_row(js * 64 + lane), s_gap, m_rp);
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
    // (fmax drops NaN operands: an iterate with NaN in it -- a step along a direction from a broken-down factorisation, 0 * NaN --
    //  would read as merit 0.  Its objective is NaN, and so must the merit be: the best saved iterate is returned then.)
    const double merit = (fabs(fval) < INFINITY && fabs(gap_rel) < INFINITY) ? fmax(rd_rel, fmax(rp_rel, gap_rel)) : INFINITY;
    fval_s = fval; merit_s = merit;
    const bool res_ok = merit <= P.tol;
    if (!(merit < INFINITY)) { flag = have_saved ? 2 : -1; break; }
    if (merit <= P.tol_loose && merit < saved_merit) {
      for (int i = tid; i < np; i += NTH) XS[i] = X_(i);
      for (int js = w; js < JT; js += W) {
        const int ix = js * 64 + lane;
        if (js < J) LAMS[ix] = aW3[ix]; else { const int i = (js - J) * 64 + lane; LAMS[ix] = i < np ? LV_(i) : 0.0; }
      }
      have_saved = 1; saved_merit = merit;
    } else if (merit > P.tol_loose && rp_rel <= P.tol_loose && gap_rel <= P.tol_loose) {
      // Only the dual residual is in the way: repair the certificate of a *copy* of the iterate by moving r_d into the
      // bound multipliers, where a finite bound of the right sign exists.
      double dgap_l = 0, m_rd2_l = 0;
      for (int js = w; js < JT; js += W) {
        if (js < J) continue;
        const int i = (js - J) * 64 + lane, ix = js * 64 + lane;
        if (i < n) {
          const double l = aL[ix], u = aU[ix], v = aV[ix];
          const double lam = LV_(i), gz = P3_(i) + lam, rr = HX_(i) + G_(i) - gz, lam2 = lam + rr;
          const bool ok = lam2 >= 0 ? l > -INFINITY : u < INFINITY;
          if (ok) dgap_l += fabs(rr) * fmax(0.0, lam2 >= 0 ? v - l : u - v);
          else {
            const double sc = fmax(1.0, fmax(fabs(G_(i)), fmax(fabs(HX_(i)), fabs(gz))));
            m_rd2_l = fmax(m_rd2_l, fabs(rr) / sc);
          }
        }
      }
      red_put(0, wave_sum(dgap_l)); red_put(1, wave_max(m_rd2_l));
      __syncthreads();
      const double merit2 = fmax(red_max(1), fmax(rp_rel, (gap + red_sum(0)) / fmax(1.0, fabs(fval))));
      red_next();
      if (merit2 <= P.tol_loose && merit2 < saved_merit) {
        for (int i = tid; i < np; i += NTH) XS[i] = X_(i);
        for (int js = w; js < JT; js += W) {
          const int ix = js * 64 + lane;
          if (js < J) LAMS[ix] = aW3[ix];
          else {
            const int i = (js - J) * 64 + lane;
            double lf = 0.0;
            if (i < n) {
              const double l = aL[ix], u = aU[ix];
              const double lam = LV_(i), rr = HX_(i) + G_(i) - (P3_(i) + lam), lam2 = lam + rr;
              const bool ok = lam2 >= 0 ? l > -INFINITY : u < INFINITY;
              lf = ok ? lam2 : lam;
            }
            LAMS[ix] = lf;
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
        for (int i = lane; i < n; i += 64) { P.dump[n * n + i] = P1_(i); P.dump[n * n + n + i] = P2_(i); P.dump[n * n + 2 * n + i] = P3_(i); P.dump[n * n + 3 * n + i] = HX_(i);
                                             P.dump[n * n + 4 * n + i] = R1_(i); P.dump[n * n + 5 * n + i] = R2_(i); }   // right-hand sides of the two step directions
      }
    }
    __syncthreads();
#endif
    STAMP(4);
    if (factor_solve2(true)) {
      flag = (res_ok || have_saved) ? 2 : -1;
      // the factorisation broke down (weights ~1e24) on an iterate that is nearly primal feasible and complementary: its working
      // set is usually the right one already, so the refinement gets a try (flag 4 -> 0 if accepted, else -1)
      if (flag == -1 && P.polish && rp_rel <= QP_BREAKDOWN_TRY_TOL && gap_rel <= QP_BREAKDOWN_TRY_TOL) flag = 4;
      break;
    }
    STAMP(7);
#ifdef QP_DEBUG_DUMP
    if (P.dump && b == 0 && P.dump_stage == 1 && it == P.dump_iter && w == 0)   // debug: the two step directions M^-1 R1, M^-1 R2
      for (int i = lane; i < n; i += 64) { P.dump[n * n + 6 * n + i] = R1_(i); P.dump[n * n + 7 * n + i] = R2_(i); }
#endif
    if (res_ok) {  // Newton-decrement test in the caller's coordinates
      double dm = 0, xm = 1.0;
      for (int i = lane; i < n; i += 64) { dm = fmax(dm, fabs(R1_(i) * EV_(i))); xm = fmax(xm, fabs(X_(i) * EV_(i))); }
      dm = wave_max(dm); xm = wave_max(xm);
      if (dm <= P.tol_x * xm) { flag = 0; break; }
    }
    if (it >= P.max_iter) { flag = have_saved ? 2 : 1; break; }

    // ================= pass 2 (my slots): va = G dxa, vc = G dxc; fused: WP[w] = A~' w_cor =================
    {
      double v[2][T], vb[2][NBB], pc[T], pcb[NBB];
#pragma unroll
      for (int t = 0; t < T; ++t) { v[0][t] = R1_(16 * t + c); v[1][t] = R2_(16 * t + c); pc[t] = 0.0; }
#pragma unroll
      for (int f = 0; f < NBB; ++f) { vb[0][f] = NB ? R1_(nc + f) : 0.0; vb[1][f] = NB ? R2_(nc + f) : 0.0; pcb[f] = 0.0; }
      for (int js = w; js < JT; js += W) {
        const int ix = js * 64 + lane;
        if (js < J) {
          {   // corrector coefficients of this slot's rows into the wave's stage
            const Row r = ld_row(ix);
            const bool hl = r.l > -INFINITY, hu = r.u < INFINITY;
            const double dl = hl ? r.zl / r.tl : 0.0, du = hu ? r.zu / r.tu : 0.0;
            WAVE_SYNC();
            CWS_(3, lane) = hu ? r.u - r.v - r.tu : 0.0; CWS_(4, lane) = du; CWS_(5, lane) = hu ? du / r.tu : 0.0;
            CWS_(0, lane) = hl ? r.v - r.l - r.tl : 0.0; CWS_(1, lane) = dl; CWS_(2, lane) = hl ? dl / r.tl : 0.0;
            WAVE_SYNC();
          }
          double ka = 0.0, kc = 0.0;
          slot_pairs(js, [&](const v2d (&bq)[T], int s0) AINL {
            const int rix = (s0 >> 4) * 64 + q * 16 + (s0 & 15), cix = q * 16 + (s0 & 15);
            const v2d rpl = *reinterpret_cast<const v2d*>(&CWS_(0, cix)), cb1 = *reinterpret_cast<const v2d*>(&CWS_(1, cix)), cc1 = *reinterpret_cast<const v2d*>(&CWS_(2, cix));
            const v2d rpu = *reinterpret_cast<const v2d*>(&CWS_(3, cix)), cb2 = *reinterpret_cast<const v2d*>(&CWS_(4, cix)), cc2 = *reinterpret_cast<const v2d*>(&CWS_(5, cix));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              double ds0 = 0.0, ds1 = 0.0;
#pragma unroll
              for (int t = 0; t < T; ++t) { ds0 = fma(bq[t][h], v[0][t], ds0); ds1 = fma(bq[t][h], v[1][t], ds1); }
              ds0 = grp16_sum(ds0); ds1 = grp16_sum(ds1);
#pragma unroll
              for (int f = 0; f < NB; ++f) { const double a_ = AB_(f, rix + h); ds0 = fma(a_, vb[0][f], ds0); ds1 = fma(a_, vb[1][f], ds1); }
              if (c == ((s0 + h) & 15)) { ka = ds0; kc = ds1; }
              const double dl_ = ds0 + rpl[h], du_ = rpu[h] - ds0;
              const double wc = dl_ * fma(cc1[h], dl_, cb1[h]) - du_ * fma(cc2[h], du_, cb2[h]);
#pragma unroll
              for (int t = 0; t < T; ++t) pc[t] = fma(wc, bq[t][h], pc[t]);
#pragma unroll
              for (int f = 0; f < NB; ++f) pcb[f] = fma(wc, AB_(f, rix + h), pcb[f]);
            }
          });
          aVA[ix] = ka; aVC[ix] = kc;
        } else { const int i = (js - J) * 64 + lane; aVA[ix] = i < n ? R1_(i) : 0.0; aVC[ix] = i < n ? R2_(i) : 0.0; }
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
    for (int js = w; js < JT; js += W) {
      const int ix = js * 64 + lane;
      const Row r = ld_row(ix);
      const double va = aVA[ix];
      const bool hl = r.l > -INFINITY, hu = r.u < INFINITY;
      double wv = 0.0;
      if (hl) {
        const double dl = r.zl / r.tl;
        const double dt = va + (r.v - r.l - r.tl), dz = -r.zl - dl * dt;
        if (dt < 0) a_aff = fmin(a_aff, -r.tl / dt);
        if (dz < 0) a_aff = fmin(a_aff, -r.zl / dz);
        s1 += r.tl * dz + r.zl * dt; s2 += dt * dz;
        wv -= dt * dz / r.tl;
      }
      if (hu) {
        const double du = r.zu / r.tu;
        const double dt = -va + (r.u - r.v - r.tu), dz = -r.zu - du * dt;
        if (dt < 0) a_aff = fmin(a_aff, -r.tu / dt);
        if (dz < 0) a_aff = fmin(a_aff, -r.zu / dz);
        s1 += r.tu * dz + r.zu * dt; s2 += dt * dz;
        wv += dt * dz / r.tu;
      }
      if (js >= J) { const int i = (js - J) * 64 + lane; if (i < np) W1V_(i) = wv; }   // (A rows: fused in pass 2)
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
      rows_Av(VEC(V_DX), aWC);
    } else {   // no corrector this iteration
      for (int i = tid; i < np; i += NTH) DX_(i) = 0.0;
      for (int js = w; js < JT; js += W) aWC[js * 64 + lane] = 0.0;
    }
    STAMP(11);
    // ================= row phase 3: step length (Mehrotra heuristic on the blocking pair) =================
    double amax = 1e300, bp = 0, bdp = 0, bd = 0, bdd = 0, q1 = 0.0, q2 = 0.0;
    for (int js = w; js < JT; js += W) {
      const int ix = js * 64 + lane;
      const Row r = ld_row(ix);
      const double va = aVA[ix];
      const bool hl = r.l > -INFINITY, hu = r.u < INFINITY;
      const double dv = va + smu * aVC[ix] + aWC[ix];
      aVC[ix] = dv;  // keep the full G dx for the update
      if (hl) {
        const double dl = r.zl / r.tl;
        const double rpl = r.v - r.l - r.tl;
        const double dta = va + rpl, dza = -r.zl - dl * dta;
        const double cl = smu - cw * dta * dza;
        const double dt = dv + rpl, dz = -r.zl + cl / r.tl - dl * dt;
        if (dt < 0 && -r.tl / dt < amax) { amax = -r.tl / dt; bp = r.tl; bdp = dt; bd = r.zl; bdd = dz; }
        if (dz < 0 && -r.zl / dz < amax) { amax = -r.zl / dz; bp = r.zl; bdp = dz; bd = r.tl; bdd = dt; }
        q1 += r.tl * dz + r.zl * dt; q2 += dt * dz;
      }
      if (hu) {
        const double du = r.zu / r.tu;
        const double rpu = r.u - r.v - r.tu;
        const double dta = -va + rpu, dza = -r.zu - du * dta;
        const double cu = smu - cw * dta * dza;
        const double dt = -dv + rpu, dz = -r.zu + cu / r.tu - du * dt;
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
    for (int js = w; js < JT; js += W) {
      const int ix = js * 64 + lane;
      Row r = ld_row(ix);
      const double va = aVA[ix], dv = aVC[ix];
      const bool hl = r.l > -INFINITY, hu = r.u < INFINITY;
      if (hl) {
        const double dl = r.zl / r.tl;
        const double rpl = r.v - r.l - r.tl;
        const double dta = va + rpl, dza = -r.zl - dl * dta;
        const double cl = smu - cw * dta * dza;
        const double dt = dv + rpl, dz = -r.zl + cl / r.tl - dl * dt;
        r.tl += alpha * dt; r.zl += alpha * dz;
        aTL[ix] = r.tl; aZL[ix] = r.zl;
        zn = fmax(zn, r.zl);
      }
      if (hu) {
        const double du = r.zu / r.tu;
        const double rpu = r.u - r.v - r.tu;
        const double dta = -va + rpu, dza = -r.zu - du * dta;
        const double cu = smu - cw * dta * dza;
        const double dt = -dv + rpu, dz = -r.zu + cu / r.tu - du * dt;
        r.tu += alpha * dt; r.zu += alpha * dz;
        aTU[ix] = r.tu; aZU[ix] = r.zu;
        zn = fmax(zn, r.zu);
      }
      r.v += alpha * dv;
      aV[ix] = r.v;
      row_weights(js, r, s_gap, m_rp);
    }
    // full direction dx = dxa + smu*dxc + dxcor (R1, R2, DX are stable since the last barrier)
    for (int i = tid; i < n; i += NTH) { const double xv = X_(i) + alpha * (R1_(i) + smu * R2_(i) + DX_(i)); X_(i) = xv; xn = fmax(xn, fabs(xv)); }
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
    if (stall > (have_saved ? 5 : 25)) { flag = have_saved ? 2 : (rp_prev > 1e-6 ? -2 : (P.polish ? 5 : 1)); break; }   // 5: stalled, the refinement may still certify (else 1)
  }
  __syncthreads();

  // ---- the returned point ----
  // The last iterate (or the best saved one) is returned whatever the exit code: the reference keeps driving on whatever the
  // solver handed back (main.m:163-175).  Multipliers of the returned point: row array W3 (A rows) / LDS vector LV (bounds).
  const bool v_current = flag == 0 || flag == 4;   // aV still equals G x and fval_s is the objective at x (not so after a restore / an update)
  if (flag == 2) {  // restore the best iterate that met tol_loose
    for (int i = tid; i < np; i += NTH) X_(i) = XS[i];
    for (int js = w; js < JT; js += W) {
      const int ix = js * 64 + lane;
      if (js < J) aW3[ix] = LAMS[ix]; else { const int i = (js - J) * 64 + lane; if (i < np) LV_(i) = LAMS[ix]; }
    }
    flag = 0; merit_s = saved_merit;
  }
  __syncthreads();

  // ---- active-set refinement: from the interior-point point to the vertex an active-set solver (qpOASES) stops at ----
  // Same algorithm as the one-wavefront kernel (qp_solver.hip, where the construction is described): working set W from the
  // multipliers (side active iff |lambda| exceeds its slack); active bounds pinned (1e16 on the diagonal, zero right-hand side);
  // active rows A_W z = b by conjugate gradients on the dual of the augmented problem, operator S = A_W M^-1 A_W' with
  // M = H~ + pin + rho A_W'A_W (resident Cholesky factor); one fused stream over A~ per CG step; up to QP_REFINE_ATTEMPTS attempts
  // with add / drop corrections; accepted only if a fresh evaluation says the point is a KKT point of the full QP.
  // Row arrays of this phase (A rows AND variable rows, owner layout): PS working-set side (+1 lower, -1 upper), PA rho on the
  // working set, PB target, PY multiplier, PC constraint residual, PP CG direction, QV = A~ (vector), YV = y - rho c / bound multiplier.
  // n-vectors: Z = R2 (the point), ATR = R1, ATP = P3 (A_W'r, A_W'p by recurrence), DX work vector, P1 / P2 pass outputs;
  // variable rows as n-vectors: DV pin weight (read by factor_solve2), W1V side, W2V bound value.
  if ((flag == 0 || flag == 4 || flag == 5) && P.polish) {
    const double rho = 1e6, pin = 1e16, rinv = 1.0 / rho;
    double* PS = rowarr(R_CB2); double* PA = rowarr(R_CB1); double* PB = rowarr(R_RPL);
    double* PY = rowarr(R_CC1); double* PC = rowarr(R_RPU); double* PP = rowarr(R_CC2);
    double* QV = aVA; double* YV = aVC;
    if (!v_current) rows_Av(VEC(V_X), aV);
    for (int js = w; js < JT; js += W) {
      const int ix = js * 64 + lane;
      const double l = aL[ix], u = aU[ix], v = aV[ix];
      double lam;
      if (js < J) lam = aW3[ix]; else { const int i = (js - J) * 64 + lane; lam = i < np ? LV_(i) : 0.0; }
      const bool lo = l > -INFINITY && lam > 0 && lam > fabs(v - l);
      const bool up = u < INFINITY && lam < 0 && -lam > fabs(u - v);
      PS[ix] = lo ? 1.0 : (up ? -1.0 : 0.0); PY[ix] = ((lo || up) && js < J) ? lam : 0.0;
    }
    for (int i = tid; i < np; i += NTH) XS[i] = X_(i);   // z of the refinement between attempts (the fall-back copy is no longer needed)
    __syncthreads();

    // fused pass of the refinement over my slots: q = A~ v (-> QV), pen = cf0 (q - cf1), y^ = cf2 - pen (-> YV),
    // P1 = A~' y^, P2 = A~' pen.  mode 0: evaluation at (z, y), cf = (PA, PB, PY); mode 1: CG step, cf = (PA, 0, 0).
    auto pass_fused = [&](int oV, int mode) AINL {
      double v[T], vb[NBB], pc[T], pd[T], pcb[NBB], pdb[NBB];
#pragma unroll
      for (int t = 0; t < T; ++t) { v[t] = slds[oV + 16 * t + c]; pc[t] = 0.0; pd[t] = 0.0; }
#pragma unroll
      for (int f = 0; f < NBB; ++f) { vb[f] = NB ? slds[oV + nc + f] : 0.0; pcb[f] = 0.0; pdb[f] = 0.0; }
      for (int js = w; js < J; js += W) {
        const int ix = js * 64 + lane;
        WAVE_SYNC();
        CWS_(0, lane) = PA[ix]; CWS_(1, lane) = mode == 0 ? PB[ix] : 0.0; CWS_(2, lane) = mode == 0 ? PY[ix] : 0.0;
        WAVE_SYNC();
        double kq = 0.0, ky = 0.0;
        slot_pairs(js, [&](const v2d (&bq)[T], int s0) AINL {
          const int rix = (s0 >> 4) * 64 + q * 16 + (s0 & 15), cix = q * 16 + (s0 & 15);
          const v2d c0 = *reinterpret_cast<const v2d*>(&CWS_(0, cix)), c1 = *reinterpret_cast<const v2d*>(&CWS_(1, cix)), c2 = *reinterpret_cast<const v2d*>(&CWS_(2, cix));
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            double dsum = 0.0;
#pragma unroll
            for (int t = 0; t < T; ++t) dsum = fma(bq[t][h], v[t], dsum);
            dsum = grp16_sum(dsum);
#pragma unroll
            for (int f = 0; f < NB; ++f) dsum = fma(AB_(f, rix + h), vb[f], dsum);
            const double pen = c0[h] * (dsum - c1[h]);
            const double ynew = c2[h] - pen;
            if (c == ((s0 + h) & 15)) { kq = dsum; ky = ynew; }
#pragma unroll
            for (int t = 0; t < T; ++t) { pc[t] = fma(ynew, bq[t][h], pc[t]); pd[t] = fma(pen, bq[t][h], pd[t]); }
#pragma unroll
            for (int f = 0; f < NB; ++f) { const double a_ = AB_(f, rix + h); pcb[f] = fma(ynew, a_, pcb[f]); pdb[f] = fma(pen, a_, pdb[f]); }
          }
        });
        QV[ix] = kq; YV[ix] = ky;
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const double pv = q_sum(pc[t]), dv_ = q_sum(pd[t]);
        if (q == 0) { slds[oWP + w * np + 16 * t + c] = pv; slds[oP2 + w * np + 16 * t + c] = dv_; }
      }
      {
        double pvb = 0.0, dvb = 0.0;
#pragma unroll
        for (int f = 0; f < NB; ++f) { const double s = q_sum(pcb[f]), s2_ = q_sum(pdb[f]); if (lane == f) { pvb = s; dvb = s2_; } }
        if (NB > 0 && lane < 16) { slds[oWP + w * np + nc + lane] = pvb; slds[oP2 + w * np + nc + lane] = dvb; }
      }
      __syncthreads();
      for (int i = tid; i < np; i += NTH) {
        double s = 0.0, s2_ = 0.0;
        for (int ww = 0; ww < W; ++ww) { s += slds[oWP + ww * np + i]; s2_ += slds[oP2 + ww * np + i]; }
        P1_(i) = s; P2_(i) = s2_;
      }
      __syncthreads();
    };
    // gradient of the augmented Lagrangian at (z, y): QV = A~z, YV = y - rho c, P1 = A~'y^, P2 = rho A_W'c, HX = H~z
    auto eval_zy = [&]() AINL { pass_fused(VEC(V_R2), 0); hx_keep(VEC(V_R2)); };
    // index of the variable row i in the owner-layout row arrays
    auto vix = [&](int i) AINL { return (J + (i >> 6)) * 64 + (i & 63); };

    for (int attempt = 0; attempt < QP_REFINE_ATTEMPTS && flag_polished <= 0; ++attempt) {
      for (int js = w; js < JT; js += W) {
        const int ix = js * 64 + lane;
        const double sd = PS[ix];
        const double tgt = sd > 0 ? aL[ix] : (sd < 0 ? aU[ix] : 0.0);
        PA[ix] = sd != 0.0 ? rho : 0.0; PB[ix] = tgt; PC[ix] = 0.0; PP[ix] = 0.0;
        if (js < J) { aD[ix] = sd != 0.0 ? rho : 0.0; aW1[ix] = 0.0; aW2[ix] = 0.0; }   // (W3 keeps the interior-point multipliers: they are returned if the refinement is rejected)
        else { const int i = (js - J) * 64 + lane; if (i < np) { DV_(i) = sd != 0.0 ? pin : 0.0; W1V_(i) = sd; W2V_(i) = tgt; } }
      }
      __syncthreads();
      acc_init();
      pass_syrk();
      for (int i = tid; i < np; i += NTH) { R1_(i) = 0.0; R2_(i) = 0.0; }
      __syncthreads();
      bool pok = factor_solve2(false) == 0, retry = false;
      if (!pok) { flag_polished = -5; break; }
      // z: the point reached so far, pinned variables on their bounds
      for (int i = tid; i < np; i += NTH) R2_(i) = (i < n && W1V_(i) != 0.0) ? W2V_(i) : XS[i];
      __syncthreads();
      eval_zy();
      for (int i = tid; i < np; i += NTH) DX_(i) = (i < n && W1V_(i) == 0.0) ? -(HX_(i) + G_(i) - P1_(i)) : 0.0;
      __syncthreads();
      solve1(VEC(V_DX));
      for (int i = tid; i < n; i += NTH) if (W1V_(i) == 0.0) R2_(i) += DX_(i);
      __syncthreads();
      eval_zy();   // c(z): QV = A~z; P2 = rho A_W'c
      double rs_l = 0.0;
      for (int js = w; js < J; js += W) {
        const int ix = js * 64 + lane;
        const double cc_ = PA[ix] != 0.0 ? QV[ix] - PB[ix] : 0.0;
        PC[ix] = cc_; PP[ix] = -cc_; rs_l = fma(cc_, cc_, rs_l);
      }
      red_put(0, wave_sum(rs_l));
      for (int i = tid; i < np; i += NTH) { const double a_ = -P2_(i) * rinv; R1_(i) = a_; P3_(i) = a_; }   // ATR, ATP
      __syncthreads();
      double rs = red_sum(0);
      red_next();
      for (int cgit = 0; cgit < 12 && pok; ++cgit) {
        double m_eq = 0.0, m_cy = 0.0;
        for (int js = w; js < J; js += W) {
          const int ix = js * 64 + lane;
          const double pc_ = PC[ix];
          m_eq = fmax(m_eq, fabs(pc_) / fmax(1.0, fabs(PB[ix])));
          m_cy = fmax(m_cy, fabs(pc_ * PY[ix]));
        }
        red_put(0, wave_max(m_eq)); red_put(1, wave_max(m_cy));
        __syncthreads();
        m_eq = red_max(0); m_cy = red_max(1);
        red_next();
        if (m_eq <= 1e-11 && m_cy <= 1e-11 * fmax(1.0, fabs(fval_s))) break;
        if (cgit == 11) { pok = false; flag_polished = -6; break; }
        for (int i = tid; i < np; i += NTH) DX_(i) = (i < n && W1V_(i) == 0.0) ? P3_(i) : 0.0;
        __syncthreads();
        solve1(VEC(V_DX));                            // w = M^-1 A_W'p
        pass_fused(VEC(V_DX), 1);                     // QV = A~w; P2 = rho A_W'(A_W w)
        double pq_l = 0.0;
        for (int js = w; js < J; js += W) { const int ix = js * 64 + lane; if (PA[ix] != 0.0) pq_l = fma(PP[ix], QV[ix], pq_l); }
        red_put(0, wave_sum(pq_l));
        __syncthreads();
        const double pq = red_sum(0);
        red_next();
        if (!(pq > 0.0) || !(rs > 0.0)) {   // dependent / inconsistent working set: drop the row that carries the stalled direction
          pok = false; flag_polished = -7;
          if (attempt < QP_REFINE_ATTEMPTS - 1) {
            double my = 0.0; int myix = -1;
            for (int js = w; js < J; js += W) { const int ix = js * 64 + lane; if (PA[ix] != 0.0 && fabs(PP[ix]) > my) { my = fabs(PP[ix]); myix = ix; } }
            red_put(0, wave_max(my));
            __syncthreads();
            const double mx = red_max(0);
            red_next();
            if (mx > 0.0 && my == mx && myix >= 0) { PS[myix] = 0.0; PY[myix] = 0.0; }
            for (int i = tid; i < np; i += NTH) XS[i] = R2_(i);
            __syncthreads();
            retry = true;
          }
          break;
        }
        const double alpha_ = rs / pq;
        double rsn_l = 0.0;
        for (int js = w; js < J; js += W) {
          const int ix = js * 64 + lane;
          if (PA[ix] != 0.0) {
            PY[ix] = fma(alpha_, PP[ix], PY[ix]);
            const double cc_ = fma(alpha_, QV[ix], PC[ix]);
            PC[ix] = cc_; rsn_l = fma(cc_, cc_, rsn_l);
          }
        }
        red_put(0, wave_sum(rsn_l));
        __syncthreads();
        const double rsn = red_sum(0);
        red_next();
        const double beta_ = rsn / rs;
        for (int js = w; js < J; js += W) { const int ix = js * 64 + lane; if (PA[ix] != 0.0) PP[ix] = fma(beta_, PP[ix], -PC[ix]); }
        for (int i = tid; i < np; i += NTH) {
          if (i < n && W1V_(i) == 0.0) R2_(i) = fma(alpha_, DX_(i), R2_(i));
          const double atr = fma(-alpha_ * rinv, P2_(i), R1_(i));
          R1_(i) = atr;
          P3_(i) = fma(beta_, P3_(i), atr);
        }
        rs = rsn;
        __syncthreads();
      }
      if (pok) {
        // fresh evaluation of the candidate (z, y): everything recomputed from a stream over A~ and H~
        eval_zy();
        double m_rd = 0, m_rp = 0, m_sg = 0, m_cp = 0, fl2 = 0;
        for (int i = tid; i < n; i += NTH) {
          const int ix = vix(i);
          const double r = HX_(i) + G_(i) - P1_(i);                   // free variable: must vanish; pinned variable: its bound multiplier
          const double sc = fmax(1.0, fmax(fabs(G_(i)), fmax(fabs(HX_(i)), fabs(P1_(i)))));
          const double sd = W1V_(i), zi = R2_(i), l = aL[ix], u = aU[ix];
          if (sd == 0.0) m_rd = fmax(m_rd, fabs(r) / sc);
          else m_sg = fmax(m_sg, (sd > 0 ? -r : r) / sc);
          YV[ix] = sd != 0.0 ? r : 0.0;                               // multiplier of the variable-bound row
          double viol = 0.0;
          if (l > -INFINITY && zi < l) viol = l - zi;
          if (u < INFINITY && zi > u) viol = fmax(viol, zi - u);
          m_rp = fmax(m_rp, viol / fmax(1.0, fabs(zi)));
          fl2 += 0.5 * zi * HX_(i) + G_(i) * zi + 0.0 * r;   // (0 * r: a non-finite residual must poison the sum -- fmax drops NaN operands)
        }
        for (int js = w; js < J; js += W) {
          const int ix = js * 64 + lane;
          const double l = aL[ix], u = aU[ix], v = QV[ix], y = YV[ix], sd = PS[ix];
          if (l > -INFINITY || u < INFINITY) {   // (padding rows carry infinite bounds)
            double sc = fmax(1.0, fabs(v));
            if (l > -INFINITY) sc = fmax(sc, fabs(l));
            if (u < INFINITY) sc = fmax(sc, fabs(u));
            double viol = 0.0;
            if (sd != 0.0) { viol = fabs(v - PB[ix]); m_cp = fmax(m_cp, fabs(y) * viol); }
            if (l > -INFINITY && v < l) viol = fmax(viol, l - v);
            if (u < INFINITY && v > u) viol = fmax(viol, v - u);
            m_rp = fmax(m_rp, viol / sc);
            m_sg = fmax(m_sg, (sd > 0 ? -y : (sd < 0 ? y : 0.0)) / fmax(1.0, fabs(y)));
            fl2 += 0.0 * (v + y);
          }
        }
        red_put(0, wave_max(m_rd)); red_put(1, wave_max(m_rp)); red_put(2, wave_max(m_sg)); red_put(3, wave_max(m_cp)); red_put(4, wave_sum(fl2));
        __syncthreads();
        m_rd = red_max(0); m_rp = red_max(1); m_sg = red_max(2); m_cp = red_max(3);
        const double f2 = red_sum(4);
        red_next();
        // acceptance: relative stationarity 1e-8, feasibility and complementarity 1e-10, multipliers of the right sign (the same
        // thresholds as the one-wavefront kernel)
        pok = m_rd <= 1e-8 && m_rp <= 1e-10 && m_cp <= 1e-10 * fmax(1.0, fabs(f2)) && m_sg <= 1e-8 && fabs(f2) < INFINITY;   // (f2 is NaN if anything in the candidate is not finite)
        if (!pok) flag_polished = !(fabs(f2) < INFINITY) ? -5 : (!(m_rd <= 1e-8) ? -1 : (!(m_rp <= 1e-10) ? -2 : (!(m_sg <= 1e-8) ? -4 : -3)));
        if (!pok && m_rd <= 1e-8 && attempt < QP_REFINE_ATTEMPTS - 1 && (m_rp > 1e-10 || m_sg > 1e-8)) {
          // single correction of the working set: add the most violated inactive row, else drop the worst wrong-sign row
          double my = 0.0; int myix = -1; double myside = 0.0;
          const bool add = m_rp > 1e-10;
          for (int js = w; js < JT; js += W) {
            const int ix = js * 64 + lane;
            const double l = aL[ix], u = aU[ix], sd = PS[ix];
            if (!(l > -INFINITY || u < INFINITY)) continue;
            double v;
            if (js < J) v = QV[ix]; else { const int i = (js - J) * 64 + lane; v = i < np ? R2_(i) : 0.0; }
            if (add) {
              if (sd != 0.0) continue;
              double sc = fmax(1.0, fabs(v));
              if (js < J) { if (l > -INFINITY) sc = fmax(sc, fabs(l)); if (u < INFINITY) sc = fmax(sc, fabs(u)); }
              const double vl = l > -INFINITY ? (l - v) / sc : -1.0, vu = u < INFINITY ? (v - u) / sc : -1.0;
              const double vv = fmax(vl, vu);
              if (vv > my) { my = vv; myix = ix; myside = vl >= vu ? 1.0 : -1.0; }
            } else {
              if (sd == 0.0) continue;
              const double y = YV[ix];
              const double sc = js < J ? fmax(1.0, fabs(y)) : 1.0;
              const double sg = (sd > 0 ? -y : y) / sc;
              if (sg > my) { my = sg; myix = ix; myside = 0.0; }
            }
          }
          red_put(0, wave_max(my));
          __syncthreads();
          const double mx = red_max(0);
          red_next();
          if (mx > 0.0 && my == mx && myix >= 0) { PS[myix] = myside; PY[myix] = 0.0; }
          __syncthreads();
          for (int js = w; js < J; js += W) { const int ix = js * 64 + lane; PY[ix] = PS[ix] != 0.0 ? PY[ix] : 0.0; }
          for (int i = tid; i < np; i += NTH) XS[i] = R2_(i);
          __syncthreads();
          continue;   // next attempt from the point reached (R2) with the corrected working set
        }
        if (!pok && !(m_rd <= 1e-8) && attempt < QP_REFINE_ATTEMPTS - 1 && m_rd <= 1e-4) {   // stationarity above the floor: one more exact step from here
          for (int i = tid; i < np; i += NTH) XS[i] = R2_(i);
          for (int js = w; js < J; js += W) { const int ix = js * 64 + lane; PY[ix] = PS[ix] != 0.0 ? YV[ix] : 0.0; }
          __syncthreads();
          continue;
        }
        if (!pok) break;
        for (int i = tid; i < np; i += NTH) X_(i) = R2_(i);
        for (int js = w; js < JT; js += W) {
          const int ix = js * 64 + lane;
          const double y = YV[ix], sd = PS[ix];
          const double lam = sd > 0 ? fmax(y, 0.0) : (sd < 0 ? fmin(y, 0.0) : 0.0);
          if (js < J) aW3[ix] = lam; else { const int i = (js - J) * 64 + lane; if (i < np) LV_(i) = lam; }
        }
        flag_polished = 1 + attempt;
        fval_s = f2; merit_s = fmax(m_rd, fmax(m_rp, m_cp / fmax(1.0, fabs(f2))));
        flag = 0;
        __syncthreads();
      } else if (!retry) break;
    }   // attempts
    __syncthreads();
  }
  if (flag == 4) flag = -1;   // not certified
  if (flag == 5) flag = 1;

  // ---- outputs ----
  double* xo = P.x + (size_t)b * d.nu;       // caller's indexing (QpDims::nu: dummy padding variables are skipped)
  for (int i = tid; i < n; i += NTH) { const int ui = qp_user_index(d, i); if (ui >= 0) xo[ui] = X_(i) * EV_(i); }
  if (P.lambda) {
    double* lo = P.lambda + (size_t)b * (d.nu + m);
    const double* __restrict__ Fs = ws + d.off_F;
    for (int js = w; js < JT; js += W) {
      if (js < J) {
        const int r = perm[js * 64 + lane];   // original row of this sorted position
        if (r >= 0) lo[d.nu + r] = aW3[js * 64 + lane] * Fs[js * 64 + lane];
      } else {
        const int i = (js - J) * 64 + lane;
        const int ui = i < n ? qp_user_index(d, i) : -1;
        if (ui >= 0) lo[ui] = LV_(i) / EV_(i);
      }
    }
  }
  if (!(v_current || flag_polished > 0)) {  // objective at the returned point (H~, g~ scaling is objective preserving)
    hx_keep(VEC(V_X));
    double fl = 0;
    for (int i = lane; i < n; i += 64) fl += 0.5 * X_(i) * HX_(i) + G_(i) * X_(i);
    fval_s = wave_sum(fl);
  }
  STAMP(14);
  STAMP_OUT;
  if (tid == 0) {
    P.fval[b] = fval_s;
    P.exitflag[b] = flag;
    P.iter[b] = it;
    if (P.polished) P.polished[b] = flag_polished;
    if (P.kkt) P.kkt[b] = merit_s;
  }
}

}  // namespace

template <int T, int NB, int W> static hipError_t launch_wg(const QpParams& P, int batch, hipStream_t st) {
  const size_t lds = qp_wg_lds_base_bytes(P.d, W, NB);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qp_wg_kernel<T, NB, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((qp_wg_kernel<T, NB, W>), dim3(batch), dim3(64 * W), lds, st, P);
  return hipGetLastError();
}
template <int T, int NB> static hipError_t launch_wg_T(const QpParams& P, int batch, hipStream_t st) { return launch_wg<T, NB, QP_WG_W>(P, batch, st); }

#ifndef QP_WG_TLO
#define QP_WG_TLO 1
#define QP_WG_THI 12
#endif
// instantiated tile counts of this translation unit: [QP_WG_TLO, QP_WG_THI]; border widths 0 and 4 (bordered shapes up to T = 5
// run on the one-wavefront kernel, qp_solver.hip; from T = 6 on 1..4 border columns run the 4-column variant, padded unit columns)
template <int T> static hipError_t launch_wg_sel(const QpParams& P, int batch, hipStream_t st) {
  if constexpr (T >= QP_WG_TLO && T <= QP_WG_THI) {
    if (P.d.T == T) {
#ifdef QP_WG_ONLY_NB
      if (P.d.NBk == QP_WG_ONLY_NB) return launch_wg_T<T, QP_WG_ONLY_NB>(P, batch, st);
      return hipErrorInvalidValue;
#else
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
#ifndef QP_WG_SYM
#define QP_WG_SYM QP_WG_TLO
#endif
hipError_t QP_WG_CAT(qp_wg_launch_, QP_WG_SYM)(const QpParams& P, int batch, hipStream_t st) { return launch_wg_sel<1>(P, batch, st); }
#endif
