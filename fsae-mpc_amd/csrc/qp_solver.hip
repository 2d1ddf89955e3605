// qp_solver.hip -- batched dense convex QP solve on MI355X (gfx950), one wavefront per QP.
//
// Replaces the qpOASES MEX call of the reference
//   (mpc/ltv/kinematic/ltvmpc_kinetmatic_curvilinear.m:52, mpc/ltv/dynamic/ltvmpc_dynamic_curvilinear.m:52,
//    contract optimizers/matlab/qpOASES/qpOASES.m:16-62)
// with a primal-dual interior-point method (Mehrotra predictor-corrector, single step length,
// OOQP-style step heuristic).  Not a port of qpOASES: the algorithm is chosen for the hardware --
// every iteration is one pass of fp64 MFMA (v_mfma_f64_16x16x4_f64) forming M = H + A'DA with the
// accumulators resident in registers, plus three light streaming passes over A.
//
// Data layout (all per QP, in the device workspace, written once by qp_prep_kernel):
//   rows             are permuted once (qp_prep_kernel): sorted by the last column tile that holds a nonzero, then dealt
//                    round-robin to the four lane groups: sorted position p <-> k-step s = p>>2, lane group q = p&3.
//                    A *trip* = 4 k-steps = 16 sorted rows; tcs[trip] = number of leading column tiles that hold a
//                    nonzero in any of them.  The condensed LTV-MPC constraints are block lower-triangular (row k only
//                    sees the inputs up to step k), so on average ~60 % of the tiles and ~47 % of the MFMAs remain; a
//                    dense A keeps tcs = T everywhere and costs nothing extra.
//   Aw               scaled A in MFMA-operand stream order: trip, pair of k-steps u, tile t < tcs[trip], lane, 2 k-steps
//                    (16-byte lane loads); lane (c=l&15,q=l>>4) of k-step s holds A~[perm[4s+q]][16t + c].
//                    aoff[trip] = start of the trip in records of 128 doubles; tend[C] = number of trips with tcs <= C.
//   Hw [T*T][4][64]  scaled H in accumulator (C/D) layout: tile (I,J), reg p, lane -> H~[16I+q+4p][16J+c]
//   row vectors      "owner layout" [slot][64]: slot js<J, lane (c,q) <-> k-step s = 16js + c, sorted position 4s + q;
//                    slots J..J+JB-1 hold the variable-bound rows i = (js-J)*64 + lane.
// fp64 MFMA lane maps (cdna_hip_programming.md section 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// C/D col = l&15, row = (l>>4) + 4*reg.  fsaempc_selftest_mfma() checks them on the device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include "qp_solver.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define DEVINL __device__ __forceinline__
#ifndef QP_SOLVE_REFINE_STEPS
#define QP_SOLVE_REFINE_STEPS 1   // refinement steps of the diagonal-block solves in vec_forward / vec_backward
#endif
// The solve kernel runs one wavefront per workgroup: its lanes exchange data through LDS (and their own rows of global
// memory) in program order, and the hardware keeps the DS / vector-memory operations of one wave in order, so a
// workgroup barrier (s_barrier + full s_waitcnt drain) is not needed -- a compiler-level fence is.
#ifndef QP_FULL_BARRIERS
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define WAVE_SYNC() __syncthreads()
#endif

namespace {

DEVINL double rl(double v, int src) {  // wave-uniform broadcast of lane `src` (src must be wave-uniform)
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int CTRL> DEVINL double dpp_f64(double v) {  // data-parallel-primitive lane move of both halves (VALU speed, no LDS)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
DEVINL double grp16_sum(double v) {  // sum over the 16 lanes sharing l>>4 (one DPP row); every lane gets the total
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);   // row_half_mirror
  v += dpp_f64<0x140>(v);   // row_mirror
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
// Exchange between the four 16-lane rows of a wave with the gfx950 lane-swap instructions (VALU speed; the ds_bpermute
// round trips of __shfl_xor cost ~100 cycles each and a wave reduction needed twelve of them):
//   v_permlane16_swap a, b : a.row1 <-> b.row0, a.row3 <-> b.row2     v_permlane32_swap a, b : a.rows23 <-> b.rows01
// with a = b = v on entry the two results are (row0,row0,row2,row2) / (row1,row1,row3,row3) resp. (rows01 x2) / (rows23 x2).
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
DEVINL double q_max(double v) { RowPair r = rows_xor16(v); v = fmax(r.a, r.b); r = rows_xor32(v); return fmax(r.a, r.b); }
DEVINL double q_min(double v) { RowPair r = rows_xor16(v); v = fmin(r.a, r.b); r = rows_xor32(v); return fmin(r.a, r.b); }
// whole-wave reductions: DPP within the four 16-lane rows, lane swaps across them (no LDS round trips)
DEVINL double wave_sum(double v) { return q_sum(grp16_sum(v)); }
DEVINL double wave_max(double v) { return q_max(grp16_max(v)); }
DEVINL double wave_min(double v) { return q_min(grp16_min(v)); }

template <int T> struct Tri {
  static constexpr int NT = T * (T + 1) / 2;
  __host__ __device__ static constexpr int idx(int I, int J) { return I * T - (I * (I - 1)) / 2 + (J - I); }
};

// Translation units: the solve kernel is instantiated for T = 1..7 and three border widths; the Makefile builds this
// file five times (-DQP_TU=0: prep kernel, dimensions, dispatch, self test; -DQP_TU=1: T = 1..4; -DQP_TU=2..4: T = 5..7).
// Without QP_TU everything lands in one object (used by the one-command diagnostic builds).
#if !defined(QP_TU) || QP_TU == 0
#define QP_MAIN_TU 1
#else
#define QP_MAIN_TU 0
#endif

#if QP_MAIN_TU
// ---------------------------------------------------------------------------------------------
// qp_prep_kernel: scaling (E columns, F rows), repack of A and H, scaled g / bounds.  One wave per QP.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void qp_prep_kernel(QpParams P) {
  // one workgroup per QP: 256 threads for small row counts, 1024 for large ones (qp_launch_prep) -- the staging tile allows one
  // workgroup per CU there, and with four wavefronts the dependent LDS -> global chains of the repack ran at a tenth of the
  // memory rate (11 ms of a 145 ms dynamic N = 60 batch)
  const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15, q = lane >> 4;
  const int NTH = blockDim.x, NW = NTH >> 6;
  const QpDims& d = P.d;
  const int n = d.n, nu = d.nu, m = d.m, T = d.T, Kq = d.Kq, J = d.J, JB = d.JB, np = d.np, nc = d.nc, nb = d.nb;
  const double* H = P.H + (P.shared_HA ? 0 : (size_t)b * nu * nu);
  const double* A = P.A + (P.shared_HA ? 0 : (size_t)b * m * nu);
  const double* g = P.g + (size_t)b * nu;
  // caller's data by SOLVER index (n = nu unless the core is padded with dummy variables, QpDims::nu): dummies have a unit Hessian
  // diagonal, nothing else
  auto U = [&](int i) { return qp_user_index(d, i); };
  auto Hat = [&](int i, int j) -> double { const int ui = U(i), uj = U(j); return (ui >= 0 && uj >= 0) ? H[(size_t)uj * nu + ui] : ((i == j && i < n) ? 1.0 : 0.0); };
  auto Aat = [&](int r, int j) -> double { const int uj = U(j); return uj >= 0 ? A[(size_t)uj * m + r] : 0.0; };
  double* ws = P.ws + (size_t)b * d.ws_per_qp;
  double* Aw = ws + d.off_Aw;
  double* Hw = ws + d.off_Hw;
  double* gw = ws + d.off_gw;
  double* Es = ws + d.off_E;
  double* Fs = ws + d.off_F;   // owner layout, J slots
  double* Ab = ws + d.off_Ab;  // 4 border columns of A~, owner layout
  double* Hb = ws + d.off_Hb;  // 4 border columns of H~ (full length np)
  int* perm_g = reinterpret_cast<int*>(ws + d.off_meta);          // owner layout: original row of each sorted position (-1: padding)
  int* tcs_g = perm_g + (size_t)(J > 0 ? J : 1) * 64;              // [ntr] tiles per trip
  int* aoff_g = tcs_g + d.ntr;                                     // [ntr+1] start of each trip in the operand stream
  int* tend_g = aoff_g + d.ntr + 1;                                // [T+1] trips with tcs <= C
  const int ntr = d.ntr;
  double* Lr = ws + d.off_rows + 0 * (size_t)d.rowlen;  // scaled lower bounds (owner layout, rows then vars)
  double* Ur = ws + d.off_rows + 1 * (size_t)d.rowlen;
  extern __shared__ double lds[];
  double* Esh = lds;            // np
  double* red = lds + np;       // 16 partial maxima
  double* tile = red + 16;       // 16 x (4Kq+1) staging for the A transpose
  const int TW = d.prep_tw;     // columns staged per pass (16, or fewer when 16 x 4Kq doubles would not fit the LDS)
  int* cls_sh = reinterpret_cast<int*>(tile + TW * (4 * Kq + 1));   // [4Kq] tile class of every original row
  int* perm_sh = cls_sh + 4 * Kq;                                    // [16 ntr] original row of every sorted position
  int* cnt_sh = perm_sh + 16 * ntr;                                  // [T+1] class counts / starts ; then tcs [ntr], aoff [ntr+1]
  int* tcs_sh = cnt_sh + 16;
  int* aoff_sh = tcs_sh + ntr;
  int* chc_sh = aoff_sh + ntr + 1;                                   // [ceil(m/64)][16] rows of each class per 64-row chunk, then their starts
  double* Fr_sh = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(chc_sh + ((m + 63) >> 6) * 16) + 7) & ~(uintptr_t)7);   // [m] row scale of every original row

  // ---- column scaling E_j = 1/sqrt(H_jj), or 1/max|A_:j| where H_jj ~ 0 (slack columns) ----
  for (int j = tid; j < np; j += NTH) {
    double e = 1.0;
    if (j < n) {
      double hjj = Hat(j, j);
      if (hjj > 1e-12) e = 1.0 / sqrt(hjj);
      else e = -1.0;  // resolved below with a cooperative column max
    } else e = 0.0;   // padded columns carry zeros
    Esh[j] = e;
  }
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    if (Esh[j] < 0) {  // uniform over the workgroup
      double cm = 0;
      for (int r = tid; r < m; r += NTH) cm = fmax(cm, fabs(Aat(r, j)));
      cm = wave_max(cm);
      if (lane == 0) red[w] = cm;
      __syncthreads();
      if (tid == 0) { double mx = 0.0; for (int i = 0; i < NW; ++i) mx = fmax(mx, red[i]); Esh[j] = mx > 1e-12 ? 1.0 / mx : 1.0; }
      __syncthreads();
    }
  }
  int bad = 0;   // NaN / Inf anywhere in H, g, A or NaN in a bound: the solve kernel answers -1 before its first iteration (the
                 // reference's MEX gateway rejects such a call; a device entry cannot look at the data before the launch)
  for (int j = tid; j < np; j += NTH) { const int uj = j < n ? U(j) : -1; const double gj = uj >= 0 ? g[uj] : 0.0; Es[j] = Esh[j]; gw[j] = gj * Esh[j]; bad |= !(fabs(gj) < INFINITY); }

  // ---- one pass over A, thread = row (coalesced: consecutive threads read consecutive rows of a column, the loads of a thread are
  //      independent): row scaling F_r = 1 / max_j |A[r][j] E_j| and the row's class = last core column tile with a nonzero.
  //      (Round 2 scanned every row twice with dependent / uncoalesced loads and ranked the rows with an O(m^2) loop: 14 ms of
  //      the 200 ms of a dynamic N = 60 batch.) ----
  __syncthreads();   // Esh complete
  const int ncols = nc < n ? nc : n;
  const int nchunk = (m + 63) >> 6;
  for (int i = tid; i < nchunk * 16; i += NTH) chc_sh[i] = 0;
  if (tid < 16) cnt_sh[tid] = 0;
  __syncthreads();
  for (int r0 = 0; r0 < m; r0 += NTH) {
    const int r = r0 + tid;
    int e = 0;
    double rm = 0.0;
    if (r < m) {
      // (eight columns of loads in flight: with one load per iteration the loop ran at one memory latency per column, 30 % of
      //  this kernel on the headline shape)
      for (int col0 = 0; col0 < n; col0 += 8) {
        double av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = col0 + u < n ? Aat(r, col0 + u) : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int col = col0 + u;
          const double a = fabs(av[u]);
          if (col < n) { rm = fmax(rm, a * Esh[col]); if (a != 0.0 && col < ncols) e = col >> 4; }
        }
      }
      cls_sh[r] = e;
      Fr_sh[r] = rm > 1e-12 ? 1.0 / rm : 1.0;
    }
    // stable counting sort by class, part 1: rows of each class in this 64-row chunk (the lanes of a wave hold consecutive rows)
    for (int cl = 0; cl < T; ++cl) {
      const unsigned long long mk = __ballot(r < m && e == cl);
      if (lane == 0 && r0 + 64 * w < m) chc_sh[((r0 >> 6) + w) * 16 + cl] = __popcll(mk);
    }
  }
  __syncthreads();
  if (tid < T) {   // per class: running start of every chunk (after the total of the lower classes is known: two steps)
    int tot = 0;
    for (int ch = 0; ch < nchunk; ++ch) tot += chc_sh[ch * 16 + tid];
    cnt_sh[tid] = tot;
  }
  __syncthreads();
  if (tid < T) {
    int start = 0;
    for (int cl = 0; cl < tid; ++cl) start += cnt_sh[cl];
    for (int ch = 0; ch < nchunk; ++ch) { const int cn = chc_sh[ch * 16 + tid]; chc_sh[ch * 16 + tid] = start; start += cn; }
  }
  for (int p = tid; p < 16 * ntr; p += NTH) perm_sh[p] = -1;
  __syncthreads();
  for (int r0 = 0; r0 < m; r0 += NTH) {   // part 2: position = start of (chunk, class) + rank among the chunk's earlier rows of the class
    const int r = r0 + tid;
    const int e = r < m ? cls_sh[r] : -1;
    for (int cl = 0; cl < T; ++cl) {
      const unsigned long long mk = __ballot(e == cl);
      if (e == cl) perm_sh[chc_sh[((r0 >> 6) + w) * 16 + cl] + __popcll(mk & ((1ull << lane) - 1ull))] = r;
    }
  }
  __syncthreads();
  // tiles per trip (16 sorted positions), stream offsets, phase ends
  for (int tr = tid; tr < ntr; tr += NTH) {
    int tc = 1;
    for (int p = 16 * tr; p < 16 * tr + 16; ++p) { const int r = perm_sh[p]; if (r >= 0) tc = max(tc, cls_sh[r] + 1); }
    tcs_sh[tr] = tc;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int tr = 0; tr < ntr; ++tr) { aoff_sh[tr] = run; run += 2 * tcs_sh[tr]; }
    aoff_sh[ntr] = run;
    for (int C = 0; C <= T; ++C) { int cn = 0; for (int tr = 0; tr < ntr; ++tr) cn += (tcs_sh[tr] <= C); tend_g[C] = cn; }
  }
  __syncthreads();
  for (int tr = tid; tr <= ntr; tr += NTH) { aoff_g[tr] = aoff_sh[tr]; if (tr < ntr) tcs_g[tr] = tcs_sh[tr]; }

  // ---- row scaling F_r = 1/max_j |A[r][j] E_j| ; rows handled in owner layout, one slot per wavefront and trip ----
  int nexcl = 0;   // rows / bounds that exclude x = 0: the difficulty estimate behind the launch order (QpParams::order)
  for (int js = w; js < J; js += NW) {
    const int s = 16 * js + c, pp = 4 * s + q;
    const int r = (s < 4 * ntr) ? perm_sh[pp] : -1;
    const bool valid = r >= 0;
    const double f = valid ? Fr_sh[r] : 0.0;
    Fs[js * 64 + lane] = f;
    perm_g[js * 64 + lane] = r;
    for (int bb = 0; bb < 4; ++bb)
    { const double v = (valid && bb < nb) ? Aat(r, nc + bb) * Esh[nc + bb] * f : 0.0; bad |= !(fabs(v) < INFINITY); Ab[(size_t)bb * J * 64 + js * 64 + lane] = v; }
    double l = -INFINITY, u = INFINITY;
    if (valid) {
      double lr = P.lbA[(size_t)b * m + r], ur = P.ubA[(size_t)b * m + r];
      bad |= (lr != lr) || (ur != ur);
      nexcl += (lr > 0.0) || (ur < 0.0);
      l = lr > -P.inf_bound ? lr * f : -INFINITY;
      u = ur < P.inf_bound ? ur * f : INFINITY;
    }
    Lr[js * 64 + lane] = l;
    Ur[js * 64 + lane] = u;
  }
  for (int jb = w; jb < JB; jb += NW) {
    const int i = jb * 64 + lane;
    double l = -INFINITY, u = INFINITY;
    const int ui = i < n ? U(i) : -1;
    if (ui >= 0) {
      double lr = P.lb[(size_t)b * nu + ui], ur = P.ub[(size_t)b * nu + ui];
      bad |= (lr != lr) || (ur != ur);
      nexcl += (lr > 0.0) || (ur < 0.0);
      l = lr > -P.inf_bound ? lr / Esh[i] : -INFINITY;
      u = ur < P.inf_bound ? ur / Esh[i] : INFINITY;
    }
    Lr[(J + jb) * 64 + lane] = l;
    Ur[(J + jb) * 64 + lane] = u;
  }
  __syncthreads();  // Fs visible to the whole workgroup (global memory, same CU)

  // ---- A -> operand stream.  Column tile by column tile: coalesced column reads -> LDS -> lane order ----
  const int R4 = 4 * Kq, mp1 = R4 + 1;  // padded LDS row length (odd => conflict-free across the 16 columns)
  for (int t = 0; t < T; ++t)
    for (int c0 = 0; c0 < 16; c0 += TW) {
      for (int e = tid; e < TW * R4; e += NTH) {
        const int cc = e / R4, r = e - cc * R4;
        const int col = 16 * t + c0 + cc;
        double v = 0.0;
        if (col < nc && col < n && r < m) v = Aat(r, col) * Esh[col];
        bad |= !(fabs(v) < INFINITY);
        tile[cc * mp1 + r] = v;
      }
      __syncthreads();
      for (int s = w; s < 4 * ntr; s += NW) {   // k-steps are stored in pairs (16 B per lane and load); padded positions carry zeros
        const int tr = s >> 2, tc = tcs_sh[tr];
        if (t < tc && c >= c0 && c < c0 + TW) {
          const int r = perm_sh[4 * s + q];
          const double v = r >= 0 ? tile[(c - c0) * mp1 + r] * Fr_sh[r] : 0.0;   // (row scale from LDS: the owner-layout copy in global memory cost a load latency per k-step)
          Aw[((size_t)aoff_sh[tr] + ((s >> 1) & 1) * tc + t) * 128 + lane * 2 + (s & 1)] = v;
        }
      }
      __syncthreads();
    }

  // ---- H -> accumulator-layout tiles (T x T grid of the core; symmetric read for coalescing) ----
  for (int idx = w; idx < T * T * 4; idx += NW) {
    const int p = idx & 3, IJ = idx >> 2, I = IJ / T, Jt = IJ - I * T;
    const int row = 16 * I + q + 4 * p, col = 16 * Jt + c;
    double v = 0.0;
    if (row < nc && col < nc && row < n && col < n) v = Hat(col, row) * Esh[row] * Esh[col];  // H[col][row] == H[row][col]
    bad |= !(fabs(v) < INFINITY);
    Hw[(size_t)idx * 64 + lane] = v;
  }
  for (int e = tid; e < 4 * np; e += NTH) {
    const int bb = e / np, i = e - bb * np;
    const double v = (bb < nb && i < n) ? Hat(i, nc + bb) * Esh[nc + bb] * Esh[i] : 0.0;
    bad |= !(fabs(v) < INFINITY);
    Hb[e] = v;
  }
  bad = __syncthreads_or(bad);
  if (tid == 0) ws[d.off_bad] = bad ? 1.0 : 0.0;
  if (P.score) {
    __shared__ int score_sh;
    if (tid == 0) score_sh = 0;
    __syncthreads();
    if (nexcl) atomicAdd(&score_sh, nexcl);
    __syncthreads();
    if (tid == 0) P.score[b] = P.score_in ? P.score_in[b] : score_sh;
  }
}

// ---------------------------------------------------------------------------------------------
// qp_order_kernel: order[] = instance ids by descending score, ties in index order (a stable counting sort by one workgroup:
// scores clipped to 1023; eight wavefronts each count and place a contiguous range of the ids, no atomics, so the order -- and
// with it the timing of a batch -- is the same every run)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void qp_order_kernel(const int* __restrict__ score, int* __restrict__ order, int batch) {
  constexpr int NBIN = 1024, NWV = 8;   // (10-bit keys: group() below)
  __shared__ int cnt[NWV][NBIN], scan[NBIN];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  for (int e = tid; e < NWV * NBIN; e += 1024) (&cnt[0][0])[e] = 0;
  __syncthreads();
  const int per = ((batch + NWV - 1) / NWV + 63) & ~63;   // ids per counting wavefront, a multiple of 64
  const int lo = w * per, hi = min(batch, lo + per);
  auto bin = [&](int i) { return NBIN - 1 - min(max(score[i], 0), NBIN - 1); };   // bin 0 = highest score
  // rank of this lane among the lanes of its wavefront holding the same bin (lower lanes first), and that group's size
  auto group = [&](int k, bool on, int& rank, int& size) {
    unsigned long long same = __builtin_amdgcn_ballot_w64(on);   // lanes holding the same bin: one ballot per key bit
#pragma unroll
    for (int bit = 0; bit < 10; ++bit) {
      const bool one = (k >> bit) & 1;
      const unsigned long long v = __builtin_amdgcn_ballot_w64(one);
      same &= one ? v : ~v;
    }
    rank = __popcll(same & ((1ull << lane) - 1ull)); size = __popcll(same);
  };
  // (eight rounds of ids per trip: their score loads are in flight together -- one dependent load per round made this kernel 44 us)
  if (w < NWV)
    for (int i0 = lo; i0 < hi; i0 += 512) {
      int kv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) { const int i = i0 + 64 * r + lane; kv[r] = i < hi ? bin(i) : -1; }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const bool on = kv[r] >= 0; const int k = on ? kv[r] : 0;
        int rank, size; group(k, on, rank, size);
        if (on && rank == 0) cnt[w][k] += size;
      }
    }
  __syncthreads();
  int v = 0;
  for (int ww = 0; ww < NWV; ++ww) v += cnt[ww][tid];
  scan[tid] = v;
  __syncthreads();
  for (int o = 1; o < NBIN; o <<= 1) {
    const int a = tid >= o ? scan[tid - o] : 0;
    __syncthreads();
    scan[tid] += a;
    __syncthreads();
  }
  int run = scan[tid] - v;   // start of bin tid; then the start of each wavefront's share of it
  for (int ww = 0; ww < NWV; ++ww) { const int c = cnt[ww][tid]; cnt[ww][tid] = run; run += c; }
  __syncthreads();
  if (w < NWV)
    for (int i0 = lo; i0 < hi; i0 += 512) {
      int kv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) { const int i = i0 + 64 * r + lane; kv[r] = i < hi ? bin(i) : -1; }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const bool on = kv[r] >= 0; const int k = on ? kv[r] : 0;
        int rank, size; group(k, on, rank, size);
        int base = on ? cnt[w][k] : 0;
        if (on) order[base + rank] = i0 + 64 * r + lane;
        if (on && rank == 0) cnt[w][k] = base + size;   // (one writer per bin and round; read above by every lane of the group first)
      }
    }
}

#endif  // QP_MAIN_TU

// ---------------------------------------------------------------------------------------------
// solve kernel
// ---------------------------------------------------------------------------------------------
struct Ctx {
  int n, m, T, Kq, J, JB, JT, np, ld, lane, c, q, nc, nb, ntr;
  const int* perm; const int* tcs; const int* aoff; const int* tend;   // row order and operand-stream directory (qp_prep_kernel)
  const double* Aw; const double* Hw; const double* Ab; const double* Hb;
  double* rows;  // base of owner-layout row arrays
  double* ring;  // LDS: operand ring of the streaming passes (records of 1 KB)
  double* cof;   // LDS: per-slot coefficient staging of the streaming passes ([array][64])
  int rowlen;
  double* Ms;    // LDS n x ld
  double* vec;   // LDS n-vectors, np each
};
// (enum RowArr: qp_solver.h, shared with the workgroup kernel)
enum VecArr { V_X = 0, V_G, V_HX, V_R1, V_R2, V_P1, V_P2, V_P3, V_DX, V_E, V_NARR };   // + 4 border-column vectors MB[b] behind them

DEVINL double* rowp(const Ctx& k, int arr) { return k.rows + (size_t)arr * k.rowlen; }
DEVINL double* vecp(const Ctx& k, int arr) { return k.vec + arr * k.np; }

DEVINL bool row_valid(const Ctx& k, int js) {
  if (js < k.J) { const int s = 16 * js + k.c; return s < k.Kq && 4 * s + k.q < k.m; }
  return (js - k.J) * 64 + k.lane < k.n;
}

// Hx through the full symmetric tile grid; the loads of tile row I+1 are in flight while row I is consumed.
// (Fusing this with the accumulator initialisation was measured 5x slower: the unrolled form spills.)
template <int T> DEVINL void hx_tiles(const Ctx& k, const double* X, double* HX) {
  double hx[T], hn[T][4], hc[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t) hx[t] = 0.0;
#pragma unroll
  for (int Jt = 0; Jt < T; ++Jt)
#pragma unroll
    for (int p = 0; p < 4; ++p) hn[Jt][p] = k.Hw[((size_t)Jt * 4 + p) * 64 + k.lane];
#pragma unroll 1
  for (int I = 0; I < T; ++I) {
#pragma unroll
    for (int Jt = 0; Jt < T; ++Jt)
#pragma unroll
      for (int p = 0; p < 4; ++p) hc[Jt][p] = hn[Jt][p];
    if (I + 1 < T) {
#pragma unroll
      for (int Jt = 0; Jt < T; ++Jt)
#pragma unroll
        for (int p = 0; p < 4; ++p) hn[Jt][p] = k.Hw[((size_t)((I + 1) * T + Jt) * 4 + p) * 64 + k.lane];
    }
    double xk[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) xk[p] = X[16 * I + k.q + 4 * p];
#pragma unroll
    for (int Jt = 0; Jt < T; ++Jt)
#pragma unroll
      for (int p = 0; p < 4; ++p) hx[Jt] = fma(hc[Jt][p], xk[p], hx[Jt]);
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    double v = q_sum(hx[t]);
    if (k.q == 0) HX[16 * t + k.c] = v;
  }
}
// accumulator initialisation acc = H~ (upper tiles, C/D layout)
template <int T> DEVINL void acc_init(const Ctx& k, v4d* acc) {
#pragma unroll
  for (int I = 0; I < T; ++I)
#pragma unroll
    for (int Jt = I; Jt < T; ++Jt) {
      const double* hp = k.Hw + ((size_t)(I * T + Jt) * 4) * 64 + k.lane;
      v4d h;
#pragma unroll
      for (int p = 0; p < 4; ++p) h[p] = hp[p * 64];
      acc[Tri<T>::idx(I, Jt)] = h;
    }
}

// H~x of the core from the freshly initialised accumulators (the upper tiles of H~ that pass 1 is about to add A'DA to), so the
// 51 KB of H~ are read once per iteration instead of twice.  Tile (I,J), I <= J, in accumulator layout gives (H_IJ' x_I)[c] for
// block J with four FMAs and, for I < J, (H_IJ x_J)[q+4p] for block I with four more; the lane-group / DPP-row sums are taken
// once per block, not per tile.  XV, HX: LDS vectors.
template <int T> DEVINL void hx_from_acc(const Ctx& k, const v4d* acc, const double* XV, double* HX) {
  double xr[T][4], xc[T], sc[T], sr[T][4];
#pragma unroll
  for (int I = 0; I < T; ++I) {
    xc[I] = XV[16 * I + k.c]; sc[I] = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) { xr[I][p] = XV[16 * I + k.q + 4 * p]; sr[I][p] = 0.0; }
  }
#pragma unroll
  for (int I = 0; I < T; ++I)
#pragma unroll
    for (int J = I; J < T; ++J) {
      const v4d& h = acc[Tri<T>::idx(I, J)];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        sc[J] = fma(h[p], xr[I][p], sc[J]);                 // column form: this lane group's rows of (H_IJ' x_I)[c]
        if (J > I) sr[I][p] = fma(h[p], xc[J], sr[I][p]);   // row form: this lane's column of (H_IJ x_J)[q+4p]
      }
    }
  WAVE_SYNC();
#pragma unroll
  for (int J = 0; J < T; ++J) { const double v = q_sum(sc[J]); if (k.q == 0) HX[16 * J + k.c] = v; }
  WAVE_SYNC();
#pragma unroll
  for (int I = 0; I < T - 1; ++I)
#pragma unroll
    for (int p = 0; p < 4; ++p) { const double v = grp16_sum(sr[I][p]); if (k.c == 0) HX[16 * I + k.q + 4 * p] += v; }
}

template <int C> struct IC { static constexpr int value = C; };

// ---------------------------------------------------------------------------------------------
// Operand stream of the three passes over A~.  The stream is a plain sequence of 1 KB records (one column tile of
// one pair of k-steps) in exactly the order the passes consume it, so the producer is a linear walk: records go
// global memory -> LDS by `global_load_lds_dwordx4` (no VGPRs involved, LDS address = M0 + lane*16) into a ring of two
// halves of T records: the pair of k-steps being consumed and the pair in flight.  The consumer reads a record with
// one ds_read_b128 per lane.  The compiler does not track LDS-DMA -> ds_read dependences, so the ordering is explicit,
// and it is a FULL drain: `s_waitcnt vmcnt(0)` at the top of a pair (every vector-memory operation of this wave has
// completed, hence this pair's records are in LDS), then the DMA of the next pair is issued and flies during the
// matrix-core work of this one.
//   Round 1 kept D records in flight and waited with a counted `s_waitcnt vmcnt(D)`; that relies on every vector-memory
//   operation of the wave (LDS-DMA loads, row-array stores, register spills) retiring in issue order.  The full drain needs
//   no such assumption and costs nothing measurable (the next pair's DMA still overlaps this pair's matrix-core work).
//   (The wrong iterates of some -O2/-O3 builds were NOT a ring hazard: DESIGN.md, "Build-variant fragility: root cause".)
// The asm memory clobbers keep the compiler from moving LDS reads or DMA issues across the wait.
// ---------------------------------------------------------------------------------------------
#ifdef QP_DRAIN_STREAM   // the full-drain double buffer of the fragility investigation (kept for A/B runs): lead = one pair
template <int T> struct StreamCfg {
  static constexpr int R = 3 * T;   // (ring sized as below so that both variants share the LDS layout)
};
template <int T> struct Stream {
  static constexpr int R = StreamCfg<T>::R;
  const char* gnext;   // wave-uniform: global address of the next record to issue
  int half;            // ring half that holds the pair to consume next
  DEVINL void issue(const Ctx& k, int slot) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gnext + k.lane * 16),
                                     (__attribute__((address_space(3))) void*)(k.ring + slot * 128), 16, 0, 0);
    gnext += 1024;
  }
  DEVINL void start(const Ctx& k, int c0) {   // c0: tile count of the first pair
    gnext = reinterpret_cast<const char*>(k.Aw); half = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) if (t < c0) issue(k, t);
  }
  template <int C> DEVINL void next_pair(const Ctx& k, v2d* b, int cn) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int other = (half ^ 1) * T;
#pragma unroll
    for (int t = 0; t < T; ++t) if (t < cn) issue(k, other + t);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < C; ++t) b[t] = *reinterpret_cast<const v2d*>(k.ring + (half * T + t) * 128 + k.lane * 2);
    half ^= 1;
  }
  DEVINL void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};
#else
// Lead of D = 2T records (two pairs of k-steps at the full tile count, more in the sparse early trips) behind a COUNTED wait:
// a pair's C records are consumed after C new ones have been issued and `s_waitcnt vmcnt(D)` says that at most D vector-memory
// operations are still outstanding.  Loads (LDS-DMA, global, scratch) return in issue order, so the D newest outstanding loads
// are never this pair's; stores may retire in any order, which can only make the wait longer, never shorter.  The producer walks
// the linear record stream and runs up to D records past its end (still inside this QP's workspace).  One pair of lead (the
// full-drain variant above) left passes 2 and 3 waiting on the DMA: their VALU work per pair is shorter than the latency.
template <int T> struct StreamCfg {
  static constexpr int D = 2 * T;
  static constexpr int R = D + T;
};
template <int T> struct Stream {
  static constexpr int D = StreamCfg<T>::D, R = StreamCfg<T>::R;
  const char* gnext; int slot_i, slot_e;
  DEVINL void issue(const Ctx& k) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gnext + k.lane * 16),
                                     (__attribute__((address_space(3))) void*)(k.ring + slot_i * 128), 16, 0, 0);
    gnext += 1024;
    slot_i = slot_i + 1 == R ? 0 : slot_i + 1;
  }
  DEVINL void start(const Ctx& k, int) {
    gnext = reinterpret_cast<const char*>(k.Aw); slot_i = 0; slot_e = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) issue(k);
  }
  template <int C> DEVINL void next_pair(const Ctx& k, v2d* b, int) {
#pragma unroll
    for (int t = 0; t < C; ++t) issue(k);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D) : "memory");
#pragma unroll
    for (int t = 0; t < C; ++t) {
      int sl = slot_e + t; if (sl >= R) sl -= R;
      b[t] = *reinterpret_cast<const v2d*>(k.ring + sl * 128 + k.lane * 2);
    }
    slot_e += C; if (slot_e >= R) slot_e -= R;
  }
  DEVINL void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};
#endif

// Per-row coefficients of a pass (owner layout [slot][64] in global memory): one slot (16 k-steps) at a time is staged
// in LDS -- NA wave-wide loads per 16 k-steps instead of NA broadcast loads per k-step -- and read back as 16-byte
// pairs (both k-steps of a pair) with the lane group's row as address.
template <int NA> struct CoefStage {
  double reg[NA > 0 ? NA : 1];
  DEVINL void load(const Ctx& k, const double* const* arr, int js) {
    const int jc = js < k.J ? js : k.J - 1;
#pragma unroll
    for (int a = 0; a < NA; ++a) reg[a] = arr[a][jc * 64 + k.lane];
  }
  DEVINL void commit(const Ctx& k) {   // registers -> LDS (the previous slot is dead by now)
#pragma unroll
    for (int a = 0; a < NA; ++a) k.cof[a * 64 + k.lane] = reg[a];
  }
  DEVINL void read_pair(const Ctx& k, int s, v2d* out) const {   // s even: k-steps s, s+1 of lane group q
#pragma unroll
    for (int a = 0; a < NA; ++a) out[a] = *reinterpret_cast<const v2d*>(k.cof + a * 64 + k.q * 16 + (s & 15));
  }
};

// pass 1: acc += A~' D A~ on the matrix cores; p1 = A~'w1, p2 = A~'w2, p3 = A~'w3 on the VALU beside them.
// The stream is walked trip by trip (4 k-steps); a trip with tc column tiles only touches the tc(tc+1)/2 accumulator
// tiles it can reach.  Trips are sorted by tc, so the pass is T phases with compile-time tile counts (phase C: all
// trips with tc == C) and no branches inside a trip.
template <int T, int NB> DEVINL void pass_syrk(const Ctx& k, v4d* acc, double* P1, double* P2, double* P3, double* MB) {
  constexpr int NBB = NB > 0 ? NB : 1;
  constexpr int NA = 4 + NB;
  const int JS = k.J * 64;
  const double* arr[NA];
  arr[0] = rowp(k, R_D); arr[1] = rowp(k, R_W1); arr[2] = rowp(k, R_W2); arr[3] = rowp(k, R_W3);
#pragma unroll
  for (int e = 0; e < NB; ++e) arr[4 + e] = k.Ab + (size_t)e * JS;
  double p1[T], p2[T], p3[T];
  double pb[NBB][T], sbb[NBB][NBB], pwb[3][NBB];   // border: column of A'DA, border block, border entries of p1..p3
#pragma unroll
  for (int t = 0; t < T; ++t) { p1[t] = 0; p2[t] = 0; p3[t] = 0; }
#pragma unroll
  for (int e = 0; e < NBB; ++e) {
#pragma unroll
    for (int t = 0; t < T; ++t) pb[e][t] = 0;
#pragma unroll
    for (int f = 0; f < NBB; ++f) sbb[e][f] = 0;
    pwb[0][e] = pwb[1][e] = pwb[2][e] = 0;
  }
  Stream<T> st;
  CoefStage<NA> cs;
  int tr = 0;
  if (k.ntr > 0) { st.start(k, k.tcs[0]); cs.load(k, arr, 0); }
  auto phase = [&](auto Cc) __attribute__((always_inline)) {
    constexpr int C = decltype(Cc)::value;
    const int tr_end = k.tend[C];
    for (; tr < tr_end; ++tr) {
      if ((tr & 3) == 0) { cs.commit(k); cs.load(k, arr, (tr >> 2) + 1); }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        v2d b[C], cf[NA];
        st.template next_pair<C>(k, b, u == 0 ? C : (tr + 1 < tr_end ? C : (tr + 1 < k.ntr ? k.tcs[tr + 1] : 0)));
        cs.read_pair(k, 4 * tr + 2 * u, cf);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double a[C], ab[NBB];
          const double dd = cf[0][h], w1 = cf[1][h], w2 = cf[2][h], w3 = cf[3][h];
#pragma unroll
          for (int e = 0; e < NB; ++e) ab[e] = cf[4 + e][h];
#pragma unroll
          for (int t = 0; t < C; ++t) a[t] = dd * b[t][h];
#pragma unroll
          for (int I = 0; I < C; ++I)
#pragma unroll
            for (int Jt = I; Jt < C; ++Jt)
              acc[Tri<T>::idx(I, Jt)] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], b[Jt][h], acc[Tri<T>::idx(I, Jt)], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < C; ++t) {
            p1[t] = fma(w1, b[t][h], p1[t]); p2[t] = fma(w2, b[t][h], p2[t]); p3[t] = fma(w3, b[t][h], p3[t]);
          }
#pragma unroll
          for (int e = 0; e < NB; ++e) {
            const double dab = dd * ab[e];
#pragma unroll
            for (int t = 0; t < C; ++t) pb[e][t] = fma(dab, b[t][h], pb[e][t]);
#pragma unroll
            for (int f = e; f < NB; ++f) sbb[e][f] = fma(dab, ab[f], sbb[e][f]);
            pwb[0][e] = fma(w1, ab[e], pwb[0][e]); pwb[1][e] = fma(w2, ab[e], pwb[1][e]); pwb[2][e] = fma(w3, ab[e], pwb[2][e]);
          }
        }
      }
    }
  };
  if constexpr (T >= 1) phase(IC<1>{});
  if constexpr (T >= 2) phase(IC<2>{});
  if constexpr (T >= 3) phase(IC<3>{});
  if constexpr (T >= 4) phase(IC<4>{});
  if constexpr (T >= 5) phase(IC<5>{});
  if constexpr (T >= 6) phase(IC<6>{});
  if constexpr (T >= 7) phase(IC<7>{});
  if constexpr (T >= 8) phase(IC<8>{});
  if (k.ntr > 0) st.drain();
#pragma unroll
  for (int t = 0; t < T; ++t) {
    double v1 = q_sum(p1[t]), v2 = q_sum(p2[t]), v3 = q_sum(p3[t]);
    if (k.q == 0) { P1[16 * t + k.c] = v1; P2[16 * t + k.c] = v2; P3[16 * t + k.c] = v3; }
  }
#pragma unroll
  for (int e = 0; e < NB; ++e) {
#pragma unroll
    for (int t = 0; t < T; ++t) { const double vb = q_sum(pb[e][t]); if (k.q == 0) MB[e * k.np + 16 * t + k.c] = vb; }
    // border scalars are identical on the 16 lanes of a group: sum the four groups, lane 0 writes
    const double v1 = q_sum(pwb[0][e]), v2 = q_sum(pwb[1][e]), v3 = q_sum(pwb[2][e]);
    if (k.lane == 0) { P1[k.nc + e] = v1; P2[k.nc + e] = v2; P3[k.nc + e] = v3; }
#pragma unroll
    for (int f = 0; f < NB; ++f) {
      const double sv = q_sum(f >= e ? sbb[e][f] : sbb[f][e]);
      if (k.lane == 0) MB[e * k.np + k.nc + f] = sv;
    }
  }
}

// y = A~ v for NVEC vectors (LDS n-vectors) -> owner-layout row arrays.  FUSE: the first vector is the affine
// direction; as soon as a row's va = a_r' dxa is reduced, the second-order weight
//   w_r = (va+a1)(b1 + c1 (va+a1)) - (a2-va)(b2 + c2 (a2-va))      (a,b,c: per-row coefficients of row phase 1)
// is formed and p_cor += w_r a_r is accumulated in the same pass (saves one full stream over A per iteration).
// FUSE 3 is the polish step (see the kernel).  Same operand stream and phase structure as pass 1.
template <int T, int NB, int NVEC, int FUSE> DEVINL void pass_Av(const Ctx& k, const double* const* vin, double* const* rout, double* Pcor, double* Pcor2 = nullptr, const double* const* cfarr = nullptr) {
  constexpr int NBB = NB > 0 ? NB : 1;
  constexpr int NC = FUSE == 1 ? 6 : (FUSE >= 2 ? 3 : 0);   // per-row coefficient arrays of the fused part
  constexpr int NA = NC + NB;
  const int JS = k.J * 64;
  const double* arr[NA > 0 ? NA : 1];
  if (FUSE == 1) { arr[0] = rowp(k, R_RPL); arr[1] = rowp(k, R_CB1); arr[2] = rowp(k, R_CC1); arr[3] = rowp(k, R_RPU); arr[4] = rowp(k, R_CB2); arr[5] = rowp(k, R_CC2); }
  if (FUSE >= 2) { arr[0] = cfarr ? cfarr[0] : rowp(k, R_CB1); arr[1] = cfarr ? cfarr[1] : rowp(k, R_RPL); arr[2] = cfarr ? cfarr[2] : rowp(k, R_CC1); }   // refinement: rho*act, target b, multiplier y
#pragma unroll
  for (int f = 0; f < NB; ++f) arr[NC + f] = k.Ab + (size_t)f * JS;
  double v[NVEC][T], vb[NVEC][NBB], pc[T], pcb[NBB], pd[FUSE >= 2 ? T : 1], pdb[NBB];
#pragma unroll
  for (int e = 0; e < NVEC; ++e) {
#pragma unroll
    for (int t = 0; t < T; ++t) v[e][t] = vin[e][16 * t + k.c];
#pragma unroll
    for (int f = 0; f < NB; ++f) vb[e][f] = vin[e][k.nc + f];
  }
#pragma unroll
  for (int t = 0; t < T; ++t) pc[t] = 0.0;
#pragma unroll
  for (int f = 0; f < NBB; ++f) { pcb[f] = 0.0; pdb[f] = 0.0; }
#pragma unroll
  for (int t = 0; t < (FUSE >= 2 ? T : 1); ++t) pd[t] = 0.0;
  double keep[NVEC + 1];   // last entry: the updated multiplier of the polish modes (written to rout[NVEC])
#pragma unroll
  for (int e = 0; e < NVEC + 1; ++e) keep[e] = 0.0;
  Stream<T> st;
  CoefStage<NA> cs;
  int tr = 0;
  if (k.ntr > 0) { st.start(k, k.tcs[0]); if (NA > 0) cs.load(k, arr, 0); }
  auto phase = [&](auto Cc) __attribute__((always_inline)) {
    constexpr int C = decltype(Cc)::value;
    const int tr_end = k.tend[C];
    for (; tr < tr_end; ++tr) {
      if (NA > 0 && (tr & 3) == 0) { cs.commit(k); cs.load(k, arr, (tr >> 2) + 1); }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        v2d b[C], cf[NA > 0 ? NA : 1];
        st.template next_pair<C>(k, b, u == 0 ? C : (tr + 1 < tr_end ? C : (tr + 1 < k.ntr ? k.tcs[tr + 1] : 0)));
        if (NA > 0) cs.read_pair(k, 4 * tr + 2 * u, cf);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int s = 4 * tr + 2 * u + h;
          const int cc = s & 15;
#pragma unroll
          for (int e = 0; e < NVEC; ++e) {
            double dsum = 0.0;
#pragma unroll
            for (int t = 0; t < C; ++t) dsum = fma(b[t][h], v[e][t], dsum);
            dsum = grp16_sum(dsum);
#pragma unroll
            for (int f = 0; f < NB; ++f) dsum = fma(cf[NC + f][h], vb[e][f], dsum);
            if (k.c == cc) keep[e] = dsum;
            if (FUSE >= 2 && e == 0) {   // polish: pen = rho*act*(v - b), y^ = y - pen; accumulate A~'y^ and A~'pen
              const double pen = cf[0][h] * (dsum - cf[1][h]);
              const double ynew = cf[2][h] - pen;
              if (k.c == cc) keep[NVEC] = ynew;
#pragma unroll
              for (int t = 0; t < C; ++t) { pc[t] = fma(ynew, b[t][h], pc[t]); pd[t] = fma(pen, b[t][h], pd[t]); }
#pragma unroll
              for (int f = 0; f < NB; ++f) { pcb[f] = fma(ynew, cf[NC + f][h], pcb[f]); pdb[f] = fma(pen, cf[NC + f][h], pdb[f]); }
            }
            if (FUSE == 1 && e == 0) {
              const double dl_ = dsum + cf[0][h], du_ = cf[3][h] - dsum;
              const double w = dl_ * fma(cf[2][h], dl_, cf[1][h]) - du_ * fma(cf[5][h], du_, cf[4][h]);
#pragma unroll
              for (int t = 0; t < C; ++t) pc[t] = fma(w, b[t][h], pc[t]);
#pragma unroll
              for (int f = 0; f < NB; ++f) pcb[f] = fma(w, cf[NC + f][h], pcb[f]);
            }
          }
          if (cc == 15 || s + 1 == 4 * k.ntr) {
            const int js = s >> 4;
#pragma unroll
            for (int e = 0; e < NVEC; ++e) { rout[e][js * 64 + k.lane] = keep[e]; keep[e] = 0.0; }
            if (FUSE >= 2) { rout[NVEC][js * 64 + k.lane] = keep[NVEC]; keep[NVEC] = 0.0; }
          }
        }
      }
    }
  };
  if constexpr (T >= 1) phase(IC<1>{});
  if constexpr (T >= 2) phase(IC<2>{});
  if constexpr (T >= 3) phase(IC<3>{});
  if constexpr (T >= 4) phase(IC<4>{});
  if constexpr (T >= 5) phase(IC<5>{});
  if constexpr (T >= 6) phase(IC<6>{});
  if constexpr (T >= 7) phase(IC<7>{});
  if constexpr (T >= 8) phase(IC<8>{});
  if (k.ntr > 0) st.drain();
  if (FUSE) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      double pv = q_sum(pc[t]);
      if (k.q == 0) Pcor[16 * t + k.c] = pv;
    }
#pragma unroll
    for (int f = 0; f < NB; ++f) { const double pv = q_sum(pcb[f]); if (k.lane == 0) Pcor[k.nc + f] = pv; }
  }
  if (FUSE >= 2) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      double pv = q_sum(pd[t]);
      if (k.q == 0) Pcor2[16 * t + k.c] = pv;
    }
#pragma unroll
    for (int f = 0; f < NB; ++f) { const double pv = q_sum(pdb[f]); if (k.lane == 0) Pcor2[k.nc + f] = pv; }
  }
}

// p = A~' w (w: owner-layout row array) -> LDS n-vector (only used by the initial point)
template <int T, int NB> DEVINL void pass_Atw(const Ctx& k, const double* W, double* Pout) {
  constexpr int NBB = NB > 0 ? NB : 1;
  const int JS = k.J * 64;
  double p[T], pbv[NBB];
#pragma unroll
  for (int t = 0; t < T; ++t) p[t] = 0.0;
#pragma unroll
  for (int f = 0; f < NBB; ++f) pbv[f] = 0.0;
  for (int s = 0; s < 4 * k.ntr; ++s) {
    const int ri = (s >> 4) * 64 + k.q * 16 + (s & 15);
    const double w = W[ri];
    const int tr = s >> 2, tc = k.tcs[tr];
    const double* src = k.Aw + ((size_t)k.aoff[tr] + ((s >> 1) & 1) * tc) * 128 + k.lane * 2 + (s & 1);
#pragma unroll
    for (int t = 0; t < T; ++t)
      if (t < tc) p[t] = fma(w, src[t * 128], p[t]);
#pragma unroll
    for (int f = 0; f < NB; ++f) pbv[f] = fma(w, k.Ab[(size_t)f * JS + ri], pbv[f]);
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    double v = q_sum(p[t]);
    if (k.q == 0) Pout[16 * t + k.c] = v;
  }
#pragma unroll
  for (int f = 0; f < NB; ++f) { const double pv = q_sum(pbv[f]); if (k.lane == 0) Pout[k.nc + f] = pv; }
}

// ---------------------------------------------------------------------------------------------
// Register-resident blocked Cholesky M = U'U and triangular solves on the matrix cores.
//
// A 16x16 tile X held in accumulator (C/D) layout -- lane (c,q), reg p <-> X[q+4p][c] -- can be fed straight
// back as an MFMA operand: as the A operand it acts as X' (A[i][k] = X[k][i]), as the B operand as X.  Four
// MFMAs (p = 0..3) therefore compute X'Y for any two resident tiles with no data movement, which is all an
// upper-form blocked Cholesky needs:  U_KJ = U_KK^-T M_KJ (done as row operations on the whole block row),
// M_IJ -= U_KI' U_KJ.  Right-hand sides ride along as one more tile column (16 slots), so the forward solve
// U'y = b is a by-product of the factorisation.  The backward solve U x = y needs U_KJ' tiles; those are
// transposed one at a time through a 2 KiB LDS scratch.
// ---------------------------------------------------------------------------------------------
template <int T> DEVINL void mfma4_sub(const v4d& X, const v4d& Y, v4d& Dst) {  // Dst -= X' Y
#pragma unroll
  for (int p = 0; p < 4; ++p) Dst = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[p], Y[p], Dst, 0, 0, 0);
}
DEVINL v4d mfma4_new(const v4d& X, const v4d& Y) {  // X' Y
  v4d Z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int p = 0; p < 4; ++p) Z = __builtin_amdgcn_mfma_f64_16x16x4f64(X[p], Y[p], Z, 0, 0, 0);
  return Z;
}

// tile transpose through the LDS scratch: returns Z with Z[row][col] = X[col][row] (C/D layout both sides)
DEVINL v4d tile_transpose(const Ctx& k, double* scratch, const v4d& Xt) {
  WAVE_SYNC();
#pragma unroll
  for (int p = 0; p < 4; ++p) scratch[(k.q + 4 * p) * 17 + k.c] = Xt[p];
  WAVE_SYNC();
  v4d Z;
#pragma unroll
  for (int p = 0; p < 4; ++p) Z[p] = scratch[k.c * 17 + k.q + 4 * p];
  return Z;
}

// resident LDS tiles (row-major, 17-double rows): store once, read back either as stored or transposed
DEVINL void tile_store(const Ctx& k, double* slot, const v4d& Xt) {
#pragma unroll
  for (int p = 0; p < 4; ++p) slot[(k.q + 4 * p) * 17 + k.c] = Xt[p];
}
DEVINL v4d tile_load(const Ctx& k, const double* slot) {
  v4d Z;
#pragma unroll
  for (int p = 0; p < 4; ++p) Z[p] = slot[(k.q + 4 * p) * 17 + k.c];
  return Z;
}
DEVINL v4d tile_load_t(const Ctx& k, const double* slot) {
  v4d Z;
#pragma unroll
  for (int p = 0; p < 4; ++p) Z[p] = slot[k.c * 17 + k.q + 4 * p];
  return Z;
}

// Factorise one 16x16 diagonal tile D = U'U in place, four 4-row panels, and apply the same row operations to two
// companion tiles: Yk (enters as the identity, leaves as U^-T) and the right-hand-side tile rk (leaves as U^-T rk).
// Per panel p: the 4x4 diagonal block (10 numbers, read with v_readlane) is factorised and inverted redundantly by
// every lane -- W = R^-T, wave-uniform -- and applied to the panel rows of the three tiles as one K=4 MFMA each
// (A operand = W scattered to the panel's rows); the rows of the later panels are then updated by one more K=4 MFMA
// per tile.  No cross-lane data movement besides the readlanes, no LDS.  (The former version walked the 16 rows one
// by one with three ds_bpermute round trips per row: 12k cycles per tile, 80 % of the whole factorisation.)
// Only these three tiles see VALU work; every other tile of the factorisation is touched by the matrix cores alone.
DEVINL int diag_factor(const Ctx& k, v4d& Ud, v4d& Yk, v4d& rk, double floor_abs) {
  int bad = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    // D[a][b] = M[4p+a][4p+b] lives in lane (c = 4p+b, q = a), register p
    const double d00 = rl(Ud[p], 4 * p + 0), d01 = rl(Ud[p], 4 * p + 1), d02 = rl(Ud[p], 4 * p + 2), d03 = rl(Ud[p], 4 * p + 3);
    const double d11 = rl(Ud[p], 16 + 4 * p + 1), d12 = rl(Ud[p], 16 + 4 * p + 2), d13 = rl(Ud[p], 16 + 4 * p + 3);
    const double d22 = rl(Ud[p], 32 + 4 * p + 2), d23 = rl(Ud[p], 32 + 4 * p + 3);
    const double d33 = rl(Ud[p], 48 + 4 * p + 3);
    auto piv = [&](double t) { if (!(t > floor_abs)) { if (!(fabs(t) < INFINITY)) bad = 1; t = floor_abs; } return rsqrt(t); };
    // R'R = D (R upper triangular), i_a = 1/R[a][a]
    const double i0 = piv(d00);
    const double r01 = d01 * i0, r02 = d02 * i0, r03 = d03 * i0;
    const double i1 = piv(fma(-r01, r01, d11));
    const double r12 = fma(-r01, r02, d12) * i1, r13 = fma(-r01, r03, d13) * i1;
    const double i2 = piv(fma(-r12, r12, fma(-r02, r02, d22)));
    const double r23 = fma(-r12, r13, fma(-r02, r03, d23)) * i2;
    const double i3 = piv(fma(-r23, r23, fma(-r13, r13, fma(-r03, r03, d33))));
    // W = (R')^-1, lower triangular
    const double w10 = -r01 * i0 * i1;
    const double w20 = -fma(r12, w10, r02 * i0) * i2, w21 = -r12 * i1 * i2;
    const double w30 = -fma(r23, w20, fma(r13, w10, r03 * i0)) * i3, w31 = -fma(r23, w21, r13 * i1) * i3, w32 = -r23 * i2 * i3;
    // A operand: lane (i = c, kq = q) holds W[c-4p][q] on the panel's rows, 0 elsewhere.  (The compiler turns these selects into
    // ~12 divergent regions per panel and sinks the products above into them.  A branch-free construction -- 0/1 masks times the
    // ten values -- was measured in round 2: 34 % fewer instructions in the factorisation, but 464 instead of 79 spilled
    // registers in the kernel and 17 % slower overall; a leaner rsqrt alone was 2 % slower for the same reason.)
    const int a_ = k.c - 4 * p;
    double wa = 0.0;
    if (k.q == 0) wa = a_ == 0 ? i0 : (a_ == 1 ? w10 : (a_ == 2 ? w20 : (a_ == 3 ? w30 : 0.0)));
    if (k.q == 1) wa = a_ == 1 ? i1 : (a_ == 2 ? w21 : (a_ == 3 ? w31 : 0.0));
    if (k.q == 2) wa = a_ == 2 ? i2 : (a_ == 3 ? w32 : 0.0);
    if (k.q == 3) wa = a_ == 3 ? i3 : 0.0;
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    const v4d nu = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, Ud[p], z, 0, 0, 0);
    const v4d ny = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, Yk[p], z, 0, 0, 0);
    const v4d nr = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, rk[p], z, 0, 0, 0);
    Ud[p] = nu[p]; Yk[p] = ny[p]; rk[p] = nr[p];
    if (p < 3) {
      const double a = (k.c > 4 * p + 3) ? -Ud[p] : 0.0;
      Yk = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Yk[p], Yk, 0, 0, 0);
      rk = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rk[p], rk, 0, 0, 0);
      Ud = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Ud[p], Ud, 0, 0, 0);
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) if (k.c < k.q + 4 * p) Ud[p] = 0.0;   // strictly lower part only ever held the symmetric copy
  return bad;
}

// Blocked Cholesky of acc (upper tiles) in place.  Yt[K] = U_KK^-T and Wt[K] = U_KK^-1 are kept for the solves;
// rh rides along and leaves as y = U^-T b.
template <int T, int K, bool RHS> struct FactorStep {
  static DEVINL int run(const Ctx& k, v4d* acc, double* YL, v4d* rh, double floor_abs) {
    v4d Yk, none = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < 4; ++p) Yk[p] = (k.q + 4 * p == k.c) ? 1.0 : 0.0;
    int bad = diag_factor(k, acc[Tri<T>::idx(K, K)], Yk, RHS ? rh[K] : none, floor_abs);
    tile_store(k, YL + K * 272, Yk);          // U_KK^-T stays in LDS for the solves of this iteration
    WAVE_SYNC();
    const v4d Wk = tile_load_t(k, YL + K * 272);   // U_KK^-1
    // U_KJ = U_KK^-T M_KJ through the explicit inverse, then one step of refinement against U_KK itself:
    // U_KJ += U_KK^-T (M_KJ - U_KK' U_KJ).  The product with the explicit inverse alone leaves a backward error of
    // cond(U_KK) eps in the block row (measured on the normal matrix of kinematic N = 40 id 6585 at iteration 14,
    // cond(U_00) = 6e7: |M - U'U| / |M| = 1.8e-14 against 1e-15 for a substitution; the refined row reaches 7e-16), which
    // reappears as noise in the dual residual of the next iterate (10x the CPU oracle's) and jams the end game.
    const v4d& UKK = acc[Tri<T>::idx(K, K)];
#pragma unroll
    for (int Jt = K + 1; Jt < T; ++Jt) {
      v4d Rr = acc[Tri<T>::idx(K, Jt)];
      v4d Ukj = mfma4_new(Wk, Rr);
#ifndef QP_NO_ROW_REFINE
      mfma4_sub<T>(UKK, Ukj, Rr);                                                   // M_KJ - U_KK' U_KJ
#pragma unroll
      for (int p = 0; p < 4; ++p) Ukj = __builtin_amdgcn_mfma_f64_16x16x4f64(Wk[p], Rr[p], Ukj, 0, 0, 0);
#endif
      acc[Tri<T>::idx(K, Jt)] = Ukj;
    }
#pragma unroll
    for (int I = K + 1; I < T; ++I) {
      const v4d& UKI = acc[Tri<T>::idx(K, I)];
#pragma unroll
      for (int Jt = I; Jt < T; ++Jt) mfma4_sub<T>(UKI, acc[Tri<T>::idx(K, Jt)], acc[Tri<T>::idx(I, Jt)]);
      if (RHS) mfma4_sub<T>(UKI, rh[K], rh[I]);
    }
    return bad | FactorStep<T, K + 1, RHS>::run(k, acc, YL, rh, floor_abs);
  }
};
template <int T, bool RHS> struct FactorStep<T, T, RHS> {
  static DEVINL int run(const Ctx&, v4d*, double*, v4d*, double) { return 0; }
};
// with right-hand sides riding along as a tile column (they leave as y = U^-T b) ...
template <int T> DEVINL int reg_factor(const Ctx& k, v4d* acc, double* YL, v4d* rh, double floor_abs) {
  return FactorStep<T, 0, true>::run(k, acc, YL, rh, floor_abs);
}
// ... or the factor alone (the solves then run on the VALU: vec_forward / vec_backward)
template <int T> DEVINL int reg_factor_only(const Ctx& k, v4d* acc, double* YL, double floor_abs) {
  return FactorStep<T, 0, false>::run(k, acc, YL, nullptr, floor_abs);
}

// forward solve U'y = b on a fresh right-hand-side tile column: y_K = U_KK^-T (b_K - sum_{I<K} U_IK' y_I)
template <int T> DEVINL void reg_forward(const Ctx& k, const v4d* acc, const double* YL, v4d* rh) {
#pragma unroll
  for (int K = 0; K < T; ++K) {
    rh[K] = mfma4_new(tile_load_t(k, YL + K * 272), rh[K]);
#pragma unroll
    for (int I = K + 1; I < T; ++I) mfma4_sub<T>(acc[Tri<T>::idx(K, I)], rh[K], rh[I]);
  }
}

// backward solve U x = y in place: x_K = U_KK^-1 (y_K - sum_{J>K} U_KJ x_J); U_KJ' comes through the LDS scratch
template <int T> DEVINL void reg_backward(const Ctx& k, const v4d* acc, const double* YL, v4d* rh, double* scratch) {
#pragma unroll
  for (int K = T - 1; K >= 0; --K) {
#pragma unroll
    for (int Jt = K + 1; Jt < T; ++Jt) {
      const v4d Lt = tile_transpose(k, scratch, acc[Tri<T>::idx(K, Jt)]);   // Lt[kappa][i] = U_KJ[i][kappa]
      mfma4_sub<T>(Lt, rh[Jt], rh[K]);
    }
    rh[K] = mfma4_new(tile_load(k, YL + K * 272), rh[K]);                    // (U_KK^-T)' = U_KK^-1
  }
}

// ---------------------------------------------------------------------------------------------
// Triangular solves of ONE right-hand side on the VALU.  A 16-wide right-hand-side tile column on the matrix cores spends 16x
// the work on a single vector and chains four dependent 64-cycle MFMAs per tile; the same products here are four FMAs per tile
// plus DPP / lane-swap reductions, and the backward sweep needs no tile transposes through LDS.
// Vector layouts: "by column": lane (c, .) holds v[c];  "by row": reg p of lane (., q) holds v[q + 4p]  (each replicated
// over the other lane coordinate).  Tiles are in accumulator layout: lane (c,q), reg p <-> X[q+4p][c].
// ---------------------------------------------------------------------------------------------
// REFINE: the diagonal blocks are applied through their explicit inverses U_KK^-T (LDS tiles); one step of refinement against
// U_KK itself makes that as accurate as a substitution.  The two step directions of an iteration need it (with right-hand sides
// riding along the factorisation, as in round 1, they went through substitution-like panel operations; without it 1 of 4096
// kinematic N = 20 instances diverged), the corrector solve never had it.
// U'y = b:  t_K = b_K - sum_{I<K} U_IK' y_I (by column),  y_K = U_KK^-T t_K (by row).  B: LDS vector (core part).
template <int T, bool REFINE> DEVINL void vec_forward(const Ctx& k, const v4d* acc, const double* YL, const double* B, double (&y)[T][4]) {
#pragma unroll
  for (int K = 0; K < T; ++K) {
    double s = 0.0;
#pragma unroll
    for (int I = 0; I < K; ++I)
#pragma unroll
      for (int p = 0; p < 4; ++p) s = fma(acc[Tri<T>::idx(I, K)][p], y[I][p], s);   // this lane group's rows of (U_IK' y_I)[c]
    const double t = B[16 * K + k.c] - (K > 0 ? q_sum(s) : 0.0);
    const v4d Yt = tile_load(k, YL + K * 272);                                        // U_KK^-T
#pragma unroll
    for (int p = 0; p < 4; ++p) y[K][p] = grp16_sum(Yt[p] * t);                       // y_K[q+4p] = sum_c Y[q+4p][c] t[c]
    if (REFINE) {
      const v4d& U = acc[Tri<T>::idx(K, K)];
#pragma unroll
      for (int rep = 0; rep < QP_SOLVE_REFINE_STEPS; ++rep) {
        double r = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) r = fma(U[p], y[K][p], r);
        r = t - q_sum(r);                                                             // t - U_KK' y  (by column)
#pragma unroll
        for (int p = 0; p < 4; ++p) y[K][p] += grp16_sum(Yt[p] * r);
      }
    }
  }
}
// U x = y:  w_K = y_K - sum_{J>K} U_KJ x_J (by row),  x_K = U_KK^-1 w_K = (U_KK^-T)' w_K (by column) -> X (LDS vector, core part)
template <int T, bool REFINE> DEVINL void vec_backward(const Ctx& k, const v4d* acc, const double* YL, const double (&y)[T][4], double* X) {
  double x[T];
#pragma unroll
  for (int K = T - 1; K >= 0; --K) {
    const v4d Yt = tile_load(k, YL + K * 272);
    double w[4], s2 = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      double s = 0.0;
#pragma unroll
      for (int J = K + 1; J < T; ++J) s = fma(acc[Tri<T>::idx(K, J)][p], x[J], s);    // this lane's column of (U_KJ x_J)[q+4p]
      w[p] = y[K][p] - (K < T - 1 ? grp16_sum(s) : 0.0);
      s2 = fma(Yt[p], w[p], s2);
    }
    x[K] = q_sum(s2);
    if (REFINE) {
      const v4d& U = acc[Tri<T>::idx(K, K)];
#pragma unroll
      for (int rep = 0; rep < QP_SOLVE_REFINE_STEPS; ++rep) {
        double d = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) d = fma(Yt[p], w[p] - grp16_sum(U[p] * x[K]), d);  // Y' (w - U_KK x)
        x[K] += q_sum(d);
      }
    }
    if (k.q == 0) X[16 * K + k.c] = x[K];
  }
}
// by-row vector <-> LDS vector
template <int T> DEVINL void vec_rows_store(const Ctx& k, const double (&y)[T][4], double* V) {
  if (k.c == 0) {
#pragma unroll
    for (int K = 0; K < T; ++K)
#pragma unroll
      for (int p = 0; p < 4; ++p) V[16 * K + k.q + 4 * p] = y[K][p];
  }
}
template <int T> DEVINL void vec_rows_load(const Ctx& k, const double* V, double (&y)[T][4]) {
#pragma unroll
  for (int K = 0; K < T; ++K)
#pragma unroll
    for (int p = 0; p < 4; ++p) y[K][p] = V[16 * K + k.q + 4 * p];
}

// right-hand sides: NS LDS vectors <-> slots 0..NS-1 (= lane column c) of the rhs tile column (core rows only)
template <int T, int NS> DEVINL void rhs_load(const Ctx& k, v4d* rh, const double* const* vecs) {
#pragma unroll
  for (int K = 0; K < T; ++K)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int i = 16 * K + k.q + 4 * p;
      double v = 0.0;
#pragma unroll
      for (int e = 0; e < NS; ++e) if (k.c == e) v = vecs[e][i];
      rh[K][p] = v;
    }
}
template <int T, int NS> DEVINL void rhs_store(const Ctx& k, const v4d* rh, double* const* vecs) {
#pragma unroll
  for (int K = 0; K < T; ++K)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int i = 16 * K + k.q + 4 * p;
#pragma unroll
      for (int e = 0; e < NS; ++e) if (k.c == e) vecs[e][i] = rh[K][p];
    }
}

#ifndef QP_STAMPS
#define QP_STAMPS 0
#endif
#if QP_STAMPS
#define STAMP_DECL unsigned long long st_acc[16]; for (int i_ = 0; i_ < 16; ++i_) st_acc[i_] = 0; unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#define STAMP(id) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[id] += t_ - st_t0; st_t0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_OUT do { if (P.dump && P.dump_stage == 9 && lane == 0) for (int i_ = 0; i_ < 16; ++i_) P.dump[(size_t)b * 16 + i_] = (double)st_acc[i_]; } while (0)
#else
#define STAMP_DECL
#define STAMP(id) do { } while (0)
#define STAMP_OUT do { } while (0)
#endif

#ifndef QP_REFINE_ATTEMPTS
#define QP_REFINE_ATTEMPTS 3
#endif
#ifndef QP_WAVES_PER_SIMD
#define QP_WAVES_PER_SIMD 1
#endif
template <int T, int NB> __global__ __launch_bounds__(64, QP_WAVES_PER_SIMD) void qp_solve_kernel(QpParams P) {
  const int b = P.order ? P.order[blockIdx.x] : blockIdx.x;   // launch order: hardest-looking instances first
  Ctx k;
  const QpDims& d = P.d;
  k.n = d.n; k.m = d.m; k.T = T; k.Kq = d.Kq; k.J = d.J; k.JB = d.JB; k.JT = d.J + d.JB; k.np = d.np; k.ld = 0;
  k.lane = threadIdx.x; k.c = k.lane & 15; k.q = k.lane >> 4; k.nc = d.nc; k.nb = d.nb;
  double* ws = P.ws + (size_t)b * d.ws_per_qp;
  k.Aw = ws + d.off_Aw; k.Hw = ws + d.off_Hw; k.Ab = ws + d.off_Ab; k.Hb = ws + d.off_Hb;
  k.rows = ws + d.off_rows; k.rowlen = d.rowlen; k.ntr = d.ntr;
  k.perm = reinterpret_cast<const int*>(ws + d.off_meta); k.tcs = k.perm + (size_t)(d.J > 0 ? d.J : 1) * 64;
  k.aoff = k.tcs + d.ntr; k.tend = k.aoff + d.ntr + 1;
  extern __shared__ double lds[];
  k.Ms = nullptr; k.vec = lds;
  double* MB = lds + (size_t)V_NARR * d.np;    // NB border-column vectors (A'DA border, then U^-T m_b)
  double* YL = MB + (size_t)(NB > 0 ? NB : 1) * d.np;   // T resident tiles U_KK^-T (MB: one vector per border column)
  k.ring = YL + T * 272;                        // operand ring of the streaming passes
  k.cof = k.ring + StreamCfg<T>::R * 128;       // coefficient staging, (6 + NB) arrays of 64
  double* SCR = k.ring;                         // 16 x 17 tile-transpose scratch of the MFMA solves (A/B builds): the ring is idle then
  const double* gw = ws + d.off_gw;
  const double* Es = ws + d.off_E;
  const double* Fs = ws + d.off_F;
  const int lane = k.lane, n = k.n, JT = k.JT, J = k.J, nc = k.nc, nb = k.nb;
  constexpr int NT = Tri<T>::NT;
  constexpr int NBB = NB > 0 ? NB : 1;

#define X vecp(k, V_X)
#define G vecp(k, V_G)
#define HX vecp(k, V_HX)
#define R1 vecp(k, V_R1)
#define R2 vecp(k, V_R2)
#define P1 vecp(k, V_P1)
#define P2 vecp(k, V_P2)
#define P3 vecp(k, V_P3)
#define DX vecp(k, V_DX)
#define EV vecp(k, V_E)
#define aL rowp(k, R_L)
#define aU rowp(k, R_U)
#define aTL rowp(k, R_TL)
#define aTU rowp(k, R_TU)
#define aZL rowp(k, R_ZL)
#define aZU rowp(k, R_ZU)
#define aV rowp(k, R_V)
#define aD rowp(k, R_D)
#define aW1 rowp(k, R_W1)
#define aW2 rowp(k, R_W2)
#define aW3 rowp(k, R_W3)
#define aVA rowp(k, R_VA)
#define aVC rowp(k, R_VC)
#define aRPL rowp(k, R_RPL)
#define aRPU rowp(k, R_RPU)

  // Hx = H~ x: core through the tile grid, border columns (full-length vectors Hb[b]) on the VALU
  auto hx_border = [&](const double* XV) __attribute__((always_inline)) {
    if (NB > 0) {
      WAVE_SYNC();
      double xb[NBB], sb[NBB];
#pragma unroll
      for (int e = 0; e < NB; ++e) { xb[e] = XV[nc + e]; sb[e] = 0.0; }
      for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        if (i < n) {
          double add = 0.0;
#pragma unroll
          for (int e = 0; e < NB; ++e) { const double hbi = k.Hb[(size_t)e * k.np + i]; add = fma(hbi, xb[e], add); sb[e] = fma(hbi, XV[i], sb[e]); }
          if (i < nc) HX[i] += add;
        }
      }
#pragma unroll
      for (int e = 0; e < NB; ++e) { const double tot = wave_sum(sb[e]); if (lane == 0) HX[nc + e] = tot; }
    }
  };
  auto hx_full = [&](const double* XV) __attribute__((always_inline)) { hx_tiles<T>(k, XV, HX); hx_border(XV); };

  // ---- load n-vectors, initial x = clamp(0, l, u) (scaled), count finite sides ----
  for (int i = lane; i < k.np; i += 64) { G[i] = gw[i]; EV[i] = Es[i]; R1[i] = 0; R2[i] = 0; DX[i] = 0; }
  int cnt_local = 0, infeas = 0;
  for (int js = 0; js < JT; ++js) {
    const int ix = js * 64 + lane;
    const bool valid = row_valid(k, js);
    double l = aL[ix], u = aU[ix];
    if (valid) {
      if (l > -INFINITY && u < INFINITY) {
        if (l > u) infeas = 1;
        if (!(u > l)) {  // equality row: open a tiny interior (documented relaxation)
          const double eps = 1e-9 * fmax(1.0, fabs(l));
          l -= eps; u += eps; aL[ix] = l; aU[ix] = u;
        }
      }
      cnt_local += (l > -INFINITY) + (u < INFINITY);
    }
    if (js >= J) {
      const int i = (js - J) * 64 + lane;
      if (i < k.np) {
        double xi = 0.0;
        if (P.x_init) { const int ui = i < k.n ? qp_user_index(d, i) : -1; if (ui >= 0) { const double xs = P.x_init[(size_t)b * d.nu + ui] / EV[i]; if (fabs(xs) < INFINITY) xi = xs; } }
        if (valid) { if (l > -INFINITY && xi < l) xi = l; if (u < INFINITY && xi > u) xi = u; }
        X[i] = xi;
      }
    }
  }
  const double cnt = fmax(1.0, wave_sum((double)cnt_local));
  infeas = wave_max((double)infeas) > 0;
  WAVE_SYNC();

  STAMP_DECL
  int flag = 1, it = 0, flag_polished = 0;
  double fval_s = 0.0, merit_s = INFINITY;   // objective / relative KKT residual of the point that is returned
  if (infeas) { flag = -2; }
  if (ws[d.off_bad] != 0.0) flag = -1;   // NaN / Inf in this QP's data (found by the prep kernel): -1 after 0 iterations, x = clamp(0, lb, ub)

  // ---- v = G x ----
  {
    const double* vin[1] = {X}; double* rout[1] = {aV};
    pass_Av<T, NB, 1, 0>(k, vin, rout, nullptr);
    for (int jb = 0; jb < k.JB; ++jb) { const int i = jb * 64 + lane; aV[(J + jb) * 64 + lane] = i < n ? X[i] : 0.0; }
  }
  // ---- initial slacks / multipliers in the equilibrated problem: t = max(resid, T0), z = Z0 (a scan over the
  //      synthetic LTV-MPC families: (10,100) needs 8-16 % fewer iterations than (1,1)) ----
  const double T0 = 10.0, Z0 = 100.0;
  for (int js = 0; js < JT; ++js) {
    const int ix = js * 64 + lane;
    const bool valid = row_valid(k, js);
    const double l = aL[ix], u = aU[ix], v = aV[ix];
    const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
    aTL[ix] = hl ? fmax(v - l, T0) : 1.0;
    aTU[ix] = hu ? fmax(u - v, T0) : 1.0;
    aZL[ix] = hl ? Z0 : 0.0;
    aZU[ix] = hu ? Z0 : 0.0;
    aW3[ix] = (js < J) ? ((hl ? Z0 : 0.0) - (hu ? Z0 : 0.0)) : 0.0;
  }
  WAVE_SYNC();
  // bound multipliers absorb the initial dual residual r = Hx + g - A'(zl - zu)
  {
    hx_full(X);
    pass_Atw<T, NB>(k, aW3, P3);
    WAVE_SYNC();
    for (int jb = 0; jb < k.JB; ++jb) {
      const int i = jb * 64 + lane, ix = (J + jb) * 64 + lane;
      if (i < n) {
        const double r = HX[i] + G[i] - P3[i];
        if (aL[ix] > -INFINITY) aZL[ix] = fmax(r, 0.0) + Z0;
        if (aU[ix] < INFINITY) aZU[ix] = fmax(-r, 0.0) + Z0;
      }
    }
    WAVE_SYNC();
  }

  // fall-back iterate (best one that met tol_loose)
  double saved_merit = INFINITY, best_res = INFINITY;
  int have_saved = 0, stall = 0;
  double* XS = ws + d.off_save;            // np
  double* LAMS = ws + d.off_save + k.np;   // rowlen

  // residuals and barrier weights of one row (owner lane): everything pass 1 / pass 2 need, in owner layout
  auto row1_body = [&](int ix, bool valid, double l, double u, double v, double tl, double tu, double zl, double zu,
                       double& s_gap, double& m_rp) {
    const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
    const double rpl = hl ? v - l - tl : 0.0, rpu = hu ? u - v - tu : 0.0;
    const double dl_ = hl ? zl / tl : 0.0, du_ = hu ? zu / tu : 0.0;
    aRPL[ix] = rpl; aRPU[ix] = rpu;
    rowp(k, R_CB1)[ix] = dl_; rowp(k, R_CC1)[ix] = hl ? dl_ / tl : 0.0;
    rowp(k, R_CB2)[ix] = du_; rowp(k, R_CC2)[ix] = hu ? du_ / tu : 0.0;
    aD[ix] = dl_ + du_;
    aW1[ix] = -dl_ * rpl + du_ * rpu;                              // affine rhs weight
    aW2[ix] = (hl ? 1.0 / tl : 0.0) - (hu ? 1.0 / tu : 0.0);      // centering weight (times sigma*mu)
    aW3[ix] = (hl ? zl : 0.0) - (hu ? zu : 0.0);                  // current multiplier (for the dual residual)
    s_gap += (hl ? tl * zl : 0.0) + (hu ? tu * zu : 0.0);
    const double sc = fmax(1.0, fabs(v));
    if (hl) m_rp = fmax(m_rp, fabs(rpl) / fmax(sc, fabs(l)));
    if (hu) m_rp = fmax(m_rp, fabs(rpu) / fmax(sc, fabs(u)));
  };
  struct Slot { int ix; bool valid; double l, u, v, tl, tu, zl, zu, va, vc, w2, rpl, rpu; };
  auto load_slot = [&](int js) {   // all loads of one owner-layout slot, issued together (one latency per trip)
    Slot s_;
    const int jc = js < JT ? js : JT - 1;
    s_.ix = jc * 64 + lane; s_.valid = js < JT && row_valid(k, jc);
    const int ix = s_.ix;
    s_.l = aL[ix]; s_.u = aU[ix]; s_.v = aV[ix]; s_.tl = aTL[ix]; s_.tu = aTU[ix]; s_.zl = aZL[ix]; s_.zu = aZU[ix];
    s_.va = aVA[ix]; s_.vc = aVC[ix]; s_.w2 = aW2[ix]; s_.rpl = aRPL[ix]; s_.rpu = aRPU[ix];
    return s_;
  };
  double gap = 0.0, rp_rel = 0.0;   // carried across iterations (produced by the update sweep)

  v4d acc[NT];          // upper tiles of M, then of its Cholesky factor U
  v4d rh[T];            // right-hand-side tile column
  double Ubb[NBB][NBB]; // Cholesky factor of the border Schur complement (wave-uniform scalars)
  // border part of a solve: R holds y_c = U^-T b_c (core) and b_b (border); leaves the border solution in R[nc+e]
  // and y_c - sum_e u_e x_e in the core, ready for the backward sweep
  auto border_solve = [&](double* R) __attribute__((always_inline)) {
    if (NB > 0) {
      double yb[NBB], xb[NBB];
#pragma unroll
      for (int e = 0; e < NB; ++e) {
        double dsum = 0.0;
        for (int h = 0; h < 2; ++h) { const int i = lane + 64 * h; if (i < nc) dsum = fma(MB[e * k.np + i], R[i], dsum); }
        double tt = R[nc + e] - wave_sum(dsum);
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
      WAVE_SYNC();
      for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        if (i < nc) {
          double r = R[i];
#pragma unroll
          for (int e = 0; e < NB; ++e) r = fma(-MB[e * k.np + i], xb[e], r);
          R[i] = r;
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int e = 0; e < NB; ++e) R[nc + e] = xb[e];
      }
      WAVE_SYNC();
    }
  };
  // M = (acc from pass 1) + diag(aD on the variable rows), border columns = H~ border + A'DA border (MB); factorise in
  // registers and solve for the two right-hand sides in R1, R2 (in place).  Returns 1 on a non-finite pivot.
  auto factor_solve2 = [&](int it_now) __attribute__((always_inline)) -> int {
    double dmax_l = 0;
#pragma unroll
    for (int K = 0; K < T; ++K) {
      const int i = 16 * K + k.c;                       // diagonal element of tile (K,K) lives on lane c with q = c&3, reg c>>2
      const int ix = (J + (i >> 6)) * 64 + (i & 63);
      const double dadd = i < n ? aD[ix] : 1.0;         // padded indices get a unit diagonal
      const bool mine = (k.q == (k.c & 3));
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (mine && p == (k.c >> 2)) { acc[Tri<T>::idx(K, K)][p] += dadd; dmax_l = fmax(dmax_l, acc[Tri<T>::idx(K, K)][p]); }
    }
    if (NB > 0) {
      for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        if (i < k.np) {
          const int ix = (J + (i >> 6)) * 64 + (i & 63);
#pragma unroll
          for (int e = 0; e < NB; ++e) {   // border column e of M: H~ column + A'DA column (+ its variable-bound weight on the diagonal)
            double v = i < n ? k.Hb[(size_t)e * k.np + i] + MB[e * k.np + i] : 0.0;
            if (i == nc + e) { v += e < nb ? aD[ix] : 1.0; dmax_l = fmax(dmax_l, v); }
            MB[e * k.np + i] = v;
          }
        }
      }
    }
    const double dmax = wave_max(dmax_l);
    WAVE_SYNC();
#ifdef QP_DEBUG_DUMP   // diagnostic build only (libfsaempc_dbg.so): the shipped kernel carries no dump branches
    if (P.dump && b == 0 && P.dump_stage == 1 && it_now == P.dump_iter) {  // debug: M, p1, p2, p3, Hx
#pragma unroll
      for (int I = 0; I < T; ++I)
#pragma unroll
        for (int Jt = I; Jt < T; ++Jt)
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int r = 16 * I + k.q + 4 * p, cc = 16 * Jt + k.c;
            if (r < n && cc < n) { P.dump[r * n + cc] = acc[Tri<T>::idx(I, Jt)][p]; if (I != Jt || cc >= r) P.dump[cc * n + r] = acc[Tri<T>::idx(I, Jt)][p]; }
          }
      for (int e = 0; e < nb; ++e)
        for (int i = lane; i < n; i += 64) { P.dump[i * n + nc + e] = MB[e * k.np + i]; P.dump[(nc + e) * n + i] = MB[e * k.np + i]; }
      for (int i = lane; i < n; i += 64) { P.dump[n * n + i] = P1[i]; P.dump[n * n + n + i] = P2[i]; P.dump[n * n + 2 * n + i] = P3[i]; P.dump[n * n + 3 * n + i] = HX[i]; }
    }
#endif
#ifdef QP_MFMA_SOLVES   // round-1 form: right-hand sides as a 16-wide tile column on the matrix cores (kept for A/B runs)
    {
      const double* vin[6] = {R1, R2, MB, MB + k.np, MB + 2 * k.np, MB + 3 * k.np};
      rhs_load<T, 2 + NB>(k, rh, vin);
    }
    int fbad = reg_factor<T>(k, acc, YL, rh, 1e-30 * dmax);
#else
    int fbad = reg_factor_only<T>(k, acc, YL, 1e-30 * dmax);
    WAVE_SYNC();
    STAMP(5);
#endif
    if (NB > 0) {   // bordered factor: u_e = U^-T m_e; S = M_bb - u'u is factorised as scalars
#ifdef QP_MFMA_SOLVES
      double* vout[6] = {R1, R2, MB, MB + k.np, MB + 2 * k.np, MB + 3 * k.np};
      rhs_store<T, 2 + NB>(k, rh, vout);
      WAVE_SYNC();
#else
#pragma unroll
      for (int e = 0; e < NB; ++e) {
        double ue[T][4];
        vec_forward<T, true>(k, acc, YL, MB + e * k.np, ue);
        WAVE_SYNC();
        vec_rows_store<T>(k, ue, MB + e * k.np);
      }
      WAVE_SYNC();
#endif
      double S[NBB][NBB];
#pragma unroll
      for (int e = 0; e < NB; ++e)
#pragma unroll
        for (int f = e; f < NB; ++f) {
          double dsum = 0.0;
          for (int h = 0; h < 2; ++h) { const int i = lane + 64 * h; if (i < nc) dsum = fma(MB[e * k.np + i], MB[f * k.np + i], dsum); }
          S[e][f] = MB[e * k.np + nc + f] - wave_sum(dsum);
        }
#pragma unroll
      for (int e = 0; e < NB; ++e) {
        double dd = S[e][e];
#pragma unroll
        for (int g2 = 0; g2 < e; ++g2) dd -= Ubb[g2][e] * Ubb[g2][e];
        if (!(dd > 1e-30 * dmax)) { if (!(fabs(dd) < INFINITY)) fbad = 1; dd = 1e-30 * dmax; }
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
#ifdef QP_MFMA_SOLVES
    if (NB > 0) {
      border_solve(R1); border_solve(R2);
      const double* vin[2] = {R1, R2};
      rhs_load<T, 2>(k, rh, vin);
    }
    reg_backward<T>(k, acc, YL, rh, SCR);
    { double* vout[2] = {R1, R2}; rhs_store<T, 2>(k, rh, vout); }
#else
    {
      double y1[T][4], y2[T][4];
      vec_forward<T, true>(k, acc, YL, R1, y1);
      vec_forward<T, true>(k, acc, YL, R2, y2);
      if (NB > 0) {
        WAVE_SYNC();
        vec_rows_store<T>(k, y1, R1); vec_rows_store<T>(k, y2, R2);
        WAVE_SYNC();
        border_solve(R1); border_solve(R2);
        vec_rows_load<T>(k, R1, y1); vec_rows_load<T>(k, R2, y2);
      }
      WAVE_SYNC();
      vec_backward<T, true>(k, acc, YL, y1, R1);
      vec_backward<T, true>(k, acc, YL, y2, R2);
    }
#endif
    WAVE_SYNC();
    return 0;
  };
  // one more solve with the resident factor: V <- M^-1 V (LDS n-vector, in place), on the VALU
  auto solve1 = [&](double* V) __attribute__((always_inline)) {
#ifdef QP_MFMA_SOLVES
    const double* vin[1] = {V}; double* vout[1] = {V};
    rhs_load<T, 1>(k, rh, vin);
    reg_forward<T>(k, acc, YL, rh);
    if (NB > 0) {
      rhs_store<T, 1>(k, rh, vout);
      WAVE_SYNC();
      border_solve(V);
      rhs_load<T, 1>(k, rh, vin);
    }
    reg_backward<T>(k, acc, YL, rh, SCR);
    rhs_store<T, 1>(k, rh, vout);
#else
    double yv[T][4];
    vec_forward<T, false>(k, acc, YL, V, yv);
    if (NB > 0) {
      WAVE_SYNC();
      vec_rows_store<T>(k, yv, V);
      WAVE_SYNC();
      border_solve(V);
      vec_rows_load<T>(k, V, yv);
    }
    WAVE_SYNC();
    vec_backward<T, false>(k, acc, YL, yv, V);
#endif
  };

  STAMP(0);
  for (it = 0; flag == 1; ++it) {
    // ================= row phase 1: residuals, weights (only on entry; afterwards fused into the update sweep) =================
    if (it == 0) {
      double s_gap = 0, m_rp = 0;
      for (int js = 0; js < JT; ++js) {
        const int ix = js * 64 + lane;
        row1_body(ix, row_valid(k, js), aL[ix], aU[ix], aV[ix], aTL[ix], aTU[ix], aZL[ix], aZU[ix], s_gap, m_rp);
      }
      gap = wave_sum(s_gap);
      rp_rel = wave_max(m_rp);
      WAVE_SYNC();
    }
    const double mu = gap / cnt;

    STAMP(1);
    // ================= pass 1: M = H + A'DA (MFMA), p1, p2, p3; Hx =================
#ifdef QP_HX_SEPARATE   // round-1 form (A/B runs): H~ read once for H~x and once more for the accumulators
    hx_full(X);
    STAMP(2);
    acc_init<T>(k, acc);
#else
    acc_init<T>(k, acc);
    hx_from_acc<T>(k, acc, X, HX);
    hx_border(X);
    STAMP(2);
#endif
    pass_syrk<T, NB>(k, acc, P1, P2, P3, MB);
    WAVE_SYNC();
    STAMP(3);
    // objective, dual residual
    double fl = 0, m_rd = 0;
    for (int h = 0; h < 2; ++h) {
      const int i = lane + 64 * h;
      if (i < n) {
        const int ix = (J + (i >> 6)) * 64 + (i & 63);
        const double gz = P3[i] + aW3[ix];
        fl += 0.5 * X[i] * HX[i] + G[i] * X[i];
        const double sc = fmax(1.0, fmax(fabs(G[i]), fmax(fabs(HX[i]), fabs(gz))));
        m_rd = fmax(m_rd, fabs(HX[i] + G[i] - gz) / sc);
      }
    }
    const double fval = wave_sum(fl);
    const double rd_rel = wave_max(m_rd);
    const double gap_rel = gap / fmax(1.0, fabs(fval));
    // (fmax drops NaN operands: an iterate with NaN in it -- a step along a direction from a broken-down factorisation, 0 * NaN --
    //  would read as merit 0.  Its objective is NaN, and so must the merit be: the best saved iterate is returned then.)
    const double merit = (fabs(fval) < INFINITY && fabs(gap_rel) < INFINITY) ? fmax(rd_rel, fmax(rp_rel, gap_rel)) : INFINITY;
    fval_s = fval; merit_s = merit;
    const bool res_ok = merit <= P.tol;
#ifdef QP_DEBUG_DUMP
    if (P.dump && P.dump_stage == 5 && b == P.dump_iter && lane == 0 && it < 120) {   // debug: iteration trace of instance `dump_iter`
      double* o_ = P.dump + 16 * it;
      o_[0] = merit; o_[1] = rd_rel; o_[2] = rp_rel; o_[3] = gap_rel; o_[4] = mu; o_[5] = fval; o_[6] = (double)have_saved; o_[7] = saved_merit;
    }
#endif
    if (!(merit < INFINITY)) { flag = have_saved ? 2 : -1; break; }
    if (merit <= P.tol_loose && merit < saved_merit) {
      for (int i = lane; i < k.np; i += 64) XS[i] = X[i];
      for (int js = 0; js < JT; ++js) LAMS[js * 64 + lane] = aW3[js * 64 + lane];
      have_saved = 1; saved_merit = merit;
    } else if (merit > P.tol_loose && rp_rel <= P.tol_loose && gap_rel <= P.tol_loose) {
      // Only the dual residual is in the way (the Newton steps lose accuracy once z/t passes ~1e19 and r_d creeps up):
      // repair the certificate of a *copy* of the iterate by moving r_d into the bound multipliers, where a finite bound
      // of the right sign exists; the copy qualifies as fall-back if its complementarity stays within tol_loose.
      double dgap = 0, m_rd2 = 0, lamfix[2] = {0.0, 0.0};
      for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        if (i < n) {
          const int ix = (J + (i >> 6)) * 64 + (i & 63);
          const double lam = aW3[ix], gz = P3[i] + lam, r = HX[i] + G[i] - gz;
          const double lam2 = lam + r, l = aL[ix], u = aU[ix], v = aV[ix];
          const bool ok = lam2 >= 0 ? l > -INFINITY : u < INFINITY;
          if (ok) { lamfix[h] = lam2; dgap += fabs(r) * fmax(0.0, lam2 >= 0 ? v - l : u - v); }
          else {
            lamfix[h] = lam;
            const double sc = fmax(1.0, fmax(fabs(G[i]), fmax(fabs(HX[i]), fabs(gz))));
            m_rd2 = fmax(m_rd2, fabs(r) / sc);
          }
        }
      }
      const double merit2 = fmax(wave_max(m_rd2), fmax(rp_rel, (gap + wave_sum(dgap)) / fmax(1.0, fabs(fval))));
      if (merit2 <= P.tol_loose && merit2 < saved_merit) {
        for (int i = lane; i < k.np; i += 64) XS[i] = X[i];
        for (int js = 0; js < J; ++js) LAMS[js * 64 + lane] = aW3[js * 64 + lane];
        for (int h = 0; h < 2; ++h) { const int i = lane + 64 * h; if (i < k.np) LAMS[(J + (i >> 6)) * 64 + (i & 63)] = i < n ? lamfix[h] : 0.0; }
        have_saved = 1; saved_merit = merit2;
      }
      if (have_saved) { flag = 2; break; }
    } else if (have_saved && merit > P.tol_loose) { flag = 2; break; }
    if (merit < 0.9 * best_res) { best_res = merit; stall = 0; } else ++stall;

    // ================= factorise (registers, MFMA) with the affine / centering right-hand sides riding along =================
    for (int h = 0; h < 2; ++h) {
      const int i = lane + 64 * h;
      if (i < k.np) {
        const int ix = (J + (i >> 6)) * 64 + (i & 63);
        R1[i] = i < n ? -(HX[i] + G[i]) + P1[i] + aW1[ix] : 0.0;
        R2[i] = i < n ? P2[i] + aW2[ix] : 0.0;
      }
    }
    STAMP(4);
    if (factor_solve2(it)) {
      flag = (res_ok || have_saved) ? 2 : -1;
      // the factorisation broke down (weights ~1e24) on an iterate that is nearly primal feasible and complementary: its working
      // set is usually the right one already, so the refinement gets a try -- it accepts nothing that is not a KKT point of the
      // full QP by a fresh evaluation (flag 4 -> 0 if accepted, else -1)
      if (flag == -1 && P.polish && rp_rel <= QP_BREAKDOWN_TRY_TOL && gap_rel <= QP_BREAKDOWN_TRY_TOL) flag = 4;
      break;
    }
    STAMP(6);
#ifdef QP_DEBUG_DUMP
    if (P.dump && b == 0 && P.dump_stage == 2 && it == P.dump_iter) {
      for (int i = lane; i < n; i += 64) { P.dump[i] = R1[i]; P.dump[n + i] = R2[i]; }
    }
#endif
    if (res_ok) {  // Newton-decrement test in the caller's coordinates
      double dm = 0, xm = 1.0;
      for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        if (i < n) { dm = fmax(dm, fabs(R1[i] * EV[i])); xm = fmax(xm, fabs(X[i] * EV[i])); }
      }
      dm = wave_max(dm); xm = wave_max(xm);
      if (dm <= P.tol_x * xm) { flag = 0; break; }
    }
    if (it >= P.max_iter) { flag = have_saved ? 2 : 1; break; }

    // ================= pass 2: va = G dxa, vc = G dxc =================
    {
      const double* vin[2] = {R1, R2}; double* rout[2] = {aVA, aVC};
      pass_Av<T, NB, 2, 1>(k, vin, rout, P1);   // fused: P1 = A~' w_cor
      for (int jb = 0; jb < k.JB; ++jb) {
        const int i = jb * 64 + lane;
        aVA[(J + jb) * 64 + lane] = i < n ? R1[i] : 0.0;
        aVC[(J + jb) * 64 + lane] = i < n ? R2[i] : 0.0;
      }
    }
    STAMP(7);
    // ================= row phase 2: affine step length, sigma, corrector weights (one sweep) =================
    // mu_aff(alpha) = [S0 + alpha S1 + alpha^2 S2]/cnt with S0 = sum t z, S1 = sum (t dz + z dt), S2 = sum dt dz
    double a_aff = 1.0, s1 = 0.0, s2 = 0.0;
    auto row2_body = [&](const Slot& r, int js) {
      const bool hl = r.valid && r.l > -INFINITY, hu = r.valid && r.u < INFINITY;
      double w = 0.0;
      if (hl) {
        const double dt = r.va + r.rpl, dz = -r.zl - (r.zl / r.tl) * dt;
        if (dt < 0) a_aff = fmin(a_aff, -r.tl / dt);
        if (dz < 0) a_aff = fmin(a_aff, -r.zl / dz);
        s1 += r.tl * dz + r.zl * dt; s2 += dt * dz;
        w -= dt * dz / r.tl;
      }
      if (hu) {
        const double dt = -r.va + r.rpu, dz = -r.zu - (r.zu / r.tu) * dt;
        if (dt < 0) a_aff = fmin(a_aff, -r.tu / dt);
        if (dz < 0) a_aff = fmin(a_aff, -r.zu / dz);
        s1 += r.tu * dz + r.zu * dt; s2 += dt * dz;
        w += dt * dz / r.tu;
      }
      if (js >= J && js < JT) aW1[r.ix] = w;   // second-order weight of the variable-bound rows (A rows: fused in pass 2)
    };
    for (int js = 0; js < JT; js += 2) {
      const Slot r0 = load_slot(js), r1 = load_slot(js + 1);
      row2_body(r0, js); row2_body(r1, js + 1);
    }
    a_aff = wave_min(a_aff);
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const double mu_aff = fmax(0.0, gap + a_aff * (s1 + a_aff * s2)) / cnt;
    double sigma = mu > 0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
    if (sigma > 1.0) sigma = 1.0;
    {
      const double mu_floor = 1e-5 * P.tol * fmax(1.0, fabs(fval)) / cnt;
      if (mu > 0 && sigma < mu_floor / mu) sigma = fmin(1.0, mu_floor / mu);
    }
    const double smu = sigma * mu;
    // the second-order term is dropped when the affine step is tiny (it then models nothing and makes the iteration
    // cycle on low-speed instances); this also saves the corrector solve and pass 3 for that iteration
    const double cw = a_aff >= 0.05 ? 1.0 : 0.0;
    WAVE_SYNC();
    STAMP(8);
    if (cw != 0.0) {
    // ================= corrector: P1 = A' w_cor came out of the fused pass 2 =================
    STAMP(9);
    for (int h = 0; h < 2; ++h) {
      const int i = lane + 64 * h;
      if (i < k.np) { const int ix = (J + (i >> 6)) * 64 + (i & 63); DX[i] = i < n ? P1[i] + aW1[ix] : 0.0; }
    }
    WAVE_SYNC();
    solve1(DX);
    WAVE_SYNC();
    STAMP(10);
    // ================= pass 4: G dx_cor =================
    {
      const double* vin[1] = {DX}; double* rout[1] = {aW2};  // W2 reused for G dx_cor
      pass_Av<T, NB, 1, 0>(k, vin, rout, nullptr);
      for (int jb = 0; jb < k.JB; ++jb) { const int i = jb * 64 + lane; aW2[(J + jb) * 64 + lane] = i < n ? DX[i] : 0.0; }
    }
    STAMP(11);
    } else {   // no corrector this iteration
      for (int i = lane; i < k.np; i += 64) DX[i] = 0.0;
      for (int js = 0; js < JT; ++js) aW2[js * 64 + lane] = 0.0;
      WAVE_SYNC();
    }
    // full direction dx = dxa + smu*dxc + dxcor ; dv likewise
    for (int h = 0; h < 2; ++h) {
      const int i = lane + 64 * h;
      if (i < n) DX[i] = R1[i] + smu * R2[i] + DX[i];
    }
    // ================= row phase 3: step length (Mehrotra heuristic on the blocking pair), update =================
    double amax = 1e300, bp = 0, bdp = 0, bd = 0, bdd = 0, q1 = 0.0, q2 = 0.0;
    auto row3a_body = [&](const Slot& r, int js) {
      const bool hl = r.valid && r.l > -INFINITY, hu = r.valid && r.u < INFINITY;
      const double dv = r.va + smu * r.vc + r.w2;
      if (js < JT) aVC[r.ix] = dv;  // keep the full G dx for the update
      if (hl) {
        const double dta = r.va + r.rpl, dza = -r.zl - (r.zl / r.tl) * dta;
        const double cl = smu - cw * dta * dza;
        const double dt = dv + r.rpl, dz = -r.zl + cl / r.tl - (r.zl / r.tl) * dt;
        if (dt < 0 && -r.tl / dt < amax) { amax = -r.tl / dt; bp = r.tl; bdp = dt; bd = r.zl; bdd = dz; }
        if (dz < 0 && -r.zl / dz < amax) { amax = -r.zl / dz; bp = r.zl; bdp = dz; bd = r.tl; bdd = dt; }
        q1 += r.tl * dz + r.zl * dt; q2 += dt * dz;
      }
      if (hu) {
        const double dta = -r.va + r.rpu, dza = -r.zu - (r.zu / r.tu) * dta;
        const double cu = smu - cw * dta * dza;
        const double dt = -dv + r.rpu, dz = -r.zu + cu / r.tu - (r.zu / r.tu) * dt;
        if (dt < 0 && -r.tu / dt < amax) { amax = -r.tu / dt; bp = r.tu; bdp = dt; bd = r.zu; bdd = dz; }
        if (dz < 0 && -r.zu / dz < amax) { amax = -r.zu / dz; bp = r.zu; bdp = dz; bd = r.tu; bdd = dt; }
        q1 += r.tu * dz + r.zu * dt; q2 += dt * dz;
      }
    };
    for (int js = 0; js < JT; js += 2) {
      const Slot r0 = load_slot(js), r1 = load_slot(js + 1);
      row3a_body(r0, js); row3a_body(r1, js + 1);
    }
    const double amax_w = wave_min(amax);
    double alpha = 1.0;
    if (amax_w < 1e299) {
      // blocking pair = the one on the lane that attains the minimum (first such lane)
      const unsigned long long msk = __ballot(amax == amax_w);
      const int src = __ffsll((long long)msk) - 1;
      bp = rl(bp, src); bdp = rl(bdp, src); bd = rl(bd, src); bdd = rl(bdd, src);
      q1 = wave_sum(q1); q2 = wave_sum(q2);
      const double gamma_f = 0.99, gamma_a = 1.0 / (1.0 - gamma_f);
      const double mufull = fmax(0.0, gap + amax_w * (q1 + amax_w * q2)) / cnt / gamma_a;
      const double a_h = (-bp + mufull / (bd + amax_w * bdd)) / bdp;
      alpha = fmin(1.0, fmin(0.99999999 * amax_w, fmax(a_h, gamma_f * amax_w)));
    }
#ifdef QP_DEBUG_DUMP
    if (P.dump && P.dump_stage == 5 && b == P.dump_iter && lane == 0 && it < 120) {
      double* o_ = P.dump + 16 * it;
      o_[8] = a_aff; o_[9] = sigma; o_[10] = alpha; o_[11] = cw; o_[12] = amax_w; o_[13] = (double)stall;
    }
    if (P.dump && b == 0 && P.dump_stage == 4 && it == P.dump_iter) {   // debug: step-length pipeline of this iteration
      double c1 = 0, c2 = 0, c3 = 0, c4 = 0;
      for (int js = 0; js < JT; ++js) { const int ix = js * 64 + lane; c1 += aVA[ix]; c2 += aVC[ix]; c3 += aW2[ix]; c4 += aW1[ix]; }
      c1 = wave_sum(c1); c2 = wave_sum(c2); c3 = wave_sum(c3); c4 = wave_sum(c4);
      if (lane == 0) {
        double* o_ = P.dump;
        o_[0] = a_aff; o_[1] = mu_aff; o_[2] = sigma; o_[3] = smu; o_[4] = cw; o_[5] = amax_w; o_[6] = alpha; o_[7] = gap; o_[8] = mu;
        o_[9] = c1; o_[10] = c2; o_[11] = c3; o_[12] = c4; o_[13] = s1; o_[14] = s2; o_[15] = q1; o_[16] = q2;
      }
      for (int i = lane; i < n; i += 64) { P.dump[32 + i] = DX[i]; P.dump[32 + n + i] = R1[i]; P.dump[32 + 2 * n + i] = R2[i]; P.dump[32 + 3 * n + i] = P1[i]; }
    }
#endif
    // update, fused with the residual / weight phase of the next iteration
    double xn = 0, zn = 0, s_gap = 0, m_rp = 0;
    auto row3b_body = [&](const Slot& r, int js) {
      if (js >= JT) return;
      const bool hl = r.valid && r.l > -INFINITY, hu = r.valid && r.u < INFINITY;
      const double dv = r.vc;   // full G dx stored by the previous sweep
      double tl = r.tl, zl = r.zl, tu = r.tu, zu = r.zu;
      if (hl) {
        const double dta = r.va + r.rpl, dza = -zl - (zl / tl) * dta;
        const double cl = smu - cw * dta * dza;
        const double dt = dv + r.rpl, dz = -zl + cl / tl - (zl / tl) * dt;
        tl += alpha * dt; zl += alpha * dz;
        aTL[r.ix] = tl; aZL[r.ix] = zl;
        zn = fmax(zn, zl);
      }
      if (hu) {
        const double dta = -r.va + r.rpu, dza = -zu - (zu / tu) * dta;
        const double cu = smu - cw * dta * dza;
        const double dt = -dv + r.rpu, dz = -zu + cu / tu - (zu / tu) * dt;
        tu += alpha * dt; zu += alpha * dz;
        aTU[r.ix] = tu; aZU[r.ix] = zu;
        zn = fmax(zn, zu);
      }
      const double v = r.v + alpha * dv;
      aV[r.ix] = v;
      row1_body(r.ix, r.valid, r.l, r.u, v, tl, tu, zl, zu, s_gap, m_rp);
    };
    for (int js = 0; js < JT; js += 2) {
      const Slot r0 = load_slot(js), r1 = load_slot(js + 1);
      row3b_body(r0, js); row3b_body(r1, js + 1);
    }
    for (int h = 0; h < 2; ++h) {
      const int i = lane + 64 * h;
      if (i < n) { X[i] += alpha * DX[i]; xn = fmax(xn, fabs(X[i])); }
    }
    const double rp_prev = rp_rel;
    gap = wave_sum(s_gap);
    rp_rel = wave_max(m_rp);
    xn = wave_max(xn); zn = wave_max(zn);
    WAVE_SYNC();
    STAMP(12);
    // divergence heuristics -> qpOASES exit codes (qpOASES.m:43-47)
    // a diverging iterate is 'unbounded' (-3) only if it is primal feasible and the objective follows it to -infinity;
    // with a primal residual it is the signature of an infeasible QP (-2); on a bounded feasible problem (every LTV-MPC QP:
    // boxed inputs, slacks with positive linear cost) it is an internal failure (-1)
    if (xn > 1e13) { flag = rp_prev > 1e-6 ? -2 : (fval < -1e13 ? -3 : -1); break; }
    if (zn > 1e15 && rp_prev > 1e-6) { flag = -2; break; }
    // once an iterate met tol_loose, a handful of non-improving iterations means the end game lost its numerical
    // footing: return the saved iterate (this also bounds the iteration tail, i.e. the kernel's drain time)
    if (stall > (have_saved ? 5 : 25)) { flag = have_saved ? 2 : (rp_prev > 1e-6 ? -2 : (P.polish ? 5 : 1)); break; }   // 5: stalled, the refinement may still certify (else 1)
  }

  // ---- outputs ----
  // The last iterate (or the best saved one) is returned whatever the exit code: the reference keeps driving on whatever the
  // solver handed back (main.m:163-175).
  const bool v_current = flag == 0 || flag == 4;   // aV still equals G x and fval_s is the objective at x (not so after a restore / an update)
  if (flag == 2) {  // restore the best iterate that met tol_loose
    for (int i = lane; i < k.np; i += 64) X[i] = XS[i];
    for (int js = 0; js < JT; ++js) aW3[js * 64 + lane] = LAMS[js * 64 + lane];
    flag = 0; merit_s = saved_merit;
    WAVE_SYNC();
  } else {
    for (int js = 0; js < JT; ++js) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(k, js);
      const bool hl = valid && aL[ix] > -INFINITY, hu = valid && aU[ix] < INFINITY;
      aW3[ix] = (hl ? aZL[ix] : 0.0) - (hu ? aZU[ix] : 0.0);
    }
    WAVE_SYNC();
  }
  // ---- active-set refinement: from the interior-point point to the vertex an active-set solver (qpOASES) stops at ----
  // Working set W from the multipliers (side active iff |lambda| exceeds its slack).  Active *bounds* are eliminated
  // exactly: the variable is pinned (huge diagonal, zero right-hand side, value reset after every update) and its
  // multiplier is read off the stationarity residual.  Active *rows* A_W z = b: conjugate gradients on the dual of the
  // augmented problem, operator S = A_W M^-1 A_W' with M = H~ + pin + rho A_W'A_W (resident Cholesky factor).  Its
  // spectrum is clustered at 1/rho plus a few small outliers from nearly dependent active rows (long stretches of the
  // horizon on a track limit): CG removes the outliers in one step each, the fixed-step method of multipliers of round 1
  // could not (rejected 10-20 % of the instances).  One fused stream over A~ per CG step (q = A_W w and A_W'q together),
  // A'p kept by recurrence.  The result is accepted only if a fresh evaluation says it is a KKT point of the full QP;
  // otherwise the interior-point iterate is returned.
#ifdef QP_DEBUG_DUMP
  if (P.dump && P.dump_stage == 5 && b == P.dump_iter && lane == 0) { double* o_ = P.dump + 16 * 120; o_[4] = (double)flag; for (int i_ = 5; i_ < 48; ++i_) o_[i_] = 0.0; }
#endif
  if ((flag == 0 || flag == 4 || flag == 5) && P.polish) {
    const double rho = 1e6, pin = 1e16, rinv = 1.0 / rho;
    double* PA = rowp(k, R_CB1); double* PB = rowp(k, R_RPL); double* PY = rowp(k, R_CC1); double* PS = rowp(k, R_CB2);
    double* PC = rowp(k, R_RPU); double* PP = rowp(k, R_CC2); double* PZ0 = aW2;   // constraint residual c, CG direction p, zeros
    double* ATR = R1; double* ATP = P3;                                               // A_W'r and A_W'p (n-vectors, by recurrence)
    if (!v_current) {
      const double* vin[1] = {X}; double* rout[1] = {aV};
      pass_Av<T, NB, 1, 0>(k, vin, rout, nullptr);
      for (int jb = 0; jb < k.JB; ++jb) { const int i = jb * 64 + lane; aV[(J + jb) * 64 + lane] = i < n ? X[i] : 0.0; }
    }
    for (int js = 0; js < JT; ++js) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(k, js);
      const double l = aL[ix], u = aU[ix], v = aV[ix], lam = aW3[ix];
      const bool lo = valid && l > -INFINITY && lam > 0 && lam > fabs(v - l);
      const bool up = valid && u < INFINITY && lam < 0 && -lam > fabs(u - v);
      PS[ix] = lo ? 1.0 : (up ? -1.0 : 0.0); PY[ix] = ((lo || up) && js < J) ? lam : 0.0;
    }
    for (int i = lane; i < k.np; i += 64) XS[i] = X[i];   // z of the refinement between attempts (the fall-back copy is no longer needed)
    WAVE_SYNC();
    // up to QP_REFINE_ATTEMPTS attempts: a refinement that ends on a violated inactive row / a multiplier of the wrong sign adds / drops
    // that one row and starts over from the point it reached (single add-drop corrections of an active-set method)
    for (int attempt = 0; attempt < QP_REFINE_ATTEMPTS && flag_polished <= 0; ++attempt) {
    for (int js = 0; js < JT; ++js) {
      const int ix = js * 64 + lane;
      const double sd = PS[ix];
      PA[ix] = sd != 0.0 ? rho : 0.0; PB[ix] = sd > 0 ? aL[ix] : (sd < 0 ? aU[ix] : 0.0);
      aD[ix] = js < J ? PA[ix] : (sd != 0.0 ? pin : 0.0); aW1[ix] = 0.0; aW2[ix] = 0.0; PC[ix] = 0.0; PP[ix] = 0.0;
    }
    WAVE_SYNC();
    acc_init<T>(k, acc);
    pass_syrk<T, NB>(k, acc, P1, P2, P3, MB);
    WAVE_SYNC();
    for (int i = lane; i < k.np; i += 64) { R1[i] = 0.0; R2[i] = 0.0; }
    WAVE_SYNC();
    bool pok = factor_solve2(-1) == 0, retry = false;
    if (!pok) { flag_polished = -5; break; }
    // z: the point reached so far (R1, R2 were the right-hand sides of the factorisation), pinned variables on their bounds
    for (int h = 0; h < 2; ++h) {
      const int i = lane + 64 * h;
      if (i < k.np) { const int ix = (J + (i >> 6)) * 64 + (i & 63); R2[i] = (i < n && PS[ix] != 0.0) ? PB[ix] : XS[i]; }
    }
    WAVE_SYNC();
    const double* cf_eval[3] = {PA, PB, PY};
    const double* cf_cg[3] = {PA, PZ0, PZ0};
    // gradient of the augmented Lagrangian at (z, y): v = A~z, y^ = y - rho c, P1 = A~'y^, P2 = rho A_W'c; then grad = H~z + g - P1
    auto eval_zy = [&]() __attribute__((always_inline)) {
      const double* vin[1] = {R2}; double* rout[2] = {aVA, aVC};
      pass_Av<T, NB, 1, 3>(k, vin, rout, P1, P2, cf_eval);
      hx_full(R2);
      WAVE_SYNC();
    };
    if (pok) {
      eval_zy();
      for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        if (i < k.np) { const int ix = (J + (i >> 6)) * 64 + (i & 63); DX[i] = (i < n && PS[ix] == 0.0) ? -(HX[i] + G[i] - P1[i]) : 0.0; }
      }
      WAVE_SYNC();
      solve1(DX);
      WAVE_SYNC();
      for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        if (i < n) { const int ix = (J + (i >> 6)) * 64 + (i & 63); if (PS[ix] == 0.0) R2[i] += DX[i]; }
      }
      WAVE_SYNC();
      eval_zy();   // c(z): aVA = A~z; P2 = rho A_W'c
      double rs_l = 0.0;
      for (int js = 0; js < J; ++js) {
        const int ix = js * 64 + lane;
        const double cc_ = PA[ix] != 0.0 ? aVA[ix] - PB[ix] : 0.0;
        PC[ix] = cc_; PP[ix] = -cc_; rs_l = fma(cc_, cc_, rs_l);
      }
      double rs = wave_sum(rs_l);
      for (int h = 0; h < 2; ++h) { const int i = lane + 64 * h; if (i < k.np) { ATR[i] = -P2[i] * rinv; ATP[i] = ATR[i]; } }
      WAVE_SYNC();
      for (int cgit = 0; cgit < 12 && pok; ++cgit) {
        double m_eq = 0.0, m_cy = 0.0;
        for (int js = 0; js < J; ++js) {
          const int ix = js * 64 + lane;
          m_eq = fmax(m_eq, fabs(PC[ix]) / fmax(1.0, fabs(PB[ix])));
          m_cy = fmax(m_cy, fabs(PC[ix] * PY[ix]));
        }
        m_eq = wave_max(m_eq); m_cy = wave_max(m_cy);
        if (m_eq <= 1e-11 && m_cy <= 1e-11 * fmax(1.0, fabs(fval_s))) break;
        if (cgit == 11) { pok = false; flag_polished = -6; break; }
        for (int h = 0; h < 2; ++h) {
          const int i = lane + 64 * h;
          if (i < k.np) { const int ix = (J + (i >> 6)) * 64 + (i & 63); DX[i] = (i < n && PS[ix] == 0.0) ? ATP[i] : 0.0; }
        }
        WAVE_SYNC();
        solve1(DX);                                   // w = M^-1 A_W'p
        WAVE_SYNC();
        {
          const double* vin[1] = {DX}; double* rout[2] = {aVA, aVC};
          pass_Av<T, NB, 1, 3>(k, vin, rout, P1, P2, cf_cg);   // aVA = A~w; P2 = rho A_W'(A_W w)
        }
        WAVE_SYNC();
        double pq_l = 0.0;
        for (int js = 0; js < J; ++js) { const int ix = js * 64 + lane; if (PA[ix] != 0.0) pq_l = fma(PP[ix], aVA[ix], pq_l); }
        const double pq = wave_sum(pq_l);
        if (!(pq > 0.0) || !(rs > 0.0)) {   // dependent / inconsistent working set: drop the row that carries the stalled direction
          pok = false; flag_polished = -7;
          if (attempt < QP_REFINE_ATTEMPTS - 1) {
            double my = 0.0; int myix = -1;
            for (int js = 0; js < J; ++js) { const int ix = js * 64 + lane; if (PA[ix] != 0.0 && fabs(PP[ix]) > my) { my = fabs(PP[ix]); myix = ix; } }
            const double mx = wave_max(my);
            if (mx > 0.0 && my == mx && myix >= 0) { PS[myix] = 0.0; PY[myix] = 0.0; }
            for (int i = lane; i < k.np; i += 64) XS[i] = R2[i];
            WAVE_SYNC();
            retry = true;
          }
          break;
        }
        const double alpha_ = rs / pq;
        double rsn_l = 0.0;
        for (int js = 0; js < J; ++js) {
          const int ix = js * 64 + lane;
          if (PA[ix] != 0.0) {
            PY[ix] = fma(alpha_, PP[ix], PY[ix]);
            const double cc_ = fma(alpha_, aVA[ix], PC[ix]);
            PC[ix] = cc_; rsn_l = fma(cc_, cc_, rsn_l);
          }
        }
        const double rsn = wave_sum(rsn_l);
        const double beta_ = rsn / rs;
        for (int js = 0; js < J; ++js) { const int ix = js * 64 + lane; if (PA[ix] != 0.0) PP[ix] = fma(beta_, PP[ix], -PC[ix]); }
        for (int h = 0; h < 2; ++h) {
          const int i = lane + 64 * h;
          if (i < k.np) {
            const int ix = (J + (i >> 6)) * 64 + (i & 63);
            if (i < n && PS[ix] == 0.0) R2[i] = fma(alpha_, DX[i], R2[i]);
            ATR[i] = fma(-alpha_ * rinv, P2[i], ATR[i]);
            ATP[i] = fma(beta_, ATP[i], ATR[i]);
          }
        }
        rs = rsn;
        WAVE_SYNC();
        // the masked residual lives in PC; aVA is overwritten by the next A~w, so the test above uses |b| and the last A~w only as scale
      }
    }
    if (pok) {
      // fresh evaluation of the candidate (z, y): everything recomputed from a stream over A~ and H~
      eval_zy();
      double m_rd = 0, m_rp = 0, m_sg = 0, m_cp = 0, fl2 = 0;
      for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        if (i < n) {
          const int ix = (J + (i >> 6)) * 64 + (i & 63);
          const double r = HX[i] + G[i] - P1[i];                    // free variable: must vanish; pinned variable: its bound multiplier
          const double sc = fmax(1.0, fmax(fabs(G[i]), fmax(fabs(HX[i]), fabs(P1[i]))));
          const double sd = PS[ix], zi = R2[i], l = aL[ix], u = aU[ix];
          if (sd == 0.0) m_rd = fmax(m_rd, fabs(r) / sc);
          else m_sg = fmax(m_sg, (sd > 0 ? -r : r) / sc);
          aVC[ix] = sd != 0.0 ? r : 0.0;                            // multiplier of the variable-bound row
          double viol = 0.0;
          if (l > -INFINITY && zi < l) viol = l - zi;
          if (u < INFINITY && zi > u) viol = fmax(viol, zi - u);
          m_rp = fmax(m_rp, viol / fmax(1.0, fabs(zi)));
          fl2 += 0.5 * zi * HX[i] + G[i] * zi + 0.0 * r;   // (0 * r: a non-finite residual must poison the sum -- fmax drops NaN operands)
        }
      }
      for (int js = 0; js < J; ++js) {
        const int ix = js * 64 + lane;
        if (row_valid(k, js)) {
          const double l = aL[ix], u = aU[ix], v = aVA[ix], y = aVC[ix], sd = PS[ix];
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
      m_rd = wave_max(m_rd); m_rp = wave_max(m_rp); m_sg = wave_max(m_sg); m_cp = wave_max(m_cp);
      const double f2 = wave_sum(fl2);
#ifdef QP_DEBUG_DUMP
      if (P.dump && b < 32 && P.dump_stage == 3 && lane == 0) { double* o_ = P.dump + 64 * b; o_[0] = m_rd; o_[1] = m_rp; o_[2] = m_sg; o_[3] = m_cp; o_[5] = f2; }
      if (P.dump && P.dump_stage == 5 && b == P.dump_iter && lane == 0 && attempt < 8) { double* o_ = P.dump + 16 * 120 + 8 + 5 * attempt; o_[0] = 1.0; o_[1] = m_rd; o_[2] = m_rp; o_[3] = m_sg; o_[4] = m_cp; }
#endif
      // acceptance: relative stationarity 1e-8 (the 1e8 slack cost of ltvmpc_*.m:35 puts cancellations of 1e8 eps into A'y of
      // the active soft rows: the floor of any fp64 evaluation of this residual; qpOASES' own terminationTolerance is
      // 5e6 eps = 1.1e-9, qpOASES_options.m:190), feasibility and complementarity 1e-10, multipliers of the right sign
      // (round-off level wrong signs are zeroed)
      pok = m_rd <= 1e-8 && m_rp <= 1e-10 && m_cp <= 1e-10 * fmax(1.0, fabs(f2)) && m_sg <= 1e-8 && fabs(f2) < INFINITY;   // (f2 is NaN if anything in the candidate is not finite)
      if (!pok) flag_polished = !(fabs(f2) < INFINITY) ? -5 : (!(m_rd <= 1e-8) ? -1 : (!(m_rp <= 1e-10) ? -2 : (!(m_sg <= 1e-8) ? -4 : -3)));
      if (!pok && m_rd <= 1e-8 && attempt < QP_REFINE_ATTEMPTS - 1 && (m_rp > 1e-10 || m_sg > 1e-8)) {
        // single correction of the working set: add the most violated inactive row, else drop the worst wrong-sign row
        double my = 0.0; int myix = -1; double myside = 0.0;
        const bool add = m_rp > 1e-10;
        for (int js = 0; js < JT; ++js) {
          const int ix = js * 64 + lane;
          if (!row_valid(k, js)) continue;
          const double l = aL[ix], u = aU[ix], sd = PS[ix];
          const double v = js < J ? aVA[ix] : R2[(js - J) * 64 + lane];
          if (add) {
            if (sd != 0.0) continue;
            double sc = fmax(1.0, fabs(v));
            if (js < J) { if (l > -INFINITY) sc = fmax(sc, fabs(l)); if (u < INFINITY) sc = fmax(sc, fabs(u)); }
            const double vl = l > -INFINITY ? (l - v) / sc : -1.0, vu = u < INFINITY ? (v - u) / sc : -1.0;
            const double vv = fmax(vl, vu);
            if (vv > my) { my = vv; myix = ix; myside = vl >= vu ? 1.0 : -1.0; }
          } else {
            if (sd == 0.0) continue;
            const double y = aVC[ix];
            const double sc = js < J ? fmax(1.0, fabs(y)) : 1.0;
            const double sg = (sd > 0 ? -y : y) / sc;
            if (sg > my) { my = sg; myix = ix; myside = 0.0; }
          }
        }
        const double mx = wave_max(my);
        if (mx > 0.0 && my == mx && myix >= 0) { PS[myix] = myside; PY[myix] = 0.0; }
        for (int js = 0; js < J; ++js) PY[js * 64 + lane] = PS[js * 64 + lane] != 0.0 ? PY[js * 64 + lane] : 0.0;
        for (int i = lane; i < k.np; i += 64) XS[i] = R2[i];
        WAVE_SYNC();
        continue;   // next attempt from the point reached (R2) with the corrected working set
      }
      if (!pok && !(m_rd <= 1e-8) && attempt < QP_REFINE_ATTEMPTS - 1 && m_rd <= 1e-4) {   // stationarity above the floor: one more exact step from here
        for (int i = lane; i < k.np; i += 64) XS[i] = R2[i];
        for (int js = 0; js < J; ++js) { const int ix = js * 64 + lane; PY[ix] = PS[ix] != 0.0 ? aVC[ix] : 0.0; }
        WAVE_SYNC();
        continue;
      }
      if (!pok) break;
      if (pok) {
        for (int i = lane; i < k.np; i += 64) X[i] = R2[i];
        for (int js = 0; js < JT; ++js) {
          const int ix = js * 64 + lane;
          const double y = aVC[ix], sd = PS[ix];
          aW3[ix] = sd > 0 ? fmax(y, 0.0) : (sd < 0 ? fmin(y, 0.0) : 0.0);
        }
        flag_polished = 1 + attempt;
        fval_s = f2; merit_s = fmax(m_rd, fmax(m_rp, m_cp / fmax(1.0, fabs(f2))));
        flag = 0;
      }
      WAVE_SYNC();
    } else if (!retry) break;
    }   // attempts
  }
  if (flag == 4) flag = -1;   // not certified
  if (flag == 5) flag = 1;
  double* xo = P.x + (size_t)b * d.nu;       // caller's indexing (QpDims::nu: dummy padding variables are skipped)
  for (int i = lane; i < n; i += 64) { const int ui = qp_user_index(d, i); if (ui >= 0) xo[ui] = X[i] * EV[i]; }
  if (P.lambda) {
    double* lo = P.lambda + (size_t)b * (d.nu + k.m);
    for (int jb = 0; jb < k.JB; ++jb) {
      const int i = jb * 64 + lane;
      const int ui = i < n ? qp_user_index(d, i) : -1;
      if (ui >= 0) lo[ui] = aW3[(J + jb) * 64 + lane] / EV[i];
    }
    for (int js = 0; js < J; ++js) {
      const int r = k.perm[js * 64 + lane];   // original row of this sorted position
      if (r >= 0) lo[d.nu + r] = aW3[js * 64 + lane] * Fs[js * 64 + lane];
    }
  }
  if (!(v_current || flag_polished > 0)) {  // objective at the returned point, in the caller's units (H~,g~ scaling is objective preserving)
    WAVE_SYNC();
    hx_full(X);
    WAVE_SYNC();
    double fl = 0;
    for (int h = 0; h < 2; ++h) { const int i = lane + 64 * h; if (i < n) fl += 0.5 * X[i] * HX[i] + G[i] * X[i]; }
    fval_s = wave_sum(fl);
  }
  STAMP(13);
  STAMP_OUT;
#ifdef QP_DEBUG_DUMP
  if (P.dump && P.dump_stage == 5 && b == P.dump_iter && lane == 0) { double* o_ = P.dump + 16 * 120; o_[0] = (double)flag; o_[1] = (double)it; o_[2] = (double)flag_polished; o_[3] = merit_s; }
#endif
  if (lane == 0) {
    P.fval[b] = fval_s;
    P.exitflag[b] = flag;
    P.iter[b] = it;
    if (P.polished) P.polished[b] = flag_polished;
    if (P.kkt) P.kkt[b] = merit_s;
  }
}

#undef aL
#undef aU
#undef aTL
#undef aTU
#undef aZL
#undef aZU
#undef aV
#undef aD
#undef aW1
#undef aW2
#undef aW3
#undef aVA
#undef aVC
#undef aRPL
#undef aRPU
#undef X
#undef G
#undef HX
#undef R1
#undef R2
#undef P1
#undef P2
#undef P3
#undef DX
#undef EV
// ---------------------------------------------------------------------------------------------
// MFMA layout self test
// ---------------------------------------------------------------------------------------------
#if QP_MAIN_TU
__global__ void mfma_selftest_kernel(const double* Am, const double* Bm, double* Cm) {
  // Am: 16x4 row-major (A[i][k]), Bm: 4x16 row-major (B[k][j]), Cm: 16x16 row-major out
  const int lane = threadIdx.x;
  const double a = Am[(lane & 15) * 4 + (lane >> 4)];
  const double bb = Bm[(lane >> 4) * 16 + (lane & 15)];
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c, 0, 0, 0);
#pragma unroll
  for (int p = 0; p < 4; ++p) Cm[((lane >> 4) + 4 * p) * 16 + (lane & 15)] = c[p];
}
#endif  // QP_MAIN_TU

}  // namespace

#if QP_MAIN_TU

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
#ifdef QP_PROBE
// Diagnostic build only: pass 1 alone (same code, no surrounding solver state) to measure what the matrix-core loop
// costs when the register allocator has nothing else to keep alive.  out[b] = cycles per pass, out[batch+b] = checksum.
template <int T, int NB> __global__ __launch_bounds__(64) void syrk_probe_kernel(QpParams P, int reps) {
  const int b = blockIdx.x;
  Ctx k;
  const QpDims& d = P.d;
  k.n = d.n; k.m = d.m; k.T = T; k.Kq = d.Kq; k.J = d.J; k.JB = d.JB; k.JT = d.J + d.JB; k.np = d.np; k.ld = 0;
  k.lane = threadIdx.x; k.c = k.lane & 15; k.q = k.lane >> 4; k.nc = d.nc; k.nb = d.nb;
  double* ws = P.ws + (size_t)b * d.ws_per_qp;
  k.Aw = ws + d.off_Aw; k.Hw = ws + d.off_Hw; k.Ab = ws + d.off_Ab; k.Hb = ws + d.off_Hb;
  k.rows = ws + d.off_rows; k.rowlen = d.rowlen; k.ntr = d.ntr;
  k.perm = reinterpret_cast<const int*>(ws + d.off_meta); k.tcs = k.perm + (size_t)(d.J > 0 ? d.J : 1) * 64;
  k.aoff = k.tcs + d.ntr; k.tend = k.aoff + d.ntr + 1;
  extern __shared__ double lds[];
  k.Ms = nullptr; k.vec = lds;
  double* MB = lds + (size_t)V_NARR * d.np;
  k.ring = MB + (size_t)(NB > 0 ? NB : 1) * d.np + T * 272; k.cof = k.ring + StreamCfg<T>::R * 128;
  for (int js = 0; js < k.JT; ++js) {
    const int ix = js * 64 + k.lane;
    rowp(k, R_D)[ix] = 1.0; rowp(k, R_W1)[ix] = 0.5; rowp(k, R_W2)[ix] = 0.25; rowp(k, R_W3)[ix] = 2.0;
  }
  __syncthreads();
  v4d acc[Tri<T>::NT];
  acc_init<T>(k, acc);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    pass_syrk<T, NB>(k, acc, vecp(k, V_P1), vecp(k, V_P2), vecp(k, V_P3), MB);
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double cs = 0;
#pragma unroll
  for (int i = 0; i < Tri<T>::NT; ++i) cs += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  cs = wave_sum(cs);
  if (k.lane == 0) { P.dump[b] = (double)(t1 - t0) / reps; P.dump[gridDim.x + b] = cs; }
}
#endif

#ifdef QP_PROBE
// Diagnostic build only: the register Cholesky alone.  out[b][0..5] = cycles of: whole reg_factor, T diag_factor calls,
// T LDS round trips (tile_store + tile_load_t), forward+backward solve of one right-hand-side column.
template <int T> __global__ __launch_bounds__(64) void factor_probe_kernel(QpParams P, int reps) {
  const int b = blockIdx.x;
  Ctx k;
  const QpDims& d = P.d;
  k.n = d.n; k.m = d.m; k.T = T; k.Kq = d.Kq; k.J = d.J; k.JB = d.JB; k.JT = d.J + d.JB; k.np = d.np; k.ld = 0;
  k.lane = threadIdx.x; k.c = k.lane & 15; k.q = k.lane >> 4; k.nc = d.nc; k.nb = d.nb;
  double* ws = P.ws + (size_t)b * d.ws_per_qp;
  k.Aw = ws + d.off_Aw; k.Hw = ws + d.off_Hw; k.Ab = ws + d.off_Ab; k.Hb = ws + d.off_Hb;
  k.rows = ws + d.off_rows; k.rowlen = d.rowlen; k.ntr = d.ntr;
  extern __shared__ double lds[];
  k.Ms = nullptr; k.vec = lds;
  double* MB = lds + (size_t)V_NARR * d.np;
  double* YL = MB + (size_t)1 * d.np;
  k.ring = YL + T * 272; k.cof = k.ring + StreamCfg<T>::R * 128;
  double* SCR = k.ring;
  v4d acc[Tri<T>::NT], rh[T];
  unsigned long long tt[4] = {0, 0, 0, 0};
  double cs = 0;
  for (int r = 0; r < reps; ++r) {
    acc_init<T>(k, acc);
#pragma unroll
    for (int K = 0; K < T; ++K) {   // make it safely positive definite: add 20 to the diagonal
#pragma unroll
      for (int p = 0; p < 4; ++p) { if (k.q + 4 * p == k.c) acc[Tri<T>::idx(K, K)][p] += 20.0; rh[K][p] = 1.0 + k.c; }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    reg_factor<T>(k, acc, YL, rh, 1e-30);
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    tt[0] += t1 - t0;
    v4d Yk;
#pragma unroll
    for (int K = 0; K < T; ++K) {
#pragma unroll
      for (int p = 0; p < 4; ++p) { Yk[p] = (k.q + 4 * p == k.c) ? 1.0 : 0.0; if (k.q + 4 * p == k.c) acc[Tri<T>::idx(K, K)][p] += 30.0; }
      diag_factor(k, acc[Tri<T>::idx(K, K)], Yk, rh[K], 1e-30);
      cs += Yk[0];
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    tt[1] += t2 - t1;
#pragma unroll
    for (int K = 0; K < T; ++K) { tile_store(k, YL + K * 272, acc[Tri<T>::idx(K, K)]); __syncthreads(); acc[Tri<T>::idx(K, K)] = tile_load_t(k, YL + K * 272); }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    tt[2] += t3 - t2;
    reg_forward<T>(k, acc, YL, rh);
    reg_backward<T>(k, acc, YL, rh, SCR);
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t4 = __builtin_amdgcn_s_memtime();
    tt[3] += t4 - t3;
#pragma unroll
    for (int K = 0; K < T; ++K) cs += rh[K][0] + acc[Tri<T>::idx(K, K)][1];
  }
  cs = wave_sum(cs);
  if (k.lane == 0) { for (int i = 0; i < 4; ++i) P.dump[(size_t)b * 8 + i] = (double)tt[i] / reps; P.dump[(size_t)b * 8 + 4] = cs; }
}
#endif

void qp_make_dims(int n, int m, QpDims* d) {
  d->n = n; d->nu = n; d->m = m;
  // 1..4 trailing variables (the slack columns of the LTV-MPC QPs: nV = 2N + 1 or 2N + 4) are a *border*: they are
  // handled on the VALU instead of costing a whole 16-wide tile row/column of MFMA work and operand traffic
  const int rem = n % 16;
  if (n >= 16 && rem >= 1 && rem <= 4) { d->T = n / 16; d->nb = rem; } else { d->T = (n + 15) / 16; d->nb = 0; }
  // The slack columns of an LTV-MPC QP are touched by most rows (every soft constraint of every stage), while the input columns of
  // a row end at its own stage: with the slack columns inside the last core tile every trip of the operand stream reaches all T
  // tiles and the structural sparsity of the condensed constraints is lost (dynamic N = 60: nV = 124 = 7 x 16 + 12: measured
  // 10.8 k instead of 4.6 k MFMAs per iteration in pass 1).  Shapes with the row / column signature of the reference's QPs
  // (kinematic: nC = 6N, nV = 2N + 1; dynamic: nC = 20N, nV = 2N + 4 -- ltvmpc_*.m:38-41, *_state_constraints.m) therefore keep
  // their 1 / 4 trailing slack columns as the border whatever nV mod 16 is.  It began as a performance policy applied only where it
  // pays (a tile row / column saved, or T >= 6); it is applied to every such shape since the sweeps of the final round-3 build: with the
  // zero-curvature, 1e8-cost slack column INSIDE the MFMA core the one-wavefront kernel lost kinematic N = 20 instance 15377 (a step of
  // the end game went non-finite; ids 0..16383 all solved, vertex rate 98.7 -> 99.0 % with the column as the border; the cost measured
  // on that shape is within noise now, 910 k vs 914 k QP/s).  FSAEMPC_SLACK_BORDER=0 switches it off, =1 is the old "where it pays"
  // rule (A/B runs and tests).
  if (d->nb == 0 && n >= 20) {
    const char* sb = getenv("FSAEMPC_SLACK_BORDER");
    const int ns = (m == 10 * (n - 4) && ((n - 4) % 2) == 0) ? 4 : ((m == 3 * (n - 1) && ((n - 1) % 2) == 0) ? 1 : 0);
    const int Tp = (n - ns + 15) / 16;
    const bool pays = !(sb && sb[0] == '1') || Tp < d->T || Tp >= 6;
    if (ns > 0 && pays && !(sb && sb[0] == '0') && Tp <= QP_MAX_T) {
      d->T = Tp; d->nb = ns;
      d->n = 16 * d->T + ns;     // the solver's variable count: the core padded with dummy variables (qp_solver.h: QpDims::nu)
    }
  }
  n = d->n;
  d->NB = d->nb == 0 ? 0 : (d->nb == 1 ? 1 : 4);
  d->nc = 16 * d->T;
  d->np = d->nc + (d->nb ? 16 : 0);
  d->Kq = (m + 3) / 4;
  d->J = (d->Kq + 15) / 16;
  d->JB = (d->np + 63) / 64;
  d->rowlen = (d->J + d->JB) * 64;
  size_t off = 0;
  d->ntr = (d->Kq + 3) / 4;   // trips of 4 k-steps (16 rows)
  d->off_Aw = off; off += (size_t)(2 * d->ntr) * d->T * 128;   // dense upper bound of the operand stream
  d->off_meta = off; off += ((size_t)(d->J > 0 ? d->J : 1) * 64 + 2 * (size_t)d->ntr + 1 + 16 + 1) / 2 + 1;   // int arrays: perm, tcs, aoff, tend
  off = (off + 1) & ~(size_t)1;   // keep every array 16-byte aligned (pairs of k-steps are read as 16-byte lane loads)
  d->off_Hw = off; off += (size_t)d->T * d->T * 4 * 64;
  d->off_gw = off; off += d->np;
  d->off_E = off; off += d->np;
  d->off_F = off; off += (size_t)(d->J > 0 ? d->J : 1) * 64;
  d->off_Ab = off; off += (size_t)4 * (d->J > 0 ? d->J : 1) * 64;
  d->off_Hb = off; off += (size_t)4 * d->np;
  d->off_rows = off; off += (size_t)R_NARR * d->rowlen;
  d->off_save = off; off += d->np + d->rowlen;
  d->off_bad = off; off += 2;
  off = (off + 1) & ~(size_t)1;
  d->off_U = off; off += (size_t)(d->T * (d->T + 1) / 2) * 256;   // (workgroup kernel)
  off = (off + 63) & ~(size_t)63;
  d->ws_per_qp = off;
  {
    d->lds_solve = ((size_t)(V_NARR + (d->NB ? d->NB : 1)) * d->np + (size_t)d->T * 272 + (size_t)(3 * d->T) * 128 + (size_t)(6 + d->NB) * 64) * sizeof(double);   // ring = StreamCfg<T>::R = 3T records
  }
  {   // workgroup solve kernel (qp_wg.hip), W = 8 wavefronts per QP
    d->W = QP_WG_W;
    d->NBk = d->nb == 0 ? 0 : 4;
    d->wg_ring = 1;
    d->lds_wg = qp_wg_lds_base_bytes(*d, d->W, d->NBk);
  }
  d->prep_tw = 16;
  for (;;) {
    d->lds_prep = ((size_t)d->np + 16 + (size_t)d->prep_tw * (4 * d->Kq + 1) + (size_t)m + 2) * sizeof(double) +
                  ((size_t)4 * d->Kq + 16 * (size_t)d->ntr + 16 + 2 * (size_t)d->ntr + 1 + (size_t)((m + 63) / 64) * 16 + 6) * sizeof(int);
    if (d->lds_prep <= 96 * 1024 || d->prep_tw == 2) break;   // keep the staging tile small enough for more than one workgroup per CU
    d->prep_tw >>= 1;
  }
}

#endif  // QP_MAIN_TU

template <int T, int NB> static hipError_t launch_solve_TN(const QpParams& P, int batch, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qp_solve_kernel<T, NB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.d.lds_solve);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((qp_solve_kernel<T, NB>), dim3(batch), dim3(64), P.d.lds_solve, st, P);
  return hipGetLastError();
}
template <int T> static hipError_t launch_solve_T(const QpParams& P, int batch, hipStream_t st) {
  switch (P.d.NB) {
    case 0: return launch_solve_TN<T, 0>(P, batch, st);
    case 1: return launch_solve_TN<T, 1>(P, batch, st);
    case 4: return launch_solve_TN<T, 4>(P, batch, st);
    default: return hipErrorInvalidValue;
  }
}

#if defined(QP_TU)
hipError_t qp_launch_solve_g1(const QpParams& P, int batch, hipStream_t st);   // T = 1..4
hipError_t qp_launch_solve_g2(const QpParams& P, int batch, hipStream_t st);   // T = 5
hipError_t qp_launch_solve_g3(const QpParams& P, int batch, hipStream_t st);   // T = 6
hipError_t qp_launch_solve_g4(const QpParams& P, int batch, hipStream_t st);   // T = 7
#if QP_TU == 1
hipError_t qp_launch_solve_g1(const QpParams& P, int batch, hipStream_t st) {
  switch (P.d.T) {
    case 1: return launch_solve_T<1>(P, batch, st);
    case 2: return launch_solve_T<2>(P, batch, st);
    case 3: return launch_solve_T<3>(P, batch, st);
    case 4: return launch_solve_T<4>(P, batch, st);
    default: return hipErrorInvalidValue;
  }
}
#elif QP_TU == 2
hipError_t qp_launch_solve_g2(const QpParams& P, int batch, hipStream_t st) { return launch_solve_T<5>(P, batch, st); }
#elif QP_TU == 3
hipError_t qp_launch_solve_g3(const QpParams& P, int batch, hipStream_t st) { return launch_solve_T<6>(P, batch, st); }
#elif QP_TU == 4
hipError_t qp_launch_solve_g4(const QpParams& P, int batch, hipStream_t st) { return launch_solve_T<7>(P, batch, st); }
#endif
#endif

#if QP_MAIN_TU
// Kernel selection.  Two solve kernels share the prep kernel, the workspace layout and the algorithm:
//   - the one-wavefront kernel of this file (T = 1..7 column tiles, border widths 0 / 1 / 4: nV <= 116), built as four
//     translation units by tile count;
//   - the workgroup-per-QP kernel of qp_wg.hip (T = 1..12, nV <= 196: eight wavefronts share one QP), five more units.
// qp_use_wavefront_kernel() holds the measured choice per shape.  Every unit goes through the build's
// assembly check (tools/check_isa_exec_prologue.py: the compiler defect behind the wrong iterates of some -O2/-O3 builds,
// DESIGN.md "Build-variant fragility: root cause") and tests/test_gpu_parity.py::test_shipped_build_matches_O1_build
// compares every instantiation of both kernels with an -O1 build on the GPU.  A development build (-DQP_WG_ONE_TU,
// `make devlib`) has its own selection of tile counts in one unit; FSAEMPC_QP_KERNEL=wg|v1 overrides the choice (A/B runs).
#define QP_V1_MAX_T 7
// Measured on MI355X at 4096 QPs per launch (profiles/round2/kernel_ab.txt): the one-wavefront kernel is 1.1-2x faster up to
// T = 7 (nV <= 116); from T = 8 on its accumulators no longer fit the register file (it spills) and the workgroup kernel wins
// (dynamic N = 60: 18.6k vs 9.6k QP/s), so T = 8 is not instantiated here.  np <= 128 is also a hard limit of this kernel
// (two lanes-worth of n-vector elements per sweep).
static bool qp_use_wavefront_kernel(const QpDims& d) {
  return d.T <= QP_V1_MAX_T;
}
bool qp_runs_wavefront_kernel(const QpDims& d) {   // what qp_launch will pick (capi.hip checks the LDS budget of that kernel)
  static const char* force = getenv("FSAEMPC_QP_KERNEL");   // A/B runs only: "wg" or "v1"
  return d.T <= QP_V1_MAX_T && ((force && force[0] == 'v') || (!(force && force[0] == 'w') && qp_use_wavefront_kernel(d)));
}
#ifdef QP_WG_ONE_TU
hipError_t qp_wg_launch_1(const QpParams& P, int batch, hipStream_t st);
static hipError_t qp_wg_launch(const QpParams& P, int batch, hipStream_t st) { return qp_wg_launch_1(P, batch, st); }
#else
// one translation unit of qp_wg.hip per tile count from T = 6 on (T = 1..5 share one: no bordered variants there): the build is
// bound by the largest kernels, and they compile side by side this way
hipError_t qp_wg_launch_1(const QpParams& P, int batch, hipStream_t st);
hipError_t qp_wg_launch_6(const QpParams& P, int batch, hipStream_t st);
hipError_t qp_wg_launch_7(const QpParams& P, int batch, hipStream_t st);
hipError_t qp_wg_launch_8(const QpParams& P, int batch, hipStream_t st);
hipError_t qp_wg_launch_9(const QpParams& P, int batch, hipStream_t st);
hipError_t qp_wg_launch_10(const QpParams& P, int batch, hipStream_t st);
hipError_t qp_wg_launch_11(const QpParams& P, int batch, hipStream_t st);
hipError_t qp_wg_launch_12(const QpParams& P, int batch, hipStream_t st);
static hipError_t qp_wg_launch(const QpParams& P, int batch, hipStream_t st) {
  switch (P.d.T) {
    case 1: case 2: case 3: case 4: case 5: return qp_wg_launch_1(P, batch, st);
    case 6: return qp_wg_launch_6(P, batch, st);
    case 7: return qp_wg_launch_7(P, batch, st);
    case 8: return qp_wg_launch_8(P, batch, st);
    case 9: return qp_wg_launch_9(P, batch, st);
    case 10: return qp_wg_launch_10(P, batch, st);
    case 11: return qp_wg_launch_11(P, batch, st);
    case 12: return qp_wg_launch_12(P, batch, st);
  }
  return hipErrorInvalidValue;
}
#endif
static int prep_thr() { static const char* e = getenv("FSAEMPC_PREP_THR"); return e ? atoi(e) : 256; }   // (A/B runs)
hipError_t qp_launch(const QpParams& P, int batch, hipStream_t st, hipEvent_t ev_mid) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qp_prep_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.d.lds_prep);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(qp_prep_kernel, dim3(batch), dim3(P.d.m > prep_thr() ? 1024 : 256), P.d.lds_prep, st, P);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (P.order) {
    hipLaunchKernelGGL(qp_order_kernel, dim3(1), dim3(1024), 0, st, P.score, P.order, batch);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (ev_mid) { e = hipEventRecord(ev_mid, st); if (e != hipSuccess) return e; }
#ifdef QP_PROBE
  if (P.dump && P.dump_stage == 7 && P.d.T == 5) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_probe_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.d.lds_solve);
    hipLaunchKernelGGL((factor_probe_kernel<5>), dim3(batch), dim3(64), P.d.lds_solve, st, P, 8);
    return hipGetLastError();
  }
  if (P.dump && P.dump_stage == 8 && P.d.T == 5 && P.d.NB == 1) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&syrk_probe_kernel<5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.d.lds_solve);
    hipLaunchKernelGGL((syrk_probe_kernel<5, 1>), dim3(batch), dim3(64), P.d.lds_solve, st, P, 16);
    return hipGetLastError();
  }
#endif
  if (!qp_runs_wavefront_kernel(P.d)) return qp_wg_launch(P, batch, st);
#if !defined(QP_TU)
  switch (P.d.T) {
#if defined(QP_ONLY_T)
    case QP_ONLY_T: return launch_solve_T<QP_ONLY_T>(P, batch, st);
#else
    case 1: return launch_solve_T<1>(P, batch, st);
    case 2: return launch_solve_T<2>(P, batch, st);
    case 3: return launch_solve_T<3>(P, batch, st);
    case 4: return launch_solve_T<4>(P, batch, st);
    case 5: return launch_solve_T<5>(P, batch, st);
    case 6: return launch_solve_T<6>(P, batch, st);
    case 7: return launch_solve_T<7>(P, batch, st);
#endif
    default: return hipErrorInvalidValue;
  }
#else
  switch (P.d.T) {
    case 1: case 2: case 3: case 4: return qp_launch_solve_g1(P, batch, st);
    case 5: return qp_launch_solve_g2(P, batch, st);
    case 6: return qp_launch_solve_g3(P, batch, st);
    case 7: return qp_launch_solve_g4(P, batch, st);
    default: return hipErrorInvalidValue;
  }
#endif
}

int qp_selftest_mfma(char* msg, int msglen) {
  double hA[64], hB[64], hC[256], ref[256];
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = (double)(1 + i * 7 + kk * 3);   // asymmetric integers
  for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = (double)(2 + kk * 11 - j * 5);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j];
    ref[i * 16 + j] = s;
  }
  double *dA = 0, *dB = 0, *dC = 0;
  if (hipMalloc(&dA, sizeof(hA)) != hipSuccess || hipMalloc(&dB, sizeof(hB)) != hipSuccess || hipMalloc(&dC, sizeof(hC)) != hipSuccess) return -1;
  (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  (void)hipMemset(dC, 0, sizeof(hC));
  hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { snprintf(msg, msglen, "selftest launch: %s", hipGetErrorString(e)); (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); return -1; }
  (void)hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (hC[i] != ref[i]) { if (!bad) snprintf(msg, msglen, "mfma layout mismatch at (%d,%d): got %g want %g", i / 16, i % 16, hC[i], ref[i]); ++bad; }
  return bad;
}
#endif  // QP_MAIN_TU
