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
//   Aw [Kq][T][64]   scaled A in MFMA-operand stream order: k-step s, column tile t, lane (c=l&15,q=l>>4)
//                    holds A~[r = q*Kq + s][16t + c]   (the K order of the MFMA is permuted so that each
//                    lane group walks a contiguous row range; every load is one coalesced 512 B line set)
//   Hw [T*T][4][64]  scaled H in accumulator (C/D) layout: tile (I,J), reg p, lane -> H~[16I+q+4p][16J+c]
//   row vectors      "owner layout" [slot][64]: slot js<J, lane (c,q) <-> row r = q*Kq + 16js + c;
//                    slots J..J+JB-1 hold the variable-bound rows i = (js-J)*64 + lane.
// fp64 MFMA lane maps (cdna_hip_programming.md section 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// C/D col = l&15, row = (l>>4) + 4*reg.  fsaempc_selftest_mfma() checks them on the device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "qp_solver.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define DEVINL __device__ __forceinline__

namespace {

DEVINL double rl(double v, int src) {  // wave-uniform broadcast of lane `src` (src must be wave-uniform)
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
DEVINL double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
DEVINL double wave_max(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
DEVINL double wave_min(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
DEVINL double grp16_sum(double v) {  // sum over the 16 lanes sharing l>>4
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}
DEVINL double q_sum(double v) {  // sum over the 4 lane groups (same l&15)
  v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
  return v;
}

template <int T> struct Tri {
  static constexpr int NT = T * (T + 1) / 2;
  __host__ __device__ static constexpr int idx(int I, int J) { return I * T - (I * (I - 1)) / 2 + (J - I); }
};

// ---------------------------------------------------------------------------------------------
// qp_prep_kernel: scaling (E columns, F rows), repack of A and H, scaled g / bounds.  One wave per QP.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void qp_prep_kernel(QpParams P) {
  const int b = blockIdx.x, lane = threadIdx.x, c = lane & 15, q = lane >> 4;
  const QpDims& d = P.d;
  const int n = d.n, m = d.m, T = d.T, Kq = d.Kq, J = d.J, JB = d.JB, np = d.np;
  const double* H = P.H + (P.shared_HA ? 0 : (size_t)b * n * n);
  const double* A = P.A + (P.shared_HA ? 0 : (size_t)b * m * n);
  const double* g = P.g + (size_t)b * n;
  double* ws = P.ws + (size_t)b * d.ws_per_qp;
  double* Aw = ws + d.off_Aw;
  double* Hw = ws + d.off_Hw;
  double* gw = ws + d.off_gw;
  double* Es = ws + d.off_E;
  double* Fs = ws + d.off_F;   // owner layout, J slots
  double* Lr = ws + d.off_rows + 0 * (size_t)d.rowlen;  // scaled lower bounds (owner layout, rows then vars)
  double* Ur = ws + d.off_rows + 1 * (size_t)d.rowlen;
  extern __shared__ double lds[];
  double* Esh = lds;            // np
  double* tile = lds + np;      // 16 x (mp+1) staging for the A transpose

  // ---- column scaling E_j = 1/sqrt(H_jj), or 1/max|A_:j| where H_jj ~ 0 (slack columns) ----
  for (int j = lane; j < np; j += 64) {
    double e = 1.0;
    if (j < n) {
      double hjj = H[(size_t)j * n + j];
      if (hjj > 1e-12) e = 1.0 / sqrt(hjj);
      else e = -1.0;  // resolved below with a wave-cooperative column max
    } else e = 0.0;   // padded columns carry zeros
    Esh[j] = e;
  }
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    if (Esh[j] < 0) {  // wave-uniform
      double cm = 0;
      for (int r = lane; r < m; r += 64) cm = fmax(cm, fabs(A[(size_t)j * m + r]));
      cm = wave_max(cm);
      __syncthreads();
      if (lane == 0) Esh[j] = cm > 1e-12 ? 1.0 / cm : 1.0;
      __syncthreads();
    }
  }
  for (int j = lane; j < np; j += 64) { Es[j] = Esh[j]; gw[j] = j < n ? g[j] * Esh[j] : 0.0; }

  // ---- row scaling F_r = 1/max_j |A[r][j] E_j| ; rows handled in owner layout ----
  for (int js = 0; js < J; ++js) {
    const int s = 16 * js + c, r = q * Kq + s;
    const bool valid = s < Kq && r < m;
    double rm = 0;
    if (valid)
      for (int j = 0; j < n; ++j) rm = fmax(rm, fabs(A[(size_t)j * m + r]) * Esh[j]);
    double f = (valid && rm > 1e-12) ? 1.0 / rm : (valid ? 1.0 : 0.0);
    Fs[js * 64 + lane] = f;
    double l = -INFINITY, u = INFINITY;
    if (valid) {
      double lr = P.lbA[(size_t)b * m + r], ur = P.ubA[(size_t)b * m + r];
      l = lr > -P.inf_bound ? lr * f : -INFINITY;
      u = ur < P.inf_bound ? ur * f : INFINITY;
    }
    Lr[js * 64 + lane] = l;
    Ur[js * 64 + lane] = u;
  }
  for (int jb = 0; jb < JB; ++jb) {
    const int i = jb * 64 + lane;
    double l = -INFINITY, u = INFINITY;
    if (i < n) {
      double lr = P.lb[(size_t)b * n + i], ur = P.ub[(size_t)b * n + i];
      l = lr > -P.inf_bound ? lr / Esh[i] : -INFINITY;
      u = ur < P.inf_bound ? ur / Esh[i] : INFINITY;
    }
    Lr[(J + jb) * 64 + lane] = l;
    Ur[(J + jb) * 64 + lane] = u;
  }
  __syncthreads();  // Fs visible (global, same wave) -- also orders the LDS tile reuse below

  // ---- A -> operand stream.  Column tile by column tile: coalesced column reads -> LDS -> lane order ----
  const int mp1 = 4 * Kq + 1;  // padded LDS row length (odd => conflict-free across the 16 columns)
  for (int t = 0; t < T; ++t) {
    for (int cc = 0; cc < 16; ++cc) {
      const int col = 16 * t + cc;
      for (int r = lane; r < 4 * Kq; r += 64) {
        double v = 0.0;
        // LDS index space is (q*Kq + s) = position in the permuted K order; r here is that position
        const int qq = r / Kq, ss = r - qq * Kq;
        const int row = qq * Kq + ss;  // == r (rows are laid out contiguously per lane group)
        if (col < n && row < m) v = A[(size_t)col * m + row] * Esh[col];
        tile[cc * mp1 + r] = v;
      }
    }
    __syncthreads();
    for (int s = 0; s < Kq; ++s) {
      const int r = q * Kq + s;
      const double f = Fs[(s >> 4) * 64 + q * 16 + (s & 15)];
      Aw[((size_t)s * T + t) * 64 + lane] = tile[c * mp1 + r] * f;
    }
    __syncthreads();
  }

  // ---- H -> accumulator-layout tiles (full T x T grid; symmetric read for coalescing) ----
  for (int I = 0; I < T; ++I)
    for (int Jt = 0; Jt < T; ++Jt)
      for (int p = 0; p < 4; ++p) {
        const int row = 16 * I + q + 4 * p, col = 16 * Jt + c;
        double v = 0.0;
        if (row < n && col < n) v = H[(size_t)row * n + col] * Esh[row] * Esh[col];  // H[col][row] == H[row][col]
        Hw[((size_t)(I * T + Jt) * 4 + p) * 64 + lane] = v;
      }
}

// ---------------------------------------------------------------------------------------------
// solve kernel
// ---------------------------------------------------------------------------------------------
struct Ctx {
  int n, m, T, Kq, J, JB, JT, np, ld, lane, c, q;
  const double* Aw; const double* Hw;
  double* rows;  // base of owner-layout row arrays
  int rowlen;
  double* Ms;    // LDS n x ld
  double* vec;   // LDS n-vectors, np each
};
enum RowArr { R_L = 0, R_U, R_TL, R_TU, R_ZL, R_ZU, R_V, R_D, R_W1, R_W2, R_W3, R_VA, R_VC, R_RPL, R_RPU, R_NARR };
enum VecArr { V_X = 0, V_G, V_HX, V_R1, V_R2, V_P1, V_P2, V_P3, V_DX, V_E, V_NARR };

DEVINL double* rowp(const Ctx& k, int arr) { return k.rows + (size_t)arr * k.rowlen; }
DEVINL double* vecp(const Ctx& k, int arr) { return k.vec + arr * k.np; }

DEVINL bool row_valid(const Ctx& k, int js) {
  if (js < k.J) { const int s = 16 * js + k.c; return s < k.Kq && (k.q * k.Kq + s) < k.m; }
  return (js - k.J) * 64 + k.lane < k.n;
}

// Hx through the full symmetric tile grid (runtime loop over tile rows keeps the register footprint small)
template <int T> DEVINL void hx_tiles(const Ctx& k, const double* X, double* HX) {
  double hx[T];
#pragma unroll
  for (int t = 0; t < T; ++t) hx[t] = 0.0;
#pragma unroll 1
  for (int I = 0; I < T; ++I) {
    double xk[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) xk[p] = X[16 * I + k.q + 4 * p];
#pragma unroll
    for (int Jt = 0; Jt < T; ++Jt) {
      const double* hp = k.Hw + ((size_t)(I * T + Jt) * 4) * 64 + k.lane;
#pragma unroll
      for (int p = 0; p < 4; ++p) hx[Jt] = fma(hp[p * 64], xk[p], hx[Jt]);
    }
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

// pass 1: acc += A~' D A~ on the matrix cores; p1 = A~'w1, p2 = A~'w2, p3 = A~'w3 on the VALU beside them
template <int T> DEVINL void pass_syrk(const Ctx& k, v4d* acc, double* P1, double* P2, double* P3) {
  const double* D = rowp(k, R_D); const double* W1 = rowp(k, R_W1);
  const double* W2 = rowp(k, R_W2); const double* W3 = rowp(k, R_W3);
  double p1[T], p2[T], p3[T], bn[T], bc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) { p1[t] = 0; p2[t] = 0; p3[t] = 0; bn[t] = 0; }
  double dn = 0, w1n = 0, w2n = 0, w3n = 0;
  if (k.Kq > 0) {
#pragma unroll
    for (int t = 0; t < T; ++t) bn[t] = k.Aw[(size_t)t * 64 + k.lane];
    const int ri = k.q * 16;
    dn = D[ri]; w1n = W1[ri]; w2n = W2[ri]; w3n = W3[ri];
  }
  for (int s = 0; s < k.Kq; ++s) {
    const double dd = dn, w1 = w1n, w2 = w2n, w3 = w3n;
#pragma unroll
    for (int t = 0; t < T; ++t) bc[t] = bn[t];
    if (s + 1 < k.Kq) {  // prefetch the next k-step while the matrix cores work on this one
      const int s1 = s + 1;
#pragma unroll
      for (int t = 0; t < T; ++t) bn[t] = k.Aw[((size_t)s1 * T + t) * 64 + k.lane];
      const int ri = (s1 >> 4) * 64 + k.q * 16 + (s1 & 15);
      dn = D[ri]; w1n = W1[ri]; w2n = W2[ri]; w3n = W3[ri];
    }
    double a[T];
#pragma unroll
    for (int t = 0; t < T; ++t) a[t] = dd * bc[t];
#pragma unroll
    for (int I = 0; I < T; ++I)
#pragma unroll
      for (int Jt = I; Jt < T; ++Jt)
        acc[Tri<T>::idx(I, Jt)] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], bc[Jt], acc[Tri<T>::idx(I, Jt)], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      p1[t] = fma(w1, bc[t], p1[t]); p2[t] = fma(w2, bc[t], p2[t]); p3[t] = fma(w3, bc[t], p3[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    double v1 = q_sum(p1[t]), v2 = q_sum(p2[t]), v3 = q_sum(p3[t]);
    if (k.q == 0) { P1[16 * t + k.c] = v1; P2[16 * t + k.c] = v2; P3[16 * t + k.c] = v3; }
  }
}

// y = A~ v for NV vectors (LDS n-vectors) -> owner-layout row arrays
template <int T, int NVEC> DEVINL void pass_Av(const Ctx& k, const double* const* vin, double* const* rout) {
  double v[NVEC][T];
#pragma unroll
  for (int e = 0; e < NVEC; ++e)
#pragma unroll
    for (int t = 0; t < T; ++t) v[e][t] = vin[e][16 * t + k.c];
  for (int js = 0; js < k.J; ++js) {
    double keep[NVEC];
#pragma unroll
    for (int e = 0; e < NVEC; ++e) keep[e] = 0.0;
    const int smax = min(16, k.Kq - 16 * js);
    for (int cc = 0; cc < smax; ++cc) {
      const int s = 16 * js + cc;
      double bc[T];
#pragma unroll
      for (int t = 0; t < T; ++t) bc[t] = k.Aw[((size_t)s * T + t) * 64 + k.lane];
#pragma unroll
      for (int e = 0; e < NVEC; ++e) {
        double dsum = 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) dsum = fma(bc[t], v[e][t], dsum);
        dsum = grp16_sum(dsum);
        if (k.c == cc) keep[e] = dsum;
      }
    }
#pragma unroll
    for (int e = 0; e < NVEC; ++e) rout[e][js * 64 + k.lane] = keep[e];
  }
}

// p = A~' w (w: owner-layout row array) -> LDS n-vector
template <int T> DEVINL void pass_Atw(const Ctx& k, const double* W, double* Pout) {
  double p[T];
#pragma unroll
  for (int t = 0; t < T; ++t) p[t] = 0.0;
  for (int s = 0; s < k.Kq; ++s) {
    const double w = W[(s >> 4) * 64 + k.q * 16 + (s & 15)];
#pragma unroll
    for (int t = 0; t < T; ++t) p[t] = fma(w, k.Aw[((size_t)s * T + t) * 64 + k.lane], p[t]);
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    double v = q_sum(p[t]);
    if (k.q == 0) Pout[16 * t + k.c] = v;
  }
}

// accumulators (upper tiles, C/D layout) -> LDS lower triangle, row-major: Ms[i*ld + j], i >= j
template <int T> DEVINL void acc_to_lds(const Ctx& k, const v4d* acc) {
#pragma unroll
  for (int I = 0; I < T; ++I)
#pragma unroll
    for (int Jt = I; Jt < T; ++Jt)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int a = 16 * I + k.q + 4 * p;  // row of the upper tile element
        const int bcol = 16 * Jt + k.c;      // its column;  M[a][bcol] == M[bcol][a], stored at row bcol
        if (bcol < k.n && a <= bcol) k.Ms[bcol * k.ld + a] = acc[Tri<T>::idx(I, Jt)][p];
      }
}

// in-LDS Cholesky of the n x n lower triangle, one wave.  Returns 0 ok / 1 non-finite pivot.
DEVINL int chol_lds(const Ctx& k, double floor_abs) {
  int bad = 0;
  for (int j = 0; j < k.n; ++j) {
    const double* rowj = k.Ms + j * k.ld;
    double s[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = k.lane + 64 * h;
      double acc = 0.0;
      if (i >= j && i < k.n) {
        const double* rowi = k.Ms + i * k.ld;
        acc = rowi[j];
        int kk = 0;
        for (; kk + 1 < j; kk += 2) { acc = fma(-rowi[kk], rowj[kk], acc); acc = fma(-rowi[kk + 1], rowj[kk + 1], acc); }
        if (kk < j) acc = fma(-rowi[kk], rowj[kk], acc);
      }
      s[h] = acc;
    }
    double piv = rl(j < 64 ? s[0] : s[1], j & 63);
    if (!(piv > floor_abs)) { if (!(fabs(piv) < INFINITY)) bad = 1; piv = floor_abs; }
    const double inv = 1.0 / sqrt(piv);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = k.lane + 64 * h;
      if (i == j) k.Ms[i * k.ld + j] = piv * inv;  // sqrt(piv)
      else if (i > j && i < k.n) k.Ms[i * k.ld + j] = s[h] * inv;
    }
    __syncthreads();
  }
  return bad;
}

// solve L L' x = r for NR right-hand sides held in LDS n-vectors (in place); x kept in registers during sweeps
template <int NR> DEVINL void chol_solve_lds(const Ctx& k, double* const* R) {
  double r[NR][2];
#pragma unroll
  for (int e = 0; e < NR; ++e)
#pragma unroll
    for (int h = 0; h < 2; ++h) { const int i = k.lane + 64 * h; r[e][h] = i < k.n ? R[e][i] : 0.0; }
  double dg[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) { const int i = k.lane + 64 * h; dg[h] = i < k.n ? 1.0 / k.Ms[i * k.ld + i] : 0.0; }
  // forward: L y = r  (column oriented; column kk of L is read with stride ld)
  for (int kk = 0; kk < k.n; ++kk) {
    const double dk = rl(kk < 64 ? dg[0] : dg[1], kk & 63);
    double y[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) y[e] = rl(kk < 64 ? r[e][0] : r[e][1], kk & 63) * dk;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = k.lane + 64 * h;
      if (i > kk && i < k.n) {
        const double lik = k.Ms[i * k.ld + kk];
#pragma unroll
        for (int e = 0; e < NR; ++e) r[e][h] = fma(-lik, y[e], r[e][h]);
      } else if (i == kk) {
#pragma unroll
        for (int e = 0; e < NR; ++e) r[e][h] = y[e];
      }
    }
  }
  // backward: L' x = y  (row kk of L is contiguous)
  for (int kk = k.n - 1; kk >= 0; --kk) {
    const double dk = rl(kk < 64 ? dg[0] : dg[1], kk & 63);
    double x[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) x[e] = rl(kk < 64 ? r[e][0] : r[e][1], kk & 63) * dk;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = k.lane + 64 * h;
      if (i < kk) {
        const double lki = k.Ms[kk * k.ld + i];
#pragma unroll
        for (int e = 0; e < NR; ++e) r[e][h] = fma(-lki, x[e], r[e][h]);
      } else if (i == kk) {
#pragma unroll
        for (int e = 0; e < NR; ++e) r[e][h] = x[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < NR; ++e)
#pragma unroll
    for (int h = 0; h < 2; ++h) { const int i = k.lane + 64 * h; if (i < k.n) R[e][i] = r[e][h]; }
}

struct RowSide { double t, z, dt, dz; };

template <int T> __global__ __launch_bounds__(64) void qp_solve_kernel(QpParams P) {
  const int b = blockIdx.x;
  Ctx k;
  const QpDims& d = P.d;
  k.n = d.n; k.m = d.m; k.T = T; k.Kq = d.Kq; k.J = d.J; k.JB = d.JB; k.JT = d.J + d.JB; k.np = d.np; k.ld = d.ld;
  k.lane = threadIdx.x; k.c = k.lane & 15; k.q = k.lane >> 4;
  double* ws = P.ws + (size_t)b * d.ws_per_qp;
  k.Aw = ws + d.off_Aw; k.Hw = ws + d.off_Hw;
  k.rows = ws + d.off_rows; k.rowlen = d.rowlen;
  extern __shared__ double lds[];
  k.Ms = lds; k.vec = lds + (size_t)d.n * d.ld;
  const double* gw = ws + d.off_gw;
  const double* Es = ws + d.off_E;
  const double* Fs = ws + d.off_F;
  const int lane = k.lane, n = k.n, JT = k.JT, J = k.J;
  constexpr int NT = Tri<T>::NT;

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
        if (valid) { if (l > -INFINITY && xi < l) xi = l; if (u < INFINITY && xi > u) xi = u; }
        X[i] = xi;
      }
    }
  }
  const double cnt = fmax(1.0, wave_sum((double)cnt_local));
  infeas = wave_max((double)infeas) > 0;
  __syncthreads();

  int flag = 1, it = 0;
  double fval_s = 0.0;
  if (infeas) { flag = -2; }

  // ---- v = G x ----
  {
    const double* vin[1] = {X}; double* rout[1] = {aV};
    pass_Av<T, 1>(k, vin, rout);
    for (int jb = 0; jb < k.JB; ++jb) { const int i = jb * 64 + lane; aV[(J + jb) * 64 + lane] = i < n ? X[i] : 0.0; }
  }
  // ---- initial slacks / multipliers: t = max(resid,1), z = 1 on general rows ----
  for (int js = 0; js < JT; ++js) {
    const int ix = js * 64 + lane;
    const bool valid = row_valid(k, js);
    const double l = aL[ix], u = aU[ix], v = aV[ix];
    const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
    aTL[ix] = hl ? fmax(v - l, 1.0) : 1.0;
    aTU[ix] = hu ? fmax(u - v, 1.0) : 1.0;
    aZL[ix] = hl ? 1.0 : 0.0;
    aZU[ix] = hu ? 1.0 : 0.0;
    aW3[ix] = (js < J) ? ((hl ? 1.0 : 0.0) - (hu ? 1.0 : 0.0)) : 0.0;
  }
  __syncthreads();
  // bound multipliers absorb the initial dual residual r = Hx + g - A'(zl - zu)
  {
    hx_tiles<T>(k, X, HX);
    pass_Atw<T>(k, aW3, P3);
    __syncthreads();
    for (int jb = 0; jb < k.JB; ++jb) {
      const int i = jb * 64 + lane, ix = (J + jb) * 64 + lane;
      if (i < n) {
        const double r = HX[i] + G[i] - P3[i];
        if (aL[ix] > -INFINITY) aZL[ix] = fmax(r, 0.0) + 1.0;
        if (aU[ix] < INFINITY) aZU[ix] = fmax(-r, 0.0) + 1.0;
      }
    }
    __syncthreads();
  }

  // fall-back iterate (best one that met tol_loose)
  double saved_merit = INFINITY, best_res = INFINITY;
  int have_saved = 0, stall = 0;
  double* XS = ws + d.off_save;            // np
  double* LAMS = ws + d.off_save + k.np;   // rowlen

  for (it = 0; flag == 1; ++it) {
    // ================= row phase 1: residuals, weights =================
    double s_gap = 0, m_rp = 0;
    for (int js = 0; js < JT; ++js) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(k, js);
      const double l = aL[ix], u = aU[ix], v = aV[ix];
      const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
      const double tl = aTL[ix], tu = aTU[ix], zl = aZL[ix], zu = aZU[ix];
      const double rpl = hl ? v - l - tl : 0.0, rpu = hu ? u - v - tu : 0.0;
      const double dl_ = hl ? zl / tl : 0.0, du_ = hu ? zu / tu : 0.0;
      aRPL[ix] = rpl; aRPU[ix] = rpu;
      aD[ix] = dl_ + du_;
      aW1[ix] = -dl_ * rpl + du_ * rpu;                              // affine rhs weight
      aW2[ix] = (hl ? 1.0 / tl : 0.0) - (hu ? 1.0 / tu : 0.0);      // centering weight (times sigma*mu)
      aW3[ix] = (hl ? zl : 0.0) - (hu ? zu : 0.0);                  // current multiplier (for the dual residual)
      s_gap += (hl ? tl * zl : 0.0) + (hu ? tu * zu : 0.0);
      const double sc = fmax(1.0, fabs(v));
      if (hl) m_rp = fmax(m_rp, fabs(rpl) / fmax(sc, fabs(l)));
      if (hu) m_rp = fmax(m_rp, fabs(rpu) / fmax(sc, fabs(u)));
    }
    const double gap = wave_sum(s_gap);
    const double mu = gap / cnt;
    const double rp_rel = wave_max(m_rp);
    __syncthreads();

    // ================= pass 1: M = H + A'DA (MFMA), p1, p2, p3; Hx =================
    hx_tiles<T>(k, X, HX);
    v4d acc[NT];
    acc_init<T>(k, acc);
    pass_syrk<T>(k, acc, P1, P2, P3);
    __syncthreads();
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
    const double merit = fmax(rd_rel, fmax(rp_rel, gap_rel));
    fval_s = fval;
    const bool res_ok = merit <= P.tol;
    if (!(merit < INFINITY)) { flag = have_saved ? 2 : -1; break; }
    if (merit <= P.tol_loose && merit < saved_merit) {
      for (int i = lane; i < k.np; i += 64) XS[i] = X[i];
      for (int js = 0; js < JT; ++js) LAMS[js * 64 + lane] = aW3[js * 64 + lane];
      have_saved = 1; saved_merit = merit;
    } else if (have_saved && merit > P.tol_loose) { flag = 2; break; }
    if (merit < 0.9 * best_res) { best_res = merit; stall = 0; } else ++stall;

    // ================= factorise =================
    acc_to_lds<T>(k, acc);
    __syncthreads();
    double dmax_l = 0;
    for (int h = 0; h < 2; ++h) {
      const int i = lane + 64 * h;
      if (i < n) {
        const int ix = (J + (i >> 6)) * 64 + (i & 63);
        const double mii = k.Ms[i * k.ld + i] + aD[ix];
        k.Ms[i * k.ld + i] = mii;
        dmax_l = fmax(dmax_l, mii);
      }
    }
    const double dmax = wave_max(dmax_l);
    __syncthreads();
    if (P.dump && b == 0 && P.dump_stage == 1 && it == P.dump_iter) {  // debug: M, p1, p2, p3, Hx
      for (int i = lane; i < n * n; i += 64) { const int r = i / n, cc = i % n; P.dump[i] = r >= cc ? k.Ms[r * k.ld + cc] : k.Ms[cc * k.ld + r]; }
      for (int i = lane; i < n; i += 64) { P.dump[n * n + i] = P1[i]; P.dump[n * n + n + i] = P2[i]; P.dump[n * n + 2 * n + i] = P3[i]; P.dump[n * n + 3 * n + i] = HX[i]; }
    }
    if (chol_lds(k, 1e-30 * dmax)) { flag = (res_ok || have_saved) ? 2 : -1; break; }

    // ================= affine + centering directions =================
    for (int h = 0; h < 2; ++h) {
      const int i = lane + 64 * h;
      if (i < n) {
        const int ix = (J + (i >> 6)) * 64 + (i & 63);
        R1[i] = -(HX[i] + G[i]) + P1[i] + aW1[ix];
        R2[i] = P2[i] + aW2[ix];
      }
    }
    __syncthreads();
    { double* R[2] = {R1, R2}; chol_solve_lds<2>(k, R); }
    __syncthreads();
    if (P.dump && b == 0 && P.dump_stage == 2 && it == P.dump_iter) {
      for (int i = lane; i < n; i += 64) { P.dump[i] = R1[i]; P.dump[n + i] = R2[i]; }
    }
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
      pass_Av<T, 2>(k, vin, rout);
      for (int jb = 0; jb < k.JB; ++jb) {
        const int i = jb * 64 + lane;
        aVA[(J + jb) * 64 + lane] = i < n ? R1[i] : 0.0;
        aVC[(J + jb) * 64 + lane] = i < n ? R2[i] : 0.0;
      }
    }
    // ================= row phase 2: affine step length, sigma, corrector weights =================
    double a_aff = 1.0;
    for (int js = 0; js < JT; ++js) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(k, js);
      const double l = aL[ix], u = aU[ix];
      const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
      const double va = aVA[ix];
      if (hl) {
        const double tl = aTL[ix], zl = aZL[ix], dt = va + aRPL[ix], dz = -zl - (zl / tl) * dt;
        if (dt < 0) a_aff = fmin(a_aff, -tl / dt);
        if (dz < 0) a_aff = fmin(a_aff, -zl / dz);
      }
      if (hu) {
        const double tu = aTU[ix], zu = aZU[ix], dt = -va + aRPU[ix], dz = -zu - (zu / tu) * dt;
        if (dt < 0) a_aff = fmin(a_aff, -tu / dt);
        if (dz < 0) a_aff = fmin(a_aff, -zu / dz);
      }
    }
    a_aff = wave_min(a_aff);
    double s_mu_aff = 0;
    for (int js = 0; js < JT; ++js) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(k, js);
      const double l = aL[ix], u = aU[ix];
      const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
      const double va = aVA[ix];
      if (hl) {
        const double tl = aTL[ix], zl = aZL[ix], dt = va + aRPL[ix], dz = -zl - (zl / tl) * dt;
        s_mu_aff += (tl + a_aff * dt) * (zl + a_aff * dz);
      }
      if (hu) {
        const double tu = aTU[ix], zu = aZU[ix], dt = -va + aRPU[ix], dz = -zu - (zu / tu) * dt;
        s_mu_aff += (tu + a_aff * dt) * (zu + a_aff * dz);
      }
    }
    const double mu_aff = wave_sum(s_mu_aff) / cnt;
    double sigma = mu > 0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
    if (sigma > 1.0) sigma = 1.0;
    {
      const double mu_floor = 1e-5 * P.tol * fmax(1.0, fabs(fval)) / cnt;
      if (mu > 0 && sigma < mu_floor / mu) sigma = fmin(1.0, mu_floor / mu);
    }
    const double smu = sigma * mu;
    // second-order weights: w = -(dt_a dz_a)/t per side (the sigma*mu centering part is smu * dxc)
    for (int js = 0; js < JT; ++js) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(k, js);
      const double l = aL[ix], u = aU[ix];
      const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
      const double va = aVA[ix];
      double w = 0.0;
      if (hl) { const double tl = aTL[ix], zl = aZL[ix], dt = va + aRPL[ix], dz = -zl - (zl / tl) * dt; w -= dt * dz / tl; }
      if (hu) { const double tu = aTU[ix], zu = aZU[ix], dt = -va + aRPU[ix], dz = -zu - (zu / tu) * dt; w += dt * dz / tu; }
      aW1[ix] = w;
    }
    __syncthreads();
    // ================= pass 3: p = A' w_cor ; solve for the corrector part =================
    pass_Atw<T>(k, aW1, P1);
    __syncthreads();
    for (int h = 0; h < 2; ++h) {
      const int i = lane + 64 * h;
      if (i < n) { const int ix = (J + (i >> 6)) * 64 + (i & 63); DX[i] = P1[i] + aW1[ix]; }
    }
    __syncthreads();
    { double* R[1] = {DX}; chol_solve_lds<1>(k, R); }
    __syncthreads();
    // ================= pass 4: G dx_cor =================
    {
      const double* vin[1] = {DX}; double* rout[1] = {aW2};  // W2 reused for G dx_cor
      pass_Av<T, 1>(k, vin, rout);
      for (int jb = 0; jb < k.JB; ++jb) { const int i = jb * 64 + lane; aW2[(J + jb) * 64 + lane] = i < n ? DX[i] : 0.0; }
    }
    // full direction dx = dxa + smu*dxc + dxcor ; dv likewise
    for (int h = 0; h < 2; ++h) {
      const int i = lane + 64 * h;
      if (i < n) DX[i] = R1[i] + smu * R2[i] + DX[i];
    }
    // ================= row phase 3: step length (Mehrotra heuristic on the blocking pair), update =================
    double amax = 1e300, bp = 0, bdp = 0, bd = 0, bdd = 0;
    for (int js = 0; js < JT; ++js) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(k, js);
      const double l = aL[ix], u = aU[ix];
      const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
      const double va = aVA[ix];
      const double dv = va + smu * aVC[ix] + aW2[ix];
      aVC[ix] = dv;  // keep the full G dx for the update
      if (hl) {
        const double tl = aTL[ix], zl = aZL[ix];
        const double dta = va + aRPL[ix], dza = -zl - (zl / tl) * dta;
        const double cl = smu - dta * dza;
        const double dt = dv + aRPL[ix], dz = -zl + cl / tl - (zl / tl) * dt;
        if (dt < 0 && -tl / dt < amax) { amax = -tl / dt; bp = tl; bdp = dt; bd = zl; bdd = dz; }
        if (dz < 0 && -zl / dz < amax) { amax = -zl / dz; bp = zl; bdp = dz; bd = tl; bdd = dt; }
      }
      if (hu) {
        const double tu = aTU[ix], zu = aZU[ix];
        const double dta = -va + aRPU[ix], dza = -zu - (zu / tu) * dta;
        const double cu = smu - dta * dza;
        const double dt = -dv + aRPU[ix], dz = -zu + cu / tu - (zu / tu) * dt;
        if (dt < 0 && -tu / dt < amax) { amax = -tu / dt; bp = tu; bdp = dt; bd = zu; bdd = dz; }
        if (dz < 0 && -zu / dz < amax) { amax = -zu / dz; bp = zu; bdp = dz; bd = tu; bdd = dt; }
      }
    }
    const double amax_w = wave_min(amax);
    double alpha = 1.0;
    if (amax_w < 1e299) {
      // blocking pair = the one on the lane that attains the minimum (first such lane)
      const unsigned long long msk = __ballot(amax == amax_w);
      const int src = __ffsll((long long)msk) - 1;
      bp = rl(bp, src); bdp = rl(bdp, src); bd = rl(bd, src); bdd = rl(bdd, src);
      double s_full = 0;
      for (int js = 0; js < JT; ++js) {
        const int ix = js * 64 + lane;
        const bool valid = row_valid(k, js);
        const double l = aL[ix], u = aU[ix];
        const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
        const double va = aVA[ix], dv = aVC[ix];
        if (hl) {
          const double tl = aTL[ix], zl = aZL[ix];
          const double dta = va + aRPL[ix], dza = -zl - (zl / tl) * dta;
          const double cl = smu - dta * dza;
          const double dt = dv + aRPL[ix], dz = -zl + cl / tl - (zl / tl) * dt;
          s_full += (tl + amax_w * dt) * (zl + amax_w * dz);
        }
        if (hu) {
          const double tu = aTU[ix], zu = aZU[ix];
          const double dta = -va + aRPU[ix], dza = -zu - (zu / tu) * dta;
          const double cu = smu - dta * dza;
          const double dt = -dv + aRPU[ix], dz = -zu + cu / tu - (zu / tu) * dt;
          s_full += (tu + amax_w * dt) * (zu + amax_w * dz);
        }
      }
      const double gamma_f = 0.99, gamma_a = 1.0 / (1.0 - gamma_f);
      const double mufull = wave_sum(s_full) / cnt / gamma_a;
      const double a_h = (-bp + mufull / (bd + amax_w * bdd)) / bdp;
      alpha = fmin(1.0, fmin(0.99999999 * amax_w, fmax(a_h, gamma_f * amax_w)));
    }
    // update
    for (int js = 0; js < JT; ++js) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(k, js);
      const double l = aL[ix], u = aU[ix];
      const bool hl = valid && l > -INFINITY, hu = valid && u < INFINITY;
      const double va = aVA[ix], dv = aVC[ix];
      if (hl) {
        const double tl = aTL[ix], zl = aZL[ix];
        const double dta = va + aRPL[ix], dza = -zl - (zl / tl) * dta;
        const double cl = smu - dta * dza;
        const double dt = dv + aRPL[ix], dz = -zl + cl / tl - (zl / tl) * dt;
        aTL[ix] = tl + alpha * dt; aZL[ix] = zl + alpha * dz;
      }
      if (hu) {
        const double tu = aTU[ix], zu = aZU[ix];
        const double dta = -va + aRPU[ix], dza = -zu - (zu / tu) * dta;
        const double cu = smu - dta * dza;
        const double dt = -dv + aRPU[ix], dz = -zu + cu / tu - (zu / tu) * dt;
        aTU[ix] = tu + alpha * dt; aZU[ix] = zu + alpha * dz;
      }
      aV[ix] = aV[ix] + alpha * dv;
    }
    double xn = 0, zn = 0;
    for (int h = 0; h < 2; ++h) {
      const int i = lane + 64 * h;
      if (i < n) { X[i] += alpha * DX[i]; xn = fmax(xn, fabs(X[i])); }
    }
    for (int js = 0; js < JT; ++js) {
      const int ix = js * 64 + lane;
      if (row_valid(k, js)) zn = fmax(zn, fmax(aL[ix] > -INFINITY ? aZL[ix] : 0.0, aU[ix] < INFINITY ? aZU[ix] : 0.0));
    }
    xn = wave_max(xn); zn = wave_max(zn);
    __syncthreads();
    // divergence heuristics -> qpOASES exit codes (qpOASES.m:43-47)
    if (xn > 1e13) { flag = -3; break; }
    if (zn > 1e15 && rp_rel > 1e-6) { flag = -2; break; }
    if (stall > 25) { flag = rp_rel > 1e-6 ? -2 : (have_saved ? 2 : 1); break; }
  }

  // ---- outputs ----
  if (flag == 2) {  // restore the best iterate that met tol_loose
    for (int i = lane; i < k.np; i += 64) X[i] = XS[i];
    for (int js = 0; js < JT; ++js) aW3[js * 64 + lane] = LAMS[js * 64 + lane];
    flag = 0;
    __syncthreads();
  } else if (flag == 0 || flag == 1) {
    for (int js = 0; js < JT; ++js) {
      const int ix = js * 64 + lane;
      const bool valid = row_valid(k, js);
      const bool hl = valid && aL[ix] > -INFINITY, hu = valid && aU[ix] < INFINITY;
      aW3[ix] = (hl ? aZL[ix] : 0.0) - (hu ? aZU[ix] : 0.0);
    }
    __syncthreads();
  }
  const bool have_x = flag == 0 || flag == 1;
  double* xo = P.x + (size_t)b * n;
  for (int i = lane; i < n; i += 64) xo[i] = have_x ? X[i] * EV[i] : NAN;
  if (P.lambda) {
    double* lo = P.lambda + (size_t)b * (n + k.m);
    for (int jb = 0; jb < k.JB; ++jb) {
      const int i = jb * 64 + lane;
      if (i < n) lo[i] = have_x ? aW3[(J + jb) * 64 + lane] / EV[i] : NAN;
    }
    for (int js = 0; js < J; ++js) {
      const int s = 16 * js + k.c, r = k.q * k.Kq + s;
      if (s < k.Kq && r < k.m) lo[n + r] = have_x ? aW3[js * 64 + lane] * Fs[js * 64 + lane] : NAN;
    }
  }
  if (have_x) {  // objective at the returned point, in the caller's units (H~,g~ scaling is objective preserving)
    __syncthreads();
    hx_tiles<T>(k, X, HX);
    __syncthreads();
    double fl = 0;
    for (int h = 0; h < 2; ++h) { const int i = lane + 64 * h; if (i < n) fl += 0.5 * X[i] * HX[i] + G[i] * X[i]; }
    fval_s = wave_sum(fl);
  }
  if (lane == 0) {
    P.fval[b] = have_x ? fval_s : NAN;
    P.exitflag[b] = flag;
    P.iter[b] = it;
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

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void qp_make_dims(int n, int m, QpDims* d) {
  d->n = n; d->m = m;
  d->T = (n + 15) / 16; d->np = 16 * d->T;
  d->Kq = (m + 3) / 4;
  d->J = (d->Kq + 15) / 16;
  d->JB = (d->np + 63) / 64;
  d->ld = n | 1;
  d->rowlen = (d->J + d->JB) * 64;
  size_t off = 0;
  d->off_Aw = off; off += (size_t)d->Kq * d->T * 64;
  d->off_Hw = off; off += (size_t)d->T * d->T * 4 * 64;
  d->off_gw = off; off += d->np;
  d->off_E = off; off += d->np;
  d->off_F = off; off += (size_t)(d->J > 0 ? d->J : 1) * 64;
  d->off_rows = off; off += (size_t)R_NARR * d->rowlen;
  d->off_save = off; off += d->np + d->rowlen;
  off = (off + 63) & ~(size_t)63;
  d->ws_per_qp = off;
  d->lds_solve = ((size_t)n * d->ld + (size_t)V_NARR * d->np) * sizeof(double);
  d->lds_prep = ((size_t)d->np + 16 * (size_t)(4 * d->Kq + 1)) * sizeof(double);
}

template <int T> static hipError_t launch_solve_T(const QpParams& P, int batch, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qp_solve_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.d.lds_solve);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(qp_solve_kernel<T>, dim3(batch), dim3(64), P.d.lds_solve, st, P);
  return hipGetLastError();
}

hipError_t qp_launch(const QpParams& P, int batch, hipStream_t st, hipEvent_t ev_mid) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qp_prep_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.d.lds_prep);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(qp_prep_kernel, dim3(batch), dim3(64), P.d.lds_prep, st, P);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (ev_mid) { e = hipEventRecord(ev_mid, st); if (e != hipSuccess) return e; }
  switch (P.d.T) {
    case 1: return launch_solve_T<1>(P, batch, st);
    case 2: return launch_solve_T<2>(P, batch, st);
    case 3: return launch_solve_T<3>(P, batch, st);
    case 4: return launch_solve_T<4>(P, batch, st);
    case 5: return launch_solve_T<5>(P, batch, st);
    case 6: return launch_solve_T<6>(P, batch, st);
    case 7: return launch_solve_T<7>(P, batch, st);
    case 8: return launch_solve_T<8>(P, batch, st);
    default: return hipErrorInvalidValue;
  }
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
