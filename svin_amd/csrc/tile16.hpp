// 16 x 16 tile helpers shared by the dense solvers (kernels.hip) and the marginalisation kernels (marg.hip): the diagonal-tile
// factorisation in the accumulator layout of v_mfma_f64_16x16x4 and the cross-lane moves it is built from.  Moved here verbatim from
// kernels.hip in round 5 (the certified Cholesky route of the prior, k_marg_final_chol, factorises its diagonal tiles with it).
#pragma once
#include <hip/hip_runtime.h>

namespace svin {

typedef double d4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlaneD(double v, int srcLane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srcLane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srcLane);
  return __hiloint2double(hi, lo);
}
// 1/sqrt(x) for normal positive x: v_rsq_f64 and two Newton steps (the pivots of S are far from the denormals)
__device__ __forceinline__ double rsqrtNewton(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  double e = __builtin_fma(-h * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-h * y, y, 0.5);
  return __builtin_fma(y, e, y);
}
constexpr int kPanelLd = 17;  // leading dimension of the 16x16 LDS tiles of the dense solvers
__device__ __forceinline__ void allGatherRows(double v, double (&out)[4]) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);  // rows [v0 v0 v2 v2], [v1 v1 v3 v3]
  const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const auto l02 = __builtin_amdgcn_permlane32_swap(l16[0], l16[0], false, false);  // [v0 x4], [v2 x4]
  const auto h02 = __builtin_amdgcn_permlane32_swap(h16[0], h16[0], false, false);
  const auto l13 = __builtin_amdgcn_permlane32_swap(l16[1], l16[1], false, false);  // [v1 x4], [v3 x4]
  const auto h13 = __builtin_amdgcn_permlane32_swap(h16[1], h16[1], false, false);
  out[0] = __hiloint2double((int)h02[0], (int)l02[0]);
  out[2] = __hiloint2double((int)h02[1], (int)l02[1]);
  out[1] = __hiloint2double((int)h13[0], (int)l13[0]);
  out[3] = __hiloint2double((int)h13[1], (int)l13[1]);
}
__device__ __forceinline__ double selectByRow(int g, double v0, double v1, double v2, double v3) {
  double v = v0;
  v = (g == 1) ? v1 : v;
  v = (g == 2) ? v2 : v;
  v = (g == 3) ? v3 : v;
  return v;
}
// 1/x for a pivot: v_rcp_f64 (>= 24 bits) and r (1 + e + e^2), e = 1 - x r: relative error e^3, one dependent
// operation less than two Newton steps
__device__ __forceinline__ double rcpPivot(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
// Factorises the tile D (16 x kPanelLd in LDS, full symmetric block) in place: lower triangle <- L, strict upper
// triangle <- transposed strict lower triangle of L^-1 (D[c][r] = Linv[r][c], c < r); dinv[i] = 1/L_ii.
// The tile lives in the accumulator layout of v_mfma_f64_16x16x4 (lane = 16 g + c, register r = entry (row g + 4r,
// column c)), so register b of the four lane rows IS the 4 x 16 row block of pivots 4b .. 4b+3.  Per block of four
// pivots: the 4 x 4 diagonal block is read with v_readlane (uniform) and factorised redundantly by every lane
// (square-root-free: A = Lt D Lt^T, Lt unit lower, one reciprocal per pivot on the serial chain), the row block is
// all-gathered across the four lane rows with gfx950's v_permlane16/32_swap (6 swaps), every lane finishes the four
// pivot rows at its column with the uniform multipliers, and the rank-4 trailing update is ONE MFMA.  The same row
// operations applied to a running identity give Lt^-1 (second MFMA, same A operand), so the inverse needs no second
// pass.  One rsqrt per lane scales both factors on their way to LDS: L = Lt D^1/2, L^-1 = D^-1/2 Lt^-1.
// (2.5 k cycles; the one-row-per-lane routine it replaces -- pivot row through 2(16-k) v_readlane per pivot, every
// lane updating its whole row -- took 5.5 k: tools/ubench/choldiag.hip keeps both.)
// `acc` = the tile in the accumulator layout (what an MFMA update of it leaves in registers); D receives the factors.
// kInvOnly: D <- the TRANSPOSED inverse factor alone (upper triangle and diagonal L^-T, zeros below): every consumer that
// multiplies with L^-1 reads the tile as it is, without a select per operand (k_chol_solve_lds never reads L of a pivot tile)
template <bool kInvOnly = false>
__device__ __forceinline__ void cholDiag16Acc(d4_t acc, double* D, double* dinv, int laneIn, int* failFlag, long long* cyc = nullptr) {
  // opaque copy of the lane id: keeps the compiler from hoisting the per-lane masks of this routine out of the
  // caller's block-column loop
  int lane = laneIn;
  asm volatile("" : "+v"(lane));
  const int c = lane & 15, g = lane >> 4;
#ifdef SVIN_CHOL_TIMING
  const long long qd0 = __builtin_readcyclecounter();
#endif
  d4_t xacc;
#pragma unroll
  for (int r = 0; r < 4; ++r) xacc[r] = (g + 4 * r == c) ? 1.0 : 0.0;
  bool bad = false;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    // the 4 x 4 diagonal block (lower triangle), uniform: entry (4b + i, 4b + j) sits in lane 16 i + 4b + j
    const double b00 = readlaneD(acc[b], 4 * b);
    const double b10 = readlaneD(acc[b], 16 + 4 * b), b11 = readlaneD(acc[b], 16 + 4 * b + 1);
    const double b20 = readlaneD(acc[b], 32 + 4 * b), b21 = readlaneD(acc[b], 32 + 4 * b + 1), b22 = readlaneD(acc[b], 32 + 4 * b + 2);
    const double b30 = readlaneD(acc[b], 48 + 4 * b), b31 = readlaneD(acc[b], 48 + 4 * b + 1), b32 = readlaneD(acc[b], 48 + 4 * b + 2),
                 b33 = readlaneD(acc[b], 48 + 4 * b + 3);
    double P[4], PX[4];
    allGatherRows(acc[b], P);
    if (b == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) PX[q] = (c == q) ? 1.0 : 0.0;
    } else {
      allGatherRows(xacc[b], PX);
    }
    const double d0 = b00;
    bad = bad || !(d0 > 0);
    const double r0 = rcpPivot(d0 > 0 ? d0 : 1.0);  // branch-free: 1 for a failed pivot
    const double l10 = b10 * r0, l20 = b20 * r0, l30 = b30 * r0;
    const double d1 = __builtin_fma(-l10, b10, b11);
    bad = bad || !(d1 > 0);
    const double r1 = rcpPivot(d1 > 0 ? d1 : 1.0);
    const double u21 = __builtin_fma(-l20, b10, b21), u31 = __builtin_fma(-l30, b10, b31);
    const double l21 = u21 * r1, l31 = u31 * r1;
    const double d2 = __builtin_fma(-l21, u21, __builtin_fma(-l20, b20, b22));
    bad = bad || !(d2 > 0);
    const double r2 = rcpPivot(d2 > 0 ? d2 : 1.0);
    const double u32 = __builtin_fma(-l31, u21, __builtin_fma(-l30, b20, b32));
    const double l32 = u32 * r2;
    const double d3 = __builtin_fma(-l32, u32, __builtin_fma(-l31, u31, __builtin_fma(-l30, b30, b33)));
    bad = bad || !(d3 > 0);
    const double r3 = rcpPivot(d3 > 0 ? d3 : 1.0);
    // the four finished pivot rows at my column, and the same row operations on the inverse
    const double U0 = P[0];
    const double U1 = __builtin_fma(-l10, U0, P[1]);
    const double U2 = __builtin_fma(-l21, U1, __builtin_fma(-l20, U0, P[2]));
    const double U3 = __builtin_fma(-l32, U2, __builtin_fma(-l31, U1, __builtin_fma(-l30, U0, P[3])));
    const double X0 = PX[0];
    const double X1 = __builtin_fma(-l10, X0, PX[1]);
    const double X2 = __builtin_fma(-l21, X1, __builtin_fma(-l20, X0, PX[2]));
    const double X3 = __builtin_fma(-l32, X2, __builtin_fma(-l31, X1, __builtin_fma(-l30, X0, PX[3])));
    const double Um = selectByRow(g, U0, U1, U2, U3), Xm = selectByRow(g, X0, X1, X2, X3);
    const double rm = selectByRow(g, r0, r1, r2, r3), dm = selectByRow(g, d0, d1, d2, d3);
    if (b < 3) {  // rows > 4b+3: a_ij -= sum_k (U_k[i] / d_k) U_k[j]; rows and columns <= 4b+3 of acc are dead from here on
      const double aop = -Um * rm;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Um, acc, 0, 0, 0);
      xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Xm, xacc, 0, 0, 0);
    }
    const double rs = rsqrtNewton(dm > 0 ? dm : 1.0);  // 1/L_kk (1 for a failed pivot)
    const int k = 4 * b + g;
    if (kInvOnly) D[c * kPanelLd + k] = (c > k) ? 0.0 : Xm * rs;   // Linv[k][c] on and above the diagonal
    else D[c * kPanelLd + k] = ((c >= k) ? Um : Xm) * rs;  // L[c][k] below / on the diagonal, Linv[k][c] above
    if (c == 0) dinv[k] = rs;
  }
  if (bad && lane == 0) atomicOr(failFlag, 2);
#ifdef SVIN_CHOL_TIMING
  // (kept in LDS-free form: a global read-modify-write here cost the instrumented chain ~1.3 k cycles per tile)
  if (cyc) *cyc += __builtin_readcyclecounter() - qd0;
#endif
}

using lds_f64 = __attribute__((address_space(3))) double;
__device__ __forceinline__ lds_f64* tileToLds(double* p) { return (lds_f64*)p; }
__device__ __forceinline__ void tileLdsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- blocked Cholesky and inverse of the factor on an LDS image, NW waves, 16 x 16 tiles on v_mfma_f64_16x16x4 (marg.hip: k_marg_final_chol,
// k_marg_dense with 16 waves; kernels.hip: k_chol_border_prepare with 4).  A: NP x ld image (NP = 16 nT, ld = NP + 1), full symmetric, identity beyond the matrix; lower tiles <- L (off-diagonal
// tiles), the diagonal tiles' L and L^-1 go to scratch tiles DgGen (16 x kPanelLd each: L lower, L^-1 transposed strict upper) and
// dinvGen (1 / L_ii).  Three barriers per block column; *sFail gets a bit when a pivot is not positive (checked by the caller
// behind the last barrier).
__device__ __forceinline__ double tileLinvAt(const lds_f64* Dg, const lds_f64* dinv, int K, int row, int col) {
  const double off = Dg[K * 16 * kPanelLd + col * kPanelLd + row], dg = dinv[16 * K + row];
  return (col < row) ? off : ((col == row) ? dg : 0.0);
}
template <int NW>
__device__ __forceinline__ void tileCholFactor(lds_f64* A, int ld, int nT, double* DgGen, double* dinvGen, int* sFail) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, g = lane >> 4, c = lane & 15;
  const lds_f64* Dg = tileToLds(DgGen);
  const lds_f64* dinv = tileToLds(dinvGen);
  auto loadAcc = [&](int I, int J) {
    d4_t x;
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = A[(16 * I + g + 4 * r) * ld + 16 * J + c];
    return x;
  };
  auto storeAcc = [&](int I, int J, const d4_t& x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) A[(16 * I + g + 4 * r) * ld + 16 * J + c] = x[r];
  };
  for (int K = 0; K < nT; ++K) {
    if (wave == 0) cholDiag16Acc<false>(loadAcc(K, K), DgGen + K * 16 * kPanelLd, dinvGen + 16 * K, lane, sFail);
    __syncthreads();
    {
      for (int I = K + 1 + wave; I < nT; I += NW) {   // panel tile (I, K) <- A_IK L_KK^-T
        d4_t x = {0.0, 0.0, 0.0, 0.0};
        double av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          av[q] = A[(16 * I + c) * ld + 16 * K + 4 * q + g];
          bv[q] = tileLinvAt(Dg, dinv, K, c, 4 * q + g);   // B[k][j] = L^-1[j][k]
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], x, 0, 0, 0);
        storeAcc(I, K, x);
      }
    }
    tileLdsBarrier();
    const int m = nT - 1 - K;
    for (int id = wave; id < m * (m + 1) / 2; id += NW) {
      int r = 0;
      while ((r + 1) * (r + 2) / 2 <= id) ++r;
      const int I = K + 1 + r, J = K + 1 + (id - r * (r + 1) / 2);
      d4_t acc = loadAcc(I, J);
      double av[4], bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        av[q] = -A[(16 * I + c) * ld + 16 * K + 4 * q + g];
        bv[q] = A[(16 * J + c) * ld + 16 * K + 4 * q + g];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
      storeAcc(I, J, acc);
    }
    tileLdsBarrier();
  }
}
// Y = L^-1, block column J on wave J without a barrier: Y_JJ is the scratch tile's inverse, Y_IJ = -L_II^-1 sum_K L_IK Y_KJ with the
// running sum in the accumulator layout (which IS the B operand of the product with L_II^-1); Y_IJ^T goes to the upper tile (J, I), so
// element Y[i][j] of two different tile rows sits at A[j * ld + i].  Returns this lane's part of |Y|_F^2 over the n x n matrix.
template <int NW>
__device__ __forceinline__ double tileCholInverse(lds_f64* A, int ld, int nT, int n, const lds_f64* Dg, const lds_f64* dinv) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, g = lane >> 4, c = lane & 15;
  double fro = 0.0;
  for (int J = wave; J < nT; J += NW) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // the diagonal tile's own inverse: entries (row 4q + g, column c)
      const int row = 4 * q + g;
      const double v = tileLinvAt(Dg, dinv, J, row, c);
      if (16 * J + row < n && 16 * J + c < n) fro = __builtin_fma(v, v, fro);
    }
    for (int I = J + 1; I < nT; ++I) {
      d4_t sacc = {0.0, 0.0, 0.0, 0.0};
      for (int K = J; K < I; ++K) {
        double av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          av[q] = A[(16 * I + c) * ld + 16 * K + 4 * q + g];                                                           // L_IK[i = c][k]
          bv[q] = (K == J) ? tileLinvAt(Dg, dinv, J, 4 * q + g, c) : (double)A[(16 * J + c) * ld + 16 * K + 4 * q + g];   // Y_KJ[k][j = c]
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], sacc, 0, 0, 0);
      }
      d4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < 4; ++q) y = __builtin_amdgcn_mfma_f64_16x16x4f64(-tileLinvAt(Dg, dinv, I, c, 4 * q + g), sacc[q], y, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        A[(16 * J + c) * ld + 16 * I + g + 4 * r] = y[r];   // Y_IJ[i = g + 4r][j = c], transposed into the upper tile (J, I)
        fro = __builtin_fma(y[r], y[r], fro);               // (rows / columns of the padding are exactly zero here)
      }
    }
  }
  return fro;
}


}  // namespace svin
