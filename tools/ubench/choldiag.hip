// micro-benchmark: the 16x16 diagonal-tile factorisation of the dense solvers (cholDiag16Reg in kernels.hip, one row
// per lane, pivot row through v_readlane) against a 4-lanes-per-row variant (lane = 16 g + i holds A[i][4g..4g+3],
// pivot row / multipliers through ds_bpermute).  Same contract: tile D (16 x 17 in LDS, full symmetric) <- L in the
// lower triangle, strict upper <- transposed strict lower of L^-1, dinv = 1 / L_ii.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kLd = 17;
__device__ __forceinline__ double rcpNewton(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double rsqrtNewton(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  double e = __builtin_fma(-h * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-h * y, y, 0.5);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double readlaneD(double v, int srcLane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srcLane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srcLane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bpermD(double v, int srcLane) {
  const int lo = __builtin_amdgcn_ds_bpermute(srcLane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(srcLane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// ---- reference: as in kernels.hip
__device__ __forceinline__ void cholOld(double* D, double* dinv, int lane) {
  const int li = lane & 15;
  double a[16], x[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { a[j] = D[li * kLd + j]; x[j] = (j == li) ? 1.0 : 0.0; }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    double pr[16];
#pragma unroll
    for (int j = k; j < 16; ++j) pr[j] = readlaneD(a[j], k);
    const bool ok = pr[k] > 0;
    const double rk = rcpNewton(ok ? pr[k] : 1.0);
    const double m = a[k] * rk, mx = x[k] * rk;
#pragma unroll
    for (int j = k + 1; j < 16; ++j) a[j] = __builtin_fma(-m, pr[j], a[j]);
#pragma unroll
    for (int j = k + 1; j < 16; ++j) x[j] = __builtin_fma(-mx, pr[j], x[j]);
    __builtin_amdgcn_sched_barrier(0);
  }
  double dk = a[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) dk = (li == k) ? a[k] : dk;
  const double rs = rsqrtNewton(dk > 0 ? dk : 1.0);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double rsk = readlaneD(rs, k);
    a[k] *= rsk;
    x[k] *= rsk;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) D[li * kLd + j] = (j > li) ? x[j] : a[j];
  dinv[li] = rs;
}
// ---- 4 lanes per row: lane = 16 g + i, a[c] = A[i][4g + c], x[c] = column 4g + c of the unit-lower inverse, row i...
// x follows the same recurrence as in the reference: lane (i, g) slot c carries X[i][4g + c] where X starts as I and
// receives x[j] -= (x[k] / a_kk) * pr[j] for j > k  (row operations identical to the ones applied to A)
__device__ __forceinline__ void cholNew(double* D, double* dinv, int lane) {
  const int i = lane & 15, g = lane >> 4;
  double a[4], x[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { a[c] = D[i * kLd + 4 * g + c]; x[c] = (4 * g + c == i) ? 1.0 : 0.0; }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    constexpr int dummy = 0; (void)dummy;
    const int gk = k >> 2, sk = k & 3;
    // pivot row entries of my column group: from lane (row k, group g)
    double pr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) pr[c] = bpermD(a[c], 16 * g + k);
    // a_kk (uniform) and my row's entries in column k (from lane (row i, group gk), slot sk)
    const double akk = readlaneD(a[sk], 16 * gk + k);
    const double aik = bpermD(a[sk], 16 * gk + i);
    const double xik = bpermD(x[sk], 16 * gk + i);
    const bool ok = akk > 0;
    const double rk = rcpNewton(ok ? akk : 1.0);
    const double m = aik * rk, mx = xik * rk;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool upd = 4 * g + c > k;   // columns j > k only (uniform per group and slot)
      a[c] = upd ? __builtin_fma(-m, pr[c], a[c]) : a[c];
      x[c] = upd ? __builtin_fma(-mx, pr[c], x[c]) : x[c];
    }
  }
  // 1/L_kk: d_k = a_kk after elimination, held by lane (k, k>>2) slot k&3
  double dself = 0;   // d_i for my own row i: from lane (i, i>>2) slot i&3
  {
    double v = a[0];
#pragma unroll
    for (int c = 1; c < 4; ++c) v = ((i & 3) == c) ? a[c] : v;
    dself = bpermD(v, 16 * (i >> 2) + i);
  }
  const double rsI = rsqrtNewton(dself > 0 ? dself : 1.0);   // 1 / L_ii (same in the 4 lanes of row i)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = 4 * g + c;
    const double rsj = bpermD(rsI, j);   // 1 / L_jj (lane j = row j, group 0)
    const double L = a[c] * rsj;         // L[i][j] for j <= i
    const double Li = x[c] * rsj;        // Linv[j][i] for j >= i
    D[i * kLd + j] = (j > i) ? Li : L;
  }
  if (g == 0) dinv[i] = rsI;
}

// ---- MFMA-blocked variant: the tile lives in the accumulator layout of v_mfma_f64_16x16x4 (lane = 16 g + c, register r
// = entry (row g + 4 r, column c)), so register b of the four lane rows IS the 4 x 16 row block of pivots 4b .. 4b+3.
// Per block: all-gather that register across the four lane rows (gfx950 v_permlane16/32_swap), read the 4 x 4 diagonal
// block with v_readlane (uniform), factor it redundantly in every lane, finish the four pivot rows locally, and apply
// the rank-4 trailing update -- to the matrix and to the running inverse -- with ONE MFMA each.
typedef double d4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gather4(double v, double (&out)[4]) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);   // [v0 v0 v2 v2], [v1 v1 v3 v3]
  const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const auto l02 = __builtin_amdgcn_permlane32_swap(l16[0], l16[0], false, false);   // [v0 x4], [v2 x4]
  const auto h02 = __builtin_amdgcn_permlane32_swap(h16[0], h16[0], false, false);
  const auto l13 = __builtin_amdgcn_permlane32_swap(l16[1], l16[1], false, false);   // [v1 x4], [v3 x4]
  const auto h13 = __builtin_amdgcn_permlane32_swap(h16[1], h16[1], false, false);
  out[0] = __hiloint2double((int)h02[0], (int)l02[0]);
  out[2] = __hiloint2double((int)h02[1], (int)l02[1]);
  out[1] = __hiloint2double((int)h13[0], (int)l13[0]);
  out[3] = __hiloint2double((int)h13[1], (int)l13[1]);
}
__device__ __forceinline__ double sel4(int g, double v0, double v1, double v2, double v3) {
  double v = v0;
  v = (g == 1) ? v1 : v;
  v = (g == 2) ? v2 : v;
  v = (g == 3) ? v3 : v;
  return v;
}
__device__ __forceinline__ void cholMfma(double* D, double* dinv, int lane, int* failFlag) {
  const int c = lane & 15, g = lane >> 4;
  d4_t acc, xacc;
#pragma unroll
  for (int r = 0; r < 4; ++r) { acc[r] = D[(g + 4 * r) * kLd + c]; xacc[r] = (g + 4 * r == c) ? 1.0 : 0.0; }
  bool bad = false;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    double P[4], PX[4];
    gather4(acc[b], P);
    if (b == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) PX[q] = (c == q) ? 1.0 : 0.0;
    } else {
      gather4(xacc[b], PX);
    }
    // the 4 x 4 diagonal block (lower triangle), uniform
    const double b00 = readlaneD(P[0], 4 * b);
    const double b10 = readlaneD(P[1], 4 * b), b11 = readlaneD(P[1], 4 * b + 1);
    const double b20 = readlaneD(P[2], 4 * b), b21 = readlaneD(P[2], 4 * b + 1), b22 = readlaneD(P[2], 4 * b + 2);
    const double b30 = readlaneD(P[3], 4 * b), b31 = readlaneD(P[3], 4 * b + 1), b32 = readlaneD(P[3], 4 * b + 2), b33 = readlaneD(P[3], 4 * b + 3);
    const double d0 = b00;
    bad = bad || !(d0 > 0);
    const double r0 = rcpNewton(d0 > 0 ? d0 : 1.0);
    const double l10 = b10 * r0, l20 = b20 * r0, l30 = b30 * r0;
    const double d1 = __builtin_fma(-l10, b10, b11);
    bad = bad || !(d1 > 0);
    const double r1 = rcpNewton(d1 > 0 ? d1 : 1.0);
    const double u21 = __builtin_fma(-l20, b10, b21), u31 = __builtin_fma(-l30, b10, b31);
    const double l21 = u21 * r1, l31 = u31 * r1;
    const double d2 = __builtin_fma(-l21, u21, __builtin_fma(-l20, b20, b22));
    bad = bad || !(d2 > 0);
    const double r2 = rcpNewton(d2 > 0 ? d2 : 1.0);
    const double u32 = __builtin_fma(-l31, u21, __builtin_fma(-l30, b20, b32));
    const double l32 = u32 * r2;
    const double d3 = __builtin_fma(-l32, u32, __builtin_fma(-l31, u31, __builtin_fma(-l30, b30, b33)));
    bad = bad || !(d3 > 0);
    const double r3 = rcpNewton(d3 > 0 ? d3 : 1.0);
    // the four pivot rows at my column, and the same row operations on the inverse
    const double U0 = P[0];
    const double U1 = __builtin_fma(-l10, U0, P[1]);
    const double U2 = __builtin_fma(-l21, U1, __builtin_fma(-l20, U0, P[2]));
    const double U3 = __builtin_fma(-l32, U2, __builtin_fma(-l31, U1, __builtin_fma(-l30, U0, P[3])));
    const double X0 = PX[0];
    const double X1 = __builtin_fma(-l10, X0, PX[1]);
    const double X2 = __builtin_fma(-l21, X1, __builtin_fma(-l20, X0, PX[2]));
    const double X3 = __builtin_fma(-l32, X2, __builtin_fma(-l31, X1, __builtin_fma(-l30, X0, PX[3])));
    const double Um = sel4(g, U0, U1, U2, U3), Xm = sel4(g, X0, X1, X2, X3);
    const double rm = sel4(g, r0, r1, r2, r3), dm = sel4(g, d0, d1, d2, d3);
    if (b < 3) {
      const double aop = -Um * rm;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Um, acc, 0, 0, 0);
      xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Xm, xacc, 0, 0, 0);
    }
    const double rs = rsqrtNewton(dm > 0 ? dm : 1.0);
    const int k = 4 * b + g;
    D[c * kLd + k] = ((c >= k) ? Um : Xm) * rs;
    if (c == 0) dinv[k] = rs;
  }
  if (bad && lane == 0) atomicOr(failFlag, 2);
}

// ---- variant 3: as cholMfma, but the four finished pivot rows come out of an MFMA as well: U = M P with M = Lt44^-1
// (the 4 x 4 unit-lower inverse, uniform values placed into the A operand of lanes c < 4) and P = the row block, which
// is the B operand as it stands (own accumulator register).  No all-gather, no per-lane recurrence, no select.
__device__ __forceinline__ double rcpPivot(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
__device__ __forceinline__ void cholMfma2(double* D, double* dinv, int lane, int* failFlag) {
  const int c = lane & 15, g = lane >> 4;
  d4_t acc, xacc;
#pragma unroll
  for (int r = 0; r < 4; ++r) { acc[r] = D[(g + 4 * r) * kLd + c]; xacc[r] = (g + 4 * r == c) ? 1.0 : 0.0; }
  const double mBase = (lane == 0 || lane == 17 || lane == 34 || lane == 51) ? 1.0 : 0.0;
  double dmin = 1.0, dlast = 1.0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const double b00 = readlaneD(acc[b], 4 * b);
    const double b10 = readlaneD(acc[b], 16 + 4 * b), b11 = readlaneD(acc[b], 16 + 4 * b + 1);
    const double b20 = readlaneD(acc[b], 32 + 4 * b), b21 = readlaneD(acc[b], 32 + 4 * b + 1), b22 = readlaneD(acc[b], 32 + 4 * b + 2);
    const double b30 = readlaneD(acc[b], 48 + 4 * b), b31 = readlaneD(acc[b], 48 + 4 * b + 1), b32 = readlaneD(acc[b], 48 + 4 * b + 2),
                 b33 = readlaneD(acc[b], 48 + 4 * b + 3);
    const double d0 = b00;
    const double r0 = rcpPivot(d0);
    const double l10 = b10 * r0, l20 = b20 * r0, l30 = b30 * r0;
    const double d1 = __builtin_fma(-l10, b10, b11);
    const double r1 = rcpPivot(d1);
    const double u21 = __builtin_fma(-l20, b10, b21), u31 = __builtin_fma(-l30, b10, b31);
    const double l21 = u21 * r1, l31 = u31 * r1;
    const double d2 = __builtin_fma(-l21, u21, __builtin_fma(-l20, b20, b22));
    const double r2 = rcpPivot(d2);
    const double u32 = __builtin_fma(-l31, u21, __builtin_fma(-l30, b20, b32));
    const double l32 = u32 * r2;
    const double d3 = __builtin_fma(-l32, u32, __builtin_fma(-l31, u31, __builtin_fma(-l30, b30, b33)));
    const double r3 = rcpPivot(d3);
    dmin = fmin(fmin(dmin, d0), fmin(d1, fmin(d2, d3)));
    dlast = d3;
    // M = Lt44^-1
    const double m20 = __builtin_fma(l21, l10, -l20), m31 = __builtin_fma(l32, l21, -l31);
    const double m30 = __builtin_fma(-l32, m20, __builtin_fma(l31, l10, -l30));
    double mop = mBase;
    mop = (lane == 1) ? -l10 : mop;
    mop = (lane == 2) ? m20 : mop;
    mop = (lane == 18) ? -l21 : mop;
    mop = (lane == 3) ? m30 : mop;
    mop = (lane == 19) ? m31 : mop;
    mop = (lane == 35) ? -l32 : mop;
    const d4_t zero = {0, 0, 0, 0};
    const d4_t Ur = __builtin_amdgcn_mfma_f64_16x16x4f64(mop, acc[b], zero, 0, 0, 0);    // reg 0: U_g[c]
    const d4_t Xr = __builtin_amdgcn_mfma_f64_16x16x4f64(mop, xacc[b], zero, 0, 0, 0);
    const double Um = Ur[0], Xm = Xr[0];
    const double rm = sel4(g, r0, r1, r2, r3);
    if (b < 3) {
      const double aop = -Um * rm;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Um, acc, 0, 0, 0);
      xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Xm, xacc, 0, 0, 0);
    }
    const double rs = rm * rsqrtNewton(rm);   // sqrt(1/d) = 1/L_kk
    const int k = 4 * b + g;
    D[c * kLd + k] = ((c >= k) ? Um : Xm) * rs;
    if (c == 0) dinv[k] = rs;
  }
  if ((!(dmin > 0) || !(dlast > 0)) && lane == 0) atomicOr(failFlag, 2);
}

// ---- variant 4: variant 3 with the dependent-operation chain of the 4 x 4 block cut down.  A dependent f64 operation
// costs ~20 cycles of latency and the block factorisation is one long chain, so: fraction-free (Bareiss) elimination --
// p1 = d0 d1, p2 = d0 d1 d2, p3 = d0 d1 d2 d3 come out of mul/fma pairs without waiting for any reciprocal, the
// reciprocals run beside the chain -- and sums of disjoint selects instead of select chains.
__device__ __forceinline__ double rcpFast(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
__device__ __forceinline__ double pick4(int g, double v0, double v1, double v2, double v3) {
  return ((g == 0 ? v0 : 0.0) + (g == 1 ? v1 : 0.0)) + ((g == 2 ? v2 : 0.0) + (g == 3 ? v3 : 0.0));
}
__device__ __forceinline__ void cholMfma3(double* D, double* dinv, int lane, int* failFlag) {
  const int c = lane & 15, g = lane >> 4;
  d4_t acc, xacc;
#pragma unroll
  for (int r = 0; r < 4; ++r) { acc[r] = D[(g + 4 * r) * kLd + c]; xacc[r] = (g + 4 * r == c) ? 1.0 : 0.0; }
  const double mBase = (lane == 0 || lane == 17 || lane == 34 || lane == 51) ? 1.0 : 0.0;
  double pmin = 1.0, plast = 1.0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const double b00 = readlaneD(acc[b], 4 * b);
    const double b10 = readlaneD(acc[b], 16 + 4 * b), b11 = readlaneD(acc[b], 16 + 4 * b + 1);
    const double b20 = readlaneD(acc[b], 32 + 4 * b), b21 = readlaneD(acc[b], 32 + 4 * b + 1), b22 = readlaneD(acc[b], 32 + 4 * b + 2);
    const double b30 = readlaneD(acc[b], 48 + 4 * b), b31 = readlaneD(acc[b], 48 + 4 * b + 1), b32 = readlaneD(acc[b], 48 + 4 * b + 2),
                 b33 = readlaneD(acc[b], 48 + 4 * b + 3);
    const double r0 = rcpFast(b00);
    // step 0 (scaled by d0)
    const double c11 = __builtin_fma(-b10, b10, b11 * b00), c21 = __builtin_fma(-b20, b10, b21 * b00), c31 = __builtin_fma(-b30, b10, b31 * b00);
    const double c22 = __builtin_fma(-b20, b20, b22 * b00), c32 = __builtin_fma(-b30, b20, b32 * b00), c33 = __builtin_fma(-b30, b30, b33 * b00);
    const double rp1 = rcpFast(c11);
    // step 1 (scaled by d0 d1)
    const double e22 = __builtin_fma(-c21, c21, c22 * c11) * r0, e32 = __builtin_fma(-c31, c21, c32 * c11) * r0, e33 = __builtin_fma(-c31, c31, c33 * c11) * r0;
    const double rp2 = rcpFast(e22);
    // step 2 (scaled by d0 d1 d2)
    const double f33 = __builtin_fma(-e32, e32, e33 * e22) * rp1;
    const double rp3 = rcpFast(f33);
    pmin = fmin(fmin(pmin, b00), fmin(c11, fmin(e22, f33)));
    plast = f33;
    const double r1 = b00 * rp1, r2 = c11 * rp2, r3 = e22 * rp3;   // 1 / d_k
    const double l10 = b10 * r0, l20 = b20 * r0, l30 = b30 * r0, l21 = c21 * rp1, l31 = c31 * rp1, l32 = e32 * rp2;
    const double m20 = __builtin_fma(l21, l10, -l20), m31 = __builtin_fma(l32, l21, -l31);
    const double m30 = __builtin_fma(-l32, m20, __builtin_fma(l31, l10, -l30));
    const double mop = ((mBase + (lane == 1 ? -l10 : 0.0)) + ((lane == 2 ? m20 : 0.0) + (lane == 18 ? -l21 : 0.0))) +
                       (((lane == 3 ? m30 : 0.0) + (lane == 19 ? m31 : 0.0)) + (lane == 35 ? -l32 : 0.0));
    const d4_t zero = {0, 0, 0, 0};
    const d4_t Ur = __builtin_amdgcn_mfma_f64_16x16x4f64(mop, acc[b], zero, 0, 0, 0);    // reg 0: U_g[c]
    const d4_t Xr = __builtin_amdgcn_mfma_f64_16x16x4f64(mop, xacc[b], zero, 0, 0, 0);
    const double Um = Ur[0], Xm = Xr[0];
    const double rm = pick4(g, r0, r1, r2, r3);
    if (b < 3) {
      const double aop = -Um * rm;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Um, acc, 0, 0, 0);
      xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Xm, xacc, 0, 0, 0);
    }
    const double rs = rm * rsqrtNewton(rm);   // sqrt(1/d) = 1/L_kk
    const int k = 4 * b + g;
    D[c * kLd + k] = ((c >= k) ? Um : Xm) * rs;
    if (c == 0) dinv[k] = rs;
  }
  if ((!(pmin > 0) || !(plast > 0)) && lane == 0) atomicOr(failFlag, 2);
}
template <int VARIANT>
__global__ __launch_bounds__(64) void bench(const double* A, double* out, double* dinvOut, int reps, long long* cyc) {
  __shared__ double T[16 * kLd], dv[16];
  const int lane = threadIdx.x;
  long long tot = 0;
  for (int r = 0; r < reps; ++r) {
    for (int e = lane; e < 256; e += 64) T[(e >> 4) * kLd + (e & 15)] = A[e];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (VARIANT == 0) cholOld(T, dv, lane); else if (VARIANT == 1) cholNew(T, dv, lane); else if (VARIANT == 2) cholMfma(T, dv, lane, (int*)(cyc + 8)); else if (VARIANT == 3) cholMfma2(T, dv, lane, (int*)(cyc + 8)); else cholMfma3(T, dv, lane, (int*)(cyc + 8));
    __syncthreads();
    tot += __builtin_readcyclecounter() - t0;
  }
  for (int e = lane; e < 256; e += 64) out[e] = T[(e >> 4) * kLd + (e & 15)];
  if (lane < 16) dinvOut[lane] = dv[lane];
  if (lane == 0) cyc[0] = tot / reps;
}
int main() {
  std::vector<double> A(256), B(256);
  srand(3);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) B[i * 16 + j] = (rand() % 2001 - 1000) / 1000.0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? 4.0 : 0.0; for (int k = 0; k < 16; ++k) s += B[i * 16 + k] * B[j * 16 + k]; A[i * 16 + j] = s; }
  double *dA, *dO, *dD; long long* dC;
  OK(hipMalloc(&dA, 256 * 8)); OK(hipMalloc(&dO, 5 * 256 * 8)); OK(hipMalloc(&dD, 5 * 16 * 8)); OK(hipMalloc(&dC, 128));
  OK(hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice));
  OK(hipMemset(dC, 0, 128));
  hipLaunchKernelGGL(bench<0>, dim3(1), dim3(64), 0, 0, dA, dO, dD, 200, dC);
  hipLaunchKernelGGL(bench<1>, dim3(1), dim3(64), 0, 0, dA, dO + 256, dD + 16, 200, dC + 1);
  hipLaunchKernelGGL(bench<2>, dim3(1), dim3(64), 0, 0, dA, dO + 512, dD + 32, 200, dC + 2);
  hipLaunchKernelGGL(bench<3>, dim3(1), dim3(64), 0, 0, dA, dO + 768, dD + 48, 200, dC + 3);
  hipLaunchKernelGGL(bench<4>, dim3(1), dim3(64), 0, 0, dA, dO + 1024, dD + 64, 200, dC + 4);
  OK(hipDeviceSynchronize());
  std::vector<double> O(1280), Dv(80); long long c[5];
  OK(hipMemcpy(O.data(), dO, 1280 * 8, hipMemcpyDeviceToHost)); OK(hipMemcpy(Dv.data(), dD, 80 * 8, hipMemcpyDeviceToHost)); OK(hipMemcpy(c, dC, 40, hipMemcpyDeviceToHost));
  double worst = 0, worstD = 0;
  for (int e = 0; e < 256; ++e) worst = fmax(worst, fabs(O[e] - O[256 + e]));
  for (int e = 0; e < 16; ++e) worstD = fmax(worstD, fabs(Dv[e] - Dv[16 + e]));
  // check L L^T = A for the new variant
  double res = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) s += O[256 + i * 16 + k] * O[256 + j * 16 + k]; res = fmax(res, fabs(s - A[i * 16 + j])); }
  printf("cycles per tile: one row per lane %lld, four lanes per row %lld; max |difference| tile %.3e dinv %.3e; |L L^T - A| %.3e\n", c[0], c[1], worst, worstD, res);
  double w2 = 0, w2d = 0;
  for (int e = 0; e < 256; ++e) w2 = fmax(w2, fabs(O[e] - O[512 + e]));
  for (int e = 0; e < 16; ++e) w2d = fmax(w2d, fabs(Dv[e] - Dv[32 + e]));
  printf("MFMA-blocked: %lld cycles per tile; max |difference to one-row-per-lane| tile %.3e dinv %.3e\n", c[2], w2, w2d);
  double w3 = 0, w3d = 0;
  for (int e = 0; e < 256; ++e) w3 = fmax(w3, fabs(O[e] - O[768 + e]));
  for (int e = 0; e < 16; ++e) w3d = fmax(w3d, fabs(Dv[e] - Dv[48 + e]));
  printf("MFMA-blocked, pivot rows through MFMA: %lld cycles per tile; max |difference to one-row-per-lane| tile %.3e dinv %.3e\n", c[3], w3, w3d);
  double w4 = 0, w4d = 0;
  for (int e = 0; e < 256; ++e) w4 = fmax(w4, fabs(O[e] - O[1024 + e]));
  for (int e = 0; e < 16; ++e) w4d = fmax(w4d, fabs(Dv[e] - Dv[64 + e]));
  printf("MFMA-blocked, fraction-free 4x4 block: %lld cycles per tile; max |difference to one-row-per-lane| tile %.3e dinv %.3e\n", c[4], w4, w4d);
  return 0;
}
