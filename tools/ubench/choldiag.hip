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
template <int VARIANT>
__global__ __launch_bounds__(64) void bench(const double* A, double* out, double* dinvOut, int reps, long long* cyc) {
  __shared__ double T[16 * kLd], dv[16];
  const int lane = threadIdx.x;
  long long tot = 0;
  for (int r = 0; r < reps; ++r) {
    for (int e = lane; e < 256; e += 64) T[(e >> 4) * kLd + (e & 15)] = A[e];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (VARIANT == 0) cholOld(T, dv, lane); else cholNew(T, dv, lane);
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
  OK(hipMalloc(&dA, 256 * 8)); OK(hipMalloc(&dO, 2 * 256 * 8)); OK(hipMalloc(&dD, 2 * 16 * 8)); OK(hipMalloc(&dC, 16));
  OK(hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(bench<0>, dim3(1), dim3(64), 0, 0, dA, dO, dD, 200, dC);
  hipLaunchKernelGGL(bench<1>, dim3(1), dim3(64), 0, 0, dA, dO + 256, dD + 16, 200, dC + 1);
  OK(hipDeviceSynchronize());
  std::vector<double> O(512), Dv(32); long long c[2];
  OK(hipMemcpy(O.data(), dO, 512 * 8, hipMemcpyDeviceToHost)); OK(hipMemcpy(Dv.data(), dD, 32 * 8, hipMemcpyDeviceToHost)); OK(hipMemcpy(c, dC, 16, hipMemcpyDeviceToHost));
  double worst = 0, worstD = 0;
  for (int e = 0; e < 256; ++e) worst = fmax(worst, fabs(O[e] - O[256 + e]));
  for (int e = 0; e < 16; ++e) worstD = fmax(worstD, fabs(Dv[e] - Dv[16 + e]));
  // check L L^T = A for the new variant
  double res = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) s += O[256 + i * 16 + k] * O[256 + j * 16 + k]; res = fmax(res, fabs(s - A[i * 16 + j])); }
  printf("cycles per tile: one row per lane %lld, four lanes per row %lld; max |difference| tile %.3e dinv %.3e; |L L^T - A| %.3e\n", c[0], c[1], worst, worstD, res);
  return 0;
}
