// micro-benchmark: ONE wave running the chain of k_chol_solve_lds's backward substitution -- per step two dependent 16 x 16
// matrix-vector products (rv = r - L^T y, y' = M^T rv) -- (a) as two chains of four v_mfma_f64_16x16x4_f64 on a vector that is
// replicated over the columns, (b) on the VALU: four FMAs per lane, the sum over the four lane rows with v_permlane swaps, and
// the 'lane column -> lane row + 4 q' transposition of the vector with DPP (row_ror on three rows, then row_newbcast).
// Prints cycles per step and the largest difference between the two results.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dppMov(double old, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), kCtrl, kRowMask, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), kCtrl, kRowMask, 0xf, false);
  return __hiloint2double(hi, lo);
}
// v replicated over the lane rows (lane (g, c) holds v[c]) -> out[q] = v[4 q + g]
__device__ __forceinline__ void colToRowForm(double v, double (&out)[4]) {
  double u = v;                               // u(g, c) = v[(c + g) mod 16]: rotate row g left by g = right by 16 - g
  u = dppMov<0x120 + 15, 0x2>(u, v);
  u = dppMov<0x120 + 14, 0x4>(u, v);
  u = dppMov<0x120 + 13, 0x8>(u, v);
  out[0] = dppMov<0x150 + 0, 0xf>(0.0, u);    // row_newbcast: lane 4 q of every row to the whole row
  out[1] = dppMov<0x150 + 4, 0xf>(0.0, u);
  out[2] = dppMov<0x150 + 8, 0xf>(0.0, u);
  out[3] = dppMov<0x150 + 12, 0xf>(0.0, u);
}
__device__ __forceinline__ double sumLaneRows(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double s = __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);
  const unsigned slo = (unsigned)__double2loint(s), shi = (unsigned)__double2hiint(s);
  const auto l32 = __builtin_amdgcn_permlane32_swap(slo, slo, false, false);
  const auto h32 = __builtin_amdgcn_permlane32_swap(shi, shi, false, false);
  return __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);
}

__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int n, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  if (wave != 0) return;
  // L[k][i] (k = 4 q + g, i = c) and M[k][i]: small entries so that the iteration stays bounded
  double aL[4], aM[4];
  for (int q = 0; q < 4; ++q) {
    const int kk = 4 * q + g;
    aL[q] = 0.01 * sin(0.37 * kk + 1.3 * c);
    aM[q] = (kk == c) ? 0.9 : 0.02 * cos(0.11 * kk - 0.7 * c);
  }
  if (mode == 2) {   // layout check of colToRowForm
    double t[4];
    colToRowForm((double)c, t);
    for (int q = 0; q < 4; ++q) out[lane * 4 + q] = t[q];
    return;
  }
  long long t0 = __builtin_readcyclecounter();
  if (mode == 0) {
    d4_t y;   // row form: register r = y[g + 4 r]
    for (int r = 0; r < 4; ++r) y[r] = 1.0 + 0.1 * (g + 4 * r);
    for (int i = 0; i < n; ++i) {
      d4_t t1 = {0, 0, 0, 0};
      for (int q = 0; q < 4; ++q) t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aL[q], y[q], t1, 0, 0, 0);
      d4_t rv;
      for (int r = 0; r < 4; ++r) rv[r] = (1.0 + 0.1 * (g + 4 * r)) - t1[r];
      d4_t z = {0, 0, 0, 0};
      for (int q = 0; q < 4; ++q) z = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[q], rv[q], z, 0, 0, 0);
      y = z;
    }
    for (int r = 0; r < 4; ++r) if (c == 0) out[g + 4 * r] = y[r];
  } else {
    double y = 1.0 + 0.1 * c;   // column form, replicated over g
    for (int i = 0; i < n; ++i) {
      double yq[4];
      colToRowForm(y, yq);
      double p = 0;
      for (int q = 0; q < 4; ++q) p = __builtin_fma(aL[q], yq[q], p);
      const double rv = (1.0 + 0.1 * c) - sumLaneRows(p);
      double rq[4];
      colToRowForm(rv, rq);
      double p2 = 0;
      for (int q = 0; q < 4; ++q) p2 = __builtin_fma(aM[q], rq[q], p2);
      y = sumLaneRows(p2);
    }
    if (g == 0) out[c] = y;
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 64);
  double h[256]; long long hc;
  hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, 0, 2);
  hipMemcpy(h, out, 256 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) for (int q = 0; q < 4; ++q) if (h[lane * 4 + q] != 4 * q + (lane >> 4)) ++bad;
  printf("colToRowForm: %d wrong entries of 256 (lane 17: %g %g %g %g, lane 50: %g %g %g %g)\n", bad, h[68], h[69], h[70], h[71], h[200], h[201], h[202], h[203]);
  const int n = 2000;
  double res[2][16];
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, n, mode);
    hipDeviceSynchronize();
    hipMemcpy(res[mode], out, 16 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s: %.0f cycles per step (two dependent 16 x 16 matrix-vector products)\n", mode == 0 ? "mfma" : "valu + dpp", (double)hc / n);
  }
  double worst = 0;
  for (int i = 0; i < 16; ++i) worst = fmax(worst, fabs(res[0][i] - res[1][i]));
  printf("largest difference of the results: %.3e (y[0] = %.15g)\n", worst, res[0][0]);
  return 0;
}
