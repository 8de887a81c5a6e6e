// micro-benchmark: latency of dependent / independent v_mfma_f64_16x16x4_f64 and f64 FMA chains on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ void k(double* out, long long* cyc, int n) {
  const int lane = threadIdx.x;
  double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
  d4_t acc = {0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  long long t1 = __builtin_readcyclecounter();
  // dependent through the B operand (result feeds the next op)
  d4_t v = acc;
  for (int i = 0; i < n; ++i) { d4_t z = {0, 0, 0, 0}; v = __builtin_amdgcn_mfma_f64_16x16x4f64(a, v[0] * 1e-30, z, 0, 0, 0); }
  long long t2 = __builtin_readcyclecounter();
  double x = a;
  for (int i = 0; i < n; ++i) x = __builtin_fma(x, b, a);
  long long t3 = __builtin_readcyclecounter();
  d4_t a0 = {0,0,0,0}, a1 = {0,0,0,0};
  for (int i = 0; i < n; ++i) { a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, a1, 0, 0, 0); }
  long long t4 = __builtin_readcyclecounter();
  double y[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = __builtin_fma(y[j], b, a);
  long long t5 = __builtin_readcyclecounter();
  double w = a;
  for (int i = 0; i < n; ++i) w = w + b;
  long long t6 = __builtin_readcyclecounter();
  float fx = (float)a, fb2 = (float)b;
  for (int i = 0; i < n; ++i) fx = __builtin_fmaf(fx, fb2, fb2);
  long long t7 = __builtin_readcyclecounter();
  double y2[2] = {a, b};
  for (int i = 0; i < n; ++i) { y2[0] = __builtin_fma(y2[0], b, a); y2[1] = __builtin_fma(y2[1], b, a); }
  long long t8 = __builtin_readcyclecounter();
  x += y[0] + y[1] + y[2] + y[3] + y[4] + y[5] + y[6] + y[7] + w + fx + y2[0] + y2[1];
  if (lane == 0) { cyc[4] = t5 - t4; cyc[5] = t6 - t5; cyc[6] = t7 - t6; cyc[7] = t8 - t7; }
  out[lane] = acc[0] + v[1] + x + a0[2] + a1[3];
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64);
  const int n = 4096;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("8 indep f64 fma: %.1f cyc per 8 | f64 add dependent %.1f | f32 fma dependent %.1f | 2 indep f64 fma %.1f per 2\n", (double)h[4] / n, (double)h[5] / n, (double)h[6] / n, (double)h[7] / n);
    printf("n=%d  mfma acc-chain %.1f cyc/op | mfma B-dependent %.1f | fma f64 dependent %.1f | 2 indep mfma chains %.1f cyc/pair | kernel %.3f ms, total cycles %lld -> %.2f GHz\n",
           n, (double)h[0] / n, (double)h[1] / n, (double)h[2] / n, (double)h[3] / n, ms, h[0] + h[1] + h[2] + h[3], (h[0] + h[1] + h[2] + h[3]) / (ms * 1e6));
  }
  return 0;
}
